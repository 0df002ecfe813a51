"""Stage-by-stage diff of the HIP path against the oracle (developer tool, run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import indexed_bzip2_amd as m
from oracle import oracle as O
import datagen


def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return n if len(a) != len(b) else -1


def run(name, enc, dec):
    offs = O.find_magic(enc)
    dec.set_input(enc)
    res, total = dec.decode_batch(offs)
    out = dec.copy_output(0, total)
    ok = True
    for i, (o, r) in enumerate(zip(offs, res)):
        od, payload, lcol, rle = O.decode_block(enc, o, want_stages=True)
        diffs = {k: (r[k], od[k]) for k in od if k in r and r[k] != od[k]}
        msg = []
        if diffs:
            msg.append(f"fields {diffs}")
        if od["status"] == 0:
            gl = dec.debug_stage(i, 0)
            d = first_diff(gl, lcol)
            if d >= 0:
                msg.append(f"L differs at {d}/{len(lcol)} gpu={gl[d:d+8].hex()} ref={lcol[d:d+8].hex()}")
            gr = dec.debug_stage(i, 2)
            d = first_diff(gr, rle)
            if d >= 0:
                msg.append(f"R differs at {d}/{len(rle)} gpu={gr[d:d+8].hex()} ref={rle[d:d+8].hex()}")
            got = out[r["data_offset"]:r["data_offset"] + r["decoded_size"]]
            d = first_diff(got, payload)
            if d >= 0:
                msg.append(f"OUT differs at {d}/{len(payload)} gpu={got[d:d+8].hex()} ref={payload[d:d+8].hex()}")
        if msg:
            ok = False
            print(f"[FAIL] {name} block {i} @{o}: " + "; ".join(msg))
    print(f"[{'ok' if ok else 'FAIL'}] {name}: {len(offs)} blocks, {total} bytes, timings {dec.timings()}")
    return ok


def main():
    dec = m.Decoder(flags=m.Decoder.KEEP_STAGES)
    fx = os.path.join(ROOT, "tests", "golden", "fixtures")
    allok = True
    for f in sorted(os.listdir(fx)):
        if f.endswith(".bz2"):
            allok &= run(f, open(os.path.join(fx, f), "rb").read(), dec)
    for name, (raw, level) in sorted(datagen.corpus_small().items()):
        allok &= run(name, datagen.compress(raw, level), dec)
    print("ALL OK" if allok else "SOME FAILED")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
