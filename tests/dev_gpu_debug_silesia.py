"""Developer tool: stage-by-stage diff against the oracle on a slice of the Silesia-style corpus."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import indexed_bzip2_amd as m
from oracle import oracle as O
import silesia_like, bz2build
from gpu_debug import first_diff

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
    data = silesia_like.generate(n)
    enc, nb, offs = bz2build.build(data, 1, piece_size=9_000_000, threads=16, find_magic=m.find_magic)
    dec = m.Decoder(flags=m.Decoder.KEEP_STAGES)
    dec.set_input(enc)
    res, total = dec.decode_batch(offs)
    out = dec.copy_output(0, total)
    bad = 0
    for i, (o, r) in enumerate(zip(offs, res)):
        od, payload, lcol, rle = O.decode_block(enc, o, want_stages=True)
        diffs = {k: (r[k], od[k]) for k in od if k in r and r[k] != od[k]}
        msg = []
        if diffs: msg.append(f"fields {diffs}")
        gl = dec.debug_stage(i, 0); d = first_diff(gl, lcol)
        if d >= 0: msg.append(f"L differs at {d}/{len(lcol)} gpu={gl[d:d+8].hex()} ref={lcol[d:d+8].hex()}")
        if msg:
            bad += 1
            if bad <= 6: print(f"[FAIL] block {i} @{o}: " + "; ".join(msg), flush=True)
    print(f"{len(offs)} blocks, {bad} bad")

if __name__ == "__main__":
    main()
