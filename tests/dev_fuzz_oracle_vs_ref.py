"""Developer fuzz (authoring container only: needs oracle/_ref/ref_bz2, the real reference compiled from its own
sources): damaged inputs, the oracle's status per block against the exception the reference throws for the same block.
Usage: python tests/dev_fuzz_oracle_vs_ref.py [cases] [seed]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from oracle import oracle as O
from test_oracle import expected_status

REF = os.path.join(ROOT, "oracle", "_ref", "ref_bz2")


def probe(path, off):
    out = subprocess.run([REF, "probe", path, str(off)], capture_output=True, text=True, timeout=120).stdout.strip()
    parts = out.split(" ", 2)
    if parts[0] == "OK":
        f = out.split()
        return {"verdict": "OK", "size": int(f[2]), "header_crc": int(f[3], 16), "calc_crc": int(f[4], 16), "decoded": int(f[5])}
    return {"verdict": "EXC", "exception": parts[1], "what": parts[2] if len(parts) > 2 else ""}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    sources = [datagen.text_like(120_000, 91), datagen.random_bytes(90_000, 92),
               bytes(rng.integers(0, 5, 150_000, dtype=np.uint8)), b"abc" * 30_000]
    checked = 0
    crc_checked = [0]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "x.bz2")
        for case in range(cases):
            raw = sources[case % len(sources)]
            enc = bytearray(datagen.compress(raw, 1 if case % 3 else 9))
            offs = O.find_magic(bytes(enc))
            mode = int(rng.integers(0, 4))
            if mode == 0:
                o = offs[int(rng.integers(0, len(offs)))] // 8
                p = min(len(enc) - 1, o + int(rng.integers(6, 400)))
                enc[p] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                enc[int(rng.integers(4, len(enc)))] ^= 1 << int(rng.integers(0, 8))
            elif mode == 2:
                p = int(rng.integers(4, max(5, len(enc) - 8)))
                enc[p:p + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
            else:
                enc = enc[:int(rng.integers(5, len(enc)))]
            enc = bytes(enc)
            with open(path, "wb") as f:
                f.write(enc)
            for off in offs:
                if off + 48 > len(enc) * 8:
                    continue
                d, payload = O.decode_block(enc, off)
                ref = probe(path, off)
                want = expected_status(ref)
                if d["status"] != want:
                    print(f"MISMATCH case {case} mode {mode} off {off}: oracle {d['status']} reference {ref}")
                    sys.exit(1)
                if want == 15:
                    # "Calculated CRC <hex> for block mismatches <hex>" (bzip2.hpp:900-907): the bytes of a damaged
                    # block must be the reference's too
                    words = ref["what"].split()
                    assert (int(words[2], 16), int(words[-1], 16)) == (d["computed_crc"], d["header_crc"]), (case, off, d, ref)
                    crc_checked[0] += 1
                if want == 0 and not d["is_eos"]:
                    assert (d["decoded_size"], d["header_crc"], d["computed_crc"], d["encoded_size_bits"]) == \
                           (ref["decoded"], ref["header_crc"], ref["calc_crc"], ref["size"]), (case, off, d, ref)
                checked += 1
    print(f"{cases} damaged files, {checked} blocks ({crc_checked[0]} with a CRC mismatch whose calculated CRC was compared): "
          f"oracle == reference (seed {seed})")


if __name__ == "__main__":
    main()
