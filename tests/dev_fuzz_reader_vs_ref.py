"""Developer fuzz (GPU box; needs oracle/_ref/ref_bz2): damaged multi-block / multi-stream files through the READER --
the reference's ParallelBZ2Reader (ref_bz2 decode <file> <P> -) against indexed_bzip2_amd's reader: decoded size when
both succeed, otherwise the reference's exception against our status.
Usage: python tests/dev_fuzz_reader_vs_ref.py [cases] [seed]"""
import io
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
import indexed_bzip2_amd as m
from test_oracle import expected_status

REF = os.path.join(ROOT, "oracle", "_ref", "ref_bz2")


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    parts = [datagen.text_like(330_000, 101), datagen.random_bytes(120_000, 102), b"", datagen.runs(150_000, 103)]
    base = datagen.multistream(parts, 1)
    stats = {"equal_ok": 0, "equal_error": 0}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "x.bz2")
        for case in range(cases):
            enc = bytearray(base)
            mode = case % 5
            if mode == 0:
                enc[int(rng.integers(4, len(enc)))] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                p = int(rng.integers(4, len(enc) - 8))
                enc[p:p + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
            elif mode == 2:
                enc = enc[:int(rng.integers(40, len(enc)))]
            elif mode == 3:
                enc += rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8).tobytes()   # trailing garbage
            else:
                pass   # undamaged
            with open(path, "wb") as f:
                f.write(bytes(enc))
            P = int(rng.integers(2, 5))
            out = subprocess.run([REF, "decode", path, str(P), "-"], capture_output=True, text=True, timeout=300).stdout.strip()
            out = out.splitlines()[-1] if out else ""
            try:
                with m.IndexedBzip2FileRaw(path, P) as f:
                    mine = ("OK", len(f.readall()))
            except m.Bz2Error as e:
                mine = ("EXC", e.status)
            except ValueError as e:
                mine = ("EXC", str(e))
            if out.startswith("EXC"):
                parts_ = out.split(" ", 2)
                want = expected_status({"verdict": "EXC", "exception": parts_[1], "what": parts_[2] if len(parts_) > 2 else ""})
                if mine != ("EXC", want):
                    print(f"MISMATCH case {case} mode {mode} P {P}: reference {out[:160]!r} -> status {want}, ours {mine}")
                    sys.exit(1)
                stats["equal_error"] += 1
            else:
                if mine != ("OK", int(out)):
                    print(f"MISMATCH case {case} mode {mode} P {P}: reference decoded {out}, ours {mine}")
                    sys.exit(1)
                stats["equal_ok"] += 1
    print(f"{cases} files through the readers: {stats} (seed {seed})")


if __name__ == "__main__":
    main()
