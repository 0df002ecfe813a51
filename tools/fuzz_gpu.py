"""Developer fuzz run: seeded random inputs (sizes, alphabets, runs, periods, levels) -> bz2 -> GPU decode == input.
Usage: python tools/fuzz_gpu.py [cases] [seed]"""
import bz2
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import indexed_bzip2_amd as m


def make(rng):
    n = int(rng.integers(1, 2_500_000))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        k = int(rng.integers(1, 257))
        return rng.integers(0, k, n, dtype=np.uint8).tobytes()
    if kind == 1:
        lengths = rng.geometric(float(rng.uniform(0.002, 0.3)), size=n // 4 + 1)
        values = rng.integers(0, int(rng.integers(1, 257)), lengths.size, dtype=np.uint8)
        return np.repeat(values, lengths)[:n].tobytes()
    if kind == 2:
        period = int(rng.integers(1, 5000))
        unit = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
        return (unit * (n // period + 1))[:n]
    if kind == 3:
        k = int(rng.integers(2, 200))
        w = 1.0 / np.arange(1, k + 1) ** float(rng.uniform(0.5, 2.5))
        return rng.choice(k, size=n, p=w / w.sum()).astype(np.uint8).tobytes()
    if kind == 4:   # sorted / nearly sorted
        a = np.sort(rng.integers(0, 256, n, dtype=np.uint8))
        return a.tobytes()
    parts = []
    while sum(len(p) for p in parts) < n:
        parts.append(make_small(rng))
    return b"".join(parts)[:n]


def make_small(rng):
    k = int(rng.integers(1, 257))
    return rng.integers(0, k, int(rng.integers(1, 200_000)), dtype=np.uint8).tobytes() * int(rng.integers(1, 4))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dec = m.Decoder()
    total = 0
    for case in range(cases):
        raw = make(rng)
        level = int(rng.integers(1, 10))
        enc = bz2.compress(raw, level)
        if case % 7 == 0:   # multi-stream
            extra = make_small(rng)
            enc += bz2.compress(extra, int(rng.integers(1, 10)))
            raw += extra
        offs = m.find_magic(enc)
        dec.set_input(enc)
        res, n = dec.decode_batch(offs)
        bad = [r for r in res if r["status"] != 0]
        out = dec.copy_output(0, n)
        if bad or out != raw:
            print(f"MISMATCH case {case} seed {seed}: len {len(raw)} level {level} blocks {len(offs)} bad {bad[:1]}", flush=True)
            sys.exit(1)
        total += len(raw)
    print(f"{cases} cases, {total / 1e6:.0f} MB decoded, all equal (seed {seed})")


if __name__ == "__main__":
    main()
