"""Developer fuzz (GPU box): random seek / read / tell sequences through the reader against the raw bytes, with and
without an imported block map."""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
import indexed_bzip2_amd as m


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rng = np.random.default_rng(seed)
    parts = [datagen.text_like(700_000, 111), datagen.random_bytes(250_000, 112), b"", b"q", datagen.runs(400_000, 113)]
    raw = b"".join(parts)
    enc = datagen.multistream(parts, 1)
    index = None
    ops = 0
    for round_ in range(8):
        P = int(rng.choice([1, 2, 3, 8, 0]))
        f = m.open(io.BytesIO(enc), P)
        if index is not None and round_ % 2:
            f.set_block_offsets(index)
        pos = 0
        for _ in range(120):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                target = int(rng.integers(0, len(raw) + 1000))
                got = f.seek(target)
                pos = min(target, len(raw))
                assert got == pos, (P, target, got, pos)
            elif kind == 1:
                delta = int(rng.integers(-200_000, 200_000))
                want = min(max(0, pos + delta), len(raw))
                if pos + delta < 0:
                    continue
                got = f.seek(delta, io.SEEK_CUR)
                pos = want
                assert got == pos, (P, delta, got, pos)
            elif kind == 2:
                back = int(rng.integers(0, len(raw)))
                got = f.seek(-back, io.SEEK_END)
                pos = len(raw) - back
                assert got == pos
            n = int(rng.choice([0, 1, 7, 1000, 65536, 300_000]))
            data = f.read(n)
            assert data == raw[pos:pos + n], (P, pos, n, len(data))
            pos += len(data)
            assert f.tell() == pos
            ops += 1
        if index is None:
            index = f.block_offsets()
        f.close()
    print(f"{ops} random seek/read operations equal to the raw data (seed {seed})")


if __name__ == "__main__":
    main()
