"""Developer probe: decode a multi-stream file of N MB through decode_batch and print stage timings."""
import bz2
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
import indexed_bzip2_amd as m


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    kind = sys.argv[2] if len(sys.argv) > 2 else "text"
    piece = 9_000_000
    n = max(1, mb * 1_000_000 // piece)
    t0 = time.time()
    base = datagen.text_like(piece, 7) if kind == "text" else datagen.random_bytes(piece, 7)
    def comp(i):
        salt = bytes([i & 255, (i >> 8) & 255]) * 8
        return bz2.compress(salt + base[16:], 9)
    with ThreadPoolExecutor(16) as ex:
        streams = list(ex.map(comp, range(n)))
    enc = b"".join(streams)
    print(f"generated {n * piece / 1e6:.0f} MB -> {len(enc) / 1e6:.1f} MB compressed in {time.time() - t0:.1f}s", flush=True)
    offs = m.find_magic(enc)
    print(len(offs), "blocks", flush=True)
    dec = m.Decoder()
    dec.set_input(enc)
    for it in range(3):
        t0 = time.time()
        res, total = dec.decode_batch(offs)
        dt = time.time() - t0
        bad = [r for r in res if r["status"] != 0]
        print(f"iter {it}: {total / 1e6:.1f} MB in {dt * 1e3:.1f} ms = {total / dt / 1e6:.0f} MB/s, bad={len(bad)}, timings={dec.timings()}", flush=True)


if __name__ == "__main__":
    main()
