"""Developer probe: latency of tiny batches (1, 2, 8 blocks) through the batch API, with the kernel timeline of the device
trace (MI355X_BZ2_TRACE=1)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401
import bench
import indexed_bzip2_amd as m


def main():
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, "/tmp/indexed_bzip2_amd_bench", 0, 1, lambda: None)
    offsets = meta["offsets"]
    sizes = [(b - a, i) for i, (a, b) in enumerate(zip(offsets, offsets[1:]))]
    small = min(sizes)[1]
    large = max(sizes)[1]
    median = sorted(sizes)[len(sizes) // 2][1]
    dec = m.Decoder(device=0, max_batch_blocks=64)
    dec.set_input(enc)
    for name, idx in (("median block", [median]), ("smallest block", [small]), ("largest block", [large]),
                      ("8 consecutive blocks", list(range(100, 108)))):
        offs = [offsets[i] for i in idx]
        a, r = dec.make_arrays(offs)
        dec.decode_batch_into(a, len(offs), r)
        best = 1e9
        for rep in range(5):
            os.environ["MI355X_BZ2_TRACE"] = "1" if rep == 4 else "0"
            t0 = time.perf_counter()
            total = dec.decode_batch_into(a, len(offs), r)
            best = min(best, time.perf_counter() - t0)
        os.environ["MI355X_BZ2_TRACE"] = "0"
        bits = [offsets[i + 1] - offsets[i] for i in idx]
        print(f"{name}: {1e3 * best:.2f} ms for {total} bytes ({sum(bits) // 8} compressed)", flush=True)


if __name__ == "__main__":
    main()
