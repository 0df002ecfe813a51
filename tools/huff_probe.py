"""Developer probe: latency of small batches -- single blocks, eight, a 320-block batch, the whole 2 560-block batch -- on one
context.  Prints the best wall time of four per batch and the per-kernel event durations.  Arguments: labels, one line each
(set the MI355X_BZ2_* knobs of INTEGRATION.md in the environment to compare forms)."""
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
    modes = sys.argv[1:] or ["scan"]
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, "/tmp/indexed_bzip2_amd_bench", 0, 1, lambda: None)
    offsets = meta["offsets"]
    sizes = [(b - a, i) for i, (a, b) in enumerate(zip(offsets, offsets[1:]))]
    large = max(sizes)[1]
    median = sorted(sizes)[len(sizes) // 2][1]
    dec = m.Decoder(device=0, max_batch_blocks=len(offsets))
    dec.set_input(enc)
    cases = (("median block", [median]), ("largest block", [large]), ("8 blocks", list(range(100, 108))),
             ("320 blocks", list(range(0, 320))), ("all blocks", list(range(len(offsets)))))
    for name, idx in cases:
        offs = [offsets[i] for i in idx]
        a, r = dec.make_arrays(offs)
        for mode in modes:
            dec.decode_batch_into(a, len(offs), r)
            best = 1e9
            for rep in range(4):
                t0 = time.perf_counter()
                total = dec.decode_batch_into(a, len(offs), r)
                best = min(best, time.perf_counter() - t0)
            t = dec.timings()
            k = {n: round(v, 2) for n, v in t["kernels"].items() if v > 0.005}
            print(f"{name:14s} {mode:7s}: {1e3 * best:8.2f} ms  pipeline {dec.pipeline_ms():7.2f} ms  {k}", flush=True)


if __name__ == "__main__":
    main()
