"""Developer probe: sequential throughput of the reader API (host path, D2H included) vs parallelization."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401  (one HIP runtime)
import bench
import indexed_bzip2_amd as m


def main():
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, "/tmp/indexed_bzip2_amd_bench", 0, 1, lambda: None)
    for P in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "64,256,640,1280")]:
        fd = os.open("/dev/null", os.O_WRONLY)
        t0 = time.perf_counter()
        f = m.IndexedBzip2FileRaw(path, P)
        n = f.bz2reader.read_to_fd(fd)
        dt = time.perf_counter() - t0
        st = f.bz2reader.statistics()
        f.close()
        os.close(fd)
        print(f"P={P}: {n / dt / 1e6:.0f} MB/s ({dt:.2f} s), batches={st['batches']} decode_s={st['decode_seconds']:.2f} "
              f"wait_s={st['wait_seconds']:.2f}", flush=True)


if __name__ == "__main__":
    main()
