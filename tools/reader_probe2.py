"""Developer probe: phases of a sequential read through the reader API on a larger (concatenated) file."""
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
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, "/tmp/indexed_bzip2_amd_bench", 0, 1, lambda: None)
    big = path + f".x{copies}"
    with open(big, "wb") as f:
        for _ in range(copies):
            f.write(enc)
    if os.environ.get("PROBE_WARMUP") == "1":      # what an application that calls ibz2.warmup() while it starts sees
        t = time.perf_counter()
        m.warmup(background=False)
        print(f"warmup {1e3 * (time.perf_counter() - t):.0f} ms", flush=True)
    for P in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1280,2560")]:
        fd = os.open("/dev/null", os.O_WRONLY)
        t0 = time.perf_counter()
        f = m.IndexedBzip2FileRaw(big, P)
        f.bz2reader.set_verify_stream_crc(True)
        t1 = time.perf_counter()
        n = f.bz2reader.read_to_fd(fd, 1)
        tb = time.perf_counter()
        n += f.bz2reader.read_to_fd(fd, (1 << 20) - 1)
        t2 = time.perf_counter()
        marks = []
        while True:
            got = f.bz2reader.read_to_fd(fd, 1 << 30)
            if got == 0:
                break
            n += got
            marks.append(time.perf_counter())
        dt = time.perf_counter() - t0
        st = f.bz2reader.statistics()
        assert n == copies * meta["decoded_bytes"] and f.bz2reader.streams_verified() == copies, (n, f.bz2reader.streams_verified())
        f.close()
        os.close(fd)
        per_gib = [round((b - a) * 1e3) for a, b in zip([t2] + marks[:-1], marks)]
        print(f"P={P}: {n / dt / 1e6:.0f} MB/s total ({dt:.2f} s); open {1e3 * (t1 - t0):.0f} ms, first byte {1e3 * (tb - t0):.0f} ms, first MiB {1e3 * (t2 - t0):.0f} ms, "
              f"ms per GiB {per_gib}; batches={st['batches']} decode_s={st['decode_seconds']:.2f} wait_s={st['wait_seconds']:.2f}",
              flush=True)


if __name__ == "__main__":
    main()
