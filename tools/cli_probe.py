"""Developer probe: wall time of the command line tool on an 8 GiB (decoded) file, process start to exit."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench

CLI = os.path.join(ROOT, "indexed_bzip2_amd", "ibzip2-mi355x")


def main():
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, "/tmp/indexed_bzip2_amd_bench", 0, 1, lambda: None)
    big = "/tmp/big.bz2"
    with open(big, "wb") as f:
        for _ in range(4):
            f.write(enc)
    size = 4 * meta["decoded_bytes"]
    null = open(os.devnull, "wb")
    for label, args in (("-d -c > /dev/null", ["-d", "-c", big]), ("again -vv", ["-d", "-v", "-v", "-c", big]),
                        ("-t (stream CRC)", ["-d", "-t", "-c", big]), ("-P 1280", ["-d", "-P", "1280", "-c", big]),
                        ("-o /tmp/big.out", ["-d", "-f", "-o", "/tmp/big.out", big])):
        t0 = time.perf_counter()
        rc = subprocess.run([CLI] + args, stdout=null).returncode
        dt = time.perf_counter() - t0
        print(f"{label}: rc={rc} {dt:.2f} s = {size / dt / 1e6:.0f} MB/s", flush=True)
    print("output file", os.path.getsize("/tmp/big.out"), "expected", size)
    os.remove("/tmp/big.out")


if __name__ == "__main__":
    main()
