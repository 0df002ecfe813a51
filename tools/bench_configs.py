"""The other measurement configurations of SURVEY.md section 8(d) (bench.py itself runs config 4 at N GPUs), each with the
REAL reference (oracle/_ref/ref_bz2 = the reference's own headers compiled by oracle/Makefile) timed beside it on the same
file and the same box, as section 8(d) "CPU reference timing" asks:

  config 2  batch-size sweep on the first 640 blocks of the Silesia-style file (blocks per decode_batch: 32 ... all);
            reference ParallelBZ2Reader on the same 537 MB at parallelization 1 (bounded sample) and all cores
  config 3  2 GiB of seeded random bytes, bzip2 -9 (incompressible: C ~ 1.0045 D, all 258 symbols, 6 tables), through
            bench.py --workload urandom: the bench's own step (host -> HBM inside, four contexts) and one context with the
            input resident; reference at parallelization 1 (bounded sample) and all cores
  config 5  random pread through the reader API with an imported block map: 1000 x (seek, read 64 KiB) at seeded uniform
            offsets, latencies; the reference's ParallelBZ2Reader with the same map (setBlockOffsets) at the same offsets
            (`ref_bz2 pread`), whose bytes must hash to the same value as ours

Prints one JSON object; tools/gpu_round.sh style use:  python tools/bench_configs.py > gpurun_out/configs.json
usage: bench_configs.py [2] [3] [5] [all]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

REF = os.path.join(ROOT, "oracle", "_ref", "ref_bz2")
CACHE = os.environ.get("BZ2_BENCH_CACHE", "/tmp/indexed_bzip2_amd_bench")
CORES = os.cpu_count() or 1


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def reference_bench(path, parallelization, max_bytes):
    """decode-only read of the first max_bytes decoded bytes through the reference's ParallelBZ2Reader"""
    if not os.path.exists(REF):
        return None
    r = subprocess.run([REF, "bench", path, str(parallelization), "1", str(max_bytes)], capture_output=True, text=True, timeout=900)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return {"MBps": round(d["MBps"], 2), "parallelization": parallelization, "decoded_bytes": d["decoded_bytes"],
            "seconds": round(d["seconds"], 3)}


def run_bench(extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + extra
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=e)
    if r.returncode != 0:
        log(r.stderr[-3000:])
        raise SystemExit("bench.py failed")
    return json.loads(r.stdout.strip().splitlines()[-1])


def config2(out):
    import torch   # first: one HIP runtime per process
    import bench
    import indexed_bzip2_amd as m
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, CACHE, 0, 1, lambda: None)
    offsets = meta["offsets"]
    d_in = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
    dec = m.Decoder(device=0, max_batch_blocks=len(offsets))
    dec.set_input_device(d_in.data_ptr(), len(enc), keepalive=d_in)
    sweep = []
    sub = offsets[:640]
    # the first quarter of the file (640 blocks ~ 512 MiB decoded) in batches of B blocks
    for B in (32, 64, 128, 256, 640):
        arrays = [dec.make_arrays(sub[i:i + B]) for i in range(0, len(sub), B)]
        for a, r in arrays:
            dec.decode_batch_into(a, len(a), r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total = 0
        for a, r in arrays:
            total += dec.decode_batch_into(a, min(B, len(a)), r)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sweep.append({"blocks_per_batch": B, "MBps": round(total / dt / 1e6, 1), "ms_per_batch": round(dt / len(arrays) * 1e3, 2)})
        log("config2", sweep[-1])
    dec.close()
    del d_in
    torch.cuda.empty_cache()
    out["config2_batch_sweep"] = {
        "workload": "first 640 blocks (537 MB decoded) of the config-4 file, input resident in HBM, output left in HBM, one context",
        "results": sweep,
        "reference": {"what": "reference ParallelBZ2Reader, decode-only read of the same 537 MB (first 100 MB at parallelization 1)",
                      "cores": CORES,
                      "P1": reference_bench(path, 1, 100_000_000),
                      "all_cores": reference_bench(path, CORES, 537_000_000)}}
    log("config2 reference", out["config2_batch_sweep"]["reference"])


def config3(out):
    common = ["--workload", "urandom", "--no-host-output"]
    four = run_bench(common + ["--steps", "12", "--warmup", "6"])
    one = run_bench(common + ["--steps", "4", "--warmup", "2", "--contexts", "1", "--resident"])
    path = os.path.join(CACHE, "urandom-214748364-x2147483648-l9-v3.bz2")
    out["config3_urandom"] = {
        "workload": four["config"]["workload"],
        "bench_step_four_contexts": {"MBps": four["value"], "ms_per_step": four["ms_per_step"],
                                     "achieved_GBps": four["roofline"]["achieved"], "frac_of_hbm_peak": four["roofline"]["frac"],
                                     "step": four["config"]["step"]},
        "one_context_input_resident": {"MBps": one["value"], "ms_per_step": one["ms_per_step"],
                                       "achieved_GBps": one["roofline"]["achieved"]},
        "kernels_ms": four["roofline"]["kernels_ms"],
        "reference": {"what": "reference ParallelBZ2Reader, decode-only, same file (first 150 MB at parallelization 1, first 1 GB on all cores)",
                      "cores": CORES,
                      "P1": reference_bench(path, 1, 150_000_000),
                      "all_cores": reference_bench(path, CORES, 1_000_000_000)}}
    log("config3", out["config3_urandom"])


def config5(out):
    import numpy as np
    import torch   # noqa: F401  (one HIP runtime per process)
    import bench
    import indexed_bzip2_amd as m
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, CACHE, 0, 1, lambda: None)
    with m.open(path, parallelization=0) as f:
        t0 = time.perf_counter()
        index = f.block_offsets()
        log(f"config5: full decode for the index took {time.perf_counter() - t0:.2f} s")
    size = meta["decoded_bytes"]
    positions = np.random.default_rng(0x5EEC).integers(0, size - 65536, 1000)
    os.makedirs("gpurun_out", exist_ok=True)
    index_path, positions_path = "/tmp/config5.index", "/tmp/config5.positions"
    with open(index_path, "w") as f:
        f.write("".join(f"{bits} {nbytes}\n" for bits, nbytes in index.items()))
    with open(positions_path, "w") as f:
        f.write("".join(f"{int(p)}\n" for p in positions))
    # zlib's CRC-32 over all bytes read, in read order: `ref_bz2 pread` prints the same for the reference's reads
    import zlib
    digests = {}
    from indexed_bzip2_amd.reader import IndexedBzip2FileRaw
    # "raw": seek + read of exactly 64 KiB through the C ABI reader (IndexedBzip2FileRaw), which is what the reference's C++
    # harness does; "buffered": through ibz2.open() = io.BufferedReader with the reference's 1 MiB buffer
    # (indexed_bzip2.pyx:337), which fills its buffer behind every seek, i.e. reads 1 MiB = two or three blocks per request
    for kind, P in (("raw", 1), ("raw", 4), ("raw", 0), ("buffered", 1), ("buffered", 0)):
        with (IndexedBzip2FileRaw(path, P) if kind == "raw" else m.open(path, parallelization=P)) as g:
            reader = g.bz2reader
            reader.set_block_offsets(index)
            lat = []
            crc = 0
            buf = bytearray(65536)
            t_all = time.perf_counter()
            for pos in positions:
                t0 = time.perf_counter()
                g.seek(int(pos))
                if kind == "raw":
                    got = 0
                    while got < 65536:
                        n = g.readinto(memoryview(buf)[got:])
                        assert n > 0
                        got += n
                    data = buf
                else:
                    data = g.read(65536)
                lat.append(time.perf_counter() - t0)
                assert len(data) == 65536
                crc = zlib.crc32(data, crc)
            wall = time.perf_counter() - t_all
            st = reader.statistics()
        digests[(kind, P)] = crc
        lat = np.array(lat) * 1e3
        out[f"config5_random_pread_{kind}_P{P}"] = {
            "reads": 1000, "read_bytes": 65536, "seed": "0x5EEC", "parallelization": P, "through": kind,
            "latency_ms": {"p50": round(float(np.percentile(lat, 50)), 3), "p95": round(float(np.percentile(lat, 95)), 3),
                           "p99": round(float(np.percentile(lat, 99)), 3), "mean": round(float(lat.mean()), 3)},
            "MBps_of_requested_bytes": round(1000 * 65536 / wall / 1e6, 2),
            "blocks_decoded": st["blocks_decoded"], "gpu_batches": st["batches"], "zlib_crc32_of_all_reads": crc}
        log("config5", out[f"config5_random_pread_{kind}_P{P}"])
    assert len(set(digests.values())) == 1, "the three parallelizations read different bytes"
    if os.path.exists(REF):
        ref = {}
        for P in (1, 4, CORES):
            r = subprocess.run([REF, "pread", path, index_path, positions_path, str(P), "65536"], capture_output=True, text=True,
                               timeout=1500)
            ref[f"P{P}"] = json.loads(r.stdout.strip().splitlines()[-1])
            log("config5 reference", ref[f"P{P}"])
        # the reference read the same bytes as this reader
        assert {v["zlib_crc32"] for v in ref.values()} == set(digests.values()), (ref, digests)
        out["config5_reference"] = {"what": "reference ParallelBZ2Reader + setBlockOffsets (ParallelBZ2Reader.hpp:365-378), the same "
                                            "1000 seek + read(65536) (oracle/ref_harness.cpp `pread`), same file, same box",
                                    "cores": CORES, **ref}


def main():
    out = {"host_cores": CORES}
    which = set(sys.argv[1:]) or {"all"}
    if which & {"3", "all"}:
        config3(out)
    if which & {"2", "all"}:
        config2(out)
    if which & {"5", "all"}:
        config5(out)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
