"""The other measurement configurations of SURVEY.md section 8(d) (bench.py itself runs config 4 at N GPUs):

  config 2  batch-size sweep on the Silesia-style workload (blocks per decode_batch: 32 ... all)
  config 3  2 GiB of seeded random bytes, bzip2 -9 (incompressible: C ~ 1.0045 D, all 258 symbols, 6 tables)
  config 5  random pread through the reader API with an imported block map: 1000 x (seek, read 64 KiB), latencies

Prints one JSON object; tools/gpu_round.sh style use:  python tools/bench_configs.py > gpurun_out/configs.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch   # first: one HIP runtime per process

import bench
import bz2build
import indexed_bzip2_amd as m


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def timed_steps(dec, offs_c, n, res_c, steps):
    dec.decode_batch_into(offs_c, n, res_c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = 0
    for _ in range(steps):
        total = dec.decode_batch_into(offs_c, n, res_c)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, total


def config2_and_5(out, only5=False):
    path, enc, meta = bench.build_workload(2 * 1024**3, 214_748_364, "/tmp/indexed_bzip2_amd_bench", 0, 1, lambda: None)
    offsets = meta["offsets"]
    d_in = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
    dec = m.Decoder(device=0, max_batch_blocks=64 if only5 else len(offsets))
    dec.set_input_device(d_in.data_ptr(), len(enc), keepalive=d_in)
    sweep = []
    sub = offsets[:640] if not only5 else []
    # the first quarter of the file (640 blocks ~ 512 MiB decoded) in batches of B blocks
    for B in (32, 64, 128, 256, 640) if not only5 else ():
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
    if not only5:
      out["config2_batch_sweep"] = {"workload": "first 640 blocks (537 MB decoded) of the config-4 file, input resident in HBM, "
                                              "output left in HBM", "results": sweep}
    dec.close()
    del d_in

    # config 5: random pread with a precomputed map (host path: decoded blocks are copied D2H per batch)
    with m.open(path, parallelization=0) as f:
        t0 = time.perf_counter()
        index = f.block_offsets()
        log(f"config5: full decode for the index took {time.perf_counter() - t0:.2f} s")
    size = meta["decoded_bytes"]
    rng = np.random.default_rng(0x5EEC)
    positions = rng.integers(0, size - 65536, 1000)
    for P in (1, 4, 0):
        with m.open(path, parallelization=P) as g:
            g.set_block_offsets(index)
            lat = []
            t_all = time.perf_counter()
            for pos in positions:
                t0 = time.perf_counter()
                g.seek(int(pos))
                data = g.read(65536)
                lat.append(time.perf_counter() - t0)
                assert len(data) == 65536
            wall = time.perf_counter() - t_all
            st = g.statistics()
        lat = np.array(lat) * 1e3
        out[f"config5_random_pread_P{P}"] = {
            "reads": 1000, "read_bytes": 65536, "seed": "0x5EEC", "parallelization": P,
            "latency_ms": {"p50": round(float(np.percentile(lat, 50)), 3), "p95": round(float(np.percentile(lat, 95)), 3),
                           "p99": round(float(np.percentile(lat, 99)), 3), "mean": round(float(lat.mean()), 3)},
            "MBps_of_requested_bytes": round(1000 * 65536 / wall / 1e6, 2),
            "blocks_decoded": st["blocks_decoded"], "gpu_batches": st["batches"]}
        log("config5", out[f"config5_random_pread_P{P}"])


def config3(out):
    cache = "/tmp/indexed_bzip2_amd_bench/urandom-214748364-x10-l9.bz2"
    if not os.path.exists(cache):
        rng = np.random.Generator(np.random.PCG64(0xBADC0DE))
        base = rng.integers(0, 256, 214_748_364, dtype=np.uint8)
        t0 = time.time()
        streams = bz2build.compress_pieces(base, piece_size=9_000_000, level=9, threads=min(32, os.cpu_count() or 8))
        enc, nblocks, offsets = bz2build.stitch(streams, 10, 9, m.find_magic)
        log(f"config3: built in {time.time() - t0:.1f} s, {len(enc) / 1e6:.1f} MB, {nblocks} blocks")
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        with open(cache, "wb") as f:
            f.write(enc)
        json.dump({"offsets": offsets}, open(cache + ".json", "w"))
    enc = open(cache, "rb").read()
    offsets = json.load(open(cache + ".json"))["offsets"]
    d_in = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
    dec = m.Decoder(device=0, max_batch_blocks=len(offsets))
    dec.set_input_device(d_in.data_ptr(), len(enc), keepalive=d_in)
    res, total = dec.decode_batch(offsets)
    assert all(r["status"] == 0 for r in res) and total == 2_147_483_640
    offs_c, res_c = dec.make_arrays(offsets)
    sec, total = timed_steps(dec, offs_c, len(offsets), res_c, 3)
    t = dec.timings()
    alg = sum(r["encoded_size_bits"] / 8 + 10 * r["bwt_length"] + r["decoded_size"] for r in res)
    out["config3_urandom"] = {"workload": "214.7 MB of PCG64(0xBADC0DE) bytes x10, single-stream bzip2 -9",
                              "blocks": len(offsets), "compressed_bytes": len(enc), "decoded_bytes": total,
                              "MBps": round(total / sec / 1e6, 1), "ms_per_step": round(sec * 1e3, 2),
                              "algorithmic_bytes": int(alg), "achieved_GBps": round(alg / sec / 1e9, 1),
                              "kernels_ms": {k: round(v, 2) for k, v in t["kernels"].items()}}
    log("config3", out["config3_urandom"])
    dec.close()


def main():
    """usage: bench_configs.py [3] [5] [all]   (default: all = config 3, then 2 and 5)"""
    out = {}
    which = set(sys.argv[1:]) or {"all"}
    if which & {"3", "all"}:
        config3(out)
    if which & {"5", "all"}:
        config2_and_5(out, only5="all" not in which)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
