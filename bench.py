#!/usr/bin/env python3
"""bench.py -- headline benchmark: decompressed MB/s of the MI355X bzip2 block decoder.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json metric, configs[3] at N = 1, 2, 4, 8): ONE single-stream `bzip2 -9` file of a seeded
Silesia-style corpus repeated to 2 GiB (tools/silesia_like.py + tools/bz2build.py; the real Silesia corpus is not
available offline): 2 560 blocks, 590 MB compressed.

A "step" = one pass of the hot path over the file, starting from the compressed bytes in (page-locked) HOST memory and
ending with the decoded bytes in HBM: H2D copy of the compressed bytes -> mi355x_bz2_decode_batch
(group-start scan, symbols, MTF -> inverse BWT -> RLE1 -> CRC, all on the GPU).  The copy is queued on the decoder's
stream (mi355x_bz2_set_input_host_async), so with two decoder contexts the transfer of step k+1 runs beside the kernels
of step k.  Every block's CRC is verified on the device in every step; the combination of the device-computed block
CRCs of all ranks is checked against the stream CRC stored in the file.

Multi-GPU (default, "scaling": "strong"): the file's block queue is partitioned into contiguous ranges of about equal
compressed size (indexed_bzip2_amd.distributed.shard_blocks); rank r copies and decodes ONLY its range, and the decoded
extents are gathered into rank 0's HBM over RCCL/xGMI (size all_gather + grouped isend/irecv, RCCL has no gatherv).
`value` = 2 GiB x steps / max-over-ranks time.  `--weak` keeps the round-1 mode (every rank its own whole copy).

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(total_bytes, base_bytes, cache_dir, rank, world, barrier, kind="silesia"):
    """Rank 0 builds (or finds cached) the .bz2 file; everyone loads it.  kind "urandom" = config 3 of BASELINE.json
    (seeded random bytes: incompressible), for tools/bench_configs.py."""
    key = f"silesia_like-{base_bytes}-x{total_bytes}-l9-v3" if kind == "silesia" else f"urandom-{base_bytes}-x{total_bytes}-l9-v3"
    path = os.path.join(cache_dir, key + ".bz2")
    meta_path = path + ".json"
    if rank == 0 and not (os.path.exists(path) and os.path.exists(meta_path)):
        import silesia_like
        import bz2build
        import indexed_bzip2_amd as m
        os.makedirs(cache_dir, exist_ok=True)
        t0 = time.time()
        if kind == "silesia":
            base = silesia_like.generate(base_bytes, threads=min(16, os.cpu_count() or 8))
        else:
            import numpy
            base = numpy.random.Generator(numpy.random.PCG64(0xBADC0DE)).integers(0, 256, base_bytes, dtype=numpy.uint8)
        repeat = max(1, total_bytes // base_bytes)
        t1 = time.time()
        streams = bz2build.compress_pieces(base, piece_size=9_000_000, level=9, threads=min(32, os.cpu_count() or 8))
        t2 = time.time()
        enc, nblocks, offsets = bz2build.stitch(streams, repeat, 9, m.find_magic)
        t3 = time.time()
        sha = hashlib.sha256(base.tobytes()).hexdigest()
        with open(path + ".tmp", "wb") as f:
            f.write(enc)
        os.replace(path + ".tmp", path)
        with open(meta_path, "w") as f:
            json.dump({"decoded_bytes": int(len(base)) * repeat, "base_sha256": sha, "repeat": repeat,
                       "blocks": nblocks, "offsets": offsets}, f)
        log(f"[bench] workload built: generate {t1 - t0:.1f}s compress {t2 - t1:.1f}s stitch {t3 - t2:.1f}s "
            f"-> {len(enc) / 1e6:.1f} MB compressed, {nblocks} blocks")
    barrier()
    with open(path, "rb") as f:
        enc = f.read()
    meta = json.load(open(meta_path))
    return path, enc, meta


def cpu_baseline(path, meta, budget_seconds):
    """Reference ParallelBZ2Reader (oracle/_ref/ref_bz2, the real reference compiled from its own sources) timed on the
    host cores on a bounded prefix of the same file.  Falls back to the oracle port if the binary is absent."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_bz2")
    cores = os.cpu_count() or 1
    if os.path.exists(ref):
        # single-thread figure (the >= 10x target refers to it): ~65 MB/s -> a sample of about 11 s
        sample1 = min(meta["decoded_bytes"], 700_000_000)
        t0 = time.time()
        r1 = json.loads(subprocess.run([ref, "bench", path, "1", "1", str(sample1)], capture_output=True, text=True,
                                       timeout=600).stdout.strip().splitlines()[-1])
        sampleN = min(meta["decoded_bytes"], max(400_000_000, int(r1["MBps"] * 1e6 * cores * 0.5 * 8)))
        rN = None
        if time.time() - t0 < budget_seconds:
            rN = json.loads(subprocess.run([ref, "bench", path, str(cores), "1", str(sampleN)], capture_output=True,
                                           text=True, timeout=600).stdout.strip().splitlines()[-1])
        out = {"value": round(r1["MBps"], 2), "unit": "MB/s", "cores": 1, "kind": "reference",
               "sample": f"first {r1['decoded_bytes'] / 1e6:.0f} MB (decoded) of the same file, reference "
                         f"ParallelBZ2Reader parallelization=1, decode-only ({r1['seconds']:.1f} s); the reference is "
                         f"built by oracle/Makefile with g++ -O3 -DNDEBUG -march=x86-64-v2 (portable across the pool's "
                         f"hosts; SURVEY 6 timed it with -march=native)"}
        if rN is not None:
            out["all_cores"] = {"value": round(rN["MBps"], 2), "cores": cores,
                                "sample": f"first {rN['decoded_bytes'] / 1e6:.0f} MB, parallelization={cores} "
                                          f"({rN['seconds']:.1f} s)"}
        return out
    from oracle import oracle as O
    enc = open(path, "rb").read()
    offs = meta["offsets"][:24]
    t0 = time.time()
    n = 0
    for o in offs:
        d, _ = O.decode_block(enc, o)
        n += d["decoded_size"]
    dt = time.time() - t0
    return {"value": round(n / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": f"first {len(offs)} blocks ({n / 1e6:.0f} MB decoded) through oracle/bz2_oracle.c ({dt:.1f} s)"}


class _DevicePtr:
    """Zero-copy view of a raw device pointer for torch.as_tensor (plumbing for the RCCL gather)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--total-bytes", type=int, default=2 * 1024**3)
    ap.add_argument("--base-bytes", type=int, default=214_748_364)   # x10 = 2 GiB - 8 bytes
    ap.add_argument("--cache-dir", default=os.environ.get("BZ2_BENCH_CACHE", "/tmp/indexed_bzip2_amd_bench"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-host-output", action="store_true", help="skip the decoded-bytes-in-host-memory figure")
    ap.add_argument("--workload", default="silesia", choices=["silesia", "urandom"],
                    help="silesia = the headline workload (config 4 of BASELINE.json); urandom = config 3 (2 GiB of seeded "
                         "random bytes at bzip2 -9: incompressible), used by tools/bench_configs.py")
    ap.add_argument("--weak", action="store_true",
                    help="round-1 mode: every rank decodes its own whole copy of the file (per-GPU work fixed)")
    ap.add_argument("--resident", action="store_true",
                    help="secondary figure only: compressed bytes already resident in HBM before the timed region")
    ap.add_argument("--contexts", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 8],
                    help="decoder contexts used in turn: step k+1 is queued on the next context before step k is "
                         "finished, so the latency-bound first kernels of one step overlap the throughput kernels of the "
                         "others.  0 = 4 (measured on one box, 2 560 blocks per step: 3 contexts 74.5 ms, 4: 72.1, 5: 83.6; "
                         "a rank's share of the file at 8 GPUs, 310 blocks: 4 contexts 14.0 ms, 5: 15.7)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 flow on ONE GPU: ranks share cuda:0 and extents travel via host memory")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.contexts == 0:
        args.contexts = 4
    # every decoder context drives up to 4 HIP streams: without this the runtime maps them onto 4 hardware queues and
    # streams that share a queue serialize (must be set before the HIP runtime starts)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(4 * args.contexts))
    import torch   # first: the process must use ONE HIP runtime (torch's), the extension binds to the loaded one
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}: use torch.distributed.run for N>1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")

    def barrier():
        if world > 1:
            dist.barrier()

    from indexed_bzip2_amd import build as hipbuild
    if rank == 0:
        hipbuild.build()
    barrier()
    import numpy as np
    import indexed_bzip2_amd as m
    from indexed_bzip2_amd.distributed import (shard_blocks, shard_byte_range, gather_extents, crc_chain,
                                               combine_crc_chains)

    path, enc, meta = build_workload(args.total_bytes, args.base_bytes, args.cache_dir, rank, world, barrier, args.workload)
    offsets = meta["offsets"]
    n_file_blocks = len(offsets)
    file_decoded = meta["decoded_bytes"]

    # ---- this rank's part of the block queue: a contiguous range of about 1/N of the compressed bytes ----
    strong = not args.weak
    lo, hi = shard_blocks(offsets, len(enc) * 8, rank, world) if strong else (0, n_file_blocks)
    byte0, byte1, my_offsets = shard_byte_range(offsets, len(enc), lo, hi)
    n_blocks = len(my_offsets)
    my_bytes = byte1 - byte0
    # compressed bytes in page-locked host memory: where every step starts from
    host_in = torch.empty(max(my_bytes, 1), dtype=torch.uint8).pin_memory()
    host_in[:my_bytes] = torch.frombuffer(enc, dtype=torch.uint8)[byte0:byte1]
    host_ptr = host_in.data_ptr()

    decs = [m.Decoder(device=device_index, max_batch_blocks=max(n_blocks, 1)) for _ in range(args.contexts)]
    dec = decs[0]

    # block offsets: the GPU magic scan over the resident range must find exactly the blocks the file was built from
    dec.set_input_host_async(host_ptr, my_bytes, keepalive=host_in)
    dec.find_magic()
    t_scan = time.perf_counter()
    gpu_offsets = dec.find_magic()
    scan_ms = (time.perf_counter() - t_scan) * 1e3
    assert gpu_offsets == my_offsets, "GPU magic scan disagrees with the block offsets of the workload"

    # ---- correctness gate (also the first warm-up): statuses, sizes, and the file's stream CRC over ALL ranks ----
    results, my_decoded = dec.decode_batch(my_offsets)
    bad = [r for r in results if r["status"] != 0]
    assert not bad, f"{len(bad)} blocks failed: {bad[:2]}"
    chain = crc_chain(r["computed_crc"] for r in results)
    if world > 1:
        parts = [None] * world
        # (the block sizes and checksums of every rank: rank 0 checks the bytes that ARRIVE in its gather buffer against them)
        dist.all_gather_object(parts, (chain, n_blocks, my_decoded,
                                       [(r["decoded_size"], r["computed_crc"]) for r in results] if strong else None))
    else:
        parts = [(chain, n_blocks, my_decoded, None)]
    extent_sizes = [p[2] for p in parts]
    if strong:
        stream_crc = combine_crc_chains((p[0], p[1]) for p in parts)
        assert sum(p[2] for p in parts) == file_decoded, (parts, file_decoded)
    else:
        stream_crc = chain
        assert my_decoded == file_decoded
    eos_bit = offsets[-1] + (results[-1]["encoded_size_bits"] if hi == n_file_blocks else 0)
    if rank == world - 1 or not strong:
        stored = 0
        for i in range(32):
            b = eos_bit + 48 + i
            stored = (stored << 1) | ((enc[b >> 3] >> (7 - (b & 7))) & 1)
        assert stream_crc == stored, f"checksum of block checksums {stream_crc:08x} != stream CRC in file {stored:08x}"
    # (the per-block CRCs cover every decoded byte; tests/ compare full payloads against the oracle at smaller sizes)
    view = torch.as_tensor(_DevicePtr(dec.output_device_ptr(), my_decoded), device="cuda")
    assert bytes(view[:4096].cpu().numpy()) == dec.copy_output(0, 4096)
    assert bytes(view[my_decoded - 4096:].cpu().numpy()) == dec.copy_output(my_decoded - 4096, 4096)

    gather_buf = None
    gather_done = {}

    def gather(total, decoder):
        """Decoded extents of all ranks -> rank 0 HBM (RCCL over xGMI)."""
        nonlocal gather_buf
        if world == 1 or args.no_gather:
            return
        mine = torch.as_tensor(_DevicePtr(decoder.output_device_ptr(), total), device="cuda")
        if args.backend == "gloo":
            mine = mine.cpu()
        # the sizes are those of the correctness pass (exchanged once): no size exchange, no host sync per step
        buf, _sizes = gather_extents(mine, rank, world, gather_buf, sizes=extent_sizes if strong else None)
        if rank == 0:
            gather_buf = buf
        if args.backend == "nccl":
            # the sends read the context's output buffer asynchronously: the context's next batch writes its output (queued
            # by its begin_batch, long before its end_batch) only behind this event
            done = torch.cuda.Event()
            done.record()
            gather_done[id(decoder)] = done
            decoder.hold_output_until(done.cuda_event, keepalive=done)

    # ---- content check of the gather, once, outside the timed region: every block of every peer, as it lies in rank 0's
    # gather buffer, has the checksum its sender computed (= the one stored in the file: status OK above) ----
    gathered_blocks_checked = 0
    if world > 1 and strong and not args.no_gather:
        gather(my_decoded, dec)
        pending = gather_done.pop(id(dec), None)
        if pending is not None:
            pending.synchronize()
        if rank == 0:
            received = gather_buf if gather_buf.is_cuda else gather_buf.cuda()
            torch.cuda.synchronize()
            at = 0
            for r in range(1, world):
                blocks = parts[r][3]
                assert sum(size for size, _ in blocks) == extent_sizes[r]
                # the extent of rank r starts at `at`; k_crc wants a 16-byte aligned base: checksum from an aligned copy if not
                extent = received[at:at + extent_sizes[r]]
                if extent.data_ptr() % 16:
                    extent = extent.clone()
                got = dec.crc32_device(extent.data_ptr(), [size for size, _ in blocks])
                want = [crc for _, crc in blocks]
                assert got == want, f"gathered extent of rank {r}: {sum(a != b for a, b in zip(got, want))} blocks differ"
                gathered_blocks_checked += len(blocks)
                at += extent_sizes[r]
            log(f"[bench] gather verified: {gathered_blocks_checked} blocks of {world - 1} peers have their senders' checksums")
        barrier()

    offs_c, _ = dec.make_arrays(my_offsets)
    res_cs = [d.make_arrays(my_offsets)[1] for d in decs]
    status_views = [np.frombuffer(r, dtype=np.int32).reshape(max(n_blocks, 1), -1)[:, -1] for r in res_cs]   # BlockResult.status
    resident_in = None

    host_out = []          # page-locked host buffers, one per context, for the "decoded bytes in host memory" figure

    def finish(k, to_host=False):
        """Second half of step k on its context: every block's status checked; decoded extents gathered for N > 1, or
        (to_host) copied to page-locked host memory in the background, beside the context's next step."""
        d = decs[k % len(decs)]
        pending = gather_done.pop(id(d), None)
        if pending is not None:
            pending.synchronize()   # long finished in practice: a whole step has passed
        if to_host:
            d.copy_output_end()     # the copy of this context's previous step: its host buffer is written again below
        total = d.end_batch(res_cs[k % len(decs)])
        assert total == my_decoded and not status_views[k % len(decs)].any(), "a block failed"
        if to_host:
            buf = host_out[k % len(decs)]
            d.copy_output_begin_to(0, total, buf.data_ptr(), keepalive=buf)
        else:
            gather(total, d)

    def run_steps(count, resident=False, to_host=False):
        """`count` passes of the hot path, each through the C ABI with preallocated arrays (no per-block Python
        objects).  A pass = the H2D copy of the compressed bytes, then the batch.  With several contexts step k+1 is
        queued before step k is finished, and the copy for a context's next step is queued as soon as its current step
        has been launched (into the context's second input buffer); all `count` copies and steps begin and end inside
        this call."""
        gpu_ms = 0.0
        depth = len(decs)
        prefetch = os.environ.get("BENCH_NO_PREFETCH") != "1"     # development: copy and batch of a step strictly in turn
        stagger = float(os.environ.get("BENCH_STAGGER_MS", "0")) / 1e3   # development: pause between the first launches
        for k in range(count + depth):
            if k >= depth:                       # step k - depth holds the context that step k needs
                finish(k - depth, to_host)
                gpu_ms += decs[(k - depth) % depth].pipeline_ms()
            if k < count:
                d = decs[k % depth]
                if not resident and (k < depth or not prefetch):
                    d.set_input_host_async(host_ptr, my_bytes, keepalive=host_in)
                d.begin_batch(offs_c, n_blocks)
                if stagger and k + 1 < depth:
                    time.sleep(stagger)
                if not resident and prefetch and k + depth < count:
                    # the bytes of this context's NEXT step (step k + depth): their copy runs beside the kernels of this one
                    d.set_input_host_async(host_ptr, my_bytes, keepalive=host_in)
        if to_host:
            for d in decs:
                d.copy_output_end()
        return gpu_ms

    run_steps(max(len(decs), args.warmup - 1))   # warm-up; also sizes the scratch of every context

    alg_bytes = sum(r["encoded_size_bits"] / 8 + 10 * r["bwt_length"] + r["decoded_size"] for r in results)
    io_floor = sum(r["encoded_size_bits"] / 8 + r["decoded_size"] for r in results)

    def timed(count, resident=False, to_host=False):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gpu_ms = run_steps(count, resident, to_host)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, gpu_ms

    dt, gpu_ms = timed(args.steps, resident=args.resident)
    # per-kernel HIP-event durations of the LAST timed step (its events are still there)
    t = decs[(args.steps - 1) % len(decs)].timings()
    ksum = dict(t["kernels"])
    kernel_ms = t["ms_kernel_sum"]
    # secondary figure: the same steps with the compressed bytes already resident in HBM (round-1's headline)
    resident_dt = None
    if not args.resident:
        for d in decs:
            d.set_input_host_async(host_ptr, my_bytes, keepalive=host_in)
            d.decode_batch(my_offsets[:1])       # orders the copy; the input stays
        resident_steps = max(2, min(10, args.steps))      # enough steps for the contexts' pipeline to fill
        resident_dt, _ = timed(resident_steps, resident=True)

    # second figure of SURVEY 8(d): "decoded bytes in host memory" -- every step's output copied to page-locked host memory
    # (mi355x_bz2_copy_output_begin / _end: on a copy stream, beside the context's next step); single GPU only
    host_dt = host_steps = None
    if world == 1 and not args.resident and not args.no_host_output:
        host_out.extend(torch.empty(my_decoded, dtype=torch.uint8).pin_memory() for _ in decs)
        run_steps(len(decs), to_host=True)
        host_steps = max(2, min(10, args.steps))
        host_dt, _ = timed(host_steps, to_host=True)
        # the bytes that arrived: head and tail of the last step's buffer against the device's
        last = host_out[(host_steps - 1) % len(decs)]
        assert bytes(last[:4096].numpy()) == decs[(host_steps - 1) % len(decs)].copy_output(0, 4096)
        assert bytes(last[my_decoded - 4096:].numpy()) == decs[(host_steps - 1) % len(decs)].copy_output(my_decoded - 4096, 4096)
        host_out.clear()
    # every kernel launched ONCE over the whole batch, nothing beside it (no block groups, one context, input resident):
    # the per-kernel durations that mean something on their own
    alone = None
    if world == 1:
        os.environ["MI355X_BZ2_NO_SPLIT"] = "1"
        try:
            d0 = decs[0]
            for _ in range(2):
                d0.begin_batch(offs_c, n_blocks)
                d0.end_batch(res_cs[0])
            alone = d0.timings()
        finally:
            del os.environ["MI355X_BZ2_NO_SPLIT"]

    if rank == 0:
        steps = args.steps
        job_decoded = file_decoded if strong else file_decoded * world
        value = job_decoded * steps / dt / 1e6
        dom = max(ksum, key=ksum.get)
        # one "launch" = the kernel pipeline of one decode_batch: its groups of blocks run on several HIP streams and
        # overlap; pipeline_ms = HIP events before the first and after the last kernel of a step.  With two contexts
        # consecutive steps overlap as well, so the rate is taken over the whole timed region (device-synchronized on
        # both sides), which is never shorter than what the events of a single step would give.
        pipeline_ms = gpu_ms / steps
        launch_ms = dt / steps * 1e3
        achieved = alg_bytes / (launch_ms / 1e3) / 1e9
        traffic = None
        # PMC traffic of one step of this workload (tools/profile_round.sh traffic: separate FETCH_SIZE / WRITE_SIZE passes,
        # corrected as calibrated by tools/fetch_calib.sh), measured for the single-GPU case
        tpath = os.path.join(ROOT, "profiles", "r03_traffic_bench.json")
        if os.path.exists(tpath) and world == 1 and n_blocks == 2560 and args.workload == "silesia":
            traffic = json.load(open(tpath)).get("hbm_bytes_per_step")
        out = {
            "metric": "decompressed MB/s (whole node), 2 GiB Silesia bz2-9" if args.workload == "silesia"
                      else "decompressed MB/s, 2 GiB urandom bz2-9 (config 3)",
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"silesia-style corpus ({args.base_bytes / 1e6:.0f} MB, seed 0x51E51A)" if args.workload == "silesia"
                                    else f"PCG64(0xBADC0DE) random bytes ({args.base_bytes / 1e6:.0f} MB)") + " repeated to "
                                   f"{file_decoded / 2**30:.2f} GiB, ONE single-stream bzip2 -9 file, {n_file_blocks} blocks, "
                                   f"{len(enc) / 1e6:.0f} MB compressed" + ("" if strong else ", one whole copy per GPU"),
                       "blocks_rank0": n_blocks, "compressed_bytes_rank0": my_bytes, "decoded_bytes_rank0": my_decoded,
                       "ratio": round(file_decoded / len(enc), 3),
                       "parallelism": (f"block queue of the file partitioned over {world} GPU(s) in contiguous ranges of equal "
                                       f"compressed size" if strong else f"every one of {world} GPU(s) decodes a whole copy")
                                      + ("" if world == 1 or args.no_gather else ", RCCL gather of decoded extents to rank 0"),
                       "decoder_contexts": len(decs),
                       "gathered_blocks_verified_on_rank0": gathered_blocks_checked if world > 1 else None,
                       "input_resident_in_hbm": bool(args.resident), "output_left_in_hbm": True,
                       "step": ("compressed bytes resident in HBM -> decoded bytes in HBM" if args.resident else
                                "compressed bytes in page-locked host memory -> H2D (on the context's input stream, beside the "
                                "kernels of the step in front of it) -> decoded bytes in HBM"),
                       "block_offsets": "known before the timed region (index / finder thread); the same offsets from the "
                                        "GPU magic scan k_find_magic take %.2f ms (not part of a step)" % scan_ms},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_step": int(alg_bytes), "io_floor_bytes_per_step": int(io_floor),
                         "definition": "rank 0: sum over its blocks of (C + 10 N + D) of one step / (duration of the timed "
                                       "region / steps), the region being device-synchronized on both sides.  "
                                       "pipeline_ms_per_step = HIP events on the launch streams from before the first to "
                                       "after the last kernel of a step; with two contexts consecutive steps overlap, so "
                                       "these add up to more than the region.  kernels_ms = HIP-event duration of every "
                                       "kernel launched ONCE over the whole batch with nothing beside it (no block groups, "
                                       "one context, input resident; measured after the timed region); "
                                       "kernels_ms_in_the_crowd = durations of the last timed step summed over its block "
                                       "groups, which run concurrently on separate streams beside three other steps (a sum "
                                       "can exceed ms_per_step)",
                         "pipeline_ms_per_step": round(pipeline_ms, 3),
                         "kernel_ms_sum_per_step": round(kernel_ms, 3), "dominant_kernel": dom,
                         "kernels_ms": {k: round(v, 3) for k, v in (alone["kernels"] if alone else ksum).items() if v > 0},
                         "kernels_ms_in_the_crowd": {k: round(v, 3) for k, v in ksum.items() if v > 0}},
        }
        if alone:
            out["roofline"]["kernel_ms_sum_alone"] = round(alone["ms_kernel_sum"], 3)
            out["roofline"]["dominant_kernel"] = max(alone["kernels"], key=alone["kernels"].get)
        memory = [d.device_memory() for d in decs]
        out["config"]["device_memory_GB"] = {
            "contexts": len(decs),
            "scratch": round(sum(m["scratch_bytes"] for m in memory) / 1e9, 2),
            "output_buffers": round(sum(m["output_bytes"] for m in memory) / 1e9, 2),
            "scratch_MB_per_block": round(memory[0]["scratch_bytes"] / max(1, n_blocks) / 1e6, 2)}
        if host_dt is not None:
            out["config"]["host_output_MBps"] = round(job_decoded * host_steps / host_dt / 1e6, 1)
            out["config"]["host_output_ms_per_step"] = round(host_dt / host_steps * 1e3, 3)
            out["config"]["host_output"] = ("same steps, each step's decoded bytes copied to page-locked host memory on a copy "
                                            "stream beside the context's next step (mi355x_bz2_copy_output_begin/_end)")
        if resident_dt is not None:
            out["config"]["resident_input_MBps"] = round(job_decoded * resident_steps / resident_dt / 1e6, 1)
            out["config"]["resident_input_ms_per_step"] = round(resident_dt / resident_steps * 1e3, 3)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(path, meta, 40.0)
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
