"""CPU suite, part 3: the N>1 path -- block-queue sharding and the gather of decoded extents to rank 0 -- on the gloo
backend with world_size 2 (the GPU path uses the same code with backend nccl = RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401  (puts the repo on sys.path)
import datagen


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, enc, raw, q):
    import sys
    sys.path.insert(0, ROOT)
    from indexed_bzip2_amd import distributed as D
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        offs = O.find_magic(enc)
        lo, hi = D.shard_blocks(offs, len(enc) * 8, rank, world)
        # this rank sees ONLY its byte range of the file, as bench.py --gpus N copies only that range to its GPU
        first, end, local = D.shard_byte_range(offs, len(enc), lo, hi)
        mine_enc = enc[first:end]
        assert O.find_magic(mine_enc) == local, "a magic scan of the range must find exactly the rank's blocks"
        # stand-in for the GPU decode of this rank's shard: the oracle (checker) produces the extent bytes
        decoded = [O.decode_block(mine_enc, o) for o in local]
        assert all(d["status"] == 0 for d, _ in decoded)
        extent = b"".join(payload for _, payload in decoded)
        chain = D.crc_chain(d["computed_crc"] for d, _ in decoded)
        parts = [None] * world
        dist.all_gather_object(parts, (chain, len(local)))
        mine = torch.frombuffer(bytearray(extent), dtype=torch.uint8) if extent else torch.empty(0, dtype=torch.uint8)
        buf, sizes = D.gather_extents(mine, rank, world)
        if rank == 0:
            whole = extent + buf[:sum(sizes[1:])].numpy().tobytes()
            # the file's own end-of-stream CRC, from the partial chains of all ranks
            eos = O.find_magic(enc, O.MAGIC_EOS)[-1]
            stored = O.read_block_header(enc, eos)["header_crc"]
            q.put(("ok", whole == raw and D.combine_crc_chains(parts) == stored, sizes, (lo, hi)))
        else:
            q.put(("peer", True, sizes, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_decode_and_gather_gloo(world):
    raw = datagen.text_like(700_000, 51) + datagen.random_bytes(200_000, 52) + datagen.runs(300_000, 53)
    enc = datagen.compress(raw, 1)      # ~12 blocks of 100 kB
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, enc, raw, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ok = [r for r in results if r[0] == "ok"][0]
    assert ok[1], "gathered extents do not reproduce the decoded stream"
    ranges = sorted(r[3] for r in results)
    assert ranges[0][0] == 0 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and ranges[-1][1] >= ranges[-1][0]
    assert sum(ok[2]) == len(raw)


def test_shard_blocks_properties():
    from indexed_bzip2_amd import distributed as D
    offs = [32, 1000, 5000, 5100, 90000, 90500, 200000]
    end = 250000
    for world in (1, 2, 3, 4, 8, 16):
        ranges = [D.shard_blocks(offs, end, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == len(offs)
        for a, b in zip(ranges, ranges[1:]):
            assert a[1] == b[0] and a[0] <= a[1]
    assert D.shard_blocks([], 0, 0, 2) == (0, 0)


def test_crc_chain_combination():
    import random
    from indexed_bzip2_amd import distributed as D
    r = random.Random(7)
    crcs = [r.getrandbits(32) for _ in range(1000)]
    whole = D.crc_chain(crcs)
    for cuts in ([0, 1000], [0, 1, 1000], [0, 333, 334, 900, 1000], [0, 32, 64, 1000], [0, 0, 500, 500, 1000]):
        parts = [(D.crc_chain(crcs[a:b]), b - a) for a, b in zip(cuts, cuts[1:])]
        assert D.combine_crc_chains(parts) == whole


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_bench_sharded_rehearsal_on_one_gpu(world, tmp_path):
    """bench.py's N > 1 flow on ONE GPU (gloo, ranks share cuda:0, extents travel through host memory): the block queue
    of one file partitioned over the ranks, every rank copies and decodes only its byte range, the extents are gathered
    to rank 0, and the file's stream CRC is checked against the combination of all ranks' block-CRC chains (bench.py
    asserts it on the last rank).  A 64 MB workload keeps it short."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BZ2_BENCH_CACHE=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline",
           "--total-bytes", str(64_000_000), "--base-bytes", str(16_000_000)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["input_resident_in_hbm"] is False
    assert line["config"]["blocks_rank0"] < 80 // world + 8      # rank 0 holds its share of the ~72 blocks only
