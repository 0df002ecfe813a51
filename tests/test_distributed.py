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
        # stand-in for the GPU decode of this rank's shard: the oracle (checker) produces the extent bytes
        extent = b"".join(O.decode_block(enc, o)[1] for o in offs[lo:hi])
        mine = torch.frombuffer(bytearray(extent), dtype=torch.uint8) if extent else torch.empty(0, dtype=torch.uint8)
        buf, sizes = D.gather_extents(mine, rank, world)
        if rank == 0:
            whole = extent + bytes(buf[:sum(sizes[1:])].tolist() if sum(sizes[1:]) < 1 << 16 else buf[:sum(sizes[1:])].numpy().tobytes())
            q.put(("ok", whole == raw, sizes, (lo, hi)))
        else:
            q.put(("peer", True, sizes, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
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
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] >= ranges[1][0]
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
