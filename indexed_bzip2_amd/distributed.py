"""Multi-GPU plumbing (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" for the CPU tests).

bzip2 blocks are independent, so the decode itself needs no collective: the block queue is partitioned across ranks
(SURVEY 8e).  The only exchange step is the gather of the decoded extents into rank 0 -- RCCL has no gatherv, so sizes
are all-gathered and the payloads move with grouped point-to-point sends (each peer rides its own xGMI link).
"""
import torch
import torch.distributed as dist


def shard_blocks(block_bit_offsets, end_bit, rank, world):
    """Contiguous block range [lo, hi) for `rank`, balanced by COMPRESSED size (the cost of the latency-bound
    Huffman stage is proportional to it).  Deterministic on every rank; ranges tile [0, n) in order."""
    n = len(block_bit_offsets)
    if world <= 1:
        return 0, n
    sizes = [(block_bit_offsets[i + 1] if i + 1 < n else end_bit) - block_bit_offsets[i] for i in range(n)]
    total = sum(sizes)
    bounds = [0]
    acc, r = 0, 1
    for i, s in enumerate(sizes):
        acc += s
        while r < world and acc >= total * r / world:
            bounds.append(i + 1)
            r += 1
    while len(bounds) < world:
        bounds.append(n)
    bounds.append(n)
    bounds = [min(b, n) for b in bounds]
    for k in range(1, len(bounds)):
        bounds[k] = max(bounds[k], bounds[k - 1])
    return bounds[rank], bounds[rank + 1]


def shard_byte_range(block_bit_offsets, file_size, lo, hi):
    """Bytes [first, end) of the file that blocks [lo, hi) need -- what a rank copies to its GPU -- and the blocks' bit
    offsets relative to `first`.  The range starts at the 4-byte word that holds the first block's magic and ends where
    block `hi` starts (or at the end of the file): a block never reads its successor's bytes, and the next block's
    magic is not completely inside, so a magic scan of the range finds exactly these blocks."""
    n = len(block_bit_offsets)
    if lo >= hi:
        return 0, 0, []
    first = (block_bit_offsets[lo] // 8) & ~3
    end = (block_bit_offsets[hi] + 7) // 8 if hi < n else file_size
    return first, end, [o - 8 * first for o in block_bit_offsets[lo:hi]]


def rotl32(value, count):
    count %= 32
    return ((value << count) | (value >> (32 - count))) & 0xFFFFFFFF if count else value & 0xFFFFFFFF


def crc_chain(block_crcs):
    """bzip2's stream CRC of a sequence of block CRCs: c = rotl(c, 1) ^ crc (BZ2Reader.hpp:481-484)."""
    chain = 0
    for crc in block_crcs:
        chain = rotl32(chain, 1) ^ crc
    return chain


def combine_crc_chains(parts):
    """Stream CRC of the whole file from the (chain, number of blocks) pairs of consecutive block ranges, in order.
    rotl is linear over xor, so chain(A then B) = rotl(chain(A), |B|) ^ chain(B): every rank folds its own blocks and
    rank 0 checks the file's end-of-stream CRC without seeing a single block CRC of the others."""
    total = 0
    for chain, count in parts:
        total = rotl32(total, count) ^ chain
    return total


def gather_extents(mine, rank, world, out=None, sizes=None):
    """Gather ragged 1-D uint8 tensors (one decoded extent per rank) into rank 0, in rank order.

    Returns (buffer, sizes) on rank 0 -- buffer holds the extents of ranks 1..world-1 back to back (rank 0's own
    extent stays where it is) -- and (None, sizes) elsewhere.  `out` may be a preallocated receive buffer.  `sizes`: the
    extent sizes of all ranks if the caller knows them already (the same file decoded again: the exchange of sizes and
    the host synchronization that reading them costs are skipped)."""
    if world == 1:
        return None, [int(mine.numel())]
    device = mine.device
    if sizes is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([mine.numel()], dtype=torch.int64, device=device))
        sizes = [int(s.item()) for s in sizes]
    assert sizes[rank] == mine.numel()
    if rank == 0:
        need = sum(sizes[1:])
        if out is None or out.numel() < need:
            out = torch.empty(max(need, 1), dtype=torch.uint8, device=device)
        ops, pos = [], 0
        for r in range(1, world):
            if sizes[r] > 0:
                ops.append(dist.P2POp(dist.irecv, out[pos:pos + sizes[r]], r))
            pos += sizes[r]
    else:
        ops = [dist.P2POp(dist.isend, mine, 0)] if mine.numel() > 0 else []
        out = None
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    return out, sizes
