"""Build large single-stream .bz2 files quickly for benchmarks and tests (data preparation, not product code).

`bzip2 -9` on 2 GiB takes minutes single-threaded.  Here the input is cut into pieces, each piece is compressed by
CPython's bz2 module (libbz2) on a thread pool, and the pieces' BLOCKS (bit ranges between the 4-byte stream header
and the EOS marker) are concatenated bit-exactly into ONE stream: "BZh9" + blocks... + EOS magic + combined CRC + pad.
The combined CRC follows the format rule crc = rotl(crc, 1) ^ blockCRC (reference: BZ2Reader.hpp:481-484).
A corpus can be repeated R times by re-appending the same block bit range (blocks are position independent),
which is how the "Silesia-repeat" inputs of BASELINE.json are produced.
"""
import bz2
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MAGIC_BLOCK = 0x314159265359
MAGIC_EOS = 0x177245385090


class BitWriter:
    def __init__(self, capacity_bytes):
        self.buf = np.zeros(capacity_bytes + 16, dtype=np.uint8)
        self.nbits = 0

    def append(self, src, a, b):
        """Append bits [a, b) of the uint8 array src."""
        nbits = b - a
        if nbits <= 0:
            return
        first, sa = a >> 3, a & 7
        nbytes = (nbits + 7) >> 3
        seg = src[first:first + nbytes + 1].astype(np.uint16)
        if len(seg) < nbytes + 1:
            seg = np.concatenate([seg, np.zeros(nbytes + 1 - len(seg), dtype=np.uint16)])
        aligned = (((seg[:-1] << sa) | (seg[1:] >> (8 - sa))) & 0xFF).astype(np.uint8) if sa else seg[:-1].astype(np.uint8)
        tail = nbits & 7
        if tail:
            aligned[-1] &= (0xFF << (8 - tail)) & 0xFF
        pos, r = self.nbits >> 3, self.nbits & 7
        need = pos + nbytes + 2
        if need > len(self.buf):
            self.buf = np.concatenate([self.buf, np.zeros(max(need - len(self.buf), len(self.buf) // 2), dtype=np.uint8)])
        if r == 0:
            self.buf[pos:pos + nbytes] = aligned
        else:
            self.buf[pos:pos + nbytes] |= aligned >> r
            self.buf[pos + 1:pos + 1 + nbytes] |= (aligned.astype(np.uint16) << (8 - r)).astype(np.uint8)
        self.nbits += nbits

    def append_value(self, value, nbits):
        b = np.frombuffer(int(value).to_bytes((nbits + 7) // 8, "big"), dtype=np.uint8)
        pad = len(b) * 8 - nbits
        self.append(b, pad, pad + nbits)

    def tobytes(self):
        return self.buf[:(self.nbits + 7) >> 3].tobytes()


def _read_bits(arr, pos, n):
    v = 0
    for i in range(n):
        bit = (int(arr[(pos + i) >> 3]) >> (7 - ((pos + i) & 7))) & 1
        v = (v << 1) | bit
    return v


def compress_pieces(data, piece_size=18_000_000, level=9, threads=16):
    """data: bytes or uint8 array -> list of independent bz2 streams (bytes)."""
    mv = memoryview(data)
    pieces = [mv[i:i + piece_size] for i in range(0, len(mv), piece_size)]
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda p: bz2.compress(p, level), pieces))


def stitch(streams, repeat=1, level=9, find_magic=None):
    """Concatenate the blocks of `streams` (each a complete single-stream bz2) into one stream, `repeat` times.
    Returns (bytes, n_blocks, block_bit_offsets)."""
    if find_magic is None:
        import indexed_bzip2_amd
        find_magic = indexed_bzip2_amd.find_magic
    ranges = []
    crcs = []
    est = 0
    for s in streams:
        arr = np.frombuffer(s, dtype=np.uint8)
        blocks = find_magic(s, MAGIC_BLOCK)
        eos = find_magic(s, MAGIC_EOS)
        if not blocks:
            continue   # empty piece
        end = eos[-1]
        ranges.append((arr, 32, end, [b - 32 for b in blocks]))
        crcs.extend(_read_bits(arr, b + 48, 32) for b in blocks)
        est += len(s)
    w = BitWriter(est * repeat + 64)
    w.append(np.frombuffer(b"BZh" + str(level).encode(), dtype=np.uint8), 0, 32)
    offsets = []
    stream_crc = 0
    for _ in range(repeat):
        for arr, a, b, rel in ranges:
            base = w.nbits
            offsets.extend(base + r for r in rel)
            w.append(arr, a, b)
        for c in crcs:
            stream_crc = (((stream_crc << 1) | (stream_crc >> 31)) & 0xFFFFFFFF) ^ c
    w.append_value(MAGIC_EOS, 48)
    w.append_value(stream_crc, 32)
    if w.nbits & 7:
        w.append_value(0, 8 - (w.nbits & 7))
    return w.tobytes(), len(offsets), offsets


def build(data, repeat=1, piece_size=18_000_000, level=9, threads=16, find_magic=None):
    return stitch(compress_pieces(data, piece_size, level, threads), repeat, level, find_magic)


if __name__ == "__main__":
    import time
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import silesia_like
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    t0 = time.time()
    d = silesia_like.generate(n)
    t1 = time.time()
    enc, nb, offs = build(d, rep, threads=8)
    t2 = time.time()
    print(f"gen {t1 - t0:.1f}s build {t2 - t1:.1f}s: {n * rep / 1e6:.0f} MB -> {len(enc) / 1e6:.1f} MB, {nb} blocks, ratio {n * rep / len(enc):.2f}")
    t3 = time.time()
    out = bz2.decompress(enc)
    print(f"libbz2 decode {time.time() - t3:.1f}s ok={out == d.tobytes() * rep}")
