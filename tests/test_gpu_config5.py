"""Config 5 of BASELINE.json in the driver-run suite: many seeded uniform random 64 KiB reads on a several-hundred-block
Silesia-style `bzip2 -9` file whose block-offset index was imported, against the raw bytes, at parallelization 1, 4 and 0
(reference analogue: src/tests/indexed_bzip2/testParallelBZ2Reader.cpp:217-256, seeks after setBlockOffsets; the
import rules are src/indexed_bzip2/ParallelBZ2Reader.hpp:365-378).  Plus the read that round 2's finder race broke
(VERDICT r02 weak 1): one sequential read of a large multi-stream file with mi355x_bz2_warmup called first, so that the
GPU magic scan hands its list over while the host scan thread is still running.

The corpus is the bench's generator (tools/silesia_like.py) at 30 MB, compressed in pieces and stitched 11 times into one
stream (tools/bz2build.py): about 330 blocks, 330 MB decoded."""
import bz2
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

BASE_BYTES = 30_000_000
REPEAT = 11
READS = 200
READ_BYTES = 65536


@pytest.fixture(scope="module")
def silesia_file(native, tmp_path_factory):
    import bz2build
    import silesia_like
    threads = min(16, os.cpu_count() or 8)
    base = silesia_like.generate(BASE_BYTES, threads=threads)
    streams = bz2build.compress_pieces(base, piece_size=900_000 * 4, level=9, threads=threads)
    enc, nblocks, offsets = bz2build.stitch(streams, REPEAT, 9, native.find_magic)
    assert nblocks >= 300
    path = tmp_path_factory.mktemp("config5") / "silesia-like.bz2"
    path.write_bytes(enc)
    return str(path), base.tobytes(), enc, offsets


def raw_slice(base, start, length):
    """Bytes [start, start + length) of `base` repeated REPEAT times."""
    out = bytearray()
    while length > 0:
        at = start % len(base)
        piece = base[at:at + length]
        out += piece
        start += len(piece)
        length -= len(piece)
    return bytes(out)


@pytest.fixture(scope="module")
def exported_index(native, silesia_file):
    path, base, enc, offsets = silesia_file
    with native.open(path, parallelization=0) as f:
        index = f.block_offsets()
        assert f.size() == len(base) * REPEAT
    # data blocks, the end-of-stream block and the end-of-file entry; the first block right behind the stream header
    assert len(index) == len(offsets) + 2
    assert list(index.items())[0] == (32, 0)
    assert [bits for bits in index][:len(offsets)] == offsets
    return index


@pytest.mark.parametrize("parallelization", [1, 4, 0])
def test_random_pread_with_imported_index(native, silesia_file, exported_index, parallelization):
    path, base, enc, offsets = silesia_file
    size = len(base) * REPEAT
    positions = np.random.default_rng(0x5EEC).integers(0, size - READ_BYTES, READS)
    with native.open(path, parallelization=parallelization) as f:
        f.set_block_offsets(exported_index)
        assert f.block_offsets_complete() and f.size() == size
        for position in positions:
            position = int(position)
            assert f.seek(position) == position
            data = f.read(READ_BYTES)
            assert data == raw_slice(base, position, READ_BYTES), position
            assert f.tell() == position + READ_BYTES
        stats = f.statistics()
        # nothing was decoded to build an index, and random access does not decode the file front to back: a 64 KiB read
        # touches one or two blocks, and two neighbours in a row look like the start of a sequential read to the access
        # tracker (as to the reference's FetchNextAdaptive), which then decodes a few blocks ahead: measured 1.3 blocks
        # per read at parallelization 1, 6.5 at 4, 13 to 17 at 0 (how far a look-ahead gets before the next seek depends on
        # the timing)
        print(f"config 5, parallelization {parallelization}: {stats}")
        assert stats["blocks_decoded"] <= READS * (4 if parallelization == 1 else 32)
        # the imported index is what the reader reports
        assert f.block_offsets() == exported_index
        # the last bytes and the end of the file
        f.seek(-100, os.SEEK_END)
        assert f.read() == raw_slice(base, size - 100, 100)
        assert f.read(1) == b""


def test_sequential_read_after_warmup_multistream(native, silesia_file, tmp_path):
    """Three streams behind each other (4 x ~330 MB would be the probe of profiles/r02_reader.txt; three keep the suite
    short), read front to back with the default parallelization after mi355x_bz2_warmup: the device magic scan finishes
    while the host finder thread still runs -- the interleaving in which round 2 lost its block list."""
    path, base, enc, offsets = silesia_file
    streams = 3
    multi = tmp_path / "multi.bz2"
    multi.write_bytes(enc * streams)
    native.warmup(0, background=False)
    import hashlib
    want = hashlib.sha256()
    for _ in range(streams * REPEAT):
        want.update(base)
    got = hashlib.sha256()
    total = 0
    with native.open(str(multi), parallelization=0) as f:
        while True:
            chunk = f.read(64 << 20)
            if not chunk:
                break
            got.update(chunk)
            total += len(chunk)
        assert total == streams * REPEAT * len(base)
        assert got.digest() == want.digest()
        index = f.block_offsets()
        assert len(index) == streams * (len(offsets) + 1) + 1
        assert f.statistics()["blocks_decoded"] >= streams * len(offsets)
    # the same file again in the same (now warm) process, in small reads through a second reader
    with native.open(str(multi), parallelization=64) as f:
        f.seek(len(base) * REPEAT - 1000)          # across the first stream's end
        assert f.read(2000) == raw_slice(base, len(base) * REPEAT - 1000, 2000)


@pytest.mark.parametrize("parallelization", [0, 3])
def test_bounded_residency_of_the_compressed_file(native, silesia_file, exported_index, monkeypatch, parallelization):
    """A file that may not be kept on the GPU whole (here: MI355X_BZ2_INPUT_BUDGET far below its 85 MB; in production a
    file beyond the free device memory) is read through all the same: every launch brings the byte range of its own blocks
    (the reference streams through 128 KiB refills, src/core/BitReader.hpp:57, src/core/filereader/Shared.hpp:238-335).
    Front to back against the raw bytes, the block map against the resident reader's, then random reads with that map."""
    import hashlib
    path, base, enc, offsets = silesia_file
    monkeypatch.setenv("MI355X_BZ2_INPUT_BUDGET", str(1 << 20))
    size = len(base) * REPEAT
    want = hashlib.sha256()
    for _ in range(REPEAT):
        want.update(base)
    got = hashlib.sha256()
    with native.open(path, parallelization=parallelization) as f:
        while True:
            chunk = f.read(32 << 20)
            if not chunk:
                break
            got.update(chunk)
        assert f.tell() == size and got.digest() == want.digest()
        assert f.block_offsets() == exported_index
        stats = f.statistics()
        assert stats["input_resident"] == 0
        # every block's bytes went to the GPU once (plus each launch's margin for where its last block may end), not the
        # whole file per launch
        assert len(enc) * 0.9 < stats["input_bytes_uploaded"] < len(enc) + stats["batches"] * 2_400_000 + (1 << 20)
    with native.open(path, parallelization=parallelization) as g:
        g.set_block_offsets(exported_index)
        for position in np.random.default_rng(7).integers(0, size - READ_BYTES, 40):
            position = int(position)
            g.seek(position)
            assert g.read(READ_BYTES) == raw_slice(base, position, READ_BYTES), position
        assert g.statistics()["input_resident"] == 0
    monkeypatch.delenv("MI355X_BZ2_INPUT_BUDGET")
    with native.open(path, parallelization=parallelization) as h:
        assert h.read(1000) == base[:1000]
        assert h.statistics()["input_resident"] == 1


def test_libbz2_agrees_on_a_slice(silesia_file):
    """The stitched file is a valid single stream: CPython's bz2 (libbz2) decodes its head to the raw bytes."""
    path, base, enc, offsets = silesia_file
    d = bz2.BZ2Decompressor()
    out = d.decompress(enc[:4 << 20], 3_000_000)
    assert out == base[:len(out)] and len(out) > 1_000_000
