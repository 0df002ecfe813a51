"""The decompression matrix of the reference's Python test, src/tests/testPythonWrappers.py:171-245 (testDecompression)
and :476-553 (the b'A' * size regression sizes of rapidgzip/issues/55 and the parameter grid): sizes x {random, AB
stripes of 1, 2, 8, 123, 257, 2048, 100000} x encoders {CPython bz2, /usr/bin/bzip2} x levels 1..9, every file read
through read(bufferSize) for bufferSize in {-1, 128, 333, 500, 1024, 1 Mi, 64 Mi} and compared by SHA-1 with the raw
data, then seeks (position 0, size - 1, a seeded random one) before and after a block-offset export / import, for
parallelization 1, 2, 3, 8.  pbzip2 is not installed here.

The reference walks the full grid (2 016 files per parallelization) on a process pool; every open() here creates a GPU
decoder, so each (size, pattern) cell takes ONE encoder and ONE level, rotated so that all encoders and levels occur for
every pattern and every size class; all buffer sizes and all four parallelizations are kept.
"""
import bz2
import hashlib
import io
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 4, 5, 10, 20, 30, 100, 1000, 10000, 100000, 200000, 0]
PATTERNS = [None, 1, 2, 8, 123, 257, 2048, 100000]          # None = random bytes
BUFFER_SIZES = [-1, 128, 333, 500, 1024, 1024 * 1024, 64 * 1024 * 1024]
ENCODERS = ["pybz2"] + (["bzip2"] if os.path.exists("/usr/bin/bzip2") else [])


def make_data(size, pattern, seed):
    if pattern is None:
        return np.random.default_rng(seed).integers(0, 256, size, dtype=np.uint8).tobytes()
    data = b""
    while len(data) < size:                                    # createStripedCompressedFile, :145-152
        for char in (b"A", b"B"):
            data += char * min(pattern, size - len(data))
    return data


def encode(data, level, encoder):
    if encoder == "pybz2":
        return bz2.compress(data, level)
    return subprocess.check_output(["/usr/bin/bzip2", f"-{level}"], input=data)


def sha1_of(file, buffer_size):
    h = hashlib.sha1()
    while True:
        piece = file.read(buffer_size)
        if not piece:
            break
        h.update(piece)
    return h.digest()


def check_seek(raw, file, pos):
    assert file.seek(pos) == pos
    got = file.read(256)
    assert pos <= file.tell() <= pos + 256
    assert got == raw[pos:pos + 256]


def cells():
    k = 0
    for size in SIZES:
        for pattern in PATTERNS:
            yield size, pattern, 1 + k % 9, ENCODERS[(k // 9) % len(ENCODERS)]
            k += 1


@pytest.mark.parametrize("parallelization", [1, 2, 3, 8])
def test_decompression_matrix(native, tmp_path, parallelization):
    rng = np.random.default_rng(0xC0FFEE + parallelization)
    path = str(tmp_path / "matrix.bz2")
    levels, encoders = set(), set()
    used_buffer_sizes = set()
    for cell, (size, pattern, level, encoder) in enumerate(cells()):
        raw = make_data(size, pattern, seed=size * 31 + (pattern or 7))
        enc = encode(raw, level, encoder)
        levels.add(level)
        encoders.add(encoder)
        with open(path, "wb") as f:
            f.write(enc)
        want = hashlib.sha1(raw).digest()
        where = (size, pattern, level, encoder, parallelization)
        for buffer_size in [BUFFER_SIZES[(cell + k) % len(BUFFER_SIZES)] for k in (0, 2, 4)]:   # checkDecompression, :78-88
            used_buffer_sizes.add(buffer_size)
            # like the reference's CompressedFileReader(name) these readers take the default parallelization; the
            # parameter's value is used for the seek checks below (testPythonWrappers.py:193, 209, 228)
            with native.IndexedBzip2File(path, parallelization if buffer_size in (-1, 333, 1024) else 1) as f:
                assert sha1_of(f, buffer_size) == want, (where, buffer_size)
        if size > 0:
            f = native.IndexedBzip2File(path, parallelization)
            for pos in (int(rng.integers(0, size)), 0, size - 1):
                check_seek(raw, f, pos)
            offsets = f.block_offsets()
            f.close()
            assert f.closed
            for pos in (int(rng.integers(0, size)), 0, size - 1):
                g = native.IndexedBzip2File(path, parallelization=parallelization)
                g.set_block_offsets(offsets)
                check_seek(raw, g, pos)
                g.close()
                assert g.closed
    assert levels == set(range(1, 10)) and encoders == set(ENCODERS) and used_buffer_sizes == set(BUFFER_SIZES)


@pytest.mark.parametrize("size", [512 * 1024 + 2, 1, 2, 4, 128, 1000, 1024, 128 * 1024, 100_000, 200_000, 400_000,
                                  1024 * 1024])
def test_regression_sizes_of_identical_bytes(native, size):
    """testPythonWrappers.py:476-500 (rapidgzip/issues/55): b'A' * size at level 1, through a file object without
    fileno, with parallelization 1, 2 and the default."""
    original = b"A" * size
    compressed = bz2.compress(original, compresslevel=1)
    assert bz2.decompress(compressed) == original
    for opener in (lambda x: native.IndexedBzip2File(x, parallelization=1),
                   lambda x: native.IndexedBzip2File(x, parallelization=2),
                   native.open):
        with opener(io.BytesIO(compressed)) as file:
            decompressed = file.read()
            assert len(decompressed) == size
            assert decompressed == original
