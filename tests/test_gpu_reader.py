"""Reader-level parity on the GPU: mirrors src/tests/indexed_bzip2/testParallelBZ2Reader.cpp (every read/seek/tell
mirrored against the raw bytes, index export/import) and src/tests/testPythonWrappers.py (io interface contract)."""
import io
import os

import pytest

from conftest import fixture_names, read_fixture, FIXTURES
import datagen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def three_block_file(tmp_path_factory):
    """2 MiB of 'A'+rand()%25 text, bzip2 -9 -> 3 blocks (testParallelBZ2Reader.cpp:259-270)."""
    raw = datagen.random_text_file(2 * 1024 * 1024, 16)
    enc = datagen.compress(raw, 9)
    p = tmp_path_factory.mktemp("bz2") / "random-text.bz2"
    p.write_bytes(enc)
    return str(p), raw, enc


class Mirror:
    """Mirrors every call on the reader and on an io.BytesIO of the raw data (testParallelBZ2Reader.cpp:59-137)."""

    def __init__(self, reader, raw):
        self.r = reader
        self.raw = io.BytesIO(raw)
        self.size = len(raw)

    def seek(self, offset, whence=io.SEEK_SET):
        got = self.r.seek(offset, whence)
        want = min(self.raw.seek(max(0, offset) if whence == io.SEEK_SET else offset, whence), self.size)
        self.raw.seek(want)
        assert got == want, (offset, whence, got, want)
        assert self.r.tell() == want

    def read(self, n):
        got = self.r.read(n)
        want = self.raw.read(n)
        assert len(got) == len(want), (n, len(got), len(want))
        assert got == want
        assert self.r.tell() == self.raw.tell()


@pytest.mark.parametrize("parallelization", [1, 2, 3, 8, 0])
def test_first_time_decoding_script(native, three_block_file, parallelization):
    path, raw, enc = three_block_file
    with native.open(path, parallelization) as f:
        m = Mirror(f, raw)
        # testParallelBZ2Reader.cpp:139-203
        m.read(0)
        m.read(1)
        m.read(2)
        m.read(10)
        m.read(100)
        m.read(1000)
        m.read(10000)
        m.read(1000000)        # across block boundaries
        m.read(7 * 1024 * 1024)  # past EOF
        m.seek(0)
        m.read(5 * 1024 * 1024)
        m.seek(2)
        m.read(2)
        m.seek(-2, io.SEEK_END)
        m.read(2)
        m.seek(-2 * 1024 * 1024 + 1, io.SEEK_END)
        m.read(100)
        m.seek(len(raw) + 1000)   # past EOF
        m.read(10)
        m.seek(900000)
        m.read(200000)
        m.seek(100, io.SEEK_CUR)
        m.read(17)
        f.join_threads()          # :205-215
        m.seek(1234567)
        m.read(4321)


def test_block_offsets_roundtrip(native, oracle, three_block_file):
    path, raw, enc = three_block_file
    st, out, want_map, tg = oracle.decode_file(enc)
    assert st == 0 and out == raw and len(want_map) == 5   # 3 blocks + EOS + sentinel
    with native.open(path, 4) as f:
        assert not f.block_offsets_complete()
        assert f.size() == 0
        assert f.read(50000) == raw[:50000]
        avail = f.available_block_offsets()
        assert list(avail.items())[0] == (32, 0)           # testBZ2Reader.cpp:284-285
        offsets = f.block_offsets()
        assert offsets == want_map
        assert f.block_offsets_complete()
        assert f.size() == len(raw)
        assert f.tell_compressed() in offsets
    # import on a fresh reader, then seek without decoding everything (testParallelBZ2Reader.cpp:217-256)
    with native.open(path, 4) as g:
        g.set_block_offsets(offsets)
        assert g.block_offsets_complete()
        assert g.size() == len(raw)
        g.seek(1500000)
        assert g.read(1000) == raw[1500000:1501000]
        assert g.statistics()["blocks_decoded"] <= 3
        assert g.block_offsets() == want_map
        g.seek(0, io.SEEK_END)
        assert g.tell() == len(raw)
        assert g.read(1) == b""
    with native.open(path, 4) as h:
        assert h.read(10) == raw[:10]
        h.set_block_offsets(offsets)                         # after a partial read
        h.seek(-1000, io.SEEK_END)
        assert h.read() == raw[-1000:]
        with pytest.raises(ValueError):
            h.set_block_offsets({})


@pytest.mark.parametrize("name", fixture_names())
def test_fixtures_through_reader(native, oracle, name):
    enc, raw = read_fixture(name)
    st, out, want_map, tg = oracle.decode_file(enc)
    with native.open(os.path.join(FIXTURES, name + ".bz2"), 3) as f:
        assert f.read() == raw
        assert f.block_offsets() == want_map
        assert f.tell() == len(raw)


def test_io_interface_contract(native, tmp_path):
    """testPythonWrappers.py:248-288 on b'Hello\\nWorld!\\n' via path / file object with fileno / BytesIO."""
    raw = b"Hello\nWorld!\n"
    enc = datagen.compress(raw, 9)
    p = tmp_path / "hello.bz2"
    p.write_bytes(enc)
    for P in (1, 2, 3, 8):
        sources = [str(p), open(str(p), "rb"), io.BytesIO(enc)]
        for src in sources:
            f = native.open(src, P)
            assert not f.closed
            assert f.readable() and f.seekable() and not f.writable()
            assert f.tell() == 0
            assert f.read(1) == b"H"
            assert f.tell() == 1
            assert f.peek(1)[:1] == b"e"
            assert f.readline() == b"ello\n"
            assert f.readlines() == [b"World!\n"]
            assert f.read() == b""
            assert f.seek(6) == 6
            b = bytearray(3)
            assert f.readinto(b) == 3 and bytes(b) == b"Wor"
            assert f.seek(-1, io.SEEK_END) == len(raw) - 1
            assert f.read() == b"\n"
            f.close()
            assert f.closed
            with pytest.raises(ValueError):
                f.read(1)
            if hasattr(src, "close"):
                src.close()


def test_multistream_and_trailing_garbage(native, oracle, tmp_path):
    parts = [datagen.text_like(250_000, 31), datagen.random_bytes(150_000, 32), b"", b"x"]
    enc = datagen.multistream(parts, 1)
    st, out, want_map, tg = oracle.decode_file(enc)
    assert st == 0 and out == b"".join(parts) and not tg
    with native.open(io.BytesIO(enc), 4) as f:
        assert f.read() == out
        assert f.block_offsets() == want_map
    garbage = enc + b"\x00" * 37 + datagen.compress(b"must not be read", 9)
    st, out2, want_map2, tg = oracle.decode_file(garbage)
    assert st == 0 and tg and out2 == out
    with native.open(io.BytesIO(garbage), 4) as f:
        assert f.read() == out
        assert f.block_offsets() == want_map2


def test_corrupt_block_surfaces_on_read(native, tmp_path):
    raw = datagen.random_text_file(2 * 1024 * 1024, 16)
    enc = bytearray(datagen.compress(raw, 9))
    enc[len(enc) // 2] ^= 0x55       # inside the second block
    # raw (unbuffered) reader: a 100 kB read stays inside the first block, whose decode is fine although the prefetch
    # of the corrupt second block failed silently (BlockFetcher.hpp:424-432); the failure surfaces when it is needed
    with native.IndexedBzip2FileRaw(io.BytesIO(bytes(enc)), 4) as f:
        assert f.read(100000) == raw[:100000]
        with pytest.raises(native.Bz2Error) as e:
            f.readall()
        assert e.value.status == 15


def _corrupt_stream_crc(enc, oracle):
    """Flip one bit of the combined CRC stored in the (first) end-of-stream block; block CRCs stay valid."""
    eos = oracle.find_magic(enc, oracle.MAGIC_EOS)[0]
    bad = bytearray(enc)
    bit = eos + 48 + 5
    bad[bit >> 3] ^= 0x80 >> (bit & 7)
    return bytes(bad)


def test_stream_crc_is_verified_like_the_serial_reference_reader(native, oracle):
    """BZ2Reader.hpp:406-416 (serial reader, = parallelization 1 in the reference) throws on a wrong stream CRC;
    ParallelBZ2Reader never looks at it.  Same matrix here, plus an explicit switch."""
    parts = [datagen.text_like(300_000, 41), datagen.random_bytes(50_000, 42), b"tail"]
    enc = datagen.multistream(parts, 1)
    raw = b"".join(parts)
    with native.open(io.BytesIO(enc), 1) as f:          # default on
        assert f.read() == raw
        assert f.streams_verified() == 3
    with native.open(io.BytesIO(enc), 4) as f:          # default off
        assert f.read() == raw
        assert f.streams_verified() == 0
    bad = _corrupt_stream_crc(enc, oracle)
    with native.open(io.BytesIO(bad), 4) as f:          # parallel reader semantics: not noticed
        assert f.read() == raw
    with native.open(io.BytesIO(bad), 4) as f:
        f.set_verify_stream_crc(True)
        with pytest.raises(native.Bz2Error) as e:
            f.read()
        assert e.value.status == 17 and "Stream CRC" in str(e.value)
    with native.IndexedBzip2FileRaw(io.BytesIO(bad), 1) as f:
        with pytest.raises(native.Bz2Error) as e:
            f.readall()
        assert e.value.status == 17
    with native.open(io.BytesIO(bad), 1) as f:
        f.set_verify_stream_crc(False)
        assert f.read() == raw


def test_text_index_roundtrip(native, oracle, three_block_file, tmp_path):
    """The `-L` text format (ibzip2.cpp:83-93) as an on-disk index: export, import into a fresh reader, seek."""
    path, raw, enc = three_block_file
    st, out, want_map, tg = oracle.decode_file(enc)
    index = tmp_path / "index.csv"
    with native.open(path, 4) as f:
        native.write_block_offsets(f.block_offsets(), str(index))
    assert index.read_text() == "".join(f"{k},{v}\n" for k, v in sorted(want_map.items()))
    with native.open(path, 4) as g:
        g.set_block_offsets(native.read_block_offsets(str(index)))
        g.seek(1_700_000)
        assert g.read(5000) == raw[1_700_000:1_705_000]
        assert g.statistics()["blocks_decoded"] <= 3
    with pytest.raises(ValueError):
        native.read_block_offsets(io.StringIO("12;5\n"))


@pytest.mark.parametrize("parallelization", [64, 100, 700])
def test_many_batches_in_flight(native, oracle, parallelization):
    """Several contexts and a prefetch window of (contexts + 2) P: a file of ~285 small blocks read sequentially goes
    through many overlapping batches (three contexts at P = 64 and 100, two at 700 with the file shorter than the
    window); block magics come from the GPU scan.  Every byte, the block map and the stream CRCs must come out as with
    one block at a time."""
    parts = [datagen.text_like(9_000_000, 81), datagen.random_bytes(3_000_000, 82), datagen.runs(4_000_000, 83),
             datagen.random_text_file(16_000_000, 84)]
    raw = b"".join(parts)
    enc = datagen.multistream(parts, 1)
    want_offsets = oracle.find_magic(enc, oracle.MAGIC_BLOCK)
    assert len(want_offsets) > 250
    with native.open(io.BytesIO(enc), parallelization) as f:
        f.set_verify_stream_crc(True)
        got = bytearray()
        while True:
            piece = f.read(3_000_001)
            if not piece:
                break
            got += piece
        assert bytes(got) == raw
        assert f.streams_verified() == len(parts)
        offsets = f.block_offsets()
        st = f.statistics()
        assert st["batches"] >= (2 if parallelization < 200 else 1) and st["failed_prefetches"] == 0
        # data blocks of the map are exactly the magics found (+ EOS entries and the end), sizes add up
        assert [o for o in sorted(offsets) if o in set(want_offsets)] == want_offsets
        assert max(offsets.values()) == len(raw)
        # backwards seek into an evicted region, then on to the end
        f.seek(len(raw) // 3)
        assert f.read(100_000) == raw[len(raw) // 3:len(raw) // 3 + 100_000]
        f.seek(-5, io.SEEK_END)
        assert f.read() == raw[-5:]
