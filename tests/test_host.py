"""CPU suite, part 2: host logic of the product library (no GPU): every symbol include/mi355x_bz2.h declares is
exported, the magic scan and stream-header parse agree with the oracle / known answers, and the product path fails
loudly without a GPU (no CPU fallback)."""
import io
import os
import re

import pytest

from conftest import ROOT, fixture_names, read_fixture
import datagen

MAGIC = 0x314159265359


def test_library_exports_every_declared_symbol(native):
    header = open(os.path.join(ROOT, "include", "mi355x_bz2.h")).read()
    declared = set(re.findall(r"\b(mi355x_bz2_[a-z0-9_]+)\s*\(", header))
    declared -= {"mi355x_bz2_status"}
    bound = {name for name, _, _ in native._native.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    L = native.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.mi355x_bz2_abi_version() == 2
    assert native.status_string(0) == "OK" and "CRC" in native.status_string(15)


def test_no_gpu_means_loud_failure(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.Bz2Error) as e:
        native.Decoder()
    assert e.value.status == 102
    with pytest.raises(native.Bz2Error):
        native.open(io.BytesIO(datagen.compress(b"abc", 9))).read()


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "indexed_bzip2_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "bz2_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
    text = open(os.path.join(ROOT, "include", "mi355x_bz2.h")).read()
    assert "oracle" not in text


@pytest.mark.parametrize("threads", [1, 2, 5])
def test_find_magic_matches_oracle(native, oracle, threads):
    M = bytes([0x31, 0x41, 0x59, 0x26, 0x53, 0x59])
    cases = [b"", M[:5], M, b"\0" + M, bytes([0x18, 0xA0, 0xAC, 0x93, 0x29, 0xAC, 0x80]),
             bytes([0x00, 0x62, 0x82, 0xB2, 0x4C, 0xA6, 0xB2]),
             bytes([0x31, 0x41, 0x59, 0x26, 0x53, 0x58])]
    base = b"\0\0\0\0" + M + b"\0\0"
    for gap in (1, 100, 123, 1024, 4095, 4096, 28 * 1024, 1 << 20, (1 << 20) - 3, 4 * 1024 * 1024):
        cases.append(base + b"\0" * gap + M)
    # every bit phase, straddling the scanner's 1 MiB sub-chunk and the thread chunk boundaries
    big = bytearray(9 * (1 << 20))
    want = []
    for k, byte_pos in enumerate([0, 1000, (1 << 20) - 3, (1 << 20) + 9, 2 * (1 << 20) - 6, 3 * (1 << 20) + 5,
                                  (4 << 20) - 4, (4 << 20) + 11, 9 * (1 << 20) - 7]):
        s = k % 8
        v = (MAGIC << (8 - s)).to_bytes(7, "big")
        for i, b in enumerate(v):
            if byte_pos + i < len(big):
                big[byte_pos + i] |= b
        want.append(byte_pos * 8 + s)
    cases.append(bytes(big))
    for data in cases:
        assert native.find_magic(data, MAGIC, threads) == oracle.find_magic(data, MAGIC)
    assert native.find_magic(bytes(big), MAGIC, threads) == sorted(want)
    for name in fixture_names():
        enc, _ = read_fixture(name)
        assert native.find_magic(enc, MAGIC, threads) == oracle.find_magic(enc, MAGIC)
        assert native.find_magic(enc, oracle.MAGIC_EOS, threads) == oracle.find_magic(enc, oracle.MAGIC_EOS)


def test_stream_header(native, oracle):
    L = native.lib()
    for lvl in range(1, 10):
        enc = datagen.compress(b"hello", lvl)
        assert L.mi355x_bz2_read_stream_header(enc, len(enc), 0) == lvl == oracle.lib().orc_read_stream_header(enc, len(enc), 0)
    for bad in (b"BZh0xxxx", b"BZi9xxxx", b"BZh", b"", b"XZh9"):
        assert L.mi355x_bz2_read_stream_header(bad, len(bad), 0) == 0
        assert oracle.lib().orc_read_stream_header(bad, len(bad), 0) == 0
    # second stream of a concatenated file starts right after the padded EOS
    enc = datagen.multistream([b"a" * 1000, b"b" * 1000], 5)
    eos = oracle.find_magic(enc, oracle.MAGIC_EOS)
    h = oracle.read_block_header(enc, eos[0])
    nxt = h["encoded_offset_bits"] + h["encoded_size_bits"]
    assert L.mi355x_bz2_read_stream_header(enc, len(enc), nxt) == 5


def test_text_index_helpers(native, tmp_path):
    """`-L` text format of the reference CLI (ibzip2.cpp:83-93): '<compressed bits>,<decoded bytes>' per line."""
    offsets = {32: 0, 1845: 900000, 99999: 1800000, 100047: 1800000}
    p = tmp_path / "map.csv"
    native.write_block_offsets(offsets, str(p))
    assert p.read_text() == "32,0\n1845,900000\n99999,1800000\n100047,1800000\n"
    assert native.read_block_offsets(str(p)) == offsets
    import io
    buf = io.StringIO()
    native.write_block_offsets(offsets, buf)
    assert native.read_block_offsets(io.BytesIO(buf.getvalue().encode())) == offsets
    with pytest.raises(ValueError):
        native.read_block_offsets(io.StringIO("1,2,3\n"))
