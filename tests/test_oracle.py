"""CPU suite, part 1: pin the oracle (oracle/bz2_oracle.c) against
  - the reference's own fixtures and raw twins (tests/golden/fixtures, from src/tests/data),
  - golden vectors produced by the REAL reference compiled in the authoring container
    (tests/golden/reference_vectors.json, made by tests/golden/make_golden.py from oracle/_ref/ref_bz2),
  - the RUNA/RUNB known-answer table (src/tests/indexed_bzip2/testRunAB.cpp:12-77),
  - the magic-scan known answers (src/tests/core/testBitStringFinder.cpp:119-146),
  - CPython's bz2 module (libbz2) as an independent third decoder,
  - the live reference binary when it is present (oracle/_ref/ref_bz2).
"""
import bz2
import hashlib
import json
import os

import pytest

from conftest import FIXTURES, ROOT, fixture_names, read_fixture
import datagen

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
MAGIC = 0x314159265359


def fnv64(data):
    h = 0xcbf29ce484222325
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def check_against_golden(oracle, enc, gold, raw=None):
    if hashlib.sha256(enc).hexdigest() != gold["enc_sha256"]:
        pytest.skip("local libbz2 produced a different (still valid) compressed stream than the golden run")
    st, out, bmap, tg = oracle.decode_file(enc)
    assert st == 0
    assert [[k, v] for k, v in bmap.items()] == gold["map"]
    if raw is not None:
        assert out == raw
    pos = 0
    for off, size, hcrc, ccrc, dsize, fnv in gold["blocks"]:
        d, payload = oracle.decode_block(enc, off)
        assert (d["status"], d["encoded_size_bits"], d["header_crc"], d["computed_crc"], d["decoded_size"]) == \
               (0, size, hcrc, ccrc, dsize)
        if dsize <= 300_000:
            assert fnv64(payload) == fnv
        assert payload == out[pos:pos + dsize]
        pos += dsize
    assert pos == len(out)


@pytest.mark.parametrize("name", fixture_names())
def test_fixtures_vs_raw_twin_and_reference(oracle, name):
    enc, raw = read_fixture(name)
    check_against_golden(oracle, enc, GOLDEN["fixtures"][name], raw)
    assert bz2.decompress(enc) == raw


def test_1B_index_known_answer(oracle):
    # src/tests/rapidgzip/testParallelGzipReader.cpp:1133: {{4*8,0},{37*8,1}} are the first and last entries
    enc, raw = read_fixture("1B")
    st, out, bmap, tg = oracle.decode_file(enc)
    items = list(bmap.items())
    assert items[0] == (4 * 8, 0) and items[-1] == (37 * 8, 1)


def test_survey_maps(oracle):
    # maps recorded from the compiled reference in SURVEY.md section 4
    want = {
        "1B": {32: 0, 211: 1, 296: 1}, "empty": {0: 0}, "zeros": {32: 0, 253: 1024, 336: 1024},
        "base64-256KiB": {32: 0, 1590568: 262144, 1590648: 262144},
        "random-128KiB": {32: 0, 1056556: 131072, 1056640: 131072},
    }
    for name, m in want.items():
        enc, raw = read_fixture(name)
        assert oracle.decode_file(enc)[2] == m


@pytest.mark.parametrize("name", sorted(GOLDEN["generated"]))
def test_generated_vs_reference(oracle, name):
    raw, level = datagen.corpus_small()[name]
    assert hashlib.sha256(raw).hexdigest() == GOLDEN["generated"][name]["raw_sha256"]
    enc = datagen.compress(raw, level)
    check_against_golden(oracle, enc, GOLDEN["generated"][name], raw)


def test_multistream_and_trailing_garbage_vs_reference(oracle):
    parts = [datagen.text_like(250_000, 31), datagen.random_bytes(150_000, 32), b"", b"x", datagen.runs(99_999, 33)]
    enc = datagen.multistream(parts, 1)
    check_against_golden(oracle, enc, GOLDEN["multistream"]["five-streams-l1"], b"".join(parts))
    garbage = enc + b"\x00" * 37 + datagen.compress(b"must not be read", 9)
    gold = GOLDEN["multistream"]["trailing-garbage"]
    if hashlib.sha256(garbage).hexdigest() == gold["enc_sha256"]:
        st, out, bmap, tg = oracle.decode_file(garbage)
        assert st == 0 and tg and out == b"".join(parts)
        assert [[k, v] for k, v in bmap.items()] == gold["map"]


# reference exception -> status code (include/mi355x_bz2.h)
def expected_status(probe):
    if probe["verdict"] == "OK":
        return 0
    exc, what = probe["exception"], probe["what"]
    if "EndOfFileReached" in exc:
        return 1
    if "bad_optional_access" in exc:
        return 11
    if "Calculated CRC" in what:
        return 15
    if "larger than buffer size" in what:
        return 4
    if "Constructing a Huffman coding" in what or "code length" in what.lower() and "start_huffman" not in what:
        return 9
    if "origPtr error" in what:
        return 14
    if "start_huffman_length" in what:
        return 8
    if "invalid compressed magic" in what:
        return 2
    if "isRandomized" in what:
        return 3
    if "group count" in what:
        return 5
    if "number of selectors" in what:
        return 6
    if "zero termination" in what:
        return 7
    if "selector" in what and "out of maximum range" in what:
        return 10
    if "dbufCount + hh" in what:
        return 12
    if "dbufCount" in what:
        return 13
    raise AssertionError(f"unmapped reference exception: {probe}")


def corrupt_inputs():
    raw = datagen.text_like(200_000, 41)
    enc = datagen.compress(raw, 9)
    cases = {}
    for flip in (len(enc) // 2, len(enc) // 3, 40, 60, 100, 2000, 14, 15, 16, 20, 30):
        b = bytearray(enc)
        b[flip] ^= 0x10
        cases[f"flip-{flip}"] = bytes(b)
    for cut in (len(enc) - 11, len(enc) // 2, 100, 20, 9):
        cases[f"cut-{cut}"] = enc[:cut]
    return cases


def test_error_statuses_vs_reference(oracle):
    checked = 0
    for name, data in corrupt_inputs().items():
        probe = GOLDEN["probes"].get(name)
        if probe is None or hashlib.sha256(data).hexdigest() != probe["enc_sha256"]:
            continue
        d, _ = oracle.decode_block(data, 32)
        assert d["status"] == expected_status(probe), (name, d, probe)
        checked += 1
    if checked == 0:
        pytest.skip("compressed stream differs from the golden run")


def test_runab_known_answers(oracle):
    table = json.load(open(os.path.join(ROOT, "tests", "golden", "runab_table.json")))["table"]
    assert len(table) == 64
    for length in range(1, len(table)):
        digits = [0 if c == "A" else 1 for c in table[length]]
        assert oracle.run_length(digits) == length


def test_magic_scan_known_answers(oracle):
    # src/tests/core/testBitStringFinder.cpp:119-146
    M = bytes([0x31, 0x41, 0x59, 0x26, 0x53, 0x59])
    cases = [
        (bytes([0x11, 0x41, 0x59, 0x26, 0x53, 0x59]), []),
        (bytes([0x31, 0x41, 0x59, 0x26, 0x53, 0x58]), []),
        (M, [0]),
        (M + b"\0\0", [0]),
        (b"\0" + M + b"\0\0", [8]),
        (b"\0\0" + M + b"\0\0", [16]),
        (b"\0\0\0" + M + b"\0\0", [24]),
        (b"\0\0\0\0" + M + b"\0\0", [32]),
        (bytes([0x18, 0xA0, 0xAC, 0x93, 0x29, 0xAC, 0x80]), [1]),
        (bytes([0x00, 0x62, 0x82, 0xB2, 0x4C, 0xA6, 0xB2]), [7]),
    ]
    base = b"\0\0\0\0" + M + b"\0\0"
    for gap in (1, 100, 123, 1024, 2000, 4095, 4096, 28 * 1024, 4 * 1024 * 1024):
        cases.append((base + b"\0" * gap + M, [32, (len(base) + gap) * 8]))
    for data, want in cases:
        assert oracle.find_magic(data, MAGIC) == want


def test_crc_known_answers(oracle):
    # bzip2's CRC of b"" / b"123456789" (CRC-32/BZIP2 check value 0xFC891918)
    assert oracle.crc32(b"") ^ 0xFFFFFFFF == 0
    assert oracle.crc32(b"123456789") ^ 0xFFFFFFFF == 0xFC891918


def test_stream_crc_matches_eos(oracle):
    # serial reader rule (BZ2Reader.hpp:481-484): rotl(stream,1) ^ block, checked against the stored EOS CRC
    raw = datagen.text_like(400_000, 15)
    enc = datagen.compress(raw, 1)
    offs = oracle.find_magic(enc, MAGIC)
    s = 0
    for o in offs:
        d, _ = oracle.decode_block(enc, o)
        s = oracle.lib().orc_stream_crc_combine(s, d["computed_crc"])
    eos = oracle.find_magic(enc, oracle.MAGIC_EOS)
    assert len(eos) == 1
    h = oracle.read_block_header(enc, eos[0])
    assert h["is_eos"] == 1 and h["header_crc"] == s


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_bz2")),
                    reason="compiled reference not present")
def test_live_reference_agrees(oracle, tmp_path):
    """When oracle/_ref/ref_bz2 exists (authoring container, GPU box snapshot) compare against it directly."""
    raw = datagen.runs(700_000, 77) + datagen.random_bytes(100_000, 78)
    enc = datagen.compress(raw, 4)
    p = tmp_path / "live.bz2"
    p.write_bytes(enc)
    ref_map = {int(a): int(b) for a, b in (l.split() for l in oracle.ref_run("map", p, 3).strip().splitlines())}
    st, out, bmap, tg = oracle.decode_file(enc)
    assert st == 0 and out == raw and bmap == ref_map
    for line in oracle.ref_run("blocks", p).strip().splitlines():
        off, size, hcrc, ccrc, dsize, fnv = line.split()
        d, payload = oracle.decode_block(enc, int(off))
        assert (d["encoded_size_bits"], d["header_crc"], d["computed_crc"], d["decoded_size"]) == \
               (int(size), int(hcrc, 16), int(ccrc, 16), int(dsize))


def _check_damaged(decode_block):
    """Every (damaged file, block offset) of tests/golden/reference_vectors.json["damaged"] -- verdicts of the REAL
    reference (oracle/_ref/ref_bz2 probe): status, and the calculated CRC even where the block fails its CRC."""
    corpus = datagen.damaged_corpus()
    checked = 0
    for name, (data, offs) in sorted(corpus.items()):
        gold = GOLDEN["damaged"][name]
        assert hashlib.sha256(data).hexdigest() == gold["enc_sha256"], "generator drifted: rerun tests/golden/make_golden.py"
        assert sorted(int(o) for o in gold["blocks"]) == sorted(offs)
        for off in offs:
            ref = gold["blocks"][str(off)]
            d = decode_block(data, off)
            assert d["status"] == expected_status(ref), (name, off, d, ref)
            if ref["verdict"] == "OK" and not d["is_eos"]:
                assert (d["decoded_size"], d["header_crc"], d["computed_crc"], d["encoded_size_bits"]) == \
                       (ref["decoded"], ref["header_crc"], ref["calc_crc"], ref["size"]), (name, off, d, ref)
            if "Calculated CRC" in ref["what"]:
                words = ref["what"].split()
                assert (int(words[2], 16), int(words[-1], 16)) == (d["computed_crc"], d["header_crc"]), (name, off, d, ref)
            checked += 1
    assert checked == 60


def test_damaged_blocks_vs_reference(oracle):
    _check_damaged(lambda data, off: oracle.decode_block(data, off)[0])


def test_exotic_valid_streams(oracle):
    """Streams from tests/bz2enc.py (20-bit codes, 2..6 tables, declared-but-unused symbols, surplus selectors):
    libbz2 (CPython bz2) and the oracle must both decode them; the real reference accepted the same streams when this
    test was written (oracle/_ref/ref_bz2 probe)."""
    import bz2
    for name, (raw, enc) in sorted(datagen.exotic_streams().items()):
        assert bz2.decompress(enc) == raw, name
        d, payload = oracle.decode_block(enc, 32)
        assert d["status"] == 0 and payload == raw and d["computed_crc"] == d["header_crc"], (name, d)
        st, out, block_map, garbage = oracle.decode_file(enc)
        assert st == 0 and out == raw and not garbage
        if oracle.ref_available():
            import tempfile
            with tempfile.NamedTemporaryFile(suffix=".bz2") as f:
                f.write(enc)
                f.flush()
                assert oracle.ref_run("probe", f.name, 32).startswith("OK"), name


def test_faulty_streams_one_per_throw_site(oracle):
    """tests/bz2enc.py with deliberate violations; expected statuses recorded from the real reference."""
    for name, (enc, status) in sorted(datagen.faulty_streams().items()):
        d, _ = oracle.decode_block(enc, 32)
        assert d["status"] == status, (name, d)
        if oracle.ref_available():
            import tempfile
            with tempfile.NamedTemporaryFile(suffix=".bz2") as f:
                f.write(enc)
                f.flush()
                out = oracle.ref_run("probe", f.name, 32).strip().split(" ", 2)
                assert out[0] == "EXC" and expected_status({"verdict": "EXC", "exception": out[1], "what": out[2]}) == status
