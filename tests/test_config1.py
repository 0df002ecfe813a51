"""BASELINE config 1 / SURVEY 8(d): 10 000 000 bytes of seeded wiki-style text (tests/datagen.enwik_like, committed word
list) at `bzip2 -1` -- about 100 blocks of 100 kB -- read with parallelization=1.

CPU half ("plumbing + bit-exact check, no GPU", BASELINE.json configs[0]): the oracle decodes the file to the raw bytes,
agrees with libbz2, and -- where the real reference is built (oracle/_ref/ref_bz2) -- with the reference's serial and
parallel block maps and per-block records.
GPU half: indexed_bzip2_amd.open(path, parallelization=1): sha256 of the output, block map == oracle map, stream CRC
verified (parallelization 1 = the reference's serial reader, which checks it), every block record == the oracle's.
"""
import bz2
import hashlib
import os
import subprocess

import pytest

import datagen

RAW_BYTES = 10_000_000


@pytest.fixture(scope="module")
def config1(tmp_path_factory):
    raw = datagen.enwik_like(RAW_BYTES)
    assert len(raw) == RAW_BYTES
    directory = tmp_path_factory.mktemp("config1")
    raw_path = directory / "enwik-like"
    raw_path.write_bytes(raw)
    if os.path.exists("/usr/bin/bzip2"):
        subprocess.run(["/usr/bin/bzip2", "-1", "-k", str(raw_path)], check=True)   # the encoder BASELINE names
        path = str(raw_path) + ".bz2"
        enc = open(path, "rb").read()
    else:
        enc = bz2.compress(raw, 1)
        path = str(raw_path) + ".bz2"
        open(path, "wb").write(enc)
    return path, raw, enc


def test_config1_oracle_plumbing(oracle, config1):
    path, raw, enc = config1
    assert bz2.decompress(enc) == raw
    status, out, offsets, trailing = oracle.decode_file(enc)
    assert status == 0 and not trailing
    assert hashlib.sha256(out).hexdigest() == hashlib.sha256(raw).hexdigest()
    data_blocks = [oracle.decode_block(enc, o)[0] for o in oracle.find_magic(enc)]
    assert len(data_blocks) >= 100
    assert all(b["status"] == 0 and b["computed_crc"] == b["header_crc"] for b in data_blocks)
    # map: first entry {32: 0}, EOS entry and sentinel at the decoded size
    keys = sorted(offsets)
    assert keys[0] == 32 and offsets[32] == 0
    assert offsets[keys[-1]] == offsets[keys[-2]] == RAW_BYTES and keys[-1] == len(enc) * 8
    if oracle.ref_available():
        # the real reference at parallelization 1 (serial BZ2Reader) and 2 (ParallelBZ2Reader)
        # (the serial reader's map has no sentinel entry behind the end-of-stream block)
        serial = {k: v for k, v in offsets.items() if k != keys[-1]}
        for cmd, want in ((("smap", path), serial), (("map", path, 2), offsets)):
            lines = oracle.ref_run(*cmd).strip().splitlines()
            ref_map = {int(a): int(b) for a, b in (l.split() for l in lines)}
            assert ref_map == want, cmd
        ref_blocks = [l.split() for l in oracle.ref_run("blocks", path).strip().splitlines()]
        assert len(ref_blocks) == len(data_blocks)
        for want, got in zip(ref_blocks, data_blocks):
            assert (int(want[0]), int(want[1]), int(want[2], 16), int(want[3], 16), int(want[4])) == \
                (got["encoded_offset_bits"], got["encoded_size_bits"], got["header_crc"], got["computed_crc"],
                 got["decoded_size"])


@pytest.mark.gpu
def test_config1_gpu_parallelization_1(native, oracle, config1):
    path, raw, enc = config1
    want_sha = hashlib.sha256(raw).hexdigest()
    offsets = oracle.decode_file(enc)[2]
    with native.open(path) as f:                       # parallelization defaults to 1, as in the reference
        sha = hashlib.sha256()
        while True:
            piece = f.read(1 << 20)
            if not piece:
                break
            sha.update(piece)
        assert sha.hexdigest() == want_sha
        assert f.tell() == RAW_BYTES and f.size() == RAW_BYTES
        assert f.block_offsets() == offsets
        assert f.streams_verified() == 1               # serial-reader semantics: the stream CRC was checked
    with native.open(path, parallelization=1) as f:
        assert f.block_offsets() == offsets            # index creation without reading through read()
        f.seek(7_654_321)
        assert f.read(100_000) == raw[7_654_321:7_754_321]
    # every block's record, through the batch C ABI, against the oracle's
    data_blocks = [oracle.decode_block(enc, o)[0] for o in oracle.find_magic(enc)]
    dec = native.Decoder()
    dec.set_input(enc)
    results, total = dec.decode_batch([b["encoded_offset_bits"] for b in data_blocks])
    assert total == RAW_BYTES
    out = dec.copy_output(0, total)
    assert hashlib.sha256(out).hexdigest() == want_sha
    for got, want in zip(results, data_blocks):
        for key in ("encoded_offset_bits", "encoded_size_bits", "decoded_size", "header_crc", "computed_crc",
                    "bwt_length", "orig_ptr", "n_symbols", "is_eos", "is_eof", "status"):
            assert got[key] == want[key], (key, got[key], want[key])
    dec.close()
