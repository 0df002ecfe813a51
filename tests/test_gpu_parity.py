"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on identical inputs. Bit-exact.

Run on the GPU box: python -m pytest tests -m gpu -x -q
"""
import hashlib
import os

import pytest

from conftest import fixture_names, read_fixture
import datagen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["stages-kept", "as-shipped"])
def dec(native, request):
    """Every test that takes `dec` runs twice: on a context that keeps the per-stage buffers addressable (the L column and the
    inverse BWT's bytes are compared with the oracle's) and on one made the way the reader and the bench make theirs (the two
    share their memory: round 3 broke damaged blocks there, and nothing that only ran on the first kind could notice)."""
    keep = request.param == "stages-kept"
    d = native.Decoder(flags=native.Decoder.KEEP_STAGES if keep else 0)
    d.keeps_stages = keep
    yield d
    d.close()


def check_blocks(native, oracle, dec, enc, expect_raw=None, check_stages=True):
    check_stages = check_stages and getattr(dec, "keeps_stages", True)
    offs = oracle.find_magic(enc)
    assert native.find_magic(enc) == offs
    dec.set_input(enc)
    results, total = dec.decode_batch(offs)
    out = dec.copy_output(0, total)
    pos = 0
    for i, (o, r) in enumerate(zip(offs, results)):
        if check_stages:
            od, payload, lcol, rle = oracle.decode_block(enc, o, want_stages=True)
        else:
            od, payload = oracle.decode_block(enc, o)
        for key in ("encoded_offset_bits", "encoded_size_bits", "decoded_size", "header_crc", "computed_crc",
                    "bwt_length", "orig_ptr", "n_symbols", "is_eos", "is_eof", "status"):
            assert r[key] == od[key], f"block {i} @bit {o}: {key}: gpu {r[key]} != oracle {od[key]}"
        if od["status"] == 0:
            if check_stages:
                assert dec.debug_stage(i, 0) == lcol, f"block {i}: L column differs"
                assert dec.debug_stage(i, 2) == rle, f"block {i}: inverse-BWT output differs"
            assert r["data_offset"] == pos
            got = out[pos:pos + r["decoded_size"]]
            assert hashlib.sha256(got).digest() == hashlib.sha256(payload).digest(), f"block {i}: payload differs"
            pos += r["decoded_size"]
    assert pos == total
    if expect_raw is not None:
        assert out == expect_raw
    return results


@pytest.mark.parametrize("name", fixture_names())
def test_reference_fixtures(native, oracle, dec, name):
    enc, raw = read_fixture(name)
    check_blocks(native, oracle, dec, enc, raw)


@pytest.mark.parametrize("name", sorted(datagen.corpus_small()))
def test_generated(native, oracle, dec, name):
    raw, level = datagen.corpus_small()[name]
    enc = datagen.compress(raw, level)
    check_blocks(native, oracle, dec, enc, raw)


def test_silesia_style_slice(native, oracle, dec):
    """24 MB of the benchmark corpus (all segment types: long Huffman codes, high-entropy blocks, long runs) as one
    stitched single-stream file: every block bit-exact against the oracle, stages included."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import silesia_like, bz2build
    data = silesia_like.generate(24_000_000, threads=8)
    enc, nb, offs = bz2build.build(data, 1, piece_size=6_000_000, threads=8, find_magic=native.find_magic)
    assert nb >= 24
    check_blocks(native, oracle, dec, enc, data.tobytes())
    # and the stream CRC stored in the stitched file equals the combination of the block CRCs (bz2build + GPU)
    assert oracle.decode_file(enc)[0] == 0


@pytest.mark.parametrize("copies", [10, 31])
def test_batches_beside_other_contexts(native, oracle, copies):
    """Three or more contexts alive on a device make the launcher choose its kernels for a crowd (workgroups and claims of the
    walk, waves per block in the scan, lanes per block in k_mtf, slices of the table build -- by batch size): batches of about 280
    and 870 blocks of the benchmark corpus on a context as shipped, with three others alive beside it.  (bench.py checks
    every block's CRC in that situation; here the records and the bytes are compared with the oracle's.)"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import silesia_like, bz2build
    data = silesia_like.generate(24_000_000, threads=8)
    enc, nb, offs = bz2build.build(data, 1, piece_size=6_000_000, threads=8, find_magic=native.find_magic)
    blocks = oracle.find_magic(enc)
    assert len(blocks) >= 24
    others = [native.Decoder() for _ in range(3)]
    d = native.Decoder()
    try:
        offsets = [o for _ in range(copies) for o in blocks]
        d.set_input(enc)
        results, total = d.decode_batch(offsets)
        out = d.copy_output(0, total)
        want = {o: oracle.decode_block(enc, o) for o in blocks}
        pos = 0
        for o, r in zip(offsets, results):
            od, payload = want[o]
            for key in ("encoded_offset_bits", "encoded_size_bits", "decoded_size", "header_crc", "computed_crc",
                        "bwt_length", "orig_ptr", "n_symbols", "is_eos", "is_eof", "status"):
                assert r[key] == od[key], (o, key)
            assert r["status"] == 0 and r["data_offset"] == pos
            assert out[pos:pos + r["decoded_size"]] == payload, o
            pos += r["decoded_size"]
        assert pos == total == copies * len(data)
    finally:
        d.close()
        for other in others:
            other.close()


def test_gpu_magic_scan(native, oracle, dec):
    """k_find_magic against the oracle / the known answers of src/tests/core/testBitStringFinder.cpp:119-146."""
    M = bytes([0x31, 0x41, 0x59, 0x26, 0x53, 0x59])
    cases = [M, b"\0" + M, b"\0\0\0\0" + M + b"\0\0", bytes([0x18, 0xA0, 0xAC, 0x93, 0x29, 0xAC, 0x80]),
             bytes([0x00, 0x62, 0x82, 0xB2, 0x4C, 0xA6, 0xB2]), bytes([0x31, 0x41, 0x59, 0x26, 0x53, 0x58]), M[:5] + b"\0" * 9]
    base = b"\0\0\0\0" + M + b"\0\0"
    for gap in (1, 100, 123, 1024, 4095, 4096, 28 * 1024, 1 << 20):
        cases.append(base + b"\0" * gap + M)
    for name in fixture_names():
        cases.append(read_fixture(name)[0])
    cases.append(datagen.multistream([datagen.text_like(250_000, 31), datagen.random_bytes(150_000, 32)], 1))
    for data in cases:
        dec.set_input(data)
        for magic in (oracle.MAGIC_BLOCK, oracle.MAGIC_EOS):
            assert dec.find_magic(magic) == oracle.find_magic(data, magic), (len(data), hex(magic))


def test_multistream(native, oracle, dec):
    parts = [datagen.text_like(250_000, 31), datagen.random_bytes(150_000, 32), b"", b"x", datagen.runs(99_999, 33)]
    enc = datagen.multistream(parts, 1)
    check_blocks(native, oracle, dec, enc, b"".join(parts))


def test_eos_offset_and_errors(native, oracle, dec):
    enc, raw = read_fixture("base64-32KiB")
    eos = oracle.find_magic(enc, oracle.MAGIC_EOS)
    assert len(eos) == 1
    dec.set_input(enc)
    # EOS offset, a non-magic offset, and an offset past the end
    offs = [eos[0], 33, len(enc) * 8 + 5, 32]
    results, total = dec.decode_batch(offs)
    for o, r in zip(offs, results):
        od, payload = oracle.decode_block(enc, o)
        for key in ("encoded_size_bits", "decoded_size", "header_crc", "is_eos", "is_eof", "status"):
            assert r[key] == od[key], (o, key, r[key], od[key])
    assert results[0]["is_eos"] == 1 and results[0]["is_eof"] == 1
    assert results[1]["status"] == 2 and results[2]["status"] == 1
    assert dec.copy_output(results[3]["data_offset"], results[3]["decoded_size"]) == raw


def test_corrupt_crc_and_truncation(native, oracle, dec):
    raw = datagen.text_like(200_000, 41)
    enc = bytearray(datagen.compress(raw, 9))
    # flip a bit in the middle of the Huffman data: CRC mismatch or a structural error -- must agree with the oracle
    for flip in (len(enc) // 2, len(enc) // 3, 40, 60, 100, 2000):
        bad = bytearray(enc)
        bad[flip] ^= 0x10
        bad = bytes(bad)
        dec.set_input(bad)
        results, total = dec.decode_batch([32])
        od, _ = oracle.decode_block(bad, 32)
        assert results[0]["status"] == od["status"], (flip, results[0], od)
    for cut in (len(enc) - 11, len(enc) // 2, 100, 20, 9):
        bad = bytes(enc[:cut])
        dec.set_input(bad)
        results, total = dec.decode_batch([32])
        od, _ = oracle.decode_block(bad, 32)
        assert results[0]["status"] == od["status"], (cut, results[0], od)


def test_begin_end_halves_and_two_contexts(native, oracle):
    """mi355x_bz2_decode_batch_begin/_end: same results as the one-call form, one batch in flight per context, two
    contexts interleaved (what bench.py does to overlap consecutive batches)."""
    import numpy as np
    parts = [datagen.text_like(1_200_000, 51), datagen.random_bytes(700_000, 52)]
    encs = [datagen.compress(p, 9) for p in parts]
    decs, arrays, wants = [], [], []
    for enc, raw in zip(encs, parts):
        d = native.Decoder()
        d.set_input(enc)
        offs = native.find_magic(enc)
        decs.append(d)
        arrays.append(d.make_arrays(offs) + (len(offs),))
        wants.append((d.decode_batch(offs), raw))
    for _ in range(2):
        decs[0].begin_batch(arrays[0][0], arrays[0][2])
        with pytest.raises(native.Bz2Error):
            decs[0].begin_batch(arrays[0][0], arrays[0][2])        # a batch is already in flight on this context
        decs[1].begin_batch(arrays[1][0], arrays[1][2])
        for i in (0, 1):
            total = decs[i].end_batch(arrays[i][1])
            (want_results, want_total), raw = wants[i]
            assert total == want_total == len(raw)
            got = [arrays[i][1][k].as_dict() for k in range(arrays[i][2])]
            assert got == want_results
            assert decs[i].copy_output(0, total) == raw
        assert decs[0].end_batch(arrays[0][1]) == 0                  # nothing in flight: empty batch
    for d in decs:
        d.close()


def test_device_memory_and_the_stage_buffers(native, oracle):
    """mi355x_bz2_device_memory; a context made without KEEP_STAGES lets the inverse BWT's bytes take the place of the last
    column (0.9 MB per block less) and refuses to show the two, one made with the flag keeps both."""
    raw = datagen.text_like(1_200_000, 77)
    enc = datagen.compress(raw, 9)
    offs = native.find_magic(enc)
    plain, keeping = native.Decoder(), native.Decoder(flags=native.Decoder.KEEP_STAGES)
    try:
        assert plain.device_memory()["output_bytes"] == 0          # (scratch for a first small batch comes with the context)
        for d in (plain, keeping):
            d.set_input(enc)
            results, total = d.decode_batch(offs)
            assert total == len(raw) and d.copy_output(0, total) == raw
        small, large = plain.device_memory(), keeping.device_memory()
        assert 0 < small["scratch_bytes"] < large["scratch_bytes"]
        assert small["output_bytes"] >= len(raw) and large["output_bytes"] >= len(raw)
        # one more buffer of 900 096 bytes per block slot
        more = large["scratch_bytes"] - small["scratch_bytes"]
        slots = round(more / 900_096)
        assert slots >= 8 and abs(more - slots * 900_096) <= 512
        lcol = oracle.decode_block(enc, offs[0], want_stages=True)[2]
        assert keeping.debug_stage(0, 0) == lcol
        for stage in (0, 2):
            with pytest.raises(native.Bz2Error):
                plain.debug_stage(0, stage)
        assert len(plain.debug_stage(0, 1)) == 4 * len(lcol)       # the table is its own buffer either way
        # the same records from both for damaged blocks (a walk that closes early lays its period out again through a
        # buffer of its own: it must not be one of the two that share their memory)
        keys = ("encoded_size_bits", "decoded_size", "computed_crc", "bwt_length", "status")
        for name, (bad, offs_bad) in datagen.damaged_corpus().items():
            if not offs_bad:
                continue
            got = []
            for d in (plain, keeping):
                d.set_input(bad)
                results_bad, total_bad = d.decode_batch(offs_bad)
                got.append(([{k: r[k] for k in keys} for r in results_bad], d.copy_output(0, total_bad)))
            assert got[0] == got[1], name
    finally:
        plain.close()
        keeping.close()


def test_two_contexts_over_one_resident_input(native, oracle):
    """mi355x_bz2_share_input: the second context decodes from the bytes the first one uploaded (what the reader does
    for its two contexts); halves of the block list on each, interleaved."""
    raw = datagen.text_like(2_000_000, 53) + datagen.random_bytes(500_000, 54)
    enc = datagen.compress(raw, 9)
    offs = native.find_magic(enc)
    a = native.Decoder()
    a.set_input(enc)
    b = native.Decoder()
    with pytest.raises(native.Bz2Error):
        b.share_input(native.Decoder())          # nothing resident there
    b.share_input(a)
    want, total = a.decode_batch(offs)
    assert a.copy_output(0, total) == raw
    got, total_b = b.decode_batch(offs)
    assert (got, total_b) == (want, total) and b.copy_output(0, total_b) == raw
    half = len(offs) // 2
    arr_a, arr_b = a.make_arrays(offs[:half]), b.make_arrays(offs[half:])
    a.begin_batch(arr_a[0], half)
    b.begin_batch(arr_b[0], len(offs) - half)
    ta, tb = a.end_batch(arr_a[1]), b.end_batch(arr_b[1])
    assert a.copy_output(0, ta) + b.copy_output(0, tb) == raw
    assert b.find_magic() == offs                 # the scan sees the shared bytes too
    b.close()
    a.close()


def test_background_copy_overlaps_the_next_batch(native, oracle):
    """mi355x_bz2_copy_output_begin/_end (the reader's per-context loop): the bytes of batch k are copied while batch k + 1
    -- of a different size, so that the second output buffer is allocated and later grows -- is decoded; afterwards
    both are intact, the output pointer names the batch finished last, and growing buffers change nothing."""
    parts = [datagen.text_like(700_000, 61), datagen.random_bytes(1_900_000, 62), datagen.text_like(3_100_000, 63)]
    raw = b"".join(parts)
    enc = b"".join(datagen.compress(p, 9) for p in parts)
    offs = native.find_magic(enc)
    d = native.Decoder()
    d.set_input(enc)
    want, total = d.decode_batch(offs)
    assert d.copy_output(0, total) == raw
    cuts = [1, 3, len(offs)]                      # growing batches: 1 block, 3 blocks, all
    pending = None
    for round_ in range(2):
        for n in cuts:
            results, size = d.decode_batch(offs[:n])
            assert results == want[:n]
            if pending is not None:
                d.copy_output_end()
                assert bytes(pending[0])[:pending[1]] == raw[:pending[1]]
            assert d.copy_output(0, size) == raw[:size]           # the synchronous form reads the same batch
            pending = (d.copy_output_begin(0, size), size)
            assert d.output_device_ptr() != 0
    d.copy_output_end()
    assert bytes(pending[0])[:pending[1]] == raw[:pending[1]]
    with pytest.raises(native.Bz2Error):
        d.copy_output_begin(0, total + 1)          # beyond the last batch's bytes
    arrays = d.make_arrays(offs)
    d.begin_batch(arrays[0], len(offs))
    with pytest.raises(native.Bz2Error):
        d.copy_output_begin(0, 1)                  # a batch is in flight: its bytes are not there yet
    assert d.end_batch(arrays[1]) == total
    d.close()


def test_next_input_is_copied_while_a_batch_is_in_flight(native, oracle):
    """mi355x_bz2_set_input_host_async during a batch (what bench.py does every step): the bytes of the NEXT batch go
    into the context's second input buffer and leave the batch in flight alone; batches alternate between two different
    files, of different sizes so that both buffers are allocated and one of them grows."""
    import ctypes
    raws = [datagen.text_like(1_500_000, 71) + datagen.random_bytes(300_000, 72), datagen.random_bytes(900_000, 73),
            datagen.text_like(2_600_000, 74)]
    encs = [datagen.compress(r, 9) for r in raws]
    offs = [native.find_magic(e) for e in encs]
    pinned = [(ctypes.c_ubyte * len(e)).from_buffer_copy(e) for e in encs]      # stays alive and in place
    d = native.Decoder()
    arrays = [d.make_arrays(o) for o in offs]
    d.set_input_host_async(ctypes.addressof(pinned[0]), len(encs[0]), keepalive=pinned[0])
    order = [0, 1, 2, 1, 0, 2, 2, 0]
    for step, which in enumerate(order):
        d.begin_batch(arrays[which][0], len(offs[which]))
        if step + 1 < len(order):
            nxt = order[step + 1]
            d.set_input_host_async(ctypes.addressof(pinned[nxt]), len(encs[nxt]), keepalive=pinned[nxt])
        total = d.end_batch(arrays[which][1])
        assert total == len(raws[which]), (step, which)
        assert all(arrays[which][1][k].status == 0 for k in range(len(offs[which]))), (step, which)
        assert d.copy_output(0, total) == raws[which], (step, which)
    d.close()


def test_hold_output_until_an_event(native, oracle):
    """mi355x_bz2_hold_output_until (bench.py: an RCCL send reads the decoded extent on the device): the next batch's
    output kernels are ordered behind the caller's event; results are unchanged, the call is refused while a batch is in
    flight.  The event is a plain hipEvent_t made through the HIP runtime the library itself uses."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = datagen.text_like(1_400_000, 81) + datagen.random_bytes(400_000, 82)
    enc = datagen.compress(raw, 9)
    offs = native.find_magic(enc)
    d = native.Decoder()
    d.set_input(enc)
    want, total = d.decode_batch(offs)
    event = ctypes.c_void_p()
    assert hip.hipEventCreate(ctypes.byref(event)) == 0
    for _ in range(3):
        assert hip.hipEventRecord(event, None) == 0          # "somebody has read the output" on the null stream
        d.hold_output_until(event.value, keepalive=event)
        got, total2 = d.decode_batch(offs)
        assert (got, total2) == (want, total)
        assert d.copy_output(0, total) == raw
    arrays = d.make_arrays(offs)
    d.begin_batch(arrays[0], len(offs))
    with pytest.raises(native.Bz2Error):
        d.hold_output_until(event.value)
    d.end_batch(arrays[1])
    d.close()
    assert hip.hipEventDestroy(event) == 0


def test_warmup(native):
    """mi355x_bz2_warmup: optional, idempotent, and says so if the device does not exist."""
    native.warmup(0, background=False)
    native.warmup(0, background=False)
    thread = native.warmup(0)
    thread.join()
    with pytest.raises(native.Bz2Error):
        native.warmup(4096, background=False)


def test_crc32_of_device_pieces(native, oracle, dec):
    """mi355x_bz2_crc32_device: bzip2's CRC-32 (bzip2.hpp:59-91, 833, 901) of consecutive pieces of a device buffer --
    what rank 0 of the bench runs over the extents it receives.  Against the oracle's updateCRC32 on the same bytes, for
    pieces of awkward sizes (empty, one byte, across the kernel's 64 KiB tiles), and against the block records."""
    enc, raw = read_fixture("base64-256KiB")
    offs = oracle.find_magic(enc)
    dec.set_input(enc)
    results, total = dec.decode_batch(offs)
    assert total == len(raw)
    base = dec.output_device_ptr()
    # the blocks themselves: the receiver's view of a sender's extent
    sizes = [r["decoded_size"] for r in results]
    assert dec.crc32_device(base, sizes) == [r["computed_crc"] for r in results] == [r["header_crc"] for r in results]
    # arbitrary pieces
    sizes = [0, 1, 15, 16, 17, 255, 65535, 65536, 65537, 3, 0, 10000]
    sizes.append(len(raw) - sum(sizes))
    want, at = [], 0
    for n in sizes:
        want.append(oracle.crc32(raw[at:at + n]) ^ 0xFFFFFFFF)
        at += n
    assert dec.crc32_device(base, sizes) == want
    with pytest.raises(native.Bz2Error):
        dec.crc32_device(base + 4, [16])           # not 16-byte aligned
    assert dec.crc32_device(base, []) == []


@pytest.mark.parametrize("symbols", [1, 2, 15, 16, 17, 127, 128, 129, 255, 256])
def test_alphabet_sizes_around_the_list_variants(native, oracle, dec, symbols):
    """k_mtf runs in two instances (128-entry and 256-entry lists) chosen by the block's symbol count; 16-entry groups
    inside a list.  Data with exactly `symbols` distinct byte values, skewed so that deep list positions occur."""
    import numpy as np
    rng = np.random.default_rng(1000 + symbols)
    alphabet = rng.permutation(256)[:symbols].astype(np.uint8)
    weights = 1.0 / np.arange(1, symbols + 1) ** 1.1
    data = alphabet[rng.choice(symbols, size=260_000, p=weights / weights.sum())]
    data[100_000:100_000 + symbols] = alphabet            # every value really occurs
    raw = data.tobytes()
    for level in (1, 9):
        check_blocks(native, oracle, dec, datagen.compress(raw, level), raw)


def test_randomized_inputs(native, oracle, dec):
    """Seeded random structure: size, alphabet, run lengths, periodic stretches, level; every block record, the L column,
    the pre-RLE1 stream and the payload against the oracle."""
    import numpy as np
    rng = np.random.default_rng(0xB21F)
    for case in range(24):
        n = int(rng.integers(1, 400_000))
        kind = case % 4
        if kind == 0:      # random bytes over a random alphabet
            k = int(rng.integers(1, 257))
            raw = rng.integers(0, k, n, dtype=np.uint8).tobytes()
        elif kind == 1:    # runs of random lengths (exercises RLE1 counts and RUNA/RUNB)
            lengths = rng.geometric(0.02, size=n // 20 + 1)
            values = rng.integers(0, 256, lengths.size, dtype=np.uint8)
            raw = np.repeat(values, lengths)[:n].tobytes()
        elif kind == 2:    # periodic with a random period (LF permutation splits into cycles)
            period = int(rng.integers(1, 50))
            unit = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
            raw = (unit * (n // period + 1))[:n]
        else:              # text-like with a few long runs
            raw = bytearray(datagen.text_like(n, 700 + case))
            for _ in range(5):
                p = int(rng.integers(0, max(1, n - 3000)))
                raw[p:p + int(rng.integers(1, 3000))] = bytes([int(rng.integers(0, 256))]) * len(raw[p:p + 1])
            raw = bytes(raw[:n])
        level = int(rng.integers(1, 10))
        check_blocks(native, oracle, dec, datagen.compress(raw, level), raw, check_stages=case % 3 == 0)


def test_maximal_rle_expansion(native, oracle, dec):
    """One value repeated 60 MB: RLE1 packs 255 + 4 bytes into 5, so a 900 kB block decodes to ~46 MB (the case the
    reference's chunk decoder guards with its 64 MiB limit, Bzip2Chunk.hpp:171-184).  Output offsets and the RLE scan
    work far beyond N here; 0xFB as the value also makes every count byte equal to the data byte."""
    for value in (0, 0xFB):
        raw = bytes([value]) * 60_000_000
        enc = datagen.compress(raw, 9)
        results = check_blocks(native, oracle, dec, enc, raw, check_stages=False)
        assert max(r["decoded_size"] for r in results) > 40_000_000


def test_corruption_fuzz_statuses_match_the_oracle(native, oracle, dec):
    """Seeded random damage (bit flips in the header / selector / code-length region and in the data, short bursts,
    truncations) to several kinds of blocks: the GPU must report the same status as the oracle for every block -- the
    reference throw site for a structural error, CRC mismatch otherwise -- and every other field the reference would
    have filled in, and must neither hang nor fault."""
    import numpy as np
    rng = np.random.default_rng(int(os.environ.get("BZ2_FUZZ_SEED", str(0xC0FFEE)), 0))
    sources = [datagen.text_like(150_000, 81), datagen.random_bytes(120_000, 82),
               bytes(rng.integers(0, 7, 200_000, dtype=np.uint8)), b"ab" * 40_000 + datagen.text_like(30_000, 83)]
    for raw in sources:
        for level in (1, 9):
            enc = datagen.compress(raw, level)
            offs = oracle.find_magic(enc)
            for case in range(int(os.environ.get("BZ2_FUZZ_CASES", "14"))):
                bad = bytearray(enc)
                mode = case % 4
                if mode == 0:      # header / tables of a random block
                    o = offs[int(rng.integers(0, len(offs)))] // 8
                    p = min(len(bad) - 1, o + int(rng.integers(6, 400)))
                    bad[p] ^= 1 << int(rng.integers(0, 8))
                elif mode == 1:    # anywhere
                    p = int(rng.integers(4, len(bad)))
                    bad[p] ^= 1 << int(rng.integers(0, 8))
                elif mode == 2:    # burst
                    p = int(rng.integers(4, max(5, len(bad) - 8)))
                    bad[p:p + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
                else:              # truncation
                    bad = bad[:int(rng.integers(5, len(bad)))]
                bad = bytes(bad)
                dec.set_input(bad)
                results, total = dec.decode_batch(offs)
                for o, r in zip(offs, results):
                    od, payload = oracle.decode_block(bad, o)
                    for key in ("status", "encoded_size_bits", "decoded_size", "header_crc", "computed_crc", "is_eos", "is_eof",
                                "orig_ptr", "bwt_length", "n_symbols"):
                        assert r[key] == od[key], (case, mode, o, key, r, od)


def test_damaged_blocks_vs_reference_vectors(native, dec):
    """The GPU against the REAL reference directly (no oracle in between): 60 blocks of 48 damaged files, verdicts and
    calculated CRCs recorded from oracle/_ref/ref_bz2 in tests/golden/reference_vectors.json."""
    from test_oracle import _check_damaged

    def decode_block(data, off):
        dec.set_input(data)
        results, _total = dec.decode_batch([off])
        return results[0]
    _check_damaged(decode_block)


@pytest.mark.parametrize("name", sorted(datagen.exotic_streams()))
def test_exotic_valid_streams(native, oracle, dec, name):
    """20-bit Huffman codes (also on frequent symbols: every window goes through the long-code path), 2..6 tables with
    scrambled selectors, declared-but-unused symbols, surplus selectors -- valid streams libbz2 never writes."""
    raw, enc = datagen.exotic_streams()[name]
    check_blocks(native, oracle, dec, enc, raw)


@pytest.mark.parametrize("name", sorted(datagen.faulty_streams()))
def test_faulty_streams_one_per_throw_site(native, oracle, dec, name):
    enc, status = datagen.faulty_streams()[name]
    results = check_blocks(native, oracle, dec, enc, check_stages=False)
    assert results[0]["status"] == status


def test_batch_of_shuffled_duplicated_and_bogus_offsets(native, oracle, dec):
    """decode_batch sorts blocks into cost groups on several streams internally; results and the ragged output must
    still follow the caller's order -- for shuffled offsets, duplicates, end-of-stream offsets and offsets that are no
    block at all, with enough blocks (>= 64) for the grouping to kick in."""
    import numpy as np
    parts = [datagen.text_like(900_000, 121), datagen.random_bytes(500_000, 122), datagen.runs(700_000, 123),
             bytes(np.random.default_rng(5).integers(0, 4, 600_000, dtype=np.uint8))]
    enc = datagen.multistream(parts, 1)                      # ~30 blocks of 100 kB, four streams
    blocks = oracle.find_magic(enc)
    eos = oracle.find_magic(enc, oracle.MAGIC_EOS)
    rng = np.random.default_rng(77)
    offsets = list(blocks) * 3 + list(eos) + [blocks[2] + 1, blocks[5] + 13, len(enc) * 8 - 3, len(enc) * 8 + 100]
    rng.shuffle(offsets)
    offsets = [int(o) for o in offsets]
    assert len(offsets) >= 64
    dec.set_input(enc)
    results, total = dec.decode_batch(offsets)
    out = dec.copy_output(0, total)
    single = {}
    pos = 0
    for o, r in zip(offsets, results):
        if o not in single:
            single[o] = oracle.decode_block(enc, o)
        d, payload = single[o]
        for key in ("status", "encoded_size_bits", "decoded_size", "header_crc", "computed_crc", "is_eos", "is_eof"):
            assert r[key] == d[key], (o, key, r, d)
        assert r["encoded_offset_bits"] == o
        if d["status"] == 0 and not d["is_eos"]:
            assert r["data_offset"] == pos
            assert out[pos:pos + r["decoded_size"]] == payload, o
            pos += r["decoded_size"]
    assert pos == total


_slice = {}


def bench_slice(native):
    """24 MB of the benchmark corpus as one stitched single-stream level-9 file: (encoded, raw)."""
    if not _slice:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        import silesia_like, bz2build
        data = silesia_like.generate(24_000_000, threads=8)
        enc, nb, offs = bz2build.build(data, 1, piece_size=6_000_000, threads=8, find_magic=native.find_magic)
        assert nb >= 24
        _slice["v"] = (enc, data.tobytes())
    return _slice["v"]


@pytest.mark.parametrize("variant", ["scan-1", "spec-4", "spec-8", "mtf-256", "bwt-1", "bwt-2", "bwt-4"])
def test_huffman_stage_variants(native, oracle, variant, monkeypatch):
    """Every kernel variant that a batch size can select -- k_hscan<1> (one wavefront per block), k_hscan_spec with 4 or 8
    wavefronts per block, the 256-lane k_mtf instances, the table build with 1, 2 or 4 workgroups per block (small batches
    use 8 by themselves) -- against the oracle, whatever the batch size would select by itself: valid data of all kinds,
    streams no libbz2 writes, one invalid stream per reference throw site, and seeded damage (every field of every record)."""
    if variant.startswith("bwt-"):
        # workgroups per block in the table build: 1 = k_bwt_build (what big batches use), 2 / 4 = k_bwt_count + k_bwt_rank with
        # that many slices (small batches use 8 by themselves)
        monkeypatch.setenv("MI355X_BZ2_BWT_SPLIT", variant.split("-")[1])
        monkeypatch.setenv("MI355X_BZ2_SCAN_WAVES", "1")
        monkeypatch.setenv("MI355X_BZ2_MTF_NARROW", "1")
    elif variant == "mtf-256":
        # the 256-lane k_mtf instances, which batches of more than 256 blocks use (small batches take the 512-lane ones)
        monkeypatch.setenv("MI355X_BZ2_MTF_NARROW", "1")
    else:
        # scan-1: k_hscan<1>; spec-N: N waves on N consecutive groups (k_hscan_spec<N>)
        monkeypatch.setenv("MI355X_BZ2_SCAN_WAVES", variant.split("-")[1])
    d = native.Decoder(flags=native.Decoder.KEEP_STAGES)
    try:
        corpus = datagen.corpus_small()
        for name in ("text-1.2M-l9", "rand-300k-l9", "text-400k-l1", "runs-500k-l9", "ab-257", "all-bytes", "rand-1",
                     "zeros-3M-l9", "one-symbol-3"):
            raw, level = corpus[name]
            check_blocks(native, oracle, d, datagen.compress(raw, level), raw, check_stages=name.startswith("text-1.2M"))
        for name, (raw, enc) in datagen.exotic_streams().items():
            check_blocks(native, oracle, d, enc, raw, check_stages=False)
        for name, (enc, status) in datagen.faulty_streams().items():
            results = check_blocks(native, oracle, d, enc, check_stages=False)
            assert results[0]["status"] == status, name
        # full-size blocks of every kind of the benchmark corpus (group lengths from 60 to 400 bits: every span size)
        check_blocks(native, oracle, d, *bench_slice(native), check_stages=False)
        keys = ("encoded_offset_bits", "encoded_size_bits", "decoded_size", "header_crc", "computed_crc", "bwt_length",
                "orig_ptr", "n_symbols", "is_eos", "is_eof", "status")
        for name, (enc, offs) in datagen.damaged_corpus().items():
            if not offs:
                continue
            d.set_input(enc)
            results, _ = d.decode_batch(offs)
            for o, r in zip(offs, results):
                od = oracle.decode_block(enc, o)[0]
                for key in keys:
                    assert r[key] == od[key], (name, o, key, r[key], od[key])
    finally:
        d.close()
