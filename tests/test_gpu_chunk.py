"""mi355x_bz2_decode_chunk against the semantics of rapidgzip's Bzip2Chunk::decodeChunk / decodeUnknownBzip2Chunk
(src/rapidgzip/chunkdecoding/Bzip2Chunk.hpp:34-268), with the oracle's per-block decode as the expected content.
The reference's own chunk decoder needs rapidgzip's ChunkData machinery and is not built here: parity unpinned for the
adapter itself, pinned for every block it returns (same records as decode_batch)."""
import pytest

import datagen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def multi(oracle):
    """Three streams, 1 + 4 + 2 data blocks (level 1 = 100 kB blocks)."""
    parts = [datagen.text_like(60_000, 61), datagen.random_text_file(380_000, 62), datagen.random_bytes(150_000, 63)]
    enc = datagen.multistream(parts, 1)
    blocks = oracle.find_magic(enc, oracle.MAGIC_BLOCK)
    eos = oracle.find_magic(enc, oracle.MAGIC_EOS)
    decoded = {}
    for b in blocks:
        d, payload = oracle.decode_block(enc, b)
        assert d["status"] == 0
        decoded[b] = (d, payload)
    assert len(blocks) == 7 and len(eos) == 3
    return enc, b"".join(parts), blocks, eos, decoded


def expected_chain(enc, oracle, blocks, eos, decoded, start, until, max_decoded=2**63):
    """Blocks of the chain from `start` (a block or EOS offset): data blocks starting before `until`."""
    out, total, footers = [], 0, []
    pos = start
    end = start
    at_stream_end = False
    while True:
        if at_stream_end:
            pos += 32
            at_stream_end = False
        end = pos
        if total >= max_decoded:
            return out, footers, end, True
        if pos >= len(enc) * 8:
            return out, footers, end, False
        is_eos = pos in eos
        if (pos >= until and not is_eos) or pos == until:
            return out, footers, end, False
        if is_eos:
            h = oracle.read_block_header(enc, pos)
            pos += h["encoded_size_bits"]
            footers.append((pos, total))
            at_stream_end = True
            if pos >= len(enc) * 8:
                return out, footers, pos, False
            continue
        d, payload = decoded[pos]
        out.append(pos)
        total += len(payload)
        pos += d["encoded_size_bits"]


def check(native, dec, enc, oracle, blocks, eos, decoded, start, until, max_decoded=2**63, expect_start=None):
    chunk, recs, footers, payload = dec.decode_chunk(enc, start, until, max_decoded)
    first = start if expect_start is None else expect_start
    want_blocks, want_footers, want_end, want_stop = expected_chain(enc, oracle, blocks, eos, decoded, first, until, max_decoded)
    assert chunk["status"] == 0, chunk
    assert chunk["encoded_offset_bits"] == first
    assert [r["encoded_offset_bits"] for r in recs] == want_blocks
    assert payload == b"".join(decoded[b][1] for b in want_blocks)
    assert chunk["decoded_size"] == len(payload)
    assert footers == want_footers
    assert chunk["encoded_end_bits"] == want_end
    assert bool(chunk["stopped_preemptively"]) == want_stop
    # block boundaries: decoded offsets relative to the chunk
    off = 0
    for r in recs:
        assert r["data_offset"] == off and r["status"] == 0
        assert r["computed_crc"] == r["header_crc"] == decoded[r["encoded_offset_bits"]][0]["header_crc"]
        off += r["decoded_size"]


def test_chunks_over_a_multistream_file(native, oracle, multi):
    enc, raw, blocks, eos, decoded = multi
    dec = native.Decoder()
    dec.set_input(enc)
    end = len(enc) * 8
    # whole file from the first block
    check(native, dec, enc, oracle, blocks, eos, decoded, blocks[0], end + 1000)
    chunk, recs, footers, payload = dec.decode_chunk(enc, blocks[0], end + 1000)
    assert payload == raw and len(footers) == 3 and chunk["encoded_end_bits"] == end
    # every pair of block offsets as [start, until)
    for i, a in enumerate(blocks):
        for b in blocks[i:] + [end]:
            check(native, dec, enc, oracle, blocks, eos, decoded, a, b)
    # a chunk may start at an end-of-stream block
    check(native, dec, enc, oracle, blocks, eos, decoded, eos[0], blocks[3])
    # estimated start (not a magic): the next block magic behind it that decodes is taken (Bzip2Chunk.hpp:238-260)
    check(native, dec, enc, oracle, blocks, eos, decoded, blocks[1] + 5, blocks[4], expect_start=blocks[2])
    # preemptive stop: the limit is checked before every block
    check(native, dec, enc, oracle, blocks, eos, decoded, blocks[1], end, max_decoded=150_000)
    check(native, dec, enc, oracle, blocks, eos, decoded, blocks[1], end, max_decoded=1)
    # nothing in range
    chunk, recs, footers, payload = dec.decode_chunk(enc, blocks[-1] + 1, end)
    assert chunk["status"] == 18 and payload == b""
    dec.close()


def test_chunk_with_a_corrupt_block_falls_back_to_the_next_start(native, oracle, multi):
    enc, raw, blocks, eos, decoded = multi
    bad = bytearray(enc)
    bad[(blocks[2] >> 3) + 2000] ^= 0x40          # inside the third data block
    bad = bytes(bad)
    dec = native.Decoder()
    dec.set_input(bad)
    # starting at block 1: block 2 fails -> the attempt fails; block 2 as a start fails; block 3 works
    chunk, recs, footers, payload = dec.decode_chunk(bad, blocks[1], blocks[5])
    assert chunk["status"] == 0 and chunk["encoded_offset_bits"] == blocks[3]
    assert [r["encoded_offset_bits"] for r in recs] == [blocks[3], blocks[4]]
    assert payload == decoded[blocks[3]][1] + decoded[blocks[4]][1]
    dec.close()
