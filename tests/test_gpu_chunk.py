"""mi355x_bz2_decode_chunk against the REAL rapidgzip::Bzip2Chunk<ChunkData>::decodeChunk
(src/rapidgzip/chunkdecoding/Bzip2Chunk.hpp:34-268): tests/golden/chunk_vectors.json holds what the reference, compiled
from its own headers (oracle/_ref/ref_chunk, oracle/Makefile), returned for every request of tests/chunk_cases.py --
start, end, decoded size, block boundaries, footers, preemptive stop, NoBlockInRange -- and an FNV-64 of the bytes.
Generator: tests/golden/make_golden_chunks.py.  The inputs are regenerated here and pinned by their sha256."""
import hashlib
import json
import os

import pytest

import chunk_cases

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chunk_vectors.json")


def fnv64(data: bytes) -> str:
    h = 0xcbf29ce484222325
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


@pytest.fixture(scope="module")
def golden():
    vectors = json.load(open(GOLDEN))
    files = chunk_cases.inputs()
    for name, data in files.items():
        assert hashlib.sha256(data).hexdigest() == vectors["inputs"][name], \
            f"input {name} differs from the one the vectors were made for (libbz2 version?): run make_golden_chunks.py"
    return vectors, files


def boundaries_of(recs, footers, eos_offsets):
    """What ChunkData::appendDeflateBlockBoundary receives (Bzip2Chunk.hpp:120-122): the start of every block -- data
    and end-of-stream -- behind the first, with the bytes decoded in front of it, adjacent duplicates dropped."""
    out = []
    for r in recs:
        out.append((r["encoded_offset_bits"], r["data_offset"]))
    for behind, decoded in footers:
        start = max(o for o in eos_offsets if o < behind)
        out.append((start, decoded))
    out = sorted(b for b in out if b[1] > 0)
    return [list(b) for i, b in enumerate(out) if i == 0 or out[i - 1] != b]


def test_chunks_against_reference_vectors(native, golden):
    vectors, files = golden
    eos = {name: chunk_cases.find_bits(data, chunk_cases.MAGIC_EOS) for name, data in files.items()}
    decoders = {}
    for name, data in files.items():
        decoders[name] = native.Decoder()
        decoders[name].set_input(data)
    assert len(vectors["cases"]) >= 50
    for case in vectors["cases"]:
        name, want = case["input"], case["reference"]
        chunk, recs, footers, payload = decoders[name].decode_chunk(files[name], case["start"], case["until"],
                                                                    case["max_decoded"])
        where = (name, case["start"], case["until"], case["max_decoded"])
        if want["status"] == "NoBlockInRange":
            assert chunk["status"] == 18 and payload == b"", where
            continue
        assert want["status"] == "ok" and chunk["status"] == 0, (where, chunk)
        assert chunk["encoded_offset_bits"] == want["encoded_offset_bits"], where
        assert chunk["encoded_end_bits"] == want["encoded_end_bits"], where
        assert chunk["decoded_size"] == want["decoded_size"] == len(payload), where
        assert fnv64(payload) == want["fnv64"], where
        assert bool(chunk["stopped_preemptively"]) == want["stopped_preemptively"], where
        assert [list(f) for f in footers] == want["footers"], where
        assert boundaries_of(recs, footers, eos[name]) == want["boundaries"], where
        # the block records themselves: consistent with the chunk
        off = 0
        for r in recs:
            assert r["data_offset"] == off and r["status"] == 0 and r["computed_crc"] == r["header_crc"], where
            off += r["decoded_size"]
        assert off == chunk["decoded_size"], where
    for d in decoders.values():
        d.close()


def test_chunk_bytes_are_the_file(native, golden):
    vectors, files = golden
    raw = b"".join(chunk_cases.raw_parts())
    dec = native.Decoder()
    dec.set_input(files["multi"])
    chunk, recs, footers, payload = dec.decode_chunk(files["multi"], 32, len(files["multi"]) * 8 + 1000)
    assert payload == raw and len(footers) == 3 and chunk["encoded_end_bits"] == len(files["multi"]) * 8
    dec.close()
