"""Inputs and chunk requests shared by tests/golden/make_golden_chunks.py (which asks the real reference) and
tests/test_gpu_chunk.py (which asks the GPU and compares)."""
import datagen

MAGIC_BLOCK = 0x314159265359
MAGIC_EOS = 0x177245385090
NO_LIMIT = 2**62


def find_bits(data: bytes, magic: int):
    """All bit offsets of a 48-bit pattern (plain Python: the files are small)."""
    out = []
    pattern = magic.to_bytes(6, "big")
    for shift in range(8):
        if shift == 0:
            needle = pattern
            start = 0
            while True:
                i = data.find(needle, start)
                if i < 0:
                    break
                out.append(8 * i)
                start = i + 1
        else:
            # pattern shifted right by `shift` bits spans 7 bytes; the inner 5 bytes are fully determined
            v = magic << (8 - shift)
            inner = ((v >> 8) & ((1 << 40) - 1)).to_bytes(5, "big")
            first = (v >> 48) & 0xFF
            last = v & 0xFF
            fmask = (1 << (8 - shift)) - 1
            lmask = (0xFF << (8 - shift)) & 0xFF
            start = 1
            while True:
                i = data.find(inner, start)
                if i < 0 or i + 5 >= len(data) + 0:
                    break
                if i >= 1 and i + 5 < len(data) and (data[i - 1] & fmask) == first and (data[i + 5] & lmask) == last:
                    out.append(8 * (i - 1) + shift)
                start = i + 1
    return sorted(out)


def inputs():
    """Three streams, 1 + 4 + 2 data blocks (level 1 = 100 kB blocks), and the same file with one damaged block."""
    parts = [datagen.text_like(60_000, 61), datagen.random_text_file(380_000, 62), datagen.random_bytes(150_000, 63)]
    enc = datagen.multistream(parts, 1)
    blocks = find_bits(enc, MAGIC_BLOCK)
    bad = bytearray(enc)
    bad[(blocks[2] >> 3) + 2000] ^= 0x40          # inside the third data block
    return {"multi": enc, "bad": bytes(bad)}


def raw_parts():
    return [datagen.text_like(60_000, 61), datagen.random_text_file(380_000, 62), datagen.random_bytes(150_000, 63)]


def requests(files):
    """(input name, chunk offset, until offset, max decoded bytes)"""
    enc = files["multi"]
    blocks = find_bits(enc, MAGIC_BLOCK)
    eos = find_bits(enc, MAGIC_EOS)
    assert len(blocks) == 7 and len(eos) == 3
    end = len(enc) * 8
    out = [("multi", blocks[0], end + 1000, NO_LIMIT)]
    for i, a in enumerate(blocks):            # every pair of block offsets as [start, until)
        for b in blocks[i:] + [end]:
            out.append(("multi", a, b, NO_LIMIT))
    out.append(("multi", eos[0], blocks[3], NO_LIMIT))            # a chunk may start at an end-of-stream block
    out.append(("multi", eos[1], end, NO_LIMIT))
    out.append(("multi", blocks[1] + 5, blocks[4], NO_LIMIT))     # estimated start, not a magic
    out.append(("multi", blocks[3] - 17, end, NO_LIMIT))
    out.append(("multi", 32, blocks[2], NO_LIMIT))                # the first block of the file
    out.append(("multi", 0, blocks[2], NO_LIMIT))                 # the stream header itself is not a block
    for limit in (150_000, 1, 100_000, 250_000, 0):               # preemptive stop: checked before every block
        out.append(("multi", blocks[1], end, limit))
    out.append(("multi", blocks[-1] + 1, end, NO_LIMIT))          # nothing in range
    out.append(("multi", blocks[2] + 1, blocks[3], NO_LIMIT))
    out.append(("bad", blocks[1], blocks[5], NO_LIMIT))           # a damaged block inside the chain
    out.append(("bad", blocks[2], blocks[5], NO_LIMIT))
    out.append(("bad", blocks[0], end, NO_LIMIT))
    return out
