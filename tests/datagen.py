"""Seeded synthetic inputs for the parity tests (encoder = CPython's bz2 module, i.e. libbz2 -- an implementation
independent of both the reference decoder and this repository)."""
import bz2

import numpy as np


def rng(seed):
    return np.random.default_rng(seed)


def random_bytes(n, seed=1):
    return rng(seed).integers(0, 256, n, dtype=np.uint8).tobytes()


def text_like(n, seed=2):
    """Zipf-ish words, newline every ~80 chars: compresses ~3-4x like natural text."""
    r = rng(seed)
    vocab = [bytes(r.integers(97, 123, int(l), dtype=np.uint8)) for l in r.integers(2, 10, 4096)]
    idx = np.minimum(r.zipf(1.3, n // 4 + 16) - 1, len(vocab) - 1)
    out = bytearray()
    col = 0
    for i in idx:
        w = vocab[int(i)]
        out += w
        col += len(w) + 1
        if col > 72:
            out += b"\n"
            col = 0
        else:
            out += b" "
        if len(out) >= n:
            break
    return bytes(out[:n])


def random_text_file(n, seed=3):
    """The reference's createRandomTextFile: 'A' + rand() % 25, newline every 80 (src/core/DataGenerators.hpp:17-26)."""
    r = rng(seed)
    a = (r.integers(0, 25, n, dtype=np.uint8) + 65).astype(np.uint8)
    a[79::80] = 10
    return a.tobytes()


def runs(n, seed=4, maxrun=700):
    """Byte runs of random length (exercises RLE1 counts 0..255 and RUNA/RUNB)."""
    r = rng(seed)
    out = bytearray()
    while len(out) < n:
        out += bytes([int(r.integers(0, 256))]) * int(r.integers(1, maxrun))
    return bytes(out[:n])


def ab_stripes(n, stripe):
    """AB stripes as in src/tests/testPythonWrappers.py:171-245."""
    unit = b"A" * stripe + b"B" * stripe
    return (unit * (n // len(unit) + 1))[:n]


def compress(data, level=9):
    return bz2.compress(data, level)


def enwik_like(n=10_000_000, seed=0xE8E8):
    """BASELINE config 1 / SURVEY 8(d): "enwik-style" text -- XML-ish wiki page markup around Zipf-distributed words
    from the committed list tests/golden/wordlist.txt, seeded xorshift-style generator (numpy PCG64 seeded with 0xE8E8;
    enwik8 itself is not available offline).  At `bzip2 -1` 10 000 000 bytes give ~100 blocks of 100 kB."""
    import os
    r = rng(seed)
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "wordlist.txt"), "rb") as f:
        words = f.read().split()
    out = []
    size = 0
    page = 0
    while size < n + 4096:
        page += 1
        k = int(r.integers(150, 2500))
        idx = ( r.zipf(1.2, k) - 1 ) % len(words)   # the heavy tail is folded over the whole list
        kinds = r.integers(0, 100, k)
        title = b" ".join(words[int(i)].capitalize() for i in r.integers(0, len(words), int(r.integers(1, 4))))
        head = (b"  <page>\n    <title>" + title + b"</title>\n    <id>" + str(page * 7 + 3).encode() + b"</id>\n"
                b"    <revision>\n      <id>" + str(15898000 + page * 11).encode() + b"</id>\n      <timestamp>2006-0"
                + str(1 + page % 9).encode() + b"-1" + str(page % 10).encode() + b"T0" + str(page % 10).encode()
                + b":42:" + str(10 + page % 50).encode() + b"Z</timestamp>\n      <contributor>\n        <username>"
                + words[int(idx[0])].capitalize() + b"</username>\n        <id>" + str(int(kinds[0]) * 131).encode()
                + b"</id>\n      </contributor>\n      <text xml:space=\"preserve\">")
        body = bytearray()
        col = 0
        for w, kind in zip(idx, kinds):
            word = words[int(w)]
            if kind < 4:
                body += b"[[" + word + b"]]"
            elif kind < 6:
                body += b"[[" + word + b"|" + words[int(w) // 2] + b"]]"
            elif kind < 7:
                body += b"''" + word + b"''"
            elif kind < 8:
                body += b"\n\n== " + word.capitalize() + b" ==\n"
                col = 0
                continue
            elif kind < 9:
                body += b"{{" + word + b"|" + str(int(w)).encode() + b"}}"
            elif kind < 10:
                body += b"&quot;" + word + b"&quot;"
            elif kind < 11:
                body += b"\n* " + word
                col = 0
            elif kind < 13:
                body += word + b","
            elif kind < 16:
                body += word.capitalize() if col == 0 else word + b"."
            else:
                body += word
            col += len(word) + 1
            if col > 200 and kind % 5 == 0:
                body += b"\n\n"
                col = 0
            else:
                body += b" "
        tail = b"</text>\n    </revision>\n  </page>\n"
        piece = head + bytes(body) + tail
        out.append(piece)
        size += len(piece)
    data = b"<mediawiki xml:lang=\"en\">\n" + b"".join(out)
    return data[:n]


def corpus_small():
    """name -> (raw, level) : fast cases for both CPU and GPU suites."""
    cases = {
        "rand-1": (random_bytes(1, 11), 9),
        "rand-300k-l9": (random_bytes(300_000, 12), 9),
        "rand-250k-l1": (random_bytes(250_000, 13), 1),      # 3 blocks at level 1
        "text-1.2M-l9": (text_like(1_200_000, 14), 9),        # 2 blocks
        "text-400k-l1": (text_like(400_000, 15), 1),          # several 100k blocks
        "reftext-2MiB-l9": (random_text_file(2 * 1024 * 1024, 16), 9),  # the reference test's 3-block file
        "runs-500k-l9": (runs(500_000, 17), 9),
        "runs-short-l5": (runs(200_000, 18, 9), 5),
        "zeros-3M-l9": (bytes(3_000_000), 9),
        "ff-2M-l9": (b"\xff" * 2_000_000, 9),                 # RLE1 count byte == run byte (0xFF)
        "fb-259k": (b"\xfb" * 259_000, 9),
        "A-512Ki+2-l1": (b"A" * (512 * 1024 + 2), 1),         # testPythonWrappers.py:476-500 regression sizes
        "ab-1": (ab_stripes(100_000, 1), 9),
        "ab-2": (ab_stripes(100_000, 2), 9),
        "ab-8": (ab_stripes(100_000, 8), 9),
        "ab-123": (ab_stripes(100_000, 123), 9),
        "ab-257": (ab_stripes(200_000, 257), 9),
        "ab-2048": (ab_stripes(200_000, 2048), 3),
        "4-4-4-4": (b"\x04" * 8 * 1000, 9),
        "two-symbols": (b"ab" * 5000, 9),
        "one-symbol-3": (b"zzz", 9),
        "all-bytes": (bytes(range(256)) * 40, 9),
    }
    return cases


def multistream(parts, level=9):
    return b"".join(bz2.compress(p, level) for p in parts)


def damaged_corpus(cases=48, seed=0xDA3A6ED):
    """Deterministic damaged inputs: name -> (bytes, [block bit offsets of the UNDAMAGED stream that still lie inside]).
    Bit flips in the header / tables of a block, anywhere, 4-byte bursts, truncations, over four kinds of data and
    levels 1 and 9.  The golden vectors hold what the real reference does with every (file, offset)."""
    import numpy as np
    g = np.random.default_rng(seed)
    sources = [text_like(120_000, 91), random_bytes(90_000, 92), bytes(g.integers(0, 5, 150_000, dtype=np.uint8)),
               b"abc" * 30_000]
    out = {}
    for case in range(cases):
        raw = sources[case % len(sources)]
        enc = bytearray(compress(raw, 1 if case % 3 else 9))
        offs = _block_offsets(bytes(enc))
        mode = case % 4
        if mode == 0:
            o = offs[int(g.integers(0, len(offs)))] // 8
            p = min(len(enc) - 1, o + int(g.integers(6, 400)))
            enc[p] ^= 1 << int(g.integers(0, 8))
        elif mode == 1:
            enc[int(g.integers(4, len(enc)))] ^= 1 << int(g.integers(0, 8))
        elif mode == 2:
            p = int(g.integers(4, max(5, len(enc) - 8)))
            enc[p:p + 4] = g.integers(0, 256, 4, dtype=np.uint8).tobytes()
        else:
            enc = enc[:int(g.integers(5, len(enc)))]
        out[f"damaged-{case:02d}"] = (bytes(enc), [o for o in offs if o + 48 <= len(enc) * 8])
    return out


def _block_offsets(enc):
    """Bit offsets of the block magic 0x314159265359 (plain Python/numpy, independent of oracle and product)."""
    import numpy as np
    a = np.frombuffer(enc, dtype=np.uint8)
    bits = np.unpackbits(a)
    magic = np.unpackbits(np.frombuffer(bytes.fromhex("314159265359"), dtype=np.uint8))
    # candidates: positions where the first 16 magic bits match, then verify
    n = len(bits) - 48 + 1
    if n <= 0:
        return []
    cand = np.flatnonzero(np.lib.stride_tricks.sliding_window_view(bits, 8)[:n, :].dot(1 << np.arange(7, -1, -1)) == 0x31)
    return [int(c) for c in cand if c + 48 <= len(bits) and np.array_equal(bits[c:c + 48], magic)]


def exotic_streams():
    """Valid streams no libbz2 writes (tests/bz2enc.py): name -> (raw, encoded).  Codes of up to 20 bits on rare and on
    FREQUENT symbols, 2..6 tables with scrambled selectors, symbols that are declared but never used, surplus
    selectors.  (libbz2 limits code lengths to 17 and never declares unused symbols.)"""
    import random
    import bz2enc
    r = random.Random(0xE807)
    values = bytes(r.sample(range(256), 19))                       # alphabet 19 + 2 = 21: lengths 1..19, 20, 20
    skewed = bytes(r.choices(values, weights=[2.0 ** -i for i in range(19)], k=3000))
    flat = bytes(r.choices(values, k=2500))

    def rarest_first(t, alphabet, freq):
        ranking = sorted(range(alphabet), key=lambda s: (freq[s], s))
        return bz2enc.skewed_lengths(alphabet, ranking)

    def rotating(t, alphabet, freq):
        ranking = sorted(range(alphabet), key=lambda s: (-freq[s], s))
        ranking = ranking[t:] + ranking[:t]                        # a different code per table
        return bz2enc.skewed_lengths(alphabet, ranking)
    wide = bytes(r.choices(range(60), k=5000))
    out = {
        "skew20-2tables": (skewed, bz2enc.encode_block(skewed, n_groups=2)),
        "skew20-frequent-long-6tables": (flat, bz2enc.encode_block(flat, n_groups=6, length_fn=rarest_first,
                                                                  selector_fn=lambda g: (g * 5 + 1) % 6)),
        "rotating-codes-5tables": (skewed, bz2enc.encode_block(skewed, level=1, n_groups=5, length_fn=rotating,
                                                                selector_fn=lambda g: (g * 3) % 5)),
        "alphabet62-deep-tail": (wide, bz2enc.encode_block(wide, n_groups=3)),
        "declared-unused-extra-selectors": (skewed, bz2enc.encode_block(skewed, n_groups=4, declare_unused=(1, 200, 255),
                                                                         extra_selectors=7)),
        "tiny": (b"\x00", bz2enc.encode_block(b"\x00", n_groups=2)),
    }
    return out


def faulty_streams():
    """Structurally invalid blocks, one per reference throw site that random damage rarely reaches (tests/bz2enc.py with
    deliberate violations): name -> (encoded, status).  The statuses are those of the REAL reference for exactly these
    streams (oracle/_ref/ref_bz2 probe, exception texts in the comments)."""
    import bz2enc
    data = bytes(range(30)) * 40
    alphabet = 32                                   # 30 byte values + RUNA/RUNB ... end-of-block = symbol 31

    def enc(**faults):
        return bz2enc.encode_block(data, n_groups=3, faults=faults)
    return {
        "randomized-bit": (enc(randomized=1), 3),                  # "deprecated isRandomized bit is not supported"
        "origptr-900001": (enc(orig_ptr=900001), 4),               # "origPtr 900001 is larger than buffer size: 900000"
        "origptr-equals-n": (enc(orig_ptr=len(bz2enc.rle1(data))), 14),   # "[BZip2 block data] origPtr error 1200"
        "group-count-1": (enc(n_groups_field=1), 5),               # "Invalid Huffman coding group count 1"
        "group-count-7": (enc(n_groups_field=7), 5),
        "selector-count-0": (enc(n_selectors_field=0), 6),         # "The number of selectors 0 is invalid"
        "selectors-run-out": (enc(drop_selectors=2), 10),          # "selector 2 out of maximum range 2"
        "run-overflow": (enc(symbols=[1] * 22 + [5, alphabet - 1]), 12),        # "dbufCount + hh 8388606 > 900000"
        "data-overflow": (enc(symbols=[2, 2] * 450_001 + [alphabet - 1]), 13),  # "dbufCount 900000 > 900000 dbufSize"
    }
