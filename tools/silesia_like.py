"""Seeded "Silesia-style" synthetic corpus (no network: the real Silesia corpus is not available).

Mix modelled on the Silesia corpus' composition (text, XML, source code, database tables, executables, medical images /
PCM-like 16-bit samples, DNA-like text, a little incompressible data), tuned so that `bzip2 -9` lands at a ratio of
about 3.9 (Silesia's bzip2 ratio is 3.88: results/paper/sections/6 - Evaluation.tex:329 of the reference).
Everything is numpy-vectorised so that ~200 MB generate in seconds.
"""
import functools

import numpy as np


def _words_to_bytes(vocab, idx, sep_choices, rng):
    """Concatenate vocab[idx[i]] + separator, fully vectorised."""
    lens, table = _table(vocab)
    wl = lens[idx] + 1
    starts = np.concatenate(([0], np.cumsum(wl)[:-1]))
    total = int(wl.sum())
    word_of_char = np.repeat(np.arange(len(idx)), wl)
    pos = np.arange(total) - starts[word_of_char]
    out = table[idx[word_of_char], pos]
    sep = np.asarray(sep_choices, dtype=np.uint8)[rng.integers(0, len(sep_choices), len(idx))]
    out[starts + wl - 1] = sep
    return out


_VOCAB_CACHE = {}


def _vocab(rng, n, lo, hi, alphabet):
    """Deterministic vocabulary (own seed, cached): does not consume the caller's rng."""
    key = (n, lo, hi, alphabet)
    if key not in _VOCAB_CACHE:
        r = np.random.default_rng(hash(key) & 0xFFFFFFFF if False else (n * 1000003 + lo * 101 + hi * 7 + len(alphabet)))
        a = np.frombuffer(alphabet, dtype=np.uint8)
        lens = r.integers(lo, hi, n)
        flat = a[r.integers(0, len(a), int(lens.sum()))].tobytes()
        out, p = [], 0
        for l in lens:
            out.append(flat[p:p + int(l)])
            p += int(l)
        _VOCAB_CACHE[key] = out
    return _VOCAB_CACHE[key]


_TABLE_CACHE = {}


def _table(vocab):
    key = id(vocab)
    if key not in _TABLE_CACHE:
        lens = np.array([len(w) for w in vocab], dtype=np.int64)
        width = int(lens.max()) + 1
        table = np.zeros((len(vocab), width), dtype=np.uint8)
        for i, w in enumerate(vocab):
            table[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
        _TABLE_CACHE[key] = (vocab, lens, table)
    return _TABLE_CACHE[key][1], _TABLE_CACHE[key][2]


def prose(n, rng):
    vocab = _vocab(rng, 30000, 2, 11, b"etaoinshrdlucmfwypvbgkqjxzeeeaaaooiitnn")
    nwords = n // 5 + 64
    idx = np.minimum(rng.zipf(1.08, nwords) - 1, len(vocab) - 1)
    seps = [32] * 14 + [10, 44, 46]
    return _words_to_bytes(vocab, idx, seps, rng)[:n]


@functools.lru_cache(maxsize=None)
def _xml_vocab():
    tags = [b"<" + t + b">" for t in _vocab(None, 300, 3, 12, b"abcdefghijklmnopqrstuvwxyz")]
    tags += [b"</" + t[1:] for t in tags]
    words = _vocab(None, 20000, 1, 9, b"etaoinshrdlucmfwypvbgk0123456789")
    return tags, words, tags + words


def xml(n, rng):
    tags, words, vocab = _xml_vocab()
    nwords = n // 5 + 64
    kind = rng.random(nwords) < 0.45
    idx = np.where(kind, rng.integers(0, len(tags), nwords),
                   len(tags) + np.minimum(rng.zipf(1.35, nwords) - 1, len(words) - 1))
    return _words_to_bytes(vocab, idx, [32, 32, 10, 32, 61, 34], rng)[:n]


@functools.lru_cache(maxsize=None)
def _source_vocab():
    kw = [b"if", b"else", b"for", b"while", b"return", b"int", b"char", b"void", b"struct", b"static", b"const",
          b"#include", b"unsigned", b"sizeof", b"break", b"case", b"switch", b"NULL", b"size_t", b"uint32_t"]
    ident = _vocab(None, 12000, 3, 14, b"abcdefghijklmnopqrstuvwxyz_ABCDEFXYZ0123")
    punct = [b"(", b")", b"{", b"}", b";", b"=", b"==", b"->", b"*", b"+", b",", b"[", b"]", b"0", b"1", b"    ",
             b"        ", b"//", b"/*", b"*/", b"&&", b"<", b">"]
    return kw, punct, ident, kw + punct + ident


def source(n, rng):
    kw, punct, ident, vocab = _source_vocab()
    nwords = n // 4 + 64
    r = rng.random(nwords)
    idx = np.where(r < 0.25, rng.integers(0, len(kw), nwords),
                   np.where(r < 0.6, len(kw) + rng.integers(0, len(punct), nwords),
                            len(kw) + len(punct) + np.minimum(rng.zipf(1.15, nwords) - 1, len(ident) - 1)))
    return _words_to_bytes(vocab, idx, [32, 32, 32, 10], rng)[:n]


def database(n, rng):
    """Fixed-width records with sorted keys, like a table dump."""
    rec = 64
    k = n // rec + 1
    keys = np.cumsum(rng.integers(1, 40, k)).astype(np.int64)
    out = np.full((k, rec), 32, dtype=np.uint8)
    for d in range(10):
        out[:, 9 - d] = 48 + (keys // 10**d) % 10
    names = np.frombuffer(b"".join(_vocab(rng, 4096, 12, 13, b"ABCDEFGHIJKLMNOPRSTUVW")), dtype=np.uint8).reshape(4096, 12)
    out[:, 11:23] = names[np.minimum(rng.zipf(1.3, k) - 1, 4095)]
    vals = rng.integers(0, 100000, k)
    for d in range(6):
        out[:, 30 - d] = 48 + (vals // 10**d) % 10
    codes = np.frombuffer(b"".join(_vocab(None, 512, 28, 29, b"0123456789ABCDEF  ")), dtype=np.uint8).reshape(512, 28)
    out[:, 32:60] = codes[np.minimum(rng.zipf(1.2, k) - 1, 511)]
    flip = rng.random(k) < 0.3
    out[flip, 40:46] = rng.choice(np.frombuffer(b"0123456789", dtype=np.uint8), (int(flip.sum()), 6))
    out[:, 63] = 10
    return out.reshape(-1)[:n]


def binary(n, rng):
    """Executable-like: opcode-ish bytes from a skewed table mixed with small little-endian immediates."""
    table = rng.permutation(256).astype(np.uint8)
    ops = table[np.minimum(rng.zipf(1.9, n) - 1, 255)]
    imm = rng.random(n) < 0.30
    ops[imm] = np.where(rng.random(int(imm.sum())) < 0.6, 0, 255).astype(np.uint8)
    # repeated code fragments
    frag = 4096
    for start in rng.integers(0, max(1, n - 2 * frag), n // (frag * 2)):
        src = int(rng.integers(0, max(1, n - frag)))
        ops[start:start + frag] = ops[src:src + frag]
    return ops


def pcm16(n, rng):
    """16-bit little-endian random walk (medical image rows / audio)."""
    k = n // 2 + 1
    walk = np.cumsum(rng.integers(-6, 7, k)).astype(np.int64)
    s = (walk + (rng.integers(-1, 2, k))).astype(np.int16)
    return s.view(np.uint8)[:n]


def dna(n, rng):
    motifs = _vocab(rng, 2000, 8, 40, b"ACGT")
    idx = rng.integers(0, len(motifs), n // 20 + 8)
    body = np.frombuffer(b"".join(motifs[i] for i in idx[:4000]), dtype=np.uint8)
    base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    reps = n // max(1, len(body)) // 3
    for _ in range(reps):
        p = int(rng.integers(0, n - len(body)))
        base[p:p + len(body)] = body
    return base


def noise(n, rng):
    return rng.integers(0, 256, n, dtype=np.uint8)


# (generator, share of the corpus) -- shares roughly follow Silesia's file sizes by type
MIX = [(prose, 0.28), (xml, 0.12), (source, 0.14), (database, 0.12), (binary, 0.16), (pcm16, 0.11), (dna, 0.05),
       (noise, 0.02)]   # bzip2 -9 ratio 3.76 on 106 MB (target 3.9 +- 0.3)


def generate(n_bytes, seed=0x51E51A, segment=6_000_000, threads=8):
    """Interleave ~6 MB segments of each type (like the files of a tar) up to n_bytes.  Segments are planned with one
    rng and generated independently (own seeded rng each) on a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    plan_rng = np.random.default_rng(seed)
    gens = [g for g, _ in MIX]
    shares = np.array([s for _, s in MIX])
    shares = shares / shares.sum()
    produced = np.zeros(len(gens))
    plan = []
    total = 0
    while total < n_bytes:
        deficit = shares * (total + segment) - produced   # the type furthest below its share goes next
        k = int(np.argmax(deficit))
        seglen = max(1, int(min(segment * (0.5 + plan_rng.random()), n_bytes - total)))
        plan.append((k, seglen, int(plan_rng.integers(0, 2**31))))
        produced[k] += seglen
        total += seglen
    # build the shared vocabularies before going parallel
    for g in gens:
        g(1000, np.random.default_rng(1))

    def make(item):
        k, seglen, sd = item
        return np.ascontiguousarray(gens[k](seglen, np.random.default_rng(sd))[:seglen])

    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(make, plan))
    return np.concatenate(parts)[:n_bytes]


if __name__ == "__main__":
    import bz2
    import sys
    import time
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
    rng = np.random.default_rng(1)
    for g, share in MIX:
        t0 = time.time()
        d = g(4_000_000, rng).tobytes()
        t1 = time.time()
        c = len(bz2.compress(d, 9))
        print(f"{g.__name__:10s} share {share:.2f} gen {t1 - t0:.2f}s ratio {len(d) / c:.2f}")
    t0 = time.time()
    d = generate(n).tobytes()
    t1 = time.time()
    c = len(bz2.compress(d, 9))
    print(f"mix: {n / 1e6:.0f} MB gen {t1 - t0:.2f}s ratio {len(d) / c:.3f}")
