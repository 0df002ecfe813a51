"""A small bzip2 ENCODER for tests: produces valid single-block streams that libbz2 never would -- Huffman codes up to
20 bits, any number of tables (2..6), arbitrary selector patterns, unused symbols in the map -- so that the decoder
paths for them can be checked against the oracle, CPython's bz2 and the reference ("gaps the build must cover
itself").  Plain Python, small inputs only (the BWT sorts rotations)."""


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, value, n):
        for i in range(n - 1, -1, -1):
            self.bits.append((value >> i) & 1)

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self):
        self.align()
        out = bytearray()
        for i in range(0, len(self.bits), 8):
            b = 0
            for bit in self.bits[i:i + 8]:
                b = (b << 1) | bit
            out.append(b)
        return bytes(out)


def crc32_bzip2(data, crc=0xFFFFFFFF):
    for byte in data:
        crc ^= byte << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    return crc


def rle1(data):
    out = bytearray()
    i = 0
    while i < len(data):
        j = i
        while j < len(data) and data[j] == data[i] and j - i < 255 + 4:
            j += 1
        run = j - i
        if run >= 4:
            out += bytes([data[i]]) * 4 + bytes([run - 4])
        else:
            out += bytes([data[i]]) * run
        i = j
    return bytes(out)


def bwt(s):
    n = len(s)
    doubled = s + s
    order = sorted(range(n), key=lambda i: doubled[i:i + n])
    last = bytes(doubled[i + n - 1] for i in order)
    return last, order.index(0)


def mtf_rle2(last):
    used = sorted(set(last))
    lst = used[:]
    symbols = []
    run = 0

    def flush():
        nonlocal run
        while run > 0:            # bijective base 2: RUNA = 1, RUNB = 2
            if run & 1:
                symbols.append(0)
                run = (run - 1) >> 1
            else:
                symbols.append(1)
                run = (run - 2) >> 1
    for b in last:
        p = lst.index(b)
        if p == 0:
            run += 1
            continue
        flush()
        symbols.append(p + 1)
        del lst[p]
        lst.insert(0, b)
    flush()
    symbols.append(len(used) + 1)   # end of block
    return used, symbols


def canonical_codes(lengths):
    order = sorted(range(len(lengths)), key=lambda s: (lengths[s], s))
    codes = [0] * len(lengths)
    code = 0
    prev = lengths[order[0]]
    for s in order:
        code <<= lengths[s] - prev
        prev = lengths[s]
        codes[s] = code
        code += 1
    return codes


def skewed_lengths(alphabet, ranking, max_len=20):
    """Lengths 1, 2, 3, ... along `ranking` (a permutation of the alphabet), the tail spread over the deepest level so
    that the code is complete.  alphabet <= max_len + 1 gives the fully skewed tree 1..max_len-1, max_len, max_len."""
    lengths = [0] * alphabet
    depth = 1
    remaining = alphabet
    for k, s in enumerate(ranking):
        if remaining > 2 and depth < max_len - 6:
            lengths[s] = depth          # one leaf at this depth, the rest goes deeper
            depth += 1
            remaining -= 1
        else:
            # remaining symbols share the subtree below `depth - 1`: a complete tree over them
            import math
            extra = max(1, math.ceil(math.log2(remaining)))
            base = depth - 1 + extra
            assert base <= max_len, (alphabet, base)
            # leaves: some at base, the others at base - 1 if the tree is not full
            full = 1 << extra
            short = full - remaining            # this many leaves can be one level up
            rest = ranking[k:]
            for i, t in enumerate(rest):
                lengths[t] = base - 1 if i < short else base
            break
    return lengths


def encode_block(data, level=9, n_groups=2, length_fn=None, selector_fn=None, extra_selectors=0, declare_unused=(),
                 faults=None):
    """Single-stream, single-block .bz2 of `data` (non-empty, short).
    length_fn(table_index, alphabet, frequencies) -> code lengths; selector_fn(group_index) -> table index.
    faults: dict of deliberate violations for error-path tests -- randomized (bit), orig_ptr (value written),
    n_groups_field (3-bit field written), n_selectors_field (15-bit field written), drop_selectors (selectors left
    out), symbols (replaces the symbol list; must use the block's alphabet and end with end-of-block)."""
    assert data
    faults = faults or {}
    block_crc = crc32_bzip2(data) ^ 0xFFFFFFFF
    last, orig_ptr = bwt(rle1(data))
    used, symbols = mtf_rle2(last)
    declared = sorted(set(used) | set(declare_unused))
    if declared != used:
        # symbols of the map that never occur: positions shift, re-run MTF over the declared list
        lst = declared[:]
        symbols = []
        run = 0

        def flush():
            nonlocal run
            while run > 0:
                if run & 1:
                    symbols.append(0)
                    run = (run - 1) >> 1
                else:
                    symbols.append(1)
                    run = (run - 2) >> 1
        for b in last:
            p = lst.index(b)
            if p == 0:
                run += 1
                continue
            flush()
            symbols.append(p + 1)
            del lst[p]
            lst.insert(0, b)
        flush()
        symbols.append(len(declared) + 1)
    alphabet = len(declared) + 2
    if "symbols" in faults:
        symbols = list(faults["symbols"])
    freq = [0] * alphabet
    for s in symbols:
        freq[s] += 1
    tables = []
    for t in range(n_groups):
        if length_fn is None:
            ranking = sorted(range(alphabet), key=lambda s: -freq[s])
            lengths = skewed_lengths(alphabet, ranking)
        else:
            lengths = length_fn(t, alphabet, freq)
        assert all(1 <= l <= 20 for l in lengths) and abs(sum(2.0 ** -l for l in lengths) - 1.0) < 1e-12, lengths
        tables.append((lengths, canonical_codes(lengths)))
    n_sel = (len(symbols) + 49) // 50
    selectors = [(selector_fn(g) if selector_fn else g % n_groups) for g in range(n_sel)]
    selectors_written = selectors + [0] * extra_selectors
    if faults.get("drop_selectors"):
        selectors_written = selectors_written[:-faults["drop_selectors"]]

    w = BitWriter()
    w.put(0x425A68, 24)
    w.put(ord("0") + level, 8)
    w.put(0x314159265359, 48)
    w.put(block_crc, 32)
    w.put(faults.get("randomized", 0), 1)
    w.put(faults.get("orig_ptr", orig_ptr), 24)
    groups = [any((16 * g + j) in declared for j in range(16)) for g in range(16)]
    w.put(sum(1 << (15 - g) for g in range(16) if groups[g]), 16)
    for g in range(16):
        if groups[g]:
            w.put(sum(1 << (15 - j) for j in range(16) if (16 * g + j) in declared), 16)
    w.put(faults.get("n_groups_field", n_groups), 3)
    w.put(faults.get("n_selectors_field", len(selectors_written)), 15)
    mtf = list(range(n_groups))
    for s in selectors_written:
        p = mtf.index(s)
        w.put((1 << (p + 1)) - 2, p + 1)      # p ones, then a zero
        del mtf[p]
        mtf.insert(0, s)
    for lengths, _ in tables:
        cur = lengths[0]
        w.put(cur, 5)
        for l in lengths:
            while cur < l:
                w.put(2, 2)
                cur += 1
            while cur > l:
                w.put(3, 2)
                cur -= 1
            w.put(0, 1)
    for i, s in enumerate(symbols):
        lengths, codes = tables[selectors[i // 50]]
        w.put(codes[s], lengths[s])
    w.put(0x177245385090, 48)
    w.put(block_crc, 32)      # one block: stream CRC = rotl(0, 1) ^ blockCRC
    return w.bytes()
