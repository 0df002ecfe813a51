"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libbz2oracle.so (the plain-C restatement, bz2_oracle.c).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package
(indexed_bzip2_amd) never imports this module.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbz2oracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "ref_bz2")

MAGIC_BLOCK = 0x314159265359
MAGIC_EOS = 0x177245385090

STATUS_NAMES = {
    0: "OK", 1: "EOF", 2: "BAD_MAGIC", 3: "RANDOMIZED", 4: "ORIGPTR_RANGE", 5: "GROUP_COUNT", 6: "SELECTOR_COUNT",
    7: "SELECTOR_UNARY", 8: "CODE_LENGTH", 9: "HUFFMAN_LENGTHS", 10: "SELECTOR_OVERRUN", 11: "INVALID_CODE",
    12: "RUN_OVERFLOW", 13: "DATA_OVERFLOW", 14: "ORIGPTR_DATA", 15: "CRC", 16: "STREAM_HEADER", 100: "OUTPUT_CAPACITY",
}


class BlockResult(ctypes.Structure):
    _fields_ = [
        ("encoded_offset_bits", ctypes.c_uint64),
        ("encoded_size_bits", ctypes.c_uint64),
        ("decoded_size", ctypes.c_uint64),
        ("header_crc", ctypes.c_uint32),
        ("computed_crc", ctypes.c_uint32),
        ("bwt_length", ctypes.c_uint32),
        ("orig_ptr", ctypes.c_uint32),
        ("n_symbols", ctypes.c_uint32),
        ("is_eos", ctypes.c_int32),
        ("is_eof", ctypes.c_int32),
        ("status", ctypes.c_int32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


def build():
    """Compile the C restatement (gcc only). Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", _HERE, "libbz2oracle.so"], check=True)
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=False)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.c_char_p
        L.orc_read_stream_header.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint64]
        L.orc_read_stream_header.restype = ctypes.c_int
        L.orc_read_block_header.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(BlockResult)]
        L.orc_read_block_header.restype = ctypes.c_int
        L.orc_decode_block.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.POINTER(BlockResult), ctypes.c_void_p, ctypes.c_void_p]
        L.orc_decode_block.restype = ctypes.c_int
        L.orc_find_magic.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
        L.orc_find_magic.restype = ctypes.c_uint64
        L.orc_decode_file.argtypes = [u8p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                      ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
        L.orc_decode_file.restype = ctypes.c_int
        L.orc_crc32_update.argtypes = [ctypes.c_uint32, u8p, ctypes.c_uint64]
        L.orc_crc32_update.restype = ctypes.c_uint32
        L.orc_run_length.argtypes = [u8p, ctypes.c_uint32]
        L.orc_run_length.restype = ctypes.c_uint32
        L.orc_stream_crc_combine.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        L.orc_stream_crc_combine.restype = ctypes.c_uint32
        _lib = L
    return _lib


def find_magic(data: bytes, magic: int = MAGIC_BLOCK):
    n = lib().orc_find_magic(data, len(data), magic, None, 0)
    arr = (ctypes.c_uint64 * max(1, n))()
    lib().orc_find_magic(data, len(data), magic, arr, n)
    return list(arr[:n])


def read_block_header(data: bytes, bit_offset: int) -> dict:
    res = BlockResult()
    lib().orc_read_block_header(data, len(data), bit_offset, ctypes.byref(res))
    return res.as_dict()


def decode_block(data: bytes, bit_offset: int, want_stages: bool = False, capacity: int = None):
    """Returns (result dict, decoded bytes[, L column, pre-RLE1 bytes])."""
    res = BlockResult()
    # size pass
    lib().orc_decode_block(data, len(data), bit_offset, None, 0, ctypes.byref(res), None, None)
    d = res.as_dict()
    if d["status"] not in (0, 15) or d["is_eos"]:
        return (d, b"", b"", b"") if want_stages else (d, b"")
    cap = d["decoded_size"] if capacity is None else capacity
    out = ctypes.create_string_buffer(max(1, cap))
    lbuf = ctypes.create_string_buffer(max(1, d["bwt_length"])) if want_stages else None
    rbuf = ctypes.create_string_buffer(max(1, d["bwt_length"])) if want_stages else None
    lib().orc_decode_block(data, len(data), bit_offset, out, cap, ctypes.byref(res), lbuf, rbuf)
    d = res.as_dict()
    payload = out.raw[:min(cap, d["decoded_size"])]
    if want_stages:
        return d, payload, lbuf.raw[:d["bwt_length"]], rbuf.raw[:d["bwt_length"]]
    return d, payload


def decode_file(data: bytes, capacity: int = None):
    """Emulates ParallelBZ2Reader: returns (status, decoded bytes, block map dict{bits: bytes}, trailing_garbage)."""
    size = ctypes.c_uint64()
    mlen = ctypes.c_uint64()
    tg = ctypes.c_int()
    mcap = 1 << 16
    bits = (ctypes.c_uint64 * mcap)()
    byts = (ctypes.c_uint64 * mcap)()
    if capacity is None:
        st = lib().orc_decode_file(data, len(data), None, 0, ctypes.byref(size), bits, byts, mcap,
                                   ctypes.byref(mlen), ctypes.byref(tg))
        capacity = size.value
    out = ctypes.create_string_buffer(max(1, capacity))
    st = lib().orc_decode_file(data, len(data), out, capacity, ctypes.byref(size), bits, byts, mcap,
                               ctypes.byref(mlen), ctypes.byref(tg))
    n = min(mlen.value, mcap)
    return st, out.raw[:min(capacity, size.value)], {bits[i]: byts[i] for i in range(n)}, bool(tg.value)


def crc32(data: bytes, crc: int = 0xFFFFFFFF) -> int:
    return lib().orc_crc32_update(crc, data, len(data))


def run_length(digits) -> int:
    b = bytes(digits)
    return lib().orc_run_length(b, len(b))


def ref_available() -> bool:
    return os.path.exists(REF_BIN)


def ref_run(*args) -> str:
    """Run the compiled REAL reference (oracle/_ref/ref_bz2). Test infrastructure only."""
    return subprocess.run([REF_BIN, *map(str, args)], check=False, capture_output=True, text=True).stdout
