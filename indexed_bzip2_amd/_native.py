"""ctypes binding of the C ABI declared in include/mi355x_bz2.h (libmi355x_bz2.so, built by build.py).

The product path has no CPU fallback: if the HIP library is missing or no MI355X is present, the functions here
raise -- nothing in this package imports or calls oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MI355X_BZ2_LIBRARY: development only, an A/B build of the same ABI)
LIB_PATH = os.environ.get("MI355X_BZ2_LIBRARY") or os.path.join(_HERE, "libmi355x_bz2.so")

MAGIC_BLOCK = 0x314159265359
MAGIC_EOS = 0x177245385090

OK = 0
ERR_CRC = 15
ERR_STREAM_CRC = 17
ERR_NO_DEVICE = 102


class Config(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("max_batch_blocks", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class BlockResult(ctypes.Structure):
    _fields_ = [
        ("encoded_offset_bits", ctypes.c_uint64),
        ("encoded_size_bits", ctypes.c_uint64),
        ("decoded_size", ctypes.c_uint64),
        ("data_offset", ctypes.c_uint64),
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


MAX_KERNELS = 16


class Timings(ctypes.Structure):
    _fields_ = [("ms_total", ctypes.c_float), ("ms_kernel_sum", ctypes.c_float), ("n_kernels", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("ms_kernel", ctypes.c_float * MAX_KERNELS)]

    def as_dict(self):
        names = [lib().mi355x_bz2_kernel_name(i).decode() for i in range(self.n_kernels)]
        return {"ms_total": self.ms_total, "ms_kernel_sum": self.ms_kernel_sum,
                "kernels": {names[i]: self.ms_kernel[i] for i in range(self.n_kernels)}}


class ChunkBoundary(ctypes.Structure):
    _fields_ = [("encoded_offset_bits", ctypes.c_uint64), ("decoded_offset", ctypes.c_uint64)]


class ChunkResult(ctypes.Structure):
    _fields_ = [("encoded_offset_bits", ctypes.c_uint64), ("encoded_end_bits", ctypes.c_uint64),
                ("decoded_size", ctypes.c_uint64), ("data_offset", ctypes.c_uint64),
                ("n_blocks", ctypes.c_uint32), ("n_footers", ctypes.c_uint32),
                ("stopped_preemptively", ctypes.c_int32), ("status", ctypes.c_int32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class ReaderStats(ctypes.Structure):
    _fields_ = [("gets", ctypes.c_uint64), ("cache_hits", ctypes.c_uint64), ("prefetch_hits", ctypes.c_uint64),
                ("on_demand_fetches", ctypes.c_uint64), ("prefetches_submitted", ctypes.c_uint64),
                ("batches", ctypes.c_uint64), ("blocks_decoded", ctypes.c_uint64),
                ("failed_prefetches", ctypes.c_uint64),
                ("decode_seconds", ctypes.c_double), ("wait_seconds", ctypes.c_double),
                ("input_resident", ctypes.c_uint64), ("input_bytes_uploaded", ctypes.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# every symbol include/mi355x_bz2.h declares: (name, restype, argtypes)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
SYMBOLS = [
    ("mi355x_bz2_status_string", ctypes.c_char_p, [ctypes.c_int]),
    ("mi355x_bz2_abi_version", ctypes.c_int, []),
    ("mi355x_bz2_create", ctypes.c_int, [ctypes.POINTER(Config), ctypes.POINTER(_vp)]),
    ("mi355x_bz2_warmup", ctypes.c_int, [ctypes.c_int32]),
    ("mi355x_bz2_destroy", None, [_vp]),
    ("mi355x_bz2_last_error", ctypes.c_char_p, [_vp]),
    ("mi355x_bz2_set_input_host", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_uint64]),
    ("mi355x_bz2_set_input_host_async", ctypes.c_int, [_vp, ctypes.c_void_p, ctypes.c_uint64]),
    ("mi355x_bz2_set_input_host_streamed", ctypes.c_int, [_vp, ctypes.c_void_p, ctypes.c_uint64]),
    ("mi355x_bz2_input_resident", ctypes.c_int, [_vp]),
    ("mi355x_bz2_set_input_device", ctypes.c_int, [_vp, _vp, ctypes.c_uint64]),
    ("mi355x_bz2_decode_batch", ctypes.c_int, [_vp, _u64p, ctypes.c_uint32, ctypes.POINTER(BlockResult), _u64p]),
    ("mi355x_bz2_decode_batch_begin", ctypes.c_int, [_vp, _u64p, ctypes.c_uint32]),
    ("mi355x_bz2_decode_batch_end", ctypes.c_int, [_vp, ctypes.POINTER(BlockResult), _u64p]),
    ("mi355x_bz2_output_device", _vp, [_vp]),
    ("mi355x_bz2_copy_output", ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    ("mi355x_bz2_copy_output_begin", ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    ("mi355x_bz2_copy_output_end", ctypes.c_int, [_vp]),
    ("mi355x_bz2_hold_output_until", ctypes.c_int, [_vp, _vp]),
    ("mi355x_bz2_last_timings", ctypes.c_int, [_vp, ctypes.POINTER(Timings)]),
    ("mi355x_bz2_last_pipeline_ms", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    ("mi355x_bz2_kernel_name", ctypes.c_char_p, [ctypes.c_uint32]),
    ("mi355x_bz2_stream", _vp, [_vp]),
    ("mi355x_bz2_device_memory", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("mi355x_bz2_debug_copy_stage", ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_int, _vp, ctypes.c_uint64]),
    ("mi355x_bz2_find_magic", ctypes.c_uint64, [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, _u64p,
                                                 ctypes.c_uint64, ctypes.c_uint32]),
    ("mi355x_bz2_share_input", ctypes.c_int, [_vp, _vp]),
    ("mi355x_bz2_find_magic_device", ctypes.c_int, [_vp, ctypes.c_uint64, _u64p, ctypes.c_uint64, _u64p]),
    ("mi355x_bz2_crc32_device", ctypes.c_int, [_vp, _vp, _u64p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]),
    ("mi355x_bz2_read_stream_header", ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64]),
    ("mi355x_bz2_reader_open_path", ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int32, ctypes.POINTER(_vp)]),
    ("mi355x_bz2_reader_open_fd", ctypes.c_int, [ctypes.c_int, ctypes.c_uint32, ctypes.c_int32, ctypes.POINTER(_vp)]),
    ("mi355x_bz2_reader_open_memory", ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32,
                                                      ctypes.POINTER(_vp)]),
    ("mi355x_bz2_reader_close", None, [_vp]),
    ("mi355x_bz2_reader_last_error", ctypes.c_char_p, [_vp]),
    ("mi355x_bz2_reader_read", ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_uint64, _u64p]),
    ("mi355x_bz2_reader_seek", ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _u64p]),
    ("mi355x_bz2_reader_tell", ctypes.c_uint64, [_vp]),
    ("mi355x_bz2_reader_eof", ctypes.c_int, [_vp]),
    ("mi355x_bz2_reader_closed", ctypes.c_int, [_vp]),
    ("mi355x_bz2_reader_size", ctypes.c_int, [_vp, _u64p]),
    ("mi355x_bz2_reader_tell_compressed", ctypes.c_uint64, [_vp]),
    ("mi355x_bz2_reader_block_offsets_complete", ctypes.c_int, [_vp]),
    ("mi355x_bz2_reader_block_offsets", ctypes.c_int, [_vp, _u64p, _u64p, ctypes.c_uint64, _u64p]),
    ("mi355x_bz2_reader_available_block_offsets", ctypes.c_int, [_vp, _u64p, _u64p, ctypes.c_uint64, _u64p]),
    ("mi355x_bz2_reader_set_block_offsets", ctypes.c_int, [_vp, _u64p, _u64p, ctypes.c_uint64]),
    ("mi355x_bz2_reader_join_threads", ctypes.c_int, [_vp]),
    ("mi355x_bz2_reader_set_verify_stream_crc", ctypes.c_int, [_vp, ctypes.c_int]),
    ("mi355x_bz2_reader_streams_verified", ctypes.c_uint64, [_vp]),
    ("mi355x_bz2_reader_statistics", ctypes.c_int, [_vp, ctypes.POINTER(ReaderStats)]),
    ("mi355x_bz2_decode_chunk", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                               ctypes.c_uint64, ctypes.POINTER(ChunkResult), ctypes.POINTER(BlockResult),
                                               ctypes.c_uint32, ctypes.POINTER(ChunkBoundary), ctypes.c_uint32]),
]

_lib = None


def lib():
    """Load libmi355x_bz2.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -m indexed_bzip2_amd.build). "
                "indexed_bzip2_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def warmup(device: int = 0, background: bool = True):
    """Start the HIP runtime and load the kernels now (0.2 s once per process) instead of inside the first open():
    on a daemon thread by default, so the caller carries on.  Optional; not for processes that fork workers later."""
    L = lib()
    if not background:
        rc = L.mi355x_bz2_warmup(device)
        if rc != 0:
            raise Bz2Error(rc)
        return None
    import threading
    thread = threading.Thread(target=L.mi355x_bz2_warmup, args=(device,), daemon=True)
    thread.start()
    return thread


def status_string(status: int) -> str:
    return lib().mi355x_bz2_status_string(status).decode()


class Bz2Error(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = status_string(status)
        if detail:
            msg += f" ({detail})"
        super().__init__(msg)


def find_magic(data: bytes, magic: int = MAGIC_BLOCK, threads: int = 0):
    L = lib()
    n = L.mi355x_bz2_find_magic(data, len(data), magic, None, 0, threads)
    arr = (ctypes.c_uint64 * max(1, n))()
    L.mi355x_bz2_find_magic(data, len(data), magic, arr, n, threads)
    return list(arr[:n])


class _KeptInputs(tuple):
    """(previous, current) host buffers of set_input_host_async."""


class Decoder:
    """One decoder context = one GPU + one HIP stream (mi355x_bz2_ctx)."""

    KEEP_STAGES = 1

    def __init__(self, device: int = -1, max_batch_blocks: int = 0, flags: int = 0):
        self._h = _vp()
        cfg = Config(device, max_batch_blocks, flags, 0)
        rc = lib().mi355x_bz2_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != OK:
            raise Bz2Error(rc)
        self._input_ref = None
        self.last_results = []

    def close(self):
        if self._h:
            lib().mi355x_bz2_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise Bz2Error(rc, lib().mi355x_bz2_last_error(self._h).decode())

    def set_input(self, data: bytes):
        self._check(lib().mi355x_bz2_set_input_host(self._h, data, len(data)))

    def set_input_host_async(self, ptr: int, size: int, keepalive=None):
        """Queue the H2D copy of `size` bytes at host address `ptr` (page-locked memory) on the decoder's input stream
        and return: the next begin_batch / decode_batch is ordered behind it.  May be called while a batch is in flight
        (the bytes of the next one).  The memory must stay valid until the batch that uses it has ended."""
        # the previous input may still be on its way to the GPU or in use by the batch in flight: keep it too, and no more
        # (the context has two input buffers: the batch in flight's and the next one's)
        previous = self._input_ref
        self._input_ref = (previous[1] if isinstance(previous, _KeptInputs) else previous, keepalive)
        self._input_ref = _KeptInputs(self._input_ref)
        self._check(lib().mi355x_bz2_set_input_host_async(self._h, ptr, size))

    def set_input_device(self, ptr: int, size: int, keepalive=None):
        self._input_ref = keepalive
        self._check(lib().mi355x_bz2_set_input_device(self._h, ptr, size))

    def share_input(self, other: "Decoder"):
        """Decode from the bytes `other` made resident (no copy); `other` is kept alive."""
        self._input_ref = other
        self._check(lib().mi355x_bz2_share_input(self._h, other._h))

    def decode_batch(self, offsets):
        n = len(offsets)
        offs = (ctypes.c_uint64 * max(1, n))(*offsets)
        res = (BlockResult * max(1, n))()
        total = ctypes.c_uint64()
        self._check(lib().mi355x_bz2_decode_batch(self._h, offs, n, res, ctypes.byref(total)))
        self.last_results = [res[i].as_dict() for i in range(n)]
        return self.last_results, total.value

    @staticmethod
    def make_arrays(offsets):
        """ctypes arrays for decode_batch_into: (offsets, results), reusable across calls."""
        n = len(offsets)
        return (ctypes.c_uint64 * max(1, n))(*offsets), (BlockResult * max(1, n))()

    def decode_batch_into(self, offsets_array, n: int, results_array) -> int:
        """decode_batch without building Python dicts (2 560 blocks x 12 fields cost milliseconds): the results stay in
        `results_array` (ctypes BlockResult[n], e.g. viewed through numpy.frombuffer).  Returns the decoded size."""
        total = ctypes.c_uint64()
        self._check(lib().mi355x_bz2_decode_batch(self._h, offsets_array, n, results_array, ctypes.byref(total)))
        self.last_results = None
        return total.value

    def begin_batch(self, offsets_array, n: int):
        """First half of decode_batch_into: queues the batch up to its decoded sizes and returns at once."""
        self._check(lib().mi355x_bz2_decode_batch_begin(self._h, offsets_array, n))

    def end_batch(self, results_array) -> int:
        """Second half: output offsets, RLE expansion, CRC; returns the decoded size."""
        total = ctypes.c_uint64()
        self._check(lib().mi355x_bz2_decode_batch_end(self._h, results_array, ctypes.byref(total)))
        self.last_results = None
        return total.value

    def decode_chunk(self, data: bytes, chunk_offset: int, until_offset: int, max_decoded: int = 2**63):
        """rapidgzip Bzip2Chunk::decodeChunk counterpart (mi355x_bz2_decode_chunk): `data` = the bytes given to
        set_input().  Returns (chunk dict, block dicts, footers [(encoded bits, decoded offset)], chunk bytes)."""
        cap = 4096
        res = ChunkResult()
        blocks = (BlockResult * cap)()
        footers = (ChunkBoundary * cap)()
        self._check(lib().mi355x_bz2_decode_chunk(self._h, data, len(data), chunk_offset, until_offset, max_decoded,
                                                  ctypes.byref(res), blocks, cap, footers, cap))
        d = res.as_dict()
        payload = self.copy_output(d["data_offset"], d["decoded_size"]) if d["status"] == OK and d["decoded_size"] else b""
        return (d, [blocks[i].as_dict() for i in range(min(cap, d["n_blocks"]))],
                [(footers[i].encoded_offset_bits, footers[i].decoded_offset) for i in range(min(cap, d["n_footers"]))],
                payload)

    def find_magic(self, magic: int = MAGIC_BLOCK):
        """Magic-bit scan of the resident input on the GPU (k_find_magic)."""
        n = ctypes.c_uint64()
        self._check(lib().mi355x_bz2_find_magic_device(self._h, magic, None, 0, ctypes.byref(n)))
        arr = (ctypes.c_uint64 * max(1, n.value))()
        self._check(lib().mi355x_bz2_find_magic_device(self._h, magic, arr, n.value, ctypes.byref(n)))
        return list(arr[:n.value])

    def crc32_device(self, device_ptr: int, sizes):
        """bzip2 CRC-32 of consecutive pieces (`sizes` bytes each) of a 16-byte aligned device buffer."""
        n = len(sizes)
        arr = (ctypes.c_uint64 * max(1, n))(*sizes)
        out = (ctypes.c_uint32 * max(1, n))()
        self._check(lib().mi355x_bz2_crc32_device(self._h, ctypes.c_void_p(device_ptr), arr, n, out))
        return list(out[:n])

    def output_device_ptr(self) -> int:
        return lib().mi355x_bz2_output_device(self._h) or 0

    def stream_ptr(self) -> int:
        return lib().mi355x_bz2_stream(self._h) or 0

    def device_memory(self) -> dict:
        """bytes of HBM this context holds: per-block scratch and output buffers"""
        scratch, output = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(lib().mi355x_bz2_device_memory(self._h, ctypes.byref(scratch), ctypes.byref(output)))
        return {"scratch_bytes": scratch.value, "output_bytes": output.value}

    def hold_output_until(self, hip_event: int, keepalive=None):
        """The next batch's output kernels wait for `hip_event` (e.g. torch.cuda.Event.cuda_event recorded behind a
        collective that reads output_device_ptr()); the event object must stay alive: pass it as `keepalive`."""
        self._output_hold = keepalive
        self._check(lib().mi355x_bz2_hold_output_until(self._h, hip_event))

    def copy_output_begin(self, offset: int, size: int):
        """Background D2H of the last batch's bytes; returns the ctypes buffer, valid after copy_output_end()."""
        buf = (ctypes.c_ubyte * max(1, size))()
        self._check(lib().mi355x_bz2_copy_output_begin(self._h, offset, size, buf))
        return buf

    def copy_output_begin_to(self, offset: int, size: int, host_ptr: int, keepalive=None):
        """The same into caller-owned (page-locked) host memory at address `host_ptr`; valid after copy_output_end()."""
        self._copy_ref = keepalive
        self._check(lib().mi355x_bz2_copy_output_begin(self._h, offset, size, ctypes.c_void_p(host_ptr)))

    def copy_output_end(self):
        self._check(lib().mi355x_bz2_copy_output_end(self._h))

    def copy_output(self, offset: int, size: int) -> bytes:
        buf = ctypes.create_string_buffer(max(1, size))
        self._check(lib().mi355x_bz2_copy_output(self._h, offset, size, buf))
        return buf.raw[:size]

    def pipeline_ms(self) -> float:
        """GPU time of the last batch between HIP events before its first and after its last kernel."""
        ms = ctypes.c_float(0)
        self._check(lib().mi355x_bz2_last_pipeline_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def timings(self) -> dict:
        t = Timings()
        self._check(lib().mi355x_bz2_last_timings(self._h, ctypes.byref(t)))
        return t.as_dict()

    def debug_stage(self, index: int, stage: int) -> bytes:
        if stage >= 3:      # k_hscan hand-off: 3 = group starts (u32 each), 4 = ScanMeta, 5 = selectors
            n = {3: 18048 * 4, 4: 32, 5: 32768}[stage]
        else:
            n = self.last_results[index]["bwt_length"] * (4 if stage == 1 else 1)
        buf = ctypes.create_string_buffer(max(1, n))
        self._check(lib().mi355x_bz2_debug_copy_stage(self._h, index, stage, buf, n))
        return buf.raw[:n]
