"""Host-side mirror of the reference's Python module (python/indexed_bzip2/indexed_bzip2.pyx:87-345).

Same names and argument meaning: open(), IndexedBzip2File(io.BufferedReader), IndexedBzip2FileRaw(io.RawIOBase) with
tell_compressed, block_offsets, set_block_offsets, block_offsets_complete, available_block_offsets, size,
join_threads.  The C++ ParallelBZ2Reader the Cython classes wrap is replaced by mi355x_bz2_reader_* (C ABI), whose
blocks are decoded on the GPU.  `parallelization` keeps the reference's meaning and default
(indexed_bzip2.pyx:293, 321, 340): 1 (default) = no parallelism, i.e. one block per GPU launch and -- like the serial
BZ2Reader that the reference selects for 1 -- the stream CRC of every end-of-stream block is verified
(BZ2Reader.hpp:406-416); 0 = as much as the machine wants (reference: all cores; here: the default batch of 512
blocks); N > 1 = N blocks per GPU batch, no stream-CRC check (ParallelBZ2Reader never checks it).  There is no CPU
reader: every value decodes on the GPU.
"""
import ctypes
import builtins
import io
import os

from . import _native as N


def _has_valid_fileno(file):
    # indexed_bzip2.pyx:78-84
    try:
        fileno = file.fileno()
        return isinstance(fileno, int) and fileno >= 0
    except Exception:
        return False


def _is_file_object(file):
    # indexed_bzip2.pyx:69-76
    return all(hasattr(file, name) for name in ("read", "seekable", "seek", "tell"))


class _IndexedBzip2FileParallel:
    """Mirror of cdef class _IndexedBzip2FileParallel (indexed_bzip2.pyx:186-288)."""

    def __init__(self, file, parallelization=1, device=-1):
        if not isinstance(parallelization, int):
            raise TypeError(f"Parallelization argument must be an integer not '{parallelization}'!")
        self._h = ctypes.c_void_p()
        self._keepalive = None
        L = N.lib()
        if isinstance(file, int):
            rc = L.mi355x_bz2_reader_open_fd(file, parallelization, device, ctypes.byref(self._h))
        elif _has_valid_fileno(file):
            rc = L.mi355x_bz2_reader_open_fd(file.fileno(), parallelization, device, ctypes.byref(self._h))
        elif _is_file_object(file):
            # pure-Python file object (the reference wraps it in PythonFileReader, filereader/Python.hpp:321-585):
            # the compressed bytes are pulled once; decoding needs them resident in HBM anyway
            pos = file.tell() if file.seekable() else None
            if pos is not None:
                file.seek(0)
            data = file.read()
            if pos is not None:
                file.seek(pos)
            rc = L.mi355x_bz2_reader_open_memory(data, len(data), parallelization, device, ctypes.byref(self._h))
        elif isinstance(file, (str, os.PathLike)):
            rc = L.mi355x_bz2_reader_open_path(os.fsencode(file), parallelization, device, ctypes.byref(self._h))
        else:
            raise Exception("Expected file name string, file descriptor integer, "
                            "or file-like object for ParallelBZ2Reader!")
        if rc != N.OK:
            self._h = ctypes.c_void_p()
            raise N.Bz2Error(rc)

    # -- helpers
    def _check(self, rc):
        if rc != N.OK:
            detail = N.lib().mi355x_bz2_reader_last_error(self._h).decode(errors="replace")
            if rc == 103:
                raise ValueError(detail or N.status_string(rc))
            raise N.Bz2Error(rc, detail)

    def _require(self):
        if not self._h:
            raise Exception("Invalid file object!")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if self._h:
            N.lib().mi355x_bz2_reader_close(self._h)
            self._h = ctypes.c_void_p()

    def closed(self):
        return (not self._h) or bool(N.lib().mi355x_bz2_reader_closed(self._h))

    def seekable(self):
        self._require()
        return True

    def readinto(self, bytes_like):
        self._require()
        view = memoryview(bytes_like).cast("B")
        n = len(view)
        if n == 0:
            return 0
        buf = (ctypes.c_char * n).from_buffer(view)
        got = ctypes.c_uint64()
        self._check(N.lib().mi355x_bz2_reader_read(self._h, -1, buf, n, ctypes.byref(got)))
        return got.value

    def read_to_fd(self, fd, n=2**64 - 1):
        """read( fd, nullptr, n ): decode straight into a file descriptor (BZ2ReaderInterface.hpp:35-57)."""
        self._require()
        got = ctypes.c_uint64()
        self._check(N.lib().mi355x_bz2_reader_read(self._h, fd, None, n, ctypes.byref(got)))
        return got.value

    def seek(self, offset, whence=io.SEEK_SET):
        self._require()
        pos = ctypes.c_uint64()
        self._check(N.lib().mi355x_bz2_reader_seek(self._h, offset, whence, ctypes.byref(pos)))
        return pos.value

    def tell(self):
        self._require()
        return N.lib().mi355x_bz2_reader_tell(self._h)

    def size(self):
        self._require()
        s = ctypes.c_uint64()
        return s.value if N.lib().mi355x_bz2_reader_size(self._h, ctypes.byref(s)) else 0

    def tell_compressed(self):
        self._require()
        return N.lib().mi355x_bz2_reader_tell_compressed(self._h)

    def block_offsets_complete(self):
        self._require()
        return bool(N.lib().mi355x_bz2_reader_block_offsets_complete(self._h))

    def _offsets(self, fn):
        n = ctypes.c_uint64()
        self._check(fn(self._h, None, None, 0, ctypes.byref(n)))
        bits = (ctypes.c_uint64 * max(1, n.value))()
        byts = (ctypes.c_uint64 * max(1, n.value))()
        self._check(fn(self._h, bits, byts, n.value, ctypes.byref(n)))
        return {bits[i]: byts[i] for i in range(n.value)}

    def block_offsets(self):
        self._require()
        return self._offsets(N.lib().mi355x_bz2_reader_block_offsets)

    def available_block_offsets(self):
        self._require()
        return self._offsets(N.lib().mi355x_bz2_reader_available_block_offsets)

    def set_block_offsets(self, offsets):
        self._require()
        items = sorted(dict(offsets).items())
        n = len(items)
        bits = (ctypes.c_uint64 * max(1, n))(*[k for k, _ in items])
        byts = (ctypes.c_uint64 * max(1, n))(*[v for _, v in items])
        self._check(N.lib().mi355x_bz2_reader_set_block_offsets(self._h, bits, byts, n))

    def join_threads(self):
        self._require()
        self._check(N.lib().mi355x_bz2_reader_join_threads(self._h))

    def set_verify_stream_crc(self, enable: bool):
        """Check every end-of-stream CRC against the block CRCs in front of it (default: only with parallelization=1,
        like the reference, whose serial reader checks and whose parallel reader does not)."""
        self._require()
        self._check(N.lib().mi355x_bz2_reader_set_verify_stream_crc(self._h, 1 if enable else 0))

    def streams_verified(self) -> int:
        self._require()
        return int(N.lib().mi355x_bz2_reader_streams_verified(self._h))

    def statistics(self):
        self._require()
        st = N.ReaderStats()
        self._check(N.lib().mi355x_bz2_reader_statistics(self._h, ctypes.byref(st)))
        return st.as_dict()


class IndexedBzip2FileRaw(io.RawIOBase):
    """indexed_bzip2.pyx:290-317"""

    def __init__(self, filename, parallelization=1, device=-1):
        self.bz2reader = _IndexedBzip2FileParallel(filename, parallelization, device)
        self.name = filename
        self.mode = "rb"

        self.readinto = self.bz2reader.readinto
        self.seek = self.bz2reader.seek
        self.tell = self.bz2reader.tell
        self.seekable = self.bz2reader.seekable
        self.join_threads = self.bz2reader.join_threads

    def close(self):
        if self.closed:
            return
        super().close()
        self.bz2reader.close()

    def readable(self):
        return True


class IndexedBzip2File(io.BufferedReader):
    """indexed_bzip2.pyx:320-337"""

    def __init__(self, filename, parallelization=1, device=-1):
        fobj = IndexedBzip2FileRaw(filename, parallelization, device)
        self.bz2reader = fobj.bz2reader

        self.tell_compressed = self.bz2reader.tell_compressed
        self.block_offsets = self.bz2reader.block_offsets
        self.set_block_offsets = self.bz2reader.set_block_offsets
        self.block_offsets_complete = self.bz2reader.block_offsets_complete
        self.available_block_offsets = self.bz2reader.available_block_offsets
        self.size = self.bz2reader.size
        self.join_threads = self.bz2reader.join_threads
        self.statistics = self.bz2reader.statistics
        self.set_verify_stream_crc = self.bz2reader.set_verify_stream_crc
        self.streams_verified = self.bz2reader.streams_verified

        super().__init__(fobj, buffer_size=1024**2)


builtins_open = builtins.open


def open(filename, parallelization=1, device=-1):
    """
    filename: can be a file path, a file descriptor, or a file object
              with suitable read, seekable, seek, and tell methods.          (indexed_bzip2.pyx:340-345)
    parallelization: 1 (default, as in the reference) = block by block with the stream-CRC check of the reference's
              serial reader; 0 = default GPU batch (512 blocks); N = N blocks per GPU batch.
    """
    return IndexedBzip2File(filename, parallelization, device)


def write_block_offsets(offsets, file):
    """Block map as text, one "<compressed bit offset>,<decoded byte offset>" per line: the format `ibzip2 -L` writes
    (src/tools/ibzip2.cpp:83-93) and `ibzip2-mi355x -L` reproduces.  `file` is a path or a text file object."""
    text = "".join(f"{int(bits)},{int(byts)}\n" for bits, byts in sorted(offsets.items()))
    if hasattr(file, "write"):
        file.write(text)
    else:
        with builtins_open(file, "w") as f:
            f.write(text)


def read_block_offsets(file):
    """Inverse of write_block_offsets: returns the dict that set_block_offsets() takes."""
    if hasattr(file, "read"):
        text = file.read()
    else:
        with builtins_open(file, "r") as f:
            text = f.read()
    if isinstance(text, bytes):
        text = text.decode("ascii")
    offsets = {}
    for number, line in enumerate(text.splitlines(), 1):
        line = line.strip()
        if not line:
            continue
        parts = line.split(",")
        if len(parts) != 2:
            raise ValueError(f"line {number}: expected '<compressed bits>,<decoded bytes>', got {line!r}")
        offsets[int(parts[0])] = int(parts[1])
    return offsets
