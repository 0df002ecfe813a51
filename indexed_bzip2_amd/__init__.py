"""indexed_bzip2_amd -- MI355X-native parallel bzip2 block decoder behind the indexed_bzip2 API.

Host-side mirror of python/indexed_bzip2/indexed_bzip2.pyx (open, IndexedBzip2File, ...) over the C ABI of
include/mi355x_bz2.h.  All decoding happens in hand-written HIP kernels on gfx950; there is no CPU fallback.
"""
__version__ = "0.1.0"

import os as _os

# the reader drives two or three decoder contexts with up to eight HIP streams each (block groups, input, copy-out): the
# runtime's default of 4 hardware queues serialises them (measured: 8 queues 9.9 GB/s, 24 queues 11.1 GB/s for a cold 8 GiB
# read; the bench at 8 queues 85.8 ms per step, at 16 or more 67).  The HIP runtime reads this when it starts, so it only
# helps if nothing has touched the GPU yet; an existing setting wins
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from ._native import Bz2Error, Decoder, find_magic, lib, status_string, warmup  # noqa: F401
from .reader import (IndexedBzip2File, IndexedBzip2FileRaw, open, read_block_offsets,  # noqa: F401
                     write_block_offsets)

if _os.environ.get("MI355X_BZ2_WARMUP") == "1":     # opt-in: see warmup()
    warmup()
