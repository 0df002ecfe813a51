"""indexed_bzip2_amd -- MI355X-native parallel bzip2 block decoder behind the indexed_bzip2 API.

Host-side mirror of python/indexed_bzip2/indexed_bzip2.pyx (open, IndexedBzip2File, ...) over the C ABI of
include/mi355x_bz2.h.  All decoding happens in hand-written HIP kernels on gfx950; there is no CPU fallback.
"""
__version__ = "0.1.0"

import os as _os

# the reader drives two or three decoder contexts with up to eight HIP streams each (block groups, input, copy-out): the
# runtime's default of 4 hardware queues serialises them.  Measured: the bench (four contexts, 2 560 blocks per batch) 85.8 ms
# per step with 8 queues, 67 with 16 or more; batches of 310 blocks on four contexts 12.4 ms per step with 16 queues but 17.3
# with 24 or 32.  The HIP runtime reads this when it starts, so it only helps if nothing has touched the GPU yet; an
# existing setting wins
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from ._native import Bz2Error, Decoder, find_magic, lib, status_string, warmup  # noqa: F401
from .reader import (IndexedBzip2File, IndexedBzip2FileRaw, open, read_block_offsets,  # noqa: F401
                     write_block_offsets)

if _os.environ.get("MI355X_BZ2_WARMUP") == "1":     # opt-in: see warmup()
    warmup()
