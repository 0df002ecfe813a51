/*
 * mi355x_bz2.h -- C ABI of the MI355X-native parallel bzip2 block decoder.
 *
 * This is the drop-in boundary for ONE path of WeGoToMars/indexed_bzip2: decoding independent bzip2 blocks behind
 * ParallelBZ2Reader / ibzip2.open().  Everything is `extern "C"`, plain pointers and sizes.  Each entry point cites
 * the reference interface it replaces (paths relative to the reference repository root).
 *
 * Layers exported here:
 *   1. Block codec on the GPU (the operator seam):      mi355x_bz2_create / _set_input_* / _decode_batch / ...
 *        replaces  BZ2BlockFetcher::decodeBlock          src/indexed_bzip2/BZ2BlockFetcher.hpp:85-138
 *                  (virtual BlockFetcher::decodeBlock    src/core/BlockFetcher.hpp:580-582)
 *        and everything it calls: bzip2::Block           src/indexed_bzip2/bzip2.hpp:145-461, 479-910
 *   2. Host magic-bit scan:                              mi355x_bz2_find_magic
 *        replaces  BitStringFinder<48>::find             src/core/BitStringFinder.hpp:158-285
 *                  ParallelBitStringFinder<48>::find     src/core/ParallelBitStringFinder.hpp:159-265
 *   3. Reader (scheduler + block map + user API):        mi355x_bz2_reader_*
 *        replaces  indexed_bzip2::ParallelBZ2Reader      src/indexed_bzip2/ParallelBZ2Reader.hpp:39-498
 *                  (BZ2ReaderInterface                   src/indexed_bzip2/BZ2ReaderInterface.hpp:15-103)
 *        as bound by the Cython module                   python/indexed_bzip2/indexed_bzip2.pyx:26-67
 *   4. Chunk decoding for rapidgzip:                     mi355x_bz2_decode_chunk
 *        replaces  Bzip2Chunk::decodeChunk               src/rapidgzip/chunkdecoding/Bzip2Chunk.hpp:34-268
 *
 * No exceptions cross this ABI: every reference throw site on the path has a status code below; the host-side
 * scheduler turns a non-OK status back into the reference's behaviour (prefetch failures are silent, on-demand
 * failures surface from read(); src/core/BlockFetcher.hpp:305, 424-432).
 */
#ifndef MI355X_BZ2_H
#define MI355X_BZ2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_BZ2_ABI_VERSION 2

/* ------------------------------------------------------------------------------------------------ status codes */
typedef enum mi355x_bz2_status {
    MI355X_BZ2_OK = 0,
    MI355X_BZ2_ERR_EOF = 1,               /* BitReader::EndOfFileReached            src/core/BitReader.hpp:82-84 */
    MI355X_BZ2_ERR_BAD_MAGIC = 2,         /* "invalid compressed magic"             bzip2.hpp:502-507 */
    MI355X_BZ2_ERR_RANDOMIZED = 3,        /* "isRandomized bit is not supported"    bzip2.hpp:509-512 */
    MI355X_BZ2_ERR_ORIGPTR_RANGE = 4,     /* "origPtr ... larger than buffer size"  bzip2.hpp:514-519 */
    MI355X_BZ2_ERR_GROUP_COUNT = 5,       /* "Invalid Huffman coding group count"   bzip2.hpp:579-583 */
    MI355X_BZ2_ERR_SELECTOR_COUNT = 6,    /* "number of selectors ... invalid"      bzip2.hpp:593-597 */
    MI355X_BZ2_ERR_SELECTOR_UNARY = 7,    /* "Could not find zero termination"      bzip2.hpp:625-629 */
    MI355X_BZ2_ERR_CODE_LENGTH = 8,       /* "start_huffman_length ..."             bzip2.hpp:657-662 */
    MI355X_BZ2_ERR_HUFFMAN_LENGTHS = 9,   /* Error::INVALID_CODE_LENGTHS            bzip2.hpp:680-683 */
    MI355X_BZ2_ERR_SELECTOR_OVERRUN = 10, /* "selector ... out of maximum range"    bzip2.hpp:714-718 */
    MI355X_BZ2_ERR_INVALID_CODE = 11,     /* std::bad_optional_access               bzip2.hpp:723 */
    MI355X_BZ2_ERR_RUN_OVERFLOW = 12,     /* "dbufCount + hh ... > dbufSize"        bzip2.hpp:751-756 */
    MI355X_BZ2_ERR_DATA_OVERFLOW = 13,    /* "dbufCount ... > dbufSize"             bzip2.hpp:776-780 */
    MI355X_BZ2_ERR_ORIGPTR_DATA = 14,     /* "origPtr error"                        bzip2.hpp:794-798 */
    MI355X_BZ2_ERR_CRC = 15,              /* "Calculated CRC ... mismatches"        bzip2.hpp:900-907 */
    MI355X_BZ2_ERR_STREAM_HEADER = 16,    /* readBzip2Header                        bzip2.hpp:114-142 */
    MI355X_BZ2_ERR_STREAM_CRC = 17,       /* "Stream CRC ... does not match"        BZ2Reader.hpp:406-416 */
    MI355X_BZ2_ERR_NO_BLOCK_IN_RANGE = 18,/* rapidgzip::NoBlockInRange              Bzip2Chunk.hpp:262-266 */

    /* errors of this implementation, no reference counterpart */
    MI355X_BZ2_ERR_OUTPUT_CAPACITY = 100,
    MI355X_BZ2_ERR_DEVICE = 101,          /* HIP runtime error (see mi355x_bz2_last_error) */
    MI355X_BZ2_ERR_NO_DEVICE = 102,       /* no gfx950 device / HIP extension unusable: the product path has NO CPU fallback */
    MI355X_BZ2_ERR_INVALID_ARGUMENT = 103,
    MI355X_BZ2_ERR_IO = 104,
    MI355X_BZ2_ERR_CLOSED = 105,
    MI355X_BZ2_ERR_LOGIC = 106
} mi355x_bz2_status;

const char* mi355x_bz2_status_string( int status );
int mi355x_bz2_abi_version( void );

/* ------------------------------------------------------------------------------------------------ 1. block codec */

typedef struct mi355x_bz2_ctx mi355x_bz2_ctx;

typedef struct mi355x_bz2_config {
    int32_t  device;              /* HIP device ordinal; -1 = current device */
    uint32_t max_batch_blocks;    /* initial scratch capacity in blocks (grows on demand); 0 = default 64 */
    uint32_t flags;               /* MI355X_BZ2_FLAG_* */
    uint32_t reserved;
} mi355x_bz2_config;

#define MI355X_BZ2_FLAG_KEEP_STAGES 1u   /* keep per-stage buffers addressable for mi355x_bz2_debug_copy_stage */

/* Mirrors indexed_bzip2::BlockData / BlockHeaderData (src/indexed_bzip2/BZ2BlockFetcher.hpp:18-34); `data` is the
 * byte range [data_offset, data_offset + decoded_size) of the batch output buffer. */
typedef struct mi355x_bz2_block_result {
    uint64_t encoded_offset_bits;   /* BlockHeaderData::encodedOffsetInBits */
    uint64_t encoded_size_bits;     /* BlockHeaderData::encodedSizeInBits */
    uint64_t decoded_size;          /* BlockData::data.size()  (D) */
    uint64_t data_offset;           /* offset of this block's bytes in the batch output buffer */
    uint32_t header_crc;            /* BlockHeaderData::expectedCRC (stream CRC for an EOS block) */
    uint32_t computed_crc;          /* BlockData::calculatedCRC */
    uint32_t bwt_length;            /* N: symbols in dbuf after readBlockData (roofline accounting) */
    uint32_t orig_ptr;
    uint32_t n_symbols;             /* Huffman symbols decoded incl. end-of-block */
    int32_t  is_eos;                /* BlockHeaderData::isEndOfStreamBlock */
    int32_t  is_eof;                /* BlockHeaderData::isEndOfFile */
    int32_t  status;                /* mi355x_bz2_status */
} mi355x_bz2_block_result;

/* Device time of every kernel launch of the last decode_batch, measured with HIP events recorded on the ctx stream
 * around each launch (the stream the kernels run on).  Kernel i is named by mi355x_bz2_kernel_name(i). */
#define MI355X_BZ2_MAX_KERNELS 16
typedef struct mi355x_bz2_timings {
    float    ms_total;                           /* first kernel start .. last kernel end (includes the one host sync) */
    float    ms_kernel_sum;                      /* sum of ms_kernel[] */
    uint32_t n_kernels;
    uint32_t reserved;
    float    ms_kernel[MI355X_BZ2_MAX_KERNELS];
} mi355x_bz2_timings;

const char* mi355x_bz2_kernel_name( uint32_t index );

/* Create / destroy a decoder context bound to one GPU and one HIP stream.
 * Fails with MI355X_BZ2_ERR_NO_DEVICE when no usable gfx950 device exists: there is no CPU fallback. */
int  mi355x_bz2_create( const mi355x_bz2_config* config, mi355x_bz2_ctx** ctx );
/* Optional: pay the one-time cost of the HIP runtime and of loading this library's kernels (0.2 s per process, otherwise
 * part of the first mi355x_bz2_create / the first launch) at a moment of the caller's choosing, e.g. on a thread while
 * the application starts.  No reference counterpart (the reference's thread pool starts lazily too, BlockFetcher.hpp:620);
 * never needed for correctness.  Do not call it in a process that forks workers afterwards. */
int  mi355x_bz2_warmup( int32_t device );
void mi355x_bz2_destroy( mi355x_bz2_ctx* ctx );
const char* mi355x_bz2_last_error( const mi355x_bz2_ctx* ctx );

/* Make the compressed file (or any byte range of it; bit offsets below are relative to `bytes[0]`) resident in HBM.
 * All forms COPY into ctx-owned memory, in file byte order: the kernels read the stream with 16-byte loads and no bounds
 * checks, so the copy is zero padded by >= 256 bytes (they swap each 32-bit word as they load it; there is no swapped
 * copy and no swap pass).  _host copies H2D (pageable or page-locked host memory) and waits; _device copies D2D from a
 * device pointer that is only read during the call (size need not be padded).  The caller's buffer may be freed
 * afterwards; the copy costs `size` bytes of HBM.
 * _host_async only QUEUES the H2D copy (in 64-MiB pieces, on a stream of the context's own) and returns: the next
 * decode_batch[_begin] on this context is ordered behind it, so the transfer overlaps with whatever is running.  It may
 * be called while a batch is in flight on this context -- these are then the bytes of the NEXT batch, copied into a second
 * buffer beside the kernels of the current one (what bench.py does).  `bytes` should be page-locked (hipHostMalloc /
 * hipHostRegister) and must stay valid until that batch's decode_batch_end has returned.
 * Replaces the BitReader/SharedFileReader clone + pread of BZ2BlockFetcher.hpp:89-90. */
int mi355x_bz2_set_input_host( mi355x_bz2_ctx* ctx, const uint8_t* bytes, uint64_t size );
int mi355x_bz2_set_input_host_async( mi355x_bz2_ctx* ctx, const uint8_t* bytes, uint64_t size );
/* _host_streamed returns at once and copies in the background, in 32-MiB pieces on a thread and a stream of its own
 * (pageable memory is fine: a memory-mapped file): every decode_batch[_begin] waits only for the pieces in which its
 * blocks lie, so the first blocks of a large file decode while the rest is still on its way
 * (mi355x_bz2_find_magic_device needs all of it and waits for all of it).  `bytes` must stay valid until
 * mi355x_bz2_input_resident returns 1, the next set_input_* or the destruction of the context.  Contexts that share the
 * input (mi355x_bz2_share_input) share the copy in progress. */
int mi355x_bz2_set_input_host_streamed( mi355x_bz2_ctx* ctx, const uint8_t* bytes, uint64_t size );
int mi355x_bz2_input_resident( const mi355x_bz2_ctx* ctx );
int mi355x_bz2_set_input_device( mi355x_bz2_ctx* ctx, const void* device_bytes, uint64_t size );
/* Several contexts over ONE resident copy of the input (the reader keeps two contexts to overlap consecutive batches):
 * `ctx` decodes from the bytes `from` made resident.  Nothing is copied; `from` must outlive `ctx`'s use of them and
 * keep its input unchanged.  Both contexts must be on the same device. */
int mi355x_bz2_share_input( mi355x_bz2_ctx* ctx, mi355x_bz2_ctx* from );

/* A batch holds at most this many blocks (grid dimensions and 16-bit segment ids are sized for it; 65 535 level-9
 * blocks are 59 GB of decoded data and 850 GB of scratch): larger requests fail with MI355X_BZ2_ERR_INVALID_ARGUMENT. */
#define MI355X_BZ2_MAX_BATCH_BLOCKS 65535u

/* Decode n_blocks independent blocks whose magic starts at block_bit_offsets[i] (from the finder or the index).
 * = n calls of BZ2BlockFetcher::decodeBlock (BZ2BlockFetcher.hpp:85-138).  An offset pointing at an EOS magic
 * returns is_eos=1 and no data (ibid. :101-104).  results[i].status reports per-block failures; the function's
 * own return value is non-OK only for argument/device failures.  Decoded bytes stay in HBM (ctx-owned, valid
 * until the next decode_batch/destroy); *total_decoded = sum of decoded_size. */
int mi355x_bz2_decode_batch( mi355x_bz2_ctx* ctx, const uint64_t* block_bit_offsets, uint32_t n_blocks,
                             mi355x_bz2_block_result* results, uint64_t* total_decoded );

/* The same in two halves, for callers that keep the GPU busy across batches.
 *   _begin plans the batch and queues ALL of it without waiting: the group-start scan, symbols, MTF, table build, walk,
 *          the output offsets (computed on the device, k_offsets), the run-length expansion into the output buffer, the
 *          CRC, and the copy of the block records to the host.  The output buffer is chosen here for 900 000 bytes per
 *          block; a mi355x_bz2_hold_output_until event given before _begin is waited for by the output kernels only.
 *   _end   waits for the batch and fills `results` (n entries, in the order given to _begin).  Only when the batch
 *          decoded to more than the buffer chosen in _begin (blocks beyond the usual 900 000 bytes) does it repeat the
 *          expansion and the CRC with a buffer of the right size.
 * One batch per context can be in flight; contexts used in turn (begin(A), begin(B), end(A), begin(A'), end(B), ...)
 * overlap the latency-bound first kernels of one batch with the throughput kernels of the others.  The input of the NEXT
 * batch may be queued while one is in flight (mi355x_bz2_set_input_host_async: second input buffer, stream of its own).
 * A context drives up to nine HIP streams (up to four block groups and, for small batches, a side stream each for the
 * second k_mtf instance; one stream each for input copies, output copies and the device magic scan): set
 * GPU_MAX_HW_QUEUES (16 measured best for four contexts) before the HIP runtime starts, or streams share hardware queues
 * and serialize. */
int mi355x_bz2_decode_batch_begin( mi355x_bz2_ctx* ctx, const uint64_t* block_bit_offsets, uint32_t n_blocks );
int mi355x_bz2_decode_batch_end( mi355x_bz2_ctx* ctx, mi355x_bz2_block_result* results, uint64_t* total_decoded );

/* Device pointer of the last batch's ragged output buffer (block i at data_offset). */
const void* mi355x_bz2_output_device( const mi355x_bz2_ctx* ctx );
/* Copy [offset, offset+size) of the last batch's output to host memory (D2H). */
int mi355x_bz2_copy_output( mi355x_bz2_ctx* ctx, uint64_t offset, uint64_t size, void* host_dst );
/* The same copy in the background, on a stream of its own: _begin (after decode_batch_end, before the next
 * decode_batch_begin) queues it and returns, _end waits for the copy started last.  The context then writes its next
 * batch into a second output buffer, so the next decode_batch_begin may follow at once: the copy of batch k overlaps the
 * kernels of batch k + 1 (the reader's per-context loop; ParallelBZ2Reader gets the same overlap from its thread pool,
 * BlockFetcher.hpp:620-642).  `host_dst` should be page-locked and must stay valid until _end has returned.
 * mi355x_bz2_output_device keeps pointing at the batch finished last. */
int mi355x_bz2_copy_output_begin( mi355x_bz2_ctx* ctx, uint64_t offset, uint64_t size, void* host_dst );
/* For callers that read mi355x_bz2_output_device themselves, asynchronously (a collective that sends the decoded extent
 * to another GPU): `hip_event` (a hipEvent_t recorded behind that read) is waited for by the context's stream before the
 * NEXT batch writes its output -- not before its other kernels.  The event is not owned and must stay valid until the next
 * decode_batch_end has returned.  Call between decode_batch_end and the next decode_batch_begin. */
int mi355x_bz2_hold_output_until( mi355x_bz2_ctx* ctx, void* hip_event );
int mi355x_bz2_copy_output_end( mi355x_bz2_ctx* ctx );
int mi355x_bz2_last_timings( const mi355x_bz2_ctx* ctx, mi355x_bz2_timings* timings );
/* Only the duration of the last batch's kernel pipeline (HIP events before the first and after the last kernel): one
 * event query instead of the ~90 that the per-kernel breakdown of mi355x_bz2_last_timings needs. */
int mi355x_bz2_last_pipeline_ms( const mi355x_bz2_ctx* ctx, float* milliseconds );
/* The hipStream_t the context launches on (as void*), so callers can order their own work after it. */
void* mi355x_bz2_stream( const mi355x_bz2_ctx* ctx );
/* Device memory the context holds right now: the per-block scratch (sized by the largest batch so far) and its output
 * buffers (sized by the largest batch's decoded bytes; two once a background copy has been used).  Either pointer may
 * be NULL.  No counterpart in the reference (its decoders keep ~5 bytes per decoded byte on the heap, bzip2.hpp:425-441). */
int mi355x_bz2_device_memory( const mi355x_bz2_ctx* ctx, uint64_t* scratch_bytes, uint64_t* output_bytes );

/* Debug/parity hook (needs MI355X_BZ2_FLAG_KEEP_STAGES -- without it stages 0 and 2 share their memory and are refused): copy an intermediate stage of block `index` of the last
 * batch to host. stage 0 = L column (N bytes, bzip2.hpp:789 dbuf low bytes), 1 = packed LF table (4N bytes),
 * 2 = inverse-BWT output before RLE1 (N bytes). */
int mi355x_bz2_debug_copy_stage( mi355x_bz2_ctx* ctx, uint32_t index, int stage, void* host_dst, uint64_t capacity );

/* bzip2's CRC-32 (MSB-first, polynomial 0x04C11DB7, createCRC32LookupTable / updateCRC32, bzip2.hpp:59-91, with the
 * initial value and final inversion of bzip2.hpp:833, 901) of `n_pieces` consecutive pieces of a DEVICE buffer: piece i
 * is the `sizes[i]` bytes behind the pieces in front of it.  For consumers of decoded extents that sit on another GPU
 * (the gather of bench.py): the receiver recomputes every block's checksum over the bytes that ARRIVED and compares it
 * with the checksum the sender's block record carries.  `device_bytes` must be 16-byte aligned; runs on the context's
 * stream and waits for the result; no batch may be in flight on the context. */
int mi355x_bz2_crc32_device( mi355x_bz2_ctx* ctx, const void* device_bytes, const uint64_t* sizes, uint32_t n_pieces,
                             uint32_t* crcs );

/* ------------------------------------------------------------------------------------------------ 2. magic scan */

#define MI355X_BZ2_MAGIC_BLOCK 0x314159265359ULL   /* bzip2.hpp:103 */
#define MI355X_BZ2_MAGIC_EOS   0x177245385090ULL   /* bzip2.hpp:104 */

/* All bit offsets (ascending) at which the 48-bit pattern occurs in bytes[0,size).  Returns the number found; at most
 * `capacity` are written.  `threads` = 0 picks the host's core count.
 * Replaces ParallelBitStringFinder<48>::find (src/core/ParallelBitStringFinder.hpp:159-265). */
uint64_t mi355x_bz2_find_magic( const uint8_t* bytes, uint64_t size, uint64_t magic48,
                                uint64_t* bit_offsets, uint64_t capacity, uint32_t threads );

/* The same scan on the GPU over the input made resident with mi355x_bz2_set_input_* (ascending offsets; at most
 * `capacity` are written, *n_found = number of matches).  At GPU decode rates the host scan would otherwise be the
 * critical path. */
int mi355x_bz2_find_magic_device( mi355x_bz2_ctx* ctx, uint64_t magic48, uint64_t* bit_offsets, uint64_t capacity,
                                  uint64_t* n_found );

/* bzip2::readBzip2Header (bzip2.hpp:114-142) at a byte-aligned bit offset: returns level 1..9, or 0 if invalid. */
int mi355x_bz2_read_stream_header( const uint8_t* bytes, uint64_t size, uint64_t bit_offset );

/* ------------------------------------------------------------------------------------------------ 3. reader */

typedef struct mi355x_bz2_reader mi355x_bz2_reader;

/* ParallelBZ2Reader( filePath | fd | memory, parallelization )    ParallelBZ2Reader.hpp:50-89
 * parallelization = number of blocks kept in flight per GPU batch (0 = default). */
int mi355x_bz2_reader_open_path( const char* path, uint32_t parallelization, int32_t device, mi355x_bz2_reader** r );
int mi355x_bz2_reader_open_fd( int fd, uint32_t parallelization, int32_t device, mi355x_bz2_reader** r );
int mi355x_bz2_reader_open_memory( const uint8_t* bytes, uint64_t size, uint32_t parallelization, int32_t device,
                                   mi355x_bz2_reader** r );
void mi355x_bz2_reader_close( mi355x_bz2_reader* r );                        /* close()        :104-111 */
const char* mi355x_bz2_reader_last_error( const mi355x_bz2_reader* r );

/* read( fd, buffer, n ): writes to `fd` if fd >= 0, else copies to `buffer` if non-NULL, else discards
 * (BZ2ReaderInterface.hpp:35-57 + ParallelBZ2Reader.hpp:167-269).  *n_read = bytes produced. */
int mi355x_bz2_reader_read( mi355x_bz2_reader* r, int fd, void* buffer, uint64_t n_bytes, uint64_t* n_read );
/* seek( offset, whence ) with SEEK_SET/SEEK_CUR/SEEK_END          ParallelBZ2Reader.hpp:271-325 */
int mi355x_bz2_reader_seek( mi355x_bz2_reader* r, int64_t offset, int whence, uint64_t* new_position );
uint64_t mi355x_bz2_reader_tell( const mi355x_bz2_reader* r );                /* tell()         :129-142 */
int      mi355x_bz2_reader_eof( const mi355x_bz2_reader* r );                 /* eof()          :119-123 */
int      mi355x_bz2_reader_closed( const mi355x_bz2_reader* r );              /* closed()       :113-117 */
/* size(): returns 1 and *size if the block map is finalized, else 0             :144-151 */
int      mi355x_bz2_reader_size( const mi355x_bz2_reader* r, uint64_t* size );
uint64_t mi355x_bz2_reader_tell_compressed( const mi355x_bz2_reader* r );     /* tellCompressed :385-393 */
int      mi355x_bz2_reader_block_offsets_complete( const mi355x_bz2_reader* r ); /*             :329-333 */

/* blockOffsets() (forces a full decode) / availableBlockOffsets(): two-call protocol -- pass capacity 0 to get the
 * count in *n, then call again with arrays of that size.                         :339-363 */
int mi355x_bz2_reader_block_offsets( mi355x_bz2_reader* r, uint64_t* bits, uint64_t* bytes, uint64_t capacity,
                                     uint64_t* n );
int mi355x_bz2_reader_available_block_offsets( const mi355x_bz2_reader* r, uint64_t* bits, uint64_t* bytes,
                                               uint64_t capacity, uint64_t* n );
/* setBlockOffsets( map )                                                       :365-378 */
int mi355x_bz2_reader_set_block_offsets( mi355x_bz2_reader* r, const uint64_t* bits, const uint64_t* bytes,
                                         uint64_t n );
/* joinThreads()                                                                :404-409 */
int mi355x_bz2_reader_join_threads( mi355x_bz2_reader* r );

/* Check of the combined CRC in every end-of-stream block against the block CRCs decoded in front of it
 * (crc = rotl( crc, 1 ) ^ blockCRC, BZ2Reader.hpp:481-484; a mismatch fails the read with MI355X_BZ2_ERR_STREAM_CRC as
 * BZ2Reader.hpp:406-416 throws).  The reference only does this in its serial reader, which is what parallelization
 * == 1 selects there; ParallelBZ2Reader never checks.  Same default here: on for parallelization == 1, else off.
 * The check covers streams whose blocks are decoded for the first time in order, i.e. not after set_block_offsets. */
int mi355x_bz2_reader_set_verify_stream_crc( mi355x_bz2_reader* r, int enable );
/* number of end-of-stream CRCs that have been compared (and matched) so far */
uint64_t mi355x_bz2_reader_streams_verified( const mi355x_bz2_reader* r );

/* BlockFetcher::Statistics subset (src/core/BlockFetcher.hpp:52-173) */
typedef struct mi355x_bz2_reader_stats {
    uint64_t gets, cache_hits, prefetch_hits, on_demand_fetches, prefetches_submitted, batches, blocks_decoded;
    uint64_t failed_prefetches;
    double   decode_seconds, wait_seconds;
    /* (ABI 2) 1: the whole compressed file is kept on the GPU; 0: bounded residency -- the file does not fit beside the
     * decoders' scratch, or exceeds MI355X_BZ2_INPUT_BUDGET bytes (environment), and every launch copies the byte range of
     * its own blocks (the reference streams through 128 KiB refills, src/core/BitReader.hpp:57) */
    uint64_t input_resident;
    uint64_t input_bytes_uploaded;   /* bounded residency: compressed bytes copied to the GPU so far, all launches */
} mi355x_bz2_reader_stats;
int mi355x_bz2_reader_statistics( const mi355x_bz2_reader* r, mi355x_bz2_reader_stats* stats );

/* ------------------------------------------------------------------------------------------------ 4. chunk decoding */

/* Counterpart of rapidgzip's bzip2 chunk decoder, Bzip2Chunk<ChunkData>::decodeChunk
 * (src/rapidgzip/chunkdecoding/Bzip2Chunk.hpp:215-268) and decodeUnknownBzip2Chunk (:34-212): decodes the run of
 * consecutive blocks that starts at chunk_offset_bits -- or, if nothing decodes there, at the first block magic behind it
 * that does -- through end-of-stream blocks and the headers of following streams, up to (excluding) the first block that
 * starts at or behind until_offset_bits, or until max_decoded_bytes have been produced (stopped_preemptively).
 * `bytes` is the host view of the input that was given to mi355x_bz2_set_input_* (magic scan and headers are read on the
 * host, all blocks of the range are decoded in ONE GPU batch).  The chunk's bytes are
 * [data_offset, data_offset + decoded_size) of mi355x_bz2_output_device / mi355x_bz2_copy_output.
 * blocks[k] (k < n_blocks): the records of the chunk's data blocks, data_offset relative to the chunk -- the
 * {encoded offset, decoded offset} pairs are what ChunkData::appendDeflateBlockBoundary receives; footers[k]: the position
 * behind every end-of-stream block and the decoded size so far (ChunkData::appendFooter).
 * result->status: MI355X_BZ2_OK or MI355X_BZ2_ERR_NO_BLOCK_IN_RANGE (the reference throws NoBlockInRange). */
typedef struct mi355x_bz2_chunk_boundary {
    uint64_t encoded_offset_bits;
    uint64_t decoded_offset;
} mi355x_bz2_chunk_boundary;

typedef struct mi355x_bz2_chunk_result {
    uint64_t encoded_offset_bits;   /* where the chunk really starts */
    uint64_t encoded_end_bits;      /* ChunkData::finalize( nextBlockOffset ) */
    uint64_t decoded_size;
    uint64_t data_offset;           /* of the chunk in the context's output buffer */
    uint32_t n_blocks;
    uint32_t n_footers;
    int32_t  stopped_preemptively;
    int32_t  status;
} mi355x_bz2_chunk_result;

int mi355x_bz2_decode_chunk( mi355x_bz2_ctx* ctx, const uint8_t* bytes, uint64_t size,
                             uint64_t chunk_offset_bits, uint64_t until_offset_bits, uint64_t max_decoded_bytes,
                             mi355x_bz2_chunk_result* result,
                             mi355x_bz2_block_result* blocks, uint32_t blocks_capacity,
                             mi355x_bz2_chunk_boundary* footers, uint32_t footers_capacity );

#ifdef __cplusplus
}
#endif
#endif /* MI355X_BZ2_H */
