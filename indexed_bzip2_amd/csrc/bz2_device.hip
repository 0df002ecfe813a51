/**
 * bz2_device.hip -- decoder context + batch launcher behind the C ABI of include/mi355x_bz2.h (section 1).
 *
 * Replaces BZ2BlockFetcher::decodeBlock (src/indexed_bzip2/BZ2BlockFetcher.hpp:85-138): instead of one block per
 * host thread, a whole batch of independent blocks is pushed through the kernels of bz2_kernels.hip.h on one HIP
 * stream.  Per-block scratch lives in HBM and is sized for the largest block the format allows (900 000 symbols):
 *   L column 0.9 MB, packed LF table 4 MiB, pre-RLE1 stream 0.9 MB, selectors 32 KiB, segment records ~100 KB.
 * There is NO CPU fallback: without a usable gfx950 device every entry point fails with MI355X_BZ2_ERR_NO_DEVICE.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi355x_bz2.h"
#include "bz2_kernels.hip.h"
#include "bz2_stage1.hip.h"
#include "bz2_hscan.hip.h"
#include "bz2_walk.hip.h"

using namespace bz2gpu;

constexpr uint32_t MAX_CHUNKS = 3;          /* groups of cheap blocks; one more stream than hardware queues (4 by
                                               default) would serialize two groups */
constexpr int MAX_GROUPS = MAX_CHUNKS + 1;   /* + the expensive group */
/* register budgets (wavefronts per SIMD that a kernel's registers leave room for, see k_hscan) and groups per chunk of k_hsym:
 * the other values that round 3 built (2 / 5, 2 / 3, 128 / 512) made no difference for a step (profiles/r03_ab_registers.txt) */
constexpr uint32_t BWT_SPLIT_BLOCKS = 640;   /* batches up to this size build their tables with several workgroups per block */
constexpr uint32_t REGS_SCAN = 4, SYM_GROUPS = 256, REGS_MTF = 4;

/** A host -> HBM copy of the input that runs in pieces on a thread and a stream of its own
 * (mi355x_bz2_set_input_host_streamed): a batch only waits for the piece in which its last block ends.  Shared by the
 * contexts that share the input. */
struct InputUpload
{
    static constexpr uint64_t PIECE = uint64_t( 32 ) << 20;
    static constexpr uint64_t HEAD = uint64_t( 64 ) << 20;   /* files beyond this get their beginning a second time, see below */

    /* Allocating the copy of a multi-GB file takes longer (page clearing, up to 70 ms per GB) than decoding its first
     * blocks: the first HEAD bytes go into a small buffer of their own first, from which batches that lie wholly inside
     * decode while the full buffer is still being allocated and filled by the thread. */
    uint8_t* head{ nullptr };
    uint64_t headBytes{ 0 };
    bool headQueued{ false };
    hipEvent_t headDone{ nullptr };
    uint8_t* main{ nullptr };      /* the whole file; allocated by the thread */

    std::thread worker;
    std::mutex mutex;
    std::condition_variable changed;
    uint64_t queued{ 0 };          /* bytes whose copy is queued on `stream`, with done[piece] recorded behind it */
    uint64_t total{ 0 };
    bool failed{ false };
    int device{ 0 };
    hipStream_t stream{ nullptr };
    std::vector<hipEvent_t> done;

    ~InputUpload()
    {
        if ( worker.joinable() ) worker.join();
        (void)hipSetDevice( device );
        if ( stream ) (void)hipStreamSynchronize( stream );
        for ( auto& e : done ) {
            if ( e ) (void)hipEventDestroy( e );
        }
        if ( headDone ) (void)hipEventDestroy( headDone );
        if ( stream ) (void)hipStreamDestroy( stream );
        (void)hipFree( head );
        (void)hipFree( main );
    }
};

struct mi355x_bz2_ctx
{
    int device{ 0 };
    uint32_t flags{ 0 };
    hipStream_t stream{ nullptr };
    hipStream_t gstream[MAX_GROUPS]{};   /* gstream[0] == stream; higher groups = more expensive blocks, higher priority */
    /* small batches: the two k_mtf instances of a group (each block belongs to one of them) side by side, see begin */
    hipStream_t sideStream[MAX_GROUPS]{};
    hipEvent_t evFork[MAX_GROUPS]{}, evJoin[MAX_GROUPS]{};
    std::string lastError;
    mutable std::mutex mutex;

    /* input */
    /* ctx-owned copies of the input.  A copy queued while a batch is in flight (set_input_host_async: the bytes of the
     * NEXT batch, beside the kernels of this one) goes into the second buffer, on a stream of its own */
    struct InBuffer
    {
        uint8_t* bytes{ nullptr };
        uint64_t capacity{ 0 };
    };
    InBuffer in[2];
    int inCurrent{ 0 };                 /* the buffer c->dIn points into, if it is ctx-owned */
    int inFlight{ -1 };                 /* the buffer the batch in flight reads, -1 if none of the two */
    hipStream_t inStream{ nullptr };
    hipEvent_t inReady{ nullptr };      /* behind the last copy queued on inStream */
    bool inPending{ false };            /* a copy has been queued that no batch has been ordered behind yet */
    const uint8_t* dIn{ nullptr };
    uint64_t inSize{ 0 };
    std::shared_ptr<InputUpload> upload;   /* set while / after a streamed copy of the input */

    /* Buffers that have been replaced by larger ones.  hipFree / hipHostFree wait for the whole device, i.e. for the
     * batches of every other context: a growing buffer is put aside instead and freed when the context goes, or when
     * what has been put aside outweighs what is in use (then the wait is paid once, see retire). */
    struct Retired
    {
        void* pointer{ nullptr };
        uint64_t bytes{ 0 };
        bool host{ false };
    };
    std::vector<Retired> retired;
    uint64_t retiredBytes{ 0 };

    /* per-block scratch, capacity in blocks: one device and one page-locked host allocation, carved up by ensureScratch */
    uint32_t capacity{ 0 };
    uint64_t scratchBytes{ 0 }, scratchHostBytes{ 0 };
    uint8_t* dScratch{ nullptr };
    uint8_t* hScratch{ nullptr };
    uint64_t* dOffsets{ nullptr };
    uint32_t* dOrder{ nullptr };
    uint32_t* hOrder{ nullptr };       /* pinned */
    BlockMeta* dMeta{ nullptr };
    uint8_t* dSel{ nullptr };
    uint16_t* dSym{ nullptr };
    uint8_t* dStb{ nullptr };
    HuffMeta* dHmeta{ nullptr };
    ScanMeta* dSmeta{ nullptr };      /* k_hscan -> k_hsym */
    HuffTables* dHtab{ nullptr };     /* decode tables per block */
    uint32_t* dGpos{ nullptr };       /* [cap][GPOS_STRIDE]: bit position of every 50-symbol group */
    uint8_t* dL{ nullptr };
    uint32_t* dTab{ nullptr };
    uint8_t* dR{ nullptr };
    uint32_t* dSegLen{ nullptr };
    uint32_t* dSegSucc{ nullptr };
    uint32_t* dSegCont{ nullptr };    /* [cap][SEG_STRIDE]: where a segment longer than STASH_BYTES goes on */
    uint2*    dChain{ nullptr };      /* [cap][SEG_STRIDE]: segments in cycle order {offset, length, segment} */
    uint32_t* dStash{ nullptr };      /* [cap][SEG_STRIDE][STASH_BYTES / 4]: first bytes of every segment */
    WalkPlan* dPlan{ nullptr };       /* [MAX_GROUPS]: one per group */
    uint32_t* dWalkBlk{ nullptr };    /* [MAX_GROUPS][cap + 16] */
    uint32_t* dWalkPre{ nullptr };
    uint32_t* hSlotOf{ nullptr };     /* pinned: original index -> slot */
    uint32_t* dSlotOf{ nullptr };
    uint64_t* dTotals{ nullptr };     /* k_offsets: {total decoded bytes, does not fit} */
    uint32_t* dScanQueue{ nullptr };  /* [MAX_GROUPS]: block counters of k_hscan<1> launched with a capped grid */
    uint32_t* dBwtCounts{ nullptr };  /* [min( cap, BWT_SPLIT_BLOCKS )][BWT_COUNTS_PER_BLOCK]: byte counts per chunk, small batches */
    uint64_t* hTotals{ nullptr };     /* pinned */
    BlockMeta* hMeta{ nullptr };       /* pinned */
    uint64_t* hOffsets{ nullptr };     /* pinned */

    /* output: dOut holds the last finished batch.  A caller that copies it out in the background
     * (mi355x_bz2_copy_output_begin) gets the next batch written into a second buffer, so that the copy and the next
     * decode overlap; callers that never do keep a single buffer. */
    struct OutBuffer
    {
        uint8_t* bytes{ nullptr };
        uint64_t capacity{ 0 };
        hipEvent_t copied{ nullptr };     /* behind the last background copy out of this buffer */
        bool copyIssued{ false };
    };
    OutBuffer out[2];
    int outCurrent{ 0 };
    int outLastCopy{ 0 };
    hipStream_t copyStream{ nullptr };
    uint8_t* dOut{ nullptr };             /* == out[outCurrent].bytes */
    uint64_t outSizeHint{ 0 };            /* the largest batch output so far */
    hipEvent_t outputHold{ nullptr };     /* mi355x_bz2_hold_output_until: not owned */

    /* mi355x_bz2_find_magic_device: a stream and buffers of its own, so that a scan neither queues behind a batch nor
     * stops one (hipFree waits for the whole device) */
    std::mutex scanMutex;
    hipStream_t scanStream{ nullptr };
    uint64_t* dScanFound{ nullptr };
    uint32_t* dScanCounter{ nullptr };
    uint64_t outSize{ 0 };
    uint32_t lastBlocks{ 0 };

    CrcConsts crc{};
    hipEvent_t ev[MAX_GROUPS][2 * MI355X_BZ2_MAX_KERNELS]{};   /* [group][2 * kernel + {start, end}] */
    hipEvent_t evStep[3]{};                                     /* step start, inputs uploaded, step end */
    hipEvent_t evGroupDone[MAX_GROUPS]{};
    uint32_t launched[MAX_GROUPS]{};                            /* bit k: kernel k was launched for the group in this batch */
    /* a batch between mi355x_bz2_decode_batch_begin and _end */
    uint32_t pendingBlocks{ 0 };
    int pendingGroups{ 0 }, pendingExpensive{ -1 };
    uint32_t pendingGroupCount[MAX_GROUPS]{};
    int timingGroups{ 0 };            /* groups of the last batch */
    bool timingsResolved{ true };     /* timings.ms_kernel[] filled in for the last batch */
    uint32_t nKernels{ 0 };
    mi355x_bz2_timings timings{};
};

namespace
{
uint32_t
hostGfMul( uint32_t a, uint32_t b )
{
    uint32_t r = 0;
    for ( int i = 31; i >= 0; --i ) {
        r = ( r << 1 ) ^ ( ( r >> 31 ) ? 0x04C11DB7u : 0u );
        if ( ( a >> i ) & 1u ) r ^= b;
    }
    return r;
}

void
initCrcConsts( CrcConsts& cc )
{
    /* x^8, squared repeatedly */
    uint32_t p = 0x100u;
    for ( int k = 0; k < 32; ++k ) {
        cc.pow8[k] = p;
        p = hostGfMul( p, p );
    }
    /* x^-1 = x^(2^32 - 2) (P is primitive, the multiplicative group has order 2^32 - 1) */
    uint32_t xinv = 1u, base = 0x2u;   /* base = x */
    uint64_t e = 0xFFFFFFFEull;
    while ( e != 0 ) {
        if ( e & 1u ) xinv = hostGfMul( xinv, base );
        base = hostGfMul( base, base );
        e >>= 1;
    }
    uint32_t q = xinv;
    for ( int i = 0; i < 3; ++i ) q = hostGfMul( q, q );   /* x^-8 */
    for ( int k = 0; k < 32; ++k ) {
        cc.ipow8[k] = q;
        q = hostGfMul( q, q );
    }
}

#define HIP_TRY( ctx, expr )                                                                       \
    do {                                                                                           \
        const hipError_t err_ = ( expr );                                                          \
        if ( err_ != hipSuccess ) {                                                                \
            ( ctx )->lastError = std::string( #expr ) + ": " + hipGetErrorString( err_ );          \
            return MI355X_BZ2_ERR_DEVICE;                                                          \
        }                                                                                          \
    } while ( 0 )

void
freeRetired( mi355x_bz2_ctx* c )
{
    for ( const auto& r : c->retired ) {
        if ( r.host ) (void)hipHostFree( r.pointer ); else (void)hipFree( r.pointer );
    }
    c->retired.clear();
    c->retiredBytes = 0;
}

/** Puts a buffer that is no longer used aside (nothing on the device may still be reading it: the caller has
 * synchronised the streams that did). */
void
retire( mi355x_bz2_ctx* c, void* pointer, uint64_t bytes, bool host, uint64_t inUseNow )
{
    if ( pointer == nullptr ) return;
    c->retired.push_back( { pointer, bytes, host } );
    c->retiredBytes += bytes;
    if ( c->retiredBytes > std::max<uint64_t>( inUseNow, uint64_t( 1 ) << 30 ) ) freeRetired( c );
}

void
freeScratch( mi355x_bz2_ctx* c, bool now = true )
{
    if ( now ) {
        (void)hipFree( c->dScratch );
        (void)hipHostFree( c->hScratch );
    } else {
        retire( c, c->dScratch, c->scratchBytes, false, c->scratchBytes );
        retire( c, c->hScratch, c->scratchHostBytes, true, c->scratchBytes );
    }
    c->dScratch = nullptr;
    c->hScratch = nullptr;
    c->scratchBytes = c->scratchHostBytes = 0;
    c->dOffsets = nullptr; c->dOrder = nullptr; c->dMeta = nullptr; c->dSel = nullptr; c->dSym = nullptr; c->dStb = nullptr;
    c->dHmeta = nullptr; c->dSmeta = nullptr; c->dHtab = nullptr; c->dGpos = nullptr; c->dL = nullptr; c->dTab = nullptr;
    c->dR = nullptr; c->dSegLen = nullptr; c->dSegSucc = nullptr; c->dSegCont = nullptr;
    c->dChain = nullptr; c->dStash = nullptr; c->dPlan = nullptr; c->dWalkBlk = nullptr; c->dWalkPre = nullptr;
    c->hOrder = nullptr; c->hSlotOf = nullptr; c->hMeta = nullptr; c->hOffsets = nullptr;
    c->dSlotOf = nullptr; c->dTotals = nullptr; c->hTotals = nullptr; c->dScanQueue = nullptr; c->dBwtCounts = nullptr;
    c->capacity = 0;
}

/** Per-block scratch for `nBlocks` blocks: ONE device allocation and ONE page-locked host allocation, carved into the
 * buffers (two dozen separate allocations cost 80 ms per context, which a reader pays before its first byte). */
int
ensureScratch( mi355x_bz2_ctx* c, uint32_t nBlocks )
{
    if ( nBlocks <= c->capacity ) return MI355X_BZ2_OK;
    HIP_TRY( c, hipStreamSynchronize( c->stream ) );
    freeScratch( c, /* now */ false );
    /* about 10 MB of scratch per block: powers of two while that is cheap, multiples of 256 blocks beyond */
    uint32_t cap = 8;
    while ( cap < nBlocks && cap < 512 ) cap *= 2;
    if ( cap < nBlocks ) cap = ( nBlocks + 255u ) & ~255u;

    size_t deviceBytes = 0, hostBytes = 0;
    const auto reserve = [] ( size_t& total, size_t bytes ) {
        const size_t at = total;
        total += ( bytes + 255 ) & ~size_t( 255 );
        return at;
    };
    const size_t oOffsets = reserve( deviceBytes, (size_t)cap * sizeof( uint64_t ) );
    const size_t oOrder = reserve( deviceBytes, (size_t)cap * sizeof( uint32_t ) );
    const size_t oMeta = reserve( deviceBytes, (size_t)cap * sizeof( BlockMeta ) );
    const size_t oSel = reserve( deviceBytes, (size_t)cap * SEL_STRIDE + 256 );
    const size_t oStb = reserve( deviceBytes, (size_t)cap * 256 );
    const size_t oHmeta = reserve( deviceBytes, (size_t)cap * sizeof( HuffMeta ) );
    const size_t oSmeta = reserve( deviceBytes, (size_t)cap * sizeof( ScanMeta ) );
    const size_t oHtab = reserve( deviceBytes, (size_t)cap * sizeof( HuffTables ) );
    const size_t oGpos = reserve( deviceBytes, (size_t)cap * GPOS_STRIDE * sizeof( uint32_t ) );
    const size_t oL = reserve( deviceBytes, (size_t)cap * L_STRIDE + 256 );
    const size_t oTab = reserve( deviceBytes, (size_t)cap * TAB_STRIDE * sizeof( uint32_t ) );
    /* the bytes of the inverse BWT (k_emit -> k_rle) go where the block's last column was (k_mtf -> table build): same
     * slot, same stream, never alive together; 0.9 MB per block less.  Not when the caller wants to look at the stages. */
    const bool keepStages = ( c->flags & MI355X_BZ2_FLAG_KEEP_STAGES ) != 0;
    const size_t oR = keepStages ? reserve( deviceBytes, (size_t)cap * L_STRIDE + 256 ) : oL;
    const size_t oSegLen = reserve( deviceBytes, (size_t)cap * SEG_STRIDE * sizeof( uint32_t ) );
    const size_t oSegSucc = reserve( deviceBytes, (size_t)cap * SEG_STRIDE * sizeof( uint32_t ) );
    const size_t oSegCont = reserve( deviceBytes, (size_t)cap * SEG_STRIDE * sizeof( uint32_t ) );
    const size_t oChain = reserve( deviceBytes, (size_t)cap * SEG_STRIDE * sizeof( uint2 ) );
    const size_t oStash = reserve( deviceBytes, (size_t)cap * SEG_STRIDE * STASH_BYTES );
    const size_t oPlan = reserve( deviceBytes, MAX_GROUPS * sizeof( WalkPlan ) );
    const size_t oWalkBlk = reserve( deviceBytes, MAX_GROUPS * ( (size_t)cap + 16 ) * sizeof( uint32_t ) );
    const size_t oWalkPre = reserve( deviceBytes, MAX_GROUPS * ( (size_t)cap + 16 ) * sizeof( uint32_t ) );
    const size_t oSlotOf = reserve( deviceBytes, (size_t)cap * sizeof( uint32_t ) );
    const size_t oTotals = reserve( deviceBytes, 2 * sizeof( uint64_t ) );
    const size_t oScanQueue = reserve( deviceBytes, MAX_GROUPS * sizeof( uint32_t ) );
    const size_t oBwtCounts = reserve( deviceBytes, (size_t)std::min( cap, BWT_SPLIT_BLOCKS ) * BWT_COUNTS_PER_BLOCK * sizeof( uint32_t ) );
    const size_t hOrderAt = reserve( hostBytes, (size_t)cap * sizeof( uint32_t ) );
    const size_t hSlotOfAt = reserve( hostBytes, (size_t)cap * sizeof( uint32_t ) );
    const size_t hMetaAt = reserve( hostBytes, (size_t)cap * sizeof( BlockMeta ) );
    const size_t hOffsetsAt = reserve( hostBytes, (size_t)cap * sizeof( uint64_t ) );
    const size_t hTotalsAt = reserve( hostBytes, 2 * sizeof( uint64_t ) );

    const auto tAlloc = std::chrono::steady_clock::now();
    HIP_TRY( c, hipMalloc( &c->dScratch, deviceBytes ) );
    HIP_TRY( c, hipHostMalloc( &c->hScratch, hostBytes, hipHostMallocDefault ) );
    if ( std::getenv( "MI355X_BZ2_READER_TRACE" ) != nullptr ) {
        std::fprintf( stderr, "[device] scratch for %u blocks (%.0f MB): %.1f ms\n", cap, deviceBytes / 1e6,
                      std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - tAlloc ).count() );
    }
    uint8_t* const d = c->dScratch;
    uint8_t* const h = c->hScratch;
    c->dOffsets = reinterpret_cast<uint64_t*>( d + oOffsets );
    c->dOrder = reinterpret_cast<uint32_t*>( d + oOrder );
    c->dMeta = reinterpret_cast<BlockMeta*>( d + oMeta );
    c->dSel = d + oSel;
    c->dStb = d + oStb;
    c->dHmeta = reinterpret_cast<HuffMeta*>( d + oHmeta );
    c->dSmeta = reinterpret_cast<ScanMeta*>( d + oSmeta );
    c->dHtab = reinterpret_cast<HuffTables*>( d + oHtab );
    c->dGpos = reinterpret_cast<uint32_t*>( d + oGpos );
    c->dL = d + oL;
    c->dTab = reinterpret_cast<uint32_t*>( d + oTab );
    c->dR = d + oR;
    c->dSegLen = reinterpret_cast<uint32_t*>( d + oSegLen );
    c->dSegSucc = reinterpret_cast<uint32_t*>( d + oSegSucc );
    c->dSegCont = reinterpret_cast<uint32_t*>( d + oSegCont );
    c->dChain = reinterpret_cast<uint2*>( d + oChain );
    c->dStash = reinterpret_cast<uint32_t*>( d + oStash );
    /* the Huffman symbols of a block (k_hsym -> k_mtf) live where the block's stash will be (k_walk -> k_emit): same slot
     * size, never alive together, producers and consumers of a block slot on one stream in that order; 1.8 MB per block less */
    static_assert( (size_t)SYM_STRIDE * sizeof( uint16_t ) == (size_t)SEG_STRIDE * STASH_BYTES );
    c->dSym = reinterpret_cast<uint16_t*>( d + oStash );
    c->dPlan = reinterpret_cast<WalkPlan*>( d + oPlan );
    c->dWalkBlk = reinterpret_cast<uint32_t*>( d + oWalkBlk );
    c->dWalkPre = reinterpret_cast<uint32_t*>( d + oWalkPre );
    c->dSlotOf = reinterpret_cast<uint32_t*>( d + oSlotOf );
    c->dTotals = reinterpret_cast<uint64_t*>( d + oTotals );
    c->dScanQueue = reinterpret_cast<uint32_t*>( d + oScanQueue );
    c->dBwtCounts = reinterpret_cast<uint32_t*>( d + oBwtCounts );
    c->hOrder = reinterpret_cast<uint32_t*>( h + hOrderAt );
    c->hSlotOf = reinterpret_cast<uint32_t*>( h + hSlotOfAt );
    c->hMeta = reinterpret_cast<BlockMeta*>( h + hMetaAt );
    c->hOffsets = reinterpret_cast<uint64_t*>( h + hOffsetsAt );
    c->hTotals = reinterpret_cast<uint64_t*>( h + hTotalsAt );
    c->capacity = cap;
    c->scratchBytes = deviceBytes;
    c->scratchHostBytes = hostBytes;
    return MI355X_BZ2_OK;
}

/** The buffer the batch that is being finished writes its `size` bytes to: the other one if the last batch's bytes are
 * (or may still be) on their way to the host in the background, else the same again.  Kernels queued on c->stream behind
 * this call wait for the copy that last read from the chosen buffer. */
int
ensureOutput( mi355x_bz2_ctx* c, uint64_t size )
{
    const int target = c->out[c->outCurrent].copyIssued ? c->outCurrent ^ 1 : c->outCurrent;
    auto& buffer = c->out[target];
    if ( buffer.copyIssued ) {
        HIP_TRY( c, hipStreamWaitEvent( c->stream, buffer.copied, 0 ) );
    }
    if ( size + 256 > buffer.capacity ) {
        HIP_TRY( c, hipStreamSynchronize( c->stream ) );
        if ( buffer.copyIssued ) HIP_TRY( c, hipEventSynchronize( buffer.copied ) );
        retire( c, buffer.bytes, buffer.capacity, false, size );
        buffer.bytes = nullptr;
        buffer.capacity = 0;
        const uint64_t cap = size + size / 8 + ( 1u << 20 );
        HIP_TRY( c, hipMalloc( &buffer.bytes, cap ) );
        buffer.capacity = cap;
    }
    buffer.copyIssued = false;
    c->outCurrent = target;
    c->dOut = buffer.bytes;
    return MI355X_BZ2_OK;
}
}  // namespace

namespace
{
const char* const KERNEL_NAMES[] = {
    "(k_huff: gone)", "k_mtf<272>", "k_bwt_build", "k_walk", "k_link2", "k_emit", "k_replicate", "k_rle<false>",
    "k_rle<true>", "k_crc", "k_walk_plan", "k_mtf<144>", "k_hscan", "k_hsym"
};
constexpr uint32_t N_KERNELS = sizeof( KERNEL_NAMES ) / sizeof( KERNEL_NAMES[0] );
static_assert( N_KERNELS <= MI355X_BZ2_MAX_KERNELS );
}  // namespace

/* record an event pair around one launch so that every kernel gets its own device duration */
/* ONE k_walk at a time per process, whatever stream and context it comes from: every walk is ordered behind the one
 * launched before it (MI355X_BZ2_WALK_SERIAL=0 turns this off).  A walk keeps one or two 3.6 MB tables per XCD in its
 * 4 MB L2; walks of several block groups and contexts side by side push each other's tables out.  In turn, and with
 * fewer workgroups each (64 per XCD instead of 256: the other kernels of the crowd fill the wave slots while the walk
 * waits for its gathers), a step of the four-context bench takes 67.5 instead of 74 ms. */
struct WalkChain
{
    std::mutex mutex;
    hipEvent_t events[64]{};
    uint32_t next{ 0 };
    hipEvent_t last{ nullptr };
    std::atomic<int> liveContexts{ 0 };
};
WalkChain g_walkChains[16];      /* by device: walks on different GPUs have nothing to do with each other */
WalkChain& walkChainOf( int device ) { return g_walkChains[(unsigned)device % 16u]; }

#define TIMED_LAUNCH( ctx, group, queue, index, ... )                                      \
    do {                                                                                   \
        ( ctx )->launched[group] |= 1u << ( index );                                       \
        HIP_TRY( ctx, hipEventRecord( ( ctx )->ev[group][2 * ( index )], queue ) );        \
        hipLaunchKernelGGL( __VA_ARGS__ );                                                 \
        HIP_TRY( ctx, hipEventRecord( ( ctx )->ev[group][2 * ( index ) + 1], queue ) );    \
    } while ( 0 )

namespace
{
/** Kernels whose LDS is declared at launch (see k_hscan) and exceeds the 64 KB a launch may ask for by default. */
bool
allowLargeLds()
{
    bool ok = true;
    const auto allow = [&ok] ( const void* kernel, size_t bytes ) {
        ok = ok && hipFuncSetAttribute( kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes ) == hipSuccess;
    };
    allow( reinterpret_cast<const void*>( &k_mtf<MTF_LANE_STRIDE, MTF_THREADS, REGS_MTF> ), sizeof( MtfShared<MTF_LANE_STRIDE, MTF_THREADS> ) );
    allow( reinterpret_cast<const void*>( &k_mtf<MTF_LANE_STRIDE, 512> ), sizeof( MtfShared<MTF_LANE_STRIDE, 512> ) );
    allow( reinterpret_cast<const void*>( &k_mtf<MTF_SMALL_STRIDE, 512> ), sizeof( MtfShared<MTF_SMALL_STRIDE, 512> ) );
    allow( reinterpret_cast<const void*>( &k_mtf<MTF_SMALL_STRIDE, 1024> ), sizeof( MtfShared<MTF_SMALL_STRIDE, 1024> ) );
    allow( reinterpret_cast<const void*>( &k_link2 ), sizeof( LinkShared ) );
    return ok;
}
}  // namespace

extern "C" {

const char*
mi355x_bz2_kernel_name( uint32_t index )
{
    return index < N_KERNELS ? KERNEL_NAMES[index] : "";
}

const char*
mi355x_bz2_status_string( int status )
{
    switch ( status ) {
    case MI355X_BZ2_OK: return "OK";
    case MI355X_BZ2_ERR_EOF: return "end of file reached inside a block";
    case MI355X_BZ2_ERR_BAD_MAGIC: return "[BZip2 block header] invalid compressed magic";
    case MI355X_BZ2_ERR_RANDOMIZED: return "[BZip2 block header] deprecated isRandomized bit is not supported";
    case MI355X_BZ2_ERR_ORIGPTR_RANGE: return "[BZip2 block header] origPtr is larger than buffer size";
    case MI355X_BZ2_ERR_GROUP_COUNT: return "[BZip2 block header] Invalid Huffman coding group count";
    case MI355X_BZ2_ERR_SELECTOR_COUNT: return "[BZip2 block header] The number of selectors is invalid";
    case MI355X_BZ2_ERR_SELECTOR_UNARY: return "[BZip2 block header] Could not find zero termination";
    case MI355X_BZ2_ERR_CODE_LENGTH: return "[BZip2 block header] start_huffman_length is larger than 20 or zero";
    case MI355X_BZ2_ERR_HUFFMAN_LENGTHS: return "Invalid Huffman code lengths";
    case MI355X_BZ2_ERR_SELECTOR_OVERRUN: return "[BZip2 block data] selector out of maximum range";
    case MI355X_BZ2_ERR_INVALID_CODE: return "[BZip2 block data] no Huffman code matches (bad optional access)";
    case MI355X_BZ2_ERR_RUN_OVERFLOW: return "[BZip2 block data] dbufCount + hh > dbufSize";
    case MI355X_BZ2_ERR_DATA_OVERFLOW: return "[BZip2 block data] dbufCount > dbufSize";
    case MI355X_BZ2_ERR_ORIGPTR_DATA: return "[BZip2 block data] origPtr error";
    case MI355X_BZ2_ERR_CRC: return "Calculated CRC for block mismatches";
    case MI355X_BZ2_ERR_STREAM_CRC: return "Stream CRC does not match calculated CRC";
    case MI355X_BZ2_ERR_NO_BLOCK_IN_RANGE: return "Failed to find any valid bzip2 block in the given range";
    case MI355X_BZ2_ERR_STREAM_HEADER: return "Input header is not BZip2 magic string 'BZh' or invalid block size";
    case MI355X_BZ2_ERR_OUTPUT_CAPACITY: return "output capacity exceeded";
    case MI355X_BZ2_ERR_DEVICE: return "HIP runtime error";
    case MI355X_BZ2_ERR_NO_DEVICE: return "no usable MI355X (gfx950) device; there is no CPU fallback";
    case MI355X_BZ2_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MI355X_BZ2_ERR_IO: return "I/O error";
    case MI355X_BZ2_ERR_CLOSED: return "operation on closed reader";
    case MI355X_BZ2_ERR_LOGIC: return "internal logic error";
    default: return "unknown status";
    }
}

int
mi355x_bz2_abi_version( void )
{
    return MI355X_BZ2_ABI_VERSION;
}

int
mi355x_bz2_warmup( int32_t device )
{
    int count = 0;
    if ( hipGetDeviceCount( &count ) != hipSuccess || count <= 0 ) return MI355X_BZ2_ERR_NO_DEVICE;
    if ( device < 0 || device >= count ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    if ( hipSetDevice( device ) != hipSuccess ) return MI355X_BZ2_ERR_DEVICE;
    /* the runtime's queues and the code object of this library: an empty magic scan is the cheapest real launch */
    uint32_t* counter = nullptr;
    if ( hipMalloc( &counter, 256 ) != hipSuccess ) return MI355X_BZ2_ERR_DEVICE;
    hipLaunchKernelGGL( k_find_magic, dim3( 1 ), dim3( 256 ), 0, nullptr, reinterpret_cast<const uint32_t*>( counter ),
                        uint64_t( 0 ), MI355X_BZ2_MAGIC_BLOCK, reinterpret_cast<uint64_t*>( counter ), 0u, counter );
    const bool ok = hipDeviceSynchronize() == hipSuccess;
    (void)hipFree( counter );
    return ok ? MI355X_BZ2_OK : MI355X_BZ2_ERR_DEVICE;
}

int
mi355x_bz2_create( const mi355x_bz2_config* config, mi355x_bz2_ctx** out )
{
    if ( out == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if ( hipGetDeviceCount( &count ) != hipSuccess || count <= 0 ) {
        return MI355X_BZ2_ERR_NO_DEVICE;
    }
    int device = config != nullptr ? config->device : -1;
    if ( device < 0 ) {
        if ( hipGetDevice( &device ) != hipSuccess ) return MI355X_BZ2_ERR_NO_DEVICE;
    }
    if ( device >= count ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    hipDeviceProp_t prop{};
    if ( hipGetDeviceProperties( &prop, device ) != hipSuccess ) return MI355X_BZ2_ERR_NO_DEVICE;
    if ( std::strncmp( prop.gcnArchName, "gfx950", 6 ) != 0 ) {
        return MI355X_BZ2_ERR_NO_DEVICE;   /* kernels are built for gfx950 only */
    }
    if ( hipSetDevice( device ) != hipSuccess ) return MI355X_BZ2_ERR_DEVICE;
    {
        /* once per device (the attribute belongs to the kernel's image on a device) */
        static std::mutex once;
        static bool done[16] = {};
        const std::scoped_lock lock( once );
        if ( !done[(unsigned)device % 16u] ) {
            if ( !allowLargeLds() ) return MI355X_BZ2_ERR_DEVICE;
            done[(unsigned)device % 16u] = true;
        }
    }
    auto* c = new mi355x_bz2_ctx();
    c->device = device;
    walkChainOf( device ).liveContexts.fetch_add( 1 );
    c->flags = config != nullptr ? config->flags : 0;
    if ( hipSetDevice( device ) != hipSuccess
         || hipStreamCreateWithFlags( &c->stream, hipStreamNonBlocking ) != hipSuccess ) {
        mi355x_bz2_destroy( c );   /* releases whatever exists so far */
        return MI355X_BZ2_ERR_DEVICE;
    }
    c->gstream[0] = c->stream;
    {
        int leastPriority = 0, greatestPriority = 0;   /* numerically lower = higher priority */
        (void)hipDeviceGetStreamPriorityRange( &leastPriority, &greatestPriority );
        for ( int g = 1; g < MAX_GROUPS; ++g ) {
            /* gstream[MAX_GROUPS - 1] serves the expensive group: its scan is the longest chain of a batch */
            const int priority = g == MAX_GROUPS - 1 ? greatestPriority : ( leastPriority + greatestPriority ) / 2;
            if ( hipStreamCreateWithPriority( &c->gstream[g], hipStreamNonBlocking, priority ) != hipSuccess ) {
                mi355x_bz2_destroy( c );
                return MI355X_BZ2_ERR_DEVICE;
            }
        }
    }
    for ( auto& e : c->evGroupDone ) {
        if ( hipEventCreate( &e ) != hipSuccess ) {
            mi355x_bz2_destroy( c );
            return MI355X_BZ2_ERR_DEVICE;
        }
    }
    for ( auto& group : c->ev ) {
        for ( auto& e : group ) {
            if ( hipEventCreate( &e ) != hipSuccess ) {
                mi355x_bz2_destroy( c );
                return MI355X_BZ2_ERR_DEVICE;
            }
        }
    }
    for ( auto& e : c->evStep ) {
        if ( hipEventCreate( &e ) != hipSuccess ) {
            mi355x_bz2_destroy( c );
            return MI355X_BZ2_ERR_DEVICE;
        }
    }
    initCrcConsts( c->crc );
    const uint32_t initial = ( config != nullptr && config->max_batch_blocks > 0 ) ? config->max_batch_blocks : 64;
    const int rc = ensureScratch( c, initial );
    if ( rc != MI355X_BZ2_OK ) {
        std::fprintf( stderr, "mi355x_bz2_create: %s\n", c->lastError.c_str() );
        mi355x_bz2_destroy( c );
        return rc;
    }
    *out = c;
    return MI355X_BZ2_OK;
}

void
mi355x_bz2_destroy( mi355x_bz2_ctx* c )
{
    if ( c == nullptr ) return;
    walkChainOf( c->device ).liveContexts.fetch_sub( 1 );
    (void)hipSetDevice( c->device );
    if ( c->stream ) (void)hipStreamSynchronize( c->stream );
    for ( int g = 1; g < MAX_GROUPS; ++g ) {
        if ( c->gstream[g] ) (void)hipStreamSynchronize( c->gstream[g] );
    }
    freeScratch( c );
    freeRetired( c );
    c->upload.reset();   /* joins the copy thread (of the owner; sharers only drop their reference) before the memory goes */
    if ( c->inStream ) (void)hipStreamSynchronize( c->inStream );
    for ( auto& buffer : c->in ) (void)hipFree( buffer.bytes );
    if ( c->inReady ) (void)hipEventDestroy( c->inReady );
    if ( c->inStream ) (void)hipStreamDestroy( c->inStream );
    if ( c->scanStream ) {
        (void)hipStreamSynchronize( c->scanStream );
        (void)hipStreamDestroy( c->scanStream );
    }
    (void)hipFree( c->dScanFound );
    (void)hipFree( c->dScanCounter );
    if ( c->copyStream ) (void)hipStreamSynchronize( c->copyStream );
    for ( auto& buffer : c->out ) {
        (void)hipFree( buffer.bytes );
        if ( buffer.copied ) (void)hipEventDestroy( buffer.copied );
    }
    if ( c->copyStream ) (void)hipStreamDestroy( c->copyStream );
    for ( auto& group : c->ev ) {
        for ( auto& e : group ) {
            if ( e ) (void)hipEventDestroy( e );
        }
    }
    for ( auto& e : c->evStep ) {
        if ( e ) (void)hipEventDestroy( e );
    }
    if ( c->stream ) (void)hipStreamDestroy( c->stream );
    for ( int g = 1; g < MAX_GROUPS; ++g ) {
        if ( c->gstream[g] ) (void)hipStreamDestroy( c->gstream[g] );
    }
    for ( int g = 0; g < MAX_GROUPS; ++g ) {
        if ( c->sideStream[g] ) {
            (void)hipStreamSynchronize( c->sideStream[g] );
            (void)hipStreamDestroy( c->sideStream[g] );
        }
        if ( c->evFork[g] ) (void)hipEventDestroy( c->evFork[g] );
        if ( c->evJoin[g] ) (void)hipEventDestroy( c->evJoin[g] );
    }
    for ( auto& e : c->evGroupDone ) {
        if ( e ) (void)hipEventDestroy( e );
    }
    delete c;
}

const char*
mi355x_bz2_last_error( const mi355x_bz2_ctx* c )
{
    return c != nullptr ? c->lastError.c_str() : "null context";
}

namespace
{
/** Room for `size` input bytes + padding in the ctx-owned copy. */
int
reserveInput( mi355x_bz2_ctx* c, int which, uint64_t size )
{
    auto& buffer = c->in[which];
    const uint64_t padded = ( ( size + 255 ) & ~uint64_t( 255 ) ) + 256;
    if ( padded > buffer.capacity ) {
        /* nothing reads this buffer: the batch in flight, if any, reads the other one */
        if ( c->inStream ) HIP_TRY( c, hipStreamSynchronize( c->inStream ) );
        retire( c, buffer.bytes, buffer.capacity, false, padded );
        buffer.bytes = nullptr;
        buffer.capacity = 0;
        HIP_TRY( c, hipMalloc( &buffer.bytes, padded ) );
        buffer.capacity = padded;
    }
    return MI355X_BZ2_OK;
}

/** Orders the ctx stream behind the streamed copy of input bytes [0, needed) and says where they are (the head buffer or
 * the full copy, see InputUpload); waits on the host until that copy is queued.  Without a streamed copy: the resident
 * input as it is. */
int
awaitInput( mi355x_bz2_ctx* c, uint64_t needed, const uint8_t** base, uint64_t* size )
{
    *base = c->dIn;
    *size = c->inSize;
    const auto upload = c->upload;
    if ( !upload || upload->total == 0 ) return MI355X_BZ2_OK;
    needed = std::min( std::max<uint64_t>( needed, 1 ), upload->total );
    hipEvent_t event = nullptr;
    {
        std::unique_lock lock( upload->mutex );
        if ( ( upload->headBytes != 0 ) && ( needed <= upload->headBytes ) && ( upload->queued < needed ) ) {
            upload->changed.wait( lock, [&] { return upload->failed || upload->headQueued; } );
            *base = upload->head;
            *size = upload->headBytes;
            event = upload->headDone;
        } else {
            upload->changed.wait( lock, [&] { return upload->failed || upload->queued >= needed; } );
            *base = upload->main;
            *size = upload->total;
            event = upload->done[( needed - 1 ) / InputUpload::PIECE];
        }
        if ( upload->failed ) {
            c->lastError = "the streamed copy of the input failed";
            return MI355X_BZ2_ERR_DEVICE;
        }
    }
    HIP_TRY( c, hipStreamWaitEvent( c->stream, event, 0 ) );
    return MI355X_BZ2_OK;
}

/** Queue `size` bytes (host or device source) into a ctx-owned input copy, zero padded, on the input stream; no wait.
 * The next decode_batch_begin is ordered behind the copy.  While a batch is in flight the copy goes into the buffer
 * that batch does not read.  Copies of more than 64 MiB go in pieces, so that what other streams move is not queued
 * behind one long transfer. */
int
queueInput( mi355x_bz2_ctx* c, const void* bytes, uint64_t size, hipMemcpyKind kind )
{
    HIP_TRY( c, hipSetDevice( c->device ) );
    c->upload.reset();
    if ( c->inStream == nullptr ) HIP_TRY( c, hipStreamCreateWithFlags( &c->inStream, hipStreamNonBlocking ) );
    if ( c->inReady == nullptr ) HIP_TRY( c, hipEventCreateWithFlags( &c->inReady, hipEventDisableTiming ) );
    const int which = ( c->pendingBlocks != 0 && c->inFlight >= 0 ) ? c->inFlight ^ 1 : c->inCurrent;
    const int rc = reserveInput( c, which, size );
    if ( rc != MI355X_BZ2_OK ) return rc;
    uint8_t* const target = c->in[which].bytes;
    HIP_TRY( c, hipMemsetAsync( target + ( size & ~uint64_t( 255 ) ), 0,
                                ( ( ( size + 255 ) & ~uint64_t( 255 ) ) + 256 ) - ( size & ~uint64_t( 255 ) ), c->inStream ) );
    constexpr uint64_t PIECE = uint64_t( 64 ) << 20;
    for ( uint64_t at = 0; at < size; at += PIECE ) {
        HIP_TRY( c, hipMemcpyAsync( target + at, static_cast<const uint8_t*>( bytes ) + at, std::min( PIECE, size - at ),
                                    kind, c->inStream ) );
    }
    HIP_TRY( c, hipEventRecord( c->inReady, c->inStream ) );
    c->inPending = true;
    c->inCurrent = which;
    c->dIn = target;
    c->inSize = size;
    return MI355X_BZ2_OK;
}
}  // namespace

int
mi355x_bz2_set_input_host( mi355x_bz2_ctx* c, const uint8_t* bytes, uint64_t size )
{
    if ( c == nullptr || ( bytes == nullptr && size > 0 ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( c->pendingBlocks != 0 ) {
        c->lastError = "set_input_host: a batch is in flight";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    const int rc = queueInput( c, bytes, size, hipMemcpyHostToDevice );
    if ( rc != MI355X_BZ2_OK ) return rc;
    HIP_TRY( c, hipStreamSynchronize( c->inStream ) );
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_set_input_host_async( mi355x_bz2_ctx* c, const uint8_t* bytes, uint64_t size )
{
    if ( c == nullptr || ( bytes == nullptr && size > 0 ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    /* a batch may be in flight: these are the bytes of the next one */
    return queueInput( c, bytes, size, hipMemcpyHostToDevice );
}

int
mi355x_bz2_set_input_host_streamed( mi355x_bz2_ctx* c, const uint8_t* bytes, uint64_t size )
{
    if ( c == nullptr || ( bytes == nullptr && size > 0 ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( c->pendingBlocks != 0 ) {
        c->lastError = "set_input_host_streamed: a batch is in flight";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    HIP_TRY( c, hipSetDevice( c->device ) );
    c->upload.reset();
    if ( c->inStream ) HIP_TRY( c, hipStreamSynchronize( c->inStream ) );
    for ( auto& buffer : c->in ) {      /* the streamed copy owns its buffers */
        (void)hipFree( buffer.bytes );
        buffer = {};
    }
    c->inPending = false;
    {
        /* the whole file becomes resident (bounded residency is not implemented): say so if it cannot */
        size_t freeBytes = 0, totalBytes = 0;
        if ( hipMemGetInfo( &freeBytes, &totalBytes ) == hipSuccess && (uint64_t)freeBytes < size + ( uint64_t( 1 ) << 30 ) ) {
            c->lastError = "the compressed input (" + std::to_string( size >> 20 ) + " MiB) does not fit the free device memory ("
                           + std::to_string( freeBytes >> 20 ) + " MiB): the reader keeps the whole file resident";
            return MI355X_BZ2_ERR_DEVICE;
        }
    }
    auto upload = std::make_shared<InputUpload>();
    upload->total = size;
    upload->device = c->device;
    HIP_TRY( c, hipStreamCreateWithFlags( &upload->stream, hipStreamNonBlocking ) );
    upload->done.resize( (size_t)( ( size + InputUpload::PIECE - 1 ) / InputUpload::PIECE ), nullptr );
    for ( auto& e : upload->done ) {
        HIP_TRY( c, hipEventCreateWithFlags( &e, hipEventDisableTiming ) );
    }
    HIP_TRY( c, hipEventCreateWithFlags( &upload->headDone, hipEventDisableTiming ) );
    const auto paddedSize = [] ( uint64_t n ) { return ( ( n + 255 ) & ~uint64_t( 255 ) ) + 256; };
    if ( size > InputUpload::HEAD ) {
        upload->headBytes = InputUpload::HEAD;
        HIP_TRY( c, hipMalloc( &upload->head, paddedSize( upload->headBytes ) ) );
    }
    InputUpload* const u = upload.get();
    upload->worker = std::thread( [u, bytes, size, paddedSize] () {
        bool ok = hipSetDevice( u->device ) == hipSuccess;
        if ( ok && u->headBytes != 0 ) {
            ok = hipMemsetAsync( u->head + u->headBytes, 0, paddedSize( u->headBytes ) - u->headBytes, u->stream ) == hipSuccess
                 && hipMemcpyAsync( u->head, bytes, u->headBytes, hipMemcpyHostToDevice, u->stream ) == hipSuccess
                 && hipEventRecord( u->headDone, u->stream ) == hipSuccess;
            const std::scoped_lock guard( u->mutex );
            if ( ok ) u->headQueued = true; else u->failed = true;
            u->changed.notify_all();
        }
        if ( ok ) {
            uint8_t* buffer = nullptr;
            ok = hipMalloc( &buffer, paddedSize( size ) ) == hipSuccess
                 && hipMemsetAsync( buffer + ( size & ~uint64_t( 255 ) ), 0, paddedSize( size ) - ( size & ~uint64_t( 255 ) ), u->stream ) == hipSuccess;
            const std::scoped_lock guard( u->mutex );
            u->main = buffer;
        }
        const bool traceUpload = std::getenv( "MI355X_BZ2_READER_TRACE" ) != nullptr;
        const auto tUpload = std::chrono::steady_clock::now();
        for ( uint64_t at = 0, k = 0; ok && at < size; at += InputUpload::PIECE, ++k ) {
            const uint64_t n = std::min( InputUpload::PIECE, size - at );
            if ( traceUpload && k % 16 == 0 ) {
                std::fprintf( stderr, "[device] upload at %.0f MB: %.1f ms\n", at / 1e6,
                              std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - tUpload ).count() );
            }
            /* from pageable memory this call returns when the piece has been staged, i.e. the thread paces the copy */
            ok = hipMemcpyAsync( u->main + at, bytes + at, n, hipMemcpyHostToDevice, u->stream ) == hipSuccess
                 && hipEventRecord( u->done[k], u->stream ) == hipSuccess;
            const std::scoped_lock guard( u->mutex );
            if ( ok ) u->queued = at + n; else u->failed = true;
            u->changed.notify_all();
        }
        if ( !ok ) {
            const std::scoped_lock guard( u->mutex );
            u->failed = true;
            u->changed.notify_all();
        }
    } );
    c->upload = std::move( upload );
    c->dIn = nullptr;     /* set by the first call that needs the whole input (awaitInput gives the buffer to use) */
    c->inSize = size;
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_input_resident( const mi355x_bz2_ctx* c )
{
    if ( c == nullptr ) return 0;
    const auto upload = c->upload;
    if ( !upload ) return c->dIn != nullptr ? 1 : 0;
    const std::scoped_lock lock( upload->mutex );
    return ( upload->failed || upload->queued >= upload->total ) ? 1 : 0;
}

int
mi355x_bz2_set_input_device( mi355x_bz2_ctx* c, const void* deviceBytes, uint64_t size )
{
    if ( c == nullptr || ( deviceBytes == nullptr && size > 0 ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    /* The kernels read whole 16-byte windows without bounds checks, so the bytes are copied (device to device, once)
     * into ctx-owned memory that is zero padded past the end. */
    if ( c->pendingBlocks != 0 ) {
        c->lastError = "set_input_device: a batch is in flight";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    const int rc = queueInput( c, deviceBytes, size, hipMemcpyDeviceToDevice );
    if ( rc != MI355X_BZ2_OK ) return rc;
    HIP_TRY( c, hipStreamSynchronize( c->inStream ) );
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_share_input( mi355x_bz2_ctx* c, mi355x_bz2_ctx* from )
{
    if ( c == nullptr || from == nullptr || c == from ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex, from->mutex );
    if ( ( from->dIn == nullptr && !from->upload ) || c->device != from->device ) {
        c->lastError = "share_input: the other context has no input or lives on another device";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    if ( c->pendingBlocks != 0 ) {
        c->lastError = "share_input: a batch is in flight";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    if ( from->inReady != nullptr ) {
        /* a copy that `from` has queued (set_input_host_async): this context's batches come behind it too */
        HIP_TRY( c, hipSetDevice( c->device ) );
        HIP_TRY( c, hipStreamWaitEvent( c->stream, from->inReady, 0 ) );
    }
    c->dIn = from->dIn;      /* not owned: mi355x_bz2_destroy frees this context's own copies only */
    c->inSize = from->inSize;
    c->upload = from->upload;
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_decode_batch( mi355x_bz2_ctx* c, const uint64_t* offsets, uint32_t n,
                         mi355x_bz2_block_result* results, uint64_t* totalDecoded )
{
    if ( c == nullptr || ( n > 0 && ( offsets == nullptr || results == nullptr ) ) ) {
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    const int rc = mi355x_bz2_decode_batch_begin( c, offsets, n );
    if ( rc != MI355X_BZ2_OK ) return rc;
    return mi355x_bz2_decode_batch_end( c, results, totalDecoded );
}

int
mi355x_bz2_decode_batch_begin( mi355x_bz2_ctx* c, const uint64_t* offsets, uint32_t n )
{
    if ( c == nullptr || ( n > 0 && offsets == nullptr ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( c->pendingBlocks != 0 ) {
        c->lastError = "a batch is already in flight on this context: call mi355x_bz2_decode_batch_end first";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    c->outSize = 0;
    c->lastBlocks = 0;
    if ( n == 0 ) return MI355X_BZ2_OK;
    if ( n > MI355X_BZ2_MAX_BATCH_BLOCKS ) {
        c->lastError = "more than MI355X_BZ2_MAX_BATCH_BLOCKS blocks in one batch: split it";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    if ( c->dIn == nullptr && !c->upload ) {
        c->lastError = "no input set";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    HIP_TRY( c, hipSetDevice( c->device ) );
    const bool traceBegin = std::getenv( "MI355X_BZ2_READER_TRACE" ) != nullptr;
    const auto tBegin = std::chrono::steady_clock::now();
    int rc = ensureScratch( c, n );
    if ( rc != MI355X_BZ2_OK ) return rc;
    /* the output buffer of this batch (see the end of this function), chosen and, if need be, allocated before anything
     * is queued: growing it waits for the stream */
    rc = ensureOutput( c, std::max<uint64_t>( (uint64_t)n * 900000u, c->outSizeHint ) );
    if ( rc != MI355X_BZ2_OK ) return rc;
    const auto tScratch = std::chrono::steady_clock::now();
    /* with a streamed copy of the input this batch needs it up to where its last block can end (a block of 900 000
     * symbols is at most 900 000 x 20 bits, in practice < 1.2 MB; the scan kernels read up to 256 B further) */
    const uint8_t* inBase = nullptr;
    uint64_t inSize = 0;
    {
        uint64_t last = 0;
        for ( uint32_t i = 0; i < n; ++i ) last = std::max( last, offsets[i] );
        rc = awaitInput( c, last / 8 + 2400000, &inBase, &inSize );
        if ( rc != MI355X_BZ2_OK ) return rc;
    }
    const auto tInput = std::chrono::steady_clock::now();

    /* ---- plan: cost estimate, groups, slots, work order --------------------------------------------------------
     * cost = estimated compressed size (distance to the next requested offset, or to the end of the input).
     * The group-start scan (k_hscan<1>) is one serial chain per block: a launch lasts as long as its LARGEST block.
     * Everything behind it (symbols, MTF, BWT, walk, RLE, CRC) is throughput work of about the same size for every block.
     * The batch is therefore cut into groups, each with its own HIP stream, such that the throughput work starts early and
     * never runs dry:
     *   - the "expensive" group: blocks above 45 % of the largest cost, if they are a minority (incompressible blocks
     *     among text).  Its scan starts at once and runs beside everything else on a high-priority stream.
     *   - the other blocks, sorted by cost, in up to MAX_CHUNKS chunks of growing size.  All scans start together; a chunk
     *     of cheap blocks is through early, and its MTF .. RLE kernels run while the later chunks are still being scanned.
     * Inside a group the stage-1 kernels start their largest blocks first (LPT).
     * Slots: group g occupies slots [groupFirst[g], +groupCount[g]) of every per-block buffer; results are mapped
     * back to input order. */
    std::vector<uint64_t> cost( n );
    {
        std::vector<uint32_t> byOffset( n );
        for ( uint32_t i = 0; i < n; ++i ) byOffset[i] = i;
        std::sort( byOffset.begin(), byOffset.end(), [&] ( uint32_t a, uint32_t b ) { return offsets[a] < offsets[b]; } );
        for ( uint32_t k = 0; k < n; ++k ) {
            const uint64_t next = k + 1 < n ? offsets[byOffset[k + 1]] : inSize * 8;
            const uint64_t cur = offsets[byOffset[k]];
            cost[byOffset[k]] = next > cur ? next - cur : 0;
        }
    }
    uint64_t maxCost = 0;
    for ( const auto v : cost ) maxCost = std::max( maxCost, v );
    const char* noSplit = std::getenv( "MI355X_BZ2_NO_SPLIT" );
    const bool split = n >= 64 && !( noSplit != nullptr && noSplit[0] == '1' );

    std::vector<uint32_t> ascending( n );   /* block indices by increasing cost */
    for ( uint32_t i = 0; i < n; ++i ) ascending[i] = i;
    std::stable_sort( ascending.begin(), ascending.end(), [&] ( uint32_t a, uint32_t b ) { return cost[a] < cost[b]; } );

    uint32_t nExpensive = 0;
    if ( split ) {
        while ( nExpensive < n && cost[ascending[n - 1 - nExpensive]] * 100 > maxCost * 45 ) ++nExpensive;
        if ( nExpensive < 16 || (uint64_t)nExpensive * 100 > (uint64_t)n * 35 ) nExpensive = 0;
    }
    const uint32_t nCheap = n - nExpensive;
    uint32_t nChunks = 1;
    if ( split ) {
        /* measured on MI355X (round 1, and still the right proportion): a lone stage-1 wave takes about 5.5 ns per
         * compressed bit; the kernels behind it together about 0.04 ms per block when the GPU is full */
        const double huffMs = (double)cost[ascending[nCheap - 1]] * 5.5e-6;
        const double restMs = (double)nCheap * 0.04;
        const double ratio = restMs / std::max( huffMs, 1e-3 );
        nChunks = (uint32_t)std::min<double>( { ratio, (double)MAX_CHUNKS, (double)( nCheap / 128 ) } );
        nChunks = std::max( nChunks, 1u );
        if ( const char* forced = std::getenv( "MI355X_BZ2_CHUNKS" ); forced != nullptr && std::atoi( forced ) > 0 ) {
            nChunks = std::min<uint32_t>( (uint32_t)std::atoi( forced ), MAX_CHUNKS );
        }
    }
    const int nGroups = (int)nChunks + ( nExpensive > 0 ? 1 : 0 );
    const int expensiveGroup = nExpensive > 0 ? (int)nChunks : -1;
    uint32_t groupCount[MAX_GROUPS] = {};
    uint32_t groupFirst[MAX_GROUPS] = {};
    {
        /* chunk g ends at rank nCheap * (g + 1)(g + 2) / (K (K + 1)): 1/3, 1 for two chunks; 1/6, 1/2, 1 for three --
         * a small first chunk gets the throughput kernels going early, the later ones keep them fed */
        uint32_t begin = 0;
        for ( uint32_t g = 0; g < nChunks; ++g ) {
            const uint32_t end = (uint32_t)( (uint64_t)nCheap * ( g + 1 ) * ( g + 2 ) / ( (uint64_t)nChunks * ( nChunks + 1 ) ) );
            groupCount[g] = end - begin;
            begin = end;
        }
    }
    if ( expensiveGroup >= 0 ) groupCount[expensiveGroup] = nExpensive;
    for ( int g = 1; g < nGroups; ++g ) groupFirst[g] = groupFirst[g - 1] + groupCount[g - 1];
    /* slot = rank by cost: group g = ranks [groupFirst[g], +groupCount[g]); LPT order inside the group = descending */
    for ( uint32_t rank = 0; rank < n; ++rank ) {
        c->hSlotOf[ascending[rank]] = rank;
        c->hOffsets[rank] = offsets[ascending[rank]];
    }
    for ( int g = 0; g < nGroups; ++g ) {
        uint32_t* order = c->hOrder + groupFirst[g];
        for ( uint32_t k = 0; k < groupCount[g]; ++k ) order[k] = groupCount[g] - 1 - k;   /* group-relative slot */
    }
    for ( auto& bits : c->launched ) bits = 0;
    if ( c->inPending ) {
        HIP_TRY( c, hipStreamWaitEvent( c->stream, c->inReady, 0 ) );
        c->inPending = false;
    }
    c->inFlight = ( inBase == c->in[0].bytes && inBase != nullptr ) ? 0 : ( ( inBase == c->in[1].bytes && inBase != nullptr ) ? 1 : -1 );
    HIP_TRY( c, hipEventRecord( c->evStep[0], c->stream ) );
    HIP_TRY( c, hipMemcpyAsync( c->dOffsets, c->hOffsets, (size_t)n * sizeof( uint64_t ), hipMemcpyHostToDevice, c->stream ) );
    HIP_TRY( c, hipMemcpyAsync( c->dOrder, c->hOrder, (size_t)n * sizeof( uint32_t ), hipMemcpyHostToDevice, c->stream ) );
    HIP_TRY( c, hipMemcpyAsync( c->dSlotOf, c->hSlotOf, (size_t)n * sizeof( uint32_t ), hipMemcpyHostToDevice, c->stream ) );
    HIP_TRY( c, hipEventRecord( c->evStep[1], c->stream ) );
    auto streamOf = [&] ( int g ) { return g == expensiveGroup ? c->gstream[MAX_GROUPS - 1] : c->gstream[g]; };
    for ( int g = 1; g < nGroups; ++g ) {
        HIP_TRY( c, hipStreamWaitEvent( streamOf( g ), c->evStep[1], 0 ) );
    }

    /* three or more contexts alive on this device: batches run side by side (a reader, the bench), kernels are chosen for
     * the throughput of the crowd; one or two: for the latency of the batch */
    const bool crowd = walkChainOf( c->device ).liveContexts.load() >= 3;
    const char* wg = std::getenv( "MI355X_BZ2_WALK_WGS" );   /* tuning knob: workgroups per XCD */
    /* measured, walks in turn: four contexts in flight 64 workgroups 67.4 ms per step, 128: 68.6, 256: 72.8; a single context
     * 64: 85.6 ms per batch, 128: 81.0, 256 (walks of the block groups side by side): 84.0 */
    const uint32_t wgsPerXcd = wg != nullptr && std::atoi( wg ) > 0
                               ? (uint32_t)std::atoi( wg )
                               : ( crowd ? WALK_WGS_CROWD : WALK_WGS_PER_XCD );
    const char* wc = std::getenv( "MI355X_BZ2_WALK_CHUNK" );
    /* Segments per claim.  A lane takes a new segment whenever it has finished one, so a claim has to hold several segments
     * per lane for the lanes to stay busy (segment lengths are geometric: with one segment per lane 22 % of the lanes of a
     * gather instruction are alive, PMC) -- but the segments an XCD has claimed should not span more than a block or two,
     * or its workgroups work on more tables than its L2 holds (claims of 1 024 with 128 workgroups per XCD: FETCH_SIZE of
     * k_walk 14.8 -> 77.6 GB per step).  Workgroups x claim = 32 768 = one block's segments; measured on one box, ms per
     * step (k_walk alone): 128 x 256: 65.1 (17.3), 64 x 256: 63.3 (22.1), 64 x 512: 61.6 (18.7), 32 x 1 024: 60.9 / 62.0 (22.2),
     * 24 x 1 536: 61.6, 16 x 2 048: 65.4.  In a crowd: 32 workgroups per XCD (an eighth of the wave slots) with claims of 1 024. */
    const uint32_t walkChunk = wc != nullptr && std::atoi( wc ) > 0 ? (uint32_t)std::atoi( wc ) : ( crowd && n >= 256 ? 4 * WALK_CHUNK : WALK_CHUNK );

    const char* sw = std::getenv( "MI355X_BZ2_SCAN_WAVES" );   /* tuning knob: 1 = k_hscan<1>, 4 / 8 = k_hscan_spec<4 / 8>, whatever the batch size */
    const uint32_t forcedScanWaves = sw != nullptr && std::atoi( sw ) > 0 ? (uint32_t)std::atoi( sw ) : 0u;
    /* most workgroups of a k_hscan<1> launch (0: one per block).  Ten of them fill the LDS of a CU for as long as
     * their blocks take (10 to 30 ms): the kernels of the other groups and contexts that need LDS wait for them */
    const char* sg = std::getenv( "MI355X_BZ2_SCAN_GRID" );
    const uint32_t scanGrid = sg != nullptr ? (uint32_t)std::atoi( sg ) : 0u;
    const char* bsp = std::getenv( "MI355X_BZ2_BWT_SPLIT" );   /* tuning knob: workgroups per block of the table build (1, 2, 4, 8) */
    const uint32_t bwtSplit = bsp != nullptr ? std::min<uint32_t>( (uint32_t)std::atoi( bsp ), BWT_SPLIT_MAX ) : 0u;
    const char* smx = std::getenv( "MI355X_BZ2_SCAN_MIXED" );   /* 0: off; 4 / 8: that many waves per expensive block; default: by count */
    const uint32_t scanMixed = smx != nullptr ? (uint32_t)std::atoi( smx ) : 1u;
    const char* wsr = std::getenv( "MI355X_BZ2_WALK_SERIAL" );
    const bool walkSerial = !( wsr != nullptr && wsr[0] == '0' );
    const char* mn = std::getenv( "MI355X_BZ2_MTF_NARROW" );   /* 1: 256 lanes per block in k_mtf whatever the batch size */
    const bool mtfNarrow = mn != nullptr && mn[0] == '1';
    const char* st = std::getenv( "MI355X_BZ2_SCAN_TUNE" );
    const uint32_t scanTune = st != nullptr ? (uint32_t)std::atoi( st ) : 0u;
    for ( int launch = 0; launch < nGroups; ++launch ) {
        /* the expensive group is queued first, then the chunks from cheap to less cheap */
        const int g = expensiveGroup >= 0 ? ( launch == 0 ? expensiveGroup : launch - 1 ) : launch;
        const uint32_t m = groupCount[g], first = groupFirst[g];
        hipStream_t q = streamOf( g );
        BlockMeta* const meta = c->dMeta + first;
        HuffMeta* const hmeta = c->dHmeta + first;
        uint8_t* const sel = c->dSel + (size_t)first * SEL_STRIDE;
        uint16_t* const sym = c->dSym + (size_t)first * SYM_STRIDE;
        uint8_t* const stb = c->dStb + (size_t)first * 256;
        uint8_t* const lcol = c->dL + (size_t)first * L_STRIDE;
        uint32_t* const tab = c->dTab + (size_t)first * TAB_STRIDE;
        uint8_t* const rbuf = c->dR + (size_t)first * L_STRIDE;
        uint32_t* const segLen = c->dSegLen + (size_t)first * SEG_STRIDE;
        uint32_t* const segSucc = c->dSegSucc + (size_t)first * SEG_STRIDE;
        uint32_t* const segCont = c->dSegCont + (size_t)first * SEG_STRIDE;
        uint2* const chain = c->dChain + (size_t)first * SEG_STRIDE;
        uint32_t* const stash = c->dStash + (size_t)first * SEG_STRIDE * ( STASH_BYTES / 4 );
        const uint32_t* const order = c->dOrder + first;
        WalkPlan* const plan = c->dPlan + g;
        uint32_t* const walkBlk = c->dWalkBlk + (size_t)g * ( c->capacity + 16 );
        uint32_t* const walkPre = c->dWalkPre + (size_t)g * ( c->capacity + 16 );
        const dim3 walkGrid( WALK_QUEUES * wgsPerXcd );

        {
            ScanMeta* const smeta = c->dSmeta + first;
            HuffTables* const htab = c->dHtab + first;
            uint32_t* const gpos = c->dGpos + (size_t)first * GPOS_STRIDE;
            /* wavefronts per block: one when the batch fills the GPU by itself; four or eight, each on a group of its own
             * (k_hscan_spec, bz2_hscan.hip.h), when few blocks have to be through quickly (their LDS, one build per wave,
             * allows 4 and 2 blocks per CU).  Measured: sixteen waves gain nothing over eight (the chain from group to
             * group and the barriers grow with the waves); eight are faster than four for ONE batch of 320 blocks (15 vs
             * 18 ms) but slower when four such batches run side by side (14.5 vs 13.4 ms per batch): eight up to 384
             * blocks for a caller with one or two contexts, up to 128 in a crowd */
            /* In a big batch the launch of one wave per block lasts as long as its largest block's chain (34 ms for an
             * incompressible block, against 14.6 ms of average wave life): the expensive minority gets its own waves per
             * group (MI355X_BZ2_SCAN_MIXED=0: one wave per block for them too) */
            /* (in a crowd -- batches side by side, the scan of one under the other kernels of the rest -- one wave per block
             * from 800 blocks on: a share of 1 270 blocks 37.5 -> 33.4 ms per step, of 960 blocks 28.1 -> 26.7, of 630 blocks
             * 20.2 -> 20.7, profiles/r03_ab_share.txt) */
            uint32_t scanWaves = forcedScanWaves != 0 ? forcedScanWaves
                                                      : ( n <= ( crowd ? 128u : 384u ) ? 8u : ( n <= ( crowd ? 800u : 1280u ) ? 4u : 1u ) );
            if ( forcedScanWaves == 0 && scanWaves == 1 && g == expensiveGroup && scanMixed != 0 ) {
                scanWaves = scanMixed >= 4 ? scanMixed : ( m <= 128 ? 8u : 4u );
            }
            const auto* const inWords = reinterpret_cast<const uint32_t*>( inBase );
            if ( scanWaves >= 8 ) {
                TIMED_LAUNCH( c, g, q, 12, k_hscan_spec<8>, dim3( m ), dim3( 512 ), 0, q, inWords, inSize, c->dOffsets + first,
                              meta, hmeta, smeta, sel, stb, htab, gpos, m, order, scanTune );
            } else if ( scanWaves >= 4 ) {
                TIMED_LAUNCH( c, g, q, 12, k_hscan_spec<4>, dim3( m ), dim3( 256 ), 0, q, inWords, inSize, c->dOffsets + first,
                              meta, hmeta, smeta, sel, stb, htab, gpos, m, order, scanTune );
            } else {
                uint32_t* scanQueue = nullptr;
                uint32_t grid = m;
                if ( scanGrid != 0 && scanGrid < m ) {
                    HIP_TRY( c, hipMemsetAsync( c->dScanQueue + g, 0, sizeof( uint32_t ), q ) );
                    scanQueue = c->dScanQueue + g;
                    grid = scanGrid;
                }
#define SCAN1( W ) TIMED_LAUNCH( c, g, q, 12, ( k_hscan<1, W> ), dim3( grid ), dim3( 64 ), sizeof( ScanShared<1> ), q, inWords, inSize, \
                                 c->dOffsets + first, meta, hmeta, smeta, sel, stb, htab, gpos, m, order, scanTune, scanQueue )
                SCAN1( REGS_SCAN );
#undef SCAN1
            }
#define HSYM( T ) TIMED_LAUNCH( c, g, q, 13, k_hsym<T>, dim3( ( MAX_SCAN_GROUPS + ( T ) * SYM_CHUNKS - 1 ) / ( ( T ) * SYM_CHUNKS ), m ), dim3( T ), \
                                sizeof( SymShared<T> ), q, inWords, meta, hmeta, smeta, sel, htab, gpos, sym )
            HSYM( SYM_GROUPS );
#undef HSYM
        }
#define MTF256( STRIDE, STREAM, INDEX ) \
        TIMED_LAUNCH( c, g, STREAM, INDEX, ( k_mtf<STRIDE, MTF_THREADS, REGS_MTF> ), dim3( m ), dim3( MTF_THREADS ), \
                      sizeof( MtfShared<STRIDE, MTF_THREADS> ), STREAM, meta, hmeta, sym, stb, lcol, m, order )
        /* Every block belongs to one of the two k_mtf instances (by its symbol count), the other returns at once.  In a
         * small batch each lasts as long as its slowest block (4 and 7 ms): side by side instead of one behind the other. */
        if ( n <= 1280 ) {
            if ( c->sideStream[g] == nullptr ) {
                HIP_TRY( c, hipStreamCreateWithFlags( &c->sideStream[g], hipStreamNonBlocking ) );
                HIP_TRY( c, hipEventCreateWithFlags( &c->evFork[g], hipEventDisableTiming ) );
                HIP_TRY( c, hipEventCreateWithFlags( &c->evJoin[g], hipEventDisableTiming ) );
            }
            hipStream_t side = c->sideStream[g];
            HIP_TRY( c, hipEventRecord( c->evFork[g], q ) );
            HIP_TRY( c, hipStreamWaitEvent( side, c->evFork[g], 0 ) );
            /* (up to 640 blocks: one batch of 320 blocks 30.4 -> 28.0 ms, but four side by side 11.9 -> 12.1 ms per batch) */
            if ( n <= ( crowd ? 256u : 640u ) && !mtfNarrow ) {
                /* few blocks: 512 lanes per block, each with half the symbols */
                if ( n <= 64 ) {
                    /* (the 128-entry lists of 1 024 lanes still fit the LDS of a CU: 152 KB) */
                    TIMED_LAUNCH( c, g, side, 11, ( k_mtf<MTF_SMALL_STRIDE, 1024> ), dim3( m ), dim3( 1024 ), sizeof( MtfShared<MTF_SMALL_STRIDE, 1024> ), side, meta, hmeta, sym, stb, lcol, m, order );
                } else {
                    TIMED_LAUNCH( c, g, side, 11, ( k_mtf<MTF_SMALL_STRIDE, 512> ), dim3( m ), dim3( 512 ), sizeof( MtfShared<MTF_SMALL_STRIDE, 512> ), side, meta, hmeta, sym, stb, lcol, m, order );
                }
                HIP_TRY( c, hipEventRecord( c->evJoin[g], side ) );
                TIMED_LAUNCH( c, g, q, 1, ( k_mtf<MTF_LANE_STRIDE, 512> ), dim3( m ), dim3( 512 ), sizeof( MtfShared<MTF_LANE_STRIDE, 512> ), q, meta, hmeta, sym, stb, lcol, m, order );
            } else {
                MTF256( MTF_SMALL_STRIDE, side, 11 );
                HIP_TRY( c, hipEventRecord( c->evJoin[g], side ) );
                MTF256( MTF_LANE_STRIDE, q, 1 );
            }
            HIP_TRY( c, hipStreamWaitEvent( q, c->evJoin[g], 0 ) );
        } else {
            MTF256( MTF_SMALL_STRIDE, q, 11 );
            MTF256( MTF_LANE_STRIDE, q, 1 );
        }
#undef MTF256
        /* table build: one workgroup per block when the batch fills the GPU with that (1 024 threads each: 512 at a time);
         * fewer blocks are spread over 2, 4 or 8 workgroups each (a lone block: 1.0 -> 0.3 ms) */
        const uint32_t bwtSlices = bwtSplit != 0 ? bwtSplit : ( n > BWT_SPLIT_BLOCKS ? 1u : ( n > 256 ? 2u : ( n > 128 ? 4u : BWT_SPLIT_MAX ) ) );
        if ( bwtSlices > 1 && n <= BWT_SPLIT_BLOCKS ) {
            uint32_t* const counts = c->dBwtCounts + (size_t)first * BWT_COUNTS_PER_BLOCK;
            c->launched[g] |= 1u << 2;
            HIP_TRY( c, hipEventRecord( c->ev[g][2 * 2], q ) );
            hipLaunchKernelGGL( k_bwt_count, dim3( bwtSlices, m ), dim3( 1024 ), 0, q, meta, lcol, counts, bwtSlices );
            hipLaunchKernelGGL( k_bwt_rank, dim3( bwtSlices, m ), dim3( 1024 ), 0, q, meta, lcol, tab, counts, bwtSlices );
            HIP_TRY( c, hipEventRecord( c->ev[g][2 * 2 + 1], q ) );
        } else {
            TIMED_LAUNCH( c, g, q, 2, k_bwt_build, dim3( m ), dim3( 1024 ), 0, q, meta, lcol, tab );
        }
        TIMED_LAUNCH( c, g, q, 10, k_walk_plan, dim3( 1 ), dim3( 256 ), 0, q, meta, m, plan, walkBlk, walkPre );
        if ( walkSerial ) {
            WalkChain& walks = walkChainOf( c->device );
            const std::scoped_lock chain( walks.mutex );
            if ( walks.last != nullptr ) HIP_TRY( c, hipStreamWaitEvent( q, walks.last, 0 ) );
            TIMED_LAUNCH( c, g, q, 3, k_walk, walkGrid, dim3( WALK_THREADS ), 0, q,
                          meta, tab, plan, walkBlk, walkPre, segLen, segSucc, walkChunk, stash, segCont );
            hipEvent_t& slot = walks.events[walks.next++ % 64];
            if ( slot == nullptr ) HIP_TRY( c, hipEventCreateWithFlags( &slot, hipEventDisableTiming ) );
            HIP_TRY( c, hipEventRecord( slot, q ) );
            walks.last = slot;
        } else {
            TIMED_LAUNCH( c, g, q, 3, k_walk, walkGrid, dim3( WALK_THREADS ), 0, q,
                          meta, tab, plan, walkBlk, walkPre, segLen, segSucc, walkChunk, stash, segCont );
        }
        TIMED_LAUNCH( c, g, q, 4, k_link2, dim3( m ), dim3( LINK_THREADS ), sizeof( LinkShared ), q, meta, segLen, segSucc, chain );
        TIMED_LAUNCH( c, g, q, 5, k_emit, dim3( ( SEG_STRIDE + EMIT_THREADS * EMIT_TILES - 1 ) / ( EMIT_THREADS * EMIT_TILES ), m ), dim3( EMIT_THREADS ), 0, q,
                      meta, tab, chain, stash, segCont, rbuf );
        static_assert( (size_t)SEG_STRIDE * STASH_BYTES >= L_STRIDE );
        TIMED_LAUNCH( c, g, q, 6, k_replicate, dim3( m ), dim3( 256 ), 0, q, meta, rbuf, reinterpret_cast<uint8_t*>( stash ),
                      (size_t)SEG_STRIDE * STASH_BYTES );
        TIMED_LAUNCH( c, g, q, 7, k_rle<false>, dim3( m ), dim3( RLE_THREADS ), 0, q, meta, rbuf, (uint8_t*)nullptr );
        if ( g >= 1 ) {
            HIP_TRY( c, hipEventRecord( c->evGroupDone[g], q ) );
        }
    }
    HIP_TRY( c, hipGetLastError() );
    for ( int g = 1; g < nGroups; ++g ) {
        HIP_TRY( c, hipStreamWaitEvent( c->stream, c->evGroupDone[g], 0 ) );
    }

    /* Output offsets = exclusive scan of the decoded sizes IN INPUT ORDER (ragged, gap-free), on the device; then the
     * expansion and the CRC, queued right behind: no round trip to the host in the middle of a batch.  The output buffer
     * is chosen now, for the size that blocks of the usual compressors have at most (900 000 bytes each); if these
     * decode to more, k_offsets says so, the two kernels do nothing and decode_batch_end repeats them with a buffer of
     * the right size. */
    if ( c->outputHold != nullptr ) {
        /* somebody still reads the last batch's bytes on the device (mi355x_bz2_hold_output_until) */
        HIP_TRY( c, hipStreamWaitEvent( c->stream, c->outputHold, 0 ) );
        c->outputHold = nullptr;
    }
    hipLaunchKernelGGL( k_offsets, dim3( 1 ), dim3( OFFSETS_THREADS ), 0, c->stream, c->dMeta, c->dSlotOf, n,
                        c->out[c->outCurrent].capacity, c->dTotals );
    TIMED_LAUNCH( c, 0, c->stream, 8, k_rle<true>, dim3( n ), dim3( RLE_THREADS ), 0, c->stream, c->dMeta, c->dR, c->dOut,
                  c->dTotals + 1 );
    TIMED_LAUNCH( c, 0, c->stream, 9, k_crc, dim3( n ), dim3( CRC_THREADS ), 0, c->stream, c->dMeta, c->dOut, c->crc,
                  c->dTotals + 1 );
    HIP_TRY( c, hipEventRecord( c->evStep[2], c->stream ) );
    HIP_TRY( c, hipGetLastError() );
    HIP_TRY( c, hipMemcpyAsync( c->hMeta, c->dMeta, (size_t)n * sizeof( BlockMeta ), hipMemcpyDeviceToHost, c->stream ) );
    HIP_TRY( c, hipMemcpyAsync( c->hTotals, c->dTotals, 2 * sizeof( uint64_t ), hipMemcpyDeviceToHost, c->stream ) );
    c->pendingBlocks = n;
    c->pendingGroups = nGroups;
    c->pendingExpensive = expensiveGroup;
    for ( int g = 0; g < MAX_GROUPS; ++g ) c->pendingGroupCount[g] = groupCount[g];
    if ( traceBegin ) {
        const auto ms = [] ( auto a, auto b ) { return std::chrono::duration<double, std::milli>( b - a ).count(); };
        std::fprintf( stderr, "[device] begin of %u blocks: scratch %.1f ms, input %.1f ms, plan + launches %.1f ms\n", n,
                      ms( tBegin, tScratch ), ms( tScratch, tInput ), ms( tInput, std::chrono::steady_clock::now() ) );
    }
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_decode_batch_end( mi355x_bz2_ctx* c, mi355x_bz2_block_result* results, uint64_t* totalDecoded )
{
    if ( c == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( totalDecoded ) *totalDecoded = 0;
    const uint32_t n = c->pendingBlocks;
    if ( n == 0 ) return MI355X_BZ2_OK;   /* empty batch */
    if ( results == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    c->pendingBlocks = 0;
    const int nGroups = c->pendingGroups, expensiveGroup = c->pendingExpensive;
    const uint32_t* const groupCount = c->pendingGroupCount;
    int rc = MI355X_BZ2_OK;
    HIP_TRY( c, hipSetDevice( c->device ) );
    HIP_TRY( c, hipStreamSynchronize( c->stream ) );
    uint64_t total = c->hTotals[0];
    if ( c->hTotals[1] != 0 ) {
        /* the bytes did not fit into the buffer chosen in decode_batch_begin (blocks that decode to more than 900 000
         * bytes each): offsets on the host, a buffer of the right size, expansion and CRC once more */
        total = 0;
        for ( uint32_t i = 0; i < n; ++i ) {
            BlockMeta& m = c->hMeta[c->hSlotOf[i]];
            m.out_off = total;
            if ( m.walk_ok ) total += m.decoded_size;
        }
        rc = ensureOutput( c, total );
        if ( rc != MI355X_BZ2_OK ) return rc;
        HIP_TRY( c, hipMemcpyAsync( c->dMeta, c->hMeta, (size_t)n * sizeof( BlockMeta ), hipMemcpyHostToDevice, c->stream ) );
        TIMED_LAUNCH( c, 0, c->stream, 8, k_rle<true>, dim3( n ), dim3( RLE_THREADS ), 0, c->stream, c->dMeta, c->dR, c->dOut,
                      static_cast<const uint64_t*>( nullptr ) );
        TIMED_LAUNCH( c, 0, c->stream, 9, k_crc, dim3( n ), dim3( CRC_THREADS ), 0, c->stream, c->dMeta, c->dOut, c->crc,
                      static_cast<const uint64_t*>( nullptr ) );
        HIP_TRY( c, hipEventRecord( c->evStep[2], c->stream ) );
        HIP_TRY( c, hipGetLastError() );
        HIP_TRY( c, hipMemcpyAsync( c->hMeta, c->dMeta, (size_t)n * sizeof( BlockMeta ), hipMemcpyDeviceToHost, c->stream ) );
        HIP_TRY( c, hipStreamSynchronize( c->stream ) );
    }
    c->outSizeHint = std::max( c->outSizeHint, total );

    for ( uint32_t i = 0; i < n; ++i ) {
        const BlockMeta& m = c->hMeta[c->hSlotOf[i]];
        mi355x_bz2_block_result& r = results[i];
        r.encoded_offset_bits = m.enc_off;
        /* set by the reference only after the symbol loop AND the origPtr check (bzip2.hpp:794-806); EOS: header only */
        r.encoded_size_bits = ( m.status == ST_OK || m.status == ST_CRC ) ? m.enc_size : 0;
        r.decoded_size = m.status == ST_OK || m.status == ST_CRC ? m.decoded_size : 0;
        r.data_offset = m.out_off;
        r.header_crc = m.header_crc;
        r.computed_crc = m.computed_crc;
        /* the reference (and the oracle) only know N and the symbol count once the symbol loop has completed */
        const bool loopDone = m.status == ST_OK || m.status == ST_CRC || m.status == ST_ORIGPTR_DATA;
        r.bwt_length = loopDone ? m.n : 0;
        r.orig_ptr = m.orig_ptr;
        r.n_symbols = loopDone ? m.nsym : 0;
        r.is_eos = m.is_eos;
        r.is_eof = m.is_eof;
        r.status = m.status;
    }
    c->outSize = total;
    c->lastBlocks = n;
    if ( totalDecoded ) *totalDecoded = total;

    float ms = 0;
    c->timings = {};
    c->timings.n_kernels = N_KERNELS;
    if ( const char* trace = std::getenv( "MI355X_BZ2_TRACE" ); trace != nullptr && trace[0] == '1' ) {
        /* per-group timeline relative to the step start: [start, end] of every kernel in ms */
        for ( int g = 0; g < nGroups; ++g ) {
            std::fprintf( stderr, "[mi355x_bz2] group %d%s (%u blocks):", g, g == expensiveGroup ? " expensive" : "",
                          groupCount[g] );
            for ( uint32_t k = 0; k < N_KERNELS; ++k ) {
                if ( !( c->launched[g] & ( 1u << k ) ) ) continue;
                float t0 = 0, t1 = 0;
                (void)hipEventElapsedTime( &t0, c->evStep[0], c->ev[g][2 * k] );
                (void)hipEventElapsedTime( &t1, c->evStep[0], c->ev[g][2 * k + 1] );
                std::fprintf( stderr, " %s[%.1f-%.1f]", KERNEL_NAMES[k], t0, t1 );
            }
            std::fprintf( stderr, "\n" );
        }
    }
    if ( hipEventElapsedTime( &ms, c->evStep[0], c->evStep[2] ) == hipSuccess ) c->timings.ms_total = ms;   /* wall */
    /* the per-kernel durations are read from the events on demand (mi355x_bz2_last_timings): ~90 event queries per
     * batch cost milliseconds of host time that a caller who does not ask should not pay */
    c->timingGroups = nGroups;
    c->timingsResolved = false;
    return MI355X_BZ2_OK;
}

const void*
mi355x_bz2_output_device( const mi355x_bz2_ctx* c )
{
    return c != nullptr ? c->dOut : nullptr;
}

int
mi355x_bz2_copy_output( mi355x_bz2_ctx* c, uint64_t offset, uint64_t size, void* hostDst )
{
    if ( c == nullptr || ( hostDst == nullptr && size > 0 ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( offset + size > c->outSize ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    if ( size == 0 ) return MI355X_BZ2_OK;
    HIP_TRY( c, hipSetDevice( c->device ) );
    HIP_TRY( c, hipMemcpyAsync( hostDst, c->dOut + offset, size, hipMemcpyDeviceToHost, c->stream ) );
    HIP_TRY( c, hipStreamSynchronize( c->stream ) );
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_hold_output_until( mi355x_bz2_ctx* c, void* hipEvent )
{
    if ( c == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( c->pendingBlocks != 0 ) {
        c->lastError = "hold_output_until: a batch is in flight (its output kernels are queued already)";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    c->outputHold = static_cast<hipEvent_t>( hipEvent );
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_copy_output_begin( mi355x_bz2_ctx* c, uint64_t offset, uint64_t size, void* hostDst )
{
    if ( c == nullptr || ( hostDst == nullptr && size > 0 ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( offset + size > c->outSize || c->pendingBlocks != 0 ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    HIP_TRY( c, hipSetDevice( c->device ) );
    auto& buffer = c->out[c->outCurrent];
    if ( c->copyStream == nullptr ) HIP_TRY( c, hipStreamCreateWithFlags( &c->copyStream, hipStreamNonBlocking ) );
    if ( buffer.copied == nullptr ) HIP_TRY( c, hipEventCreateWithFlags( &buffer.copied, hipEventDisableTiming ) );
    /* decode_batch_end has synchronised c->stream: the bytes are there */
    if ( size > 0 ) HIP_TRY( c, hipMemcpyAsync( hostDst, buffer.bytes + offset, size, hipMemcpyDeviceToHost, c->copyStream ) );
    HIP_TRY( c, hipEventRecord( buffer.copied, c->copyStream ) );
    buffer.copyIssued = true;
    c->outLastCopy = c->outCurrent;
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_copy_output_end( mi355x_bz2_ctx* c )
{
    if ( c == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    hipEvent_t event = nullptr;
    {
        const std::scoped_lock lock( c->mutex );
        event = c->out[c->outLastCopy].copied;
    }
    if ( event == nullptr ) return MI355X_BZ2_OK;   /* no background copy was ever started */
    if ( hipEventSynchronize( event ) != hipSuccess ) {
        const std::scoped_lock lock( c->mutex );
        c->lastError = "hipEventSynchronize( copied ) failed";
        return MI355X_BZ2_ERR_DEVICE;
    }
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_last_timings( const mi355x_bz2_ctx* c, mi355x_bz2_timings* t )
{
    if ( c == nullptr || t == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    if ( !c->timingsResolved && c->lastBlocks > 0 ) {
        mi355x_bz2_ctx* const m = const_cast<mi355x_bz2_ctx*>( c );
        float ms = 0;
        for ( uint32_t k = 0; k < N_KERNELS; ++k ) {
            for ( int g = 0; g < c->timingGroups; ++g ) {
                if ( !( c->launched[g] & ( 1u << k ) ) ) continue;   /* e.g. 8, 9: once for the whole batch */
                if ( hipEventElapsedTime( &ms, c->ev[g][2 * k], c->ev[g][2 * k + 1] ) == hipSuccess ) {
                    m->timings.ms_kernel[k] += ms;
                    m->timings.ms_kernel_sum += ms;
                }
            }
        }
        m->timingsResolved = true;
    }
    *t = c->timings;
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_last_pipeline_ms( const mi355x_bz2_ctx* c, float* milliseconds )
{
    if ( c == nullptr || milliseconds == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    *milliseconds = c->timings.ms_total;
    return MI355X_BZ2_OK;
}

void*
mi355x_bz2_stream( const mi355x_bz2_ctx* c )
{
    return c != nullptr ? static_cast<void*>( c->stream ) : nullptr;
}

int
mi355x_bz2_device_memory( const mi355x_bz2_ctx* c, uint64_t* scratchBytes, uint64_t* outputBytes )
{
    if ( c == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( scratchBytes != nullptr ) *scratchBytes = c->scratchBytes;
    if ( outputBytes != nullptr ) *outputBytes = c->out[0].capacity + c->out[1].capacity;
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_find_magic_device( mi355x_bz2_ctx* c, uint64_t magic48, uint64_t* bitOffsets, uint64_t capacity,
                              uint64_t* nFound )
{
    if ( c == nullptr || nFound == nullptr || ( capacity > 0 && bitOffsets == nullptr ) ) {
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    *nFound = 0;
    constexpr uint32_t CAP = 1u << 20;
    const std::scoped_lock scanLock( c->scanMutex );   /* one scan at a time per context: they share the buffers below */
    std::shared_ptr<InputUpload> upload;
    const uint8_t* inBase = nullptr;
    uint64_t inSize = 0;
    {
        /* the context's own lock only for the set-up: batches may be launched on it while the scan waits or runs */
        const std::scoped_lock lock( c->mutex );
        if ( c->dIn == nullptr && !c->upload ) {
            c->lastError = "no input set";
            return MI355X_BZ2_ERR_INVALID_ARGUMENT;
        }
        if ( c->inSize < 6 ) return MI355X_BZ2_OK;
        HIP_TRY( c, hipSetDevice( c->device ) );
        if ( c->scanStream == nullptr ) HIP_TRY( c, hipStreamCreateWithFlags( &c->scanStream, hipStreamNonBlocking ) );
        if ( c->dScanFound == nullptr ) HIP_TRY( c, hipMalloc( &c->dScanFound, (size_t)CAP * sizeof( uint64_t ) ) );
        if ( c->dScanCounter == nullptr ) HIP_TRY( c, hipMalloc( &c->dScanCounter, sizeof( uint32_t ) ) );
        upload = c->upload;
        inBase = c->dIn;
        inSize = c->inSize;
        if ( !upload && c->inReady != nullptr ) {
            /* a copy queued by set_input_host_async is on the input stream */
            HIP_TRY( c, hipStreamWaitEvent( c->scanStream, c->inReady, 0 ) );
        }
    }
    const auto fail = [c] ( const char* what ) {
        const std::scoped_lock lock( c->mutex );
        c->lastError = what;
        return MI355X_BZ2_ERR_DEVICE;
    };
    if ( upload && upload->total != 0 ) {
        /* the whole streamed copy: wait until its last piece is queued, then order the scan behind it */
        std::unique_lock lock( upload->mutex );
        upload->changed.wait( lock, [&] { return upload->failed || upload->queued >= upload->total; } );
        if ( upload->failed ) {
            lock.unlock();
            return fail( "the streamed copy of the input failed" );
        }
        inBase = upload->main;
        inSize = upload->total;
        if ( hipStreamWaitEvent( c->scanStream, upload->done.back(), 0 ) != hipSuccess ) {
            lock.unlock();
            return fail( "hipStreamWaitEvent( scan ) failed" );
        }
    }
    uint32_t count = 0;
    std::vector<uint64_t> host;
    if ( hipMemsetAsync( c->dScanCounter, 0, sizeof( uint32_t ), c->scanStream ) != hipSuccess ) return fail( "k_find_magic failed" );
    hipLaunchKernelGGL( k_find_magic, dim3( 4096 ), dim3( 256 ), 0, c->scanStream,
                        reinterpret_cast<const uint32_t*>( inBase ), inSize * 8, magic48 & 0xFFFFFFFFFFFFULL,
                        c->dScanFound, CAP, c->dScanCounter );
    if ( hipMemcpyAsync( &count, c->dScanCounter, sizeof( uint32_t ), hipMemcpyDeviceToHost, c->scanStream ) != hipSuccess
         || hipStreamSynchronize( c->scanStream ) != hipSuccess ) return fail( "k_find_magic failed" );
    const uint32_t stored = std::min( count, CAP );
    host.resize( stored );
    if ( stored > 0
         && ( hipMemcpyAsync( host.data(), c->dScanFound, (size_t)stored * sizeof( uint64_t ), hipMemcpyDeviceToHost,
                              c->scanStream ) != hipSuccess
              || hipStreamSynchronize( c->scanStream ) != hipSuccess ) ) return fail( "k_find_magic failed" );
    if ( count > CAP ) return MI355X_BZ2_ERR_OUTPUT_CAPACITY;
    std::sort( host.begin(), host.end() );
    *nFound = host.size();
    for ( uint64_t i = 0; i < host.size() && i < capacity; ++i ) bitOffsets[i] = host[i];
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_crc32_device( mi355x_bz2_ctx* c, const void* deviceBytes, const uint64_t* sizes, uint32_t nPieces, uint32_t* crcs )
{
    if ( c == nullptr || ( nPieces > 0 && ( deviceBytes == nullptr || sizes == nullptr || crcs == nullptr ) ) ) {
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    if ( nPieces == 0 ) return MI355X_BZ2_OK;
    const std::scoped_lock lock( c->mutex );
    if ( c->pendingBlocks != 0 || ( reinterpret_cast<uintptr_t>( deviceBytes ) & 15u ) != 0 ) {
        c->lastError = "crc32_device: a batch is in flight, or the buffer is not 16-byte aligned";
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    HIP_TRY( c, hipSetDevice( c->device ) );
    /* k_crc as it runs behind a batch, over records that describe the pieces */
    std::vector<BlockMeta> records( nPieces );
    uint64_t at = 0;
    for ( uint32_t i = 0; i < nPieces; ++i ) {
        BlockMeta m{};
        m.decoded_size = sizes[i];
        m.out_off = at;
        m.walk_ok = 1;
        records[i] = m;
        at += sizes[i];
    }
    BlockMeta* dRecords = nullptr;
    HIP_TRY( c, hipMalloc( &dRecords, (size_t)nPieces * sizeof( BlockMeta ) ) );
    int rc = MI355X_BZ2_OK;
    if ( hipMemcpyAsync( dRecords, records.data(), (size_t)nPieces * sizeof( BlockMeta ), hipMemcpyHostToDevice, c->stream ) != hipSuccess ) {
        rc = MI355X_BZ2_ERR_DEVICE;
    } else {
        hipLaunchKernelGGL( k_crc, dim3( nPieces ), dim3( CRC_THREADS ), 0, c->stream, dRecords,
                            static_cast<const uint8_t*>( deviceBytes ), c->crc, static_cast<const uint64_t*>( nullptr ) );
        if ( hipGetLastError() != hipSuccess
             || hipMemcpyAsync( records.data(), dRecords, (size_t)nPieces * sizeof( BlockMeta ), hipMemcpyDeviceToHost, c->stream ) != hipSuccess
             || hipStreamSynchronize( c->stream ) != hipSuccess ) {
            rc = MI355X_BZ2_ERR_DEVICE;
        }
    }
    (void)hipFree( dRecords );
    if ( rc != MI355X_BZ2_OK ) {
        c->lastError = "crc32_device: the checksum kernel failed";
        return rc;
    }
    for ( uint32_t i = 0; i < nPieces; ++i ) crcs[i] = records[i].computed_crc;
    return MI355X_BZ2_OK;
}

int
mi355x_bz2_debug_copy_stage( mi355x_bz2_ctx* c, uint32_t index, int stage, void* hostDst, uint64_t capacity )
{
    if ( c == nullptr || hostDst == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    const std::scoped_lock lock( c->mutex );
    if ( index >= c->lastBlocks ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    if ( ( c->flags & MI355X_BZ2_FLAG_KEEP_STAGES ) == 0 && ( stage == 0 || stage == 2 ) ) {
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;      /* the two share their memory in a context made without the flag */
    }
    index = c->hSlotOf[index];   /* per-block buffers are in slot order */
    const uint64_t N = c->hMeta[index].n;
    const void* src = nullptr;
    uint64_t bytes = 0;
    switch ( stage ) {
    case 0: src = c->dL + (size_t)index * L_STRIDE; bytes = N; break;
    case 1: src = c->dTab + (size_t)index * TAB_STRIDE; bytes = N * 4; break;
    case 2: src = c->dR + (size_t)index * L_STRIDE; bytes = N; break;
    case 3: src = c->dGpos + (size_t)index * GPOS_STRIDE; bytes = (uint64_t)GPOS_STRIDE * 4; break;   /* group starts (k_hscan) */
    case 4: src = c->dSmeta + index; bytes = sizeof( ScanMeta ); break;
    case 5: src = c->dSel + (size_t)index * SEL_STRIDE; bytes = SEL_STRIDE; break;
    default: return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    if ( bytes > capacity ) bytes = capacity;
    if ( bytes == 0 ) return MI355X_BZ2_OK;
    HIP_TRY( c, hipSetDevice( c->device ) );
    HIP_TRY( c, hipMemcpyAsync( hostDst, src, bytes, hipMemcpyDeviceToHost, c->stream ) );
    HIP_TRY( c, hipStreamSynchronize( c->stream ) );
    return MI355X_BZ2_OK;
}

}  // extern "C"
