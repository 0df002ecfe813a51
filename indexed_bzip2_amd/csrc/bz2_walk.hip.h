/**
 * bz2_walk.hip.h -- inverse-BWT walk, XCD-affine work-queue form (replaces the first version's one-lane-per-segment grid).
 *
 * Why: PMC showed each of the 2 x 2.3 G four-byte gathers of the grid version pulling a full 64-B line across the fabric
 * (FETCH_SIZE 127 GB per pass for 9 GB of tables): ~60 blocks' tables (230 MB) were in flight at once, so nothing
 * stayed in an XCD's 4 MiB L2.  Here every block is assigned to ONE XCD queue (block i -> queue i mod 8), each
 * workgroup reads its XCC id (s_getreg HW_REG_XCC_ID) and pulls chunks of consecutive segments from its own XCD's
 * queue, so an XCD works on one or two 3.6 MB tables at a time.  Workgroups per XCD: 256 (all wave slots) are best for a
 * walk that has the GPU to itself, but it never has: with the walks of all groups and contexts taking turns (bz2_device.hip)
 * and the other kernels filling the wave slots, 128 (64 with three or more contexts in flight) give the shortest steps.  The gathers still miss
 * L2 mostly (35 B of fabric traffic per step, PMC) but are served by the 256 MB MALL rather than HBM.  Lanes refill
 * from the chunk through an LDS counter, so a short segment does not idle its lane.  A workgroup whose queue is
 * exhausted helps the other queues; nobody ever waits on another workgroup, every loop is bounded by atomic counters
 * that only grow, so the grid always drains.  Placement affects speed only.
 *
 * There is ONE gather pass: the walk keeps the first STASH_BYTES bytes of every segment (k_emit puts them in output
 * order after k_link2 has ordered the segments), only longer segments are walked again from where the stash ends.
 *
 * Reference: the N-step dependent walk of BurrowsWheelerTransformData::decodeBlock, bzip2.hpp:872-879.
 */
#pragma once

#include "bz2_kernels.hip.h"

namespace bz2gpu
{
constexpr uint32_t WALK_THREADS = 256;
constexpr uint32_t WALK_CHUNK = 256;         /* segments per queue grab */
constexpr uint32_t WALK_WGS_PER_XCD = 128;    /* one or two contexts */
constexpr uint32_t WALK_WGS_CROWD = 32;       /* three or more contexts alive, with claims of 4 x WALK_CHUNK: see bz2_device.hip */
constexpr uint32_t WALK_QUEUES = 8;
constexpr uint32_t STASH_BYTES = 128;        /* bytes of a segment the first walk keeps (two 64-B lines per segment: 1 % of the
                                                bytes lie beyond, 10 % with one line).  Round 3 measured a quarter of the
                                                segments with 512 stashed bytes each: k_link2 4.7 -> 1.8 ms, but k_walk 21 -> 32
                                                and k_emit 5.3 -> 10.8 ms for the bench's batch; gone again */
constexpr uint32_t EMIT_THREADS = 256;       /* segments (consecutive along the cycle) per k_emit workgroup */
constexpr uint32_t EMIT_TILES = 4;           /* such pieces per workgroup, one after the other */
constexpr uint32_t EMIT_STAGE = 16384;       /* LDS bytes that collect their output before it is written in whole lines */

struct WalkPlan
{
    uint32_t q_begin[WALK_QUEUES + 1];   /* entry range of queue q in blk[] / pre[] */
    uint32_t q_total[WALK_QUEUES];       /* segments in queue q */
    uint32_t ctr[WALK_QUEUES];           /* next unclaimed segment of the queue (atomic) */
    uint32_t pad[15];
};

/** One workgroup: lays out the queues (block i -> queue i mod 8) and zeroes the counters. */
__global__ __launch_bounds__( 256 ) void
k_walk_plan( const BlockMeta* __restrict__ meta, uint32_t n, WalkPlan* plan, uint32_t* blk, uint32_t* pre )
{
    __shared__ uint32_t begin[WALK_QUEUES + 1];
    const uint32_t t = threadIdx.x;
    if ( t == 0 ) {
        uint32_t acc = 0;
        for ( uint32_t q = 0; q < WALK_QUEUES; ++q ) {
            begin[q] = acc;
            acc += q < n ? ( n - q + WALK_QUEUES - 1 ) / WALK_QUEUES : 0;
        }
        begin[WALK_QUEUES] = acc;
    }
    __syncthreads();
    if ( t <= WALK_QUEUES ) plan->q_begin[t] = begin[t];
    if ( t < WALK_QUEUES ) {
        uint32_t idx = begin[t], acc = 0;
        for ( uint32_t i = t; i < n; i += WALK_QUEUES ) {
            blk[idx] = i;
            pre[idx] = acc;
            acc += meta[i].walk_ok ? meta[i].nseg : 0u;
            ++idx;
        }
        plan->q_total[t] = acc;
        plan->ctr[t] = 0;
    }
}

/* =============================================================================================================
 * k_link2: order the segments along the cycle that starts at origPtr and give each its output offset.
 * LF is always a permutation, so the successor chain returns to its first segment after c = (cycle length) steps.
 * For ordinary data c == N.  For periodic data (e.g. "abab...": the sorted rotations repeat) the permutation splits
 * into N/c cycles and the reference's N-step walk (bzip2.hpp:872-879) goes round the origPtr cycle N/c times: the
 * output is the first period repeated (k_replicate).  Segments off the cycle keep INVALID_OFF and are not emitted.
 * Corrupt data takes the same path and is caught by the CRC, exactly as in the reference.
 *
 * With up to 32 769 segments the chain is itself cut at SPLITTERS (every 128th segment id + the first segment): each
 * lane follows the successors from its splitter to the next splitter (successors in LDS as u16), lane 0 links the
 * <= 257 sub-chains, and every lane walks its sub-chain again to write the offsets -- the same multi-start idea as the
 * byte walk, one level up.
 * ============================================================================================================= */
constexpr uint32_t LINK_THREADS = 256;
constexpr uint32_t LINK_SPLIT = 128;
constexpr uint32_t LINK_MAX_SUB = KMAX / LINK_SPLIT + 2;   /* 258 */

struct alignas( 16 ) LinkShared
{
    uint16_t ssucc[SEG_STRIDE];          /* 64 KiB */
    uint32_t subLen[LINK_MAX_SUB];       /* bytes covered by sub-chain s */
    uint16_t subNext[LINK_MAX_SUB];      /* sub-chain that follows (index), 0xFFFF if broken */
    uint32_t subOff[LINK_MAX_SUB];       /* output offset of the sub-chain, INVALID_OFF if off the cycle */
    uint32_t subCnt[LINK_MAX_SUB];       /* segments in sub-chain s */
    uint32_t subRank[LINK_MAX_SUB];      /* position of its first segment along the cycle */
};

/* The body of k_link2; its LDS arrays as __restrict__ parameters (pieces of one launch-time allocation, see k_mtf). */
__device__ __forceinline__ void
link_block( BlockMeta*                   meta,
            const uint32_t* __restrict__ seg_len,
            const uint32_t* __restrict__ seg_succ,
            uint2* __restrict__          chain,
            uint16_t* __restrict__       ssucc,
            uint32_t* __restrict__       subLen,
            uint16_t* __restrict__       subNext,
            uint32_t* __restrict__       subOff,
            uint32_t* __restrict__       subCnt,
            uint32_t* __restrict__       subRank )
{
    const uint32_t b = blockIdx.x;
    const BlockMeta mt = meta[b];
    if ( !mt.walk_ok ) return;
    const uint32_t nseg = mt.nseg, N = mt.n, stride = mt.seg_stride, origPtr = mt.orig_ptr;
    const uint32_t k0 = ( N + stride - 1 ) / stride;
    const size_t base = (size_t)b * SEG_STRIDE;
    const uint32_t first = ( origPtr % stride != 0 ) ? k0 : origPtr / stride;
    const uint32_t tid = threadIdx.x;

    for ( uint32_t j = tid; j < nseg; j += LINK_THREADS ) {
        const uint32_t s = seg_succ[base + j];
        ssucc[j] = (uint16_t)( s < nseg ? s : 0xFFFFu );
    }
    /* sub-chain s < nSplit starts at segment s * LINK_SPLIT; sub-chain nSplit starts at `first` (unless that is a
     * multiple of LINK_SPLIT already) */
    const uint32_t nSplit = ( nseg + LINK_SPLIT - 1 ) / LINK_SPLIT;
    const bool firstExtra = first % LINK_SPLIT != 0;
    const uint32_t nSub = nSplit + ( firstExtra ? 1u : 0u );
    const auto splitterOf = [&] ( uint32_t seg ) -> uint32_t {   /* sub-chain index if seg starts one, else 0xFFFF */
        if ( seg == first && firstExtra ) return nSplit;
        return seg % LINK_SPLIT == 0 ? seg / LINK_SPLIT : 0xFFFFu;
    };
    for ( uint32_t s = tid; s < LINK_MAX_SUB; s += LINK_THREADS ) subOff[s] = INVALID_OFF;
    __syncthreads();

    /* pass 1: length and successor of every sub-chain.  The chain itself runs through LDS; the segment lengths come from
     * memory, eight at a time: one load per step of the chain made every step wait for memory (4.6 ms for the bench's batch
     * with one wait per step). */
    constexpr uint32_t AHEAD = 8;
    for ( uint32_t s = tid; s < nSub; s += LINK_THREADS ) {
        uint32_t cur = s < nSplit ? s * LINK_SPLIT : first;
        uint32_t sum = 0, steps = 0, nextSub = 0xFFFFu;
        bool more = true;
        while ( more ) {
            uint32_t nodes[AHEAD];
            uint32_t m = 0;
#pragma unroll
            for ( uint32_t k = 0; k < AHEAD; ++k ) {
                nodes[k] = cur;
                if ( more ) {
                    m = k + 1;
                    cur = ssucc[cur];
                    ++steps;
                    if ( cur == 0xFFFFu || steps > nseg ) {
                        more = false;                                   /* broken chain (cannot happen for a permutation) */
                    } else {
                        nextSub = splitterOf( cur );
                        more = nextSub == 0xFFFFu;
                    }
                }
            }
#pragma unroll
            for ( uint32_t k = 0; k < AHEAD; ++k ) {
                if ( k < m ) sum += seg_len[base + nodes[k]];
            }
        }
        subLen[s] = sum;
        subCnt[s] = steps;
        subNext[s] = (uint16_t)nextSub;
    }
    __syncthreads();

    /* link the sub-chains starting at the one that begins with `first` */
    if ( tid == 0 ) {
        const uint32_t start = firstExtra ? nSplit : first / LINK_SPLIT;
        uint32_t s = start, off = 0, visited = 0, rank = 0;
        bool ok = true;
        do {
            subOff[s] = off;
            subRank[s] = rank;
            off += subLen[s];
            rank += subCnt[s];
            s = subNext[s];
            ++visited;
            if ( s == 0xFFFFu || visited > nSub || off > N ) { ok = false; break; }
        } while ( s != start );
        meta[b].cycle_len = off;
        meta[b].nchain = rank;
        if ( !ok || off == 0 ) {   /* unreachable for a permutation: guards against table corruption */
            meta[b].status = ST_CRC;
            meta[b].walk_ok = 0;
        }
    }
    __syncthreads();

    /* pass 2: offsets of the segments of every sub-chain that lies on the cycle (lengths eight at a time, as above) */
    for ( uint32_t s = tid; s < nSub; s += LINK_THREADS ) {
        uint32_t off = subOff[s];
        if ( off == INVALID_OFF ) continue;
        uint32_t cur = s < nSplit ? s * LINK_SPLIT : first;
        uint32_t steps = 0;
        uint32_t rank = subRank[s];
        bool more = true;
        while ( more ) {
            uint32_t nodes[AHEAD], lens[AHEAD];
            uint32_t m = 0;
#pragma unroll
            for ( uint32_t k = 0; k < AHEAD; ++k ) {
                nodes[k] = cur;
                if ( more ) {
                    m = k + 1;
                    cur = ssucc[cur];
                    ++steps;
                    more = !( cur == 0xFFFFu || steps > nseg || splitterOf( cur ) != 0xFFFFu );
                }
            }
#pragma unroll
            for ( uint32_t k = 0; k < AHEAD; ++k ) lens[k] = k < m ? seg_len[base + nodes[k]] : 0u;
#pragma unroll
            for ( uint32_t k = 0; k < AHEAD; ++k ) {
                if ( k < m ) {
                    const uint32_t len = lens[k];
                    /* record for k_emit, in cycle order: output offset and length (20 bits each), segment (16 bits) */
                    chain[base + rank] = make_uint2( off | ( nodes[k] << 20 ), len | ( ( nodes[k] >> 12 ) << 20 ) );
                    ++rank;
                    off += len;
                }
            }
        }
    }
}

/* LDS declared at launch -- see k_hscan (bz2_hscan.hip.h): the kernel needs 38 registers and was given 169. */
__global__ __launch_bounds__( LINK_THREADS ) void
k_link2( BlockMeta*                   meta,
         const uint32_t* __restrict__ seg_len,
         const uint32_t* __restrict__ seg_succ,
         uint2* __restrict__          chain )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t ldsAtLaunch[];     /* sizeof( LinkShared ) */
    auto& shared = *reinterpret_cast<LinkShared*>( ldsAtLaunch );
    link_block( meta, seg_len, seg_succ, chain, shared.ssucc, shared.subLen, shared.subNext, shared.subOff,
                shared.subCnt, shared.subRank );
}

/* The walk of the segments.  A lane follows ONE segment at a time, sixteen steps (one 16-byte piece of the segment's stash)
 * per round, and takes its next segment at the end of the round in which it finished one.  (Up to round 2 the lanes of a
 * wave took their next segments together, when the longest of them was through: segment lengths are geometric -- mean 27,
 * the longest of 64 about 128 --, so 22 % of the lanes of a gather instruction were alive, PMC: 10.6 G lane slots for 2.3 G
 * steps; the walk is bound by the latency of its dependent gathers, i.e. by how many of them are in flight.)  Within a claim
 * of `chunk` segments the runs that belong to one block are handled one after the other, so that everything a lane needs to
 * start a segment is uniform: its number, the stride, and one table look-up. */
__global__ __launch_bounds__( WALK_THREADS ) void
k_walk( const BlockMeta* __restrict__ meta,
         const uint32_t* __restrict__  tab_buf,
         WalkPlan*                     plan,
         const uint32_t* __restrict__  blk,
         const uint32_t* __restrict__  pre,
         uint32_t*                     seg_len,
         uint32_t*                     seg_succ,
         uint32_t                      chunk,
         uint32_t*                     stash,      /* [block][SEG_STRIDE * STASH_BYTES / 4]: STASH_BYTES bytes per segment */
         uint32_t*                     seg_cont )  /* table index of byte STASH_BYTES of a longer segment */
{
    __shared__ uint32_t sBase, sNext, sK0;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t xcc;
    asm volatile( "s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"( xcc ) );
    xcc &= WALK_QUEUES - 1;

    for ( uint32_t attempt = 0; attempt < WALK_QUEUES; ++attempt ) {
        const uint32_t q = ( xcc + attempt ) & ( WALK_QUEUES - 1 );
        const uint32_t qb = plan->q_begin[q], qe = plan->q_begin[q + 1];
        const uint32_t total = plan->q_total[q];
        if ( total == 0 ) continue;
        for ( ;; ) {
            __syncthreads();   /* everyone is done with sBase / sK0 of the previous claim */
            if ( tid == 0 ) {
                const uint32_t base = atomicAdd( &plan->ctr[q], chunk );
                sBase = base;
                /* last entry k in [qb, qe) with pre[k] <= base */
                uint32_t lo = qb, hi = qe;
                while ( hi - lo > 1 ) {
                    const uint32_t mid = ( lo + hi ) >> 1;
                    if ( pre[mid] <= base ) lo = mid; else hi = mid;
                }
                sK0 = lo;
            }
            __syncthreads();
            uint32_t base = sBase;
            if ( base >= total ) break;   /* queue exhausted (the counter only grows) */
            const uint32_t claimEnd = total - base < chunk ? total : base + chunk;
            uint32_t k = sK0;
            /* the runs of the claim that belong to one block each */
            while ( base < claimEnd ) {
                while ( k + 1 < qe && base >= pre[k + 1] ) ++k;      /* (blocks without segments are skipped) */
                const uint32_t runEnd = ( k + 1 < qe && pre[k + 1] < claimEnd ) ? pre[k + 1] : claimEnd;
                const uint32_t cnt = runEnd - base;
                const uint32_t b = blk[k];
                const uint32_t j0 = base - pre[k];
                const uint32_t N = meta[b].n, stride = meta[b].seg_stride, origPtr = meta[b].orig_ptr;
                const uint32_t k0 = ( N + stride - 1 ) / stride;
                const uint32_t* const tab = tab_buf + (size_t)b * TAB_STRIDE;
                const size_t segBase = (size_t)b * SEG_STRIDE;
                uint32_t* const stashOfBlock = stash + segBase * ( STASH_BYTES / 4 );
                __syncthreads();           /* the previous run's sNext has been read by everybody */
                if ( tid == 0 ) sNext = WALK_THREADS;
                __syncthreads();

                uint32_t my = tid;         /* segment of the run this lane works on / asks for */
                bool idle = true;          /* no segment under way */
                uint32_t j = 0, p = 0, e = 0, len = 0, piece = 0;
                for ( ;; ) {
                    if ( idle && my < cnt ) {
                        j = j0 + my;
                        p = j < k0 ? j * stride : origPtr;
                        e = tab[p];
                        len = 0;
                        piece = 0;
                        idle = false;
                    }
                    if ( __ballot( !idle ) == 0 ) break;     /* this wave's lanes have found no more segments */
                    if ( !idle ) {
                        uint32_t w[4] = { 0, 0, 0, 0 };
                        bool done = false;
                        if ( piece == STASH_BYTES / 16 ) __builtin_nontemporal_store( p, seg_cont + segBase + j );   /* table index of byte STASH_BYTES */
                        /* four steps per look at `done`: a lane whose segment ends inside the four goes on in place (it
                         * re-reads its last entry, adds nothing) instead of costing every step a change of the execution mask */
#pragma unroll
                        for ( uint32_t i = 0; i < 16; i += 4 ) {
                            if ( !done ) {
#pragma unroll
                                for ( uint32_t s4 = i; s4 < i + 4; ++s4 ) {
                                    const uint32_t live = done ? 0u : 1u;
                                    w[s4 >> 2] |= done ? 0u : ( e & 0xFFu ) << ( 8 * ( s4 & 3u ) );
                                    len += live;
                                    p = done ? p : ( e >> 8 ) & LF_MASK;
                                    e = tab[p];
                                    done = ( e & MARK ) || len >= N;
                                }
                            }
                        }
                        if ( piece < STASH_BYTES / 16 ) {
                            /* the segment's first STASH_BYTES bytes, walk order, 16 bytes per store: k_emit then needs no second
                             * gather pass for them.  Written once, read much later: kept out of the way of the table lines */
                            uint32_t* const out = stashOfBlock + (size_t)j * ( STASH_BYTES / 4 ) + 4 * piece;
                            __builtin_nontemporal_store( w[0], out );
                            __builtin_nontemporal_store( w[1], out + 1 );
                            __builtin_nontemporal_store( w[2], out + 2 );
                            __builtin_nontemporal_store( w[3], out + 3 );
                        }
                        ++piece;
                        if ( done ) {
                            __builtin_nontemporal_store( len, seg_len + segBase + j );
                            const bool isOrig = ( p == origPtr ) && ( origPtr % stride != 0 );
                            __builtin_nontemporal_store( ( e & MARK ) ? ( isOrig ? k0 : p / stride ) : 0xFFFFFFFFu, seg_succ + segBase + j );
                            idle = true;
                        }
                    }
                    /* the lanes that are through take the next segments of the run: one LDS atomic per wave */
                    {
                        const uint64_t asking = __ballot( idle );
                        if ( asking != 0 ) {
                            uint32_t first = 0;
                            if ( lane == (uint32_t)__builtin_ctzll( asking ) ) first = atomicAdd( &sNext, (uint32_t)__popcll( asking ) );
                            first = (uint32_t)__builtin_amdgcn_readlane( (int)first, __builtin_ctzll( asking ) );
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi( (uint32_t)( asking >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)asking, 0 ) );
                            if ( idle ) my = first < cnt ? first + rank : cnt;     /* (the counter is left alone once it has passed the end) */
                        }
                    }
                }
                base = runEnd;
            }
        }
    }
}
/**
 * k_emit: the bytes of the walk, in output order.  A workgroup takes EMIT_THREADS segments that are CONSECUTIVE ALONG THE
 * CYCLE (k_link2 wrote them down in that order), so together they cover one contiguous piece of the output.  Every lane
 * fetches its segment's stashed line (the first 64 bytes, kept by the first walk), walks on only if the segment is
 * longer, and drops the bytes at their place in an LDS image of that piece; the image then goes to memory in whole
 * 16-byte units.  The second full gather pass and its partial-line writes (120 GB read + 25 GB written per 2 GiB, PMC)
 * are gone.  Output addresses run backwards: byte i of the segment at offset `off` is R[N - 1 - off - i].
 */
__global__ __launch_bounds__( EMIT_THREADS ) void
k_emit( const BlockMeta* __restrict__ meta,
        const uint32_t* __restrict__  tab_buf,
        const uint2* __restrict__     chain,
        const uint32_t* __restrict__  stash,
        const uint32_t* __restrict__  seg_cont,   /* table index of byte STASH of a longer segment */
        uint8_t* __restrict__         r_buf )
{
    constexpr uint32_t STASH = STASH_BYTES, STAGE = EMIT_STAGE;
    constexpr uint32_t FIRST = 2;                                /* 16-byte pieces of a segment that its own lane takes */
    constexpr uint32_t MORE = STASH / 16 - FIRST;                /* further pieces a segment can have in the stash */
    __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t image[STAGE + 16];
    __shared__ uint32_t sLow, sTop;                              /* lowest / highest output address of the piece */
    __shared__ uint32_t segAddr[EMIT_THREADS], segLen[EMIT_THREADS], segId[EMIT_THREADS];
    __shared__ uint16_t later[EMIT_THREADS * MORE];   /* pieces behind the first two: lane << 3 | piece */
    __shared__ uint32_t waveSum[EMIT_THREADS / 64];
    const uint32_t b = blockIdx.y;
    const BlockMeta mt = meta[b];
    if ( !mt.walk_ok ) return;
    const uint32_t t = threadIdx.x;
    const uint32_t N = mt.n;
    /* EMIT_TILES pieces of the output per workgroup, one after the other: a workgroup costs about 11 ns of dispatch whatever it
     * does (k_hsym with 182 000 / 91 000 / 45 000 workgroups: 4.5 / 3.5 / 3.05 ms), and 330 000 workgroups of one piece each
     * made that 3.5 of this kernel's 5.3 ms */
    for ( uint32_t tile = 0; tile < EMIT_TILES; ++tile ) {
    const uint32_t r0 = ( blockIdx.x * EMIT_TILES + tile ) * EMIT_THREADS;
    if ( r0 >= mt.nchain ) return;
    const uint32_t count = mt.nchain - r0 < EMIT_THREADS ? mt.nchain - r0 : EMIT_THREADS;
    const size_t base = (size_t)b * SEG_STRIDE;
    uint8_t* const R = r_buf + (size_t)b * L_STRIDE;
    const uint32_t* const tab = tab_buf + (size_t)b * TAB_STRIDE;
    const uint32_t* const stashOfBlock = stash + base * ( STASH_BYTES / 4 );

    uint2 rec = make_uint2( 0, 0 );
    if ( t < count ) rec = chain[base + r0 + t];
    const uint32_t off = rec.x & 0xFFFFFu, len = rec.y & 0xFFFFFu, seg = ( rec.x >> 20 ) | ( ( rec.y >> 20 ) << 12 );
    /* the piece: [low, top], top = address of the first byte of the first segment */
    if ( t == count - 1 ) sLow = N - off - len;           /* N - 1 - (off + len - 1) */
    if ( t == 0 ) sTop = N - 1 - off;
    const uint32_t kept = t < count ? ( len < STASH ? len : STASH ) : 0u;
    const uint32_t addr0 = N - 1 - off;                   /* address of byte 0 */
    uint32_t laterTotal = 0;
    {
        /* Segment lengths are geometric (mean 27): a lane that copies ITS segment runs the longest segment's eight pieces with
         * one lane in five alive (3.5 of k_emit's 5.4 ms for the bench's batch were that skeleton).  So a lane copies the
         * first two pieces of its segment -- 70 % of the segments end there -- and the pieces behind them are listed and
         * dealt out to the lanes again. */
        segAddr[t] = addr0;
        segLen[t] = kept;
        segId[t] = seg;
        const uint32_t pieces = ( kept + 15 ) >> 4;
        const uint32_t mine = pieces > FIRST ? pieces - FIRST : 0u;
        uint32_t incl = mine;
#pragma unroll
        for ( int d = 1; d < 64; d <<= 1 ) {
            const uint32_t o = (uint32_t)__shfl_up( (int)incl, d );
            if ( (int)( t & 63u ) >= d ) incl += o;
        }
        if ( ( t & 63u ) == 63u ) waveSum[t >> 6] = incl;
        __syncthreads();
        uint32_t start = incl - mine;
#pragma unroll
        for ( uint32_t w = 0; w < EMIT_THREADS / 64; ++w ) {
            if ( w < ( t >> 6 ) ) start += waveSum[w];
            laterTotal += waveSum[w];
        }
#pragma unroll
        for ( uint32_t k = 0; k < MORE; ++k ) {
            if ( k < mine ) later[start + k] = (uint16_t)( ( t << 3 ) | ( FIRST + k ) );
        }
    }
    __syncthreads();
    const uint32_t low = sLow, top = sTop;
    const uint32_t imageBase = low & ~15u;                /* the image mirrors memory from a 16-byte boundary */

    const auto put = [&] ( uint32_t addr, uint32_t byte ) {
        const uint32_t pos = addr - imageBase;
        if ( pos < STAGE ) image[pos] = (uint8_t)byte; else R[addr] = (uint8_t)byte;   /* oversized piece */
    };
    /* sixteen bytes (fewer at the segment's end) of a segment whose byte 0 goes to address a0 */
    const auto sixteen = [&] ( const uint32_t* line, uint32_t quad, uint32_t a0, uint32_t bytes ) {
        const uint4 v = reinterpret_cast<const uint4*>( line )[quad];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for ( uint32_t i = 0; i < 16; ++i ) {
            if ( quad * 16 + i < bytes ) put( a0 - ( quad * 16 + i ), ( w[i >> 2] >> ( 8 * ( i & 3 ) ) ) & 0xFFu );
        }
    };
    if ( t < count ) {
        const uint32_t* const line = stashOfBlock + (size_t)seg * ( STASH / 4 );
#pragma unroll
        for ( uint32_t quad = 0; quad < FIRST; ++quad ) {
            if ( quad * 16 < kept ) sixteen( line, quad, addr0, kept );
        }
    }
    {
        for ( uint32_t i = t; i < laterTotal; i += EMIT_THREADS ) {
            const uint32_t entry = later[i];
            const uint32_t owner = entry >> 3, quad = entry & 7u;
            sixteen( stashOfBlock + (size_t)segId[owner] * ( STASH / 4 ), quad, segAddr[owner], segLen[owner] );
        }
    }
    if ( t < count && len > STASH ) {
        /* the rare segment that is longer than its stash: the rest of its walk, again */
        uint32_t p = seg_cont[base + seg];
        uint32_t a = addr0 - STASH;
        for ( uint32_t i = STASH; i < len; ++i ) {
            const uint32_t e = tab[p];
            put( a, e & 0xFFu );
            --a;
            p = ( e >> 8 ) & LF_MASK;
        }
    }
    __syncthreads();

    /* image -> memory: whole aligned 16-byte units where the piece covers them, bytes at its two ends (the units there
     * are shared with the neighbouring pieces) */
    const uint32_t endPos = ( top - imageBase + 1 ) < STAGE ? ( top - imageBase + 1 ) : STAGE;   /* exclusive */
    const uint32_t beginPos = low - imageBase;
    for ( uint32_t unit = t * 16; unit < endPos; unit += EMIT_THREADS * 16 ) {
        if ( unit >= beginPos && unit + 16 <= endPos ) {
            *reinterpret_cast<uint4*>( R + imageBase + unit ) = *reinterpret_cast<const uint4*>( image + unit );
        } else {
            for ( uint32_t i = 0; i < 16; ++i ) {
                const uint32_t pos = unit + i;
                if ( pos >= beginPos && pos < endPos ) R[imageBase + pos] = image[pos];
            }
        }
    }
    __syncthreads();     /* the image and the lists are written again by the next piece */
    }
}
}  // namespace bz2gpu
