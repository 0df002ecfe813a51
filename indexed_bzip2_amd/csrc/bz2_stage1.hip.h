/**
 * bz2_stage1.hip.h -- RUNA/RUNB and move-to-front: the Huffman symbols of a block (k_hscan + k_hsym, bz2_hscan.hip.h) -> its
 * L column.  (Round 1's fused stage and round 2's k_huff -- one wavefront per block following the chain of code lengths
 * 64 bit positions at a time on the scalar unit, 61.5 ms for the bench's batch against 29 + 4 ms of the scan and the symbol
 * kernel -- are gone; their measurements are in DESIGN.md.)
 *
 *   k_mtf   256 lanes per block (512 / 1 024 in small batches), each owning a contiguous chunk of the symbol stream:
 *           RUNA/RUNB run lengths and move-to-front (bzip2.hpp:726-790).  MTF is order dependent, so pass A runs every
 *           chunk on an identity list (giving the chunk's permutation and output size), the permutations are composed in
 *           chunk order to get each chunk's true start list, and pass B replays the chunk with it and writes the L column.
 *           Each lane keeps its list in LDS (16-byte groups, lane stride chosen bank-conflict free).
 */
#pragma once

#include "bz2_kernels.hip.h"

namespace bz2gpu
{
constexpr uint32_t SYM_STRIDE = SEG_STRIDE * 64;   /* u16 per block slot.  A block has at most 900 100 symbols (n_sym <= N + 1); the slot
                                                    is as large as the block's slot of the walk's stash (SEG_STRIDE x 128 B), because
                                                    the two share memory: the symbols are dead when k_mtf is through, the stash is
                                                    written by k_walk behind it on the same stream (bz2_device.hip) */
constexpr uint32_t SYM_CAP = 900096;
constexpr uint32_t MTF_THREADS = 256;      /* lanes (= chunks) per block in k_mtf: one workgroup per block */
constexpr uint32_t MTF_SMALL_STRIDE = 144;  /* lists of 128 entries for blocks with few symbols, see k_mtf */
constexpr uint32_t MTF_LANE_STRIDE = 272;  /* bytes between the lists of consecutive lanes: 256 + 16, so that the 16-byte
                                              accesses of the 16 lanes served together fall on 64 distinct banks */
/** Orders LDS traffic between the lanes of ONE wavefront (no s_barrier). */
__device__ __forceinline__ void
wave_sync()
{
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    __builtin_amdgcn_wave_barrier();
    asm volatile( "" ::: "memory" );   /* LDS traffic of one wave is in order in hardware: the compiler has to keep it so */
}

/** Per-block hand-off between the Huffman stage (k_hscan + k_hsym, bz2_hscan.hip.h) and k_mtf. */
struct HuffMeta
{
    uint32_t n_stored;      /* symbols written to sym_buf (without the end-of-block symbol) */
    uint32_t symbol_count;  /* bzip2 symbolCount (used byte values) */
    int32_t  status;        /* first error of the Huffman stage (ST_OK if the end-of-block symbol was reached) */
    uint32_t active;        /* 1 if k_mtf has work (data block whose header parsed) */
};

/** Row p (0..15) of the byte-permute selectors for "insert a byte in front of a 16-byte group and drop byte p":
 * dword j of the group becomes v_perm_b32( d_j, prev_j, sel_j ), prev_j = d_(j-1), prev_0 = the inserted byte in its
 * byte 0.  Selector bytes 0-3 pick from prev_j, 4-7 from d_j: dwords below p's shift up completely, p's dword up to
 * byte p & 3, the rest stays. */
__device__ __forceinline__ uint4
mtf_perm_row( uint32_t p )
{
    const uint32_t qd = p >> 2, r = p & 3u;
    uint32_t sel[4];
#pragma unroll
    for ( uint32_t j = 0; j < 4; ++j ) {
        uint32_t v;
        if ( j < qd || ( j == qd && r == 3 ) ) v = 0x06050403u;      /* full shift */
        else if ( j > qd ) v = 0x07060504u;                          /* keep */
        else v = r == 0 ? 0x07060503u : ( r == 1 ? 0x07060403u : 0x07050403u );
        if ( j == 0 && j <= qd ) v &= 0xFFFFFF00u;                   /* byte 0 of the group = byte 0 of prev_0 */
        sel[j] = v;
    }
    return make_uint4( sel[0], sel[1], sel[2], sel[3] );
}

/** Move entry `ii` of this lane's list to the front; returns the entry.  `mine` = this lane's 256-byte list in LDS
 * (16-byte aligned), `perm_rows` = the 16 rows of mtf_perm_row in LDS.  Entries below `ii` shift up by one byte: whole
 * 16-byte groups with four v_alignbit per ds_read_b128 / ds_write_b128 trip, the group that holds the entry with four
 * v_perm_b32.  Everything the last group needs (the group, the entry, the byte carried in from the group below, the
 * selector row) is read up front, so the loads overlap and nothing depends on the trips. */
__device__ __forceinline__ uint32_t
mtf_lane_move( uint4* mine, const uint4* perm_rows, uint32_t ii )
{
    const uint32_t q4 = ii >> 4;
    const uint8_t* const bytes = reinterpret_cast<const uint8_t*>( mine );
    const uint4 a = mine[q4];
    const uint32_t x = bytes[ii];
    const uint32_t below = bytes[q4 != 0 ? 16 * q4 - 1 : 0];
    const uint4 sel = perm_rows[ii & 15u];
    uint32_t carry = x << 24;   /* the byte that moves into the next group sits in the top byte */
    for ( uint32_t k = 0; k < q4; ++k ) {
        const uint4 g = mine[k];
        uint4 m;
        m.x = __builtin_amdgcn_alignbit( g.x, carry, 24 );
        m.y = __builtin_amdgcn_alignbit( g.y, g.x, 24 );
        m.z = __builtin_amdgcn_alignbit( g.z, g.y, 24 );
        m.w = __builtin_amdgcn_alignbit( g.w, g.z, 24 );
        carry = g.w;
        mine[k] = m;
    }
    const uint32_t in = q4 != 0 ? below : x;
    uint4 n;
    n.x = __builtin_amdgcn_perm( a.x, in, sel.x );
    n.y = __builtin_amdgcn_perm( a.y, a.x, sel.y );
    n.z = __builtin_amdgcn_perm( a.z, a.y, sel.z );
    n.w = __builtin_amdgcn_perm( a.w, a.z, sel.w );
    mine[q4] = n;
    return x;
}

/** Calls step( symbol ) for symbols [begin, end) of a lane's chunk, eight symbols per 16-byte load with the extraction
 * unrolled; the next load is in flight while a group is processed.
 * step returns false to stop early. */
template<typename Step>
__device__ __forceinline__ void
for_symbols( const uint16_t* sym, uint32_t begin, uint32_t end, Step&& step )
{
    uint32_t i = begin;
    for ( ; i < end && ( i & 7u ) != 0; ++i ) {
        if ( !step( (uint32_t)sym[i] ) ) return;
    }
    if ( i + 8 <= end ) {
        uint4 next = *reinterpret_cast<const uint4*>( sym + i );
        for ( ; i + 8 <= end; i += 8 ) {
            const uint4 v = next;
            next = *reinterpret_cast<const uint4*>( sym + i + 8 );   /* the symbol buffer is padded past n */
            if ( !step( v.x & 0xFFFFu ) ) return;
            if ( !step( v.x >> 16 ) ) return;
            if ( !step( v.y & 0xFFFFu ) ) return;
            if ( !step( v.y >> 16 ) ) return;
            if ( !step( v.z & 0xFFFFu ) ) return;
            if ( !step( v.z >> 16 ) ) return;
            if ( !step( v.w & 0xFFFFu ) ) return;
            if ( !step( v.w >> 16 ) ) return;
        }
    }
    for ( ; i < end; ++i ) {
        if ( !step( (uint32_t)sym[i] ) ) return;
    }
}

/** The same walk over symbols [begin, end), but every symbol is REPLACED by what step( symbol ) returns (a u16): the groups of
 * eight go back as one 16-byte store. */
template<typename Step>
__device__ __forceinline__ void
rewrite_symbols( uint16_t* sym, uint32_t begin, uint32_t end, Step&& step )
{
    uint32_t i = begin;
    for ( ; i < end && ( i & 7u ) != 0; ++i ) sym[i] = (uint16_t)step( (uint32_t)sym[i] );
    if ( i + 8 <= end ) {
        uint4 next = *reinterpret_cast<const uint4*>( sym + i );
        for ( ; i + 8 <= end; i += 8 ) {
            const uint4 v = next;
            next = *reinterpret_cast<const uint4*>( sym + i + 8 );   /* the symbol buffer is padded past n */
            uint4 r;
            r.x = step( v.x & 0xFFFFu );
            r.x |= step( v.x >> 16 ) << 16;
            r.y = step( v.y & 0xFFFFu );
            r.y |= step( v.y >> 16 ) << 16;
            r.z = step( v.z & 0xFFFFu );
            r.z |= step( v.z >> 16 ) << 16;
            r.w = step( v.w & 0xFFFFu );
            r.w |= step( v.w >> 16 ) << 16;
            *reinterpret_cast<uint4*>( sym + i ) = r;
        }
    }
    for ( ; i < end; ++i ) sym[i] = (uint16_t)step( (uint32_t)sym[i] );
}

/** Per-lane write combiner for a sequential byte stream: whole aligned 16-byte units go out as one store.  The lanes of a
 * wave write 3.5 KB apart, so every store is a partial line of its own: byte stores cost 43 GB of fabric writes for 2.3 GB of
 * L column (PMC WRITE_SIZE), dword stores 10.3 GB, 16-byte units a third of that.  Only the unaligned head and the tail of a
 * lane's range, which share a unit with the neighbouring lane, are written bytewise.
 * Runs are what this is tuned for: half the symbols of a text block are run digits, most runs are shorter than a dword, and
 * with a loop of single bytes per run the expansion was a third of k_mtf (3.7 of 11 ms per instance for the bench's batch,
 * profiles/r03_mtf_probe.txt): fill() completes the current dword and starts the last one with two masked ORs, its only loops
 * are over the whole dwords up to the next unit (at most three) and over the whole units of a long run.  (The first 16-byte
 * form of round 3 aligned every run to a unit byte by byte: 14.5 instead of 8.9 ms.) */
struct ByteSink
{
    uint8_t* base;
    uint32_t lo;    /* first byte position of this lane's range */
    uint32_t o;     /* next byte position */
    uint32_t acc;   /* bytes of the dword that contains o, at their place */
    uint32_t w0, w1, w2;   /* the complete dwords in front of it in the 16-byte unit that contains o */

    /** bytes [from, to) of the unit that holds them all ({a, b, c, d} = its dwords) */
    __device__ __forceinline__ void
    bytes_out( uint32_t from, uint32_t to, uint32_t a, uint32_t b, uint32_t c, uint32_t d )
    {
        for ( uint32_t k = from; k < to; ++k ) {
            const uint32_t q = ( k >> 2 ) & 3u;
            const uint32_t word = q == 0 ? a : ( q == 1 ? b : ( q == 2 ? c : d ) );
            base[k] = (uint8_t)( word >> ( 8 * ( k & 3u ) ) );
        }
    }

    /** o is a multiple of 4: `acc` is the dword that ends there */
    __device__ __forceinline__ void
    word_done()
    {
        const uint32_t q = ( ( o - 4 ) >> 2 ) & 3u;
        if ( q == 3 ) {
            if ( o >= 16 && o - 16 >= lo ) {
                *reinterpret_cast<uint4*>( base + o - 16 ) = make_uint4( w0, w1, w2, acc );
            } else {
                bytes_out( lo, o, w0, w1, w2, acc );   /* the unit in which the range starts belongs to the previous lane as well */
            }
            w0 = w1 = w2 = 0;
        } else {
            w0 = q == 0 ? acc : w0;
            w1 = q == 1 ? acc : w1;
            w2 = q == 2 ? acc : w2;
        }
        acc = 0;
    }

    __device__ __forceinline__ void
    put( uint32_t byte )
    {
        acc |= byte << ( 8 * ( o & 3u ) );
        ++o;
        if ( ( o & 3u ) == 0 ) word_done();
    }

    __device__ __forceinline__ void
    fill( uint32_t byte, uint32_t count )
    {
        const uint32_t word = byte * 0x01010101u;
        const uint32_t n = o & 3u;
        if ( n != 0 ) {
            /* the rest of the current dword, or as much of it as the run has: 1 to 3 bytes */
            const uint32_t take = count < 4 - n ? count : 4 - n;
            acc |= ( word & ( ( 1u << ( 8 * take ) ) - 1u ) ) << ( 8 * n );
            o += take;
            count -= take;
            if ( ( o & 3u ) == 0 ) word_done();
        }
        /* (whatever is left starts a dword) whole dwords up to the next unit: at most three */
        while ( count >= 4 && ( o & 15u ) != 0 ) {
            acc = word;
            o += 4;
            count -= 4;
            word_done();
        }
        /* whole units */
        if ( count >= 16 ) {
            const uint4 unit = make_uint4( word, word, word, word );
            for ( uint32_t k = count >> 4; k != 0; --k ) {
                *reinterpret_cast<uint4*>( base + o ) = unit;     /* (o >= lo: inside the range) */
                o += 16;
            }
            count &= 15u;
        }
        /* whole dwords of the last unit, then the bytes that start its last dword */
        while ( count >= 4 ) {
            acc = word;
            o += 4;
            count -= 4;
            word_done();
        }
        acc |= word & ( ( 1u << ( 8 * count ) ) - 1u );
        o += count;
    }

    __device__ __forceinline__ void
    flush()
    {
        /* the last, incomplete unit (shared with the next lane) */
        const uint32_t unitStart = o & ~15u;
        const uint32_t first = unitStart > lo ? unitStart : lo;
        const uint32_t q = ( o >> 2 ) & 3u;     /* the dword `acc` stands for */
        bytes_out( first, o, q == 0 ? acc : w0, q == 1 ? acc : w1, q == 2 ? acc : w2, acc );
        acc = w0 = w1 = w2 = 0;
    }
};

/** THREADS = lanes = chunks of the symbol stream: 256, or 512 for small batches, whose blocks should be through quickly (a
 * lane's two passes are half as long; 139 / 74 KB of LDS: one or two blocks per CU).
 * LANE_STRIDE = bytes between the lists of consecutive lanes = list capacity + 16.  Two instances: 272 (any block) and
 * MTF_SMALL_STRIDE (blocks that use at most MTF_SMALL_STRIDE - 16 symbols: text), whose 37 KB of LDS let four
 * workgroups share a CU instead of two -- the kernel is bound by instruction issue at two waves per SIMD.  Both are
 * launched over all blocks, a workgroup whose block belongs to the other instance returns at once.  Both strides keep
 * the 16-byte accesses of 16 lanes on 64 distinct banks (stride / 4 mod 64 is an odd multiple of 4). */
template<uint32_t LANE_STRIDE, uint32_t THREADS>
struct alignas( 16 ) MtfShared
{
    uint8_t listBytes[THREADS * LANE_STRIDE];   /* 68 / 36 KiB with 256 lanes */
    uint4 permRows[16];
    uint8_t cur[256];
    uint32_t starts[THREADS + 1];
    unsigned long long waveTotals[THREADS / 64];
    uint32_t firstError;
};

/* The body of k_mtf.  Its LDS arrays arrive as __restrict__ parameters: they are pieces of ONE launch-time allocation (see
 * k_mtf), and as plain pointers into it they could alias each other for all the compiler knows. */
template<uint32_t LANE_STRIDE, uint32_t THREADS>
__device__ __forceinline__ void
mtf_block( BlockMeta* __restrict__       meta,
           const HuffMeta* __restrict__  hmeta,
           uint16_t* __restrict__        sym_buf,
           const uint8_t* __restrict__   stb_buf,
           uint8_t* __restrict__         l_buf,
           uint32_t                      n_blocks,
           const uint32_t* __restrict__  order,
           uint8_t* __restrict__         listBytes,
           uint4* __restrict__           permRows,
           uint8_t* __restrict__         cur,
           uint32_t* __restrict__        starts,
           unsigned long long* __restrict__ waveTotals,
           uint32_t* __restrict__        firstErrorAt )
{
    constexpr uint32_t LIST_ENTRIES = LANE_STRIDE - 16;
    /* How pass B gets its bytes.  Lists of up to 128 entries (text): the moves are replayed on the true start list -- they are
     * 0.1 ms of a pass there.  Lists of 256 entries (binary and incompressible data, whose moves shift 128 bytes of a lane's
     * list on average: the kernel is bound by the LDS traffic of that, 43 ms for 2 620 incompressible blocks): pass A leaves
     * behind, in place of every symbol, the POSITION its entry has in the chunk's start list (the list starts as the identity,
     * so that is what a move brings to the front), and pass B only looks positions up -- half the list traffic for 1.8 MB of
     * symbols written back per block. */
    constexpr bool REPLAY = LANE_STRIDE == MTF_SMALL_STRIDE;
    uint32_t& firstError = *firstErrorAt;

    const uint32_t slot = blockIdx.x;
    if ( slot >= n_blocks ) return;
    const uint32_t b = order[slot];
    const HuffMeta hm = hmeta[b];
    if ( !hm.active ) return;
    if ( ( hm.symbol_count <= MTF_SMALL_STRIDE - 16 ) != ( LANE_STRIDE == MTF_SMALL_STRIDE ) ) return;   /* other instance */
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t n = hm.n_stored;
    uint16_t* const sym = sym_buf + (size_t)b * SYM_STRIDE;
    uint8_t* const L = l_buf + (size_t)b * L_STRIDE;
    uint4* const mine = reinterpret_cast<uint4*>( listBytes + t * LANE_STRIDE );

    /* chunk boundaries: never inside a RUNA/RUNB digit sequence */
    const uint32_t S = ( n + THREADS - 1 ) / THREADS;
    uint32_t begin = t * S < n ? t * S : n;
    while ( begin < n && begin > 0 && sym[begin] <= 1 && sym[begin - 1] <= 1 ) ++begin;
    starts[t] = begin;
    if ( t == 0 ) { starts[THREADS] = n; firstError = 0xFFFFFFFFu; }
    if ( t < 256 ) cur[t] = stb_buf[(size_t)b * 256 + t];
    if ( t < 16 ) permRows[t] = mtf_perm_row( t );
    for ( uint32_t k = 0; k < LIST_ENTRIES / 16; ++k ) {
        const uint32_t e = 16 * k;
        mine[k] = make_uint4( ( e ) | ( ( e + 1 ) << 8 ) | ( ( e + 2 ) << 16 ) | ( ( e + 3 ) << 24 ),
                              ( e + 4 ) | ( ( e + 5 ) << 8 ) | ( ( e + 6 ) << 16 ) | ( ( e + 7 ) << 24 ),
                              ( e + 8 ) | ( ( e + 9 ) << 8 ) | ( ( e + 10 ) << 16 ) | ( ( e + 11 ) << 24 ),
                              ( e + 12 ) | ( ( e + 13 ) << 8 ) | ( ( e + 14 ) << 16 ) | ( ( e + 15 ) << 24 ) );
    }
    __syncthreads();
    const uint32_t end = starts[t + 1];

    /* ---- pass A: chunk permutation and output size ---- */
    unsigned long long count = 0;
    {
        uint32_t runPos = 0, hh = 0;
        const auto one = [&] ( uint32_t s ) -> uint32_t {
            if ( s <= 1 ) {
                if ( runPos == 0 ) { runPos = 1; hh = 0; }
                hh += runPos << s;
                runPos <<= 1;
                return s;
            }
            if ( runPos != 0 ) { count += hh; runPos = 0; }
            ++count;
            return mtf_lane_move( mine, permRows, s - 1 ) + 2u;    /* (run digits stay 0 and 1) */
        };
        if constexpr ( REPLAY ) {
            for_symbols( sym, begin, end, [&] ( uint32_t s ) { (void)one( s ); return true; } );
        } else {
            rewrite_symbols( sym, begin, end, one );
        }
        if ( runPos != 0 ) count += hh;
    }
    /* exclusive prefix sum of the chunk sizes */
    unsigned long long incl = count;
    for ( int d = 1; d < 64; d <<= 1 ) {
        const unsigned long long o = __shfl_up( incl, d );
        if ( (int)lane >= d ) incl += o;
    }
    if ( lane == 63 ) waveTotals[wave] = incl;
    __syncthreads();
    unsigned long long prefix = incl - count, total = 0;
    for ( uint32_t w = 0; w < THREADS / 64; ++w ) {
        if ( w < wave ) prefix += waveTotals[w];
        total += waveTotals[w];
    }

    /* ---- compose the chunk permutations in order: lane c's list becomes the list valid at the start of chunk c ---- */
    for ( uint32_t c = 0; c < THREADS; ++c ) {
        /* entries beyond the list capacity are never moved (symbols that do not occur): thread t has nothing to do */
        uint8_t* const slotC = listBytes + c * LANE_STRIDE;
        const bool mineToDo = t < LIST_ENTRIES;
        const uint8_t v = mineToDo ? cur[slotC[t]] : 0;   /* entry t after chunk c = old entry at the permuted position */
        const uint8_t o = mineToDo ? cur[t] : 0;
        __syncthreads();
        if ( mineToDo ) {
            slotC[t] = o;
            cur[t] = v;
        }
        __syncthreads();
    }

    /* ---- pass B: the chunk again with its true start list, the L column written ---- */
    {
        /* positions are 32-bit: a start beyond the buffer (only possible for damaged data, whose runs can add up to
         * anything) is clamped -- that lane then reports the overflow at its first symbol, an earlier lane wins anyway */
        const uint32_t startAt = prefix < MAX_N ? (uint32_t)prefix : MAX_N;
        ByteSink sink{ L, startAt, startAt, 0, 0, 0, 0 };
        const uint8_t* const list = reinterpret_cast<const uint8_t*>( mine );
        uint32_t front = 0;       /* !REPLAY: position (in the start list) of the entry that is at the front now */
        uint32_t runPos = 0, hh = 0;
        uint32_t err = 0;
        for_symbols( sym, begin, end, [&] ( uint32_t s ) {
            if ( s <= 1 ) {
                if ( runPos == 0 ) { runPos = 1; hh = 0; }
                hh += runPos << s;
                runPos <<= 1;
                return true;
            }
            if ( runPos != 0 ) {
                runPos = 0;
                if ( sink.o + hh > MAX_N ) { err = ST_RUN_OVERFLOW; return false; }
                sink.fill( list[REPLAY ? 0u : front], hh );
            }
            if ( sink.o >= MAX_N ) { err = ST_DATA_OVERFLOW; return false; }
            if constexpr ( REPLAY ) {
                sink.put( mtf_lane_move( mine, permRows, s - 1 ) );
            } else {
                front = s - 2u;
                sink.put( list[front] );
            }
            return true;
        } );
        /* a run that is still open where the Huffman stage FAILED is never flushed by the reference */
        if ( err == 0 && runPos != 0 && ( end < n || hm.status == ST_OK ) ) {
            if ( sink.o + hh > MAX_N ) {
                err = ST_RUN_OVERFLOW;
            } else {
                sink.fill( list[REPLAY ? 0u : front], hh );
            }
        }
        sink.flush();
        if ( err != 0 ) atomicMin( &firstError, ( t << 8 ) | err );
    }
    __syncthreads();
    if ( t == 0 ) {
        /* The first failure in symbol order wins, as in the sequential reference: an overflow found here lies before
         * the point where the Huffman stage stopped. */
        int32_t status = hm.status;
        if ( firstError != 0xFFFFFFFFu ) status = (int32_t)( firstError & 0xFFu );
        const uint32_t N = status == ST_OK ? (uint32_t)total : 0u;
        const uint32_t origPtr = meta[b].orig_ptr;
        if ( status == ST_OK && origPtr >= N ) status = ST_ORIGPTR_DATA;
        meta[b].n = status == ST_OK || status == ST_ORIGPTR_DATA ? (uint32_t)total : 0u;
        meta[b].status = status;
        /* walk segments: up to KMAX of them, a table entry in every `stride` a segment start */
        uint32_t stride = ( N + KMAX - 1 ) / KMAX;
        if ( stride < MIN_SEG_STRIDE ) stride = MIN_SEG_STRIDE;
        const uint32_t k0 = ( N + stride - 1 ) / stride;
        meta[b].seg_stride = stride;
        meta[b].nseg = k0 + ( ( N > 0 && origPtr % stride != 0 ) ? 1u : 0u );
        meta[b].walk_ok = ( status == ST_OK && N > 0 ) ? 1u : 0u;
    }
}

/* W: wavefronts per SIMD the registers leave room for, LDS declared at launch -- see k_hscan (bz2_hscan.hip.h). */
template<uint32_t LANE_STRIDE, uint32_t THREADS = MTF_THREADS, uint32_t W = 2>
__global__ __launch_bounds__( THREADS ) __attribute__( ( amdgpu_waves_per_eu( W, 8 ) ) ) void
k_mtf( BlockMeta* __restrict__       meta,
       const HuffMeta* __restrict__  hmeta,
       uint16_t* __restrict__        sym_buf,       /* the <272> instance rewrites it in place, see mtf_block */
       const uint8_t* __restrict__   stb_buf,
       uint8_t* __restrict__         l_buf,
       uint32_t                      n_blocks,
       const uint32_t* __restrict__  order )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t ldsAtLaunch[];     /* sizeof( MtfShared<LANE_STRIDE, THREADS> ) */
    auto& shared = *reinterpret_cast<MtfShared<LANE_STRIDE, THREADS>*>( ldsAtLaunch );
    mtf_block<LANE_STRIDE, THREADS>( meta, hmeta, sym_buf, stb_buf, l_buf, n_blocks, order, shared.listBytes, shared.permRows,
                                     shared.cur, shared.starts, shared.waveTotals, &shared.firstError );
}
}  // namespace bz2gpu
