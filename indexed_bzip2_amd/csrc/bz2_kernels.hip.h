/**
 * bz2_kernels.hip.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the bzip2 block-decode hot path.
 *
 * Pipeline for one batch of independent bzip2 blocks (reference: one BZ2BlockFetcher::decodeBlock call per block,
 * src/indexed_bzip2/BZ2BlockFetcher.hpp:85-138):
 *
 *   k_hscan, k_hsym (bz2_hscan.hip.h), k_mtf (bz2_stage1.hip.h)  header, tables, Huffman, RUNA/RUNB, MTF -> L column (u8[N])   bzip2.hpp:479-807
 *   k_bwt_build   per block: byte histogram, stable ranks -> packed LF table u32[N] = LF<<8 | byte | MARK
 *                 (coalesced writes; replaces the scatter of prepare(), bzip2.hpp:810-847)
 *   k_walk, k_link2, k_emit (bz2_walk.hip.h)  multi-segment form of the N-step walk  bzip2.hpp:872-879
 *   k_rle<false>  RLE1 as a 5-state scan: decoded size D per block                      bzip2.hpp:881-896
 *   k_rle<true>   expansion into the batch output buffer
 *   k_crc         bzip2 CRC-32 of the D bytes by chunk CRCs + GF(2) shift-combine        bzip2.hpp:59-91, 900-907
 *
 * All arithmetic is integer/byte work bounded by HBM traffic and load latency; there is no MFMA anywhere.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bz2gpu
{
constexpr uint32_t MAX_N = 900000;            /* bzip2.hpp:416 dbuf size */
constexpr uint32_t L_STRIDE = 900096;         /* bytes per block in the L and R buffers (multiple of 256) */
constexpr uint32_t SEL_STRIDE = 32768;        /* bzip2.hpp:451 */
constexpr uint32_t TAB_STRIDE = 1u << 20;     /* u32 entries per block: every 20-bit index stays in bounds */
constexpr uint32_t KMAX = 32768;              /* max regular walk segments per block (+1 for origPtr) */
constexpr uint32_t SEG_STRIDE = KMAX + 64;
constexpr uint32_t MIN_SEG_STRIDE = 16;
constexpr uint32_t MARK = 0x80000000u;
constexpr uint32_t LF_MASK = 0xFFFFFu;
constexpr uint32_t INVALID_OFF = 0xFFFFFFFFu;

enum Status : int32_t {
    ST_OK = 0, ST_EOF = 1, ST_BAD_MAGIC = 2, ST_RANDOMIZED = 3, ST_ORIGPTR_RANGE = 4, ST_GROUP_COUNT = 5,
    ST_SELECTOR_COUNT = 6, ST_SELECTOR_UNARY = 7, ST_CODE_LENGTH = 8, ST_HUFFMAN_LENGTHS = 9,
    ST_SELECTOR_OVERRUN = 10, ST_INVALID_CODE = 11, ST_RUN_OVERFLOW = 12, ST_DATA_OVERFLOW = 13,
    ST_ORIGPTR_DATA = 14, ST_CRC = 15
};

/** Device-side per-block record. */
struct BlockMeta
{
    uint64_t enc_off;
    uint64_t enc_size;
    uint64_t decoded_size;
    uint64_t out_off;
    uint32_t header_crc;
    uint32_t computed_crc;
    uint32_t n;          /* BWT length N */
    uint32_t orig_ptr;
    uint32_t nsym;
    int32_t  is_eos;
    int32_t  is_eof;
    int32_t  status;
    uint32_t seg_stride;
    uint32_t nseg;
    uint32_t walk_ok;    /* 1 if stages >= walk should run */
    uint32_t cycle_len;  /* length of the permutation cycle through origPtr (== N unless the block is periodic) */
    uint32_t nchain;     /* number of walk segments on that cycle (k_link2) */
    uint32_t pad;
};

struct CrcConsts
{
    uint32_t pow8[32];    /* x^(8*2^k) mod P */
    uint32_t ipow8[32];   /* x^(-8*2^k) mod P */
};

__device__ __forceinline__ uint32_t
sfl( uint32_t v )
{
    return __builtin_amdgcn_readfirstlane( v );
}

/** A 32-bit word of the resident input as 32 stream bits, most significant first: the file's bytes stay in file order
 * in HBM (no swapped copy), every reader swaps the word it loads (one v_perm_b32). */
__device__ __forceinline__ uint32_t
be32( uint32_t word )
{
    return __builtin_bswap32( word );
}

__device__ __forceinline__ uint32_t
lane_id()
{
    return threadIdx.x & 63u;
}

/* -------------------------------------------------------------------------------------------------------------
 * MSB-first bit reader over 32-bit words of the input, byte-swapped as they are loaded (reference: BitReader<true,uint64_t>,
 * src/core/BitReader.hpp:190-290, 469-476).  All state is wave-uniform and ends up in SGPRs.
 * ------------------------------------------------------------------------------------------------------------- */
struct BitRd
{
    const uint32_t* w;
    uint64_t nwords;
    uint64_t size_bits;
    uint32_t tail_mask;
    uint64_t buf;     /* valid bits left-aligned */
    uint32_t n;       /* number of valid bits in buf */
    uint64_t widx;    /* next word to load */
    uint64_t pos;     /* absolute bit position of buf's first bit */
    bool eof;

    __device__ __forceinline__ uint32_t
    loadw( uint64_t i ) const
    {
        uint32_t v = 0;
        if ( i < nwords ) {
            v = sfl( be32( w[i] ) );   /* the ctx keeps the file's bytes as they are, zero padded */
        }
        return v;
    }

    __device__ __forceinline__ void
    init( const uint32_t* words, uint64_t size_bytes, uint64_t bitpos )
    {
        w = words;
        nwords = ( size_bytes + 3 ) >> 2;
        size_bits = size_bytes * 8;
        const uint32_t r = (uint32_t)( size_bytes & 3 );
        tail_mask = r == 0 ? 0xFFFFFFFFu : ( 0xFFFFFFFFu << ( 32 - 8 * r ) );
        pos = bitpos;
        eof = false;
        widx = bitpos >> 5;
        const uint32_t sh = (uint32_t)( bitpos & 31 );
        const uint64_t hi = loadw( widx ), lo = loadw( widx + 1 );
        buf = ( ( hi << 32 ) | lo ) << sh;
        n = 64 - sh;
        widx += 2;
    }

    /** Guarantees n > 32 afterwards. */
    __device__ __forceinline__ void
    refill()
    {
        if ( n <= 32 ) {
            buf |= (uint64_t)loadw( widx ) << ( 32 - n );
            n += 32;
            ++widx;
        }
    }

    __device__ __forceinline__ uint32_t
    peek( uint32_t k ) const   /* 1 <= k <= 32, k <= n */
    {
        return (uint32_t)( buf >> ( 64 - k ) );
    }

    __device__ __forceinline__ void
    skip( uint32_t k )
    {
        buf <<= k;
        n -= k;
        pos += k;
    }

    /** read k <= 32 bits; sets eof (sticky) if the read crosses the end of the input. */
    __device__ __forceinline__ uint32_t
    read( uint32_t k )
    {
        refill();
        if ( pos + k > size_bits ) {
            eof = true;
        }
        const uint32_t v = peek( k );
        skip( k );
        return v;
    }
};

/* -------------------------------------------------------------------------------------------------------------
 * k_find_magic: every bit offset of a 48-bit pattern in the resident input (the host form is
 * mi355x_bz2_find_magic = BitStringFinder<48>, src/core/BitStringFinder.hpp:158-285).  One thread per 32 start
 * positions: three big-endian words give the 80 bits that the 32 candidate windows of a word need.  Matches are
 * rare (one per block), so they are appended through a single atomic counter and sorted on the host.
 * ------------------------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__( 256 ) void
k_find_magic( const uint32_t* __restrict__ words, uint64_t size_bits, uint64_t magic48,
              uint64_t* __restrict__ found, uint32_t capacity, uint32_t* __restrict__ counter )
{
    const uint64_t nStartWords = ( size_bits + 31 ) >> 5;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for ( uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < nStartWords; w += stride ) {
        const uint64_t hi = ( (uint64_t)be32( words[w] ) << 32 ) | be32( words[w + 1] );   /* input copy is zero padded */
        const uint64_t lo = (uint64_t)be32( words[w + 2] ) << 32;
#pragma unroll 4
        for ( uint32_t s = 0; s < 32; ++s ) {
            const uint64_t window = s == 0 ? hi : ( ( hi << s ) | ( lo >> ( 64 - s ) ) );
            if ( ( window >> 16 ) == magic48 ) {
                const uint64_t offset = w * 32 + s;
                if ( offset + 48 <= size_bits ) {
                    const uint32_t slot = atomicAdd( counter, 1u );
                    if ( slot < capacity ) found[slot] = offset;
                }
            }
        }
    }
}

/** Lanes whose `key` (low `bits` bits) equals this lane's, restricted to `valid` lanes. */
__device__ __forceinline__ uint64_t
match_any( uint32_t key, int bits, bool valid )
{
    /* per bit: one ballot, then every lane notes the peers whose bit DIFFERS from its own: ballot ^ -bit.  The notes of two
     * bits go into the collection with one three-input OR per half (5 vector instructions per bit; `bit ? ballot : ~ballot`
     * and an AND per bit cost 8, an xnor and an AND per bit 6) */
    const uint64_t all = __ballot( valid );
    uint32_t lo = 0, hi = 0;
    int b = 0;
    for ( ; b + 1 < bits; b += 2 ) {
        const uint32_t minus0 = (uint32_t)__builtin_amdgcn_sbfe( (int)key, b, 1 );       /* 0 or 0xFFFFFFFF */
        const uint32_t minus1 = (uint32_t)__builtin_amdgcn_sbfe( (int)key, b + 1, 1 );
        const uint64_t bal0 = __ballot( minus0 != 0 );
        const uint64_t bal1 = __ballot( minus1 != 0 );
        lo = lo | ( (uint32_t)bal0 ^ minus0 ) | ( (uint32_t)bal1 ^ minus1 );
        hi = hi | ( (uint32_t)( bal0 >> 32 ) ^ minus0 ) | ( (uint32_t)( bal1 >> 32 ) ^ minus1 );
    }
    if ( b < bits ) {
        const uint32_t minusBit = (uint32_t)__builtin_amdgcn_sbfe( (int)key, b, 1 );
        const uint64_t bal = __ballot( minusBit != 0 );
        lo |= (uint32_t)bal ^ minusBit;
        hi |= (uint32_t)( bal >> 32 ) ^ minusBit;
    }
    return all & ~( ( (uint64_t)hi << 32 ) | lo );
}

/** match_any for the bytes of a BWT's last column, which come in runs: if all valid lanes hold the same key (one
 * comparison and one ballot), that is the answer. */
__device__ __forceinline__ uint64_t
match_any_runs( uint32_t key, bool valid )
{
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane( (int)key );     /* (lane 0 is valid whenever any lane is) */
    if ( __ballot( valid && key != first ) == 0 ) return __ballot( valid );
    return match_any( key, 8, valid );
}

__device__ __forceinline__ uint32_t
popc_below( uint64_t mask, uint32_t lane )
{
    return __popcll( mask & ( ( 1ull << lane ) - 1ull ) );
}

/* =============================================================================================================
 * k_bwt_build: LF[i] = C[L[i]] + rank(L[i], i), written coalesced at i as (LF << 8) | L[i] | MARK.
 * The reference scatters i into dbuf[C[c]++] (the inverse permutation T) and walks forward (bzip2.hpp:817-831);
 * LF = T^-1 needs no scatter and is walked backwards from origPtr: out[N-1-k] = L[LF^k(origPtr)].
 * One workgroup of 1024 threads (16 waves) per block; wave w owns a contiguous 1/16 of the block.
 * ============================================================================================================= */
constexpr int BWT_WAVES = 16;

__global__ __launch_bounds__( 1024 ) void
k_bwt_build( const BlockMeta* __restrict__ meta,
             const uint8_t* __restrict__   l_buf,
             uint32_t* __restrict__        tab_buf )
{
    __shared__ uint32_t hist[BWT_WAVES][256];
    __shared__ uint32_t tot[256];
    const uint32_t b = blockIdx.x;
    const BlockMeta mt = meta[b];
    if ( !mt.walk_ok ) return;
    const uint32_t N = mt.n, origPtr = mt.orig_ptr, stride = mt.seg_stride;
    const uint8_t* const L = l_buf + (size_t)b * L_STRIDE;
    uint32_t* const tab = tab_buf + (size_t)b * TAB_STRIDE;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    uint32_t chunk = ( N + BWT_WAVES - 1 ) / BWT_WAVES;
    chunk = ( chunk + 255u ) & ~255u;
    const uint32_t begin = wave * chunk < N ? wave * chunk : N;
    const uint32_t end = begin + chunk < N ? begin + chunk : N;

    for ( uint32_t i = tid; i < BWT_WAVES * 256; i += 1024 ) ( &hist[0][0] )[i] = 0;
    __syncthreads();

    /* pass 1: per-wave byte histograms (4 bytes per lane per step; L buffer is padded to L_STRIDE) */
    for ( uint32_t base = begin; base < end; base += 256 ) {
        const uint32_t i = base + 4 * lane;
        if ( i < end ) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>( L + i );
            const uint32_t nb = end - i < 4 ? end - i : 4;
            for ( uint32_t k = 0; k < nb; ++k ) atomicAdd( &hist[wave][( v >> ( 8 * k ) ) & 0xFF], 1u );
        }
    }
    __syncthreads();
    if ( tid < 256 ) {
        uint32_t run = 0;
        for ( int w = 0; w < BWT_WAVES; ++w ) {
            const uint32_t t = hist[w][tid];
            hist[w][tid] = run;
            run += t;
        }
        tot[tid] = run;
    }
    __syncthreads();
    if ( tid < 64 ) {
        /* exclusive scan of 256 totals: 4 per lane + wave scan */
        uint32_t v0 = tot[4 * tid], v1 = tot[4 * tid + 1], v2 = tot[4 * tid + 2], v3 = tot[4 * tid + 3];
        const uint32_t s = v0 + v1 + v2 + v3;
        uint32_t incl = s;
        for ( int d = 1; d < 64; d <<= 1 ) {
            const uint32_t o = __shfl_up( incl, d );
            if ( (int)tid >= d ) incl += o;
        }
        uint32_t excl = incl - s;
        tot[4 * tid] = excl; excl += v0;
        tot[4 * tid + 1] = excl; excl += v1;
        tot[4 * tid + 2] = excl; excl += v2;
        tot[4 * tid + 3] = excl;
    }
    __syncthreads();
    if ( tid < 256 ) {
        const uint32_t c = tot[tid];
        for ( int w = 0; w < BWT_WAVES; ++w ) hist[w][tid] += c;
    }
    __syncthreads();

    /* pass 2: stable ranks, 64 positions per wave step; i % stride is kept incrementally (a division per position
     * would cost more than the ranking itself) */
    uint32_t rem = ( begin + lane ) % stride;
    const uint32_t remStep = 64u % stride;
    /* four steps per trip: their bytes are loaded together, so the ranking of one step hides the load latency of the
     * next ones instead of every step waiting for its own byte (the L buffer is padded: reading past `end` is fine) */
    for ( uint32_t base4 = begin; base4 < end; base4 += 256 ) {
        uint32_t keys[4];
#pragma unroll
        for ( uint32_t u = 0; u < 4; ++u ) keys[u] = L[base4 + 64 * u + lane];
#pragma unroll
        for ( uint32_t u = 0; u < 4; ++u ) {
            const uint32_t i = base4 + 64 * u + lane;
            const bool valid = i < end;
            const uint32_t key = valid ? keys[u] : 0u;
            const uint64_t same = match_any_runs( key, valid );
            const uint32_t rank = popc_below( same, lane );
            uint32_t basePos = 0;
            if ( valid ) basePos = hist[wave][key];
            if ( valid ) {
                const uint32_t lf = basePos + rank;
                const bool mark = ( rem == 0 ) || ( i == origPtr );
                tab[i] = ( lf << 8 ) | key | ( mark ? MARK : 0u );
                if ( rank == 0 ) hist[wave][key] = basePos + (uint32_t)__popcll( same );
            }
            rem += remStep;
            if ( rem >= stride ) rem -= stride;
        }
    }
}

/* -------------------------------------------------------------------------------------------------------------
 * The same table build for FEW blocks, a block spread over S workgroups (k_bwt_build runs ONE workgroup per block: a
 * millisecond for a lone block, sixteen waves over 900 000 bytes one row at a time).  k_bwt_count: workgroup (s, b) counts
 * the bytes of slice s of block b, per wavefront chunk (16 S chunks per block), into `counts`.  k_bwt_rank: workgroup
 * (s, b) turns the counts of all chunks in front of its own into the first rank of every byte value in each of its
 * chunks, then ranks its slice exactly as k_bwt_build's second pass does.
 * ------------------------------------------------------------------------------------------------------------- */
constexpr uint32_t BWT_SPLIT_MAX = 8;                                   /* most slices per block */
constexpr uint32_t BWT_COUNTS_PER_BLOCK = BWT_SPLIT_MAX * BWT_WAVES * 256;   /* u32 */

__device__ __forceinline__ void
bwt_chunk_range( uint32_t N, uint32_t slices, uint32_t slice, uint32_t wave, uint32_t& begin, uint32_t& end )
{
    uint32_t chunk = ( N + slices * BWT_WAVES - 1 ) / ( slices * BWT_WAVES );
    chunk = ( chunk + 255u ) & ~255u;
    const uint32_t index = slice * BWT_WAVES + wave;
    begin = index * chunk < N ? index * chunk : N;
    end = begin + chunk < N ? begin + chunk : N;
}

__global__ __launch_bounds__( 1024 ) void
k_bwt_count( const BlockMeta* __restrict__ meta,
             const uint8_t* __restrict__   l_buf,
             uint32_t* __restrict__        counts,     /* [block][slices * BWT_WAVES][256] */
             uint32_t                      slices )
{
    __shared__ uint32_t hist[BWT_WAVES][256];
    const uint32_t b = blockIdx.y, slice = blockIdx.x;
    const BlockMeta mt = meta[b];
    if ( !mt.walk_ok ) return;
    const uint8_t* const L = l_buf + (size_t)b * L_STRIDE;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t begin, end;
    bwt_chunk_range( mt.n, slices, slice, wave, begin, end );
    for ( uint32_t i = tid; i < BWT_WAVES * 256; i += 1024 ) ( &hist[0][0] )[i] = 0;
    __syncthreads();
    for ( uint32_t base = begin; base < end; base += 256 ) {
        const uint32_t i = base + 4 * lane;
        if ( i < end ) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>( L + i );
            const uint32_t nb = end - i < 4 ? end - i : 4;
            for ( uint32_t k = 0; k < nb; ++k ) atomicAdd( &hist[wave][( v >> ( 8 * k ) ) & 0xFF], 1u );
        }
    }
    __syncthreads();
    uint32_t* const out = counts + (size_t)b * BWT_COUNTS_PER_BLOCK + (size_t)slice * BWT_WAVES * 256;
    for ( uint32_t i = tid; i < BWT_WAVES * 256; i += 1024 ) out[i] = ( &hist[0][0] )[i];
}

__global__ __launch_bounds__( 1024 ) void
k_bwt_rank( const BlockMeta* __restrict__ meta,
            const uint8_t* __restrict__   l_buf,
            uint32_t* __restrict__        tab_buf,
            const uint32_t* __restrict__  counts,
            uint32_t                      slices )
{
    __shared__ uint32_t hist[BWT_WAVES][256];    /* first rank of every byte value in this workgroup's chunks */
    __shared__ uint32_t tot[256];
    const uint32_t b = blockIdx.y, slice = blockIdx.x;
    const BlockMeta mt = meta[b];
    if ( !mt.walk_ok ) return;
    const uint32_t N = mt.n, origPtr = mt.orig_ptr, stride = mt.seg_stride;
    const uint8_t* const L = l_buf + (size_t)b * L_STRIDE;
    uint32_t* const tab = tab_buf + (size_t)b * TAB_STRIDE;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if ( tid < 256 ) {
        /* byte value `tid`: its count in the chunks in front of this slice, in this slice's chunks, and in all */
        const uint32_t* const of = counts + (size_t)b * BWT_COUNTS_PER_BLOCK + tid;
        uint32_t run = 0;
        for ( uint32_t c = 0; c < slices * BWT_WAVES; ++c ) {
            const uint32_t t = of[(size_t)c * 256];
            if ( c / BWT_WAVES == slice ) hist[c % BWT_WAVES][tid] = run;
            run += t;
        }
        tot[tid] = run;
    }
    __syncthreads();
    if ( tid < 64 ) {
        /* exclusive scan of the 256 totals: 4 per lane + wave scan */
        uint32_t v0 = tot[4 * tid], v1 = tot[4 * tid + 1], v2 = tot[4 * tid + 2], v3 = tot[4 * tid + 3];
        const uint32_t s4 = v0 + v1 + v2 + v3;
        uint32_t incl = s4;
        for ( int d = 1; d < 64; d <<= 1 ) {
            const uint32_t o = __shfl_up( incl, d );
            if ( (int)tid >= d ) incl += o;
        }
        uint32_t excl = incl - s4;
        tot[4 * tid] = excl; excl += v0;
        tot[4 * tid + 1] = excl; excl += v1;
        tot[4 * tid + 2] = excl; excl += v2;
        tot[4 * tid + 3] = excl;
    }
    __syncthreads();
    if ( tid < 256 ) {
        const uint32_t c = tot[tid];
        for ( int w = 0; w < BWT_WAVES; ++w ) hist[w][tid] += c;
    }
    __syncthreads();

    uint32_t begin, end;
    bwt_chunk_range( N, slices, slice, wave, begin, end );
    uint32_t rem = ( begin + lane ) % stride;
    const uint32_t remStep = 64u % stride;
    for ( uint32_t base4 = begin; base4 < end; base4 += 256 ) {
        uint32_t keys[4];
#pragma unroll
        for ( uint32_t u = 0; u < 4; ++u ) keys[u] = L[base4 + 64 * u + lane];
#pragma unroll
        for ( uint32_t u = 0; u < 4; ++u ) {
            const uint32_t i = base4 + 64 * u + lane;
            const bool valid = i < end;
            const uint32_t key = valid ? keys[u] : 0u;
            const uint64_t same = match_any_runs( key, valid );
            const uint32_t rank = popc_below( same, lane );
            uint32_t basePos = 0;
            if ( valid ) basePos = hist[wave][key];
            if ( valid ) {
                const uint32_t lf = basePos + rank;
                const bool mark = ( rem == 0 ) || ( i == origPtr );
                tab[i] = ( lf << 8 ) | key | ( mark ? MARK : 0u );
                if ( rank == 0 ) hist[wave][key] = basePos + (uint32_t)__popcll( same );
            }
            rem += remStep;
            if ( rem >= stride ) rem -= stride;
        }
    }
}

/** Blocks whose LF permutation does not have origPtr on an N-cycle (cycle length c < N; see k_link2): the reference's
 * forward walk over T = LF^-1 (bzip2.hpp:872-879) goes round that cycle for N steps, out[j] = X[j mod c].  The backward
 * walk here produced Y[k] = X[c-1-k] at R[N-1-k], k < c.  In terms of k that is R[N-1-k] = Y[(k - r) mod c] with
 * r = N mod c.  Periodic data (valid streams) always has c | N: r = 0 and the first period is simply repeated.  r != 0
 * only happens for damaged blocks (whose CRC then fails in the reference, too), but the bytes and therefore the
 * calculated CRC still have to be the reference's: the period is parked in `tmp_buf` (the block's stash, which k_emit has
 * read by now; not the L column: R lies in its memory) and laid out again with the shift. */
__global__ __launch_bounds__( 256 ) void
k_replicate( const BlockMeta* __restrict__ meta,
             uint8_t*                      r_buf,
             uint8_t*                      tmp_buf,
             size_t                        tmp_stride )    /* bytes per block in tmp_buf, at least L_STRIDE */
{
    const uint32_t b = blockIdx.x;
    const BlockMeta mt = meta[b];
    if ( !mt.walk_ok || mt.cycle_len >= mt.n || mt.cycle_len == 0 ) return;
    const uint32_t N = mt.n, c = mt.cycle_len;
    uint8_t* const R = r_buf + (size_t)b * L_STRIDE;
    const uint32_t r = N % c;
    if ( r == 0 ) {
        for ( uint32_t k = c + threadIdx.x; k < N; k += 256 ) {
            R[N - 1 - k] = R[N - 1 - ( k % c )];
        }
        return;
    }
    uint8_t* const Y = tmp_buf + (size_t)b * tmp_stride;
    for ( uint32_t k = threadIdx.x; k < c; k += 256 ) Y[k] = R[N - 1 - k];
    __syncthreads();
    for ( uint32_t k = threadIdx.x; k < N; k += 256 ) {
        R[N - 1 - k] = Y[( k % c + c - r ) % c];
    }
}

/* =============================================================================================================
 * k_rle: RLE1 expansion (bzip2.hpp:881-896) as a scan over a 5-state machine.
 * State k before byte i = number of equal data bytes ending at i-1 (1..4; 4: byte i is a repeat count) or 0 after
 * a count byte.  With eq_i = (b[i] == b[i-1]):  eq: k -> (k+1) mod 5;  !eq: k -> (k == 4 ? 0 : 1).
 * Functions {0..4}->{0..4} are packed 3 bits per entry and composed associatively, so the state before every
 * chunk is an exclusive scan.  Output sizes are a second (sum) scan.  One workgroup per block walks its tiles.
 * ============================================================================================================= */
constexpr uint32_t RLE_THREADS = 512;
constexpr uint32_t RLE_BYTES_PER_THREAD = 32;   /* two 16-byte loads; the per-byte loops are unrolled with a predicate so
                                                   that the bytes stay in registers (no dynamic indexing) */
constexpr uint32_t RLE_TILE = RLE_THREADS * RLE_BYTES_PER_THREAD;
constexpr uint32_t FN_IDENTITY = 0 | ( 1 << 3 ) | ( 2 << 6 ) | ( 3 << 9 ) | ( 4 << 12 );

__device__ __forceinline__ uint32_t
fn_apply( uint32_t f, uint32_t k )
{
    return ( f >> ( 3 * k ) ) & 7u;
}

/** (g after f)(k) = g(f(k)) */
__device__ __forceinline__ uint32_t
fn_compose( uint32_t f, uint32_t g )
{
    uint32_t h = 0;
#pragma unroll
    for ( int k = 0; k < 5; ++k ) h |= fn_apply( g, fn_apply( f, k ) ) << ( 3 * k );
    return h;
}

__device__ __forceinline__ uint32_t
rle_step( uint32_t k, bool eq )
{
    return eq ? ( k == 4 ? 0u : k + 1 ) : ( k == 4 ? 0u : 1u );
}

/* ---------------------------------------------------------------------------------------------------------------
 * k_offsets: where every block's bytes go in the ragged output -- the exclusive prefix sum of the decoded sizes IN
 * THE CALLER'S ORDER (slot_of[i] = slot of the i-th requested block).  On the device, so that the expansion and the CRC
 * can be queued behind the walk without a round trip to the host.  result[0] = total bytes, result[1] = 1 if they do
 * not fit into `capacity` (k_rle<true> and k_crc then do nothing and the host takes over with a larger buffer).
 * ------------------------------------------------------------------------------------------------------------- */
constexpr uint32_t OFFSETS_THREADS = 1024;

__global__ __launch_bounds__( OFFSETS_THREADS ) void
k_offsets( BlockMeta* meta, const uint32_t* __restrict__ slot_of, uint32_t n, uint64_t capacity, uint64_t* result )
{
    __shared__ uint64_t partial[OFFSETS_THREADS];
    const uint32_t t = threadIdx.x;
    const uint32_t per = ( n + OFFSETS_THREADS - 1 ) / OFFSETS_THREADS;
    const uint32_t lo = t * per < n ? t * per : n;
    const uint32_t hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for ( uint32_t i = lo; i < hi; ++i ) {
        const BlockMeta& m = meta[slot_of[i]];
        sum += m.walk_ok ? m.decoded_size : 0;
    }
    partial[t] = sum;
    __syncthreads();
    for ( uint32_t step = 1; step < OFFSETS_THREADS; step *= 2 ) {
        const uint64_t add = t >= step ? partial[t - step] : 0;
        __syncthreads();
        partial[t] += add;
        __syncthreads();
    }
    uint64_t off = partial[t] - sum;     /* exclusive */
    for ( uint32_t i = lo; i < hi; ++i ) {
        BlockMeta& m = meta[slot_of[i]];
        m.out_off = off;
        off += m.walk_ok ? m.decoded_size : 0;
    }
    if ( t == OFFSETS_THREADS - 1 ) {
        result[0] = partial[t];
        result[1] = partial[t] + 256 > capacity ? 1 : 0;
    }
}

template<bool WRITE>
__global__ __launch_bounds__( RLE_THREADS ) void
k_rle( BlockMeta*                   meta,
       const uint8_t* __restrict__  r_buf,
       uint8_t* __restrict__        out,
       const uint64_t*              skip = nullptr )   /* non-zero: the output does not fit (k_offsets), nothing is written */
{
    __shared__ uint32_t wfn[RLE_THREADS / 64];
    __shared__ uint64_t wsum[RLE_THREADS / 64];
    __shared__ uint32_t carryFn;     /* state before the tile (as a constant function value) */
    __shared__ uint64_t carrySum;
    /* Eight bytes at a time: fnTab[e] = transition function of eight bytes whose equal-to-predecessor bits are e;
     * cmTab[k][e] = which of them are repeat counts when the group is entered in state k (low 8 bits) and the state
     * behind it (bits 8-10).  A full 32-byte chunk then costs four look-ups and three compositions instead of 32 x 5
     * state steps. */
    __shared__ uint16_t fnTab[256];
    __shared__ uint16_t cmTab[5][256];
    const uint32_t b = blockIdx.x;
    const BlockMeta mt = meta[b];
    if ( !mt.walk_ok ) return;
    if ( WRITE && skip != nullptr && *skip != 0 ) return;
    if ( threadIdx.x < 256 ) {
        const uint32_t e = threadIdx.x;
        uint32_t t[5] = { 0, 1, 2, 3, 4 };
        for ( uint32_t q = 0; q < 8; ++q ) {
            const bool eq = ( e >> q ) & 1u;
            for ( uint32_t k = 0; k < 5; ++k ) t[k] = rle_step( t[k], eq );
        }
        fnTab[e] = (uint16_t)( t[0] | ( t[1] << 3 ) | ( t[2] << 6 ) | ( t[3] << 9 ) | ( t[4] << 12 ) );
        for ( uint32_t k = 0; k < 5; ++k ) {
            uint32_t state = k, counts = 0;
            for ( uint32_t q = 0; q < 8; ++q ) {
                if ( state == 4 ) counts |= 1u << q;
                state = rle_step( state, ( e >> q ) & 1u );
            }
            cmTab[k][e] = (uint16_t)( counts | ( state << 8 ) );
        }
    }
    __syncthreads();
    const uint32_t N = mt.n;
    const uint8_t* const R = r_buf + (size_t)b * L_STRIDE;
    uint8_t* const dst = WRITE ? out + mt.out_off : nullptr;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    uint32_t stateIn = 0;     /* state before the current tile */
    uint64_t sumIn = 0;       /* output bytes before the current tile */

    for ( uint32_t tile = 0; tile < N; tile += RLE_TILE ) {
        const uint32_t i0 = tile + tid * RLE_BYTES_PER_THREAD;
        uint32_t wv[RLE_BYTES_PER_THREAD / 4] = {};
        uint32_t prev = 0;
        uint32_t nb = 0;
        if ( i0 < N ) {
            nb = N - i0 < RLE_BYTES_PER_THREAD ? N - i0 : RLE_BYTES_PER_THREAD;
            const uint4 v = *reinterpret_cast<const uint4*>( R + i0 );   /* R is padded to L_STRIDE */
            const uint4 v2 = *reinterpret_cast<const uint4*>( R + i0 + 16 );
            wv[0] = v.x; wv[1] = v.y; wv[2] = v.z; wv[3] = v.w;
            wv[4] = v2.x; wv[5] = v2.y; wv[6] = v2.z; wv[7] = v2.w;
            prev = i0 > 0 ? R[i0 - 1] : 0x100u;
        }
        /* eq mask of my bytes */
        uint32_t eqMask = 0;
        const bool full = nb == RLE_BYTES_PER_THREAD;
        if ( full ) {
            /* four bytes per step: a byte of d is zero where the byte equals its predecessor */
            uint32_t before = prev << 24;   /* prev == 0x100 (no predecessor) turns into 0: handled below */
#pragma unroll
            for ( uint32_t j = 0; j < RLE_BYTES_PER_THREAD / 4; ++j ) {
                const uint32_t d = wv[j] ^ __builtin_amdgcn_alignbit( wv[j], before, 24 );
                const uint32_t nonZero = ( ( ( d & 0x7F7F7F7Fu ) + 0x7F7F7F7Fu ) | d ) & 0x80808080u;
                const uint32_t e = ( ~nonZero >> 7 ) & 0x01010101u;
                eqMask |= ( ( e * 0x01020408u ) >> 24 & 0xFu ) << ( 4 * j );
                before = wv[j];
            }
            if ( prev > 0xFFu ) eqMask &= ~1u;   /* first byte of the block has no predecessor */
        } else {
            uint32_t p = prev;
            _Pragma( "unroll" ) for ( uint32_t q = 0; q < RLE_BYTES_PER_THREAD; ++q ) { if ( q >= nb ) break;
                const uint32_t c = ( wv[q >> 2] >> ( 8 * ( q & 3 ) ) ) & 0xFFu;
                eqMask |= ( c == p ? 1u : 0u ) << q;
                p = c;
            }
        }
        /* my chunk's transition function: 5 tracks */
        uint32_t fn = FN_IDENTITY;
        if ( full ) {
            fn = fn_compose( fn_compose( fnTab[eqMask & 0xFFu], fnTab[( eqMask >> 8 ) & 0xFFu] ),
                             fn_compose( fnTab[( eqMask >> 16 ) & 0xFFu], fnTab[eqMask >> 24] ) );
        } else if ( nb > 0 ) {
            uint32_t t0 = 0, t1 = 1, t2 = 2, t3 = 3, t4 = 4;
            _Pragma( "unroll" ) for ( uint32_t q = 0; q < RLE_BYTES_PER_THREAD; ++q ) { if ( q >= nb ) break;
                const bool eq = ( eqMask >> q ) & 1u;
                t0 = rle_step( t0, eq ); t1 = rle_step( t1, eq ); t2 = rle_step( t2, eq );
                t3 = rle_step( t3, eq ); t4 = rle_step( t4, eq );
            }
            fn = t0 | ( t1 << 3 ) | ( t2 << 6 ) | ( t3 << 9 ) | ( t4 << 12 );
        }
        /* exclusive scan of functions: wave level */
        uint32_t incl = fn;
        for ( int d = 1; d < 64; d <<= 1 ) {
            const uint32_t o = __shfl_up( incl, d );
            if ( (int)lane >= d ) incl = fn_compose( o, incl );
        }
        if ( lane == 63 ) wfn[wave] = incl;
        __syncthreads();
        uint32_t pre = FN_IDENTITY;   /* composition of all earlier waves */
        for ( uint32_t w = 0; w < wave; ++w ) pre = fn_compose( pre, wfn[w] );
        uint32_t exclWave = __shfl_up( incl, 1 );
        if ( lane == 0 ) exclWave = FN_IDENTITY;
        const uint32_t before = fn_compose( pre, exclWave );
        uint32_t k = fn_apply( before, stateIn );

        /* output size of my chunk with the true entry state; remember per-byte class */
        uint32_t mySize = 0;
        uint32_t countMask = 0;
        if ( full ) {
            uint32_t r = cmTab[k][eqMask & 0xFFu];
            countMask = r & 0xFFu;
            r = cmTab[r >> 8][( eqMask >> 8 ) & 0xFFu];
            countMask |= ( r & 0xFFu ) << 8;
            r = cmTab[r >> 8][( eqMask >> 16 ) & 0xFFu];
            countMask |= ( r & 0xFFu ) << 16;
            r = cmTab[r >> 8][eqMask >> 24];
            countMask |= ( r & 0xFFu ) << 24;
            mySize = RLE_BYTES_PER_THREAD - (uint32_t)__popc( countMask );
            if ( countMask != 0 ) {
#pragma unroll
                for ( uint32_t j = 0; j < RLE_BYTES_PER_THREAD / 4; ++j ) {
                    const uint32_t four = ( countMask >> ( 4 * j ) ) & 0xFu;
                    if ( four != 0 ) {
#pragma unroll
                        for ( uint32_t z = 0; z < 4; ++z ) {
                            if ( ( four >> z ) & 1u ) mySize += ( wv[j] >> ( 8 * z ) ) & 0xFFu;
                        }
                    }
                }
            }
        } else {
            uint32_t kk = k;
            _Pragma( "unroll" ) for ( uint32_t q = 0; q < RLE_BYTES_PER_THREAD; ++q ) { if ( q >= nb ) break;
                const uint32_t c = ( wv[q >> 2] >> ( 8 * ( q & 3 ) ) ) & 0xFFu;
                if ( kk == 4 ) { mySize += c; countMask |= 1u << q; } else { mySize += 1; }
                kk = rle_step( kk, ( eqMask >> q ) & 1u );
            }
        }
        /* exclusive sum scan */
        uint32_t inclS = mySize;
        for ( int d = 1; d < 64; d <<= 1 ) {
            const uint32_t o = __shfl_up( inclS, d );
            if ( (int)lane >= d ) inclS += o;
        }
        if ( lane == 63 ) wsum[wave] = inclS;
        __syncthreads();
        uint64_t preS = 0, tileTotal = 0;
        for ( uint32_t w = 0; w < RLE_THREADS / 64; ++w ) {
            if ( w < wave ) preS += wsum[w];
            tileTotal += wsum[w];
        }
        const uint64_t myOff = sumIn + preS + ( inclS - mySize );

        if constexpr ( WRITE ) {
            if ( countMask == 0 && nb == RLE_BYTES_PER_THREAD ) {
                /* the common case: no repeat count among my bytes, they go out as they are.  Whole aligned dwords
                 * (funnel-shifted by the misalignment of the destination), bytes only at the two ends. */
                uint8_t* const d = dst + myOff;
                const uint32_t head = ( 4u - ( (uint32_t)myOff & 3u ) ) & 3u;
                for ( uint32_t z = 0; z < head; ++z ) d[z] = (uint8_t)( wv[0] >> ( 8 * z ) );
                uint32_t* const d32 = reinterpret_cast<uint32_t*>( d + head );
#pragma unroll
                for ( uint32_t j = 0; j + 1 < RLE_BYTES_PER_THREAD / 4; ++j ) {
                    d32[j] = __builtin_amdgcn_alignbyte( wv[j + 1], wv[j], head );
                }
                constexpr uint32_t LAST = RLE_BYTES_PER_THREAD / 4 - 1;
                if ( head == 0 ) {
                    d32[LAST] = wv[LAST];
                } else {
                    for ( uint32_t z = head; z < 4; ++z ) d[4 * LAST + z] = (uint8_t)( wv[LAST] >> ( 8 * z ) );
                }
            } else {
            uint64_t o = myOff;
            uint32_t p = prev;
            _Pragma( "unroll" ) for ( uint32_t q = 0; q < RLE_BYTES_PER_THREAD; ++q ) { if ( q >= nb ) break;
                const uint32_t c = ( wv[q >> 2] >> ( 8 * ( q & 3 ) ) ) & 0xFFu;
                if ( ( countMask >> q ) & 1u ) {
                    for ( uint32_t z = 0; z < c; ++z ) dst[o + z] = (uint8_t)p;
                    o += c;
                } else {
                    dst[o++] = (uint8_t)c;
                }
                p = c;
            }
            }
        }

        /* carry to the next tile: the last thread's exit state */
        if ( tid == RLE_THREADS - 1 ) {
            carryFn = fn_apply( fn_compose( before, fn ), stateIn );
        }
        __syncthreads();
        stateIn = carryFn;
        sumIn += tileTotal;
        __syncthreads();
    }
    if ( !WRITE && tid == 0 ) {
        meta[b].decoded_size = sumIn;
    }
    (void)carrySum;
}

/* =============================================================================================================
 * k_crc: bzip2's MSB-first CRC-32 (poly 0x04C11DB7, bzip2.hpp:59-91) of each block's D output bytes.
 * Thread t CRCs a 64-byte chunk (zero init); chunks are combined with crc(A|B) = crc(A)*x^(8|B|) + crc(B) in
 * GF(2)[x]/P.  Tiles are aligned to absolute 64-byte addresses; leading pad bytes are zeros (no effect on a
 * zero-init CRC), trailing pad is undone with x^(-8*pad).  Init/final XOR are applied at the end.
 * ============================================================================================================= */
constexpr uint32_t CRC_THREADS = 256;
constexpr uint32_t CRC_CHUNK_LOG2 = 8;
constexpr uint32_t CRC_CHUNK = 1u << CRC_CHUNK_LOG2;     /* bytes per lane and tile: the combine tree (6 + 5 GF(2) multiplications
                                                             per tile) is the expensive part, so few, large tiles */
constexpr uint32_t CRC_TILE = CRC_THREADS * CRC_CHUNK;   /* 64 KiB */
constexpr uint32_t CRC_POLY = 0x04C11DB7u;

__device__ __forceinline__ uint32_t
gf_mul( uint32_t a, uint32_t b )
{
    /* (a * b) mod P, bit 31 = x^31 */
    uint32_t r = 0;
#pragma unroll 8
    for ( int i = 31; i >= 0; --i ) {
        r = ( r << 1 ) ^ ( ( r >> 31 ) ? CRC_POLY : 0u );
        if ( ( a >> i ) & 1u ) r ^= b;
    }
    return r;
}

/** x^(8*nbytes) (or its inverse) by binary decomposition over the precomputed table. */
__device__ __forceinline__ uint32_t
gf_pow8( const uint32_t* table, uint64_t nbytes )
{
    uint32_t r = 0x00000001u;   /* the polynomial 1 */
    for ( int k = 0; nbytes != 0 && k < 32; ++k, nbytes >>= 1 ) {
        if ( nbytes & 1u ) r = gf_mul( r, table[k] );
    }
    return r;
}

__global__ __launch_bounds__( CRC_THREADS ) void
k_crc( BlockMeta*                  meta,
       const uint8_t* __restrict__ out,
       CrcConsts                   cc,
       const uint64_t*             skip = nullptr )
{
    __shared__ uint32_t table[256];
    __shared__ uint32_t wcrc[CRC_THREADS / 64];
    const uint32_t b = blockIdx.x;
    const BlockMeta mt = meta[b];
    if ( !mt.walk_ok ) return;
    if ( skip != nullptr && *skip != 0 ) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    {
        uint32_t c = tid << 24;
        for ( int j = 0; j < 8; ++j ) c = ( c & 0x80000000u ) ? ( c << 1 ) ^ CRC_POLY : ( c << 1 );
        table[tid] = c;
    }
    __syncthreads();

    const uint64_t D = mt.decoded_size;
    const uint64_t a0 = mt.out_off;           /* absolute byte offsets in `out` */
    const uint64_t a1 = a0 + D;
    const uint64_t t0 = a0 & ~(uint64_t)( CRC_TILE - 1 );
    uint32_t running = 0;
    uint64_t tEnd = t0;
    const bool aligned = ( reinterpret_cast<uintptr_t>( out ) & 15u ) == 0;
    for ( uint64_t t = t0; t < a1; t += CRC_TILE ) {
        const uint64_t c0 = t + (uint64_t)tid * CRC_CHUNK;
        uint32_t crc = 0;
        if ( c0 + CRC_CHUNK > a0 && c0 < a1 ) {
            if ( aligned && c0 >= a0 && c0 + CRC_CHUNK <= a1 ) {
                const uint4* src = reinterpret_cast<const uint4*>( out + c0 );
#pragma unroll 4
                for ( uint32_t v = 0; v < CRC_CHUNK / 16; ++v ) {
                    const uint4 d = src[v];
                    const uint32_t ws[4] = { d.x, d.y, d.z, d.w };
#pragma unroll
                    for ( int q = 0; q < 4; ++q ) {
#pragma unroll
                        for ( int z = 0; z < 4; ++z ) {
                            const uint32_t byte = ( ws[q] >> ( 8 * z ) ) & 0xFFu;
                            crc = ( crc << 8 ) ^ table[( crc >> 24 ) ^ byte];
                        }
                    }
                }
            } else {
                for ( uint32_t q = 0; q < CRC_CHUNK; ++q ) {
                    const uint64_t a = c0 + q;
                    const uint32_t byte = ( a >= a0 && a < a1 ) ? out[a] : 0u;
                    crc = ( crc << 8 ) ^ table[( crc >> 24 ) ^ byte];
                }
            }
        }
        /* wave tree: after step d lane i (i % 2d == 0) holds the CRC of chunks [i, i+2d) */
        for ( int d = 1, k = CRC_CHUNK_LOG2; d < 64; d <<= 1, ++k ) {
            const uint32_t right = __shfl_down( crc, d );
            crc = gf_mul( crc, cc.pow8[k] ) ^ right;   /* pow8[k] = x^(8 * CRC_CHUNK * d) since CRC_CHUNK * d = 2^k */
        }
        if ( lane == 0 ) wcrc[wave] = crc;
        __syncthreads();
        if ( tid == 0 ) {
            uint32_t tileCrc = 0;
            for ( uint32_t w = 0; w < CRC_THREADS / 64; ++w ) {
                tileCrc = gf_mul( tileCrc, cc.pow8[CRC_CHUNK_LOG2 + 6] ) ^ wcrc[w];   /* 64 chunks per wave */
            }
            running = gf_mul( running, cc.pow8[CRC_CHUNK_LOG2 + 8] ) ^ tileCrc;       /* CRC_TILE bytes per tile */
        }
        __syncthreads();
        tEnd = t + CRC_TILE;
    }
    if ( tid == 0 ) {
        /* undo the trailing zero pad, then add the init contribution 0xFFFFFFFF * x^(8D) and invert */
        const uint64_t pad = tEnd - a1;
        uint32_t pure = D == 0 ? 0u : gf_mul( running, gf_pow8( cc.ipow8, pad ) );
        const uint32_t init = gf_mul( 0xFFFFFFFFu, gf_pow8( cc.pow8, D ) );
        const uint32_t crc = ~( pure ^ init );
        meta[b].computed_crc = crc;
        if ( crc != mt.header_crc ) meta[b].status = ST_CRC;
    }
}
}  // namespace bz2gpu
