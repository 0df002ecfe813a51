/**
 * bz2_hscan.hip.h -- Huffman stage in two kernels that break the one-serial-chain-of-codes-per-block form of a bzip2 decoder.
 *
 * The Huffman stream of a bzip2 block switches its code table every 50 symbols (Block::readBlockData,
 * src/indexed_bzip2/bzip2.hpp:709-723), so a decoder that starts in the middle of a block knows neither the code
 * alignment nor the table.  But the ONLY thing that is serial is the bit position at which every 50-symbol group starts:
 * once those positions are known, all groups (18 000 per level-9 block) decode independently.
 *
 *   k_hscan  one wavefront per block (large batches; k_hscan_spec below when few blocks have to be through quickly).  It
 *            parses header, selectors, code lengths and builds the canonical tables.  Then, group by
 *            group: all 64 lanes look up the length of the code that WOULD start at each of S consecutive bit
 *            positions (S = 64..1024, adapted to the table), which gives the successor array J1[i] = i + len(i); five
 *            rounds of pointer doubling in LDS, J2k[i] = Jk[Jk[i]], give J2, J4, J8, J16, J32, and the group that starts
 *            at bit x ends at J2[J16[J32[x]]] (32 + 16 + 2 = 50 symbols).  The dependent work per group is nine LDS
 *            round trips instead of ~50 chain steps on the scalar unit, no symbol is produced, nothing is written but
 *            one u32 per group.  End-of-block, "no code matches" and end-of-input positions are absorbing entries
 *            (they point to themselves), so the first group that runs into one of them is found exactly; consecutive
 *            groups with the same table share one set of arrays.  (The helpers keep their parameter K = wavefronts that
 *            share the rows of a build: round 2 measured K = 2, 4, 8 -- one barrier per round of doubling, 31 -> 24 ms
 *            for an incompressible block with eight waves, no more -- and a producer / consumer pair of waves, 10 to 20 %
 *            slower than one wave; both are gone, k_hscan_spec took their place.  Only K = 1 is instantiated.)
 *   k_hsym   one LANE per group: decodes its 50 symbols from the known start with the known table (per-lane bit reader,
 *            {length, symbol} look-up table in LDS, canonical range search for long codes =
 *            HuffmanCodingShortBitsCached::decode / decodeLong, src/huffman/HuffmanCodingShortBitsCached.hpp:98-150).
 *            The lane of the block's last group applies the reference's end-of-input / invalid-code rules symbol by
 *            symbol and writes the block's final record.
 *
 * k_mtf (bz2_stage1.hip.h) consumes the symbols.
 */
#pragma once

#include "bz2_stage1.hip.h"

namespace bz2gpu
{
constexpr uint32_t SCAN_LUT_BITS = 10;
constexpr uint32_t SCAN_LUT_SIZE = 1u << SCAN_LUT_BITS;
constexpr uint32_t SCAN_MAX_SPAN = 1024;        /* bit positions per build; 50 codes of 20 bits end below it */
constexpr uint32_t SCAN_ROWS = SCAN_MAX_SPAN / 64;
constexpr uint32_t GROUP_SYMS = 50;
constexpr uint32_t MAX_SCAN_GROUPS = 18002;     /* 900 100 symbols: more than any block that k_mtf accepts */
constexpr uint32_t GPOS_STRIDE = 18048;         /* u32 per block */
constexpr uint32_t LEN_STOP = 0x80u;            /* length-table flag: the code is the end-of-block symbol */
constexpr uint32_t SYM_THREADS = 128;           /* groups per chunk of a k_hsym workgroup (default: 256, bz2_device.hip) */
constexpr uint32_t SYM_CHUNKS = 4;              /* chunks per workgroup */

/** Decode tables of one block, written by k_hscan and read by every k_hsym workgroup of the block. */
struct alignas( 16 ) HuffTables
{
    uint16_t lut[6][SCAN_LUT_SIZE];   /* {len:5, sym:9 << 5} of the code that starts an index; 0: longer than the index / none */
    uint16_t perm[6][264];            /* symbols in canonical order */
    uint32_t first[6][24];            /* first code of length l */
    uint32_t count[6][24];            /* codes of length l */
    uint32_t offs[6][24];             /* index into perm of the first code of length l */
    uint32_t minmax[8];               /* minLen | maxLen << 8 per table */
};
static_assert( sizeof( HuffTables ) % 16 == 0 );

/** Per-block hand-off from k_hscan to k_hsym. */
struct ScanMeta
{
    uint64_t pos_base;      /* absolute bit position that relative position 0 stands for (multiple of 32) */
    uint32_t size_bits;     /* end of the input, relative (clamped) */
    uint32_t n_groups;      /* groups whose start is known */
    uint32_t terminal;      /* 1: the last group runs into end-of-block / an error and its lane finishes the record */
    uint32_t symbol_count;
    uint32_t pad[2];
};

constexpr uint32_t SCAN_RING_ENTRIES = 256;   /* stream words kept around the current position (8 192 bits: a producer that
                                                  has looked ahead may have to come back by a group or two) */
constexpr uint32_t SCAN_RING_MIRROR = 40;     /* entries repeated behind the end, so that a build never wraps */

constexpr uint32_t LEV_ENTRIES = SCAN_MAX_SPAN + 16;      /* entries per array: the span, OUT, TERM, padding */
constexpr uint32_t LEV_BYTES = 2 * LEV_ENTRIES;

/** The arrays of one build: J2, J16, J32 (and the intermediate levels) as byte offsets, see scan_build.  ONE array
 * object holding the three ("a", "b", "c" below), so that every access is derived from the same base: pointers that
 * walk from one member array into the next would tell the compiler that the accesses cannot alias. */
struct alignas( 16 ) ScanSlot
{
    uint16_t lev[3 * LEV_ENTRIES];

    __device__ __forceinline__ uint8_t* bytes() { return reinterpret_cast<uint8_t*>( lev ); }
    __device__ __forceinline__ uint16_t& at( uint32_t array, uint32_t entry ) { return lev[array * LEV_ENTRIES + entry]; }
};

/** What wave 0 found in front of the symbols, for the other waves and for the block's record. */
struct ScanHeader
{
    uint64_t enc_size, pos_base;
    int32_t  status, is_eos, is_eof;
    uint32_t active, header_crc, orig_ptr, symbol_count, n_sel, size_bits, p0, n_words;
};

template<uint32_t K>
struct ScanShared
{
    uint8_t  lenlut[6][SCAN_LUT_SIZE];   /* code length per index (| LEN_STOP: end-of-block), 0 = longer than the index or no code */
    uint32_t limit[6][24];               /* [t][l]: (first + count) << (20 - l), 0 outside [minLen, maxLen] */
    uint32_t eob_lo[6], eob_hi[6];       /* 20-bit windows that start with the end-of-block code: [lo, hi) */
    uint32_t minmax[6];
    /* stream words around the current position, two per entry so that one 8-byte read gives 64 stream bits in register
     * order: entry e = { word e + 1, word e }; entries [128, 168) repeat [0, 40) */
    uint32_t ring[2 * ( SCAN_RING_ENTRIES + SCAN_RING_MIRROR )];
    ScanHeader hdr;
    union alignas( 16 ) {
        struct {
            uint16_t lut[SCAN_LUT_SIZE];      /* first: copied out in 16-byte units */
            uint32_t first[24], count[24], offs[24], running[24];
            uint16_t perm[264];
            uint16_t bitmap[16];
            uint8_t  lens[264];
            uint8_t  sym_to_byte[256];
        } build;
        ScanSlot slot[K > 1 ? 2 : 1];   /* K > 1: builds alternate between two, see k_hscan */
    };
};

/** Stream word `index` (relative to the block's base word), as loaded (little-endian view of the big-endian stream), into
 * the ring: high half of entry `index`, low half of entry `index - 1`, and their mirrors.  The byte swap is done here and
 * not where the word was loaded: the load is issued one refill ahead, and a swap behind it would wait for the memory. */
template<uint32_t RING = SCAN_RING_ENTRIES>
__device__ __forceinline__ void
ring_put( uint32_t* ring, uint32_t index, uint32_t raw )
{
    const uint32_t value = be32( raw );
    const uint32_t e1 = index & ( RING - 1 );
    const uint32_t e0 = ( index - 1 ) & ( RING - 1 );
    ring[2 * e1 + 1] = value;
    ring[2 * e0] = value;
    if ( e1 < SCAN_RING_MIRROR ) ring[2 * ( RING + e1 ) + 1] = value;
    if ( e0 < SCAN_RING_MIRROR ) ring[2 * ( RING + e0 )] = value;
}

/** Barrier between the waves that share a build; a single wave only has to order its own LDS traffic. */
template<uint32_t K>
__device__ __forceinline__ void
scan_sync()
{
    if ( K > 1 ) __syncthreads(); else wave_sync();
}

/** J1 (successor of every bit position under the table of `lenlut`) and its doublings J2 .. J32 over S = 64 K RW bit
 * positions that start at relative bit position p.  The K waves of the workgroup share the rows of 64 positions: wave w
 * takes rows w, w + K, ...  Afterwards slot.b = J2, slot.c = J16, slot.a = J32 (only entry 0 if `firstOnly`: a single group
 * is chased from position 0).  Entries are BYTE offsets into these u16 arrays (2 x position), so that a round of
 * doubling is one ds_read + one ds_write per row: 2 i' for the next position, OUT = 2 S beyond the span, TERM = 2 S + 2 for
 * a position whose code is the end-of-block symbol, matches no code or ends behind the input.  OUT and TERM are entries
 * that point to themselves (written by the caller), i.e. absorbing under doubling.
 * Codes longer than the index bits of the length table are rare per position but present in most rows: their positions
 * are collected (arrays b and c are free until the first doubling) and resolved in ONE pass of range comparisons
 * instead of one per row. */
/* parts of scan_build: all of it (k_hscan); SCAN_ENDS: J1, the rounds up to J16 and then, for the first 64 x endRows
 * positions, where the 50-symbol group that starts there ends (k_hscan_spec) */
constexpr int SCAN_ALL = 0, SCAN_ENDS = 3;

template<uint32_t K, uint32_t RW, bool NEAR_END, int PART = SCAN_ALL, uint32_t RING = SCAN_RING_ENTRIES>
__device__ __forceinline__ void
scan_build( ScanSlot& slot, const uint8_t* lenlut, const uint32_t* ring, uint32_t p, uint32_t sizeBits,
            const uint32_t ( &lim )[10], uint32_t eobLo, uint32_t eobHi, bool firstOnly, uint32_t lane, uint32_t wave,
            uint32_t endRows = 0, uint32_t endBase = 0 )
{
    constexpr uint32_t S = 64 * K * RW;
    constexpr uint32_t OUT = 2 * S, TERM = 2 * S + 2;
    constexpr uint32_t STEP = 128 * K;          /* bytes between two rows of one wave */
    uint8_t* const base = slot.bytes();
    const uint8_t* const A = base;
    const uint8_t* const B = base + LEV_BYTES;
    const uint8_t* const C = base + 2 * LEV_BYTES;
    uint8_t* const mine = base + 128 * wave + 2 * lane;   /* my entry of row 0, array a */
    constexpr uint32_t TO_B = LEV_BYTES, TO_C = 2 * LEV_BYTES;
    /* positions with long codes, 1 040 / K entries per wave (arrays b and c): position << 20 | 20-bit window */
    uint32_t* const pending = reinterpret_cast<uint32_t*>( base + LEV_BYTES ) + wave * ( LEV_ENTRIES / K );
    const uint32_t first = 64 * wave + lane;     /* my position in row 0 */
    uint32_t own[RW], v20s[RW];
    {
        /* 64 stream bits from the lane's word on: one 8-byte read per row at a fixed distance from the first row's */
        const uint32_t a0 = p + first;
        const uint64_t* const w = reinterpret_cast<const uint64_t*>( ring ) + ( ( a0 >> 5 ) & ( RING - 1 ) );
        const uint32_t sh = a0 & 31u;
#pragma unroll
        for ( uint32_t j = 0; j < RW; ++j ) {
            const uint32_t bits32 = (uint32_t)( ( w[2 * K * j] << sh ) >> 32 );
            v20s[j] = bits32 >> 12;
            own[j] = lenlut[bits32 >> ( 32 - SCAN_LUT_BITS )];
        }
    }
    uint32_t nPending = 0;
#pragma unroll
    for ( uint32_t j = 0; j < RW; ++j ) {
        const uint64_t isLong = __ballot( own[j] == 0 );
        if ( isLong != 0 ) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi( (uint32_t)( isLong >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)isLong, 0 ) );
            if ( own[j] == 0 ) pending[nPending + rank] = ( ( first + 64 * K * j ) << 20 ) | v20s[j];
            nPending += (uint32_t)__popcll( isLong );
        }
    }
    const uint32_t lane2 = 2 * first;
#pragma unroll
    for ( uint32_t j = 0; j < RW; ++j ) {
        const uint32_t len = own[j];
        bool stop = len >= LEN_STOP;
        if ( NEAR_END ) stop |= p + first + 64 * K * j + ( len & 31u ) > sizeBits;
        /* 2 (i + len), clamped to OUT, with i = first + 64 K j */
        uint32_t next = 2 * len + lane2;
        next = ( next < OUT - STEP * j ? next : OUT - STEP * j ) + STEP * j;
        own[j] = stop ? TERM : next;
        *reinterpret_cast<uint16_t*>( mine + STEP * j ) = (uint16_t)own[j];
    }
    if ( nPending != 0 ) {
        /* canonical codes: the first length whose range end exceeds the window (decodeLong,
         * HuffmanCodingShortBitsCached.hpp:117-150); 21: no code matches */
        wave_sync();
        for ( uint32_t k = lane; k < nPending; k += 64 ) {
            const uint32_t e = pending[k];
            const uint32_t v20 = e & 0xFFFFFu, i = e >> 20;
            uint32_t ll = 11;
#pragma unroll
            for ( uint32_t l = 0; l < 10; ++l ) ll += v20 >= lim[l] ? 1u : 0u;
            bool stop = ( ll > 20 ) | ( ( v20 >= eobLo ) & ( v20 < eobHi ) );
            if ( NEAR_END ) stop |= p + i + ll > sizeBits;
            const uint32_t next = 2 * ( i + ll ) < OUT ? 2 * ( i + ll ) : OUT;
            *reinterpret_cast<uint16_t*>( base + 2 * i ) = (uint16_t)( stop ? TERM : next );
        }
        /* Behind the end-of-block code of a block's last group the span runs into the next block's header, where most
         * positions look like long codes: with a single wave the list may then have grown over the OUT / TERM entries of
         * arrays b and c (entries S and S + 1; a list of more than S / 2 positions), which are put back here. */
        if ( K == 1 && 2 * nPending > S && lane < 2 ) {
            const uint16_t v = (uint16_t)( OUT + 2 * lane );
            *reinterpret_cast<uint16_t*>( base + LEV_BYTES + OUT + 2 * lane ) = v;
            *reinterpret_cast<uint16_t*>( base + 2 * LEV_BYTES + OUT + 2 * lane ) = v;
        }
        wave_sync();
#pragma unroll
        for ( uint32_t j = 0; j < RW; ++j ) own[j] = *reinterpret_cast<const uint16_t*>( mine + STEP * j );
    }
    scan_sync<K>();
    /* one round of doubling: own[j] = src[own[j]], the same into dst */
#define SCAN_LEVEL( src, toDst ) \
    _Pragma( "unroll" ) for ( uint32_t j = 0; j < RW; ++j ) own[j] = *reinterpret_cast<const uint16_t*>( src + own[j] ); \
    _Pragma( "unroll" ) for ( uint32_t j = 0; j < RW; ++j ) *reinterpret_cast<uint16_t*>( mine + ( toDst ) + STEP * j ) = (uint16_t)own[j]; \
    scan_sync<K>();
    SCAN_LEVEL( A, TO_B )   /* J2 -> b (kept) */
    SCAN_LEVEL( B, TO_C )   /* J4 */
    SCAN_LEVEL( C, 0 )      /* J8 */
    SCAN_LEVEL( A, TO_C )   /* J16 -> c (kept) */
    if constexpr ( PART == SCAN_ENDS ) {
        /* 16 + 16 + 16 + 2 symbols from each of the first positions (own[] = J16 of them) -> a (J8 is not needed any more),
         * as bit positions from endBase bits in front of the span: what the chain of k_hscan_spec reads */
#pragma unroll
        for ( uint32_t j = 0; j < RW; ++j ) {
            if ( j < endRows ) {
                uint32_t e = *reinterpret_cast<const uint16_t*>( C + own[j] );
                e = *reinterpret_cast<const uint16_t*>( C + e );
                e = *reinterpret_cast<const uint16_t*>( B + e );
                e = e == OUT ? 0xFFFFu : ( e == TERM ? 0xFFFEu : endBase + ( e >> 1 ) );
                *reinterpret_cast<uint16_t*>( mine + STEP * j ) = (uint16_t)e;
            }
        }
        scan_sync<K>();
        return;
    }
    if ( firstOnly ) {
        if ( wave == 0 ) {
            const uint32_t j32 = *reinterpret_cast<const uint16_t*>( C + own[0] );
            if ( lane == 0 ) *reinterpret_cast<uint16_t*>( base ) = (uint16_t)j32;
        }
        scan_sync<K>();
    } else {
        SCAN_LEVEL( C, 0 )   /* J32 -> a (kept) */
    }
#undef SCAN_LEVEL
}

/** Header, selectors, code lengths, canonical tables of one block: Block::readBlockHeader .. readTrees
 * (bzip2.hpp:479-685), executed by ONE wavefront.  Leaves the tables in `sh` (and in `tabs` for k_hsym) and the rest
 * in sh.hdr. */
template<uint32_t K>
__device__ __forceinline__ void
scan_parse( ScanShared<K>& sh, const uint32_t* __restrict__ in_words, uint64_t in_size_bytes, uint64_t start,
            uint8_t* sel, uint8_t* __restrict__ stb_buf, HuffTables* tabs, uint32_t b, uint32_t lane )
{
    BitRd br;
    br.init( in_words, in_size_bytes, start );

    int32_t status = ST_OK;
    uint32_t headerCrc = 0, origPtr = 0;
    int32_t isEos = 0, isEof = 0;
    uint64_t encSize = 0;
    uint32_t symbolCount = 0, groupCount = 0, nSel = 0;
    uint32_t active = 0;

#define FAIL( code ) do { status = br.eof ? (int32_t)ST_EOF : (int32_t)( code ); goto done; } while ( 0 )

    /* ---- Block::readBlockHeader, bzip2.hpp:479-523 ---- */
    if ( start > br.size_bits ) {
        br.eof = true;
        FAIL( ST_EOF );
    }
    {
        const uint64_t hi = br.read( 24 );
        const uint64_t lo = br.read( 24 );
        const uint64_t magic = ( hi << 24 ) | lo;
        headerCrc = br.read( 32 );
        if ( br.eof ) {
            headerCrc = 0;   /* the reference's read throws before anything is assigned */
            FAIL( ST_EOF );
        }
        if ( magic == 0x177245385090ULL ) {
            isEos = 1;
            const uint32_t inByte = (uint32_t)( br.pos & 7 );
            if ( inByte > 0 ) {
                br.read( 8 - inByte );
                if ( br.eof ) FAIL( ST_EOF );
            }
            encSize = br.pos - start;
            isEof = br.pos >= br.size_bits;
            goto done;
        }
        if ( magic != 0x314159265359ULL ) FAIL( ST_BAD_MAGIC );
        const uint32_t randomized = br.read( 1 );
        if ( br.eof ) FAIL( ST_EOF );
        if ( randomized ) FAIL( ST_RANDOMIZED );
        origPtr = br.read( 24 );
        if ( br.eof ) {
            origPtr = 0;
            FAIL( ST_EOF );
        }
        if ( origPtr > MAX_N ) FAIL( ST_ORIGPTR_RANGE );
    }

    /* ---- Block::readSymbolMaps, bzip2.hpp:526-571 ---- */
    {
        const uint32_t used = br.read( 16 );
        for ( uint32_t v = lane; v < 256; v += 64 ) sh.build.sym_to_byte[v] = 0;   /* fresh Block: zero-initialised */
        for ( int i = 0; i < 16; ++i ) {
            uint32_t bm = 0;
            if ( used & ( 1u << ( 15 - i ) ) ) {
                bm = br.read( 16 );
            }
            if ( lane == 0 ) sh.build.bitmap[i] = (uint16_t)bm;
        }
        wave_sync();
        uint32_t total = 0;
        for ( int g = 0; g < 16; ++g ) total += __popc( sh.build.bitmap[g] );
        symbolCount = total;
        for ( uint32_t v = lane; v < 256; v += 64 ) {
            const uint32_t g = v >> 4, j = v & 15;
            const uint32_t bm = sh.build.bitmap[g];
            if ( bm & ( 1u << ( 15 - j ) ) ) {
                uint32_t rank = 0;
                for ( uint32_t gg = 0; gg < g; ++gg ) rank += __popc( sh.build.bitmap[gg] );
                rank += j == 0 ? 0 : __popc( bm >> ( 16 - j ) );
                sh.build.sym_to_byte[rank] = (uint8_t)v;
            }
        }
        wave_sync();
        if ( br.eof ) FAIL( ST_EOF );
        reinterpret_cast<uint32_t*>( stb_buf + (size_t)b * 256 )[lane] =
            reinterpret_cast<const uint32_t*>( sh.build.sym_to_byte )[lane];
    }

    /* ---- Block::readSelectors, bzip2.hpp:574-637 ---- */
    {
        groupCount = br.read( 3 );
        if ( br.eof ) FAIL( ST_EOF );
        if ( groupCount < 2 || groupCount > 6 ) FAIL( ST_GROUP_COUNT );
        nSel = br.read( 15 );
        if ( br.eof ) FAIL( ST_EOF );
        if ( nSel == 0 ) FAIL( ST_SELECTOR_COUNT );
        /* The selectors are unary codes (j ones and a zero, j < groupCount) whose values go through a move-to-front list of
         * the tables: 18 000 of them per full block, 0.85 ms when one code is read at a time.  Here 2 048 bits per round:
         * every zero bit ends one code, so lane l finds the codes that end in ITS 32 bits (their number in stream order from
         * a prefix sum of the zero counts, their length from the distance to the zero before), and the move-to-front list
         * is carried across the lanes as a composition of permutations of six entries (nibble k of a word = entry k).
         * The reference's checks, per selector in stream order: fewer than 6 bits left in front of it -> end of input
         * (peek<6>, BitReader.hpp:458-460), then j >= groupCount -> invalid. */
        uint8_t* const jbuf = reinterpret_cast<uint8_t*>( sh.build.lut );     /* 2 048 code lengths; free until the tables are built */
        static_assert( sizeof( sh.build.lut ) >= 2048 );
        uint64_t cursor = br.pos;          /* where the next selector starts */
        uint32_t mtfsel = 0x543210u;       /* nibble k = entry k */
        const auto compose = [] ( uint32_t first, uint32_t then ) {     /* the list after `first` and then `then`: first[then[k]] */
            uint32_t r = 0;
#pragma unroll
            for ( uint32_t k = 0; k < 6; ++k ) r |= ( ( first >> ( 4 * ( ( then >> ( 4 * k ) ) & 0xFu ) ) ) & 0xFu ) << ( 4 * k );
            return r;
        };
        const auto moveToFront = [] ( uint32_t list, uint32_t j ) {     /* entry j to the front */
            const uint32_t shj = 4 * j;
            const uint32_t val = ( list >> shj ) & 0xFu;
            const uint32_t low = list & ( ( 1u << shj ) - 1u );
            const uint32_t highMask = ~( ( 16u << shj ) - 1u );
            return ( list & highMask ) | ( low << 4 ) | val;
        };
        for ( uint32_t done = 0; done < nSel; ) {
            /* this lane's 32 bits */
            const uint64_t at = cursor + 32u * lane;
            uint32_t word;
            {
                const uint64_t i = at >> 5;
                const uint32_t shw = (uint32_t)( at & 31u );
                const uint32_t a = i < br.nwords ? be32( br.w[i] ) : 0u;
                const uint32_t c = i + 1 < br.nwords ? be32( br.w[i + 1] ) : 0u;
                word = shw != 0 ? ( a << shw ) | ( c >> ( 32 - shw ) ) : a;
            }
            const uint32_t before = (uint32_t)__shfl_up( (int)word, 1 );
            /* ones at the end of the words in front of this one (6 stands for "six or more": invalid anyway) */
            const uint32_t carry = lane == 0 ? 0u : ( before == 0xFFFFFFFFu ? 6u : (uint32_t)__builtin_ctz( ~before ) );
            uint32_t zeros = ~word;
            const uint32_t count = (uint32_t)__popc( zeros );
            uint32_t incl = count;
#pragma unroll
            for ( int d = 1; d < 64; d <<= 1 ) {
                const uint32_t o = (uint32_t)__shfl_up( (int)incl, d );
                if ( (int)lane >= d ) incl += o;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane( (int)incl, 63 );
            const uint32_t take = total < nSel - done ? total : nSel - done;
            if ( total == 0 ) {
                /* 2 048 ones: the selector at `cursor` is invalid, if it is not the end of the input */
                if ( cursor + 6 > br.size_bits ) br.eof = true;
                br.pos = cursor;
                FAIL( ST_SELECTOR_UNARY );
            }
            /* the codes that end in this word */
            const uint32_t firstIndex = incl - count; /* number (in this round) of the first of them */
            uint32_t index = firstIndex;
            uint32_t prev = 0;                        /* bit behind the zero before (0: the word's first bit) */
            uint32_t ones = carry;                    /* ones in front of the word's first bit */
            uint32_t firstBad = 0xFFFFFFFFu;          /* number of the first code here that fails a check */
            uint32_t endOfLast = 0;                   /* bit behind the zero of code number take - 1, if it is here */
            while ( zeros != 0 ) {
                const uint32_t q = (uint32_t)__builtin_clz( zeros );      /* position of the zero, 0 = first bit */
                zeros &= ~( 0x80000000u >> q );
                uint32_t j = q - prev + ones;
                if ( j > 6 ) j = 6;
                if ( index < take ) {
                    /* (six or more ones are invalid wherever the code starts; shorter codes start exactly here) */
                    const bool atEnd = at + q - j + 6 > br.size_bits;
                    if ( ( atEnd || j >= groupCount ) && firstBad == 0xFFFFFFFFu ) firstBad = index;
                    jbuf[index] = (uint8_t)j;
                    if ( index + 1 == take ) endOfLast = q + 1;
                }
                prev = q + 1;
                ones = 0;
                ++index;
            }
            /* the first selector in stream order that fails decides; it starts behind the zero of the one before it */
            uint32_t bad = firstBad;
#pragma unroll
            for ( int d = 32; d >= 1; d >>= 1 ) {
                const uint32_t o = (uint32_t)__shfl_xor( (int)bad, d );
                bad = o < bad ? o : bad;
            }
            if ( bad != 0xFFFFFFFFu ) {
                uint64_t startsAt = cursor;
                if ( bad > 0 ) {
                    /* the lane whose word holds the zero of code number bad - 1 */
                    uint32_t behind = 0;
                    if ( bad - 1 >= firstIndex && bad - 1 < firstIndex + count ) {
                        uint32_t z = ~word;
                        for ( uint32_t k = firstIndex; k < bad; ++k ) {
                            behind = (uint32_t)__builtin_clz( z ) + 1;
                            z &= ~( 0x80000000u >> ( behind - 1 ) );
                        }
                    }
                    const uint64_t owner = __ballot( behind != 0 );
                    const uint32_t ownerLane = (uint32_t)__builtin_ctzll( owner );
                    startsAt = cursor + 32u * ownerLane + (uint32_t)__builtin_amdgcn_readlane( (int)behind, ownerLane );
                }
                if ( startsAt + 6 > br.size_bits ) br.eof = true;
                FAIL( ST_SELECTOR_UNARY );
            }
            /* where the next round starts */
            {
                const uint64_t owner = __ballot( endOfLast != 0 );
                const uint32_t ownerLane = (uint32_t)__builtin_ctzll( owner );
                cursor += 32u * ownerLane + (uint32_t)__builtin_amdgcn_readlane( (int)endOfLast, ownerLane );
            }
            wave_sync();
            /* move-to-front over the `take` values: lane l has values [l R, l R + R) */
            const uint32_t R = ( take + 63 ) / 64;
            const uint32_t lo = lane * R < take ? lane * R : take;
            const uint32_t hi = lo + R < take ? lo + R : take;
            uint32_t mine = 0x543210u;
            for ( uint32_t k = lo; k < hi; ++k ) mine = moveToFront( mine, jbuf[k] );
            uint32_t scan = mine;                     /* inclusive: the lists of lanes 0 .. l in turn */
#pragma unroll
            for ( int d = 1; d < 64; d <<= 1 ) {
                const uint32_t o = (uint32_t)__shfl_up( (int)scan, d );
                if ( (int)lane >= d ) scan = compose( o, scan );
            }
            uint32_t list = (uint32_t)__shfl_up( (int)scan, 1 );
            list = lane == 0 ? mtfsel : compose( mtfsel, list );
            for ( uint32_t k = lo; k < hi; ++k ) {
                list = moveToFront( list, jbuf[k] );
                sel[done + k] = (uint8_t)( list & 0xFu );
            }
            mtfsel = compose( mtfsel, (uint32_t)__builtin_amdgcn_readlane( (int)scan, 63 ) );
            wave_sync();                              /* jbuf is written again in the next round */
            done += take;
        }
        br.init( br.w, in_size_bytes, cursor );
    }

    /* ---- Block::readTrees, bzip2.hpp:644-685, and the canonical tables, one table at a time ---- */
    {
        const uint32_t symCount = symbolCount + 2;
        const uint32_t eob = symbolCount + 1;
        for ( uint32_t t = 0; t < groupCount; ++t ) {
            uint32_t hh = br.read( 5 );
            if ( br.eof ) FAIL( ST_EOF );
            for ( uint32_t s = 0; s < symCount; ++s ) {
                for ( ;; ) {
                    if ( hh - 1u > 19u ) FAIL( ST_CODE_LENGTH );
                    br.refill();
                    const uint32_t b2 = br.peek( 2 );
                    if ( b2 < 2 ) {
                        if ( br.pos + 1 > br.size_bits ) { br.eof = true; FAIL( ST_EOF ); }
                        br.skip( 1 );
                        break;
                    }
                    if ( br.pos + 2 > br.size_bits ) { br.eof = true; FAIL( ST_EOF ); }
                    hh += b2 == 2 ? 1u : 0xFFFFFFFFu;
                    br.skip( 2 );
                }
                if ( lane == 0 ) sh.build.lens[s] = (uint8_t)hh;
            }
            /* the reference builds (and checks) the coding of a group before it reads the next group's lengths
             * (bzip2.hpp:679-683): an over-subscribed set here wins over a bad length further on */
            wave_sync();
            uint32_t c = 0;
            if ( lane >= 1 && lane <= 20 ) {
                for ( uint32_t s = 0; s < symCount; ++s ) c += sh.build.lens[s] == lane;
            }
            if ( lane < 24 ) sh.build.count[lane] = c;
            wave_sync();
            uint32_t minLen = 0, maxLen = 0;
            for ( uint32_t l = 1; l <= 20; ++l ) {
                if ( sh.build.count[l] != 0 ) {
                    if ( minLen == 0 ) minLen = l;
                    maxLen = l;
                }
            }
            {
                uint32_t unused = 1u << minLen;
                bool bad = false;
                for ( uint32_t l = minLen; l <= maxLen; ++l ) {
                    const uint32_t f = sh.build.count[l];
                    if ( f > unused ) { bad = true; break; }
                    unused = ( unused - f ) * 2u;
                }
                if ( bad ) FAIL( ST_HUFFMAN_LENGTHS );
            }
            if ( lane == 0 ) {
                uint32_t minCode = 0, sum = 0;
                for ( uint32_t l = 0; l < 24; ++l ) { sh.build.first[l] = 0; sh.build.offs[l] = 0; }
                for ( uint32_t l = minLen; l <= maxLen; ++l ) {
                    minCode = ( minCode + ( l > minLen ? sh.build.count[l - 1] : 0u ) ) << 1;
                    if ( l == minLen ) minCode = 0;
                    sh.build.first[l] = minCode;
                    sh.build.offs[l] = sum;
                    sh.build.running[l] = sum;
                    sum += sh.build.count[l];
                }
                sh.minmax[t] = minLen | ( maxLen << 8 );
            }
            wave_sync();
            for ( uint32_t base = 0; base < symCount; base += 64 ) {
                const uint32_t s = base + lane;
                const bool valid = s < symCount;
                const uint32_t len = valid ? sh.build.lens[s] : 0u;
                const uint64_t same = match_any( len, 5, valid );
                const uint32_t rank = popc_below( same, lane );
                uint32_t basePos = 0;
                if ( valid ) basePos = sh.build.running[len];
                if ( valid ) sh.build.perm[basePos + rank] = (uint16_t)s;
                wave_sync();
                if ( valid && rank == 0 ) sh.build.running[len] = basePos + (uint32_t)__popcll( same );
                wave_sync();
            }
            if ( lane < 24 ) {
                const bool in = lane >= minLen && lane <= maxLen;
                sh.limit[t][lane] = in ? ( ( sh.build.first[lane] + sh.build.count[lane] ) << ( 20 - lane ) ) : 0u;
            }
            if ( lane == 0 ) {
                /* canonical order is (length, symbol): the end-of-block symbol, the highest, is the last code of its length */
                const uint32_t le = sh.build.lens[eob];
                const uint32_t hi = ( sh.build.first[le] + sh.build.count[le] ) << ( 20 - le );
                sh.eob_hi[t] = hi;
                sh.eob_lo[t] = hi - ( 1u << ( 20 - le ) );
            }
            const uint32_t lutMax = maxLen < SCAN_LUT_BITS ? maxLen : SCAN_LUT_BITS;
            for ( uint32_t e = lane; e < SCAN_LUT_SIZE; e += 64 ) {
                uint32_t val = 0;
                for ( uint32_t l = minLen; l <= lutMax; ++l ) {
                    const uint32_t code = e >> ( SCAN_LUT_BITS - l );
                    const uint32_t d = code - sh.build.first[l];
                    if ( d < sh.build.count[l] ) {
                        val = l | ( (uint32_t)sh.build.perm[sh.build.offs[l] + d] << 5 );
                        break;
                    }
                }
                sh.build.lut[e] = (uint16_t)val;
                sh.lenlut[t][e] = (uint8_t)( ( val & 31u ) | ( ( val >> 5 ) == eob && val != 0 ? LEN_STOP : 0u ) );
            }
            wave_sync();
            /* the block's tables for k_hsym */
            {
                const uint4* const src = reinterpret_cast<const uint4*>( sh.build.lut );
                uint4* const dst = reinterpret_cast<uint4*>( tabs->lut[t] );
                for ( uint32_t k = lane; k < SCAN_LUT_SIZE * 2 / 16; k += 64 ) dst[k] = src[k];
                for ( uint32_t k = lane; k < 264; k += 64 ) tabs->perm[t][k] = k < symCount ? sh.build.perm[k] : (uint16_t)0;
                if ( lane < 24 ) {
                    tabs->first[t][lane] = sh.build.first[lane];
                    tabs->count[t][lane] = lane <= 20 ? sh.build.count[lane] : 0u;
                    tabs->offs[t][lane] = sh.build.offs[lane];
                }
                if ( lane == 0 ) tabs->minmax[t] = minLen | ( maxLen << 8 );
            }
            wave_sync();
        }
    }
    active = 1;

done:
#undef FAIL
    if ( lane == 0 ) {
        ScanHeader h;
        const uint64_t posBase = br.pos & ~31ull;   /* bit positions of the scan are 32-bit, relative to this word */
        const uint64_t wordsLeft = ( ( in_size_bytes + 3 ) >> 2 ) - ( posBase >> 5 );
        h.enc_size = encSize;
        h.pos_base = posBase;
        h.status = status;
        h.is_eos = isEos;
        h.is_eof = isEof;
        h.active = active;
        h.header_crc = headerCrc;
        h.orig_ptr = origPtr;
        h.symbol_count = symbolCount;
        h.n_sel = nSel;
        h.size_bits = br.size_bits - posBase < 0xFFFF0000ull ? (uint32_t)( br.size_bits - posBase ) : 0xFFFF0000u;
        h.p0 = (uint32_t)( br.pos - posBase );
        h.n_words = wordsLeft < 0x08000000ull ? (uint32_t)wordsLeft : 0x08000000u;
        sh.hdr = h;
    }
}

/** Rows of 64 positions per wave for a span of `rows` rows shared by K waves: the builds are statically unrolled, so
 * the count is rounded up to the next instance. */
template<uint32_t K>
__device__ __forceinline__ uint32_t
scan_rows_per_wave( uint32_t rows )
{
    const uint32_t rw = ( rows + K - 1 ) / K;
    constexpr uint32_t MAX = SCAN_ROWS / K;
    if ( rw <= 6 ) return rw < MAX ? rw : MAX;
    return rw <= 8 ? ( 8u < MAX ? 8u : MAX ) : ( rw <= 10 ? ( 10u < MAX ? 10u : MAX ) : ( rw <= 12 ? ( 12u < MAX ? 12u : MAX ) : MAX ) );
}

template<uint32_t K, int PART = SCAN_ALL, uint32_t RING = SCAN_RING_ENTRIES>
__device__ __forceinline__ void
scan_build_rows( uint32_t rw, bool nearEnd, ScanSlot& slot, const uint8_t* lenlut, const uint32_t* ring, uint32_t p,
                 uint32_t sizeBits, const uint32_t ( &lim )[10], uint32_t eobLo, uint32_t eobHi, bool one, uint32_t lane,
                 uint32_t wave, uint32_t endRows = 0, uint32_t endBase = 0 )
{
    constexpr uint32_t MAX = SCAN_ROWS / K;
#define SCAN_CASE( n ) \
    if constexpr ( ( n ) <= MAX ) { \
        if ( rw == ( n ) ) { scan_build<K, ( n ), false, PART, RING>( slot, lenlut, ring, p, sizeBits, lim, eobLo, eobHi, one, lane, wave, endRows, endBase ); return; } \
    }
    if ( nearEnd ) {
        scan_build<K, MAX, true, PART, RING>( slot, lenlut, ring, p, sizeBits, lim, eobLo, eobHi, one, lane, wave, endRows, endBase );
        return;
    }
    SCAN_CASE( 1 ) SCAN_CASE( 2 ) SCAN_CASE( 3 ) SCAN_CASE( 4 ) SCAN_CASE( 5 ) SCAN_CASE( 6 ) SCAN_CASE( 8 ) SCAN_CASE( 10 )
    SCAN_CASE( 12 )
    scan_build<K, MAX, false, PART, RING>( slot, lenlut, ring, p, sizeBits, lim, eobLo, eobHi, one, lane, wave, endRows, endBase );
#undef SCAN_CASE
}

/* W = wavefronts per SIMD that the kernel's registers must leave room for.  Its LDS alone allows 2.5 per SIMD, and the
 * compiler, seeing that, spends 172 registers per lane: fine for this kernel by itself, but two such waves on a SIMD leave
 * no room for ONE wave of k_mtf or k_link2 (169 to 171 registers) of the batches beside it, and only 2 048 of a big batch's
 * 2 560 blocks are resident at a time.  The LDS is therefore declared at launch (the compiler does not see its size) and
 * the register budget is given: W = 4 -> 128 registers (7 spilled, none in the chase), W = 5 -> 96. */
template<uint32_t K, uint32_t W = 2>
__global__ __launch_bounds__( 64 * K ) __attribute__( ( amdgpu_waves_per_eu( W, 8 ) ) ) void
k_hscan( const uint32_t* __restrict__ in_words,
         uint64_t                     in_size_bytes,
         const uint64_t* __restrict__ offsets,
         BlockMeta* __restrict__      meta,
         HuffMeta* __restrict__       hmeta,
         ScanMeta* __restrict__       smeta,
         uint8_t*                     sel_buf,
         uint8_t* __restrict__        stb_buf,
         HuffTables* __restrict__     tab_buf,
         uint32_t* __restrict__       gpos_buf,
         uint32_t                     n_blocks,
         const uint32_t* __restrict__ order,
         uint32_t                     tune,    /* debugging: 1 = one group per build, 2 = always the full span */
         uint32_t*                    queue )  /* K == 1 only: not null = the workgroups take blocks from this counter (zeroed by
                                                  the host) until none is left, so that a grid smaller than the batch -- one
                                                  that leaves LDS to the kernels of other streams -- still decodes all of it */
{
    static_assert( K == 1, "one wavefront per block: the forms with shared rows were measured and dropped (DESIGN.md)" );
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t ldsAtLaunch[];     /* sizeof( ScanShared<K> ) */
    ScanShared<K>& sh = *reinterpret_cast<ScanShared<K>*>( ldsAtLaunch );
    for ( uint32_t slotIndex = blockIdx.x;; slotIndex += gridDim.x ) {
    if ( K == 1 && queue != nullptr ) {
        uint32_t taken = 0;
        if ( threadIdx.x == 0 ) taken = atomicAdd( queue, 1u );
        slotIndex = sfl( taken );
    }
    if ( slotIndex >= n_blocks ) return;
    const uint32_t b = sfl( order[slotIndex] );
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = sfl( threadIdx.x >> 6 );
    uint8_t* const sel = sel_buf + (size_t)b * SEL_STRIDE;
    uint32_t* const gpos = gpos_buf + (size_t)b * GPOS_STRIDE;
    const uint64_t start = offsets[b];

    if ( wave == 0 ) {
        scan_parse<K>( sh, in_words, in_size_bytes, start, sel, stb_buf, tab_buf + b, b, lane );
        __threadfence_block();   /* the selectors are read back below */
    }
    scan_sync<K>();

    const uint32_t active = sfl( sh.hdr.active );
    int32_t status = (int32_t)sfl( (uint32_t)sh.hdr.status );
    const uint32_t nSel = sfl( sh.hdr.n_sel );
    const uint32_t sizeBits = sfl( sh.hdr.size_bits );
    const uint32_t nWords = sfl( sh.hdr.n_words );
    const uint64_t posBase = ( (uint64_t)sfl( (uint32_t)( sh.hdr.pos_base >> 32 ) ) << 32 ) | sfl( (uint32_t)sh.hdr.pos_base );
    uint32_t nGroups = 0, terminal = 0;

    /* ---- where every 50-symbol group of Block::readBlockData's loop (bzip2.hpp:709-723) starts ---- */
    if ( active ) {
        const uint32_t* const words = in_words + ( posBase >> 5 );
        uint32_t p = sfl( sh.hdr.p0 );
        uint32_t g = 0;

        /* stream ring: words [wHi - 256, wHi) are in sh.ring, the next 64 are on their way in `pend` (wave 0 fills) */
        uint32_t wHi = 0;
        uint32_t pend = 0;
        if ( wave == 0 && lane < nWords ) pend = words[lane];
        /* selectors of groups [64 k, 64 k + 64), one per lane, the next 64 on their way */
        uint32_t selV = sel[lane];
        uint32_t selPend = sel[64 + lane];
        /* per-table estimate of a group's length in bits, lane t = table t; 0 = not seen yet (full span) */
        uint32_t estV = 0;
        uint32_t gposV = 0;
        uint32_t lastS[2] = { 0, 0 }, lastT = 0xFFFFFFFFu;
        uint32_t lim[10] = {};
        uint32_t eobLo = 0, eobHi = 0;
        uint32_t which = 0;        /* the two slots alternate: a wave that is through with a chase may start the next build */
        bool forceFull = false;

        for ( ;; ) {
            if ( g >= nSel ) { status = ST_SELECTOR_OVERRUN; break; }
            if ( g >= MAX_SCAN_GROUPS ) { status = ST_DATA_OVERFLOW; break; }
            /* stream words up to (p + SCAN_MAX_SPAN + 64) >> 5 (every wave keeps count, wave 0 moves the words) */
            if ( ( p >> 5 ) + ( SCAN_MAX_SPAN + 96 ) / 32 > wHi ) {
                while ( ( p >> 5 ) + ( SCAN_MAX_SPAN + 96 ) / 32 > wHi ) {
                    if ( wave == 0 ) {
                        ring_put( sh.ring, wHi + lane, pend );
                        pend = 0;
                        if ( wHi + 64 + lane < nWords ) pend = words[wHi + 64 + lane];
                    }
                    wHi += 64;
                }
                scan_sync<K>();
            }
            /* this group's table and how many of the following groups (of this window of 64) use it as well */
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane( (int)selV, g & 63u );
            uint32_t runLen;
            {
                const uint64_t same = __ballot( selV == t ) >> ( g & 63u );   /* bit 0 = this group */
                runLen = (uint32_t)__builtin_ctzll( ~same );                    /* >= 1; 64 if all the rest match */
                if ( runLen > 64u - ( g & 63u ) ) runLen = 64u - ( g & 63u );
                if ( runLen > nSel - g ) runLen = nSel - g;
                if ( runLen > MAX_SCAN_GROUPS - g ) runLen = MAX_SCAN_GROUPS - g;
            }
            if ( t != lastT ) {
                /* range ends of the long codes of this table, flat beyond its longest code: a 20-bit window v holds a code
                 * of length 11 + #{ l : v >= lim[l] }, none if that comes to 21 */
                const uint32_t mx = sfl( sh.minmax[t] ) >> 8;
                const uint32_t limV = sh.limit[t][lane < mx ? lane : mx];
#pragma unroll
                for ( uint32_t l = 0; l < 10; ++l ) lim[l] = (uint32_t)__builtin_amdgcn_readlane( (int)limV, 11 + l );
                eobLo = sfl( sh.eob_lo[t] );
                eobHi = sfl( sh.eob_hi[t] );
                lastT = t;
            }
            const uint32_t est = (uint32_t)__builtin_amdgcn_readlane( (int)estV, t );
            const bool nearEnd = p + SCAN_MAX_SPAN + 32 > sizeBits;
            uint32_t rows = SCAN_ROWS, m = 1;
            if ( est != 0 && !forceFull && !nearEnd && !( tune & 2u ) ) {
                const uint32_t need = est + ( est >> 3 ) + 16;   /* a group of this table: last one + 12 % + 16 bits */
                m = ( SCAN_MAX_SPAN - 24 ) / need;
                if ( tune & 1u ) m = 1;
                if ( m > runLen ) m = runLen;
                if ( m < 1 ) m = 1;
                rows = ( m * need + 24 + 63 ) >> 6;
                if ( rows > SCAN_ROWS ) rows = SCAN_ROWS;
            }
            forceFull = false;
            const uint32_t rw = nearEnd ? SCAN_ROWS / K : scan_rows_per_wave<K>( rows );
            const uint32_t S = 64 * K * rw;
            ScanSlot& slot = sh.slot[which];
            if ( S != lastS[which] ) {
                /* OUT = 2 S and TERM = 2 S + 2 point to themselves in all three arrays (every wave writes the same) */
                if ( lane < 2 ) {
                    const uint16_t v = (uint16_t)( 2 * ( S + lane ) );
                    slot.at( 0, S + lane ) = v; slot.at( 1, S + lane ) = v; slot.at( 2, S + lane ) = v;
                }
                lastS[which] = S;
            }
            scan_build_rows<K>( rw, nearEnd, slot, sh.lenlut[t], sh.ring, p, sizeBits, lim, eobLo, eobHi, m == 1, lane, wave );

            /* chase: 32 + 16 + 2 symbols per group; x and the entries are byte offsets (2 x position) */
            const uint8_t* const J32 = slot.bytes();
            const uint8_t* const J16 = slot.bytes() + 2 * LEV_BYTES;
            const uint8_t* const J2 = slot.bytes() + LEV_BYTES;
            uint32_t x = 0, done = 0;
            bool stopAll = false;
            for ( uint32_t j = 0; j < m; ++j ) {
                const uint32_t q = *reinterpret_cast<const uint16_t*>( J32 + x );
                const uint32_t r16 = *reinterpret_cast<const uint16_t*>( J16 + q );
                const uint32_t u = sfl( *reinterpret_cast<const uint16_t*>( J2 + r16 ) );
                if ( u == 2 * S ) {                   /* left the span: again from here, the first group with the full span */
                    forceFull = j == 0;
                    break;
                }
                const uint32_t gg = g + j;
                gposV = lane == ( gg & 63u ) ? p + ( x >> 1 ) : gposV;
                if ( wave == 0 && ( gg & 63u ) == 63u ) gpos[( gg & ~63u ) + lane] = gposV;
                ++done;
                if ( u == 2 * S + 2 ) {               /* end-of-block, no code or end of input inside this group */
                    terminal = 1;
                    stopAll = true;
                    break;
                }
                {
                    const uint32_t d = ( u - x ) >> 1;
                    const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane( (int)estV, t );
                    const uint32_t upd = ( cur == 0 || d > cur ) ? d : cur - ( ( cur - d ) >> 2 );
                    estV = lane == t ? upd : estV;
                }
                x = u;
            }
            if ( ( ( g + done ) ^ g ) & ~63u ) {      /* next window of 64 selectors */
                selV = selPend;
                selPend = sel[( ( g + done ) & ~63u ) + 64 + lane];   /* may read past the block's selectors: never used */
            }
            g += done;
            p += x >> 1;
            if ( K > 1 ) which ^= 1u;
            if ( stopAll ) break;
        }
        nGroups = g;
        if ( wave == 0 && ( g & 63u ) != 0 && lane < ( g & 63u ) ) gpos[( g & ~63u ) + lane] = gposV;
    }

    if ( wave == 0 && lane == 0 ) {
        const uint32_t fullGroups = terminal ? nGroups - 1 : nGroups;
        BlockMeta mt;
        mt.enc_off = start;
        mt.enc_size = sh.hdr.enc_size;
        mt.decoded_size = 0;
        mt.out_off = 0;
        mt.header_crc = sh.hdr.header_crc;
        mt.computed_crc = 0xFFFFFFFFu;
        mt.n = 0;
        mt.orig_ptr = sh.hdr.orig_ptr;
        mt.nsym = fullGroups * GROUP_SYMS;    /* k_hsym's last lane finishes nsym, enc_size and status of a terminal group */
        mt.is_eos = sh.hdr.is_eos;
        mt.is_eof = sh.hdr.is_eof;
        mt.status = status;
        mt.seg_stride = MIN_SEG_STRIDE;
        mt.nseg = 0;
        mt.walk_ok = 0;
        mt.cycle_len = 0;
        mt.nchain = 0;
        mt.pad = 0;
        meta[b] = mt;
        HuffMeta hm;
        hm.n_stored = fullGroups * GROUP_SYMS;
        hm.symbol_count = sh.hdr.symbol_count;
        hm.status = status;
        hm.active = active;
        hmeta[b] = hm;
        ScanMeta sm;
        sm.pos_base = posBase;
        sm.size_bits = sizeBits;
        sm.n_groups = nGroups;
        sm.terminal = terminal;
        sm.symbol_count = sh.hdr.symbol_count;
        sm.pad[0] = sm.pad[1] = 0;
        smeta[b] = sm;
    }
    if ( K != 1 || queue == nullptr ) return;
    }
}

/* =============================================================================================================
 * k_hscan_spec: the scan with K waves per block that work on K CONSECUTIVE GROUPS at once.
 *
 * The chain from group to group is what makes a block slow when few blocks are decoded (one reader, one rank's share of a
 * file): a group's start is known only when the group in front of it has been measured.  But it is known APPROXIMATELY:
 * the groups of a table are about as long as its last ones (text: +- 12 %; incompressible data: 399 +- 1 bits).  So wave
 * w takes group g + w, lays a window over every position where that group can start -- the sum of the expected
 * lengths of the groups in front of it, minus and plus their observed deviations -- and computes, for EVERY start in
 * the window, where the group ends (J1 and the doubling rounds over the window plus one group, as in k_hscan, by the wave
 * alone and without barriers; then 16 + 16 + 16 + 2 symbols from each candidate start).  After one barrier the K groups
 * are chained with one LDS read each: the end of group w - 1 is the start of group w, if it lies in w's window.  If not
 * (a stray group, a table seen for the first time, the end of the input near), the unit ends there and the next one
 * starts at the true position: a guess decides how many groups a unit completes, never where a group starts.
 * Group g itself starts at a known position, so every unit completes at least one group.
 * ============================================================================================================= */
template<uint32_t K>
struct SpecShared
{
    static constexpr uint32_t RING = K > 8 ? 2 * SCAN_RING_ENTRIES : SCAN_RING_ENTRIES;   /* stream words around the unit */
    ScanShared<1> s;          /* tables, ring, header; its slot is wave 0's */
    ScanSlot more[K - 1];     /* the slots of waves 1 .. K - 1 */
    uint32_t wideRing[K > 8 ? 2 * ( RING + SCAN_RING_MIRROR ) : 2];   /* sixteen groups reach further than s.ring holds */

    __device__ __forceinline__ uint32_t* ring() { return K > 8 ? wideRing : s.ring; }

    __device__ __forceinline__ ScanSlot& slot( uint32_t w ) { return w == 0 ? s.slot[0] : more[w - 1]; }
};

/* bits in front of the current position that a unit may look: the stream ring holds 32 bits per entry, a build needs
 * 1 120 bits behind its start, refills come in pieces of 2 048 bits, and the current position must stay inside */
template<uint32_t K>
constexpr uint32_t SPEC_REACH = 32 * SpecShared<K>::RING - 4700;

template<uint32_t K>
__global__ __launch_bounds__( 64 * K ) void
k_hscan_spec( const uint32_t* __restrict__ in_words,
              uint64_t                     in_size_bytes,
              const uint64_t* __restrict__ offsets,
              BlockMeta* __restrict__      meta,
              HuffMeta* __restrict__       hmeta,
              ScanMeta* __restrict__       smeta,
              uint8_t*                     sel_buf,
              uint8_t* __restrict__        stb_buf,
              HuffTables* __restrict__     tab_buf,
              uint32_t* __restrict__       gpos_buf,
              uint32_t                     n_blocks,
              const uint32_t* __restrict__ order,
              uint32_t                     tune )   /* debugging: 1 = one group per unit */
{
    __shared__ SpecShared<K> shared;
    ScanShared<1>& sh = shared.s;
    const uint32_t slotIndex = blockIdx.x;
    if ( slotIndex >= n_blocks ) return;
    const uint32_t b = sfl( order[slotIndex] );
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = sfl( threadIdx.x >> 6 );
    uint8_t* const sel = sel_buf + (size_t)b * SEL_STRIDE;
    uint32_t* const gpos = gpos_buf + (size_t)b * GPOS_STRIDE;
    const uint64_t start = offsets[b];

    if ( wave == 0 ) {
        scan_parse<1>( sh, in_words, in_size_bytes, start, sel, stb_buf, tab_buf + b, b, lane );
        __threadfence_block();   /* the selectors are read back below */
    }
    __syncthreads();

    const uint32_t active = sfl( sh.hdr.active );
    int32_t status = (int32_t)sfl( (uint32_t)sh.hdr.status );
    const uint32_t nSel = sfl( sh.hdr.n_sel );
    const uint32_t sizeBits = sfl( sh.hdr.size_bits );
    const uint32_t nWords = sfl( sh.hdr.n_words );
    const uint64_t posBase = ( (uint64_t)sfl( (uint32_t)( sh.hdr.pos_base >> 32 ) ) << 32 ) | sfl( (uint32_t)sh.hdr.pos_base );
    uint32_t nGroups = 0, terminal = 0;

    if ( active ) {
        const uint32_t* const words = in_words + ( posBase >> 5 );
        ScanSlot& mySlot = shared.slot( wave );
        uint32_t p = sfl( sh.hdr.p0 );
        uint32_t g = 0;
        /* stream ring: words [wHi - 256, wHi) are in sh.ring.  A unit takes up to 3 400 bits, more than one refill of 64
         * words: the waves take turns, wave w holds the words of the next piece q (64 words) with q mod K == w, loaded K
         * refills ahead */
        uint32_t wHi = 0, pend = 0;
        if ( 64 * wave + lane < nWords ) pend = words[64 * wave + lane];
        /* selectors of groups [64 k, 64 k + 64), one per lane, and of the next 64 */
        uint32_t selV = sel[lane];
        uint32_t selNext = sel[64 + lane];
        /* per table (lane t = table t): the largest recent group length (0: table not seen yet), a running mean of the
         * lengths and the largest recent deviation from it */
        uint32_t estV = 0, midV = 0, devV = 0;
        uint32_t gposV = 0;
        uint32_t lastS = 0, lastT = 0xFFFFFFFFu;     /* of this wave's slot / of this wave's code ranges */
        uint32_t lim[10] = {};
        uint32_t eobLo = 0, eobHi = 0;
        bool forceFull = false;

        for ( ;; ) {
            if ( g >= nSel ) { status = ST_SELECTOR_OVERRUN; break; }
            if ( g >= MAX_SCAN_GROUPS ) { status = ST_DATA_OVERFLOW; break; }
            /* ---- the plan of the unit, the same in every wave: group g + w in slot w for w < n; lane w plans slot w ---- */
            uint32_t tV, loV, widthV, rowsV;     /* lane w: table, start of the window relative to p, its width, rows of the build */
            uint32_t n;
            {
                const uint32_t gw = g + lane;
                const uint32_t fromNext = ( gw >> 6 ) != ( g >> 6 ) ? 1u : 0u;
                const uint32_t tHere = (uint32_t)__shfl( (int)selV, (int)( gw & 63u ) );
                const uint32_t tNext = (uint32_t)__shfl( (int)selNext, (int)( gw & 63u ) );
                tV = ( fromNext ? tNext : tHere ) & 7u;
                const uint32_t est = (uint32_t)__shfl( (int)estV, (int)tV );
                const uint32_t mid = (uint32_t)__shfl( (int)midV, (int)tV );
                const uint32_t dev = (uint32_t)__shfl( (int)devV, (int)tV );
                const uint32_t need = est + ( est >> 3 ) + 16;       /* a group of this table: largest recent + 12 % + 16 bits */
                const uint32_t slack = dev + ( dev >> 2 ) + 2;
                const uint32_t stepLo = mid > slack + 50 ? mid - slack : 50u;      /* 50 symbols are at least 50 bits */
                const uint32_t stepHi = mid + slack;
                /* exclusive prefix sums over the first lanes (K <= 16: inside one row of 16 lanes) */
                uint32_t accLo = stepLo, accHi = stepHi;
#define SPEC_SCAN_STEP( k ) \
                if constexpr ( ( k ) < K ) { \
                    accLo += (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)accLo, 0x110 + ( k ), 0xF, 0xF, true );   /* row_shr:k, 0 shifted in */ \
                    accHi += (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)accHi, 0x110 + ( k ), 0xF, 0xF, true ); \
                }
                SPEC_SCAN_STEP( 1 ) SPEC_SCAN_STEP( 2 ) SPEC_SCAN_STEP( 4 ) SPEC_SCAN_STEP( 8 )
#undef SPEC_SCAN_STEP
                accLo -= stepLo;
                accHi -= stepHi;
                loV = accLo;
                widthV = accHi - accLo;
                rowsV = scan_rows_per_wave<1>( ( widthV + need + 24 + 63 ) >> 6 );
                const bool ok = ( lane < K ) && ( gw < nSel ) && ( gw < MAX_SCAN_GROUPS ) && !( ( tune & 1u ) && lane > 0 )
                                && ( est != 0 ) && ( widthV + need + 24 <= SCAN_MAX_SPAN ) && ( accHi <= SPEC_REACH<K> )
                                && ( p + accLo + SCAN_MAX_SPAN + 32 <= sizeBits );
                const uint64_t okMask = __ballot( ok );
                n = (uint32_t)__builtin_ctzll( ~okMask );            /* slots 0 .. n - 1 are usable */
                if ( forceFull ) n = 0;
            }
            const bool full = n == 0;     /* first group of a table, near the end of the input, or a group that left its span */
            const bool nearEnd = p + SCAN_MAX_SPAN + 32 > sizeBits;
            if ( full ) {
                n = 1;
                tV = (uint32_t)__builtin_amdgcn_readlane( (int)selV, g & 63u );
                loV = 0; widthV = 0; rowsV = SCAN_ROWS;
            }
            /* ---- stream words up to the end of the farthest build (every wave keeps count, wave 0 moves the words) ---- */
            {
                const uint32_t far = (uint32_t)__builtin_amdgcn_readlane( (int)loV, n - 1 );
                const uint32_t needWord = ( ( p + far ) >> 5 ) + ( SCAN_MAX_SPAN + 96 ) / 32;
                if ( needWord > wHi ) {
                    while ( needWord > wHi ) {
                        if ( ( ( wHi >> 6 ) & ( K - 1 ) ) == wave ) {
                            ring_put<SpecShared<K>::RING>( shared.ring(), wHi + lane, pend );
                            pend = 0;
                            if ( wHi + 64 * K + lane < nWords ) pend = words[wHi + 64 * K + lane];
                        }
                        wHi += 64;
                    }
                    __syncthreads();
                }
            }
            /* ---- every wave builds its slot ---- */
            if ( wave < n ) {
                const uint32_t myT = (uint32_t)__builtin_amdgcn_readlane( (int)tV, wave );
                const uint32_t myLo = (uint32_t)__builtin_amdgcn_readlane( (int)loV, wave );
                const uint32_t myWidth = (uint32_t)__builtin_amdgcn_readlane( (int)widthV, wave );
                const uint32_t myRows = (uint32_t)__builtin_amdgcn_readlane( (int)rowsV, wave );
                if ( myT != lastT ) {
                    const uint32_t mx = sfl( sh.minmax[myT] ) >> 8;
                    const uint32_t limV = sh.limit[myT][lane < mx ? lane : mx];
#pragma unroll
                    for ( uint32_t l = 0; l < 10; ++l ) lim[l] = (uint32_t)__builtin_amdgcn_readlane( (int)limV, 11 + l );
                    eobLo = sfl( sh.eob_lo[myT] );
                    eobHi = sfl( sh.eob_hi[myT] );
                    lastT = myT;
                }
                const uint32_t S = 64 * myRows;
                if ( S != lastS ) {
                    if ( lane < 2 ) {
                        const uint16_t v = (uint16_t)( 2 * ( S + lane ) );
                        mySlot.at( 0, S + lane ) = v; mySlot.at( 1, S + lane ) = v; mySlot.at( 2, S + lane ) = v;
                    }
                    lastS = S;
                }
                scan_build_rows<1, SCAN_ENDS, SpecShared<K>::RING>( myRows, full && nearEnd, mySlot, sh.lenlut[myT], shared.ring(), p + myLo, sizeBits, lim,
                                               eobLo, eobHi, false, lane, 0, ( myWidth >> 6 ) + 1, myLo );
            }
            __syncthreads();
            /* ---- chain the groups: every wave follows the same chain (uniform LDS reads).  A slot's candidates hold the end
             * of their group relative to p (0xFFFF: beyond the span, 0xFFFE: the group stops the block); lane w keeps the
             * end of group w ---- */
            uint32_t rel = 0, measured = 0;
            uint32_t endV = 0;
            bool stopAll = false;
            forceFull = false;
            {
                uint32_t last = 0;      /* what ended the chain: 0xFFFF, 0xFFFE or nothing special */
#pragma unroll 1
                for ( ; measured < n; ++measured ) {
                    const uint32_t y = rel - (uint32_t)__builtin_amdgcn_readlane( (int)loV, measured );   /* wraps in front of the window */
                    if ( y > (uint32_t)__builtin_amdgcn_readlane( (int)widthV, measured ) ) break;        /* outside (never for slot 0) */
                    const uint32_t u = sfl( shared.slot( measured ).at( 0, y ) );
                    if ( u >= 0xFFFEu ) {
                        last = u;
                        break;
                    }
                    rel = u;
                    endV = lane == measured ? u : endV;
                }
                forceFull = ( last == 0xFFFFu ) && ( measured == 0 );     /* a group longer than expected: the full span next */
                stopAll = last == 0xFFFEu;                                /* end-of-block, no code or end of input inside this group */
            }
            if ( stopAll ) terminal = 1;
            const uint32_t done = measured + ( stopAll ? 1u : 0u );
            /* starts of the `done` groups (lane w: group g + w) -> the lanes of their group numbers; the window of 64 is
             * written out when its last group is known, before the next window's first groups take its lanes */
            uint32_t dV;
            {
                const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)endV, 0x111, 0xF, 0xF, true );   /* row_shr:1 */
                const uint32_t startV = lane == 0 ? 0u : before;
                dV = endV - startV;
                const uint32_t idx = ( lane - g ) & 63u;                  /* which of the unit's groups this lane stands for */
                const uint32_t mine = p + (uint32_t)__shfl( (int)startV, (int)idx );
                const uint32_t inWindow = 64u - ( g & 63u );              /* groups of the unit that still belong to g's window */
                const uint32_t part = done < inWindow ? done : inWindow;
                gposV = idx < part ? mine : gposV;
                if ( done >= inWindow ) {
                    if ( wave == 0 ) gpos[( g & ~63u ) + lane] = gposV;
                    gposV = ( idx >= part && idx < done ) ? mine : gposV;
                }
            }
            const uint32_t cur = p + rel;
            /* ---- what the measured groups say about their tables (lane t = table t), once per unit: the longest and the
             * shortest group of the table in this unit ---- */
            {
                uint32_t dMax = 0, dMin = 0xFFFFFFFFu;
#pragma unroll 1
                for ( uint32_t w = 0; w < measured; ++w ) {
                    const uint32_t tw = (uint32_t)__builtin_amdgcn_readlane( (int)tV, w );
                    const uint32_t dw = (uint32_t)__builtin_amdgcn_readlane( (int)dV, w );
                    dMax = ( lane == tw && dw > dMax ) ? dw : dMax;
                    dMin = ( lane == tw && dw < dMin ) ? dw : dMin;
                }
                if ( dMax != 0 ) {
                    const bool first = estV == 0;      /* says nothing about the spread yet: assume the envelope's 12 % */
                    const uint32_t midOld = first ? dMax : midV;
                    const uint32_t offHi = dMax > midOld ? dMax - midOld : midOld - dMax;
                    const uint32_t offLo = dMin > midOld ? dMin - midOld : midOld - dMin;
                    const uint32_t off = offHi > offLo ? offHi : offLo;
                    const int32_t centre = (int32_t)( ( dMax + dMin ) >> 1 );
                    estV = ( first || dMax > estV ) ? dMax : estV - ( ( estV - dMax ) >> 2 );
                    midV = first ? dMax : (uint32_t)( (int32_t)midV + ( centre - (int32_t)midV ) / 2 );
                    devV = first ? ( dMax >> 3 ) + 8 : ( off > devV ? off : devV - ( ( devV - off + 3 ) >> 2 ) );
                }
            }
            if ( ( ( g + done ) ^ g ) & ~63u ) {      /* next window of 64 selectors */
                selV = selNext;
                selNext = sel[( ( g + done ) & ~63u ) + 64 + lane];   /* may read past the block's selectors: never used */
            }
            g += done;
            p = cur;
            __syncthreads();       /* the slots are rebuilt next */
            if ( stopAll ) break;
        }
        nGroups = g;
        if ( wave == 0 && ( g & 63u ) != 0 && lane < ( g & 63u ) ) gpos[( g & ~63u ) + lane] = gposV;
    }

    if ( wave == 0 && lane == 0 ) {
        const uint32_t fullGroups = terminal ? nGroups - 1 : nGroups;
        BlockMeta mt;
        mt.enc_off = start;
        mt.enc_size = sh.hdr.enc_size;
        mt.decoded_size = 0;
        mt.out_off = 0;
        mt.header_crc = sh.hdr.header_crc;
        mt.computed_crc = 0xFFFFFFFFu;
        mt.n = 0;
        mt.orig_ptr = sh.hdr.orig_ptr;
        mt.nsym = fullGroups * GROUP_SYMS;    /* k_hsym's last lane finishes nsym, enc_size and status of a terminal group */
        mt.is_eos = sh.hdr.is_eos;
        mt.is_eof = sh.hdr.is_eof;
        mt.status = status;
        mt.seg_stride = MIN_SEG_STRIDE;
        mt.nseg = 0;
        mt.walk_ok = 0;
        mt.cycle_len = 0;
        mt.nchain = 0;
        mt.pad = 0;
        meta[b] = mt;
        HuffMeta hm;
        hm.n_stored = fullGroups * GROUP_SYMS;
        hm.symbol_count = sh.hdr.symbol_count;
        hm.status = status;
        hm.active = active;
        hmeta[b] = hm;
        ScanMeta sm;
        sm.pos_base = posBase;
        sm.size_bits = sizeBits;
        sm.n_groups = nGroups;
        sm.terminal = terminal;
        sm.symbol_count = sh.hdr.symbol_count;
        smeta[b] = sm;
    }
}

/* ============================================================================================================= */

template<uint32_t THREADS>
struct alignas( 16 ) SymShared
{
    HuffTables tabs;
    uint16_t stage[THREADS * GROUP_SYMS];
};

/* THREADS = groups per workgroup: the block's tables (17 KB) are loaded once per workgroup.  LDS declared at launch, see
 * k_hscan: the compiler then takes the 26 registers the kernel needs instead of the 129 that its LDS-bound occupancy allows. */
template<uint32_t THREADS = SYM_THREADS>
__global__ __launch_bounds__( THREADS ) void
k_hsym( const uint32_t* __restrict__   in_words,
        BlockMeta* __restrict__        meta,
        HuffMeta* __restrict__         hmeta,
        const ScanMeta* __restrict__   smeta,
        const uint8_t* __restrict__    sel_buf,
        const HuffTables* __restrict__ tab_buf,
        const uint32_t* __restrict__   gpos_buf,
        uint16_t* __restrict__         sym_buf )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t ldsAtLaunch[];     /* sizeof( SymShared<THREADS> ) */
    SymShared<THREADS>& sh = *reinterpret_cast<SymShared<THREADS>*>( ldsAtLaunch );
    const uint32_t b = blockIdx.y;
    const ScanMeta sm = smeta[b];
    if ( blockIdx.x * THREADS * SYM_CHUNKS >= sm.n_groups ) return;
    const uint32_t tid = threadIdx.x;
    {
        const uint4* const src = reinterpret_cast<const uint4*>( tab_buf + b );
        uint4* const dst = reinterpret_cast<uint4*>( &sh.tabs );
        for ( uint32_t k = tid; k < sizeof( HuffTables ) / 16; k += THREADS ) dst[k] = src[k];
    }
    __syncthreads();

    /* SYM_CHUNKS times THREADS groups per workgroup, one chunk after the other: the tables are loaded once, and a workgroup
     * costs about 11 ns of dispatch whatever it does (182 000 / 91 000 / 45 000 workgroups: 4.5 / 3.5 / 3.05 ms) */
    for ( uint32_t chunk = 0; chunk < SYM_CHUNKS; ++chunk ) {
    const uint32_t g0 = ( blockIdx.x * SYM_CHUNKS + chunk ) * THREADS;
    if ( g0 >= sm.n_groups ) return;
    const uint32_t gi = g0 + tid;
    const uint32_t eob = sm.symbol_count + 1;
    if ( gi < sm.n_groups ) {
        const uint32_t t = sel_buf[(size_t)b * SEL_STRIDE + gi];
        const uint32_t* const words = in_words + ( sm.pos_base >> 5 );
        uint32_t pos = gpos_buf[(size_t)b * GPOS_STRIDE + gi];
        const bool last = sm.terminal && gi + 1 == sm.n_groups;
        /* bit buffer: `have` valid bits left-aligned in buf */
        uint32_t w = pos >> 5;
        uint64_t buf = ( ( (uint64_t)be32( words[w] ) << 32 ) | be32( words[w + 1] ) ) << ( pos & 31u );
        uint32_t have = 64 - ( pos & 31u );
        w += 2;
        const uint16_t* const lut = sh.tabs.lut[t];
        const uint32_t mm = sh.tabs.minmax[t];
        const uint32_t maxLen = mm >> 8;
        uint16_t* const out = sh.stage + tid * GROUP_SYMS;
        uint32_t cnt = 0;
        int32_t status = ST_OK;
        bool finished = false;
        for ( uint32_t j = 0; j < GROUP_SYMS; ++j ) {
            if ( have <= 32 ) {
                buf |= (uint64_t)be32( words[w] ) << ( 32 - have );   /* the input copy is zero padded */
                have += 32;
                ++w;
            }
            uint32_t e = lut[(uint32_t)( buf >> ( 64 - SCAN_LUT_BITS ) )];
            uint32_t len = e & 31u, sym = e >> 5;
            if ( len == 0 ) {
                const uint32_t v20 = (uint32_t)( buf >> 44 );
                for ( uint32_t l = SCAN_LUT_BITS + 1; l <= maxLen; ++l ) {
                    const uint32_t d = ( v20 >> ( 20 - l ) ) - sh.tabs.first[t][l];
                    if ( d < sh.tabs.count[t][l] ) {
                        len = l;
                        sym = sh.tabs.perm[t][sh.tabs.offs[t][l] + d];
                        break;
                    }
                }
            }
            if ( last ) {
                /* the rules of the bit reader and of the decoder at the end of the input (oracle: huff_decode) */
                if ( len == 0 ) {
                    status = pos + maxLen > sm.size_bits ? ST_EOF : ST_INVALID_CODE;
                    break;
                }
                if ( pos + len > sm.size_bits ) {
                    status = ST_EOF;
                    break;
                }
            }
            pos += len;
            buf <<= len;
            have -= len;
            if ( last && sym == eob ) {
                finished = true;
                break;
            }
            out[j] = (uint16_t)sym;
            ++cnt;
        }
        if ( last ) {
            const uint32_t stored = gi * GROUP_SYMS + cnt;
            if ( !finished && status == ST_OK ) status = ST_INVALID_CODE;   /* unreachable: k_hscan saw the group end early */
            hmeta[b].n_stored = stored;
            hmeta[b].status = status;
            meta[b].nsym = stored + ( finished ? 1u : 0u );
            meta[b].status = status;
            meta[b].enc_size = sm.pos_base + pos - meta[b].enc_off;
        }
    }
    __syncthreads();
    /* stage -> memory: the groups of a workgroup are one contiguous piece of the symbol buffer */
    {
        const uint32_t nHere = sm.n_groups - g0 < THREADS ? sm.n_groups - g0 : THREADS;
        const uint32_t units = ( nHere * GROUP_SYMS * 2 + 15 ) / 16;   /* the symbol buffer is padded */
        uint4* const dst = reinterpret_cast<uint4*>( sym_buf + (size_t)b * SYM_STRIDE + (size_t)g0 * GROUP_SYMS );
        const uint4* const src = reinterpret_cast<const uint4*>( sh.stage );
        for ( uint32_t k = tid; k < units; k += THREADS ) dst[k] = src[k];
    }
    __syncthreads();     /* the stage is written again by the next chunk */
    }
}
}  // namespace bz2gpu
