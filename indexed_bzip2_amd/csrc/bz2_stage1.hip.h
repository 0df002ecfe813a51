/**
 * bz2_stage1.hip.h -- stage 1 of the block decoder split in two kernels (replaces the fused v1 k_stage1).
 *
 *   k_huff  one wavefront per block.  Header/selector/code-length parse as before, then the Huffman bit stream is
 *           decoded a WINDOW at a time: every lane looks up the code that would start at "its" bit (64 consecutive
 *           bit positions, LUTs in LDS), and the true code boundaries are found by following the chain
 *           0 -> len -> len' ... with v_readlane on the scalar unit.  The LUT has a second 16-bit half per entry that
 *           describes ALL codes wholly inside the 10 index bits (start mask + total advance), so one scalar step
 *           usually skips 2-3 symbols.  Symbols (u16) go to HBM; no MTF here.
 *           Reference: Block::readBlockData's symbol loop + HuffmanCodingShortBitsCached::decode
 *           (src/indexed_bzip2/bzip2.hpp:709-723, src/huffman/HuffmanCodingShortBitsCached.hpp:98-150).
 *
 *   k_mtf   128 lanes per block, each owning a contiguous chunk of the symbol stream: RUNA/RUNB run lengths and
 *           move-to-front (bzip2.hpp:726-790).  MTF is order dependent, so pass A runs every chunk on an identity
 *           list (giving the chunk's permutation and output size), the permutations are composed in chunk order to
 *           get each chunk's true start list, and pass B replays the chunk with it and writes the L column.
 *           Each lane keeps its 256-entry list in LDS (64 dwords, lane-interleaved: conflict free).
 */
#pragma once

#include "bz2_kernels.hip.h"

namespace bz2gpu
{
constexpr uint32_t SYM_STRIDE = 900224;       /* u16 symbols per block (n_sym <= N + 1 <= 900001 for valid blocks) */
constexpr uint32_t SYM_CAP = 900096;
constexpr uint32_t MTF_THREADS = 256;      /* lanes (= chunks) per block in k_mtf: one workgroup per block */
constexpr uint32_t MTF_SMALL_STRIDE = 144;  /* lists of 128 entries for blocks with few symbols, see k_mtf */
constexpr uint32_t MTF_LANE_STRIDE = 272;  /* bytes between the lists of consecutive lanes: 256 + 16, so that the 16-byte
                                              accesses of the 16 lanes served together fall on 64 distinct banks */
constexpr uint32_t CHAIN_END = 54;      /* a chain step starts below this bit of the 64-bit window: the 10 start bits of
                                           an entry then stay inside the window (the asm below spells the number out) */
/* one branch-free step of the code chain in k_huff; operands as in the asm statements there */
#define HUFF_CHAIN_STEP \
    "v_readlane_b32 s96, %[M], %[cur]\n\t" \
    "s_lshr_b32 %[adv], s96, 10\n\t" \
    "s_and_b64 s[98:99], s[96:97], 0x3ff\n\t" \
    "s_lshl_b64 s[98:99], s[98:99], %[cur]\n\t" \
    "s_or_b64 %[mask], %[mask], s[98:99]\n\t" \
    "s_add_u32 %[cur], %[cur], %[adv]\n\t"
constexpr uint32_t HUFF_RING = 1024;    /* symbols in the LDS staging ring of k_huff; flushed in halves */
constexpr uint32_t HUFF_WAVES = 1;      /* independent blocks (one per wavefront) per k_huff workgroup */

/** Orders LDS traffic between the lanes of ONE wavefront (no s_barrier: the waves of a k_huff workgroup are independent). */
__device__ __forceinline__ void
wave_sync()
{
    __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" );
    __builtin_amdgcn_wave_barrier();
    asm volatile( "" ::: "memory" );   /* LDS traffic of one wave is in order in hardware: the compiler has to keep it so */
}

/** Per-block hand-off between k_huff and k_mtf. */
struct HuffMeta
{
    uint32_t n_stored;      /* symbols written to sym_buf (without the end-of-block symbol) */
    uint32_t symbol_count;  /* bzip2 symbolCount (used byte values) */
    int32_t  status;        /* first error of the Huffman stage (ST_OK if the end-of-block symbol was reached) */
    uint32_t active;        /* 1 if k_mtf has work (data block whose header parsed) */
};

struct HuffShared
{
    uint32_t lut[6][1 << LUT_BITS];   /* lo16: {len:5, sym:9} of the code starting here (0: longer than LUT_BITS / none)
                                         hi16: {mask:10 starts of all codes inside the index bits, adv:6 their total length} */
    uint16_t perm[6][260];
    uint32_t first[6][24];
    uint32_t count[6][24];
    uint32_t offs[6][24];
    uint32_t running[24];
    uint32_t limit[6][24];            /* [t][l], l in (LUT_BITS, 20]: (first + count) << (20 - l), else 0 */
    union alignas( 16 ) {
        uint8_t  lens[6][264];          /* code lengths, only while the tables are built */
        uint16_t ring[HUFF_RING];       /* decoded symbols on their way to memory, only in the symbol loop */
    };
    uint16_t bitmap[16];
    uint8_t  sym_to_byte[256];
    uint32_t minmax[6];
};

/** Symbols [base, base + count) of the staging ring -> memory, eight per lane (one 16-byte store).  Storing every
 * window's few symbols straight to memory would put a store in front of every wait for the prefetched stream words:
 * s_waitcnt vmcnt counts loads and stores together, so each window would sit out a store acknowledge. */
__device__ __forceinline__ void
huff_flush( const HuffShared& sh, uint16_t* symOut, uint32_t base, uint32_t count, uint32_t lane )
{
    wave_sync();
    if ( 8 * lane < count ) {
        const uint4 v = *reinterpret_cast<const uint4*>( &sh.ring[( base + 8 * lane ) & ( HUFF_RING - 1 )] );
        *reinterpret_cast<uint4*>( symOut + base + 8 * lane ) = v;
    }
}

/** Huffman stage of ONE block by one wavefront (see k_huff). */
__device__ __forceinline__ void
huff_block( HuffShared&                  sh,
            const uint32_t* __restrict__ in_words,
            uint64_t                     in_size_bytes,
            const uint64_t* __restrict__ offsets,
            BlockMeta* __restrict__      meta,
            HuffMeta* __restrict__       hmeta,
            uint8_t*                     sel_buf,
            uint16_t* __restrict__       sym_buf,
            uint8_t* __restrict__        stb_buf,
            const uint32_t* __restrict__ order,
            uint32_t                     slot )
{
    const uint32_t b = sfl( order[slot] );
    const uint32_t lane = threadIdx.x & 63;
    uint8_t* const sel = sel_buf + (size_t)b * SEL_STRIDE;
    uint16_t* const symOut = sym_buf + (size_t)b * SYM_STRIDE;

    const uint64_t start = offsets[b];
    BitRd br;
    br.init( in_words, in_size_bytes, start );

    int32_t status = ST_OK;
    uint32_t headerCrc = 0, origPtr = 0, nsym = 0, cnt = 0;
    int32_t isEos = 0, isEof = 0;
    uint64_t encSize = 0;
    uint32_t symbolCount = 0, groupCount = 0, nSel = 0;
    uint32_t active = 0;

#define FAIL( code ) do { status = br.eof ? (int32_t)ST_EOF : (int32_t)( code ); goto finish; } while ( 0 )

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
            goto finish;
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
        for ( uint32_t v = lane; v < 256; v += 64 ) sh.sym_to_byte[v] = 0;   /* fresh Block: zero-initialised */
        for ( int i = 0; i < 16; ++i ) {
            uint32_t bm = 0;
            if ( used & ( 1u << ( 15 - i ) ) ) {
                bm = br.read( 16 );
            }
            if ( lane == 0 ) sh.bitmap[i] = (uint16_t)bm;
        }
        wave_sync();
        uint32_t total = 0;
        for ( int g = 0; g < 16; ++g ) total += __popc( sh.bitmap[g] );
        symbolCount = total;
        for ( uint32_t v = lane; v < 256; v += 64 ) {
            const uint32_t g = v >> 4, j = v & 15;
            const uint32_t bm = sh.bitmap[g];
            if ( bm & ( 1u << ( 15 - j ) ) ) {
                uint32_t rank = 0;
                for ( uint32_t gg = 0; gg < g; ++gg ) rank += __popc( sh.bitmap[gg] );
                rank += j == 0 ? 0 : __popc( bm >> ( 16 - j ) );
                sh.sym_to_byte[rank] = (uint8_t)v;
            }
        }
        wave_sync();
        if ( br.eof ) FAIL( ST_EOF );
        reinterpret_cast<uint32_t*>( stb_buf + (size_t)b * 256 )[lane] =
            reinterpret_cast<const uint32_t*>( sh.sym_to_byte )[lane];
    }

    /* ---- Block::readSelectors, bzip2.hpp:574-637 ---- */
    {
        groupCount = br.read( 3 );
        if ( br.eof ) FAIL( ST_EOF );
        if ( groupCount < 2 || groupCount > 6 ) FAIL( ST_GROUP_COUNT );
        nSel = br.read( 15 );
        if ( br.eof ) FAIL( ST_EOF );
        if ( nSel == 0 ) FAIL( ST_SELECTOR_COUNT );
        uint32_t mtfsel = 0x543210u;   /* nibble k = entry k */
        uint32_t packed = 0;
        for ( uint32_t i = 0; i < nSel; ++i ) {
            br.refill();
            if ( br.pos + 6 > br.size_bits ) {   /* peek<6> throws at EOF, BitReader.hpp:458-460 */
                br.eof = true;
                FAIL( ST_EOF );
            }
            const uint32_t bits6 = br.peek( 6 );
            const uint32_t j = __clz( ~( bits6 << 26 ) );   /* leading ones, 6 if all set */
            br.skip( j + 1 );
            if ( j >= groupCount ) FAIL( ST_SELECTOR_UNARY );
            const uint32_t shj = 4 * j;
            const uint32_t val = ( mtfsel >> shj ) & 0xFu;
            const uint32_t low = mtfsel & ( ( 1u << shj ) - 1u );
            const uint32_t highMask = ~( ( 16u << shj ) - 1u );
            mtfsel = ( mtfsel & highMask ) | ( low << 4 ) | val;
            packed |= val << ( 8 * ( i & 3 ) );
            if ( ( i & 3 ) == 3 || i + 1 == nSel ) {
                if ( lane == 0 ) *reinterpret_cast<uint32_t*>( sel + ( i & ~3u ) ) = packed;
                packed = 0;
            }
        }
    }

    /* ---- Block::readTrees, bzip2.hpp:644-685, and the canonical tables ---- */
    {
        const uint32_t symCount = symbolCount + 2;
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
                if ( lane == 0 ) sh.lens[t][s] = (uint8_t)hh;
            }
            /* the reference builds (and checks) the coding of a group before it reads the next group's lengths
             * (bzip2.hpp:679-683): an over-subscribed set here wins over a bad length further on */
            wave_sync();
            uint32_t c = 0;
            if ( lane >= 1 && lane <= 20 ) {
                for ( uint32_t s = 0; s < symCount; ++s ) c += sh.lens[t][s] == lane;
            }
            if ( lane < 24 ) sh.count[t][lane] = c;
            wave_sync();
            {
                uint32_t minLen = 0, maxLen = 0;
                for ( uint32_t l = 1; l <= 20; ++l ) {
                    if ( sh.count[t][l] != 0 ) {
                        if ( minLen == 0 ) minLen = l;
                        maxLen = l;
                    }
                }
                uint32_t unused = 1u << minLen;
                bool bad = false;
                for ( uint32_t l = minLen; l <= maxLen; ++l ) {
                    const uint32_t f = sh.count[t][l];
                    if ( f > unused ) { bad = true; break; }
                    unused = ( unused - f ) * 2u;
                }
                if ( bad ) FAIL( ST_HUFFMAN_LENGTHS );
            }
        }
        wave_sync();

        for ( uint32_t t = 0; t < groupCount; ++t ) {
            uint32_t minLen = 0, maxLen = 0;
            for ( uint32_t l = 1; l <= 20; ++l ) {
                if ( sh.count[t][l] != 0 ) {
                    if ( minLen == 0 ) minLen = l;
                    maxLen = l;
                }
            }
            if ( lane == 0 ) {
                uint32_t minCode = 0, sum = 0;
                for ( uint32_t l = 0; l < 24; ++l ) { sh.first[t][l] = 0; sh.offs[t][l] = 0; }
                for ( uint32_t l = minLen; l <= maxLen; ++l ) {
                    minCode = ( minCode + ( l > minLen ? sh.count[t][l - 1] : 0u ) ) << 1;
                    if ( l == minLen ) minCode = 0;
                    sh.first[t][l] = minCode;
                    sh.offs[t][l] = sum;
                    sh.running[l] = sum;
                    sum += sh.count[t][l];
                }
                sh.minmax[t] = minLen | ( maxLen << 8 );
            }
            wave_sync();
            for ( uint32_t base = 0; base < symCount; base += 64 ) {
                const uint32_t s = base + lane;
                const bool valid = s < symCount;
                const uint32_t len = valid ? sh.lens[t][s] : 0u;
                const uint64_t same = match_any( len, 5, valid );
                const uint32_t rank = popc_below( same, lane );
                uint32_t basePos = 0;
                if ( valid ) basePos = sh.running[len];
                if ( valid ) sh.perm[t][basePos + rank] = (uint16_t)s;
                wave_sync();
                if ( valid && rank == 0 ) sh.running[len] = basePos + (uint32_t)__popcll( same );
                wave_sync();
            }
            if ( lane < 24 ) {
                sh.limit[t][lane] = ( lane > (uint32_t)LUT_BITS && lane <= 20 )
                                    ? ( ( sh.first[t][lane] + sh.count[t][lane] ) << ( 20 - lane ) ) : 0u;
            }
            /* single-symbol half */
            const uint32_t lutMax = maxLen < (uint32_t)LUT_BITS ? maxLen : (uint32_t)LUT_BITS;
            for ( uint32_t e = lane; e < ( 1u << LUT_BITS ); e += 64 ) {
                uint32_t val = 0;
                for ( uint32_t l = minLen; l <= lutMax; ++l ) {
                    const uint32_t code = e >> ( LUT_BITS - l );
                    const uint32_t d = code - sh.first[t][l];
                    if ( d < sh.count[t][l] ) {
                        val = l | ( (uint32_t)sh.perm[t][sh.offs[t][l] + d] << 5 );
                        break;
                    }
                }
                sh.lut[t][e] = val;
            }
            wave_sync();
            /* multi-symbol half: all codes that lie completely inside the LUT_BITS index bits */
            for ( uint32_t e = lane; e < ( 1u << LUT_BITS ); e += 64 ) {
                uint32_t p = 0, mask = 0;
                while ( p < (uint32_t)LUT_BITS ) {
                    const uint32_t idx = ( e << p ) & ( ( 1u << LUT_BITS ) - 1u );
                    const uint32_t single = sh.lut[t][idx] & 0xFFFFu;
                    const uint32_t len = single & 31u;
                    /* the end-of-block symbol is never part of a step: the chain stops in front of it */
                    if ( len == 0 || p + len > (uint32_t)LUT_BITS || ( single >> 5 ) == symbolCount + 1 ) break;
                    mask |= 1u << p;
                    p += len;
                }
                const uint32_t multi = mask | ( p << 10 );
                /* all single halves are final before any entry is rewritten (barrier above); OR keeps the low half */
                atomicOr( &sh.lut[t][e], multi << 16 );
            }
            wave_sync();
        }
    }
    active = 1;

    /* ---- symbol loop of Block::readBlockData, bzip2.hpp:709-723, window-parallel ---- */
    {
        /* Bit positions inside the loop are 32-bit and relative to the word that holds the first symbol bit: a block
         * ends long before 2^32 bits (900 096 symbols of at most 20 bits), and an input that is longer than that from
         * here is clamped, which only moves the "end of input" checks out of reach. */
        const uint64_t posBase = br.pos & ~31ull;
        const uint32_t* const words = in_words + ( posBase >> 5 );
        uint32_t pos = (uint32_t)( br.pos - posBase );
        const uint32_t sizeBits = br.size_bits - posBase < 0xFFFF0000ull ? (uint32_t)( br.size_bits - posBase ) : 0xFFFF0000u;
        const uint32_t safeEnd = sizeBits > 256 ? sizeBits - 256 : 0;   /* below this no code can cross the end */
        const uint32_t fastEnd = safeEnd > 1100 ? safeEnd - 1100 : 0;   /* groups opened below this are "fast" */
        const uint32_t eob = symbolCount + 1;
        uint32_t groupLeft = 0, fastLeft = 0, selIdx = 0, tcur = 0;
        uint32_t limitV = 0;   /* lane l in (LUT_BITS, 20]: left-aligned (20 bit) end of the length-l code range */
        bool finished = false;

        /* selectors are fetched 8 at a time, one fetch ahead of their use */
        const uint64_t* const sel64 = reinterpret_cast<const uint64_t*>( sel );
        uint64_t selCur = 0;
        uint64_t selNext = sel64[0];

        /* Each lane keeps the four stream words that start at the word of ITS bit position, loaded one window ahead:
         * a window advances by at most 64 bits, so the words needed next are among them. */
        uint32_t myWord = ( pos + lane ) >> 5;
        const auto load4 = [words] ( uint32_t at ) {
            uint4 v = *reinterpret_cast<const uint4*>( words + at );
            v.x = be32( v.x ); v.y = be32( v.y ); v.z = be32( v.z ); v.w = be32( v.w );
            return v;
        };
        uint4 D = load4( myWord );

#ifdef MI355X_BZ2_HUFF_PROFILE
        uint64_t profSetup = 0, profChain = 0, profCommit = 0, profWindows = 0, profRefresh = 0, profGeneral = 0, profGeneralCycles = 0;
#define HUFF_PROF_NOW() __builtin_readcyclecounter()
#endif
        for ( ;; ) {
#ifdef MI355X_BZ2_HUFF_PROFILE
            const uint64_t profT00 = HUFF_PROF_NOW();
#endif
            if ( groupLeft == 0 ) {
                if ( selIdx >= nSel ) { status = ST_SELECTOR_OVERRUN; break; }
                if ( ( selIdx & 7u ) == 0 ) {
                    const uint64_t fetched = selNext;
                    selCur = ( (uint64_t)sfl( (uint32_t)( fetched >> 32 ) ) << 32 ) | sfl( (uint32_t)fetched );
                    selNext = sel64[( selIdx >> 3 ) + 1];
                }
                tcur = (uint32_t)( selCur >> ( 8 * ( selIdx & 7u ) ) ) & 0xFFu;
                ++selIdx;
                groupLeft = 50;
                /* windows of this group start below pos + 50 * 20 bits: far enough from the end of the input and of
                 * the symbol buffer, the whole group may take the fast path */
                fastLeft = ( pos <= fastEnd && cnt + 50 <= SYM_CAP ) ? 50u : 0u;
                limitV = sh.limit[tcur][lane < 24 ? lane : 23];   /* first use is far away: nobody waits for it here */
            }
#ifdef MI355X_BZ2_HUFF_PROFILE
            const uint64_t profT0 = HUFF_PROF_NOW();
            profRefresh += profT0 - profT00;
#endif
            /* my 32 stream bits, from the words fetched during the previous window; then fetch for the next one */
            const uint32_t newWord = ( pos + lane ) >> 5;
            const uint32_t dsel = newWord - myWord;   /* 0..2 */
            const uint32_t hi = dsel == 0 ? D.x : ( dsel == 1 ? D.y : D.z );
            const uint32_t lo = dsel == 0 ? D.y : ( dsel == 1 ? D.z : D.w );
            myWord = newWord;
            D = load4( myWord );
            const uint32_t shv = ( pos + lane ) & 31u;
            const uint32_t bits32 = (uint32_t)( ( ( ( (uint64_t)hi << 32 ) | lo ) << shv ) >> 32 );
            const uint32_t E = sh.lut[tcur][bits32 >> ( 32 - LUT_BITS )];
            /* chain entries: lanes >= CHAIN_END never start a step (their entry reads as "stop"), which makes a step
             * at such a position a no-op and lets the first steps run without any branch */
            const uint32_t Mv = lane < CHAIN_END ? E >> 16 : 0u;

            /* Follow the code chain on the scalar unit: e = Mv[cur]; adv = e >> 10; mask |= (e & 0x3ff) << cur;
             * cur += adv.  An entry with adv == 0 ("stop": long code, end-of-block symbol first, no code, or lane >=
             * CHAIN_END) leaves cur and mask unchanged, so CHAIN_UNROLL steps are issued back to back without a branch
             * (a not-taken branch costs 13 cycles, a taken one 21, a SALU op 4.5 on this machine); the loop behind them
             * finishes windows that need more steps.  s[96:99] are scratch: s96 receives the lane value, s97 is
             * don't-care (masked by the 64-bit and). */
            uint32_t cur = 0;
            uint64_t mask = 0;
#ifdef MI355X_BZ2_HUFF_PROFILE
            asm volatile( "" :: "v"( Mv ) );
            const uint64_t profT1 = HUFF_PROF_NOW();
#endif
            {
                uint32_t adv;
                asm volatile(
                    HUFF_CHAIN_STEP HUFF_CHAIN_STEP HUFF_CHAIN_STEP HUFF_CHAIN_STEP
                    HUFF_CHAIN_STEP HUFF_CHAIN_STEP HUFF_CHAIN_STEP
                    "1:\n\t"
                    "s_cmp_lt_u32 %[cur], 54\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "v_readlane_b32 s96, %[M], %[cur]\n\t"
                    "s_lshr_b32 %[adv], s96, 10\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "s_and_b64 s[98:99], s[96:97], 0x3ff\n\t"
                    "s_lshl_b64 s[98:99], s[98:99], %[cur]\n\t"
                    "s_or_b64 %[mask], %[mask], s[98:99]\n\t"
                    "s_add_u32 %[cur], %[cur], %[adv]\n\t"
                    "s_branch 1b\n\t"
                    "2:\n\t"
                    : [cur] "+s"( cur ), [mask] "+s"( mask ), [adv] "=&s"( adv )
                    : [M] "v"( Mv )
                    : "scc", "s96", "s97", "s98", "s99" );
            }
#ifdef MI355X_BZ2_HUFF_PROFILE
            const uint64_t profT2 = HUFF_PROF_NOW();
#endif
            uint32_t mySym = ( E & 0xFFFFu ) >> 5;

            /* Fast path (almost every window): the chain ran to the end of the window, or the 50-symbol group ends
             * inside it -- then the window is cut in front of the first symbol of the next group, which was decoded
             * with the wrong table.  End-of-block cannot be among the symbols (its entries read as "stop"); the end of
             * the input and the symbol capacity were checked for the whole group when it was opened (fastLeft != 0). */
            {
                const uint32_t nAll = (uint32_t)__popcll( mask );
                const bool cut = nAll > fastLeft;
                if ( __builtin_expect( ( fastLeft != 0 ) & ( cut | ( cur >= CHAIN_END ) ), 1 ) ) {
                    const bool isStart = __builtin_amdgcn_inverse_ballot_w64( mask );
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi( (uint32_t)( mask >> 32 ),
                                                                     __builtin_amdgcn_mbcnt_lo( (uint32_t)mask, 0 ) );
                    uint32_t take = nAll, advance = cur;
                    if ( cut ) {
                        take = fastLeft;
                        advance = (uint32_t)__builtin_ctzll( __ballot( isStart && rank == take ) );
                    }
                    if ( isStart && rank < take ) sh.ring[( cnt + rank ) & ( HUFF_RING - 1 )] = (uint16_t)mySym;
                    if ( ( ( cnt + take ) ^ cnt ) & ~( HUFF_RING / 2 - 1 ) ) {
                        huff_flush( sh, symOut, cnt & ~( HUFF_RING / 2 - 1 ), HUFF_RING / 2, lane );
                    }
                    cnt += take;
                    nsym += take;
                    groupLeft -= take;
                    fastLeft -= take;
                    pos += advance;
#ifdef MI355X_BZ2_HUFF_PROFILE
                    {
                        const uint64_t profT3 = HUFF_PROF_NOW();
                        profSetup += profT1 - profT0;
                        profChain += profT2 - profT1;
                        profCommit += profT3 - profT2;
                        ++profWindows;
                    }
#endif
                    continue;
                }
            }
#ifdef MI355X_BZ2_HUFF_PROFILE
            ++profGeneral;
#endif

            /* General path.  A stop at cur < CHAIN_END is resolved here: an end-of-block symbol (its short code is in
             * the single half of the entry) ends the chain; a code longer than LUT_BITS is found by comparing its 20-bit
             * window against the per-length code range ends held one per lane (canonical codes: the first length whose
             * range end exceeds the window is the code length; = decodeLong, HuffmanCodingShortBitsCached.hpp:117-150),
             * after which the chain continues. */
            uint32_t lenOv = 0;     /* per lane: length of a long code that starts here and is on the chain */
            bool anyLong = false, invalid = false;
            while ( cur < CHAIN_END ) {
                const uint32_t shortLen = (uint32_t)__builtin_amdgcn_readlane( (int)( E & 31u ), cur );
                if ( shortLen != 0 ) {   /* end-of-block symbol starts here */
                    mask |= 1ull << cur;
                    cur += shortLen;
                    break;
                }
                if ( cur > 44 ) break;   /* a long code here could end past bit 64: leave it to the next window, so that
                                            a window never advances by more than 64 bits (the prefetched words cover 95) */
                const uint32_t v20 = (uint32_t)__builtin_amdgcn_readlane( bits32, cur ) >> 12;   /* readlane returns int */
                const uint64_t fits = __ballot( v20 < limitV );
                if ( fits == 0 ) { invalid = true; break; }
                const uint32_t l = (uint32_t)__builtin_ctzll( fits );
                lenOv = lane == cur ? l : lenOv;
                anyLong = true;
                mask |= 1ull << cur;
                cur += l;
                uint32_t adv;
                asm volatile(
                    "1:\n\t"
                    "s_cmp_lt_u32 %[cur], 54\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "v_readlane_b32 s96, %[M], %[cur]\n\t"
                    "s_lshr_b32 %[adv], s96, 10\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "s_and_b64 s[98:99], s[96:97], 0x3ff\n\t"
                    "s_lshl_b64 s[98:99], s[98:99], %[cur]\n\t"
                    "s_or_b64 %[mask], %[mask], s[98:99]\n\t"
                    "s_add_u32 %[cur], %[cur], %[adv]\n\t"
                    "s_branch 1b\n\t"
                    "2:\n\t"
                    : [cur] "+s"( cur ), [mask] "+s"( mask ), [adv] "=&s"( adv )
                    : [M] "v"( Mv )
                    : "scc", "s96", "s97", "s98", "s99" );
            }

            uint32_t nSyms = (uint32_t)__popcll( mask );
            uint32_t consumed = cur;
            uint32_t myLen = E & 31u;
            if ( anyLong ) {
                if ( lenOv != 0 ) {
                    const uint32_t code = bits32 >> ( 32 - lenOv );
                    mySym = sh.perm[tcur][sh.offs[tcur][lenOv] + code - sh.first[tcur][lenOv]];
                    myLen = lenOv;
                }
            }
            /* group boundary inside the window: keep the first groupLeft symbols, the rest use the next table */
            if ( nSyms > groupLeft ) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi( (uint32_t)( mask >> 32 ),
                                                                 __builtin_amdgcn_mbcnt_lo( (uint32_t)mask, 0 ) );
                const uint64_t cutAt = __ballot( __builtin_amdgcn_inverse_ballot_w64( mask ) && rank == groupLeft );
                const uint32_t pcut = (uint32_t)__builtin_ctzll( cutAt );
                mask &= ( 1ull << pcut ) - 1ull;
                consumed = pcut;
                nSyms = groupLeft;
                invalid = false;
            }
            /* end-of-block symbol */
            {
                const uint64_t eobMask = __ballot( mySym == eob ) & mask;
                if ( eobMask != 0 ) {
                    const uint32_t pe = (uint32_t)__builtin_ctzll( eobMask );
                    mask &= ( 1ull << pe ) - 1ull;     /* EOB itself is not stored */
                    nSyms = (uint32_t)__popcll( mask );
                    consumed = pe + (uint32_t)__builtin_amdgcn_readlane( myLen, pe );
                    finished = true;
                    invalid = false;
                }
            }
            /* a code must end inside the input (the bit reader throws otherwise); only possible near the end */
            if ( pos > safeEnd && pos + consumed > sizeBits ) {
                const uint64_t viol = __ballot( pos + lane + myLen > sizeBits ) & mask;
                if ( viol != 0 || finished ) {
                    if ( viol != 0 ) {
                        mask &= ( 1ull << (uint32_t)__builtin_ctzll( viol ) ) - 1ull;
                        nSyms = (uint32_t)__popcll( mask );
                    }
                    status = ST_EOF;
                    finished = false;
                    invalid = false;
                }
            }
            /* store the symbols of this window */
            if ( cnt + nSyms > SYM_CAP ) { status = ST_DATA_OVERFLOW; break; }
            if ( __builtin_amdgcn_inverse_ballot_w64( mask ) ) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi( (uint32_t)( mask >> 32 ),
                                                                 __builtin_amdgcn_mbcnt_lo( (uint32_t)mask, 0 ) );
                sh.ring[( cnt + rank ) & ( HUFF_RING - 1 )] = (uint16_t)mySym;
            }
            if ( ( ( cnt + nSyms ) ^ cnt ) & ~( HUFF_RING / 2 - 1 ) ) {
                huff_flush( sh, symOut, cnt & ~( HUFF_RING / 2 - 1 ), HUFF_RING / 2, lane );
            }
            cnt += nSyms;
            nsym += nSyms;
            groupLeft -= nSyms;
            fastLeft = fastLeft != 0 ? groupLeft : 0u;
            if ( status != ST_OK ) break;
            pos += consumed;
#ifdef MI355X_BZ2_HUFF_PROFILE
            profGeneralCycles += HUFF_PROF_NOW() - profT0;
#endif
            if ( finished ) { ++nsym; break; }
            if ( invalid ) {
                /* no code of any length matches at `pos`: the reference runs out of bits first if fewer than the
                 * longest code remain (oracle: huff_decode) */
                const uint32_t tMaxLen = sfl( sh.minmax[tcur] ) >> 8;
                status = ( pos + tMaxLen > sizeBits ) ? ST_EOF : ST_INVALID_CODE;
                break;
            }
        }
#ifdef MI355X_BZ2_HUFF_PROFILE
        if ( lane == 0 && ( b & 15u ) == 0 ) {
            printf( "[k_huff profile] block %u: %llu fast + %llu general windows (%.0f cycles each), cycles/window: refresh %.1f setup %.1f chain %.1f commit %.1f; bits %llu\n",
                    b, (unsigned long long)profWindows, (unsigned long long)profGeneral,
                    (double)profGeneralCycles / ( profGeneral ? profGeneral : 1 ), (double)profRefresh / profWindows, (double)profSetup / profWindows,
                    (double)profChain / profWindows, (double)profCommit / profWindows, (unsigned long long)( posBase + pos - start ) );
        }
#endif
        encSize = posBase + pos - start;
        /* what is left in the ring (whole groups of eight: the symbol buffer is padded) */
        huff_flush( sh, symOut, cnt & ~( HUFF_RING / 2 - 1 ), cnt & ( HUFF_RING / 2 - 1 ), lane );
    }

finish:
#undef FAIL
    if ( lane == 0 ) {
        BlockMeta mt;
        mt.enc_off = start;
        mt.enc_size = encSize;
        mt.decoded_size = 0;
        mt.out_off = 0;
        mt.header_crc = headerCrc;
        mt.computed_crc = 0xFFFFFFFFu;
        mt.n = 0;
        mt.orig_ptr = origPtr;
        mt.nsym = nsym;
        mt.is_eos = isEos;
        mt.is_eof = isEof;
        mt.status = status;
        mt.seg_stride = MIN_SEG_STRIDE;
        mt.nseg = 0;
        mt.walk_ok = 0;
        mt.cycle_len = 0;
        mt.nchain = 0;
        mt.pad = 0;
        meta[b] = mt;
        HuffMeta hm;
        hm.n_stored = cnt;
        hm.symbol_count = symbolCount;
        hm.status = status;
        hm.active = active;
        hmeta[b] = hm;
    }
}

__global__ __launch_bounds__( 64 * HUFF_WAVES ) void
k_huff( const uint32_t* __restrict__ in_words,
        uint64_t                     in_size_bytes,
        const uint64_t* __restrict__ offsets,
        BlockMeta* __restrict__      meta,
        HuffMeta* __restrict__       hmeta,
        uint8_t*                     sel_buf,
        uint16_t* __restrict__       sym_buf,
        uint8_t* __restrict__        stb_buf,
        uint32_t                     n_blocks,
        const uint32_t* __restrict__ order )
{
    /* HUFF_WAVES independent blocks per workgroup, one per wavefront: a 256-thread workgroup is guaranteed to put its
     * four waves on the four SIMDs of the CU, which single-wave workgroups are not (they were observed to pile up). */
    __shared__ HuffShared shAll[HUFF_WAVES];
    const uint32_t waveInGroup = sfl( threadIdx.x >> 6 );   /* wave-uniform: keeps all decoder state in SGPRs */
    /* longest-processing-time-first: slot i works on the block with the i-th largest compressed size, so the expensive
     * (incompressible) blocks start first and the cheap ones fill the tail.  The grid may be smaller than the number
     * of blocks (the host limits how many wavefronts share a CU's LDS with the other kernels of the pipeline): a
     * wavefront then takes every gridDim-th slot. */
    for ( uint32_t slot = blockIdx.x * HUFF_WAVES + waveInGroup; slot < n_blocks; slot += gridDim.x * HUFF_WAVES ) {
        huff_block( shAll[waveInGroup], in_words, in_size_bytes, offsets, meta, hmeta, sel_buf, sym_buf, stb_buf, order, slot );
        wave_sync();
    }
}

/* ============================================================================================================= */

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

/** Per-lane sequential reader of the u16 symbol stream: 8 symbols (16 bytes) per load, the next 16 bytes already in
 * flight, so the symbol loop does not wait for a global load per symbol. */
struct SymStream
{
    const uint16_t* p;
    uint32_t base;       /* index of cur's first symbol (multiple of 8) */
    uint4 cur, next;

    __device__ __forceinline__ void
    init( const uint16_t* symbols, uint32_t begin )
    {
        p = symbols;
        base = begin & ~7u;
        cur = *reinterpret_cast<const uint4*>( p + base );          /* the symbol buffer is padded past n */
        next = *reinterpret_cast<const uint4*>( p + base + 8 );
    }

    __device__ __forceinline__ uint32_t
    get( uint32_t i )   /* i must advance by one per call */
    {
        if ( i - base >= 8 ) {
            base += 8;
            cur = next;
            next = *reinterpret_cast<const uint4*>( p + base + 8 );
        }
        const uint32_t idx = i - base;
        const uint64_t half = idx < 4 ? ( (uint64_t)cur.y << 32 | cur.x ) : ( (uint64_t)cur.w << 32 | cur.z );
        return (uint32_t)( half >> ( 16 * ( idx & 3u ) ) ) & 0xFFFFu;
    }
};

/** Calls step( symbol ) for symbols [begin, end) of a lane's chunk, eight symbols per 16-byte load with the extraction
 * unrolled (static shifts instead of SymStream's dynamic ones); the next load is in flight while a group is processed.
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

/** Per-lane write combiner for a sequential byte stream: whole aligned dwords go out as one store (byte-granular
 * scattered stores cost a full write request each: 43 GB of fabric writes for 2.3 GB of L column, PMC WRITE_SIZE).  Only the
 * unaligned head and the tail of a lane's range, which share a dword with the neighbouring lane, are written bytewise.
 * Runs are what this is tuned for: half the symbols of a text block are run digits, most runs are shorter than a dword, and
 * with a loop of single bytes per run the expansion was a third of k_mtf (3.7 of 11 ms per instance for the bench's batch,
 * profiles/r03_mtf_probe.txt): fill() completes the current dword and starts the last one with two masked ORs, its only loop is over
 * the whole dwords of a long run.
 * (Round 3 also tried whole 16-byte units per store -- three more registers for the dwords in front of the current one, a
 * select per completed dword: the L column's write traffic falls, but k_mtf took 14.5 instead of 8.9 ms (<144>) and 14.5
 * instead of 10.4 ms (<272>) for the bench's batch, the step 73 instead of 66 ms.  With the stores folded into a window of
 * 256 bytes per lane the kernel is as slow as with the real addresses: it is not the write traffic that bounds it.) */
struct ByteSink
{
    uint8_t* base;
    uint32_t lo;    /* first byte position of this lane's range */
    uint32_t o;     /* next byte position */
    uint32_t acc;   /* bytes of the dword that contains o, at their place */

    __device__ __forceinline__ void
    word_done()
    {
        /* o is a multiple of 4: `acc` is the dword that ends there */
        if ( o - 4 >= lo ) {
            *reinterpret_cast<uint32_t*>( base + o - 4 ) = acc;
        } else {
            /* the dword in which the range starts belongs to the previous lane as well: bytes */
            for ( uint32_t k = lo; k < o; ++k ) base[k] = (uint8_t)( acc >> ( 8 * ( k & 3u ) ) );
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
        /* (whatever is left starts a dword) */
        for ( uint32_t k = count >> 2; k != 0; --k ) {
            *reinterpret_cast<uint32_t*>( base + o ) = word;
            o += 4;
        }
        const uint32_t rest = count & 3u;
        acc |= word & ( ( 1u << ( 8 * rest ) ) - 1u );
        o += rest;
    }

    __device__ __forceinline__ void
    flush()
    {
        /* the last, incomplete dword (shared with the next lane) */
        const uint32_t first = ( o & ~3u ) > lo ? ( o & ~3u ) : lo;
        for ( uint32_t k = first; k < o; ++k ) base[k] = (uint8_t)( acc >> ( 8 * ( k & 3u ) ) );
        acc = 0;
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
           const uint16_t* __restrict__  sym_buf,
           const uint8_t* __restrict__   stb_buf,
           uint8_t* __restrict__         l_buf,
           uint32_t                      n_blocks,
           const uint32_t* __restrict__  order,
           uint32_t                      seg_target,
           uint8_t* __restrict__         listBytes,
           uint4* __restrict__           permRows,
           uint8_t* __restrict__         cur,
           uint32_t* __restrict__        starts,
           unsigned long long* __restrict__ waveTotals,
           uint32_t* __restrict__        firstErrorAt )
{
    constexpr uint32_t LIST_ENTRIES = LANE_STRIDE - 16;
    uint32_t& firstError = *firstErrorAt;

    const uint32_t slot = blockIdx.x;
    if ( slot >= n_blocks ) return;
    const uint32_t b = order[slot];
    const HuffMeta hm = hmeta[b];
    if ( !hm.active ) return;
    if ( ( hm.symbol_count <= MTF_SMALL_STRIDE - 16 ) != ( LANE_STRIDE == MTF_SMALL_STRIDE ) ) return;   /* other instance */
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t n = hm.n_stored;
    const uint16_t* const sym = sym_buf + (size_t)b * SYM_STRIDE;
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
        for_symbols( sym, begin, end, [&] ( uint32_t s ) {
            if ( s <= 1 ) {
                if ( runPos == 0 ) { runPos = 1; hh = 0; }
                hh += runPos << s;
                runPos <<= 1;
            } else {
                if ( runPos != 0 ) { count += hh; runPos = 0; }
                mtf_lane_move( mine, permRows, s - 1 );
                ++count;
            }
            return true;
        } );
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

    /* ---- pass B: replay with the true start list, write the L column.  (Round 3 measured the alternative -- pass A replaces
     * every symbol by the position its entry has in the chunk's start list, pass B only looks positions up: the list updates
     * of a pass cost 0.1 to 1.4 ms of its 3.3 ms, the rewritten symbols 4.6 GB of writes; no gain, profiles/r03_mtf_probe.txt) ---- */
    {
        /* positions are 32-bit: a start beyond the buffer (only possible for damaged data, whose runs can add up to
         * anything) is clamped -- that lane then reports the overflow at its first symbol, an earlier lane wins anyway */
        const uint32_t startAt = prefix < MAX_N ? (uint32_t)prefix : MAX_N;
        ByteSink sink{ L, startAt, startAt, 0 };
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
                sink.fill( reinterpret_cast<const uint8_t*>( mine )[0], hh );
            }
            if ( sink.o >= MAX_N ) { err = ST_DATA_OVERFLOW; return false; }
            sink.put( mtf_lane_move( mine, permRows, s - 1 ) );
            return true;
        } );
        /* a run that is still open where the Huffman stage FAILED is never flushed by the reference */
        if ( err == 0 && runPos != 0 && ( end < n || hm.status == ST_OK ) ) {
            if ( sink.o + hh > MAX_N ) {
                err = ST_RUN_OVERFLOW;
            } else {
                sink.fill( reinterpret_cast<const uint8_t*>( mine )[0], hh );
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
        /* walk segments: about seg_target of them (<= KMAX), a table entry in every `stride` a segment start */
        uint32_t stride = ( N + seg_target - 1 ) / seg_target;
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
       const uint16_t* __restrict__  sym_buf,
       const uint8_t* __restrict__   stb_buf,
       uint8_t* __restrict__         l_buf,
       uint32_t                      n_blocks,
       const uint32_t* __restrict__  order,
       uint32_t                      seg_target )   /* walk segments per block to aim for, 1 .. KMAX */
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t ldsAtLaunch[];     /* sizeof( MtfShared<LANE_STRIDE, THREADS> ) */
    auto& shared = *reinterpret_cast<MtfShared<LANE_STRIDE, THREADS>*>( ldsAtLaunch );
    mtf_block<LANE_STRIDE, THREADS>( meta, hmeta, sym_buf, stb_buf, l_buf, n_blocks, order, seg_target, shared.listBytes, shared.permRows,
                                     shared.cur, shared.starts, shared.waveTotals, &shared.firstError );
}
}  // namespace bz2gpu
