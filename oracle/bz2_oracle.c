/*
 * TEST INFRASTRUCTURE ONLY -- see bz2_oracle.h.  Plain-C restatement of the reference's bzip2 block decoder.
 * It follows the reference's ALGORITHM (same tables, same T-vector inverse BWT, same order of checks) so that
 * results AND failure modes agree; it is deliberately scalar and simple.  Not shipped, not on the product path.
 */
#include "bz2_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MAX_GROUPS 6          /* bzip2.hpp:95 */
#define GROUP_SIZE 50         /* bzip2.hpp:96 */
#define MAX_HUFCODE_BITS 20   /* bzip2.hpp:97 */
#define MAX_SYMBOLS 258       /* bzip2.hpp:98 */
#define DBUF_SIZE 900000u     /* bzip2.hpp:416 */
#define MAGIC_BLOCK 0x314159265359ULL /* bzip2.hpp:103 */
#define MAGIC_EOS 0x177245385090ULL   /* bzip2.hpp:104 */

/* ---------------------------------------------------------------- CRC, bzip2.hpp:59-91 */
static uint32_t g_crc_table[256];
static int g_crc_ready = 0;

static void
crc_init( void )
{
    if ( g_crc_ready ) return;
    for ( uint32_t i = 0; i < 256; ++i ) {
        uint32_t c = i << 24;
        for ( int j = 0; j < 8; ++j ) {
            c = ( c & 0x80000000u ) ? ( c << 1 ) ^ 0x04C11DB7u : ( c << 1 );
        }
        g_crc_table[i] = c;
    }
    g_crc_ready = 1;
}

uint32_t
orc_crc32_update( uint32_t crc, const uint8_t* data, uint64_t n )
{
    crc_init();
    for ( uint64_t i = 0; i < n; ++i ) {
        crc = ( crc << 8 ) ^ g_crc_table[( ( crc >> 24 ) ^ data[i] ) & 0xFFu];
    }
    return crc;
}

uint32_t
orc_stream_crc_combine( uint32_t s, uint32_t b )
{
    return ( ( s << 1 ) | ( s >> 31 ) ) ^ b;   /* BZ2Reader.hpp:481-484 */
}

uint32_t
orc_run_length( const uint8_t* digits, uint32_t n )
{
    /* bzip2.hpp:726-743: hh += runPos << nextSym; runPos <<= 1 (uint32 arithmetic, wraps like the reference) */
    uint32_t hh = 0, runPos = 0;
    for ( uint32_t i = 0; i < n; ++i ) {
        if ( runPos == 0 ) { runPos = 1; hh = 0; }
        hh += runPos << digits[i];
        runPos <<= 1;
    }
    return hh;
}

/* ---------------------------------------------------------------- MSB-first bit reader, BitReader.hpp (a8) */
typedef struct {
    const uint8_t* p;
    uint64_t size_bits;
    uint64_t pos;
    int eof;  /* sticky: a read went past the end (EndOfFileReached) */
} bitrd;

static uint32_t
br_peek( const bitrd* b, unsigned n )  /* n <= 32; zero padded beyond EOF */
{
    uint64_t v = 0;
    const uint64_t byte = b->pos >> 3;
    const uint64_t nbytes = b->size_bits >> 3;
    for ( int i = 0; i < 8; ++i ) {
        v = ( v << 8 ) | ( byte + i < nbytes ? b->p[byte + i] : 0 );
    }
    v <<= ( b->pos & 7 );
    return n == 0 ? 0 : (uint32_t)( v >> ( 64 - n ) );
}

static uint32_t
br_read( bitrd* b, unsigned n )
{
    if ( b->pos + n > b->size_bits ) {
        b->eof = 1;
        b->pos = b->size_bits;
        return 0;
    }
    const uint32_t v = br_peek( b, n );
    b->pos += n;
    return v;
}

/* ---------------------------------------------------------------- canonical Huffman (a5, a7) */
typedef struct {
    uint8_t  min_len, max_len;
    uint32_t min_code[MAX_HUFCODE_BITS + 2];   /* m_minimumCodeValuesPerLevel, indexed by len - min_len */
    uint16_t offsets[MAX_HUFCODE_BITS + 2];    /* m_offsets */
    uint16_t symbols[MAX_SYMBOLS];             /* m_symbolsPerLength */
} huff;

/* HuffmanCodingSymbolsPerLength::initializeFromLengths, HuffmanCodingSymbolsPerLength.hpp:71-95, with
 * HuffmanCodingBase::{initializeMinMaxCodeLengths :46-69, checkCodeLengthFrequencies :71-112 (CHECK_OPTIMALITY=false),
 * initializeMinimumCodeValues :117-149}.  All lengths are in [1,20] here (readTrees guarantees it). */
static int
huff_init( huff* h, const uint8_t* lengths, unsigned n )
{
    uint32_t freq[MAX_HUFCODE_BITS + 1];
    memset( freq, 0, sizeof( freq ) );
    memset( h, 0, sizeof( *h ) );
    unsigned mn = 255, mx = 0;
    for ( unsigned i = 0; i < n; ++i ) {
        if ( lengths[i] > mx ) mx = lengths[i];
        if ( lengths[i] > 0 && lengths[i] < mn ) mn = lengths[i];
        ++freq[lengths[i]];
    }
    h->min_len = (uint8_t)mn;
    h->max_len = (uint8_t)mx;
    uint32_t unused = 1u << mn;
    for ( unsigned l = mn; l <= mx; ++l ) {
        if ( freq[l] > unused ) return ORC_ERR_HUFFMAN_LENGTHS;
        unused -= freq[l];
        unused *= 2;
    }
    freq[0] = 0;
    uint32_t minCode = 0;
    for ( unsigned bits = mn > 1 ? mn : 1; bits <= mx; ++bits ) {
        minCode = ( minCode + freq[bits - 1] ) << 1;
        h->min_code[bits - mn] = minCode;
    }
    unsigned sum = 0;
    for ( unsigned l = mn; l <= mx; ++l ) {
        h->offsets[l - mn] = (uint16_t)sum;
        sum += freq[l];
    }
    h->offsets[mx - mn + 1] = (uint16_t)sum;
    uint16_t sizes[MAX_HUFCODE_BITS + 2];
    memcpy( sizes, h->offsets, sizeof( sizes ) );
    for ( unsigned s = 0; s < n; ++s ) {
        if ( lengths[s] != 0 ) h->symbols[sizes[lengths[s] - mn]++] = (uint16_t)s;
    }
    return ORC_OK;
}

/* HuffmanCodingSymbolsPerLength::decode :97-124 expressed on a zero-padded 20-bit peek: grow the prefix one bit at
 * a time from min_len to max_len and test `minCode <= code && code - minCode < count`.  The LUT of
 * HuffmanCodingShortBitsCached (:98-150) is a cache of exactly this function, so it is not restated. */
static int
huff_decode( const huff* h, bitrd* b, uint16_t* sym )
{
    const uint32_t window = br_peek( b, MAX_HUFCODE_BITS );
    for ( unsigned l = h->min_len; l <= h->max_len; ++l ) {
        const uint32_t code = window >> ( MAX_HUFCODE_BITS - l );
        const unsigned k = l - h->min_len;
        if ( h->min_code[k] <= code ) {
            const uint32_t sub = h->offsets[k] + ( code - h->min_code[k] );
            if ( sub < h->offsets[k + 1] ) {
                if ( b->pos + l > b->size_bits ) { b->eof = 1; return ORC_ERR_EOF; }
                b->pos += l;
                *sym = h->symbols[sub];
                return ORC_OK;
            }
        }
    }
    /* no code matches: the reference runs out of bits first if fewer than max_len remain */
    if ( b->pos + h->max_len > b->size_bits ) { b->eof = 1; return ORC_ERR_EOF; }
    return ORC_ERR_INVALID_CODE;
}

/* ---------------------------------------------------------------- block state */
typedef struct {
    bitrd br;
    uint64_t encoded_offset, encoded_size;
    uint32_t header_crc, orig_ptr;
    int eos, eof;
    uint8_t symbol_to_byte[256];
    unsigned symbol_count;
    unsigned group_count;
    unsigned selectors_count;
    uint8_t selectors[32768];
    huff codings[MAX_GROUPS];
} blk;

/* Block::readSymbolMaps, bzip2.hpp:526-571 */
static void
read_symbol_maps( blk* s )
{
    const uint32_t used = br_read( &s->br, 16 );
    s->symbol_count = 0;
    for ( int i = 0; i < 16; ++i ) {
        if ( used & ( 1u << ( 15 - i ) ) ) {
            const uint32_t bitmap = br_read( &s->br, 16 );
            for ( int j = 0; j < 16; ++j ) {
                if ( bitmap & ( 1u << ( 15 - j ) ) ) {
                    s->symbol_to_byte[s->symbol_count++] = (uint8_t)( 16 * i + j );
                }
            }
        }
    }
}

/* Block::readSelectors, bzip2.hpp:574-637 */
static int
read_selectors( blk* s )
{
    s->group_count = br_read( &s->br, 3 );
    if ( s->br.eof ) return ORC_ERR_EOF;
    if ( s->group_count < 2 || s->group_count > MAX_GROUPS ) return ORC_ERR_GROUP_COUNT;
    s->selectors_count = br_read( &s->br, 15 );
    if ( s->br.eof ) return ORC_ERR_EOF;
    if ( s->selectors_count == 0 ) return ORC_ERR_SELECTOR_COUNT;

    uint8_t mtf[MAX_GROUPS];
    for ( unsigned i = 0; i < s->group_count; ++i ) mtf[i] = (uint8_t)i;
    for ( unsigned i = 0; i < s->selectors_count; ++i ) {
        /* BITS_TO_SELECTOR: number of leading 1 bits in a 6-bit peek, 6 if all ones (:602-619).
         * The reference peeks 6 bits; near EOF the peek itself throws only if fewer than the needed bits remain
         * AND the buffer cannot be refilled -- BitReader::peek zero-extends?  No: peek2 throws EndOfFileReached
         * when it cannot provide the bits (BitReader.hpp:391-466).  So a peek<6> within 6 bits of EOF throws. */
        if ( s->br.pos + MAX_GROUPS > s->br.size_bits ) { s->br.eof = 1; return ORC_ERR_EOF; }
        const uint32_t bits = br_peek( &s->br, MAX_GROUPS );
        unsigned j = 0;
        while ( j < MAX_GROUPS && ( bits & ( 1u << ( MAX_GROUPS - 1 - j ) ) ) ) ++j;
        s->br.pos += j + 1;   /* seekAfterPeek( j + 1 ) */
        if ( j >= s->group_count ) return ORC_ERR_SELECTOR_UNARY;
        const uint8_t uc = mtf[j];
        memmove( mtf + 1, mtf, j );
        mtf[0] = uc;
        s->selectors[i] = uc;
    }
    return ORC_OK;
}

/* Block::readTrees, bzip2.hpp:644-685 */
static int
read_trees( blk* s )
{
    const unsigned symCount = s->symbol_count + 2;
    for ( unsigned j = 0; j < s->group_count; ++j ) {
        uint8_t lengths[MAX_SYMBOLS];
        memset( lengths, 0, sizeof( lengths ) );
        uint32_t hh = br_read( &s->br, 5 );
        if ( s->br.eof ) return ORC_ERR_EOF;
        for ( unsigned sym = 0; sym < symCount; ++sym ) {
            while ( 1 ) {
                if ( (uint32_t)( MAX_HUFCODE_BITS - 1 ) < hh - 1 ) return ORC_ERR_CODE_LENGTH;
                const uint32_t more = br_read( &s->br, 1 );
                if ( s->br.eof ) return ORC_ERR_EOF;
                if ( more == 0 ) break;
                const uint32_t dec = br_read( &s->br, 1 );
                if ( s->br.eof ) return ORC_ERR_EOF;
                hh += 1 - ( dec << 1 );
            }
            lengths[sym] = (uint8_t)hh;
        }
        const int err = huff_init( &s->codings[j], lengths, symCount );
        if ( err != ORC_OK ) return err;
    }
    return ORC_OK;
}

/* Block::readBlockHeader, bzip2.hpp:479-523 (+ readBlockTrees :348-361) */
static int
read_block_header( blk* s, const uint8_t* file, uint64_t file_size, uint64_t bit_offset )
{
    s->br.p = file;
    s->br.size_bits = file_size * 8;
    s->br.pos = bit_offset;
    s->br.eof = 0;
    s->encoded_offset = bit_offset;
    s->encoded_size = 0;
    s->eos = s->eof = 0;
    s->symbol_count = 0;

    if ( bit_offset > s->br.size_bits ) return ORC_ERR_EOF;
    const uint64_t hi = br_read( &s->br, 24 );
    const uint64_t lo = br_read( &s->br, 24 );
    if ( s->br.eof ) return ORC_ERR_EOF;
    const uint64_t magic = ( hi << 24 ) | lo;
    s->header_crc = br_read( &s->br, 32 );
    if ( s->br.eof ) return ORC_ERR_EOF;
    if ( magic == MAGIC_EOS ) {
        s->eos = 1;
        const unsigned inByte = (unsigned)( s->br.pos & 7 );
        if ( inByte > 0 ) {
            br_read( &s->br, 8 - inByte );
            if ( s->br.eof ) return ORC_ERR_EOF;
        }
        s->encoded_size = s->br.pos - s->encoded_offset;
        s->eof = s->br.pos >= s->br.size_bits;
        return ORC_OK;
    }
    if ( magic != MAGIC_BLOCK ) return ORC_ERR_BAD_MAGIC;
    const uint32_t randomized = br_read( &s->br, 1 );
    if ( s->br.eof ) return ORC_ERR_EOF;
    if ( randomized ) return ORC_ERR_RANDOMIZED;
    s->orig_ptr = br_read( &s->br, 24 );
    if ( s->br.eof ) return ORC_ERR_EOF;
    if ( s->orig_ptr > DBUF_SIZE ) return ORC_ERR_ORIGPTR_RANGE;

    read_symbol_maps( s );
    if ( s->br.eof ) return ORC_ERR_EOF;
    int err = read_selectors( s );
    if ( err != ORC_OK ) return err;
    return read_trees( s );
}

static void
fill_result( const blk* s, orc_block_result* r, int status )
{
    r->encoded_offset_bits = s->encoded_offset;
    r->encoded_size_bits = s->encoded_size;
    r->header_crc = s->header_crc;
    r->orig_ptr = s->orig_ptr;
    r->is_eos = s->eos;
    r->is_eof = s->eof;
    r->status = status;
}

int
orc_read_stream_header( const uint8_t* file, uint64_t file_size, uint64_t bit_offset )
{
    bitrd b = { file, file_size * 8, bit_offset, 0 };
    const char magic[3] = { 'B', 'Z', 'h' };
    for ( int i = 0; i < 3; ++i ) {
        const uint32_t c = br_read( &b, 8 );
        if ( b.eof || (char)c != magic[i] ) return 0;
    }
    const uint32_t lvl = br_read( &b, 8 );
    if ( b.eof || lvl < '1' || lvl > '9' ) return 0;
    return (int)( lvl - '0' );
}

int
orc_read_block_header( const uint8_t* file, uint64_t file_size, uint64_t bit_offset, orc_block_result* res )
{
    blk* s = (blk*)calloc( 1, sizeof( blk ) );
    memset( res, 0, sizeof( *res ) );
    const int err = read_block_header( s, file, file_size, bit_offset );
    fill_result( s, res, err );
    free( s );
    return err;
}

int
orc_decode_block( const uint8_t* file, uint64_t file_size, uint64_t bit_offset,
                  uint8_t* out, uint64_t out_capacity, orc_block_result* res,
                  uint8_t* bwt_l_out, uint8_t* rle_out )
{
    crc_init();
    memset( res, 0, sizeof( *res ) );
    res->computed_crc = 0xFFFFFFFFu;   /* BlockData::calculatedCRC default, BZ2BlockFetcher.hpp:33 */
    blk* s = (blk*)calloc( 1, sizeof( blk ) );
    uint32_t* dbuf = NULL;
    int err = read_block_header( s, file, file_size, bit_offset );
    if ( err != ORC_OK || s->eos ) goto done;

    /* ---- Block::readBlockData, bzip2.hpp:691-807 ---- */
    dbuf = (uint32_t*)calloc( DBUF_SIZE, sizeof( uint32_t ) );
    uint32_t byteCount[256];
    memset( byteCount, 0, sizeof( byteCount ) );
    uint8_t mtf[256];
    for ( int i = 0; i < 256; ++i ) mtf[i] = (uint8_t)i;

    uint32_t dbufCount = 0, nSymbols = 0;
    {
        const huff* coding = &s->codings[0];
        uint32_t hh = 0, runPos = 0, symCount = 0, selector = 0;
        for ( ;; ) {
            if ( symCount-- == 0 ) {
                symCount = GROUP_SIZE - 1;
                if ( selector >= s->selectors_count ) { err = ORC_ERR_SELECTOR_OVERRUN; goto done; }
                coding = &s->codings[s->selectors[selector]];
                selector++;
            }
            uint16_t nextSym = 0;
            err = huff_decode( coding, &s->br, &nextSym );
            if ( err != ORC_OK ) goto done;
            ++nSymbols;

            if ( nextSym <= 1 ) {   /* RUNA / RUNB */
                if ( runPos == 0 ) { runPos = 1; hh = 0; }
                hh += runPos << nextSym;
                runPos <<= 1;
                continue;
            }
            if ( runPos != 0 ) {
                runPos = 0;
                /* The reference adds in uint32 (bzip2.hpp:751): a wrapped sum would pass its check and then overrun
                 * dbuf (undefined behaviour).  The only defined outcome is the overflow error, so widen the sum. */
                if ( (uint64_t)dbufCount + hh > DBUF_SIZE ) { err = ORC_ERR_RUN_OVERFLOW; goto done; }
                const uint8_t uc = s->symbol_to_byte[mtf[0]];
                byteCount[uc] += hh;
                while ( hh-- != 0 ) dbuf[dbufCount++] = uc;
            }
            if ( nextSym > s->symbol_count ) break;   /* end of block */
            if ( dbufCount >= DBUF_SIZE ) { err = ORC_ERR_DATA_OVERFLOW; goto done; }
            const int ii = nextSym - 1;
            uint8_t uc = mtf[ii];
            memmove( mtf + 1, mtf, (size_t)ii );
            mtf[0] = uc;
            uc = s->symbol_to_byte[uc];
            byteCount[uc]++;
            dbuf[dbufCount++] = uc;
        }
    }
    res->bwt_length = dbufCount;
    res->n_symbols = nSymbols;
    if ( s->orig_ptr >= dbufCount ) { err = ORC_ERR_ORIGPTR_DATA; goto done; }
    if ( bwt_l_out ) {
        for ( uint32_t i = 0; i < dbufCount; ++i ) bwt_l_out[i] = (uint8_t)dbuf[i];
    }

    /* ---- BurrowsWheelerTransformData::prepare, bzip2.hpp:810-847 ---- */
    {
        uint32_t cum = 0;
        for ( int i = 0; i < 256; ++i ) {
            const uint32_t n = cum + byteCount[i];
            byteCount[i] = cum;
            cum = n;
        }
        for ( uint32_t i = 0; i < dbufCount; ++i ) {
            const uint8_t uc = (uint8_t)dbuf[i];
            dbuf[byteCount[uc]] |= i << 8;
            byteCount[uc]++;
        }
    }
    s->encoded_size = s->br.pos - s->encoded_offset;

    /* ---- BurrowsWheelerTransformData::decodeBlock, bzip2.hpp:850-910 (whole block at once) ---- */
    {
        uint32_t crc = 0xFFFFFFFFu;
        uint32_t writePos = dbuf[s->orig_ptr];
        int writeCurrent = (int)( writePos & 0xFF );
        writePos >>= 8;
        int writeRun = -1;
        uint32_t writeCount = dbufCount;
        uint64_t n = 0, nrle = 0;
        int overflow = 0;
        while ( writeCount > 0 ) {
            writeCount--;
            const int previous = writeCurrent;
            writePos = dbuf[writePos];
            writeCurrent = (int)( writePos & 0xFF );
            writePos >>= 8;
            if ( rle_out ) rle_out[nrle] = (uint8_t)writeCurrent;
            ++nrle;
            if ( writeRun < 3 ) {
                if ( out ) { if ( n < out_capacity ) out[n] = (uint8_t)writeCurrent; else overflow = 1; }
                ++n;
                crc = ( crc << 8 ) ^ g_crc_table[( ( crc >> 24 ) ^ (uint32_t)writeCurrent ) & 0xFFu];
                if ( writeCurrent != previous ) writeRun = 0; else ++writeRun;
            } else {
                const uint8_t sym = (uint8_t)previous;
                for ( int k = 0; k < writeCurrent; ++k ) {
                    if ( out ) { if ( n < out_capacity ) out[n] = sym; else overflow = 1; }
                    ++n;
                    crc = ( crc << 8 ) ^ g_crc_table[( ( crc >> 24 ) ^ sym ) & 0xFFu];
                }
                writeCurrent = -1;
                writeRun = 0;
            }
        }
        res->decoded_size = n;
        res->computed_crc = ~crc;
        if ( overflow ) { err = ORC_ERR_OUTPUT_CAPACITY; goto done; }
        if ( res->computed_crc != s->header_crc ) { err = ORC_ERR_CRC; goto done; }
    }

done:
    fill_result( s, res, err );
    free( dbuf );
    free( s );
    return err;
}

/* ---------------------------------------------------------------- magic scan (a14) */
uint64_t
orc_find_magic( const uint8_t* file, uint64_t file_size, uint64_t magic48, uint64_t* offsets, uint64_t capacity )
{
    /* Same result set as BitStringFinder<48> (BitStringFinder.hpp:158-285): every bit offset o with
     * bits[o, o+48) == magic, ascending.  Restated as a sliding 64-bit window, one byte per step, 8 shifts each. */
    uint64_t found = 0;
    if ( file_size < 6 ) return 0;
    const uint64_t mask = 0xFFFFFFFFFFFFULL;
    uint64_t window = 0;   /* last 8 bytes */
    for ( uint64_t i = 0; i < file_size + 1; ++i ) {
        /* window holds bytes [i-8, i) ; test offsets whose 48 bits end within byte i-1 */
        if ( i >= 6 ) {
            /* candidates: bit offsets 8*(i-7)+1 .. 8*(i-6) -> they need bytes i-7..i-1 (7 bytes = 56 bits) */
            for ( int sh = 7; sh >= 0; --sh ) {
                /* pattern ends sh bits before the end of byte i-1: start bit = 8*i - sh - 48 */
                const int64_t start = (int64_t)( 8 * i ) - sh - 48;
                if ( start < 0 ) continue;
                if ( sh != 0 && i < 7 ) continue;
                if ( ( ( window >> sh ) & mask ) == magic48 ) {
                    if ( found < capacity ) offsets[found] = (uint64_t)start;
                    ++found;
                }
            }
        }
        if ( i < file_size ) window = ( window << 8 ) | file[i];
    }
    return found;
}

/* ---------------------------------------------------------------- whole-file emulation of ParallelBZ2Reader */
typedef struct {
    uint64_t* bits; uint64_t* bytes; uint64_t cap, len;
    uint64_t n_eos;
    uint64_t last_enc_size, last_dec_size;
} bmap;

/* BlockMap::push, BlockMap.hpp:69-119 (monotone case only: the sequential reader never re-inserts) */
static void
bmap_push( bmap* m, uint64_t off, uint64_t enc_size, uint64_t dec_size )
{
    uint64_t dec_off = 0;
    if ( m->len > 0 ) {
        const uint64_t idx = ( m->len <= m->cap ? m->len : m->cap ) - 1;
        dec_off = m->bytes[idx] + m->last_dec_size;
    }
    if ( m->len < m->cap ) { m->bits[m->len] = off; m->bytes[m->len] = dec_off; }
    m->len++;
    if ( dec_size == 0 ) m->n_eos++;
    m->last_dec_size = dec_size;
    m->last_enc_size = enc_size;
}

/* BlockMap::finalize, BlockMap.hpp:171-197 */
static void
bmap_finalize( bmap* m )
{
    if ( m->len == 0 ) {
        if ( m->cap > 0 ) { m->bits[0] = 0; m->bytes[0] = 0; }
        m->len = 1;
    } else if ( m->last_enc_size != 0 || m->last_dec_size != 0 ) {
        const uint64_t idx = ( m->len <= m->cap ? m->len : m->cap ) - 1;
        const uint64_t b = m->bits[idx] + m->last_enc_size, d = m->bytes[idx] + m->last_dec_size;
        if ( m->len < m->cap ) { m->bits[m->len] = b; m->bytes[m->len] = d; }
        m->len++;
    }
    m->last_enc_size = m->last_dec_size = 0;
}

int
orc_decode_file( const uint8_t* file, uint64_t file_size,
                 uint8_t* out, uint64_t out_capacity, uint64_t* decoded_size,
                 uint64_t* map_bits, uint64_t* map_bytes, uint64_t map_capacity, uint64_t* map_len,
                 int* trailing_garbage )
{
    bmap m = { map_bits, map_bytes, map_capacity, 0, 0, 0, 0 };
    uint64_t total = 0;
    int status = ORC_OK;
    if ( trailing_garbage ) *trailing_garbage = 0;

    /* BlockFinder over ParallelBitStringFinder<48>( MAGIC_BITS_BLOCK ): ParallelBZ2Reader.hpp:56-63 */
    uint64_t nfound = orc_find_magic( file, file_size, MAGIC_BLOCK, NULL, 0 );
    uint64_t* found = (uint64_t*)malloc( ( nfound + 1 ) * sizeof( uint64_t ) );
    orc_find_magic( file, file_size, MAGIC_BLOCK, found, nfound );

    /* ParallelBZ2Reader::read loop, ParallelBZ2Reader.hpp:167-269 */
    for ( uint64_t i = 0; ; ++i ) {
        if ( i >= nfound ) break;   /* blockFinder().get() == nullopt -> finalize, EOF (:193-197) */
        if ( i == 0 ) {
            /* BZ2BlockFetcher ctor reads the stream header once (BZ2BlockFetcher.hpp:56) */
            if ( orc_read_stream_header( file, file_size, 0 ) == 0 ) { status = ORC_ERR_STREAM_HEADER; goto out; }
        }
        orc_block_result r;
        status = orc_decode_block( file, file_size, found[i],
                                   out ? out + ( total < out_capacity ? total : out_capacity ) : NULL,
                                   out ? ( total < out_capacity ? out_capacity - total : 0 ) : 0, &r, NULL, NULL );
        if ( status != ORC_OK ) goto out;
        bmap_push( &m, r.encoded_offset_bits, r.encoded_size_bits, r.decoded_size );
        total += r.decoded_size;
        if ( !r.is_eof ) {
            orc_block_result h;
            status = orc_read_block_header( file, file_size, r.encoded_offset_bits + r.encoded_size_bits, &h );
            if ( status != ORC_OK ) goto out;
            if ( h.is_eos ) {
                bmap_push( &m, h.encoded_offset_bits, h.encoded_size_bits, 0 );
                const uint64_t next = h.encoded_offset_bits + h.encoded_size_bits;
                if ( next < file_size * 8 ) {
                    if ( orc_read_stream_header( file, file_size, next ) == 0 ) {
                        /* "Trailing garbage after EOF ignored!" -> finder truncated (:218-235) */
                        if ( trailing_garbage ) *trailing_garbage = 1;
                        nfound = m.len - m.n_eos;
                    }
                }
            }
        }
    }
    bmap_finalize( &m );
out:
    free( found );
    if ( decoded_size ) *decoded_size = total;
    if ( map_len ) *map_len = m.len;
    return status;
}
