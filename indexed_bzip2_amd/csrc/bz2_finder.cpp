/**
 * bz2_finder.cpp -- host-side magic-bit scan (C ABI section 2 of include/mi355x_bz2.h).
 *
 * Finds every bit offset of a 48-bit pattern (block magic 0x314159265359 / EOS magic 0x177245385090).
 * Same result set as the reference's BitStringFinder<48>::find / ParallelBitStringFinder<48>::find
 * (src/core/BitStringFinder.hpp:158-285, src/core/ParallelBitStringFinder.hpp:159-265), restated:
 * for each of the 8 bit phases the pattern contains 5 (phase != 0) or 6 (phase 0) whole bytes; those are located
 * with memmem() and the partial head/tail bytes are verified.  Work is split into byte ranges over std::threads.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mi355x_bz2.h"
#include "bz2_host.hpp"

namespace mi355x
{
/** All matches whose FIRST byte B lies in [begin, end); reads up to byte B+6. Sorted ascending. */
void
scanMagicRange( const uint8_t* bytes, uint64_t size, uint64_t magic48, uint64_t begin, uint64_t end,
           std::vector<uint64_t>& found )
{
    constexpr uint64_t SUB = 1u << 20;
    for ( uint64_t sub = begin; sub < end; sub += SUB ) {
        const uint64_t subEnd = std::min( end, sub + SUB );
        const size_t firstNew = found.size();
        for ( unsigned s = 0; s < 8; ++s ) {
            /* pattern placed at bit offset s inside a 7-byte (56-bit) window */
            const uint64_t win = magic48 << ( 8 - s );            /* bits 55..0, MSB first */
            const uint64_t winMask = 0xFFFFFFFFFFFFULL << ( 8 - s );
            uint8_t pat[7], msk[7];
            for ( int k = 0; k < 7; ++k ) {
                pat[k] = (uint8_t)( win >> ( 8 * ( 6 - k ) ) );
                msk[k] = (uint8_t)( winMask >> ( 8 * ( 6 - k ) ) );
            }
            const unsigned needleOff = s == 0 ? 0 : 1;
            const unsigned needleLen = s == 0 ? 6 : 5;
            const unsigned span = s == 0 ? 6 : 7;   /* bytes touched by a match */
            /* search needle start positions p = B + needleOff with B in [sub, subEnd) and B + span <= size */
            if ( size < span ) continue;
            const uint64_t lastB = std::min<uint64_t>( subEnd, size - span + 1 );   /* exclusive */
            uint64_t B = sub;
            while ( B < lastB ) {
                const uint8_t* hayBegin = bytes + B + needleOff;
                const uint64_t hayLen = ( lastB - 1 + needleOff + needleLen ) - ( B + needleOff );
                const void* hit = memmem( hayBegin, hayLen, pat + needleOff, needleLen );
                if ( hit == nullptr ) break;
                const uint64_t p = (uint64_t)( static_cast<const uint8_t*>( hit ) - bytes );
                const uint64_t mB = p - needleOff;
                bool ok = true;
                if ( s != 0 ) {
                    ok = ( ( bytes[mB] & msk[0] ) == pat[0] ) && ( ( bytes[mB + 6] & msk[6] ) == pat[6] );
                }
                if ( ok ) found.push_back( mB * 8 + s );
                B = mB + 1;
            }
        }
        std::sort( found.begin() + (std::ptrdiff_t)firstNew, found.end() );
    }
}
}  // namespace mi355x

using mi355x::scanMagicRange;

extern "C" uint64_t
mi355x_bz2_find_magic( const uint8_t* bytes, uint64_t size, uint64_t magic48,
                       uint64_t* bitOffsets, uint64_t capacity, uint32_t threads )
{
    if ( bytes == nullptr || size < 6 ) return 0;
    magic48 &= 0xFFFFFFFFFFFFULL;
    unsigned T = threads == 0 ? std::max( 1u, std::thread::hardware_concurrency() ) : threads;
    const uint64_t minChunk = 4u << 20;
    T = (unsigned)std::max<uint64_t>( 1, std::min<uint64_t>( T, ( size + minChunk - 1 ) / minChunk ) );
    std::vector<std::vector<uint64_t> > results( T );
    if ( T == 1 ) {
        scanMagicRange( bytes, size, magic48, 0, size, results[0] );
    } else {
        std::vector<std::thread> pool;
        const uint64_t chunk = ( size + T - 1 ) / T;
        for ( unsigned t = 0; t < T; ++t ) {
            const uint64_t b = std::min<uint64_t>( size, (uint64_t)t * chunk );
            const uint64_t e = std::min<uint64_t>( size, b + chunk );
            pool.emplace_back( [&, t, b, e] () { scanMagicRange( bytes, size, magic48, b, e, results[t] ); } );
        }
        for ( auto& th : pool ) th.join();
    }
    uint64_t n = 0;
    for ( const auto& r : results ) {
        for ( const auto o : r ) {
            if ( n < capacity && bitOffsets != nullptr ) bitOffsets[n] = o;
            ++n;
        }
    }
    return n;
}

extern "C" int
mi355x_bz2_read_stream_header( const uint8_t* bytes, uint64_t size, uint64_t bitOffset )
{
    /* bzip2::readBzip2Header, src/indexed_bzip2/bzip2.hpp:114-142 (works at any bit alignment) */
    if ( bytes == nullptr || bitOffset + 32 > size * 8 ) return 0;
    uint8_t b[4];
    const uint64_t byte = bitOffset >> 3;
    const unsigned sh = (unsigned)( bitOffset & 7 );
    for ( int i = 0; i < 4; ++i ) {
        const unsigned hi = bytes[byte + i];
        const unsigned lo = ( sh != 0 && byte + i + 1 < size ) ? bytes[byte + i + 1] : 0;
        b[i] = (uint8_t)( ( ( hi << 8 | lo ) >> ( 8 - sh ) ) & 0xFF );
    }
    if ( b[0] != 'B' || b[1] != 'Z' || b[2] != 'h' ) return 0;
    if ( b[3] < '1' || b[3] > '9' ) return 0;
    return b[3] - '0';
}
