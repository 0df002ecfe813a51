/**
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
 *
 * Thin driver around the REAL reference implementation (header-only C++ under /root/reference/src), compiled
 * where those headers lie by oracle/Makefile into oracle/_ref/ref_bz2 (git-ignored, travels to the GPU box).
 * No reference source is copied into this repository: this file only calls the reference's public classes
 *   indexed_bzip2::ParallelBZ2Reader   (src/indexed_bzip2/ParallelBZ2Reader.hpp:39-498)
 *   indexed_bzip2::BZ2Reader           (src/indexed_bzip2/BZ2Reader.hpp:29-492)
 *   bzip2::Block                       (src/indexed_bzip2/bzip2.hpp:145-461)
 * Uses:
 *   1. pin oracle/bz2_oracle.c (our CPU restatement) against the reference itself (tests/, tools/make_golden.py);
 *   2. CPU baseline "kind": "reference" in bench.py (decode-only, like `ibzip2 -d -o /dev/null`, src/tools/ibzip2.cpp:397).
 *
 * Commands (all print machine-readable lines on stdout):
 *   map    <file> [P]           block-offset map of ParallelBZ2Reader::blockOffsets(): "<bits> <bytes>" per line
 *   smap   <file>               same for the serial BZ2Reader
 *   blocks <file>               one line per data block: offset size headerCRC calcCRC decodedSize fnv64(data)
 *   probe  <file> <bitOffset>   decode one block at an arbitrary bit offset; prints OK ... or EXC <type> <what>
 *   decode <file> <P> <out>     full decode to <out> ("-" = discard)
 *   bench  <file> <P> <reps> [maxBytes]   decode-only timing; prints JSON
 *   pread  <file> <index> <positions> <P> <bytes>   config 5 of BASELINE.json on the reference: ParallelBZ2Reader with the
 *                               block map of <index> ("<bits> <bytes>" lines, as `map` prints) imported through
 *                               setBlockOffsets (ParallelBZ2Reader.hpp:365-378), then seek + read(<bytes>) at every
 *                               decoded offset listed in <positions> (ParallelBZ2Reader.hpp:271-325, 167-265); prints
 *                               JSON with the latency percentiles and FNV-64 / zlib CRC-32 over all bytes read
 */
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <typeinfo>
#include <vector>

#include <BitReader.hpp>
#include <filereader/Standard.hpp>
#include <BZ2Reader.hpp>
#include <ParallelBZ2Reader.hpp>
#include <bzip2.hpp>

using namespace rapidgzip;
using namespace indexed_bzip2;

static uint64_t
fnv64( const uint8_t* p, size_t n )
{
    uint64_t h = 0xcbf29ce484222325ULL;
    for ( size_t i = 0; i < n; ++i ) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}

static std::unique_ptr<FileReader>
openFile( const char* path )
{
    return std::make_unique<StandardFileReader>( std::string( path ) );
}

/* The same call sequence as BZ2BlockFetcher::decodeBlock (src/indexed_bzip2/BZ2BlockFetcher.hpp:85-138),
 * which is private there, expressed with bzip2::Block's public interface. */
struct OneBlock
{
    size_t offset{ 0 }, size{ 0 };
    uint32_t headerCRC{ 0 }, calcCRC{ 0 };
    bool eos{ false }, eof{ false };
    std::vector<uint8_t> data;
};

static OneBlock
decodeOne( const bzip2::BitReader& baseReader, size_t blockOffset )
{
    bzip2::BitReader bitReader( baseReader );
    bitReader.seekTo( blockOffset );
    bzip2::Block block( bitReader );
    OneBlock r;
    r.offset = blockOffset;
    r.eos = block.eos();
    r.eof = block.eof();
    r.headerCRC = block.headerCRC();
    if ( block.eos() ) {
        r.size = block.encodedSizeInBits;
        return r;
    }
    block.readBlockData();
    size_t decoded = 0;
    do {
        if ( r.data.empty() ) {
            r.data.resize( 9 * 100'000 + 255 );
        } else {
            r.data.resize( r.data.size() * 2 );
        }
        decoded += block.read( r.data.size() - 255U - decoded, reinterpret_cast<char*>( r.data.data() ) + decoded );
    } while ( !block.eob() );
    r.data.resize( decoded );
    r.size = block.encodedSizeInBits;
    r.calcCRC = block.dataCRC();
    return r;
}

int
main( int argc, char** argv )
{
    if ( argc < 3 ) {
        std::fprintf( stderr, "usage: %s map|smap|blocks|probe|decode|bench <file> ...\n", argv[0] );
        return 2;
    }
    const std::string cmd = argv[1];
    const char* path = argv[2];
    try {
        if ( cmd == "map" ) {
            const size_t P = argc > 3 ? std::stoul( argv[3] ) : 1;
            ParallelBZ2Reader reader( openFile( path ), P );
            for ( const auto& [bits, bytes] : reader.blockOffsets() ) {
                std::printf( "%zu %zu\n", bits, bytes );
            }
            return 0;
        }
        if ( cmd == "smap" ) {
            BZ2Reader reader( openFile( path ) );
            for ( const auto& [bits, bytes] : reader.blockOffsets() ) {
                std::printf( "%zu %zu\n", bits, bytes );
            }
            return 0;
        }
        if ( cmd == "blocks" ) {
            std::map<size_t, size_t> offsets;
            {
                ParallelBZ2Reader reader( openFile( path ), 1 );
                offsets = reader.blockOffsets();
            }
            bzip2::BitReader base( openFile( path ) );
            for ( auto it = offsets.begin(); it != offsets.end(); ++it ) {
                auto nit = std::next( it );
                if ( nit == offsets.end() || nit->second == it->second ) {
                    continue;  /* EOS / sentinel */
                }
                const auto b = decodeOne( base, it->first );
                std::printf( "%zu %zu %08x %08x %zu %016llx\n", b.offset, b.size, b.headerCRC, b.calcCRC,
                             b.data.size(), (unsigned long long)fnv64( b.data.data(), b.data.size() ) );
            }
            return 0;
        }
        if ( cmd == "probe" ) {
            const size_t off = std::stoull( argv[3] );
            bzip2::BitReader base( openFile( path ) );
            try {
                const auto b = decodeOne( base, off );
                std::printf( "OK %zu %zu %08x %08x %zu %016llx %d %d\n", b.offset, b.size, b.headerCRC, b.calcCRC,
                             b.data.size(), (unsigned long long)fnv64( b.data.data(), b.data.size() ),
                             int( b.eos ), int( b.eof ) );
            } catch ( const std::exception& e ) {
                std::string what = e.what();
                for ( auto& c : what ) { if ( c == '\n' ) c = ' '; }
                std::printf( "EXC %s %s\n", typeid( e ).name(), what.c_str() );
            }
            return 0;
        }
        if ( cmd == "decode" ) {
            const size_t P = std::stoul( argv[3] );
            const std::string out = argv[4];
            ParallelBZ2Reader reader( openFile( path ), P );
            FILE* f = out == "-" ? nullptr : std::fopen( out.c_str(), "wb" );
            size_t total = 0;
            std::vector<char> buf( 4U << 20U );
            while ( true ) {
                const auto n = reader.read( -1, f ? buf.data() : nullptr, buf.size() );
                if ( n == 0 ) break;
                if ( f ) std::fwrite( buf.data(), 1, n, f );
                total += n;
            }
            if ( f ) std::fclose( f );
            std::printf( "%zu\n", total );
            return 0;
        }
        if ( cmd == "bench" ) {
            const size_t P = std::stoul( argv[3] );
            const int reps = argc > 4 ? std::stoi( argv[4] ) : 1;
            const size_t maxBytes = argc > 5 ? std::stoull( argv[5] ) : std::numeric_limits<size_t>::max();
            double best = 1e300;
            size_t total = 0;
            for ( int r = 0; r < reps; ++r ) {
                const auto t0 = std::chrono::steady_clock::now();
                if ( P == 0 ) {
                    /* P == 0 here selects the SERIAL reader (what the Python wrapper uses for parallelization=1,
                     * python/indexed_bzip2/indexed_bzip2.pyx:295-296). */
                    BZ2Reader reader( openFile( path ) );
                    total = reader.read( -1, nullptr, maxBytes );
                } else {
                    ParallelBZ2Reader reader( openFile( path ), P );
                    total = reader.read( -1, nullptr, maxBytes );
                }
                const double dt = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
                best = std::min( best, dt );
            }
            std::printf( "{\"decoded_bytes\": %zu, \"seconds\": %.6f, \"MBps\": %.3f, \"P\": %zu, \"reps\": %d}\n",
                         total, best, total / best / 1e6, P, reps );
            return 0;
        }
        if ( cmd == "pread" ) {
            if ( argc < 7 ) {
                std::fprintf( stderr, "usage: pread <file> <index> <positions> <P> <bytes>\n" );
                return 2;
            }
            std::map<size_t, size_t> offsets;
            {
                std::ifstream in( argv[3] );
                size_t bits = 0, bytes = 0;
                while ( in >> bits >> bytes ) offsets.emplace( bits, bytes );
            }
            std::vector<size_t> positions;
            {
                std::ifstream in( argv[4] );
                size_t at = 0;
                while ( in >> at ) positions.push_back( at );
            }
            const size_t P = std::stoul( argv[5] );
            const size_t nBytes = std::stoull( argv[6] );
            const auto tOpen = std::chrono::steady_clock::now();
            ParallelBZ2Reader reader( openFile( path ), P );
            reader.setBlockOffsets( offsets );
            std::vector<char> buffer( nBytes );
            std::vector<double> latencies;
            latencies.reserve( positions.size() );
            uint64_t hash = 0xcbf29ce484222325ULL;
            /* and zlib's CRC-32 (reflected 0xEDB88320) over all bytes read, which the caller can recompute at C speed */
            uint32_t zcrcTable[256];
            for ( uint32_t i = 0; i < 256; ++i ) {
                uint32_t c = i;
                for ( int k = 0; k < 8; ++k ) c = ( c & 1U ) ? ( c >> 1U ) ^ 0xEDB88320U : ( c >> 1U );
                zcrcTable[i] = c;
            }
            uint32_t zcrc = 0xFFFFFFFFU;
            size_t total = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for ( const auto at : positions ) {
                const auto a = std::chrono::steady_clock::now();
                reader.seek( static_cast<long long>( at ) );
                const auto n = reader.read( -1, buffer.data(), nBytes );
                latencies.push_back( std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - a ).count() );
                for ( size_t i = 0; i < n; ++i ) {
                    hash ^= static_cast<uint8_t>( buffer[i] );
                    hash *= 0x100000001b3ULL;
                    zcrc = zcrcTable[( zcrc ^ static_cast<uint8_t>( buffer[i] ) ) & 0xFFU] ^ ( zcrc >> 8U );
                }
                total += n;
            }
            const double wall = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
            std::sort( latencies.begin(), latencies.end() );
            const auto pct = [&] ( double q ) {
                return latencies.empty() ? 0. : latencies[std::min( latencies.size() - 1, (size_t)( q * latencies.size() ) )];
            };
            double mean = 0;
            for ( const auto v : latencies ) mean += v;
            mean /= latencies.empty() ? 1 : latencies.size();
            std::printf( "{\"reads\": %zu, \"read_bytes\": %zu, \"P\": %zu, \"bytes_read\": %zu, \"fnv64\": \"%016llx\", \"zlib_crc32\": %u, "
                         "\"wall_seconds\": %.4f, \"open_and_import_ms\": %.2f, "
                         "\"latency_ms\": {\"p50\": %.3f, \"p95\": %.3f, \"p99\": %.3f, \"mean\": %.3f}}\n",
                         positions.size(), nBytes, P, total, (unsigned long long)hash, zcrc ^ 0xFFFFFFFFU, wall,
                         std::chrono::duration<double, std::milli>( t0 - tOpen ).count(), pct( 0.50 ), pct( 0.95 ), pct( 0.99 ), mean );
            return 0;
        }
    } catch ( const std::exception& e ) {
        std::string what = e.what();
        for ( auto& c : what ) { if ( c == '\n' ) c = ' '; }
        std::printf( "EXC %s %s\n", typeid( e ).name(), what.c_str() );
        return 1;
    }
    std::fprintf( stderr, "unknown command\n" );
    return 2;
}
