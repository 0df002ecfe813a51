/**
 * Known answers for the host-side classes of indexed_bzip2_amd/csrc/bz2_host.hpp.
 *
 * 1. tests/golden/host_vectors.txt: scripts + the answers of the REAL reference classes (FetchNextAdaptive,
 *    src/core/Prefetcher.hpp:82-217; BlockMap, src/core/BlockMap.hpp:26-295), recorded by tests/golden/make_golden_host.py
 *    from oracle/_ref/ref_host.  They include the sequences of src/tests/core/testPrefetcher.cpp:239-285 and 306-313.
 *    Replayed here on SequentialityTracker and BlockIndex: every answer has to be the same.
 * 2. src/tests/core/testCache.cpp:11-24 (replacing a key's value is not an eviction and leaves no unused entry), on RunCache.
 * 3. src/tests/core/testPrefetcher.cpp:1075-1100: a simulated reader over tracker + cache -- sequential access hits the
 *    cache > 99.5 % of the time, backward access prefetches nothing.
 *
 * usage: host_known_answers <host_vectors.txt>
 */
#include <cstdio>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>

#include "../../indexed_bzip2_amd/csrc/bz2_host.hpp"

using namespace mi355x;

namespace
{
int failures = 0;

void
check( bool ok, const std::string& what )
{
    if ( !ok ) {
        ++failures;
        if ( failures < 20 ) std::fprintf( stderr, "FAILED: %s\n", what.c_str() );
    }
}

struct TestRun
{
    size_t firstBlock, blocks;
    [[nodiscard]] size_t first() const { return firstBlock; }
    [[nodiscard]] size_t count() const { return blocks; }
};

std::string
answer( const std::string& line, std::unique_ptr<SequentialityTracker>& tracker, std::unique_ptr<BlockIndex>& index )
{
    std::istringstream in( line );
    std::string cmd;
    in >> cmd;
    char buffer[256];
    try {
        if ( cmd == "strategy" ) {
            size_t memory = 3;
            in >> memory;
            tracker = std::make_unique<SequentialityTracker>( memory );
            return "ok";
        }
        if ( cmd == "f" ) {
            size_t block = 0;
            in >> block;
            tracker->note( block );
            return "ok";
        }
        if ( cmd == "p" ) {
            size_t limit = 0;
            in >> limit;
            const auto range = tracker->ahead( limit );
            if ( range.count == 0 ) return "- 0";
            std::snprintf( buffer, sizeof( buffer ), "%zu %zu", range.first, range.count );
            return buffer;
        }
        if ( cmd == "s" ) return tracker->inOrder() ? "1" : "0";
        if ( cmd == "map" ) {
            index = std::make_unique<BlockIndex>();
            return "ok";
        }
        if ( cmd == "push" ) {
            uint64_t bits = 0, bitLength = 0, byteLength = 0;
            in >> bits >> bitLength >> byteLength;
            std::snprintf( buffer, sizeof( buffer ), "%llu", (unsigned long long)index->append( bits, bitLength, byteLength ) );
            return buffer;
        }
        if ( cmd == "find" ) {
            uint64_t offset = 0;
            in >> offset;
            const auto span = index->locate( offset );
            std::snprintf( buffer, sizeof( buffer ), "%zu %llu %llu %llu %llu %d", span.ordinal, (unsigned long long)span.bits,
                           (unsigned long long)span.bitLength, (unsigned long long)span.bytes,
                           (unsigned long long)span.byteLength, span.covers( offset ) ? 1 : 0 );
            return buffer;
        }
        if ( cmd == "finalize" ) {
            index->seal();
            return "ok";
        }
        if ( cmd == "state" ) {
            const bool empty = index->empty();
            const auto last = empty ? std::pair<uint64_t, uint64_t>{ 0, 0 } : index->last();
            std::snprintf( buffer, sizeof( buffer ), "%d %d %zu %llu %llu", index->sealed() ? 1 : 0, empty ? 1 : 0,
                           index->dataBlocks(), (unsigned long long)last.first, (unsigned long long)last.second );
            return buffer;
        }
        if ( cmd == "dump" ) {
            const auto pairs = index->snapshot();
            std::string result = std::to_string( pairs.size() );
            for ( const auto& [bits, bytes] : pairs ) result += " " + std::to_string( bits ) + ":" + std::to_string( bytes );
            return result;
        }
        if ( cmd == "set" ) {
            BlockIndex::Pairs pairs;
            std::string pair;
            while ( in >> pair ) {
                const auto colon = pair.find( ':' );
                pairs.emplace_back( std::stoull( pair.substr( 0, colon ) ), std::stoull( pair.substr( colon + 1 ) ) );
            }
            std::sort( pairs.begin(), pairs.end() );
            index->assign( pairs );
            return "ok";
        }
    } catch ( const std::exception& ) {
        return "EXC";
    }
    return "?";
}

void
replay( const char* path )
{
    std::ifstream file( path );
    check( file.good(), std::string( "cannot open " ) + path );
    std::unique_ptr<SequentialityTracker> tracker;
    std::unique_ptr<BlockIndex> index;
    std::string line;
    size_t count = 0;
    while ( std::getline( file, line ) ) {
        const auto arrow = line.find( " => " );
        if ( arrow == std::string::npos ) continue;
        const auto command = line.substr( 0, arrow ), want = line.substr( arrow + 4 );
        const auto got = answer( command, tracker, index );
        check( got == want, "line " + std::to_string( count + 1 ) + ": " + command + " -> " + got + ", reference: " + want );
        ++count;
    }
    check( count > 10000, "too few vectors: " + std::to_string( count ) );
    std::printf( "%zu reference answers replayed\n", count );
}

/* testCache.cpp:11-24 */
void
cacheReinsertion()
{
    RunCache<TestRun> cache( /* budget in blocks */ 2 );
    cache.insert( std::make_shared<TestRun>( TestRun{ 2, 1 } ) );
    cache.insert( std::make_shared<TestRun>( TestRun{ 1, 1 } ) );
    cache.insert( std::make_shared<TestRun>( TestRun{ 1, 1 } ) );   /* replacing must not evict */
    check( cache.statistics().unusedRuns == 0, "re-insertion counted an unused eviction" );
    check( cache.statistics().evictions == 0 && cache.blocks() == 2 && cache.runs() == 2, "re-insertion evicted" );
    check( cache.covers( 1 ) && cache.covers( 2 ) && !cache.covers( 3 ) && !cache.covers( 0 ), "coverage after re-insertion" );
    cache.insert( std::make_shared<TestRun>( TestRun{ 5, 1 } ) );   /* now the least recently used run (2) goes, unused */
    check( !cache.covers( 2 ) && cache.covers( 1 ) && cache.covers( 5 ), "least recently used run evicted" );
    check( cache.statistics().unusedRuns == 1 && cache.statistics().evictions == 1, "eviction statistics" );
    check( cache.find( 1 ) != nullptr && cache.find( 7 ) == nullptr, "find" );
    check( cache.statistics().hits == 1 && cache.statistics().misses == 1, "hit statistics" );
    cache.insert( std::make_shared<TestRun>( TestRun{ 10, 4 } ) );  /* bigger than the budget: kept alone */
    check( cache.runs() == 1 && cache.covers( 13 ) && !cache.covers( 14 ), "oversized run" );
    check( cache.firstGap( 10 ) == 14 && cache.firstGap( 3 ) == 3, "firstGap" );
}

/* testPrefetcher.cpp:1075-1100: one block per access, prefetches complete at once, parallelization 16 */
void
simulatedAccess( bool backward )
{
    SequentialityTracker tracker;
    RunCache<TestRun> cache( 2 * 16 );
    size_t prefetched = 0, hits = 0;
    const size_t n = 1000;
    for ( size_t k = 0; k < n; ++k ) {
        const size_t block = backward ? n - 1 - k : k;
        if ( cache.find( block ) ) ++hits; else cache.insert( std::make_shared<TestRun>( TestRun{ block, 1 } ) );
        tracker.note( block );
        const auto range = tracker.ahead( 16 );
        for ( size_t b = range.first; b < range.first + range.count; ++b ) {
            if ( !cache.covers( b ) ) {
                cache.insert( std::make_shared<TestRun>( TestRun{ b, 1 } ) );
                ++prefetched;
            }
        }
    }
    if ( backward ) {
        check( prefetched <= 16 + 2, "backward access kept prefetching: " + std::to_string( prefetched ) );
    } else {
        check( (double)hits / n > 0.995, "sequential hit rate " + std::to_string( (double)hits / n ) );
    }
}
}  // namespace

int
main( int argc, char** argv )
{
    if ( argc < 2 ) {
        std::fprintf( stderr, "usage: host_known_answers <host_vectors.txt>\n" );
        return 2;
    }
    replay( argv[1] );
    cacheReinsertion();
    simulatedAccess( false );
    simulatedAccess( true );
    if ( failures != 0 ) {
        std::printf( "%d FAILURES\n", failures );
        return 1;
    }
    std::printf( "known answers ok\n" );
    return 0;
}
