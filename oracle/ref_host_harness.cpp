/**
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
 *
 * Scripted driver around the reference's host-side scheduler classes, compiled from the reference's own headers where
 * they lie (oracle/Makefile, target _ref/ref_host).  No reference source is copied.
 *   rapidgzip::FetchingStrategy::FetchNextAdaptive   src/core/Prefetcher.hpp:82-217
 *   rapidgzip::BlockMap                              src/core/BlockMap.hpp:26-295
 * tests/golden/make_golden_host.py feeds it seeded scripts and records the answers in tests/golden/host_vectors.txt;
 * tests/native/host_known_answers.cpp replays the same scripts on this repository's own classes.
 *
 * stdin: one command per line, stdout: one answer per line.
 *   strategy <memory>          new FetchNextAdaptive
 *   f <index>                  fetch                    -> "ok"
 *   p <max>                    prefetch                 -> "<first> <count>" (lists are always consecutive), "- 0" if empty
 *   s                          isSequential             -> "0" | "1"
 *   map                        new BlockMap
 *   push <enc> <encSize> <decSize>                      -> "<decoded offset>" | "EXC"
 *   find <dataOffset>                                   -> "<index> <enc> <encSize> <dec> <decSize> <contains>"
 *   finalize                                            -> "ok"
 *   state                                               -> "<finalized> <empty> <dataBlockCount> <backEnc> <backDec>"
 *   dump                                                -> "<n> enc:dec ..."
 *   set <enc:dec> ...          setBlockOffsets          -> "ok" | "EXC"
 */
#include <cstdio>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>

#include <BlockMap.hpp>
#include <Prefetcher.hpp>

using namespace rapidgzip;

int
main()
{
    std::unique_ptr<FetchingStrategy::FetchNextAdaptive> strategy;
    std::unique_ptr<BlockMap> map;
    std::string line;
    while ( std::getline( std::cin, line ) ) {
        std::istringstream in( line );
        std::string cmd;
        in >> cmd;
        try {
            if ( cmd == "strategy" ) {
                size_t memory = 3;
                in >> memory;
                strategy = std::make_unique<FetchingStrategy::FetchNextAdaptive>( memory );
                std::printf( "ok\n" );
            } else if ( cmd == "f" ) {
                size_t index = 0;
                in >> index;
                strategy->fetch( index );
                std::printf( "ok\n" );
            } else if ( cmd == "p" ) {
                size_t maxAmount = 0;
                in >> maxAmount;
                const auto list = strategy->prefetch( maxAmount );
                bool consecutive = true;
                for ( size_t i = 1; i < list.size(); ++i ) consecutive &= list[i] == list[i - 1] + 1;
                if ( list.empty() ) std::printf( "- 0\n" );
                else if ( !consecutive ) std::printf( "NOT-CONSECUTIVE\n" );
                else std::printf( "%zu %zu\n", list.front(), list.size() );
            } else if ( cmd == "s" ) {
                std::printf( "%d\n", strategy->isSequential() ? 1 : 0 );
            } else if ( cmd == "map" ) {
                map = std::make_unique<BlockMap>();
                std::printf( "ok\n" );
            } else if ( cmd == "push" ) {
                size_t enc = 0, encSize = 0, decSize = 0;
                in >> enc >> encSize >> decSize;
                std::printf( "%zu\n", map->push( enc, encSize, decSize ) );
            } else if ( cmd == "find" ) {
                size_t offset = 0;
                in >> offset;
                const auto info = map->findDataOffset( offset );
                std::printf( "%zu %zu %zu %zu %zu %d\n", info.blockIndex, info.encodedOffsetInBits, info.encodedSizeInBits,
                             info.decodedOffsetInBytes, info.decodedSizeInBytes, info.contains( offset ) ? 1 : 0 );
            } else if ( cmd == "finalize" ) {
                map->finalize();
                std::printf( "ok\n" );
            } else if ( cmd == "state" ) {
                const bool empty = map->empty();
                std::printf( "%d %d %zu %zu %zu\n", map->finalized() ? 1 : 0, empty ? 1 : 0, map->dataBlockCount(),
                             empty ? size_t( 0 ) : map->back().first, empty ? size_t( 0 ) : map->back().second );
            } else if ( cmd == "dump" ) {
                const auto offsets = map->blockOffsets();
                std::printf( "%zu", offsets.size() );
                for ( const auto& [enc, dec] : offsets ) std::printf( " %zu:%zu", enc, dec );
                std::printf( "\n" );
            } else if ( cmd == "set" ) {
                std::map<size_t, size_t> offsets;
                std::string pair;
                while ( in >> pair ) {
                    const auto colon = pair.find( ':' );
                    offsets.emplace( std::stoull( pair.substr( 0, colon ) ), std::stoull( pair.substr( colon + 1 ) ) );
                }
                map->setBlockOffsets( offsets );
                std::printf( "ok\n" );
            } else {
                std::printf( "?\n" );
            }
        } catch ( const std::exception& ) {
            std::printf( "EXC\n" );
        }
    }
    return 0;
}
