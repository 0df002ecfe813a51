/**
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
 *
 * Driver around the REAL rapidgzip::Bzip2Chunk<ChunkData>::decodeChunk
 * (src/rapidgzip/chunkdecoding/Bzip2Chunk.hpp:211-268) compiled from the reference's own headers where they lie
 * (oracle/Makefile, target `refchunk` -> oracle/_ref/ref_chunk).  No reference source is copied.
 * Used by tests/golden/make_golden_chunks.py to record what the reference returns for the chunk requests of
 * tests/test_gpu_chunk.py; mi355x_bz2_decode_chunk is then compared with those vectors.
 *
 *   ref_chunk <file> <chunkOffsetBits> <untilOffsetBits> <maxDecodedBytes>
 * prints one JSON object: {"status": "ok", "encoded_offset_bits", "encoded_end_bits", "decoded_size", "fnv64",
 * "stopped_preemptively", "boundaries": [[bits, bytes]...], "footers": [[bits, bytes]...]} or
 * {"status": "NoBlockInRange" | "<exception type>", "what": "..."}.
 */
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <typeinfo>

#include <filereader/Shared.hpp>
#include <filereader/Standard.hpp>
#include <chunkdecoding/Bzip2Chunk.hpp>

using namespace rapidgzip;

int
main( int argc, char** argv )
{
    if ( argc < 5 ) {
        std::fprintf( stderr, "usage: ref_chunk <file> <chunkOffsetBits> <untilOffsetBits> <maxDecodedBytes>\n" );
        return 2;
    }
    const size_t chunkOffset = std::strtoull( argv[2], nullptr, 10 );
    const size_t untilOffset = std::strtoull( argv[3], nullptr, 10 );
    const size_t maxDecoded = std::strtoull( argv[4], nullptr, 10 );
    try {
        UniqueFileReader file = std::make_unique<SharedFileReader>(
            std::make_unique<StandardFileReader>( std::string( argv[1] ) ) );
        const std::atomic<bool> cancel{ false };
        ChunkData::Configuration configuration;
        configuration.encodedOffsetInBits = chunkOffset;
        configuration.fileType = FileType::BZIP2;
        configuration.crc32Enabled = false;
        auto chunk = Bzip2Chunk<ChunkData>::decodeChunk( std::move( file ), chunkOffset, untilOffset, cancel,
                                                         configuration, maxDecoded );
        uint64_t h = 0xcbf29ce484222325ULL;
        size_t total = 0;
        for ( deflate::DecodedData::Iterator it( chunk ); static_cast<bool>( it ); ++it ) {
            const auto [pointer, size] = *it;
            const auto* const bytes = static_cast<const uint8_t*>( pointer );
            for ( size_t i = 0; i < size; ++i ) {
                h ^= bytes[i];
                h *= 0x100000001b3ULL;
            }
            total += size;
        }
        std::printf( "{\"status\": \"ok\", \"encoded_offset_bits\": %zu, \"encoded_end_bits\": %zu, \"decoded_size\": %zu, "
                     "\"data_bytes\": %zu, \"fnv64\": \"%016llx\", \"stopped_preemptively\": %s, \"boundaries\": [",
                     chunk.encodedOffsetInBits, chunk.encodedOffsetInBits + chunk.encodedSizeInBits,
                     chunk.decodedSizeInBytes, total, (unsigned long long)h, chunk.stoppedPreemptively ? "true" : "false" );
        for ( size_t i = 0; i < chunk.blockBoundaries.size(); ++i ) {
            std::printf( "%s[%zu, %zu]", i ? ", " : "", chunk.blockBoundaries[i].encodedOffset,
                         chunk.blockBoundaries[i].decodedOffset );
        }
        std::printf( "], \"footers\": [" );
        for ( size_t i = 0; i < chunk.footers.size(); ++i ) {
            std::printf( "%s[%zu, %zu]", i ? ", " : "", chunk.footers[i].blockBoundary.encodedOffset,
                         chunk.footers[i].blockBoundary.decodedOffset );
        }
        std::printf( "]}\n" );
    } catch ( const NoBlockInRange& e ) {
        std::printf( "{\"status\": \"NoBlockInRange\"}\n" );
    } catch ( const std::exception& e ) {
        std::printf( "{\"status\": \"%s\"}\n", typeid( e ).name() );
    }
    return 0;
}
