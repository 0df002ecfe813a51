/**
 * bz2_chunk.cpp -- chunk decoding behind C ABI section 4 of include/mi355x_bz2.h.
 *
 * Counterpart of rapidgzip's bzip2 chunk decoder:
 *   Bzip2Chunk::decodeChunk              src/rapidgzip/chunkdecoding/Bzip2Chunk.hpp:215-268
 *   Bzip2Chunk::decodeUnknownBzip2Chunk  src/rapidgzip/chunkdecoding/Bzip2Chunk.hpp:34-212
 * A chunk is the run of consecutive blocks (through end-of-stream blocks and the headers of following streams) that
 * starts at or behind `chunk_offset_bits` and whose blocks start before `until_offset_bits`.  The reference decodes
 * them one after the other because a block's end is only known once it is decoded; here all block magics of the range
 * are decoded as ONE GPU batch and the chain is then walked on the host over the per-block records.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mi355x_bz2.h"
#include "bz2_host.hpp"

namespace
{
uint64_t
readBits( const uint8_t* bytes, uint64_t sizeBits, uint64_t& pos, unsigned count, bool& eof )
{
    uint64_t value = 0;
    for ( unsigned i = 0; i < count; ++i ) {
        if ( pos >= sizeBits ) {
            eof = true;
            return 0;
        }
        value = ( value << 1 ) | ( ( bytes[pos >> 3] >> ( 7 - ( pos & 7 ) ) ) & 1u );
        ++pos;
    }
    return value;
}

struct Walk
{
    std::vector<uint32_t> blocks;                       /* indices into the candidate list, in chain order */
    std::vector<mi355x_bz2_chunk_boundary> footers;
    uint64_t end{ 0 };
    uint64_t decoded{ 0 };
    bool stoppedPreemptively{ false };
    int status{ MI355X_BZ2_OK };
};

/** decodeUnknownBzip2Chunk's loop over the records of the decoded candidates (Bzip2Chunk.hpp:66-189). */
Walk
walkChain( const uint8_t* bytes, uint64_t size, uint64_t start, uint64_t until, uint64_t maxDecoded,
           const std::vector<uint64_t>& offsets, const std::vector<mi355x_bz2_block_result>& results )
{
    Walk w;
    const uint64_t sizeBits = size * 8;
    uint64_t pos = start;
    bool isAtStreamEnd = false;
    uint64_t nextBlockOffset = start;
    for ( ;; ) {
        if ( isAtStreamEnd ) {
            if ( ( pos & 7 ) != 0 || mi355x_bz2_read_stream_header( bytes, size, pos ) == 0 ) {
                w.status = MI355X_BZ2_ERR_STREAM_HEADER;
                return w;
            }
            pos += 32;
            isAtStreamEnd = false;
        }
        nextBlockOffset = pos;
        if ( w.decoded >= maxDecoded ) {
            w.stoppedPreemptively = true;
            break;
        }
        /* block header: magic + CRC */
        uint64_t p = pos;
        bool eof = false;
        const uint64_t hi = readBits( bytes, sizeBits, p, 24, eof );
        const uint64_t lo = readBits( bytes, sizeBits, p, 24, eof );
        (void)readBits( bytes, sizeBits, p, 32, eof );
        if ( eof ) {
            if ( pos == start ) break;   /* nothing at all at the very first position: an empty chunk */
            w.status = MI355X_BZ2_ERR_EOF;
            return w;
        }
        const uint64_t magic = ( hi << 24 ) | lo;
        const bool eos = magic == MI355X_BZ2_MAGIC_EOS;
        if ( !eos && magic != MI355X_BZ2_MAGIC_BLOCK ) {
            w.status = MI355X_BZ2_ERR_BAD_MAGIC;
            return w;
        }
        /* the reference decodes the block before it looks at the stop condition; failures of a block that would have
         * been dropped anyway therefore still fail the attempt */
        uint32_t candidate = 0;
        if ( !eos ) {
            const auto it = std::lower_bound( offsets.begin(), offsets.end(), pos );
            if ( it == offsets.end() || *it != pos ) {
                /* a data block at or behind `until` was not part of the batch: it is never emitted */
                if ( pos >= until ) break;
                w.status = MI355X_BZ2_ERR_LOGIC;
                return w;
            }
            candidate = (uint32_t)( it - offsets.begin() );
            if ( results[candidate].status != MI355X_BZ2_OK ) {
                w.status = results[candidate].status;
                return w;
            }
        }
        if ( ( nextBlockOffset >= until && !eos ) || nextBlockOffset == until ) break;
        if ( eos ) {
            if ( ( p & 7 ) != 0 ) {
                (void)readBits( bytes, sizeBits, p, 8 - (unsigned)( p & 7 ), eof );
                if ( eof ) {
                    w.status = MI355X_BZ2_ERR_EOF;
                    return w;
                }
            }
            w.footers.push_back( { p, w.decoded } );
            isAtStreamEnd = true;
            pos = p;
            if ( pos >= sizeBits ) {
                nextBlockOffset = pos;
                break;
            }
            continue;
        }
        if ( results[candidate].decoded_size > ( 64ull << 20 ) ) {
            w.status = MI355X_BZ2_ERR_OUTPUT_CAPACITY;   /* "more than 64 MiB ... not supported", Bzip2Chunk.hpp:181-184 */
            return w;
        }
        w.blocks.push_back( candidate );
        w.decoded += results[candidate].decoded_size;
        pos += results[candidate].encoded_size_bits;
    }
    w.end = nextBlockOffset;
    return w;
}
}  // namespace

extern "C" int
mi355x_bz2_decode_chunk( mi355x_bz2_ctx* ctx, const uint8_t* bytes, uint64_t size,
                         uint64_t chunkOffset, uint64_t untilOffset, uint64_t maxDecoded,
                         mi355x_bz2_chunk_result* result,
                         mi355x_bz2_block_result* blocks, uint32_t blocksCapacity,
                         mi355x_bz2_chunk_boundary* footers, uint32_t footersCapacity )
{
    if ( ctx == nullptr || bytes == nullptr || result == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    std::memset( result, 0, sizeof( *result ) );
    result->encoded_offset_bits = chunkOffset;
    result->encoded_end_bits = chunkOffset;
    const uint64_t sizeBits = size * 8;
    if ( chunkOffset > sizeBits ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;

    /* every block magic that starts in [chunkOffset, untilOffset): one GPU batch */
    std::vector<uint64_t> offsets;
    {
        const uint64_t beginByte = chunkOffset / 8;
        const uint64_t endByte = std::min<uint64_t>( size, untilOffset / 8 + 1 );
        std::vector<uint64_t> found;
        mi355x::scanMagicRange( bytes, size, MI355X_BZ2_MAGIC_BLOCK, beginByte, std::max( beginByte, endByte ), found );
        for ( const auto o : found ) {
            if ( o >= chunkOffset && o < untilOffset ) offsets.push_back( o );
        }
        std::sort( offsets.begin(), offsets.end() );
        if ( offsets.size() > 65535 ) offsets.resize( 65535 );
    }
    std::vector<mi355x_bz2_block_result> results( offsets.size() );
    uint64_t total = 0;
    if ( !offsets.empty() ) {
        const int rc = mi355x_bz2_decode_batch( ctx, offsets.data(), (uint32_t)offsets.size(), results.data(), &total );
        if ( rc != MI355X_BZ2_OK ) return rc;
    }

    /* tryToDecode( chunkOffset ), then every block magic in the range as a start (Bzip2Chunk.hpp:225-260) */
    Walk walk = walkChain( bytes, size, chunkOffset, untilOffset, maxDecoded, offsets, results );
    if ( walk.status != MI355X_BZ2_OK ) {
        bool found = false;
        for ( const auto start : offsets ) {
            if ( start == chunkOffset ) continue;   /* that was the first attempt */
            walk = walkChain( bytes, size, start, untilOffset, maxDecoded, offsets, results );
            if ( walk.status == MI355X_BZ2_OK ) {
                result->encoded_offset_bits = start;
                found = true;
                break;
            }
        }
        if ( !found ) {
            result->status = MI355X_BZ2_ERR_NO_BLOCK_IN_RANGE;   /* NoBlockInRange, Bzip2Chunk.hpp:262-266 */
            return MI355X_BZ2_OK;
        }
    }

    /* the chunk's bytes must be one contiguous piece of the batch output: true when every block of the batch that
     * produced output is on the chain; otherwise decode just the chain again (only after a failed first attempt) */
    bool contiguous = true;
    {
        uint64_t expect = walk.blocks.empty() ? 0 : results[walk.blocks.front()].data_offset;
        for ( const auto i : walk.blocks ) {
            if ( results[i].data_offset != expect ) { contiguous = false; break; }
            expect += results[i].decoded_size;
        }
    }
    uint64_t dataOffset = walk.blocks.empty() ? 0 : results[walk.blocks.front()].data_offset;
    if ( !contiguous ) {
        std::vector<uint64_t> chainOffsets;
        for ( const auto i : walk.blocks ) chainOffsets.push_back( offsets[i] );
        std::vector<mi355x_bz2_block_result> again( chainOffsets.size() );
        const int rc = mi355x_bz2_decode_batch( ctx, chainOffsets.data(), (uint32_t)chainOffsets.size(), again.data(), &total );
        if ( rc != MI355X_BZ2_OK ) return rc;
        for ( size_t k = 0; k < walk.blocks.size(); ++k ) results[walk.blocks[k]] = again[k];
        dataOffset = 0;
    }

    result->encoded_end_bits = walk.end;
    result->decoded_size = walk.decoded;
    result->data_offset = dataOffset;
    result->n_blocks = (uint32_t)walk.blocks.size();
    result->n_footers = (uint32_t)walk.footers.size();
    result->stopped_preemptively = walk.stoppedPreemptively ? 1 : 0;
    result->status = MI355X_BZ2_OK;
    for ( uint32_t k = 0; k < result->n_blocks && k < blocksCapacity && blocks != nullptr; ++k ) {
        blocks[k] = results[walk.blocks[k]];
        blocks[k].data_offset -= dataOffset;   /* relative to the chunk */
    }
    for ( uint32_t k = 0; k < result->n_footers && k < footersCapacity && footers != nullptr; ++k ) {
        footers[k] = walk.footers[k];
    }
    return MI355X_BZ2_OK;
}
