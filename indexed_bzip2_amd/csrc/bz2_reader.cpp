/**
 * bz2_reader.cpp -- reader + GPU block fetcher behind C ABI section 3 of include/mi355x_bz2.h.
 *
 *   GpuBlockFetcher  <- rapidgzip::BlockFetcher::get / prefetchNewBlocks    src/core/BlockFetcher.hpp:244-317, 446-559
 *                       + BZ2BlockFetcher::{readBlockHeader, decodeBlock}   src/indexed_bzip2/BZ2BlockFetcher.hpp:64-138
 *   ParallelReader   <- indexed_bzip2::ParallelBZ2Reader                    src/indexed_bzip2/ParallelBZ2Reader.hpp:39-498
 *
 * What is kept from the reference: the block finder running ahead on its own thread, the LRU cache + prefetch cache +
 * failed-prefetch cache, the adaptive prefetch strategy, "prefetch failures are silent, on-demand failures surface",
 * the block map and the whole read/seek state machine.
 * What is different by design (GPU backend): the thread pool of per-block tasks is replaced by ONE submission thread
 * that pushes BATCHES of blocks through mi355x_bz2_decode_batch; prefetches are issued in batches of >= P/2 blocks so
 * that every launch has enough independent blocks to fill the GPU.
 */
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <iostream>
#include <list>
#include <memory>
#include <queue>
#include <sstream>
#include <string>

#include <hip/hip_runtime.h>

#include "../../include/mi355x_bz2.h"
#include "bz2_host.hpp"

namespace mi355x
{
struct Bz2Exception : public std::runtime_error
{
    Bz2Exception( int statusCode, const std::string& message ) :
        std::runtime_error( message ),
        status( statusCode )
    {}

    int status;
};

[[noreturn]] static void
fail( int status, const std::string& detail = {} )
{
    std::string message = mi355x_bz2_status_string( status );
    if ( !detail.empty() ) {
        message += ": " + detail;
    }
    throw Bz2Exception( status, message );
}

/* ------------------------------------------------------------------------------------------------ compressed source */
class Source
{
public:
    static std::shared_ptr<Source>
    fromFd( int fd, bool closeFd )
    {
        struct stat st{};
        if ( fstat( fd, &st ) != 0 || !S_ISREG( st.st_mode ) ) {
            if ( closeFd ) ::close( fd );
            /* ParallelBZ2Reader.hpp:66-68 */
            fail( MI355X_BZ2_ERR_INVALID_ARGUMENT, "Parallel BZ2 Reader will not work on non-seekable input like stdin (yet)!" );
        }
        auto s = std::shared_ptr<Source>( new Source() );
        s->m_size = (uint64_t)st.st_size;
        if ( s->m_size > 0 ) {
            void* p = mmap( nullptr, s->m_size, PROT_READ, MAP_PRIVATE, fd, 0 );
            if ( p == MAP_FAILED ) {
                const int e = errno;
                if ( closeFd ) ::close( fd );
                fail( MI355X_BZ2_ERR_IO, std::string( "mmap: " ) + strerror( e ) );
            }
            s->m_map = p;
            s->m_bytes = static_cast<const uint8_t*>( p );
            (void)madvise( p, s->m_size, MADV_SEQUENTIAL );
        }
        if ( closeFd ) ::close( fd );
        return s;
    }

    static std::shared_ptr<Source>
    fromPath( const char* path )
    {
        const int fd = ::open( path, O_RDONLY | O_CLOEXEC );
        if ( fd < 0 ) {
            fail( MI355X_BZ2_ERR_IO, std::string( "open(" ) + path + "): " + strerror( errno ) );
        }
        return fromFd( fd, true );
    }

    static std::shared_ptr<Source>
    fromMemory( const uint8_t* bytes, uint64_t size )
    {
        auto s = std::shared_ptr<Source>( new Source() );
        s->m_copy.assign( bytes, bytes + size );
        s->m_bytes = s->m_copy.data();
        s->m_size = size;
        return s;
    }

    ~Source()
    {
        if ( m_map != nullptr ) {
            munmap( m_map, m_size );
        }
    }

    [[nodiscard]] const uint8_t* bytes() const { return m_bytes; }
    [[nodiscard]] uint64_t size() const { return m_size; }
    [[nodiscard]] uint64_t sizeInBits() const { return m_size * 8; }

private:
    Source() = default;
    const uint8_t* m_bytes{ nullptr };
    uint64_t m_size{ 0 };
    void* m_map{ nullptr };
    std::vector<uint8_t> m_copy;
};

/** MSB-first read of up to 32 bits from a byte array; sets eof if it crosses the end (BitReader.hpp:190-206). */
static uint32_t
readBits( const Source& src, uint64_t& pos, unsigned n, bool& eof )
{
    if ( pos + n > src.sizeInBits() ) {
        eof = true;
        return 0;
    }
    uint64_t v = 0;
    const uint64_t byte = pos >> 3;
    for ( int i = 0; i < 8; ++i ) {
        v = ( v << 8 ) | ( byte + i < src.size() ? src.bytes()[byte + i] : 0 );
    }
    v <<= ( pos & 7 );
    pos += n;
    return n == 0 ? 0 : (uint32_t)( v >> ( 64 - n ) );
}

/* ------------------------------------------------------------------------------------------------ host buffers */
/** Page-locked host buffers for the decoded bytes of a batch, recycled through a small pool: a pageable
 * std::vector costs a memset of the whole batch plus a staged (slow) D2H copy, pinning fresh memory per batch costs
 * more than the copy itself. */
class PinnedPool
{
public:
    struct Buffer
    {
        uint8_t* data{ nullptr };
        size_t capacity{ 0 };
    };

    ~PinnedPool()
    {
        for ( auto& b : m_free ) (void)hipHostFree( b.data );
    }

    [[nodiscard]] std::shared_ptr<const uint8_t>
    get( size_t size, const std::shared_ptr<PinnedPool>& self )
    {
        Buffer buffer;
        {
            const std::scoped_lock lock( m_mutex );
            for ( auto it = m_free.begin(); it != m_free.end(); ++it ) {
                if ( it->capacity >= size ) {
                    buffer = *it;
                    m_free.erase( it );
                    break;
                }
            }
        }
        if ( buffer.data == nullptr ) {
            /* batches of one sequential read differ a little in size: leave headroom so that a recycled buffer fits
             * the next batch instead of pinning fresh memory (which costs more than the copy) */
            const size_t grain = size < ( size_t( 8 ) << 20 ) ? ( size_t( 1 ) << 20 ) : ( size_t( 16 ) << 20 );
            buffer.capacity = ( std::max<size_t>( size + size / 8, 1 ) + grain - 1 ) / grain * grain;
            if ( hipHostMalloc( reinterpret_cast<void**>( &buffer.data ), buffer.capacity, hipHostMallocDefault ) != hipSuccess ) {
                return nullptr;
            }
        }
        /* the deleter hands the memory back; the pool lives as long as any buffer does */
        return std::shared_ptr<const uint8_t>( buffer.data, [self, buffer] ( const uint8_t* ) { self->put( buffer ); } );
    }

private:
    void
    put( Buffer buffer )
    {
        const std::scoped_lock lock( m_mutex );
        if ( m_free.size() < 4 ) {
            m_free.push_back( buffer );
            return;
        }
        /* keep the larger ones */
        auto smallest = std::min_element( m_free.begin(), m_free.end(),
                                          [] ( const Buffer& a, const Buffer& b ) { return a.capacity < b.capacity; } );
        if ( smallest->capacity < buffer.capacity ) std::swap( *smallest, buffer );
        (void)hipHostFree( buffer.data );
    }

    std::mutex m_mutex;
    std::vector<Buffer> m_free;
};

/* ------------------------------------------------------------------------------------------------ block records */
/** indexed_bzip2::BlockHeaderData / BlockData, BZ2BlockFetcher.hpp:18-34 */
struct BlockHeaderData
{
    size_t encodedOffsetInBits{ std::numeric_limits<size_t>::max() };
    size_t encodedSizeInBits{ 0 };
    uint32_t expectedCRC{ 0 };
    bool isEndOfStreamBlock{ false };
    bool isEndOfFile{ false };
};

struct BlockData : public BlockHeaderData
{
    std::shared_ptr<const uint8_t> buffer;   /* page-locked host copy of the whole batch output */
    size_t dataOffset{ 0 };
    size_t dataSize{ 0 };
    size_t batchBytes{ 0 };                  /* decoded bytes that share `buffer` */
    uint32_t calculatedCRC{ 0xFFFFFFFFu };
    int status{ MI355X_BZ2_OK };

    [[nodiscard]] const uint8_t* data() const { return buffer ? buffer.get() + dataOffset : nullptr; }
};

using BlockDataPtr = std::shared_ptr<BlockData>;

/* ------------------------------------------------------------------------------------------------ GPU block fetcher */
class GpuBlockFetcher
{
public:
    GpuBlockFetcher( std::shared_ptr<Source> source, std::shared_ptr<BlockFinder> finder,
                     size_t parallelization, int device ) :
        m_source( std::move( source ) ),
        m_blockFinder( std::move( finder ) ),
        m_parallelization( std::max<size_t>( 1, parallelization ) ),
        m_cache( std::max<size_t>( 16, m_parallelization ) ),              /* BlockFetcher.hpp:180 */
        /* BlockFetcher.hpp:181-182 use 2 P.  The prefetch window (= this capacity) bounds decoded-but-unread blocks plus
         * blocks in flight; with 2 P a second full batch could only start once the reader had consumed all of the first,
         * so launches went out with P / 2 blocks each (80 ms latency floor per launch: 5 GB/s at P = 512).  (contexts + 2) P
         * lets every context have a full batch in flight while another one is being read. */
        m_prefetchCache( ( contextCount( m_parallelization ) + 2 ) * m_parallelization ),
        m_failedPrefetchCache( ( contextCount( m_parallelization ) + 2 ) * m_parallelization )
    {
        /* BZ2BlockFetcher ctor reads the stream header once: BZ2BlockFetcher.hpp:56 */
        if ( mi355x_bz2_read_stream_header( m_source->bytes(), m_source->size(), 0 ) == 0 ) {
            fail( MI355X_BZ2_ERR_STREAM_HEADER );
        }
        /* Several decoder contexts, each with its own submission thread, once batches are large enough to be worth it:
         * while one batch is copied to the host (and consumed), the next ones are already being decoded.  A batch has a
         * latency floor of ~80 ms (one wave per block in the Huffman stage), so smaller batches want more of them in
         * flight: three contexts up to P = 640 (20 GB of scratch at the default P = 512), two above.  Small P keeps two so that
         * an on-demand block does not have to wait for a prefetch launch (see get()); P = 1 is the serial reader. */
        const size_t nContexts = contextCount( m_parallelization );
        const auto tCtor = std::chrono::steady_clock::now();
        for ( size_t i = 0; i < nContexts; ++i ) {
            mi355x_bz2_config config{};
            config.device = device;
            /* scratch (13 MB per block, ~35 ms per GB to allocate) grows with the batches that are really launched: a
             * reader that seeks and reads a little never pays for P blocks, a sequential one pays once per context, on
             * that context's own thread */
            config.max_batch_blocks = (uint32_t)std::min<size_t>( m_parallelization, 64 );
            mi355x_bz2_ctx* ctx = nullptr;
            int rc = mi355x_bz2_create( &config, &ctx );
            std::string detail;
            if ( rc == MI355X_BZ2_OK ) {
                /* one resident copy of the compressed file for all contexts */
                rc = m_ctxs.empty() ? mi355x_bz2_set_input_host( ctx, m_source->bytes(), m_source->size() )
                                    : mi355x_bz2_share_input( ctx, m_ctxs.front() );
                if ( rc != MI355X_BZ2_OK ) {
                    detail = mi355x_bz2_last_error( ctx );
                    mi355x_bz2_destroy( ctx );
                }
            }
            if ( rc != MI355X_BZ2_OK ) {
                for ( auto* const c : m_ctxs ) mi355x_bz2_destroy( c );
                m_ctxs.clear();
                fail( rc, detail );
            }
            m_ctxs.push_back( ctx );
        }
        /* The file is resident on the GPU now: let it find the block magics too (k_find_magic, a few ms per GB).  The
         * host finder threads (ParallelBitStringFinder's job, ~1.3 GB/s of compressed data on eight cores) would
         * otherwise pace the whole reader.  Same offsets, delivered at once; if the scan cannot be used (more matches
         * than its result buffer holds) the host threads take over as before. */
        const auto tInput = std::chrono::steady_clock::now();
        if ( !m_blockFinder->finalized() ) {
            /* the scan keeps up to 2^20 matches (>= 100 GB of level-9 data) and reports MI355X_BZ2_ERR_OUTPUT_CAPACITY beyond */
            std::vector<uint64_t> offsets( (size_t)std::min<uint64_t>( m_source->size() / 6 + 16, 1u << 20 ) );
            uint64_t found = 0;
            if ( ( mi355x_bz2_find_magic_device( m_ctxs.front(), MI355X_BZ2_MAGIC_BLOCK, offsets.data(), offsets.size(),
                                                 &found ) == MI355X_BZ2_OK ) && ( found <= offsets.size() ) ) {
                offsets.resize( found );
                m_blockFinder->setBlockOffsets( std::deque<size_t>( offsets.begin(), offsets.end() ) );
            }
        }
        if ( m_trace ) {
            const auto now = std::chrono::steady_clock::now();
            std::fprintf( stderr, "[reader] %zu contexts + %.0f MB resident: %.1f ms, magic scan: %.1f ms (%zu blocks)\n",
                          m_ctxs.size(), m_source->size() / 1e6,
                          std::chrono::duration<double, std::milli>( tInput - tCtor ).count(),
                          std::chrono::duration<double, std::milli>( now - tInput ).count(), m_blockFinder->size() );
        }
        for ( auto* const ctx : m_ctxs ) {
            m_workers.emplace_back( [this, ctx] () { workerMain( ctx ); } );
        }
    }

    ~GpuBlockFetcher()
    {
        {
            const std::scoped_lock lock( m_queueMutex );
            m_stop = true;
            m_queueChanged.notify_all();
        }
        for ( auto& worker : m_workers ) {
            if ( worker.joinable() ) worker.join();
        }
        m_prefetching.clear();
        for ( auto it = m_ctxs.rbegin(); it != m_ctxs.rend(); ++it ) mi355x_bz2_destroy( *it );   /* owner of the input last */
    }

    [[nodiscard]] static size_t
    contextCount( size_t parallelization )
    {
        if ( const char* const forced = std::getenv( "MI355X_BZ2_READER_CONTEXTS" ) ) {
            return std::min<size_t>( 4, std::max<size_t>( 1, std::strtoul( forced, nullptr, 10 ) ) );
        }
        return parallelization >= 64 ? ( parallelization <= 640 ? 3 : 2 ) : ( parallelization >= 2 ? 2 : 1 );
    }

    /** BZ2BlockFetcher::readBlockHeader, BZ2BlockFetcher.hpp:64-82, for the EOS / next-stream probe on the caller
     * thread.  Host-side parse of magic, CRC, randomised bit and origPtr (bzip2.hpp:479-519); the tree section of a
     * data block is validated when that block is decoded on the GPU. */
    [[nodiscard]] BlockHeaderData
    readBlockHeader( size_t blockOffset ) const
    {
        BlockHeaderData result;
        result.encodedOffsetInBits = blockOffset;
        uint64_t pos = blockOffset;
        bool eof = blockOffset > m_source->sizeInBits();
        const uint64_t hi = readBits( *m_source, pos, 24, eof );
        const uint64_t lo = readBits( *m_source, pos, 24, eof );
        result.expectedCRC = readBits( *m_source, pos, 32, eof );
        if ( eof ) fail( MI355X_BZ2_ERR_EOF );
        const uint64_t magic = ( hi << 24 ) | lo;
        if ( magic == MI355X_BZ2_MAGIC_EOS ) {
            result.isEndOfStreamBlock = true;
            if ( ( pos & 7 ) != 0 ) {
                readBits( *m_source, pos, 8 - (unsigned)( pos & 7 ), eof );
                if ( eof ) fail( MI355X_BZ2_ERR_EOF );
            }
            result.encodedSizeInBits = pos - blockOffset;
            result.isEndOfFile = pos >= m_source->sizeInBits();
            return result;
        }
        if ( magic != MI355X_BZ2_MAGIC_BLOCK ) {
            char buffer[96];
            std::snprintf( buffer, sizeof( buffer ), "0x%llx at bit offset %llu", (unsigned long long)magic,
                           (unsigned long long)blockOffset );
            fail( MI355X_BZ2_ERR_BAD_MAGIC, buffer );
        }
        const uint32_t randomized = readBits( *m_source, pos, 1, eof );
        if ( eof ) fail( MI355X_BZ2_ERR_EOF );
        if ( randomized != 0 ) fail( MI355X_BZ2_ERR_RANDOMIZED );
        const uint32_t origPtr = readBits( *m_source, pos, 24, eof );
        if ( eof ) fail( MI355X_BZ2_ERR_EOF );
        if ( origPtr > 900000 ) fail( MI355X_BZ2_ERR_ORIGPTR_RANGE );
        return result;
    }

    /** BlockFetcher::get, BlockFetcher.hpp:244-317 */
    [[nodiscard]] BlockDataPtr
    get( size_t blockOffset, std::optional<size_t> dataBlockIndex = std::nullopt )
    {
        const auto tStart = std::chrono::steady_clock::now();
        ++m_stats.gets;

        /* getFromCaches, BlockFetcher.hpp:369-391 */
        std::shared_future<BlockDataPtr> queued;
        std::optional<BlockDataPtr> cached;
        if ( const auto match = m_prefetching.find( blockOffset ); match != m_prefetching.end() ) {
            queued = match->second;
            m_prefetching.erase( match );
            ++m_stats.prefetch_hits;
        } else {
            cached = m_cache.get( blockOffset );
            if ( cached ) {
                ++m_stats.cache_hits;
            } else {
                cached = m_prefetchCache.get( blockOffset );
                if ( cached ) {
                    ++m_stats.prefetch_hits;
                    m_prefetchCache.evict( blockOffset );
                    insertIntoCache( blockOffset, *cached );
                }
            }
        }

        const auto validDataBlockIndex = dataBlockIndex ? *dataBlockIndex : m_blockFinder->find( blockOffset );

        std::vector<uint64_t> batch;
        const bool onDemand = !cached.has_value() && !queued.valid();
        if ( onDemand ) {
            ++m_stats.on_demand_fetches;
            batch.push_back( blockOffset );
        }

        m_fetchingStrategy.fetch( validDataBlockIndex );
        collectPrefetches( batch, onDemand, blockOffset );

        if ( onDemand && ( batch.size() > 1 ) && ( m_ctxs.size() >= 2 ) ) {
            /* A batch returns when its slowest block is through (an incompressible block takes 68 ms, a text block
             * 24 ms): the block the caller is waiting for goes alone, ahead of everything queued, and its prefetch
             * companions as a second launch on another context. */
            queued = submitBatch( { batch[0] }, /* urgent */ true )[0];
            const std::vector<uint64_t> companions( batch.begin() + 1, batch.end() );
            const auto futures = submitBatch( companions );
            for ( size_t i = 0; i < companions.size(); ++i ) {
                m_prefetching.emplace( companions[i], futures[i] );
                ++m_stats.prefetches_submitted;
            }
        } else if ( !batch.empty() ) {
            auto futures = submitBatch( batch, onDemand );
            size_t first = 0;
            if ( onDemand ) {
                queued = futures[0];
                first = 1;
            }
            for ( size_t i = first; i < batch.size(); ++i ) {
                m_prefetching.emplace( batch[i], futures[i] );
                ++m_stats.prefetches_submitted;
            }
        }

        if ( cached.has_value() ) {
            return *cached;
        }

        const auto tWait = std::chrono::steady_clock::now();
        auto result = queued.get();
        m_stats.wait_seconds += std::chrono::duration<double>( std::chrono::steady_clock::now() - tWait ).count();
        if ( !result ) {
            fail( MI355X_BZ2_ERR_DEVICE, m_workerError );
        }
        if ( result->status != MI355X_BZ2_OK ) {
            /* on-demand failures surface to the caller (BlockFetcher.hpp:305) */
            fail( result->status, "block at bit offset " + std::to_string( blockOffset ) );
        }
        insertIntoCache( blockOffset, result );
        (void)tStart;
        return result;
    }

    [[nodiscard]] mi355x_bz2_reader_stats
    statistics() const
    {
        auto result = m_stats;
        {
            const std::scoped_lock lock( m_queueMutex );
            result.batches = m_batches;
            result.blocks_decoded = m_blocksDecoded;
            result.decode_seconds = m_decodeSeconds;
        }
        return result;
    }

private:
    struct Request
    {
        std::vector<uint64_t> offsets;
        std::vector<std::promise<BlockDataPtr> > promises;
        std::shared_ptr<std::atomic<bool> > done;   /* set after the last promise: the futures of a batch become ready together */
    };

    struct BatchInFlight
    {
        std::shared_ptr<std::atomic<bool> > done;
        std::vector<uint64_t> offsets;
    };

    void
    insertIntoCache( size_t blockOffset, BlockDataPtr blockData )
    {
        if ( m_fetchingStrategy.isSequential() ) {
            m_cache.clear();   /* BlockFetcher.hpp:343-346 */
        } else if ( blockData && blockData->buffer
                    && ( blockData->batchBytes > 8 * std::max<size_t>( blockData->dataSize, size_t( 1 ) << 20 ) ) ) {
            /* A block shares the page-locked buffer of its whole batch.  This cache survives random access for a long
             * time: one kept block must not hold hundreds of MB of a batch whose other blocks are long gone. */
            auto own = std::make_shared<BlockData>( *blockData );
            std::shared_ptr<uint8_t> bytes( new uint8_t[std::max<size_t>( own->dataSize, 1 )], std::default_delete<uint8_t[]>() );
            std::memcpy( bytes.get(), blockData->data(), own->dataSize );
            own->buffer = std::move( bytes );
            own->dataOffset = 0;
            own->batchBytes = own->dataSize;
            blockData = std::move( own );
        }
        m_cache.insert( blockOffset, std::move( blockData ) );
    }

    [[nodiscard]] bool
    isInCacheOrQueue( size_t blockOffset ) const
    {
        return ( m_prefetching.find( blockOffset ) != m_prefetching.end() )
               || m_cache.test( blockOffset ) || m_prefetchCache.test( blockOffset );
    }

    /** processReadyPrefetches, BlockFetcher.hpp:414-438.  The reference polls every future; here all futures of a batch
     * become ready together, so one flag per batch is polled (a get() with 5 000 blocks in flight must not cost 5 000
     * future look-ups: that alone capped the reader at 5 GB/s). */
    void
    processReadyPrefetches()
    {
        for ( auto batch = m_batchesInFlight.begin(); batch != m_batchesInFlight.end(); ) {
            if ( !batch->done->load( std::memory_order_acquire ) ) {
                ++batch;
                continue;
            }
            for ( const auto offset : batch->offsets ) {
                const auto it = m_prefetching.find( offset );
                if ( it == m_prefetching.end() ) continue;     /* fetched on demand meanwhile */
                const auto result = it->second.get();
                if ( result && ( result->status == MI355X_BZ2_OK ) ) {
                    m_prefetchCache.insert( it->first, result );
                } else {
                    /* Prefetching failed: ignore result and error; it is retried (and reported) on demand. */
                    m_failedPrefetchCache.insert( it->first, true );
                    ++m_stats.failed_prefetches;
                }
                m_prefetching.erase( it );
            }
            batch = m_batchesInFlight.erase( batch );
        }
    }

    /** prefetchNewBlocks, BlockFetcher.hpp:446-559, batch-oriented. */
    void
    collectPrefetches( std::vector<uint64_t>& batch, bool onDemand, size_t requestedOffset )
    {
        processReadyPrefetches();
        /* threadPoolSaturated, BlockFetcher.hpp:453-460: one batch of up to P blocks per decoder context */
        const auto inFlight = m_prefetching.size() + ( onDemand ? 1 : 0 );
        const size_t limit = m_parallelization * m_ctxs.size();
        if ( inFlight >= limit ) return;
        /* The candidate search below looks at up to 2 P indexes.  While a launch is held back for want of candidates
         * (see the batching rule at the end) a sequential reader gains about one candidate per get(): searching again
         * on every call would cost O(P^2) per batch (1 s per 2 560 blocks), so the search is skipped for a while. */
        if ( !onDemand && ( m_skipCollect > 0 ) && !m_prefetching.empty() ) {
            --m_skipCollect;
            return;
        }
        const size_t room = std::min( limit - inFlight, m_parallelization );

        const auto indexes = m_fetchingStrategy.prefetch( m_prefetchCache.capacity() );
        std::vector<uint64_t> candidates;
        for ( const auto index : indexes ) {
            if ( candidates.size() >= room ) break;
            if ( m_blockFinder->finalized() && ( index >= m_blockFinder->size() ) ) continue;
            const auto [offset, code] = m_blockFinder->get( index, /* timeout */ 0 );
            if ( !offset ) continue;
            if ( *offset == requestedOffset || isInCacheOrQueue( *offset ) || m_failedPrefetchCache.test( *offset ) ) {
                m_prefetchCache.touch( *offset );
                m_cache.touch( *offset );
                continue;
            }
            /* Avoid cache pollution: stop when results that are still wanted would be evicted (BlockFetcher.hpp:527-535) */
            if ( m_prefetching.size() + candidates.size() + 1 > m_prefetchCache.capacity() ) break;
            candidates.push_back( *offset );
        }
        if ( candidates.empty() ) {
            m_skipCollect = std::max<size_t>( 1, m_parallelization / 8 );
            return;
        }
        /* GPU batching rule: piggy-back on an on-demand launch, otherwise wait until at least P/2 blocks can go in one
         * launch (or nothing is in flight at all). */
        const size_t batchMin = std::max<size_t>( 1, m_parallelization / 2 );
        if ( onDemand || ( candidates.size() >= batchMin ) || m_prefetching.empty() ) {
            batch.insert( batch.end(), candidates.begin(), candidates.end() );
            m_skipCollect = 0;
        } else {
            m_skipCollect = std::min( batchMin - candidates.size(), std::max<size_t>( 1, m_parallelization / 8 ) );
        }
    }

    [[nodiscard]] std::vector<std::shared_future<BlockDataPtr> >
    submitBatch( const std::vector<uint64_t>& offsets, bool urgent = false )
    {
        auto request = std::make_unique<Request>();
        request->offsets = offsets;
        request->promises.resize( offsets.size() );
        request->done = std::make_shared<std::atomic<bool> >( false );
        m_batchesInFlight.push_back( BatchInFlight{ request->done, offsets } );
        std::vector<std::shared_future<BlockDataPtr> > futures;
        futures.reserve( offsets.size() );
        for ( auto& promise : request->promises ) {
            futures.emplace_back( promise.get_future().share() );
        }
        {
            const std::scoped_lock lock( m_queueMutex );
            if ( urgent ) {
                m_queue.push_front( std::move( request ) );   /* someone is waiting for it */
            } else {
                m_queue.push_back( std::move( request ) );
            }
            m_queueChanged.notify_all();
        }
        return futures;
    }

    /** The single GPU submission thread: replaces ThreadPool workers calling decodeBlock (BlockFetcher.hpp:620-642). */
    void
    workerMain( mi355x_bz2_ctx* const ctx )
    {
        while ( true ) {
            std::unique_ptr<Request> request;
            {
                std::unique_lock lock( m_queueMutex );
                m_queueChanged.wait( lock, [this] { return m_stop || !m_queue.empty(); } );
                if ( m_queue.empty() ) {
                    return;   /* m_stop */
                }
                request = std::move( m_queue.front() );
                m_queue.pop_front();
            }
            const auto t0 = std::chrono::steady_clock::now();
            const auto n = (uint32_t)request->offsets.size();
            std::vector<mi355x_bz2_block_result> results( n );
            uint64_t total = 0;
            int rc = mi355x_bz2_decode_batch( ctx, request->offsets.data(), n, results.data(), &total );
            const auto t1 = std::chrono::steady_clock::now();
            auto t2 = t1;
            std::shared_ptr<const uint8_t> buffer;
            if ( rc == MI355X_BZ2_OK ) {
                buffer = m_hostBuffers->get( total, m_hostBuffers );
                t2 = std::chrono::steady_clock::now();
                if ( !buffer ) {
                    rc = MI355X_BZ2_ERR_DEVICE;
                } else if ( total > 0 ) {
                    rc = mi355x_bz2_copy_output( ctx, 0, total, const_cast<uint8_t*>( buffer.get() ) );
                }
            }
            if ( m_trace ) {
                const auto ms = [] ( auto a, auto b ) { return std::chrono::duration<double, std::milli>( b - a ).count(); };
                const auto t3 = std::chrono::steady_clock::now();
                std::fprintf( stderr, "[reader] t=%.1f ms: batch of %u blocks on ctx %p: decode %.1f ms, host buffer %.1f ms, "
                              "copy of %.0f MB %.1f ms\n", ms( m_created, t0 ), n, (void*)ctx, ms( t0, t1 ), ms( t1, t2 ),
                              total / 1e6, ms( t2, t3 ) );
            }
            if ( rc != MI355X_BZ2_OK ) {
                {
                    const std::scoped_lock lock( m_queueMutex );
                    m_workerError = mi355x_bz2_last_error( ctx );
                }
                for ( auto& promise : request->promises ) {
                    promise.set_value( nullptr );
                }
                request->done->store( true, std::memory_order_release );
                continue;
            }
            for ( uint32_t i = 0; i < n; ++i ) {
                const auto& r = results[i];
                auto block = std::make_shared<BlockData>();
                block->encodedOffsetInBits = request->offsets[i];
                block->encodedSizeInBits = r.encoded_size_bits;
                block->expectedCRC = r.header_crc;
                block->calculatedCRC = r.computed_crc;
                block->isEndOfStreamBlock = r.is_eos != 0;
                block->isEndOfFile = r.is_eof != 0;
                block->status = r.status;
                if ( r.status == MI355X_BZ2_OK ) {
                    block->buffer = buffer;
                    block->dataOffset = r.data_offset;
                    block->dataSize = r.decoded_size;
                    block->batchBytes = total;
                }
                request->promises[i].set_value( std::move( block ) );
            }
            request->done->store( true, std::memory_order_release );
            const std::scoped_lock lock( m_queueMutex );
            ++m_batches;
            m_blocksDecoded += n;
            m_decodeSeconds += std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
        }
    }

private:
    const std::shared_ptr<Source> m_source;
    const std::shared_ptr<BlockFinder> m_blockFinder;
    const size_t m_parallelization;

    FetchNextAdaptive m_fetchingStrategy;
    LruCache<size_t, BlockDataPtr> m_cache;
    LruCache<size_t, BlockDataPtr> m_prefetchCache;
    LruCache<size_t, bool> m_failedPrefetchCache;
    std::map<size_t, std::shared_future<BlockDataPtr> > m_prefetching;
    std::list<BatchInFlight> m_batchesInFlight;
    size_t m_skipCollect{ 0 };

    std::vector<mi355x_bz2_ctx*> m_ctxs;
    const std::shared_ptr<PinnedPool> m_hostBuffers{ std::make_shared<PinnedPool>() };
    std::vector<std::thread> m_workers;
    mutable std::mutex m_queueMutex;
    std::condition_variable m_queueChanged;
    std::deque<std::unique_ptr<Request> > m_queue;
    bool m_stop{ false };
    const bool m_trace{ std::getenv( "MI355X_BZ2_READER_TRACE" ) != nullptr };
    const std::chrono::steady_clock::time_point m_created{ std::chrono::steady_clock::now() };
    std::string m_workerError;
    uint64_t m_batches{ 0 };
    uint64_t m_blocksDecoded{ 0 };
    double m_decodeSeconds{ 0 };

    mi355x_bz2_reader_stats m_stats{};
};

/* ------------------------------------------------------------------------------------------------ the reader */
class ParallelReader
{
public:
    ParallelReader( std::shared_ptr<Source> source, size_t parallelization, int device ) :
        m_source( std::move( source ) ),
        m_parallelization( parallelization == 0 ? DEFAULT_PARALLELIZATION : parallelization ),
        m_device( device ),
        m_verifyStreamCrc( parallelization == 1 )   /* the reference's serial reader checks, the parallel one does not */
    {}

    void setVerifyStreamCrc( bool enable ) { m_verifyStreamCrc = enable; }
    [[nodiscard]] uint64_t streamsVerified() const { return m_streamsVerified; }

    /* parallelization 0: a batch of this many blocks keeps the GPU busy (6.5 GB/s at 640, 0.65 GB/s at 64) while a
     * single cold read still returns after one batch of about 80 ms */
    static constexpr size_t DEFAULT_PARALLELIZATION = 512;

    void
    close()   /* ParallelBZ2Reader.hpp:104-111 */
    {
        m_blockFetcher.reset();
        m_blockFinder.reset();
        m_source.reset();
    }

    [[nodiscard]] bool closed() const { return !m_source; }
    [[nodiscard]] bool eof() const { return m_atEndOfFile; }

    [[nodiscard]] size_t
    tell() const   /* :129-142 */
    {
        if ( m_atEndOfFile ) {
            const auto fileSize = size();
            if ( !fileSize ) {
                fail( MI355X_BZ2_ERR_LOGIC, "When the file end has been reached, the block map should have been "
                                            "finalized and the file size should be available!" );
            }
            return *fileSize;
        }
        return m_currentPosition;
    }

    [[nodiscard]] std::optional<size_t>
    size() const   /* :144-151 */
    {
        if ( !m_blockMap.finalized() ) {
            return std::nullopt;
        }
        return m_blockMap.back().second;
    }

    using WriteFunctor = std::function<void( const void*, uint64_t )>;

    /** ParallelBZ2Reader::read, ParallelBZ2Reader.hpp:167-269 */
    size_t
    read( const WriteFunctor& writeFunctor, size_t nBytesToRead = std::numeric_limits<size_t>::max() )
    {
        if ( closed() ) {
            fail( MI355X_BZ2_ERR_CLOSED, "You may not call read on closed ParallelBZ2Reader!" );
        }
        if ( eof() || ( nBytesToRead == 0 ) ) {
            return 0;
        }
        size_t nBytesDecoded = 0;
        while ( ( nBytesDecoded < nBytesToRead ) && !eof() ) {
            BlockDataPtr blockData;
            auto blockInfo = m_blockMap.findDataOffset( m_currentPosition );
            if ( !blockInfo.contains( m_currentPosition ) ) {
                /* Fetch new block for the first time and add information to block map. */
                const auto dataBlockIndex = m_blockMap.dataBlockCount();
                const auto encodedOffsetInBits = blockFinder().get( dataBlockIndex ).first;
                if ( !encodedOffsetInBits ) {
                    m_blockMap.finalize();
                    m_atEndOfFile = true;
                    break;
                }
                blockData = blockFetcher().get( *encodedOffsetInBits, dataBlockIndex );
                m_blockMap.push( blockData->encodedOffsetInBits, blockData->encodedSizeInBits, blockData->dataSize );
                /* BZ2Reader.hpp:481-484: new blocks arrive here in file order */
                m_calculatedStreamCrc = ( ( m_calculatedStreamCrc << 1U ) | ( m_calculatedStreamCrc >> 31U ) )
                                        ^ blockData->calculatedCRC;

                /* EOS blocks have a different magic and are not found by the block finder (:204-238) */
                if ( !blockData->isEndOfFile ) {
                    const auto next = blockFetcher().readBlockHeader( blockData->encodedOffsetInBits
                                                                      + blockData->encodedSizeInBits );
                    if ( next.isEndOfStreamBlock ) {
                        m_blockMap.push( next.encodedOffsetInBits, next.encodedSizeInBits, 0 );
                        /* the end-of-stream block carries the CRC of the whole stream (BZ2Reader.hpp:406-416) */
                        const auto calculated = m_calculatedStreamCrc;
                        m_calculatedStreamCrc = 0;
                        if ( m_verifyStreamCrc && m_streamCrcIntact ) {
                            if ( next.expectedCRC != calculated ) {
                                std::stringstream msg;
                                msg << "[BZip2 block header] Stream CRC 0x" << std::hex << next.expectedCRC
                                    << " does not match calculated CRC 0x" << calculated;
                                fail( MI355X_BZ2_ERR_STREAM_CRC, msg.str() );
                            }
                            ++m_streamsVerified;
                        }
                        m_streamCrcIntact = true;
                        const auto nextStreamOffsetInBits = next.encodedOffsetInBits + next.encodedSizeInBits;
                        if ( nextStreamOffsetInBits < m_source->sizeInBits() ) {
                            if ( mi355x_bz2_read_stream_header( m_source->bytes(), m_source->size(),
                                                                nextStreamOffsetInBits ) == 0 ) {
                                std::cerr << "[Warning] Trailing garbage after EOF ignored!\n";
                                m_blockFinder->finalize( m_blockMap.dataBlockCount() );
                            }
                        }
                    }
                }
                blockInfo = m_blockMap.findDataOffset( m_currentPosition );
                if ( !blockInfo.contains( m_currentPosition ) ) {
                    continue;
                }
            } else {
                blockData = blockFetcher().get( blockInfo.encodedOffsetInBits );
            }

            const auto offsetInBlock = m_currentPosition - blockInfo.decodedOffsetInBytes;
            if ( offsetInBlock >= blockData->dataSize ) {
                fail( MI355X_BZ2_ERR_LOGIC, "Block does not contain the requested offset even though it "
                                            "shouldn't be according to block map!" );
            }
            const auto nBytesToDecode = std::min( blockData->dataSize - offsetInBlock, nBytesToRead - nBytesDecoded );
            if ( writeFunctor ) {
                writeFunctor( blockData->data() + offsetInBlock, nBytesToDecode );
            }
            nBytesDecoded += nBytesToDecode;
            m_currentPosition += nBytesToDecode;
        }
        return nBytesDecoded;
    }

    /** BZ2ReaderInterface::read( fd, buffer, n ), BZ2ReaderInterface.hpp:35-57 + writeAll, FileUtils.hpp:803-826 */
    size_t
    read( int outputFileDescriptor, char* outputBuffer, size_t nBytesToRead )
    {
        if ( ( outputFileDescriptor < 0 ) && ( outputBuffer == nullptr ) ) {
            return read( WriteFunctor(), nBytesToRead );
        }
        uint64_t nBytesDecoded = 0;
        const WriteFunctor writeFunctor = [&] ( const void* buffer, uint64_t size ) {
            if ( outputFileDescriptor >= 0 ) {
                const auto* p = static_cast<const uint8_t*>( buffer );
                uint64_t written = 0;
                while ( written < size ) {
                    const auto n = ::write( outputFileDescriptor, p + written,
                                            (size_t)std::min<uint64_t>( size - written, 1u << 30 ) );
                    if ( n <= 0 ) {
                        if ( n < 0 && errno == EINTR ) continue;
                        fail( MI355X_BZ2_ERR_IO, std::string( "Failed to write all bytes because of: " ) + strerror( errno ) );
                    }
                    written += (uint64_t)n;
                }
            }
            if ( outputBuffer != nullptr ) {
                std::memcpy( outputBuffer + nBytesDecoded, buffer, size );
            }
            nBytesDecoded += size;
        };
        return read( writeFunctor, nBytesToRead );
    }

    /** ParallelBZ2Reader::seek, ParallelBZ2Reader.hpp:271-325 */
    size_t
    seek( long long int offset, int origin )
    {
        if ( closed() ) {
            fail( MI355X_BZ2_ERR_CLOSED, "You may not call seek on closed ParallelBZ2Reader!" );
        }
        if ( origin == SEEK_END ) {
            if ( !m_blockMap.finalized() ) {
                read( WriteFunctor() );
            }
        }
        const auto positiveOffset = effectiveOffset( offset, origin );
        if ( positiveOffset == tell() ) {
            return positiveOffset;
        }
        if ( positiveOffset < tell() ) {
            m_atEndOfFile = false;
            m_currentPosition = positiveOffset;
            return positiveOffset;
        }
        const auto blockInfo = m_blockMap.findDataOffset( positiveOffset );
        if ( positiveOffset < blockInfo.decodedOffsetInBytes ) {
            fail( MI355X_BZ2_ERR_LOGIC, "Block map returned unwanted block!" );
        }
        if ( blockInfo.contains( positiveOffset ) ) {
            m_atEndOfFile = false;
            m_currentPosition = positiveOffset;
            return tell();
        }
        if ( m_blockMap.finalized() ) {
            m_atEndOfFile = true;
            m_currentPosition = m_blockMap.back().second;
            return tell();
        }
        m_atEndOfFile = false;
        m_currentPosition = blockInfo.decodedOffsetInBytes + blockInfo.decodedSizeInBytes;
        read( WriteFunctor(), positiveOffset - tell() );
        return tell();
    }

    [[nodiscard]] bool
    blockOffsetsComplete() const
    {
        return m_blockMap.finalized();
    }

    /** :339-350 */
    [[nodiscard]] std::map<size_t, size_t>
    blockOffsets()
    {
        if ( !m_blockMap.finalized() ) {
            read( WriteFunctor() );
            if ( !m_blockMap.finalized() || !blockFinder().finalized() ) {
                fail( MI355X_BZ2_ERR_LOGIC, "Reading everything should have finalized the block map!" );
            }
        }
        return m_blockMap.blockOffsets();
    }

    [[nodiscard]] std::map<size_t, size_t>
    availableBlockOffsets() const
    {
        return m_blockMap.blockOffsets();
    }

    /** :365-378 */
    void
    setBlockOffsets( const std::map<size_t, size_t>& offsets )
    {
        if ( offsets.empty() ) {
            fail( MI355X_BZ2_ERR_INVALID_ARGUMENT, "May not clear offsets. Construct a new ParallelBZ2Reader instead!" );
        }
        setBlockFinderOffsets( offsets );
        if ( offsets.size() < 2 ) {
            fail( MI355X_BZ2_ERR_INVALID_ARGUMENT,
                  "Block offset map must contain at least one valid block and one EOS block!" );
        }
        m_blockMap.setBlockOffsets( offsets );
    }

    /** :385-393 */
    [[nodiscard]] size_t
    tellCompressed() const
    {
        const auto blockInfo = m_blockMap.findDataOffset( m_currentPosition );
        if ( blockInfo.contains( m_currentPosition ) ) {
            return blockInfo.encodedOffsetInBits;
        }
        return m_blockMap.empty() ? 0 : m_blockMap.back().first;
    }

    /** :404-409 */
    void
    joinThreads()
    {
        m_blockFetcher.reset();
        m_blockFinder.reset();
    }

    [[nodiscard]] mi355x_bz2_reader_stats
    statistics() const
    {
        return m_blockFetcher ? m_blockFetcher->statistics() : mi355x_bz2_reader_stats{};
    }

private:
    [[nodiscard]] size_t
    effectiveOffset( long long int offset, int origin ) const
    {
        /* FileReader::effectiveOffset, src/core/filereader/FileReader.hpp:110-140 */
        long long int base = 0;
        switch ( origin ) {
        case SEEK_SET: base = 0; break;
        case SEEK_CUR: base = (long long int)tell(); break;
        case SEEK_END:
        {
            const auto fileSize = size();
            if ( !fileSize ) fail( MI355X_BZ2_ERR_LOGIC, "File size is not available to seek from end!" );
            base = (long long int)*fileSize;
            break;
        }
        default: fail( MI355X_BZ2_ERR_INVALID_ARGUMENT, "Invalid seek origin supplied" );
        }
        auto target = base + offset;
        if ( target < 0 ) target = 0;
        const auto fileSize = size();
        if ( fileSize && ( (size_t)target > *fileSize ) ) {
            target = (long long int)*fileSize;
        }
        return (size_t)target;
    }

    /** :412-433 */
    BlockFinder&
    blockFinder()
    {
        if ( m_blockFinder ) {
            return *m_blockFinder;
        }
        const unsigned cores = std::max( 1u, std::thread::hardware_concurrency() );
        /* look-ahead 3 * hardware_concurrency in the reference (BlockFinder.hpp:213); a GPU batch wants more */
        const size_t lookAhead = std::max<size_t>( 3 * cores, 4 * m_parallelization );
        m_blockFinder = std::make_shared<BlockFinder>( m_source->bytes(), m_source->size(), MI355X_BZ2_MAGIC_BLOCK,
                                                       lookAhead, std::min( 8u, std::max( 1u, cores / 2 ) ) );
        if ( m_blockMap.finalized() ) {
            setBlockFinderOffsets( m_blockMap.blockOffsets() );
        }
        return *m_blockFinder;
    }

    /** :435-454 */
    GpuBlockFetcher&
    blockFetcher()
    {
        if ( m_blockFetcher ) {
            return *m_blockFetcher;
        }
        (void)blockFinder();
        m_blockFetcher = std::make_unique<GpuBlockFetcher>( m_source, m_blockFinder, m_parallelization, m_device );
        if ( !blockFinder().finalized() ) {
            blockFinder().startThreads();
        }
        return *m_blockFetcher;
    }

    /** :456-475 */
    void
    setBlockFinderOffsets( const std::map<size_t, size_t>& offsets )
    {
        if ( offsets.empty() ) {
            fail( MI355X_BZ2_ERR_INVALID_ARGUMENT, "A non-empty list of block offsets is required!" );
        }
        std::deque<size_t> encodedBlockOffsets;
        for ( auto it = offsets.begin(), nit = std::next( offsets.begin() ); nit != offsets.end(); ++it, ++nit ) {
            if ( it->second != nit->second ) {
                encodedBlockOffsets.push_back( it->first );
            }
        }
        blockFinder().setBlockOffsets( std::move( encodedBlockOffsets ) );
    }

private:
    std::shared_ptr<Source> m_source;
    const size_t m_parallelization;
    const int m_device;
    size_t m_currentPosition{ 0 };
    bool m_atEndOfFile{ false };

    bool m_verifyStreamCrc;
    bool m_streamCrcIntact{ true };         /* every block of the current stream went into m_calculatedStreamCrc */
    uint32_t m_calculatedStreamCrc{ 0 };
    uint64_t m_streamsVerified{ 0 };

    std::shared_ptr<BlockFinder> m_blockFinder;
    BlockMap m_blockMap;
    std::unique_ptr<GpuBlockFetcher> m_blockFetcher;
};
}  // namespace mi355x

/* ================================================================================================ C ABI */
struct mi355x_bz2_reader
{
    std::unique_ptr<mi355x::ParallelReader> reader;
    std::string lastError;
};

namespace
{
template<typename Functor>
int
guarded( mi355x_bz2_reader* r, const Functor& functor )
{
    if ( r == nullptr || !r->reader ) {
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    }
    try {
        functor( *r->reader );
        return MI355X_BZ2_OK;
    } catch ( const mi355x::Bz2Exception& e ) {
        r->lastError = e.what();
        return e.status;
    } catch ( const std::invalid_argument& e ) {
        r->lastError = e.what();
        return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    } catch ( const std::exception& e ) {
        r->lastError = e.what();
        return MI355X_BZ2_ERR_LOGIC;
    }
}

template<typename MakeSource>
int
openReader( const MakeSource& makeSource, uint32_t parallelization, int32_t device, mi355x_bz2_reader** out )
{
    if ( out == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    try {
        auto source = makeSource();
        auto* r = new mi355x_bz2_reader();
        r->reader = std::make_unique<mi355x::ParallelReader>( std::move( source ), parallelization, device );
        *out = r;
        return MI355X_BZ2_OK;
    } catch ( const mi355x::Bz2Exception& e ) {
        std::fprintf( stderr, "mi355x_bz2_reader_open: %s\n", e.what() );
        return e.status;
    } catch ( const std::exception& e ) {
        std::fprintf( stderr, "mi355x_bz2_reader_open: %s\n", e.what() );
        return MI355X_BZ2_ERR_LOGIC;
    }
}

int
copyOffsets( const std::map<size_t, size_t>& offsets, uint64_t* bits, uint64_t* bytes, uint64_t capacity, uint64_t* n )
{
    if ( n != nullptr ) *n = offsets.size();
    if ( capacity == 0 ) return MI355X_BZ2_OK;
    if ( bits == nullptr || bytes == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    uint64_t i = 0;
    for ( const auto& [b, d] : offsets ) {
        if ( i >= capacity ) break;
        bits[i] = b;
        bytes[i] = d;
        ++i;
    }
    return MI355X_BZ2_OK;
}
}  // namespace

extern "C" {

int
mi355x_bz2_reader_open_path( const char* path, uint32_t parallelization, int32_t device, mi355x_bz2_reader** r )
{
    if ( path == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    return openReader( [path] () { return mi355x::Source::fromPath( path ); }, parallelization, device, r );
}

int
mi355x_bz2_reader_open_fd( int fd, uint32_t parallelization, int32_t device, mi355x_bz2_reader** r )
{
    /* dup so that the caller's descriptor stays usable (StandardFileReader(int), filereader/Standard.hpp:50-62) */
    return openReader( [fd] () {
        const int copy = dup( fd );
        if ( copy < 0 ) mi355x::fail( MI355X_BZ2_ERR_IO, std::string( "dup: " ) + strerror( errno ) );
        return mi355x::Source::fromFd( copy, true );
    }, parallelization, device, r );
}

int
mi355x_bz2_reader_open_memory( const uint8_t* bytes, uint64_t size, uint32_t parallelization, int32_t device,
                               mi355x_bz2_reader** r )
{
    if ( bytes == nullptr && size > 0 ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    return openReader( [bytes, size] () { return mi355x::Source::fromMemory( bytes, size ); }, parallelization, device, r );
}

void
mi355x_bz2_reader_close( mi355x_bz2_reader* r )
{
    if ( r == nullptr ) return;
    if ( r->reader ) {
        try { r->reader->close(); } catch ( ... ) {}
    }
    delete r;
}

const char*
mi355x_bz2_reader_last_error( const mi355x_bz2_reader* r )
{
    return r != nullptr ? r->lastError.c_str() : "null reader";
}

int
mi355x_bz2_reader_read( mi355x_bz2_reader* r, int fd, void* buffer, uint64_t nBytes, uint64_t* nRead )
{
    if ( nRead != nullptr ) *nRead = 0;
    return guarded( r, [&] ( mi355x::ParallelReader& reader ) {
        const auto n = reader.read( fd, static_cast<char*>( buffer ), (size_t)nBytes );
        if ( nRead != nullptr ) *nRead = n;
    } );
}

int
mi355x_bz2_reader_seek( mi355x_bz2_reader* r, int64_t offset, int whence, uint64_t* newPosition )
{
    return guarded( r, [&] ( mi355x::ParallelReader& reader ) {
        const auto p = reader.seek( offset, whence );
        if ( newPosition != nullptr ) *newPosition = p;
    } );
}

uint64_t
mi355x_bz2_reader_tell( const mi355x_bz2_reader* r )
{
    uint64_t result = 0;
    guarded( const_cast<mi355x_bz2_reader*>( r ), [&] ( mi355x::ParallelReader& reader ) { result = reader.tell(); } );
    return result;
}

int
mi355x_bz2_reader_eof( const mi355x_bz2_reader* r )
{
    return ( r != nullptr && r->reader && r->reader->eof() ) ? 1 : 0;
}

int
mi355x_bz2_reader_closed( const mi355x_bz2_reader* r )
{
    return ( r == nullptr || !r->reader || r->reader->closed() ) ? 1 : 0;
}

int
mi355x_bz2_reader_size( const mi355x_bz2_reader* r, uint64_t* size )
{
    if ( r == nullptr || !r->reader ) return 0;
    const auto s = r->reader->size();
    if ( !s ) return 0;
    if ( size != nullptr ) *size = *s;
    return 1;
}

uint64_t
mi355x_bz2_reader_tell_compressed( const mi355x_bz2_reader* r )
{
    uint64_t result = 0;
    guarded( const_cast<mi355x_bz2_reader*>( r ),
             [&] ( mi355x::ParallelReader& reader ) { result = reader.tellCompressed(); } );
    return result;
}

int
mi355x_bz2_reader_block_offsets_complete( const mi355x_bz2_reader* r )
{
    return ( r != nullptr && r->reader && r->reader->blockOffsetsComplete() ) ? 1 : 0;
}

int
mi355x_bz2_reader_block_offsets( mi355x_bz2_reader* r, uint64_t* bits, uint64_t* bytes, uint64_t capacity, uint64_t* n )
{
    int inner = MI355X_BZ2_OK;
    const int rc = guarded( r, [&] ( mi355x::ParallelReader& reader ) {
        inner = copyOffsets( reader.blockOffsets(), bits, bytes, capacity, n );
    } );
    return rc != MI355X_BZ2_OK ? rc : inner;
}

int
mi355x_bz2_reader_available_block_offsets( const mi355x_bz2_reader* r, uint64_t* bits, uint64_t* bytes,
                                           uint64_t capacity, uint64_t* n )
{
    int inner = MI355X_BZ2_OK;
    const int rc = guarded( const_cast<mi355x_bz2_reader*>( r ), [&] ( mi355x::ParallelReader& reader ) {
        inner = copyOffsets( reader.availableBlockOffsets(), bits, bytes, capacity, n );
    } );
    return rc != MI355X_BZ2_OK ? rc : inner;
}

int
mi355x_bz2_reader_set_block_offsets( mi355x_bz2_reader* r, const uint64_t* bits, const uint64_t* bytes, uint64_t n )
{
    if ( n > 0 && ( bits == nullptr || bytes == nullptr ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    return guarded( r, [&] ( mi355x::ParallelReader& reader ) {
        std::map<size_t, size_t> offsets;
        for ( uint64_t i = 0; i < n; ++i ) {
            offsets.emplace( bits[i], bytes[i] );
        }
        reader.setBlockOffsets( offsets );
    } );
}

int
mi355x_bz2_reader_join_threads( mi355x_bz2_reader* r )
{
    return guarded( r, [] ( mi355x::ParallelReader& reader ) { reader.joinThreads(); } );
}

int
mi355x_bz2_reader_set_verify_stream_crc( mi355x_bz2_reader* r, int enable )
{
    return guarded( r, [enable] ( mi355x::ParallelReader& reader ) { reader.setVerifyStreamCrc( enable != 0 ); } );
}

uint64_t
mi355x_bz2_reader_streams_verified( const mi355x_bz2_reader* r )
{
    return ( r != nullptr && r->reader ) ? r->reader->streamsVerified() : 0;
}

int
mi355x_bz2_reader_statistics( const mi355x_bz2_reader* r, mi355x_bz2_reader_stats* stats )
{
    if ( stats == nullptr ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    return guarded( const_cast<mi355x_bz2_reader*>( r ),
                    [&] ( mi355x::ParallelReader& reader ) { *stats = reader.statistics(); } );
}

}  // extern "C"
