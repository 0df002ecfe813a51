/**
 * bz2_reader.cpp -- reader + GPU batch scheduler behind C ABI section 3 of include/mi355x_bz2.h.
 *
 *   BatchScheduler  the role of rapidgzip::BlockFetcher + BZ2BlockFetcher    src/core/BlockFetcher.hpp:244-317, 446-559,
 *                                                                            src/indexed_bzip2/BZ2BlockFetcher.hpp:64-138
 *   StreamReader    the role of indexed_bzip2::ParallelBZ2Reader             src/indexed_bzip2/ParallelBZ2Reader.hpp:39-498
 *
 * What a caller can observe is the reference's: read / seek / tell / eof, the block-offset map with its end-of-stream
 * entries and end-of-file entry, "a block that cannot be decoded fails the read that needs it and no read before it",
 * trailing garbage is ignored with a warning, the stream CRC check of the serial reader.  The structure is this design's:
 * the unit of work is a RUN -- a batch of consecutive blocks decoded by one launch sequence into one page-locked host
 * buffer -- not a block.  The scheduler keeps runs (in flight and finished) by the number of their first block, decides
 * from the recent access pattern which RANGE of blocks to launch next, and hands out runs; the reader walks through a
 * run's blocks without further look-ups in caches, and bytes of consecutive blocks are consecutive in the run's buffer.
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
#include <optional>
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

    /** A page-locked buffer of at least `size` bytes: the smallest free one that fits, else fresh memory (which costs
     * more than the copy it is for: 20 to 50 ms for a batch of 512 blocks).  `capacity`, if given, receives its size. */
    [[nodiscard]] std::shared_ptr<const uint8_t>
    get( size_t size, const std::shared_ptr<PinnedPool>& self, size_t* capacity = nullptr )
    {
        Buffer buffer;
        {
            const std::scoped_lock lock( m_mutex );
            auto best = m_free.end();
            for ( auto it = m_free.begin(); it != m_free.end(); ++it ) {
                if ( ( it->capacity >= size ) && ( best == m_free.end() || it->capacity < best->capacity ) ) best = it;
            }
            if ( best != m_free.end() ) {
                buffer = *best;
                m_free.erase( best );
            }
        }
        if ( buffer.data == nullptr ) {
            /* batches of one sequential read differ a little in size: leave headroom so that a recycled buffer fits
             * the next batch instead of pinning fresh memory */
            const size_t grain = size < ( size_t( 8 ) << 20 ) ? ( size_t( 1 ) << 20 ) : ( size_t( 16 ) << 20 );
            buffer.capacity = ( std::max<size_t>( size + size / 8, 1 ) + grain - 1 ) / grain * grain;
            if ( hipHostMalloc( reinterpret_cast<void**>( &buffer.data ), buffer.capacity, hipHostMallocDefault ) != hipSuccess ) {
                return nullptr;
            }
        }
        if ( capacity != nullptr ) *capacity = buffer.capacity;
        /* the deleter hands the memory back; the pool lives as long as any buffer does */
        return std::shared_ptr<const uint8_t>( buffer.data, [self, buffer] ( const uint8_t* ) { self->put( buffer ); } );
    }

private:
    void
    put( Buffer buffer )
    {
        const std::scoped_lock lock( m_mutex );
        if ( m_free.size() < 8 ) {
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

/* ------------------------------------------------------------------------------------------------ decoded runs */
/** One block of a run: indexed_bzip2::BlockHeaderData / BlockData (BZ2BlockFetcher.hpp:18-34) without owning bytes. */
struct BlockRecord
{
    uint64_t bits{ 0 }, bitLength{ 0 };
    uint64_t at{ 0 }, byteLength{ 0 };       /* the block's bytes inside the run's buffer */
    uint32_t storedCrc{ 0 }, computedCrc{ 0xFFFFFFFFu };
    int status{ MI355X_BZ2_OK };
    bool endOfStream{ false }, endOfFile{ false };
};

/** Consecutive blocks [firstBlock, firstBlock + blocks.size()) of the finder's list, decoded by one batch. */
struct DecodedRun
{
    size_t firstBlock{ 0 };
    std::vector<BlockRecord> blocks;
    std::shared_ptr<const uint8_t> bytes;    /* page-locked; block k at bytes + blocks[k].at */
    size_t totalBytes{ 0 };
    bool lookAhead{ false };                 /* launched ahead of the reader (not because a read was waiting for it) */

    [[nodiscard]] size_t first() const { return firstBlock; }
    [[nodiscard]] size_t count() const { return blocks.size(); }
    [[nodiscard]] const BlockRecord& at( size_t block ) const { return blocks[block - firstBlock]; }
};

using RunPtr = std::shared_ptr<const DecodedRun>;

/** Header fields of the block at a bit offset, parsed on the host (the GPU validates the rest when it decodes it). */
struct BlockHeader
{
    uint64_t bits{ 0 }, bitLength{ 0 };
    uint32_t storedCrc{ 0 };
    bool endOfStream{ false }, endOfFile{ false };
};

/* ------------------------------------------------------------------------------------------------ batch scheduler */
class BatchScheduler
{
public:
    BatchScheduler( std::shared_ptr<Source> source, std::shared_ptr<BlockFinder> finder, size_t parallelization, int device ) :
        m_source( std::move( source ) ),
        m_finder( std::move( finder ) ),
        m_batch( std::max<size_t>( 1, parallelization ) ),
        m_contexts( contextCount( m_batch ) ),
        /* blocks that may be decoded ahead of the reader, finished or in flight: every context a full batch in flight and
         * one more queued (a context that has finished finds its next launch without waiting for the reading thread),
         * while another two are finished and being read */
        m_window( ( m_contexts + 3 ) * m_batch ),
        m_ready( m_window + std::max<size_t>( 16, m_batch ) )
    {
        /* BZ2BlockFetcher's constructor reads the stream header once: BZ2BlockFetcher.hpp:56 */
        m_level = mi355x_bz2_read_stream_header( m_source->bytes(), m_source->size(), 0 );
        if ( m_level == 0 ) {
            fail( MI355X_BZ2_ERR_STREAM_HEADER );
        }
        m_resident = fitsDevice( device );
        /* Several decoder contexts, each with its own submission thread: while one batch is copied to the host (and
         * consumed), the next ones are being decoded.  They share ONE resident copy of the compressed file.  Only the first
         * context is created here -- the first block goes to it at once; the others are created by their threads, beside
         * the first launches (a context costs tens of milliseconds: streams, events, scratch). */
        const auto tCreate = std::chrono::steady_clock::now();
        m_ctxs.assign( m_contexts, nullptr );
        {
            std::string detail;
            const int rc = createContext( device, nullptr, m_ctxs[0], detail );
            if ( rc != MI355X_BZ2_OK ) fail( rc, detail );
        }
        if ( m_trace ) {
            std::fprintf( stderr, "[reader] first context, copy of %.0f MB started: %.1f ms\n", m_source->size() / 1e6,
                          std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - tCreate ).count() );
        }
        if ( m_resident ) {
            m_scanner = std::thread( [this] { scanOnDevice(); } );
        }
        for ( size_t i = 0; i < m_contexts; ++i ) {
            m_workers.emplace_back( [this, i, device] () {
                if ( i > 0 ) {
                    std::string detail;
                    mi355x_bz2_ctx* ctx = nullptr;
                    if ( createContext( device, m_ctxs[0], ctx, detail ) != MI355X_BZ2_OK ) {
                        const std::scoped_lock lock( m_queueMutex );
                        m_workerError = detail;
                        return;   /* the others carry on */
                    }
                    m_ctxs[i] = ctx;
                }
                workerMain( m_ctxs[i] );
            } );
        }
    }

    /** A decoder context over the file: the first one starts the copy to the GPU (in the background, in pieces: the first
     * blocks decode while the rest of a large file is still on its way), the others share it. */
    int
    createContext( int device, mi355x_bz2_ctx* shareFrom, mi355x_bz2_ctx*& out, std::string& detail ) const
    {
        mi355x_bz2_config config{};
        config.device = device;
        /* scratch (13 MB per block) grows with the batches that are really launched: a reader that seeks and reads a
         * little never pays for a full batch, a sequential one pays once per context, on that context's own thread */
        config.max_batch_blocks = 1;
        mi355x_bz2_ctx* ctx = nullptr;
        int rc = mi355x_bz2_create( &config, &ctx );
        if ( rc != MI355X_BZ2_OK ) return rc;
        if ( m_resident ) {
            rc = shareFrom == nullptr ? mi355x_bz2_set_input_host_streamed( ctx, m_source->bytes(), m_source->size() )
                                      : mi355x_bz2_share_input( ctx, shareFrom );
        }   /* else: every launch brings the bytes of its own blocks, see workerMain */
        if ( rc != MI355X_BZ2_OK ) {
            detail = mi355x_bz2_last_error( ctx );
            mi355x_bz2_destroy( ctx );
            return rc;
        }
        out = ctx;
        return MI355X_BZ2_OK;
    }

    /**
     * Whether the whole compressed file may be kept resident on the GPU next to the decoders' scratch (the fast way: one
     * copy, the magic scan on the device).  If not -- a file beyond the free device memory, or beyond
     * MI355X_BZ2_INPUT_BUDGET bytes -- residency is BOUNDED: every launch copies the byte range of its own blocks into its
     * context's input buffer (two per context: the next range travels beside the batch in flight), the way the reference
     * streams a file through 128 KiB refills of its bit reader (src/core/BitReader.hpp:57, filereader/Shared.hpp:238-335);
     * block offsets then come from the host finder threads alone.
     */
    [[nodiscard]] bool
    fitsDevice( int device ) const
    {
        if ( const char* const budget = std::getenv( "MI355X_BZ2_INPUT_BUDGET" ) ) {
            return m_source->size() <= std::strtoull( budget, nullptr, 10 );
        }
        size_t freeBytes = 0, totalBytes = 0;
        if ( ( device >= 0 && hipSetDevice( device ) != hipSuccess ) || hipMemGetInfo( &freeBytes, &totalBytes ) != hipSuccess ) {
            return true;    /* no device: creating the first context reports that */
        }
        /* scratch and output of every context's batches, and some room for whoever else uses the device */
        const uint64_t others = (uint64_t)m_contexts * m_batch * ( uint64_t( 15 ) << 20 ) + ( uint64_t( 2 ) << 30 );
        return m_source->size() + others <= (uint64_t)freeBytes;
    }

    [[nodiscard]] bool inputResident() const { return m_resident; }

    ~BatchScheduler()
    {
        {
            const std::scoped_lock lock( m_queueMutex );
            m_stop = true;
            m_queueChanged.notify_all();
        }
        if ( m_scanner.joinable() ) m_scanner.join();
        for ( auto& worker : m_workers ) {
            if ( worker.joinable() ) worker.join();
        }
        m_flights.clear();
        for ( auto it = m_ctxs.rbegin(); it != m_ctxs.rend(); ++it ) {
            if ( *it != nullptr ) mi355x_bz2_destroy( *it );   /* owner of the input last */
        }
    }

    /** Decoder contexts for a batch size: a batch has a latency floor (its largest block), so smaller batches want more
     * of them in flight: three contexts up to 640 blocks per batch, two above; small batches keep two so that a block
     * someone waits for does not queue behind a look-ahead launch; one block at a time is the serial reader. */
    [[nodiscard]] static size_t
    contextCount( size_t batch )
    {
        if ( const char* const forced = std::getenv( "MI355X_BZ2_READER_CONTEXTS" ) ) {
            return std::min<size_t>( 4, std::max<size_t>( 1, std::strtoul( forced, nullptr, 10 ) ) );
        }
        return batch >= 64 ? ( batch <= 640 ? 3 : 2 ) : ( batch >= 2 ? 2 : 1 );
    }

    /** The fields in front of a block's tables (bzip2.hpp:479-519), read on the caller's thread: what the reader needs to
     * recognise an end-of-stream block behind a data block.  Throws like the reference's block constructor would. */
    [[nodiscard]] BlockHeader
    peekHeader( uint64_t bits ) const
    {
        BlockHeader header;
        header.bits = bits;
        uint64_t pos = bits;
        bool eof = bits > m_source->sizeInBits();
        const uint64_t magic = ( (uint64_t)readBits( *m_source, pos, 24, eof ) << 24 ) | readBits( *m_source, pos, 24, eof );
        header.storedCrc = readBits( *m_source, pos, 32, eof );
        if ( eof ) fail( MI355X_BZ2_ERR_EOF );
        if ( magic == MI355X_BZ2_MAGIC_EOS ) {
            header.endOfStream = true;
            if ( ( pos & 7 ) != 0 ) {
                readBits( *m_source, pos, 8 - (unsigned)( pos & 7 ), eof );   /* padding to the next byte */
                if ( eof ) fail( MI355X_BZ2_ERR_EOF );
            }
            header.bitLength = pos - bits;
            header.endOfFile = pos >= m_source->sizeInBits();
            return header;
        }
        if ( magic != MI355X_BZ2_MAGIC_BLOCK ) {
            char text[96];
            std::snprintf( text, sizeof( text ), "0x%llx at bit offset %llu", (unsigned long long)magic, (unsigned long long)bits );
            fail( MI355X_BZ2_ERR_BAD_MAGIC, text );
        }
        const bool randomised = readBits( *m_source, pos, 1, eof ) != 0;
        if ( eof ) fail( MI355X_BZ2_ERR_EOF );
        if ( randomised ) fail( MI355X_BZ2_ERR_RANDOMIZED );
        const uint32_t origPtr = readBits( *m_source, pos, 24, eof );
        if ( eof ) fail( MI355X_BZ2_ERR_EOF );
        if ( origPtr > 900000 ) fail( MI355X_BZ2_ERR_ORIGPTR_RANGE );
        return header;
    }

    /**
     * The run that holds block number `block` of the finder's list -- from the finished runs, from a launch in flight, or
     * from a launch of its own that goes ahead of everything queued.  Along the way: the access is noted, finished
     * launches are collected, and the range of blocks that the access pattern makes worth decoding ahead is launched.
     * A block's own failure is NOT an error here: it is in its record, for the read that needs the block.
     */
    [[nodiscard]] RunPtr
    demand( size_t block )
    {
        ++m_stats.gets;
        collectFinished();
        m_pattern.note( block );
        if ( m_pattern.inOrder() ) {
            m_ready.dropBefore( block );      /* a sequential reader never comes back (BlockFetcher.hpp:343-346) */
        }

        std::shared_future<RunPtr> pending;
        bool onDemand = false;
        RunPtr run = m_ready.find( block );
        if ( run ) {
            ++( run->lookAhead ? m_stats.prefetch_hits : m_stats.cache_hits );
        } else if ( const auto flight = flightOf( block ); flight != m_flights.end() ) {
            pending = flight->second.result;
            ++m_stats.prefetch_hits;
        } else {
            /* A batch returns when its slowest block is through: the block somebody waits for goes alone, ahead of
             * everything queued, and what should follow it as a second launch on another context. */
            ++m_stats.on_demand_fetches;
            pending = launch( block, 1, /* urgent */ true, /* lookAhead */ false );
            onDemand = true;
        }
        /* (a reader that waits for a launch in flight gains nothing from a partial launch behind it) */
        launchAhead( block, /* somebodyWaits */ onDemand );

        if ( !run ) {
            const auto tWait = std::chrono::steady_clock::now();
            run = pending.get();
            m_stats.wait_seconds += std::chrono::duration<double>( std::chrono::steady_clock::now() - tWait ).count();
            if ( !run ) {
                const std::scoped_lock lock( m_queueMutex );
                fail( MI355X_BZ2_ERR_DEVICE, m_workerError );
            }
            collectFinished();
        }
        return run;
    }

    [[nodiscard]] mi355x_bz2_reader_stats
    statistics() const
    {
        auto result = m_stats;
        const std::scoped_lock lock( m_queueMutex );
        result.batches = m_batches;
        result.blocks_decoded = m_blocksDecoded;
        result.decode_seconds = m_decodeSeconds;
        result.input_resident = m_resident ? 1 : 0;
        result.input_bytes_uploaded = m_uploadedBytes.load( std::memory_order_relaxed );
        return result;
    }

private:
    struct Launch
    {
        size_t first{ 0 };
        std::vector<uint64_t> offsets;
        bool lookAhead{ false };
        std::promise<RunPtr> promise;
        std::shared_ptr<std::atomic<bool> > done;
    };

    struct Flight
    {
        size_t count{ 0 };
        std::shared_future<RunPtr> result;
        std::shared_ptr<std::atomic<bool> > done;
    };

    using Flights = std::map<size_t, Flight>;   /* by first block */

    [[nodiscard]] Flights::iterator
    flightOf( size_t block )
    {
        auto behind = m_flights.upper_bound( block );
        if ( behind == m_flights.begin() ) return m_flights.end();
        --behind;
        return block - behind->first < behind->second.count ? behind : m_flights.end();
    }

    /** A thread of its own: once the whole file is resident on the GPU, let it find the block magics too (k_find_magic,
     * a few ms per GB, on a stream of its own); the host finder threads (about 1.3 GB/s of compressed data on eight
     * cores), which serve the first blocks while the copy runs, would pace the whole reader.  Same offsets, delivered at
     * once; if the scan cannot be used (more matches than its result buffer holds) the host threads carry on. */
    void
    scanOnDevice()
    {
        const auto stopped = [this] {
            const std::scoped_lock lock( m_queueMutex );
            return m_stop;
        };
        while ( mi355x_bz2_input_resident( m_ctxs.front() ) == 0 ) {
            if ( stopped() || m_finder->complete() ) return;
            std::this_thread::sleep_for( std::chrono::milliseconds( 2 ) );
        }
        if ( stopped() || m_finder->complete() ) return;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint64_t> offsets( (size_t)std::min<uint64_t>( m_source->size() / 6 + 16, 1u << 20 ) );
        uint64_t found = 0;
        if ( ( mi355x_bz2_find_magic_device( m_ctxs.front(), MI355X_BZ2_MAGIC_BLOCK, offsets.data(), offsets.size(),
                                             &found ) == MI355X_BZ2_OK ) && ( found <= offsets.size() ) ) {
            offsets.resize( found );
            /* an index given by the caller meanwhile, or a list cut behind trailing garbage, stays: decided inside the finder */
            (void)m_finder->adopt( std::vector<size_t>( offsets.begin(), offsets.end() ), BlockFinder::Authority::SCANNER );
        }
        if ( m_trace ) {
            const auto now = std::chrono::steady_clock::now();
            std::fprintf( stderr, "[reader] t=%.1f ms: file resident, magic scan %.1f ms (%zu blocks)\n",
                          std::chrono::duration<double, std::milli>( now - m_created ).count(),
                          std::chrono::duration<double, std::milli>( now - t0 ).count(), m_finder->size() );
        }
    }

    /** Launches whose worker is through move to the finished runs (one flag per launch is polled, not one future per
     * block: thousands of blocks can be in flight). */
    void
    collectFinished()
    {
        for ( auto flight = m_flights.begin(); flight != m_flights.end(); ) {
            if ( !flight->second.done->load( std::memory_order_acquire ) ) {
                ++flight;
                continue;
            }
            if ( const auto run = flight->second.result.get() ) {
                if ( run->lookAhead ) {
                    for ( const auto& record : run->blocks ) {
                        m_stats.failed_prefetches += record.status != MI355X_BZ2_OK ? 1 : 0;
                    }
                }
                m_ready.insert( run );
                m_inFlightBlocks -= flight->second.count;
                flight = m_flights.erase( flight );
            } else {
                ++flight;   /* device failure: stays, and fails whoever asks for it */
            }
        }
    }

    /** Decide from the access pattern which blocks behind `block` to decode ahead, and launch them as ONE contiguous
     * range if that is worth a launch. */
    void
    launchAhead( size_t block, bool somebodyWaits )
    {
        /* one full batch per context in flight, and one queued: launches are made on the reading thread, which may be
         * blocked on a run when a context becomes free */
        const size_t limit = m_batch * ( m_contexts + ( m_batch >= 64 ? 1 : 0 ) );
        if ( m_inFlightBlocks >= limit ) return;
        /* While a launch is held back for want of blocks (rule at the end) a sequential reader gains about one candidate
         * per call: looking again on every call would cost O(batch^2) per batch. */
        if ( !somebodyWaits && ( m_holdOff > 0 ) && !m_flights.empty() ) {
            --m_holdOff;
            return;
        }
        const auto wanted = m_pattern.ahead( m_window );
        if ( wanted.count == 0 ) return;
        size_t end = wanted.first + wanted.count;
        if ( m_finder->complete() ) end = std::min( end, m_finder->size() );
        /* the first block of the range that is neither finished nor in flight */
        size_t from = wanted.first;
        for ( ;; ) {
            from = m_ready.firstGap( from );
            const auto flight = flightOf( from );
            if ( flight == m_flights.end() ) break;
            from = flight->first + flight->second.count;
        }
        if ( from >= end ) {
            m_holdOff = std::max<size_t>( 1, m_batch / 8 );
            return;
        }
        /* as many as fit: a batch, the room in flight, the window of blocks decoded ahead (finished ones that are still
         * wanted must not be pushed out by newer ones, BlockFetcher.hpp:527-535) */
        /* the first launches are small and grow (64, 256, 1 024 blocks): a context's scratch for a full batch (13 MB per block)
         * takes longer to allocate than the batch to decode, and the reader should not wait for that before its first bytes */
        size_t room = std::min( { m_batch, limit - m_inFlightBlocks, end - from, m_ramp } );
        const size_t ahead = m_ready.blocksWithin( wanted.first, end ) + m_inFlightBlocks;
        room = ahead >= m_window ? 0 : std::min( room, m_window - ahead );
        std::vector<uint64_t> offsets;
        for ( size_t b = from; b < from + room; ++b ) {
            if ( m_ready.covers( b ) || ( flightOf( b ) != m_flights.end() ) ) break;
            const auto offset = m_finder->at( b, BlockFinder::DO_NOT_WAIT ).bits;
            if ( !offset ) break;
            offsets.push_back( *offset );
        }
        if ( offsets.empty() ) {
            m_holdOff = std::max<size_t>( 1, m_batch / 8 );
            return;
        }
        /* A launch has a latency floor (its slowest block, 30 to 45 ms) whatever its size: go along with a launch somebody
         * waits for, otherwise wait until a whole batch can go (the ramp's size while the reader starts up) -- unless nothing is
         * in flight at all or the file ends here. */
        const size_t worthIt = m_batch;
        const bool reachesEnd = m_finder->complete() && ( from + offsets.size() >= m_finder->size() );
        if ( somebodyWaits || ( offsets.size() >= std::min( worthIt, m_ramp ) ) || m_flights.empty() || reachesEnd ) {
            m_ramp = std::min( m_batch, 4 * m_ramp );
            m_stats.prefetches_submitted += offsets.size();
            launch( from, std::move( offsets ), /* urgent */ false, /* lookAhead */ true );
            m_holdOff = 0;
        } else {
            m_holdOff = std::min( worthIt - offsets.size(), std::max<size_t>( 1, m_batch / 8 ) );
        }
    }

    std::shared_future<RunPtr>
    launch( size_t first, size_t count, bool urgent, bool lookAhead )
    {
        std::vector<uint64_t> offsets;
        for ( size_t b = first; b < first + count; ++b ) {
            const auto offset = m_finder->at( b ).bits;
            if ( !offset ) {
                fail( MI355X_BZ2_ERR_LOGIC, "block " + std::to_string( b ) + " is not in the block finder's list" );
            }
            offsets.push_back( *offset );
        }
        return launch( first, std::move( offsets ), urgent, lookAhead );
    }

    std::shared_future<RunPtr>
    launch( size_t first, std::vector<uint64_t> offsets, bool urgent, bool lookAhead )
    {
        auto work = std::make_unique<Launch>();
        work->first = first;
        work->offsets = std::move( offsets );
        work->lookAhead = lookAhead;
        work->done = std::make_shared<std::atomic<bool> >( false );
        Flight flight;
        flight.count = work->offsets.size();
        flight.result = work->promise.get_future().share();
        flight.done = work->done;
        m_inFlightBlocks += flight.count;
        const auto result = flight.result;
        m_flights.emplace( first, std::move( flight ) );
        {
            const std::scoped_lock lock( m_queueMutex );
            if ( urgent ) {
                m_queue.push_front( std::move( work ) );   /* someone is waiting for it */
            } else {
                m_queue.push_back( std::move( work ) );
            }
            m_queueChanged.notify_all();
        }
        return result;
    }

    /** One submission thread per decoder context: takes a launch, decodes the batch, copies it to a page-locked buffer,
     * publishes the run.  Replaces the thread pool of per-block tasks (BlockFetcher.hpp:620-642).
     * The copy of a batch runs in the background while the context's next batch is launched (the context writes to a
     * second output buffer meanwhile), and the page-locked buffer is fetched while the GPU is decoding. */
    void
    workerMain( mi355x_bz2_ctx* const ctx )
    {
        struct Step
        {
            std::unique_ptr<Launch> work;
            std::vector<mi355x_bz2_block_result> results;
            std::shared_ptr<const uint8_t> buffer;
            uint64_t total{ 0 };
            std::chrono::steady_clock::time_point t0;
        };
        const auto failStep = [this, ctx] ( Step& step ) {
            {
                const std::scoped_lock lock( m_queueMutex );
                m_workerError = mi355x_bz2_last_error( ctx );
            }
            step.work->promise.set_value( nullptr );
            step.work->done->store( true, std::memory_order_release );
        };
        /* the copy of `step` has been queued: wait for it and hand the run over */
        const auto publish = [this, ctx, &failStep] ( Step& step ) {
            if ( mi355x_bz2_copy_output_end( ctx ) != MI355X_BZ2_OK ) {
                failStep( step );
                return;
            }
            const auto n = step.results.size();
            auto run = std::make_shared<DecodedRun>();
            run->firstBlock = step.work->first;
            run->bytes = std::move( step.buffer );
            run->totalBytes = step.total;
            run->lookAhead = step.work->lookAhead;
            run->blocks.resize( n );
            for ( size_t i = 0; i < n; ++i ) {
                const auto& r = step.results[i];
                auto& record = run->blocks[i];
                record.bits = step.work->offsets[i];
                record.bitLength = r.encoded_size_bits;
                record.storedCrc = r.header_crc;
                record.computedCrc = r.computed_crc;
                record.endOfStream = r.is_eos != 0;
                record.endOfFile = r.is_eof != 0;
                record.status = r.status;
                if ( r.status == MI355X_BZ2_OK ) {
                    record.at = r.data_offset;
                    record.byteLength = r.decoded_size;
                }
            }
            step.work->promise.set_value( std::move( run ) );
            step.work->done->store( true, std::memory_order_release );
            const std::scoped_lock lock( m_queueMutex );
            ++m_batches;
            m_blocksDecoded += n;
            m_decodeSeconds += std::chrono::duration<double>( std::chrono::steady_clock::now() - step.t0 ).count();
        };

        std::optional<Step> copying;    /* decoded, on its way to the host */
        while ( true ) {
            Step step;
            {
                std::unique_lock lock( m_queueMutex );
                if ( copying && m_queue.empty() && !m_stop ) {
                    /* nothing to launch beside the copy: finish it first, somebody may be waiting for exactly this run */
                    lock.unlock();
                    publish( *copying );
                    copying.reset();
                    lock.lock();
                }
                m_queueChanged.wait( lock, [this] { return m_stop || !m_queue.empty(); } );
                if ( m_queue.empty() ) {
                    break;   /* m_stop */
                }
                step.work = std::move( m_queue.front() );
                m_queue.pop_front();
            }
            step.t0 = std::chrono::steady_clock::now();
            const auto n = (uint32_t)step.work->offsets.size();
            step.results.resize( n );
            int rc = MI355X_BZ2_OK;
            if ( m_resident ) {
                rc = mi355x_bz2_decode_batch_begin( ctx, step.work->offsets.data(), n );
            } else {
                /* bounded residency: the bytes from the word of the first block's magic to where the last block can end at
                 * the latest (900 000 symbols of 20 bits and the tables in front of them: below 2 400 000 bytes), and the
                 * blocks' offsets inside that range.  The copy is queued on the context's input stream; while a batch of
                 * this context is still being copied out, it lands in the context's second input buffer. */
                const auto [lowest, highest] = std::minmax_element( step.work->offsets.begin(), step.work->offsets.end() );
                const uint64_t from = ( *lowest / 8 ) & ~uint64_t( 3 );
                const uint64_t to = std::min<uint64_t>( m_source->size(), *highest / 8 + 2400000 );
                std::vector<uint64_t> relative( step.work->offsets );
                for ( auto& bits : relative ) bits -= 8 * from;
                rc = from < to ? mi355x_bz2_set_input_host_async( ctx, m_source->bytes() + from, to - from )
                               : MI355X_BZ2_ERR_INVALID_ARGUMENT;
                if ( rc == MI355X_BZ2_OK ) rc = mi355x_bz2_decode_batch_begin( ctx, relative.data(), n );
                m_uploadedBytes.fetch_add( to - from, std::memory_order_relaxed );
            }
            const auto t1 = std::chrono::steady_clock::now();
            /* while the GPU works: the run in front of this one, and the memory for this one (a block of the usual
             * compressors decodes to at most level x 100 000 bytes; if these decode to more -- later streams of the file
             * may have a higher level -- a second buffer is fetched below) */
            if ( copying ) {
                publish( *copying );
                copying.reset();
            }
            size_t capacity = 0;
            if ( rc == MI355X_BZ2_OK ) {
                step.buffer = m_hostBuffers->get( (size_t)n * m_level * 100000 + 4096, m_hostBuffers, &capacity );
            }
            const auto t2 = std::chrono::steady_clock::now();
            if ( rc == MI355X_BZ2_OK ) {
                rc = mi355x_bz2_decode_batch_end( ctx, step.results.data(), &step.total );
            }
            const auto t3 = std::chrono::steady_clock::now();
            if ( rc == MI355X_BZ2_OK ) {
                if ( step.total > capacity ) step.buffer = m_hostBuffers->get( step.total, m_hostBuffers );
                if ( !step.buffer ) {
                    rc = MI355X_BZ2_ERR_DEVICE;
                } else {
                    rc = mi355x_bz2_copy_output_begin( ctx, 0, step.total, const_cast<uint8_t*>( step.buffer.get() ) );
                }
            }
            if ( m_trace ) {
                const auto ms = [] ( auto a, auto b ) { return std::chrono::duration<double, std::milli>( b - a ).count(); };
                std::fprintf( stderr, "[reader] t=%.1f ms: blocks [%zu, +%u) on ctx %p: launch %.1f ms, previous run + host "
                              "buffer %.1f ms, rest of the decode %.1f ms, %.0f MB\n", ms( m_created, step.t0 ),
                              step.work->first, n, (void*)ctx, ms( step.t0, t1 ), ms( t1, t2 ), ms( t2, t3 ), step.total / 1e6 );
            }
            if ( rc != MI355X_BZ2_OK ) {
                failStep( step );
                continue;
            }
            copying = std::move( step );
        }
        if ( copying ) publish( *copying );
    }

private:
    const std::shared_ptr<Source> m_source;
    const std::shared_ptr<BlockFinder> m_finder;
    const size_t m_batch;        /* blocks per launch = the `parallelization` of the API */
    const size_t m_contexts;
    const size_t m_window;

    SequentialityTracker m_pattern;
    RunCache<DecodedRun> m_ready;
    Flights m_flights;
    size_t m_inFlightBlocks{ 0 };
    size_t m_holdOff{ 0 };
    size_t m_ramp{ 64 };         /* blocks of the next look-ahead launch while the reader is starting up */
    unsigned m_level{ 9 };       /* of the first stream: 100 000 bytes per block and level */
    bool m_resident{ true };     /* the whole compressed file on the GPU, or the range of every launch (fitsDevice) */
    std::atomic<uint64_t> m_uploadedBytes{ 0 };

    std::vector<mi355x_bz2_ctx*> m_ctxs;
    const std::shared_ptr<PinnedPool> m_hostBuffers{ std::make_shared<PinnedPool>() };
    std::vector<std::thread> m_workers;
    std::thread m_scanner;       /* see scanOnDevice */
    mutable std::mutex m_queueMutex;
    std::condition_variable m_queueChanged;
    std::deque<std::unique_ptr<Launch> > m_queue;
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
class StreamReader
{
public:
    /* parallelization 0: a batch of this many blocks keeps the GPU busy while a single cold read still returns after
     * one small launch */
    static constexpr size_t DEFAULT_BATCH = 512;

    using Sink = std::function<void( const uint8_t*, uint64_t )>;
    using OffsetMap = std::map<size_t, size_t>;

    StreamReader( std::shared_ptr<Source> source, size_t parallelization, int device ) :
        m_source( std::move( source ) ),
        m_batch( parallelization == 0 ? DEFAULT_BATCH : parallelization ),
        m_device( device ),
        m_checkStreamCrc( parallelization == 1 )   /* the reference's serial reader checks, the parallel one does not */
    {}

    void setVerifyStreamCrc( bool enable ) { m_checkStreamCrc = enable; }
    [[nodiscard]] uint64_t streamsVerified() const { return m_streamsVerified; }

    void
    close()
    {
        m_scheduler.reset();
        m_finder.reset();
        m_source.reset();
    }

    [[nodiscard]] bool closed() const { return !m_source; }
    [[nodiscard]] bool eof() const { return m_atEnd; }

    /** Decoded size, once the whole file has been seen (or an index was loaded). */
    [[nodiscard]] std::optional<size_t>
    size() const
    {
        if ( !m_index.sealed() ) return std::nullopt;
        return (size_t)m_index.last().second;
    }

    [[nodiscard]] size_t
    tell() const
    {
        if ( !m_atEnd ) return m_position;
        const auto total = size();
        if ( !total ) {
            fail( MI355X_BZ2_ERR_LOGIC, "at the end of the file its size has to be known" );
        }
        return *total;
    }

    /**
     * Up to `wanted` bytes from the current position into `sink` (an empty sink discards them).  Two things can happen
     * per step: the position lies in a block the index knows -- then its run is fetched and the bytes go out; or it lies
     * behind everything indexed -- then the next block of the file is decoded and indexed, which is where decode errors,
     * end-of-stream blocks, stream CRCs and the end of the file come up.
     */
    size_t
    read( const Sink& sink, size_t wanted = std::numeric_limits<size_t>::max() )
    {
        if ( closed() ) {
            fail( MI355X_BZ2_ERR_CLOSED, "read on a closed reader" );
        }
        size_t produced = 0;
        while ( ( produced < wanted ) && !m_atEnd ) {
            const auto span = m_index.locate( m_position );
            if ( !span.covers( m_position ) ) {
                if ( !indexNextBlock() ) {
                    m_atEnd = true;
                }
                continue;
            }
            const auto run = scheduler().demand( finder().numberOf( span.bits ) );
            const auto& record = recordIn( *run, span.bits );
            if ( record.status != MI355X_BZ2_OK ) {
                fail( record.status, "block at bit offset " + std::to_string( span.bits ) );
            }
            const uint64_t inBlock = m_position - span.bytes;
            if ( inBlock >= record.byteLength ) {
                fail( MI355X_BZ2_ERR_LOGIC, "the block index promises more bytes than the block decodes to" );
            }
            const uint64_t piece = std::min<uint64_t>( record.byteLength - inBlock, wanted - produced );
            if ( sink ) {
                sink( run->bytes.get() + record.at + inBlock, piece );
            }
            produced += piece;
            m_position += piece;
        }
        return produced;
    }

    /** read( fd, buffer, n ) of the interface: to a file descriptor and / or a buffer, or nowhere
     * (BZ2ReaderInterface.hpp:35-57). */
    size_t
    read( int outputFileDescriptor, char* outputBuffer, size_t wanted )
    {
        if ( ( outputFileDescriptor < 0 ) && ( outputBuffer == nullptr ) ) {
            return read( Sink(), wanted );
        }
        uint64_t copied = 0;
        return read( [&] ( const uint8_t* bytes, uint64_t size ) {
            for ( uint64_t written = 0; ( outputFileDescriptor >= 0 ) && ( written < size ); ) {
                const auto n = ::write( outputFileDescriptor, bytes + written, (size_t)std::min<uint64_t>( size - written, 1u << 30 ) );
                if ( n > 0 ) {
                    written += (uint64_t)n;
                } else if ( !( n < 0 && errno == EINTR ) ) {
                    fail( MI355X_BZ2_ERR_IO, std::string( "write: " ) + strerror( errno ) );
                }
            }
            if ( outputBuffer != nullptr ) {
                std::memcpy( outputBuffer + copied, bytes, size );
            }
            copied += size;
        }, wanted );
    }

    /** Seeking never decodes more than it has to: inside the indexed part of the file (or anywhere once the index is
     * complete) only the position changes; a target behind the indexed part is reached by decoding forward. */
    size_t
    seek( long long offset, int whence )
    {
        if ( closed() ) {
            fail( MI355X_BZ2_ERR_CLOSED, "seek on a closed reader" );
        }
        if ( ( whence == SEEK_END ) && !m_index.sealed() ) {
            read( Sink() );      /* the size is only known at the end */
        }
        long long target = offset;
        if ( whence == SEEK_CUR ) {
            target += (long long)tell();
        } else if ( whence == SEEK_END ) {
            target += (long long)size().value();
        } else if ( whence != SEEK_SET ) {
            fail( MI355X_BZ2_ERR_INVALID_ARGUMENT, "invalid seek origin" );
        }
        target = std::max( target, 0LL );
        if ( const auto total = size() ) {
            target = std::min<long long>( target, (long long)*total );
        }
        const auto goal = (size_t)target;
        const size_t here = tell();
        if ( goal == here ) {
            return here;
        }
        if ( ( goal < here ) || ( goal < m_index.frontier() ) ) {
            m_position = goal;
            m_atEnd = false;
        } else if ( m_index.sealed() ) {
            m_position = (size_t)m_index.last().second;
            m_atEnd = true;
        } else {
            m_position = (size_t)m_index.frontier();
            m_atEnd = false;
            read( Sink(), goal - m_position );
        }
        return tell();
    }

    [[nodiscard]] bool indexComplete() const { return m_index.sealed(); }

    [[nodiscard]] OffsetMap
    blockOffsets()
    {
        if ( !m_index.sealed() ) {
            read( Sink() );
            if ( !m_index.sealed() || !finder().complete() ) {
                fail( MI355X_BZ2_ERR_LOGIC, "the whole file was read but its block index is not complete" );
            }
        }
        return availableBlockOffsets();
    }

    [[nodiscard]] OffsetMap
    availableBlockOffsets() const
    {
        const auto pairs = m_index.snapshot();
        return { pairs.begin(), pairs.end() };
    }

    /** Import of a complete index: at least one block and the end-of-file entry.  Entries followed by an equal decoded
     * offset are end-of-stream blocks and are not handed to the finder (ParallelBZ2Reader.hpp:365-378, 456-475). */
    void
    setBlockOffsets( const OffsetMap& offsets )
    {
        if ( offsets.empty() ) {
            fail( MI355X_BZ2_ERR_INVALID_ARGUMENT, "an empty index cannot be loaded: open a new reader instead" );
        }
        handOffsetsToFinder( offsets );
        if ( offsets.size() < 2 ) {
            fail( MI355X_BZ2_ERR_INVALID_ARGUMENT, "an index needs at least one block and the end-of-file entry" );
        }
        m_index.assign( BlockIndex::Pairs( offsets.begin(), offsets.end() ) );
    }

    [[nodiscard]] size_t
    tellCompressed() const
    {
        const auto span = m_index.locate( m_position );
        if ( span.covers( m_position ) ) return (size_t)span.bits;
        return m_index.empty() ? 0 : (size_t)m_index.last().first;
    }

    void
    joinThreads()
    {
        m_scheduler.reset();
        m_finder.reset();
    }

    [[nodiscard]] mi355x_bz2_reader_stats
    statistics() const
    {
        return m_scheduler ? m_scheduler->statistics() : mi355x_bz2_reader_stats{};
    }

private:
    [[nodiscard]] static const BlockRecord&
    recordIn( const DecodedRun& run, uint64_t bits )
    {
        const auto record = std::lower_bound( run.blocks.begin(), run.blocks.end(), bits,
                                              [] ( const BlockRecord& r, uint64_t b ) { return r.bits < b; } );
        if ( ( record == run.blocks.end() ) || ( record->bits != bits ) ) {
            fail( MI355X_BZ2_ERR_LOGIC, "the decoded run does not hold the block it was fetched for" );
        }
        return *record;
    }

    /**
     * Decodes the first block the index does not know yet and appends it -- together with the end-of-stream block behind
     * it, if there is one (those have a magic of their own and are not in the finder's list).  False at the end of the
     * file, where the index is sealed.  The stream CRC is folded over the blocks in this order (BZ2Reader.hpp:481-484).
     */
    bool
    indexNextBlock()
    {
        const size_t number = m_index.dataBlocks();
        const auto bits = finder().at( number ).bits;
        if ( !bits ) {
            m_index.seal();
            return false;
        }
        const auto run = scheduler().demand( number );
        const auto& block = run->at( number );
        if ( block.status != MI355X_BZ2_OK ) {
            fail( block.status, "block at bit offset " + std::to_string( *bits ) );
        }
        m_index.append( block.bits, block.bitLength, block.byteLength );
        m_streamCrc = ( ( m_streamCrc << 1U ) | ( m_streamCrc >> 31U ) ) ^ block.computedCrc;
        if ( block.endOfFile ) {
            return true;
        }
        const auto next = scheduler().peekHeader( block.bits + block.bitLength );
        if ( !next.endOfStream ) {
            return true;
        }
        m_index.append( next.bits, next.bitLength, 0 );
        /* the end-of-stream block carries the CRC of the whole stream (BZ2Reader.hpp:406-416) */
        const auto folded = std::exchange( m_streamCrc, 0U );
        if ( m_checkStreamCrc ) {
            if ( next.storedCrc != folded ) {
                std::stringstream message;
                message << "stream CRC 0x" << std::hex << next.storedCrc << " in the file, 0x" << folded << " calculated";
                fail( MI355X_BZ2_ERR_STREAM_CRC, message.str() );
            }
            ++m_streamsVerified;
        }
        const uint64_t behind = next.bits + next.bitLength;
        if ( ( behind < m_source->sizeInBits() )
             && ( mi355x_bz2_read_stream_header( m_source->bytes(), m_source->size(), behind ) == 0 ) ) {
            std::cerr << "[Warning] Trailing garbage after EOF ignored!\n";
            m_finder->cut( m_index.dataBlocks() );   /* whatever the finder saw in the garbage is not a block */
        }
        return true;
    }

    BlockFinder&
    finder()
    {
        if ( !m_finder ) {
            const unsigned cores = std::max( 1u, std::thread::hardware_concurrency() );
            /* scan ahead of the decoder by a few batches (3 * hardware_concurrency blocks in the reference, BlockFinder.hpp:213) */
            m_finder = std::make_shared<BlockFinder>( m_source->bytes(), m_source->size(), MI355X_BZ2_MAGIC_BLOCK,
                                                      std::max<size_t>( 3 * cores, 4 * m_batch ),
                                                      std::min( 8u, std::max( 1u, cores / 2 ) ) );
            if ( m_index.sealed() ) {
                const auto pairs = m_index.snapshot();
                handOffsetsToFinder( OffsetMap( pairs.begin(), pairs.end() ) );
            }
        }
        return *m_finder;
    }

    BatchScheduler&
    scheduler()
    {
        if ( !m_scheduler ) {
            (void)finder();
            m_scheduler = std::make_unique<BatchScheduler>( m_source, m_finder, m_batch, m_device );
            if ( !m_finder->complete() ) {
                m_finder->startThreads();
            }
        }
        return *m_scheduler;
    }

    void
    handOffsetsToFinder( const OffsetMap& offsets )
    {
        if ( offsets.empty() ) {
            fail( MI355X_BZ2_ERR_INVALID_ARGUMENT, "a list of block offsets is required" );
        }
        std::vector<size_t> dataBlocks;
        for ( auto entry = offsets.begin(), behind = std::next( entry ); behind != offsets.end(); ++entry, ++behind ) {
            if ( entry->second != behind->second ) {
                dataBlocks.push_back( entry->first );
            }
        }
        (void)finder().adopt( std::move( dataBlocks ), BlockFinder::Authority::CALLER );
    }

private:
    std::shared_ptr<Source> m_source;
    const size_t m_batch;
    const int m_device;
    size_t m_position{ 0 };
    bool m_atEnd{ false };

    bool m_checkStreamCrc;
    uint32_t m_streamCrc{ 0 };
    uint64_t m_streamsVerified{ 0 };

    std::shared_ptr<BlockFinder> m_finder;
    BlockIndex m_index;
    std::unique_ptr<BatchScheduler> m_scheduler;
};
}  // namespace mi355x

/* ================================================================================================ C ABI */
struct mi355x_bz2_reader
{
    std::unique_ptr<mi355x::StreamReader> reader;
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
        r->reader = std::make_unique<mi355x::StreamReader>( std::move( source ), parallelization, device );
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
    return guarded( r, [&] ( mi355x::StreamReader& reader ) {
        const auto n = reader.read( fd, static_cast<char*>( buffer ), (size_t)nBytes );
        if ( nRead != nullptr ) *nRead = n;
    } );
}

int
mi355x_bz2_reader_seek( mi355x_bz2_reader* r, int64_t offset, int whence, uint64_t* newPosition )
{
    return guarded( r, [&] ( mi355x::StreamReader& reader ) {
        const auto p = reader.seek( offset, whence );
        if ( newPosition != nullptr ) *newPosition = p;
    } );
}

uint64_t
mi355x_bz2_reader_tell( const mi355x_bz2_reader* r )
{
    uint64_t result = 0;
    guarded( const_cast<mi355x_bz2_reader*>( r ), [&] ( mi355x::StreamReader& reader ) { result = reader.tell(); } );
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
             [&] ( mi355x::StreamReader& reader ) { result = reader.tellCompressed(); } );
    return result;
}

int
mi355x_bz2_reader_block_offsets_complete( const mi355x_bz2_reader* r )
{
    return ( r != nullptr && r->reader && r->reader->indexComplete() ) ? 1 : 0;
}

int
mi355x_bz2_reader_block_offsets( mi355x_bz2_reader* r, uint64_t* bits, uint64_t* bytes, uint64_t capacity, uint64_t* n )
{
    int inner = MI355X_BZ2_OK;
    const int rc = guarded( r, [&] ( mi355x::StreamReader& reader ) {
        inner = copyOffsets( reader.blockOffsets(), bits, bytes, capacity, n );
    } );
    return rc != MI355X_BZ2_OK ? rc : inner;
}

int
mi355x_bz2_reader_available_block_offsets( const mi355x_bz2_reader* r, uint64_t* bits, uint64_t* bytes,
                                           uint64_t capacity, uint64_t* n )
{
    int inner = MI355X_BZ2_OK;
    const int rc = guarded( const_cast<mi355x_bz2_reader*>( r ), [&] ( mi355x::StreamReader& reader ) {
        inner = copyOffsets( reader.availableBlockOffsets(), bits, bytes, capacity, n );
    } );
    return rc != MI355X_BZ2_OK ? rc : inner;
}

int
mi355x_bz2_reader_set_block_offsets( mi355x_bz2_reader* r, const uint64_t* bits, const uint64_t* bytes, uint64_t n )
{
    if ( n > 0 && ( bits == nullptr || bytes == nullptr ) ) return MI355X_BZ2_ERR_INVALID_ARGUMENT;
    return guarded( r, [&] ( mi355x::StreamReader& reader ) {
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
    return guarded( r, [] ( mi355x::StreamReader& reader ) { reader.joinThreads(); } );
}

int
mi355x_bz2_reader_set_verify_stream_crc( mi355x_bz2_reader* r, int enable )
{
    return guarded( r, [enable] ( mi355x::StreamReader& reader ) { reader.setVerifyStreamCrc( enable != 0 ); } );
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
                    [&] ( mi355x::StreamReader& reader ) { *stats = reader.statistics(); } );
}

}  // extern "C"
