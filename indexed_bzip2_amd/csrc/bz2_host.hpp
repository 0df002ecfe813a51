/**
 * bz2_host.hpp -- host-side pieces of the reader, batch-oriented for a GPU backend.  Product code (never touches oracle/).
 * The BEHAVIOUR of the reference's scheduler classes is kept where callers can observe it and is pinned
 * by answers recorded from the reference's own classes (tests/golden/host_vectors.txt, replayed by
 * tests/native/host_known_answers.cpp); the data structures are this design's own:
 *
 *   BlockIndex            two flat arrays (compressed bit offset, decoded byte offset) + the size of the open last block;
 *                         what rapidgzip::BlockMap answers                 src/core/BlockMap.hpp:26-295
 *   SequentialityTracker  the last few block numbers that were asked for -> ONE range [first, first + count) worth
 *                         decoding ahead; same amounts as FetchNextAdaptive  src/core/Prefetcher.hpp:82-217
 *   RunCache              decoded batches ("runs" of consecutive blocks that share one host buffer), budgeted in blocks,
 *                         least recently used run out first; the role of Cache + the prefetch cache,
 *                                                                          src/core/Cache.hpp:117-296
 *   BlockFinder           producer of block offsets, chunk-wise magic scan   src/core/BlockFinder.hpp:36-219,
 *                                                                          src/core/StreamedResults.hpp:26-156
 */
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace mi355x
{
/** Scan byte range [begin,end) (first byte of the match) of bytes[0,size) for a 48-bit pattern; bz2_finder.cpp. */
void
scanMagicRange( const uint8_t* bytes, uint64_t size, uint64_t magic48, uint64_t begin, uint64_t end,
                std::vector<uint64_t>& found );

/* ------------------------------------------------------------------------------------------------ block index */
/**
 * Where every block starts, compressed (bits) and decoded (bytes).  End-of-stream blocks are entries of decoded size 0;
 * seal() adds the end-of-file entry, after which the index is what IndexedBzip2File.block_offsets() returns.
 * Entries only ever grow at the end, so the two columns are plain sorted arrays and every look-up is a binary search.
 */
class BlockIndex
{
public:
    using Pairs = std::vector<std::pair<uint64_t, uint64_t> >;

    struct Span
    {
        size_t   ordinal{ 0 };      /* position in the index, end-of-stream entries included */
        uint64_t bits{ 0 }, bitLength{ 0 };
        uint64_t bytes{ 0 }, byteLength{ 0 };

        [[nodiscard]] bool
        covers( uint64_t byteOffset ) const
        {
            return ( byteOffset >= bytes ) && ( byteOffset - bytes < byteLength );
        }
    };

    /** A block decoded for the first time, in file order.  Returns where its bytes start.  A block that is already known
     * (decoded again after a seek) is accepted if its size agrees and is otherwise an error, like any offset that does not
     * continue the index. */
    uint64_t
    append( uint64_t bits, uint64_t bitLength, uint64_t byteLength )
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        if ( m_sealed ) {
            throw std::invalid_argument( "the block index is complete: nothing can be appended" );
        }
        if ( m_bits.empty() || ( bits > m_bits.back() ) ) {
            const uint64_t at = m_bits.empty() ? 0 : m_bytes.back() + m_openBytes;
            m_bits.push_back( bits );
            m_bytes.push_back( at );
            m_openBits = bitLength;
            m_openBytes = byteLength;
            m_emptyBlocks += byteLength == 0 ? 1 : 0;
            return at;
        }
        const auto known = std::lower_bound( m_bits.begin(), m_bits.end(), bits );
        const auto i = static_cast<size_t>( known - m_bits.begin() );
        if ( *known != bits ) {
            throw std::invalid_argument( "block offsets have to arrive in increasing order" );
        }
        if ( i + 1 == m_bits.size() ) {
            throw std::logic_error( "the open last block cannot be appended twice" );
        }
        if ( m_bytes[i + 1] - m_bytes[i] != byteLength ) {
            throw std::invalid_argument( "a known block came back with another decoded size" );
        }
        return m_bytes[i];
    }

    /** The last entry that starts at or before byteOffset (among entries with equal starts -- an end-of-stream block and
     * the block behind it -- the later one).  An all-zero Span if there is none. */
    [[nodiscard]] Span
    locate( uint64_t byteOffset ) const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        const auto behind = std::upper_bound( m_bytes.begin(), m_bytes.end(), byteOffset );
        if ( behind == m_bytes.begin() ) {
            return {};
        }
        return spanAt( static_cast<size_t>( behind - m_bytes.begin() ) - 1 );
    }

    /** Decoded bytes covered so far: the end of the last block. */
    [[nodiscard]] uint64_t
    frontier() const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        return m_bytes.empty() ? 0 : m_bytes.back() + m_openBytes;
    }

    /** Data blocks, i.e. entries that are neither end-of-stream blocks nor the end-of-file entry. */
    [[nodiscard]] size_t
    dataBlocks() const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        return m_bits.size() - m_emptyBlocks;
    }

    /** No more blocks: close the last one with the end-of-file entry {bits behind it, total decoded size}. */
    void
    seal()
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        if ( m_sealed ) {
            return;
        }
        if ( m_bits.empty() ) {
            m_bits.push_back( m_openBits );
            m_bytes.push_back( m_openBytes );
        } else if ( ( m_openBits | m_openBytes ) != 0 ) {
            m_bits.push_back( m_bits.back() + m_openBits );
            m_bytes.push_back( m_bytes.back() + m_openBytes );
        }
        m_openBits = m_openBytes = 0;
        m_sealed = true;
    }

    [[nodiscard]] bool
    sealed() const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        return m_sealed;
    }

    /** Import of a complete index (sorted by bit offset): entries followed by an equal decoded offset are end-of-stream
     * blocks, the last entry is the end of the file. */
    void
    assign( const Pairs& sortedPairs )
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        m_bits.clear();
        m_bytes.clear();
        m_bits.reserve( sortedPairs.size() );
        m_bytes.reserve( sortedPairs.size() );
        for ( const auto& [bits, bytes] : sortedPairs ) {
            m_bits.push_back( bits );
            m_bytes.push_back( bytes );
        }
        m_emptyBlocks = m_bytes.empty() ? 0 : 1;
        for ( size_t i = 0; i + 1 < m_bytes.size(); ++i ) {
            m_emptyBlocks += m_bytes[i] == m_bytes[i + 1] ? 1 : 0;
        }
        m_openBits = m_openBytes = 0;
        m_sealed = true;
    }

    [[nodiscard]] Pairs
    snapshot() const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        Pairs result( m_bits.size() );
        for ( size_t i = 0; i < m_bits.size(); ++i ) {
            result[i] = { m_bits[i], m_bytes[i] };
        }
        return result;
    }

    [[nodiscard]] std::pair<uint64_t, uint64_t>
    last() const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        if ( m_bits.empty() ) {
            throw std::out_of_range( "the block index is empty" );
        }
        return { m_bits.back(), m_bytes.back() };
    }

    [[nodiscard]] bool
    empty() const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        return m_bits.empty();
    }

private:
    [[nodiscard]] Span
    spanAt( size_t i ) const
    {
        Span span;
        span.ordinal = i;
        span.bits = m_bits[i];
        span.bytes = m_bytes[i];
        if ( i + 1 < m_bits.size() ) {
            if ( m_bytes[i + 1] < m_bytes[i] ) {
                throw std::logic_error( "decoded offsets of the block index decrease" );
            }
            span.bitLength = m_bits[i + 1] - m_bits[i];
            span.byteLength = m_bytes[i + 1] - m_bytes[i];
        } else {
            span.bitLength = m_openBits;
            span.byteLength = m_openBytes;
        }
        return span;
    }

    mutable std::mutex m_lock;
    std::vector<uint64_t> m_bits, m_bytes;          /* sorted; m_bytes non-decreasing */
    uint64_t m_openBits{ 0 }, m_openBytes{ 0 };     /* size of the last block while the index is still growing */
    size_t m_emptyBlocks{ 0 };
    bool m_sealed{ false };
};

/* ------------------------------------------------------------------------------------------------ access pattern */
/**
 * Remembers the last few block numbers a reader asked for and says how far ahead decoding is worth it: everything the
 * caller allows while the accesses came in order, nothing for random accesses, and limit^(streak / depth) in between, so
 * that one stray seek does not stop a sequential reader and a few ordered accesses restart it.  The answer is one range.
 */
class SequentialityTracker
{
public:
    struct Range
    {
        size_t first{ 0 }, count{ 0 };
    };

    explicit
    SequentialityTracker( size_t depth = 3 ) :
        m_depth( std::max<size_t>( depth, 1 ) )
    {
        m_recent.reserve( m_depth + 1 );
    }

    /** A block was asked for.  Asking for the same block again (small reads inside one block) changes nothing. */
    void
    note( size_t block )
    {
        if ( !m_recent.empty() && ( m_recent.front() == block ) ) {
            return;
        }
        m_recent.insert( m_recent.begin(), block );
        if ( m_recent.size() > m_depth ) {
            m_recent.pop_back();
        }
    }

    /** True while every remembered access followed its predecessor (also with fewer than two accesses). */
    [[nodiscard]] bool
    inOrder() const noexcept
    {
        return steps() + 1 >= m_recent.size();
    }

    [[nodiscard]] Range
    ahead( size_t limit ) const
    {
        const size_t known = m_recent.size();
        if ( ( known == 0 ) || ( limit == 0 ) ) {
            return {};
        }
        const size_t next = m_recent.front() + 1;
        if ( known == 1 ) {
            return { next, limit };   /* a fresh reader: one miss, then everything in parallel */
        }
        if ( steps() == 0 ) {
            return {};                /* nothing but jumps */
        }
        /* how much of the memory is one unbroken run up to the newest access */
        size_t streak = 0;
        while ( ( streak + 1 < known ) && ( m_recent[streak] == m_recent[streak + 1] + 1 ) ) {
            ++streak;
        }
        const double confidence = streak == 0 ? 0.0 : static_cast<double>( streak + 1 ) / static_cast<double>( known );
        const double amount = std::round( std::exp2( confidence * std::log2( static_cast<double>( limit ) ) ) );
        return { next, static_cast<size_t>( std::max( 0.0, amount ) ) };
    }

private:
    /** Number of remembered accesses that directly followed the one before them. */
    [[nodiscard]] size_t
    steps() const noexcept
    {
        size_t result = 0;
        for ( size_t i = 0; i + 1 < m_recent.size(); ++i ) {
            result += m_recent[i] == m_recent[i + 1] + 1 ? 1 : 0;
        }
        return result;
    }

    size_t m_depth;
    std::vector<size_t> m_recent;   /* newest first */
};

/* ------------------------------------------------------------------------------------------------ decoded runs */
/**
 * Decoded batches by the number of their first block.  `Run` provides first() and count() (consecutive block numbers
 * [first, first + count)).  The budget is in blocks; when it is exceeded the least recently used runs go, never the one
 * that was just put in.  Statistics in the spirit of Cache::Statistics (src/core/Cache.hpp:120-140): a run that leaves
 * without ever having been asked for counts as unused; replacing a run by a new one for the same blocks does not.
 */
template<typename Run>
class RunCache
{
public:
    struct Statistics
    {
        size_t hits{ 0 }, misses{ 0 }, unusedRuns{ 0 }, evictions{ 0 }, maxBlocks{ 0 };
    };

    explicit
    RunCache( size_t blockBudget ) :
        m_budget( blockBudget )
    {}

    /** The run that holds `block`, marked as used now; nullptr if none does. */
    [[nodiscard]] std::shared_ptr<const Run>
    find( size_t block )
    {
        const auto entry = locate( block );
        if ( entry == m_runs.end() ) {
            ++m_statistics.misses;
            return nullptr;
        }
        ++m_statistics.hits;
        ++entry->second.uses;
        entry->second.lastUse = ++m_clock;
        return entry->second.run;
    }

    [[nodiscard]] bool
    covers( size_t block ) const
    {
        return const_cast<RunCache*>( this )->locate( block ) != m_runs.end();
    }

    /** The first block at or behind `block` that no run holds. */
    [[nodiscard]] size_t
    firstGap( size_t block ) const
    {
        for ( auto entry = const_cast<RunCache*>( this )->locate( block ); entry != m_runs.end();
              entry = const_cast<RunCache*>( this )->locate( block ) ) {
            block = entry->first + entry->second.run->count();
        }
        return block;
    }

    /** How many of the blocks [first, end) some run holds. */
    [[nodiscard]] size_t
    blocksWithin( size_t first, size_t end ) const
    {
        size_t result = 0;
        for ( const auto& [runFirst, entry] : m_runs ) {
            const size_t lo = std::max( first, runFirst ), hi = std::min( end, runFirst + entry.run->count() );
            result += hi > lo ? hi - lo : 0;
        }
        return result;
    }

    void
    insert( std::shared_ptr<const Run> run )
    {
        if ( !run || ( run->count() == 0 ) ) {
            return;
        }
        const size_t first = run->first();
        if ( const auto same = m_runs.find( first ); same != m_runs.end() ) {
            /* the same blocks decoded again: not an eviction, and its use count carries over */
            m_blocks -= same->second.run->count();
            m_blocks += run->count();
            same->second.run = std::move( run );
            same->second.lastUse = ++m_clock;
        } else {
            m_blocks += run->count();
            m_runs.emplace( first, Entry{ std::move( run ), ++m_clock, 0 } );
        }
        while ( ( m_blocks > m_budget ) && ( m_runs.size() > 1 ) ) {
            auto oldest = m_runs.end();
            for ( auto entry = m_runs.begin(); entry != m_runs.end(); ++entry ) {
                if ( ( entry->first != first ) && ( ( oldest == m_runs.end() ) || ( entry->second.lastUse < oldest->second.lastUse ) ) ) {
                    oldest = entry;
                }
            }
            if ( oldest == m_runs.end() ) {
                break;
            }
            drop( oldest );
        }
        m_statistics.maxBlocks = std::max( m_statistics.maxBlocks, m_blocks );
    }

    /** A reader that goes through the file in order never comes back: forget every run that ends at or before `block`. */
    void
    dropBefore( size_t block )
    {
        for ( auto entry = m_runs.begin(); entry != m_runs.end(); ) {
            if ( entry->first + entry->second.run->count() <= block ) {
                entry = drop( entry );
            } else {
                ++entry;
            }
        }
    }

    void
    clear()
    {
        m_runs.clear();
        m_blocks = 0;
    }

    [[nodiscard]] size_t blocks() const { return m_blocks; }
    [[nodiscard]] size_t runs() const { return m_runs.size(); }
    [[nodiscard]] size_t budget() const { return m_budget; }
    [[nodiscard]] Statistics statistics() const { return m_statistics; }

private:
    struct Entry
    {
        std::shared_ptr<const Run> run;
        uint64_t lastUse{ 0 };
        size_t uses{ 0 };
    };

    using Runs = std::map<size_t, Entry>;

    typename Runs::iterator
    locate( size_t block )
    {
        auto behind = m_runs.upper_bound( block );
        if ( behind == m_runs.begin() ) {
            return m_runs.end();
        }
        --behind;
        return block - behind->first < behind->second.run->count() ? behind : m_runs.end();
    }

    typename Runs::iterator
    drop( typename Runs::iterator entry )
    {
        m_statistics.unusedRuns += entry->second.uses == 0 ? 1 : 0;
        ++m_statistics.evictions;
        m_blocks -= entry->second.run->count();
        return m_runs.erase( entry );
    }

    size_t m_budget;
    Runs m_runs;
    size_t m_blocks{ 0 };
    uint64_t m_clock{ 0 };
    Statistics m_statistics;
};

/* ------------------------------------------------------------------------------------------------ block finder */
/**
 * The list of data-block bit offsets of a file, numbered from 0, which fills while the file is being read.  Two sources
 * feed it: a scan thread of its own (bz2_finder.cpp over chunks of the host bytes, only as far ahead as somebody asks),
 * and whoever knows the whole list at once (an index given by the caller, the GPU's k_find_magic over the resident file,
 * the reader when it finds trailing garbage).  What the reference splits into BlockFinder + StreamedResults
 * (src/core/BlockFinder.hpp:36-219, src/core/StreamedResults.hpp:26-156) is ONE monitor here:
 *
 *   - every member is guarded by m_lock; a list that is `complete` never changes its source again (it can only be cut);
 *   - the scan thread carries the `epoch` it was started in and appends only while that is still the current one: adopt()
 *     and cut() advance the epoch INSIDE the critical section in which they replace the list, so nothing the thread found
 *     before can land behind a list it does not belong to, and no thread is started for a complete list (round 2 stopped
 *     the thread first and took the lock afterwards: a reader could restart the scan in between);
 *   - threads are joined outside the lock (the thread needs the lock to leave).
 */
class BlockFinder
{
public:
    /** Who hands over a whole list: the caller's word (an imported index) always counts, a scanner's only while the list
     * is still open -- an index imported meanwhile, or a cut behind trailing garbage, stays (ADVICE r02, bz2_reader.cpp:505). */
    enum class Authority { CALLER, SCANNER };

    /** Answer to "where does block `number` start": the offset if the list has it; `listComplete` tells a missing block
     * of a finished list ("there is no such block") from one the scan has not reached yet. */
    struct Answer
    {
        std::optional<size_t> bits;
        bool listComplete{ false };
    };

    static constexpr double DO_NOT_WAIT = 0;
    static constexpr double UNTIL_KNOWN = std::numeric_limits<double>::infinity();

    BlockFinder( const uint8_t* bytes, uint64_t size, uint64_t magic48, size_t lookAhead, unsigned scanThreads ) :
        m_bytes( bytes ),
        m_size( size ),
        m_magic( magic48 ),
        m_lookAhead( lookAhead ),
        m_scanThreads( std::max( 1u, scanThreads ) )
    {}

    ~BlockFinder()
    {
        stopThreads();
    }

    /** Starts the scan thread unless the list is complete or a thread is at work. */
    void
    startThreads()
    {
        std::thread finished;
        {
            const std::lock_guard<std::mutex> hold( m_lock );
            finished = ensureScanning();
        }
        if ( finished.joinable() ) {
            finished.join();
        }
    }

    /** Stops the scan where it is (a later at() resumes it there). */
    void
    stopThreads()
    {
        std::thread scanner;
        {
            const std::lock_guard<std::mutex> hold( m_lock );
            m_pause = true;
            scanner = std::move( m_thread );
            m_wake.notify_all();
        }
        if ( scanner.joinable() ) {
            scanner.join();
        }
    }

    [[nodiscard]] size_t
    size() const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        return m_offsets.size();
    }

    [[nodiscard]] bool
    complete() const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        return m_complete;
    }

    /**
     * Offset of block `number`.  Raises the scan's target to `number` + look-ahead, (re)starts the scan if the list is
     * open, and waits up to `seconds` (DO_NOT_WAIT: look only; UNTIL_KNOWN: until the block is there or the list complete).
     */
    [[nodiscard]] Answer
    at( size_t number, double seconds = UNTIL_KNOWN )
    {
        std::vector<std::thread> finished;
        Answer answer;
        {
            std::unique_lock<std::mutex> hold( m_lock );
            if ( number > m_furthestAsked ) {
                m_furthestAsked = number;
                m_wake.notify_all();
            }
            /* system_clock: waits on it are pthread_cond_timedwait, which ThreadSanitizer understands (waits on the steady
             * clock go through pthread_cond_clockwait, which gcc 11's runtime does not intercept: false reports) */
            const auto deadline = std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::nanoseconds>(
                std::chrono::duration<double>( std::isinf( seconds ) ? 0. : std::max( 0., seconds ) ) );
            const auto settled = [&] { return m_complete || ( number < m_offsets.size() ); };
            for ( ;; ) {
                if ( auto left = ensureScanning(); left.joinable() ) {
                    finished.push_back( std::move( left ) );
                }
                /* a scan that was paused under a waiting reader is taken up again, hence `!m_scanning` */
                const auto awake = [&] { return settled() || !m_scanning; };
                if ( std::isinf( seconds ) ) {
                    m_wake.wait( hold, awake );
                } else if ( !settled() ) {
                    if ( !( seconds > 0 ) || !m_wake.wait_until( hold, deadline, awake ) ) {
                        break;
                    }
                }
                if ( settled() ) {
                    break;
                }
            }
            if ( number < m_offsets.size() ) {
                answer.bits = m_offsets[number];
            }
            answer.listComplete = m_complete;
        }
        for ( auto& thread : finished ) {
            thread.join();
        }
        return answer;
    }

    /** Number of the block that starts at `bits`; an offset that is not in the list is the caller's logic error. */
    [[nodiscard]] size_t
    numberOf( size_t bits ) const
    {
        const std::lock_guard<std::mutex> hold( m_lock );
        const auto [first, behind] = std::equal_range( m_offsets.begin(), m_offsets.end(), bits );
        if ( first == behind ) {
            throw std::out_of_range( "bit offset " + std::to_string( bits ) + " is not the start of a block in the finder's list of "
                                     + std::to_string( m_offsets.size() ) + " blocks" );
        }
        return (size_t)( first - m_offsets.begin() );
    }

    /** The whole list at once; it is complete from here on.  False if a scanner came too late (see Authority). */
    bool
    adopt( std::vector<size_t> offsets, Authority who )
    {
        std::thread scanner;
        {
            const std::lock_guard<std::mutex> hold( m_lock );
            if ( m_complete && ( who == Authority::SCANNER ) ) {
                return false;
            }
            m_offsets = std::move( offsets );
            scanner = closeList();
        }
        if ( scanner.joinable() ) {
            scanner.join();
        }
        return true;
    }

    /** The list ends after its first `count` blocks (what the scan saw behind them was no block), or where it is now. */
    void
    cut( std::optional<size_t> count = {} )
    {
        std::thread scanner;
        {
            const std::lock_guard<std::mutex> hold( m_lock );
            if ( count ) {
                if ( *count > m_offsets.size() ) {
                    throw std::invalid_argument( "the finder's list has " + std::to_string( m_offsets.size() )
                                                 + " blocks and cannot be cut to " + std::to_string( *count ) );
                }
                m_offsets.resize( *count );
            }
            scanner = closeList();
        }
        if ( scanner.joinable() ) {
            scanner.join();
        }
    }

private:
    /** With m_lock held: nothing a running scan finds may be appended any more. */
    [[nodiscard]] std::thread
    closeList()
    {
        m_complete = true;
        ++m_epoch;
        m_wake.notify_all();
        return std::move( m_thread );
    }

    /** With m_lock held.  Returns the handle of a thread that has left (paused earlier) for the caller to join. */
    [[nodiscard]] std::thread
    ensureScanning()
    {
        if ( m_complete || m_scanning ) {
            return {};
        }
        std::thread finished = std::move( m_thread );
        m_pause = false;
        m_scanning = true;
        m_thread = std::thread( [this, epoch = m_epoch] { scan( epoch ); } );
        return finished;
    }

    /** The scan thread: a chunk of the file per round (split over m_scanThreads), as long as the list is `m_lookAhead`
     * blocks or less ahead of the furthest block asked for. */
    void
    scan( const uint64_t epoch )
    {
        constexpr uint64_t CHUNK = 8U << 20U;
        std::unique_lock<std::mutex> hold( m_lock );
        const auto mine = [&] { return ( m_epoch == epoch ) && !m_pause; };
        while ( mine() ) {
            if ( m_scanPosition >= m_size ) {
                m_complete = true;
                break;
            }
            if ( m_offsets.size() > m_furthestAsked + m_lookAhead ) {
                m_wake.wait( hold );
                continue;
            }
            const uint64_t begin = m_scanPosition;
            const uint64_t end = std::min<uint64_t>( m_size, begin + CHUNK * m_scanThreads );
            hold.unlock();

            std::vector<std::vector<uint64_t> > parts( m_scanThreads );
            std::vector<std::thread> helpers;
            const uint64_t share = ( end - begin + m_scanThreads - 1 ) / m_scanThreads;
            for ( unsigned t = 0; t < m_scanThreads; ++t ) {
                const uint64_t from = std::min( end, begin + t * share );
                const uint64_t to = std::min( end, from + share );
                if ( t + 1 < m_scanThreads ) {
                    helpers.emplace_back( [this, from, to, &parts, t] { scanMagicRange( m_bytes, m_size, m_magic, from, to, parts[t] ); } );
                } else {
                    scanMagicRange( m_bytes, m_size, m_magic, from, to, parts[t] );
                }
            }
            for ( auto& helper : helpers ) {
                helper.join();
            }

            hold.lock();
            if ( m_epoch != epoch ) {
                break;      /* the list was handed over or cut meanwhile: these matches belong to nobody */
            }
            for ( const auto& part : parts ) {
                m_offsets.insert( m_offsets.end(), part.begin(), part.end() );
            }
            m_scanPosition = end;
            m_wake.notify_all();
        }
        m_scanning = false;
        m_wake.notify_all();
    }

    const uint8_t* const m_bytes;
    const uint64_t m_size;
    const uint64_t m_magic;
    const size_t m_lookAhead;
    const unsigned m_scanThreads;

    mutable std::mutex m_lock;
    std::condition_variable m_wake;
    std::vector<size_t> m_offsets;      /* sorted: appended in file order or handed over whole */
    size_t m_furthestAsked{ 0 };
    uint64_t m_scanPosition{ 0 };       /* first byte the scan has not covered */
    uint64_t m_epoch{ 0 };              /* advanced whenever the list changes hands */
    bool m_complete{ false };
    bool m_pause{ false };
    bool m_scanning{ false };           /* a scan thread is inside scan() */
    std::thread m_thread;
};
}  // namespace mi355x
