/**
 * bz2_host.hpp -- host-side scheduler pieces kept from the reference's architecture, restated
 * batch-oriented for a GPU backend.  Product code (never touches oracle/).
 *
 *   BlockMap            <- rapidgzip::BlockMap                       src/core/BlockMap.hpp:26-295
 *   LruCache            <- rapidgzip::Cache + LeastRecentlyUsed      src/core/Cache.hpp:47-296
 *   FetchNextAdaptive   <- FetchingStrategy::FetchNextAdaptive       src/core/Prefetcher.hpp:82-217
 *   BlockFinder         <- rapidgzip::BlockFinder + StreamedResults  src/core/BlockFinder.hpp:36-219,
 *                                                                    src/core/StreamedResults.hpp:26-156
 */
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <limits>
#include <list>
#include <map>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

namespace mi355x
{
/** Scan byte range [begin,end) (first byte of the match) of bytes[0,size) for a 48-bit pattern; bz2_finder.cpp. */
void
scanMagicRange( const uint8_t* bytes, uint64_t size, uint64_t magic48, uint64_t begin, uint64_t end,
                std::vector<uint64_t>& found );

/* ------------------------------------------------------------------------------------------------ BlockMap */
class BlockMap
{
public:
    struct BlockInfo
    {
        [[nodiscard]] bool
        contains( size_t dataOffset ) const
        {
            return ( decodedOffsetInBytes <= dataOffset ) && ( dataOffset < decodedOffsetInBytes + decodedSizeInBytes );
        }

        size_t blockIndex{ 0 };
        size_t encodedOffsetInBits{ 0 };
        size_t encodedSizeInBits{ 0 };
        size_t decodedOffsetInBytes{ 0 };
        size_t decodedSizeInBytes{ 0 };
    };

    using Offsets = std::vector<std::pair<size_t, size_t> >;

    /** BlockMap::push, BlockMap.hpp:69-119 */
    size_t
    push( size_t encodedBlockOffset, size_t encodedSize, size_t decodedSize )
    {
        const std::scoped_lock lock( m_mutex );
        if ( m_finalized ) {
            throw std::invalid_argument( "May not insert into finalized block map!" );
        }
        std::optional<size_t> decodedOffset;
        if ( m_offsets.empty() ) {
            decodedOffset = 0;
        } else if ( encodedBlockOffset > m_offsets.back().first ) {
            decodedOffset = m_offsets.back().second + m_lastDecodedSize;
        }
        if ( decodedOffset ) {
            m_offsets.emplace_back( encodedBlockOffset, *decodedOffset );
            if ( decodedSize == 0 ) {
                m_eosBlocks.push_back( encodedBlockOffset );
            }
            m_lastDecodedSize = decodedSize;
            m_lastEncodedSize = encodedSize;
            return *decodedOffset;
        }
        const auto match = std::lower_bound( m_offsets.begin(), m_offsets.end(),
                                             std::make_pair( encodedBlockOffset, size_t( 0 ) ),
                                             [] ( const auto& a, const auto& b ) { return a.first < b.first; } );
        if ( ( match == m_offsets.end() ) || ( match->first != encodedBlockOffset ) ) {
            throw std::invalid_argument( "Inserted block offsets should be strictly increasing!" );
        }
        if ( std::next( match ) == m_offsets.end() ) {
            throw std::logic_error( "In this case, the new block should already have been appended above!" );
        }
        if ( std::next( match )->second - match->second != decodedSize ) {
            throw std::invalid_argument( "Got duplicate block offset with inconsistent size!" );
        }
        return match->second;
    }

    /** BlockMap::findDataOffset, BlockMap.hpp:125-144: last block whose decoded offset is <= dataOffset */
    [[nodiscard]] BlockInfo
    findDataOffset( size_t dataOffset ) const
    {
        const std::scoped_lock lock( m_mutex );
        /* entries have non-decreasing .second; find the LAST entry with second <= dataOffset */
        auto it = std::upper_bound( m_offsets.begin(), m_offsets.end(), dataOffset,
                                    [] ( size_t value, const auto& entry ) { return value < entry.second; } );
        if ( it == m_offsets.begin() ) {
            return {};
        }
        --it;
        return get( (size_t)std::distance( m_offsets.begin(), it ) );
    }

    [[nodiscard]] size_t
    dataBlockCount() const
    {
        const std::scoped_lock lock( m_mutex );
        return m_offsets.size() - m_eosBlocks.size();
    }

    /** BlockMap::finalize, BlockMap.hpp:171-197 */
    void
    finalize()
    {
        const std::scoped_lock lock( m_mutex );
        if ( m_finalized ) {
            return;
        }
        if ( m_offsets.empty() ) {
            m_offsets.emplace_back( m_lastEncodedSize, m_lastDecodedSize );
        } else if ( ( m_lastEncodedSize != 0 ) || ( m_lastDecodedSize != 0 ) ) {
            const auto [lastEncoded, lastDecoded] = m_offsets.back();
            m_offsets.emplace_back( lastEncoded + m_lastEncodedSize, lastDecoded + m_lastDecodedSize );
        }
        m_lastEncodedSize = 0;
        m_lastDecodedSize = 0;
        m_finalized = true;
    }

    [[nodiscard]] bool
    finalized() const
    {
        const std::scoped_lock lock( m_mutex );
        return m_finalized;
    }

    /** BlockMap::setBlockOffsets, BlockMap.hpp:207-230 */
    void
    setBlockOffsets( const std::map<size_t, size_t>& blockOffsets )
    {
        const std::scoped_lock lock( m_mutex );
        m_offsets.assign( blockOffsets.begin(), blockOffsets.end() );
        m_lastEncodedSize = 0;
        m_lastDecodedSize = 0;
        m_eosBlocks.clear();
        for ( size_t i = 0; i + 1 < m_offsets.size(); ++i ) {
            if ( m_offsets[i].second == m_offsets[i + 1].second ) {
                m_eosBlocks.push_back( m_offsets[i].first );
            }
        }
        m_eosBlocks.push_back( m_offsets.back().first );
        m_finalized = true;
    }

    [[nodiscard]] std::map<size_t, size_t>
    blockOffsets() const
    {
        const std::scoped_lock lock( m_mutex );
        return { m_offsets.begin(), m_offsets.end() };
    }

    [[nodiscard]] std::pair<size_t, size_t>
    back() const
    {
        const std::scoped_lock lock( m_mutex );
        if ( m_offsets.empty() ) {
            throw std::out_of_range( "Can not return last element of empty block map!" );
        }
        return m_offsets.back();
    }

    [[nodiscard]] bool
    empty() const
    {
        const std::scoped_lock lock( m_mutex );
        return m_offsets.empty();
    }

private:
    [[nodiscard]] BlockInfo
    get( size_t i ) const
    {
        BlockInfo result;
        result.encodedOffsetInBits = m_offsets[i].first;
        result.decodedOffsetInBytes = m_offsets[i].second;
        result.blockIndex = i;
        if ( i + 1 == m_offsets.size() ) {
            result.decodedSizeInBytes = m_lastDecodedSize;
            result.encodedSizeInBits = m_lastEncodedSize;
        } else {
            if ( m_offsets[i + 1].second < m_offsets[i].second ) {
                throw std::logic_error( "Data offsets are not monotonically increasing!" );
            }
            result.decodedSizeInBytes = m_offsets[i + 1].second - m_offsets[i].second;
            result.encodedSizeInBits = m_offsets[i + 1].first - m_offsets[i].first;
        }
        return result;
    }

    mutable std::mutex m_mutex;
    Offsets m_offsets;
    std::vector<size_t> m_eosBlocks;
    bool m_finalized{ false };
    size_t m_lastEncodedSize{ 0 };
    size_t m_lastDecodedSize{ 0 };
};

/* ------------------------------------------------------------------------------------------------ LRU cache */
template<typename Key, typename Value>
class LruCache
{
public:
    struct Statistics
    {
        size_t hits{ 0 };
        size_t misses{ 0 };
        size_t unusedEntries{ 0 };
        size_t capacity{ 0 };
        size_t maxSize{ 0 };
    };

    explicit
    LruCache( size_t capacity ) :
        m_capacity( capacity )
    {}

    [[nodiscard]] std::optional<Value>
    get( const Key& key )
    {
        const auto match = m_entries.find( key );
        if ( match == m_entries.end() ) {
            ++m_statistics.misses;
            return std::nullopt;
        }
        ++m_statistics.hits;
        ++match->second.accesses;
        touchEntry( match );
        return match->second.value;
    }

    void
    insert( Key key, Value value )
    {
        if ( m_capacity == 0 ) {
            return;
        }
        auto match = m_entries.find( key );
        if ( match == m_entries.end() ) {
            shrinkTo( m_capacity - 1 );
            m_order.push_back( key );
            Entry entry{ std::move( value ), std::prev( m_order.end() ), 0 };
            m_entries.emplace( std::move( key ), std::move( entry ) );
            m_statistics.maxSize = std::max( m_statistics.maxSize, m_entries.size() );
        } else {
            match->second.value = std::move( value );
            touchEntry( match );
        }
    }

    void
    touch( const Key& key )
    {
        const auto match = m_entries.find( key );
        if ( match != m_entries.end() ) {
            touchEntry( match );
        }
    }

    [[nodiscard]] bool
    test( const Key& key ) const
    {
        return m_entries.find( key ) != m_entries.end();
    }

    void
    clear()
    {
        m_entries.clear();
        m_order.clear();
    }

    void
    evict( const Key& key )
    {
        const auto match = m_entries.find( key );
        if ( match != m_entries.end() ) {
            m_order.erase( match->second.position );
            m_entries.erase( match );
        }
    }

    /** Key that would be evicted by the n-th hypothetical insertion (Cache::nextNthEviction, Cache.hpp:217-224). */
    [[nodiscard]] std::optional<Key>
    nextNthEviction( size_t countToBeInserted ) const
    {
        const auto freeCapacity = m_capacity - m_entries.size();
        if ( countToBeInserted <= freeCapacity ) {
            return std::nullopt;
        }
        const auto n = countToBeInserted - freeCapacity;
        if ( ( n == 0 ) || ( n > m_order.size() ) ) {
            return std::nullopt;
        }
        return *std::next( m_order.begin(), (std::ptrdiff_t)( n - 1 ) );
    }

    void
    shrinkTo( size_t newSize )
    {
        while ( m_entries.size() > newSize ) {
            const auto key = m_order.front();
            const auto match = m_entries.find( key );
            if ( match->second.accesses == 0 ) {
                ++m_statistics.unusedEntries;
            }
            m_entries.erase( match );
            m_order.pop_front();
        }
    }

    [[nodiscard]] Statistics
    statistics() const
    {
        auto result = m_statistics;
        result.capacity = m_capacity;
        return result;
    }

    [[nodiscard]] size_t
    capacity() const
    {
        return m_capacity;
    }

    [[nodiscard]] size_t
    size() const
    {
        return m_entries.size();
    }

private:
    struct Entry
    {
        Value value;
        typename std::list<Key>::iterator position;
        size_t accesses{ 0 };
    };

    void
    touchEntry( typename std::unordered_map<Key, Entry>::iterator match )
    {
        m_order.erase( match->second.position );
        m_order.push_back( match->first );
        match->second.position = std::prev( m_order.end() );
    }

    size_t m_capacity;
    std::unordered_map<Key, Entry> m_entries;
    std::list<Key> m_order;   /* front = least recently used */
    Statistics m_statistics;
};

/* ------------------------------------------------------------------------------------------------ prefetch strategy */
class FetchNextAdaptive
{
public:
    explicit
    FetchNextAdaptive( size_t memorySize = 3 ) :
        m_memorySize( memorySize )
    {}

    /** Prefetcher.hpp:91-104 */
    void
    fetch( size_t index )
    {
        if ( !m_previousIndexes.empty() && ( m_previousIndexes.front() == index ) ) {
            return;
        }
        m_previousIndexes.push_front( index );
        while ( m_previousIndexes.size() > m_memorySize ) {
            m_previousIndexes.pop_back();
        }
    }

    /** Prefetcher.hpp:106-116 */
    [[nodiscard]] bool
    isSequential() const noexcept
    {
        for ( size_t i = 0; i + 1 < m_previousIndexes.size(); ++i ) {
            if ( m_previousIndexes[i + 1] + 1 != m_previousIndexes[i] ) {
                return false;
            }
        }
        return true;
    }

    /** Prefetcher.hpp:118-183: full amount when sequential, nothing when random, exponential in between. */
    [[nodiscard]] std::vector<size_t>
    prefetch( size_t maxAmountToPrefetch ) const
    {
        const auto size = m_previousIndexes.size();
        if ( ( size == 0 ) || ( maxAmountToPrefetch == 0 ) ) {
            return {};
        }
        const auto iotaFrom = [] ( size_t first, size_t count ) {
            std::vector<size_t> result( count );
            for ( size_t i = 0; i < count; ++i ) {
                result[i] = first + i;
            }
            return result;
        };
        if ( size == 1 ) {
            return iotaFrom( m_previousIndexes.front() + 1, maxAmountToPrefetch );
        }
        size_t adjacent = 0;
        for ( size_t i = 0; i + 1 < size; ++i ) {
            if ( m_previousIndexes[i] == m_previousIndexes[i + 1] + 1 ) {
                ++adjacent;
            }
        }
        if ( adjacent == 0 ) {
            return {};
        }
        size_t lastConsecutiveCount = 0;
        for ( size_t i = 0; i + 1 < size; ++i ) {
            if ( m_previousIndexes[i] == m_previousIndexes[i + 1] + 1 ) {
                lastConsecutiveCount = lastConsecutiveCount == 0 ? 2 : lastConsecutiveCount + 1;
            } else {
                break;
            }
        }
        const auto consecutiveRatio = static_cast<double>( std::min( lastConsecutiveCount, size ) )
                                      / static_cast<double>( size );
        const auto amount = std::round( std::exp2( consecutiveRatio * std::log2( (double)maxAmountToPrefetch ) ) );
        return iotaFrom( m_previousIndexes.front() + 1, static_cast<size_t>( std::max( 0.0, amount ) ) );
    }

private:
    const size_t m_memorySize;
    std::deque<size_t> m_previousIndexes;   /* most recent at the front */
};

/* ------------------------------------------------------------------------------------------------ block finder */
class BlockFinder
{
public:
    enum class GetReturnCode { SUCCESS, TIMEOUT, FAILURE };

    BlockFinder( const uint8_t* bytes, uint64_t size, uint64_t magic48, size_t prefetchCount, unsigned scanThreads ) :
        m_bytes( bytes ),
        m_size( size ),
        m_magic( magic48 ),
        m_prefetchCount( prefetchCount ),
        m_scanThreads( std::max( 1u, scanThreads ) )
    {}

    ~BlockFinder()
    {
        stopThreads();
    }

    void
    startThreads()
    {
        const std::scoped_lock lock( m_threadMutex );
        if ( !m_thread.joinable() && !m_finalized ) {
            m_cancel = false;
            m_thread = std::thread( [this] () { finderMain(); } );
        }
    }

    void
    stopThreads()
    {
        {
            const std::scoped_lock lock( m_mutex );
            m_cancel = true;
            m_changed.notify_all();
        }
        const std::scoped_lock lock( m_threadMutex );
        if ( m_thread.joinable() ) {
            m_thread.join();
        }
    }

    [[nodiscard]] size_t
    size() const
    {
        const std::scoped_lock lock( m_mutex );
        return m_offsets.size();
    }

    /** BlockFinder::finalize, BlockFinder.hpp:91-97 */
    void
    finalize( std::optional<size_t> blockCount = {} )
    {
        stopThreads();
        const std::scoped_lock lock( m_mutex );
        if ( blockCount ) {
            if ( *blockCount > m_offsets.size() ) {
                throw std::invalid_argument( "You may not finalize to a size larger than the current results buffer!" );
            }
            m_offsets.resize( *blockCount );
        }
        m_finalized = true;
        m_changed.notify_all();
    }

    [[nodiscard]] bool
    finalized() const
    {
        return m_finalized;
    }

    /** BlockFinder::get, BlockFinder.hpp:112-131 + StreamedResults::get, StreamedResults.hpp:73-94 */
    [[nodiscard]] std::pair<std::optional<size_t>, GetReturnCode>
    get( size_t blockNumber, double timeoutInSeconds = std::numeric_limits<double>::infinity() )
    {
        if ( !m_finalized ) {
            startThreads();
        }
        std::unique_lock lock( m_mutex );
        m_highestRequested = std::max( m_highestRequested, blockNumber );
        m_changed.notify_all();
        if ( timeoutInSeconds > 0 ) {
            const auto predicate = [&] () { return m_finalized.load() || ( blockNumber < m_offsets.size() ); };
            if ( std::isfinite( timeoutInSeconds ) ) {
                m_changed.wait_for( lock, std::chrono::nanoseconds( (int64_t)( timeoutInSeconds * 1e9 ) ), predicate );
            } else {
                m_changed.wait( lock, predicate );
            }
        }
        if ( blockNumber < m_offsets.size() ) {
            return { m_offsets[blockNumber], GetReturnCode::SUCCESS };
        }
        return { std::nullopt, m_finalized ? GetReturnCode::FAILURE : GetReturnCode::TIMEOUT };
    }

    /** BlockFinder::find, BlockFinder.hpp:134-150 */
    [[nodiscard]] size_t
    find( size_t encodedBlockOffsetInBits ) const
    {
        const std::scoped_lock lock( m_mutex );
        const auto match = std::lower_bound( m_offsets.begin(), m_offsets.end(), encodedBlockOffsetInBits );
        if ( ( match == m_offsets.end() ) || ( *match != encodedBlockOffsetInBits ) ) {
            throw std::out_of_range( "No block with the specified offset exists in the block finder map!" );
        }
        return (size_t)std::distance( m_offsets.begin(), match );
    }

    /** BlockFinder::setBlockOffsets, BlockFinder.hpp:152-161 */
    void
    setBlockOffsets( std::deque<size_t> offsets )
    {
        stopThreads();
        const std::scoped_lock lock( m_mutex );
        m_offsets = std::move( offsets );
        m_finalized = true;
        m_changed.notify_all();
    }

private:
    /** BlockFinder::blockFinderMain, BlockFinder.hpp:164-197, scanning a chunk (not one match) per iteration. */
    void
    finderMain()
    {
        constexpr uint64_t CHUNK = 8u << 20;
        uint64_t position = 0;
        {
            const std::scoped_lock lock( m_mutex );
            position = m_scanPosition;
        }
        while ( true ) {
            {
                std::unique_lock lock( m_mutex );
                m_changed.wait( lock, [this] {
                    return m_cancel || ( m_offsets.size() <= m_highestRequested + m_prefetchCount );
                } );
                if ( m_cancel ) {
                    m_scanPosition = position;
                    return;
                }
            }
            if ( position >= m_size ) {
                break;
            }
            const uint64_t end = std::min<uint64_t>( m_size, position + CHUNK * m_scanThreads );
            std::vector<std::vector<uint64_t> > parts( m_scanThreads );
            std::vector<std::thread> pool;
            const uint64_t per = ( end - position + m_scanThreads - 1 ) / m_scanThreads;
            for ( unsigned t = 0; t < m_scanThreads; ++t ) {
                const uint64_t b = std::min( end, position + t * per );
                const uint64_t e = std::min( end, b + per );
                if ( t + 1 == m_scanThreads ) {
                    scanMagicRange( m_bytes, m_size, m_magic, b, e, parts[t] );
                } else {
                    pool.emplace_back( [this, b, e, &parts, t] () {
                        scanMagicRange( m_bytes, m_size, m_magic, b, e, parts[t] );
                    } );
                }
            }
            for ( auto& th : pool ) {
                th.join();
            }
            position = end;
            {
                const std::scoped_lock lock( m_mutex );
                for ( const auto& part : parts ) {
                    for ( const auto offset : part ) {
                        m_offsets.push_back( offset );
                    }
                }
                m_changed.notify_all();
            }
        }
        const std::scoped_lock lock( m_mutex );
        m_scanPosition = position;
        m_finalized = true;
        m_changed.notify_all();
    }

    const uint8_t* const m_bytes;
    const uint64_t m_size;
    const uint64_t m_magic;
    const size_t m_prefetchCount;
    const unsigned m_scanThreads;

    mutable std::mutex m_mutex;
    std::condition_variable m_changed;
    std::deque<size_t> m_offsets;
    size_t m_highestRequested{ 0 };
    uint64_t m_scanPosition{ 0 };
    std::atomic<bool> m_finalized{ false };
    bool m_cancel{ false };

    std::mutex m_threadMutex;
    std::thread m_thread;
};
}  // namespace mi355x
