/* Seeded random operations on the host-side classes (block index, access pattern, run cache, block finder thread) for the
 * sanitizer builds of tests/test_host_sanitizers.py: asserts only invariants, the answers are pinned elsewhere
 * (tests/native/host_known_answers.cpp). */
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <vector>
#include "../../include/mi355x_bz2.h"
#include "../../indexed_bzip2_amd/csrc/bz2_host.hpp"
using namespace mi355x;
struct Run { size_t f, n; size_t first() const { return f; } size_t count() const { return n; } };
int main() {
    srand(3);
    for (int iter = 0; iter < 300; ++iter) {
        BlockIndex index;
        size_t enc = 32, dec = 0;
        const int n = 1 + rand() % 50;
        for (int i = 0; i < n; ++i) {
            const size_t es = 80 + rand() % 10000, ds = (rand() % 4 == 0) ? 0 : rand() % 100000;
            assert(index.append(enc, es, ds) == dec);
            enc += es; dec += ds;
            assert(index.frontier() == dec);
            for (int q = 0; q < 5; ++q) { const auto o = rand() % (dec + 10); const auto s = index.locate(o); assert(!s.covers(o) || (s.bytes <= o && o < s.bytes + s.byteLength)); }
        }
        index.seal();
        assert(index.sealed());
        const auto pairs = index.snapshot();
        BlockIndex other; other.assign(pairs);
        assert(other.snapshot() == pairs);
        for (int q = 0; q < 50; ++q) { (void)other.locate(rand() % (dec + 100)); }
        (void)other.last(); (void)other.dataBlocks();
    }
    RunCache<Run> cache(64);
    for (int i = 0; i < 10000; ++i) {
        const size_t k = rand() % 400;
        switch (rand() % 5) {
        case 0: if (!cache.covers(k)) { size_t n = 1 + rand() % 16; while (n > 1 && cache.firstGap(k) != k + 0 && false) --n; size_t m = 0; while (m < n && !cache.covers(k + m)) ++m; cache.insert(std::make_shared<Run>(Run{k, m})); } break;
        case 1: (void)cache.find(k); break;
        case 2: (void)cache.firstGap(k); (void)cache.blocksWithin(k, k + 50); break;
        case 3: cache.dropBefore(k); break;
        default: (void)cache.covers(k); break;
        }
        assert(cache.blocks() <= cache.budget() || cache.runs() == 1);
    }
    cache.clear();
    SequentialityTracker pattern;
    size_t idx = 0;
    for (int i = 0; i < 5000; ++i) {
        if (rand() % 10 == 0) idx = rand() % 1000; else ++idx;
        pattern.note(idx);
        const size_t limit = 1 + rand() % 64;
        const auto r = pattern.ahead(limit);
        assert(r.count <= limit && (r.count == 0 || r.first == idx + 1));
        (void)pattern.inOrder();
    }
    // BlockFinder over a buffer with magics, with and without the thread
    std::vector<uint8_t> buf(3 << 20);
    for (auto& b : buf) b = rand();
    const uint8_t M[6] = {0x31,0x41,0x59,0x26,0x53,0x59};
    std::vector<size_t> want;
    for (size_t p = 4; p + 6 < buf.size(); p += 100000 + rand() % 50000) { memcpy(buf.data() + p, M, 6); want.push_back(p * 8); }
    {
        BlockFinder finder(buf.data(), buf.size(), MI355X_BZ2_MAGIC_BLOCK, 8, 2);
        finder.startThreads();
        for (size_t i = 0; i < want.size(); ++i) { const auto a = finder.at(i); assert(a.bits && *a.bits == want[i]); }
        const auto none = finder.at(want.size());
        assert(!none.bits && none.listComplete);
        assert(finder.numberOf(want[3]) == 3);
        bool threw = false;
        try { (void)finder.numberOf(want[3] + 1); } catch (const std::out_of_range&) { threw = true; }
        assert(threw);
    }
    {
        BlockFinder finder(buf.data(), buf.size(), MI355X_BZ2_MAGIC_BLOCK, 2, 1);
        (void)finder.at(1);
        const auto early = finder.at(1000, BlockFinder::DO_NOT_WAIT);
        assert(!early.bits);
        finder.cut(2);
        assert(finder.size() == 2 && finder.complete());
        assert(!finder.adopt({7, 8, 9, 10}, BlockFinder::Authority::SCANNER));      // too late: the cut list stays
        assert(finder.size() == 2);
        assert(finder.adopt({1, 2, 3}, BlockFinder::Authority::CALLER));
        assert(finder.size() == 3 && finder.numberOf(3) == 2);
        bool threw = false;
        try { finder.cut(4); } catch (const std::invalid_argument&) { threw = true; }
        assert(threw);
    }
    {   // paused and resumed
        BlockFinder finder(buf.data(), buf.size(), MI355X_BZ2_MAGIC_BLOCK, 1, 1);
        (void)finder.at(0);
        finder.stopThreads();
        const auto a = finder.at(want.size() - 1);
        assert(a.bits && *a.bits == want.back());
    }
    std::printf("host ok\n");
    return 0;
}
