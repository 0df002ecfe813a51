#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include "../../include/mi355x_bz2.h"
#include "../../indexed_bzip2_amd/csrc/bz2_host.hpp"
using namespace mi355x;
int main() {
    srand(3);
    for (int iter = 0; iter < 300; ++iter) {
        BlockMap map;
        std::vector<size_t> offs{0};
        size_t enc = 32, dec = 0;
        const int n = 1 + rand() % 50;
        for (int i = 0; i < n; ++i) {
            const size_t es = 80 + rand() % 10000, ds = (rand() % 4 == 0) ? 0 : rand() % 100000;
            map.push(enc, es, ds);
            enc += es; dec += ds;
            for (int q = 0; q < 5; ++q) { auto info = map.findDataOffset(rand() % (dec + 10)); (void)info.contains(0); }
        }
        map.finalize();
        assert(map.finalized());
        const auto m = map.blockOffsets();
        BlockMap other; other.setBlockOffsets(m);
        assert(other.blockOffsets() == m);
        for (int q = 0; q < 50; ++q) { (void)other.findDataOffset(rand() % (dec + 100)); }
        (void)other.back(); (void)other.dataBlockCount();
    }
    LruCache<size_t, int> cache(8);
    for (int i = 0; i < 10000; ++i) {
        const size_t k = rand() % 40;
        switch (rand() % 5) {
        case 0: cache.insert(k, i); break;
        case 1: (void)cache.get(k); break;
        case 2: cache.touch(k); break;
        case 3: cache.evict(k); break;
        default: (void)cache.test(k); (void)cache.nextNthEviction(1 + rand() % 3); break;
        }
        assert(cache.size() <= cache.capacity());
    }
    cache.shrinkTo(2); cache.clear();
    FetchNextAdaptive strat;
    size_t idx = 0;
    for (int i = 0; i < 5000; ++i) {
        if (rand() % 10 == 0) idx = rand() % 1000; else ++idx;
        strat.fetch(idx);
        const auto p = strat.prefetch(1 + rand() % 64);
        (void)p; (void)strat.isSequential();
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
        size_t i = 0;
        for (;; ++i) { const auto [o, code] = finder.get(i); if (!o) break; assert(i < want.size() + 5); }
        assert(finder.finalized());
        finder.stopThreads();
        printf("finder found %zu (planted %zu)\n", finder.size(), want.size());
    }
    printf("host ok\n");
    return 0;
}
