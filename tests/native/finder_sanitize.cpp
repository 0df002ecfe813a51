#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/mi355x_bz2.h"
int main() {
    srand(1);
    const uint8_t M[6] = {0x31,0x41,0x59,0x26,0x53,0x59};
    uint64_t total = 0;
    for (int iter = 0; iter < 20000; ++iter) {
        const size_t n = rand() % 200;
        std::vector<uint8_t> buf(n);   // exact size: ASAN catches any overread
        for (auto& b : buf) b = rand();
        if (n >= 6 && rand() % 2) { size_t p = rand() % (n - 5); memcpy(buf.data() + p, M, 6); }
        if (n >= 7 && rand() % 3 == 0) {   // bit-shifted copy near the end
            const int s = 1 + rand() % 7; size_t p = n - 7;
            uint64_t v = 0; for (int i = 0; i < 6; ++i) v = (v << 8) | M[i];
            v <<= (8 - s);
            for (int i = 0; i < 7; ++i) buf[p + i] = (uint8_t)(v >> (8 * (6 - i)));
        }
        std::vector<uint64_t> out(64);
        const unsigned threads = 1 + rand() % 3;
        uint64_t c = mi355x_bz2_find_magic(buf.data(), n, MI355X_BZ2_MAGIC_BLOCK, out.data(), out.size(), threads);
        total += c;
        (void)mi355x_bz2_read_stream_header(buf.data(), n, (rand() % (n + 2)) * 8);
    }
    // large buffer with threads
    std::vector<uint8_t> big(9 << 20);
    for (auto& b : big) b = rand();
    for (size_t p = 100; p + 6 < big.size(); p += 1 << 20) memcpy(big.data() + p, M, 6);
    std::vector<uint64_t> out(64);
    total += mi355x_bz2_find_magic(big.data(), big.size(), MI355X_BZ2_MAGIC_BLOCK, out.data(), out.size(), 4);
    printf("matches %llu\n", (unsigned long long)total);
    return 0;
}
