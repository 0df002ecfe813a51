/* Two (and three) threads against one BlockFinder, the interleavings the reader performs:
 * the reading thread asks for block offsets -- which (re)starts the host scan whenever the list is open -- while another
 * thread hands over the complete list at a random moment (the GPU magic scan, bz2_reader.cpp scanOnDevice), and the reading
 * thread itself may import an index or cut the list (trailing garbage) at any time.  After every round: the list is
 * complete, has exactly the expected blocks, is strictly increasing, and every offset is found under its number.
 * Built under ThreadSanitizer and ASan by tests/test_host_sanitizers.py.  Reference analogue: the CI's TSan runs over the
 * threaded classes (.github/workflows/test-cpp.yml:224-436); the classes are src/core/BlockFinder.hpp, StreamedResults.hpp. */
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "../../include/mi355x_bz2.h"
#include "../../indexed_bzip2_amd/csrc/bz2_host.hpp"
using namespace mi355x;

static void
check(const BlockFinder& finder, const std::vector<size_t>& want, const char* what, int round)
{
    if (!finder.complete() || finder.size() != want.size()) {
        std::fprintf(stderr, "%s, round %d: %zu blocks in the finder, %zu expected, complete %d\n", what, round, finder.size(), want.size(), (int)finder.complete());
        std::abort();
    }
    for (size_t i = 0; i < want.size(); ++i) {
        if (finder.numberOf(want[i]) != i) { std::fprintf(stderr, "%s, round %d: offset %zu is not block %zu\n", what, round, want[i], i); std::abort(); }
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 300;
    std::mt19937_64 rng(0xF1DE);
    std::vector<uint8_t> buf(48u << 20);
    { uint64_t x = 88172645463325252ull; for (size_t i = 0; i + 8 <= buf.size(); i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::memcpy(buf.data() + i, &x, 8); } }
    const uint8_t M[6] = {0x31,0x41,0x59,0x26,0x53,0x59};
    std::vector<size_t> want;
    for (size_t p = 4; p + 6 < buf.size(); p += 150000 + rng() % 100000) { std::memcpy(buf.data() + p, M, 6); want.push_back(p * 8); }

    for (int round = 0; round < rounds; ++round) {
        /* 1. get() loop against a scanner's hand-over at a random moment */
        {
            BlockFinder finder(buf.data(), buf.size(), MI355X_BZ2_MAGIC_BLOCK,
                               /* the reader's look-ahead (4 batches) exceeds most files' block counts: the scan never rests */
                               ( round % 4 == 3 ) ? 4 + rng() % 64 : 2048, 1 + rng() % 3);
            std::atomic<bool> stop{false};
            const auto ask = [&] {
                for (size_t i = 0; !stop; ++i) {
                    const auto a = finder.at(i % want.size(), ( ( round % 8 == 7 ) && ( i % 3 == 0 ) ) ? 0.0002 : BlockFinder::DO_NOT_WAIT);
                    if (a.bits && *a.bits != want[i % want.size()]) { std::fprintf(stderr, "round %d: block %zu at %zu, expected %zu\n", round, i % want.size(), *a.bits, want[i % want.size()]); std::abort(); }
                }
            };
            std::thread reader(ask), second(ask), third(ask);
            std::this_thread::sleep_for(std::chrono::microseconds(rng() % 1500));
            const bool took = finder.adopt(want, BlockFinder::Authority::SCANNER);
            (void)took;   /* refused only if the host scan got to the end of the buffer first: the same list then */
            std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));
            stop = true;
            reader.join(); second.join(); third.join();
            /* a scan thread that was wrongly started again needs a chunk's time (8 MB per thread) before its matches arrive */
            std::this_thread::sleep_for(std::chrono::milliseconds(15));
            check(finder, want, "hand-over", round);
        }
        /* 2. the caller imports an index while a scanner hands over the full list: the caller's stays, whatever the order */
        {
            BlockFinder finder(buf.data(), buf.size(), MI355X_BZ2_MAGIC_BLOCK, 8, 2);
            std::vector<size_t> imported(want.begin(), want.begin() + want.size() / 2);
            std::thread scanner([&] {
                std::this_thread::sleep_for(std::chrono::microseconds(rng() % 400));
                (void)finder.adopt(want, BlockFinder::Authority::SCANNER);
            });
            (void)finder.at(3, 0.0001);
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            assert(finder.adopt(imported, BlockFinder::Authority::CALLER));
            scanner.join();
            check(finder, imported, "import", round);
        }
        /* 3. the reading thread cuts the list (trailing garbage) while the scanner hands over: never longer than the cut
         *    if the cut came last, and a late scanner is refused */
        {
            BlockFinder finder(buf.data(), buf.size(), MI355X_BZ2_MAGIC_BLOCK, 8, 1);
            const auto first = finder.at(5);
            assert(first.bits && *first.bits == want[5]);
            std::thread scanner([&] {
                std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));
                (void)finder.adopt(want, BlockFinder::Authority::SCANNER);
            });
            std::this_thread::sleep_for(std::chrono::microseconds(150));
            finder.cut(4);
            scanner.join();
            check(finder, std::vector<size_t>(want.begin(), want.begin() + 4), "cut", round);
        }
    }
    std::printf("finder race ok: %d rounds, %zu blocks\n", rounds, want.size());
    return 0;
}
