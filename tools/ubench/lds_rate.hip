// Developer microbenchmark: LDS b128 read/write throughput per CU as a function of the number of active lanes (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template<int MODE>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint64_t laneMask, int iters, uint32_t stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[256 * 272];
    const uint32_t t = threadIdx.x;
    uint4* mine = reinterpret_cast<uint4*>(lds + t * stride);
    for (int k = 0; k < 16; ++k) mine[k] = make_uint4(t, k, 0, 0);
    __syncthreads();
    const bool on = (laneMask >> (t & 63)) & 1;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint64_t t0 = __builtin_readcyclecounter();
    if (on) {
        for (int i = 0; i < iters; ++i) {
            if constexpr (MODE == 0) {          // 16 independent b128 reads
#pragma unroll
                for (int k = 0; k < 16; ++k) { uint4 v = mine[k]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
                asm volatile("" ::: "memory");
            } else if constexpr (MODE == 1) {   // 16 b128 writes
#pragma unroll
                for (int k = 0; k < 16; ++k) mine[k] = make_uint4(i, k, acc.x, t);
                asm volatile("" ::: "memory");
            } else if constexpr (MODE == 2) {   // 16 b32 reads
                const uint32_t* m32 = reinterpret_cast<const uint32_t*>(mine);
#pragma unroll
                for (int k = 0; k < 16; ++k) acc.x ^= m32[k];
                asm volatile("" ::: "memory");
            }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (t == 0) out[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u && acc.y == 77) out[1000 + t] = acc.z + acc.w;
}

template<int MODE>
void run(const char* name, uint64_t mask, uint32_t stride)
{
    uint64_t* d; (void)hipMalloc(&d, 1 << 20);
    const int iters = 2000, blocks = 512;   // 2 workgroups per CU (68 KB LDS each)
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, mask, iters, stride);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, mask, iters, stride);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    // per CU: 2 WGs x 4 waves x iters x 16 instructions
    const double instrPerCu = 2.0 * 4 * iters * 16;
    printf("%-14s lanes=%2d stride=%3u  %.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, __builtin_popcountll(mask), stride,
           ms * 1e-3 * 2.4e9 / instrPerCu);
    (void)hipFree(d);
}

int main()
{
    const uint64_t masks[] = { ~0ull, 0xFFFFFFFFull, 0xFFFFull, 0xFFull, 0x1ull, 0x0101010101010101ull, 0x1111111111111111ull };
    for (uint32_t stride : { 272u, 256u }) {
        for (uint64_t m : masks) {
            run<0>("read b128", m, stride);
            run<1>("write b128", m, stride);
            run<2>("read b32", m, stride);
        }
    }
    return 0;
}
