// Developer microbenchmark: what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for known byte counts, per access
// shape.  Kernels, each launched once (names are what the PMC summary is keyed by):
//   calib_stream16   reads N bytes, 16 B per lane, coalesced                      (the shape the guide's x2 rule is for)
//   calib_stream4    reads N bytes, 4 B per lane, coalesced
//   calib_gather4    G random 4-byte gathers from a table far larger than L2 + Infinity Cache (every gather a miss)
//   calib_gather4_l2 G random 4-byte gathers from a 2 MiB table (hits after the first touch)
//   calib_write16    writes N bytes, 16 B per lane
// The host prints the byte counts to compare with: tools/fetch_calib.sh joins both.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_stream16(const uint4* __restrict__ in, uint64_t n16, uint32_t* out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_stream4(const uint32_t* __restrict__ in, uint64_t n4, uint32_t* out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) acc ^= in[i];
    if (acc == 0x9E3779B9u) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_gather4(const uint32_t* __restrict__ table, uint64_t mask, uint32_t perThread, uint32_t* out)
{
    // independent pseudo-random indices per lane (xorshift), so that the 64 lanes of a wave touch 64 different lines
    uint64_t s = 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * 256 + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint32_t k = 0; k < perThread; ++k) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        acc ^= table[s & mask];
    }
    if (acc == 0x9E3779B9u) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_gather4_l2(const uint32_t* __restrict__ table, uint64_t mask, uint32_t perThread, uint32_t* out)
{
    uint64_t s = 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * 256 + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint32_t k = 0; k < perThread; ++k) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        acc ^= table[s & mask];
    }
    if (acc == 0x9E3779B9u) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_write16(uint4* __restrict__ outp, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        outp[i] = make_uint4((uint32_t)i, 1, 2, 3);
    }
}

int main()
{
    const uint64_t streamBytes = 2ull << 30;            // 2 GiB streamed
    const uint64_t tableBytes = 8ull << 30;             // 8 GiB table: far beyond 32 MiB of L2 + 256 MiB Infinity Cache
    const uint32_t grid = 4096, perThread = 64;
    const uint64_t gathers = (uint64_t)grid * 256 * perThread;
    uint8_t* buf = nullptr; uint32_t* out = nullptr;
    CHECK(hipMalloc(&buf, tableBytes));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(buf, 1, tableBytes));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_stream16, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, streamBytes / 16, out);
    hipLaunchKernelGGL(calib_stream4, dim3(grid), dim3(256), 0, 0, (const uint32_t*)(buf + (4ull << 30)), streamBytes / 4, out);
    hipLaunchKernelGGL(calib_gather4, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, tableBytes / 4 - 1, perThread, out);
    hipLaunchKernelGGL(calib_gather4_l2, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, (2ull << 20) / 4 - 1, perThread, out);
    hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(256), 0, 0, (uint4*)buf, streamBytes / 16);
    CHECK(hipDeviceSynchronize());
    std::printf("{\"calib_stream16\": {\"read_bytes\": %llu}, \"calib_stream4\": {\"read_bytes\": %llu}, "
                "\"calib_gather4\": {\"gathers\": %llu, \"bytes_if_64B_lines\": %llu, \"bytes_if_32B_sectors\": %llu, \"bytes_if_128B_lines\": %llu}, "
                "\"calib_gather4_l2\": {\"gathers\": %llu, \"table_bytes\": %llu}, \"calib_write16\": {\"write_bytes\": %llu}}\n",
                (unsigned long long)streamBytes, (unsigned long long)streamBytes, (unsigned long long)gathers,
                (unsigned long long)(gathers * 64), (unsigned long long)(gathers * 32), (unsigned long long)(gathers * 128),
                (unsigned long long)gathers, (unsigned long long)(2ull << 20), (unsigned long long)streamBytes);
    return 0;
}
