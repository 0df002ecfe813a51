// Developer microbenchmark: per-wave issue cost of the instruction patterns in k_huff's scalar chain (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -o issue_rate issue_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template<int MODE>
__global__ void k(uint64_t* out, uint32_t seed, int iters)
{
    uint32_t s = seed, t = 1, cur = 0;
    uint32_t v = threadIdx.x * 3 + seed;
    uint64_t mask = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {   // dependent SALU adds
            asm volatile(REP64("s_add_u32 %0, %0, %1\n\t") : "+s"(s) : "s"(t) : "scc");
        } else if constexpr (MODE == 1) {   // independent SALU
            asm volatile(REP16("s_add_u32 s90, %0, %1\n\ts_add_u32 s91, %0, %1\n\ts_add_u32 s92, %0, %1\n\ts_add_u32 s93, %0, %1\n\t") : "+s"(s) : "s"(t) : "scc", "s90", "s91", "s92", "s93");
        } else if constexpr (MODE == 2) {   // readlane -> salu -> readlane (lane select depends)
            asm volatile(REP64("v_readlane_b32 s90, %1, %0\n\ts_and_b32 %0, s90, 63\n\t") : "+s"(cur) : "v"(v) : "scc", "s90");
        } else if constexpr (MODE == 3) {   // dependent VALU
            asm volatile(REP64("v_add_u32 %0, %0, %0\n\t") : "+v"(v));
        } else if constexpr (MODE == 4) {   // taken branch each
            asm volatile(REP64("s_branch 1f\n\ts_nop 0\n\t1:\n\t") ::: "scc");
        } else if constexpr (MODE == 5) {   // chain step as in k_huff (no loop branch)
            asm volatile(REP64(
                "v_readlane_b32 s96, %[M], %[cur]\n\t"
                "s_lshr_b32 s94, s96, 10\n\t"
                "s_and_b64 s[98:99], s[96:97], 0x3ff\n\t"
                "s_lshl_b64 s[98:99], s[98:99], %[cur]\n\t"
                "s_or_b64 %[mask], %[mask], s[98:99]\n\t"
                "s_add_u32 %[cur], %[cur], s94\n\t"
                "s_and_b32 %[cur], %[cur], 63\n\t")
                : [cur] "+s"(cur), [mask] "+s"(mask) : [M] "v"(v) : "scc", "s94", "s96", "s97", "s98", "s99");
        } else if constexpr (MODE == 6) {   // ds_bpermute dependent chain
            asm volatile(REP64("ds_bpermute_b32 %0, %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(v));
        } else if constexpr (MODE == 7) {   // LDS read dependent chain
            __shared__ uint32_t lds[1024];
            lds[threadIdx.x] = threadIdx.x * 4;
            asm volatile(REP64("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(v));
            (void)lds;
        } else if constexpr (MODE == 8) {   // s_nop 0 (pure issue)
            asm volatile(REP64("s_nop 0\n\t"));
        } else if constexpr (MODE == 9) {   // not-taken branches
            asm volatile(REP64("s_cmp_eq_u32 %0, 0x12345\n\ts_cbranch_scc1 9f\n\t") "9:\n\t" :: "s"(s) : "scc");
        } else if constexpr (MODE == 10) {  // v_readfirstlane -> valu using sgpr -> readfirstlane
            asm volatile(REP64("v_readfirstlane_b32 s90, %0\n\tv_add_u32 %0, s90, %0\n\t") : "+v"(v) :: "s90");
        } else if constexpr (MODE == 11) {  // readlane with constant lane then salu
            asm volatile(REP64("v_readlane_b32 s90, %1, 5\n\ts_add_u32 %0, %0, s90\n\t") : "+s"(s) : "v"(v) : "scc", "s90");
        } else if constexpr (MODE == 12) {  // v_cmp -> s_ff1 (vcc) -> lane-dependent
            asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n\ts_ff1_i32_b64 %0, vcc\n\t") : "+s"(s) : "v"(v) : "vcc", "scc");
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = s + cur + v + (uint32_t)mask; }
}

template<int MODE>
void run(const char* name, int instrPerRep, int blocks, int threads)
{
    uint64_t* d; (void)hipMalloc(&d, 4096 * 16);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1u, iters);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1u, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    // s_memtime / readcyclecounter on gfx9 counts at a fixed 100 MHz?  report raw ticks too
    const double n = (double)iters * 64 * instrPerRep;
    printf("%-44s blocks=%5d threads=%4d ticks/instr=%.3f  ns/instr=%.2f (event, whole kernel)\n", name, blocks, threads,
           (double)h[0] / n, ms * 1e6 / n);
    (void)hipFree(d);
}

int main()
{
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; (void)hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("clock %d kHz, wall clock %d kHz\n", clk, wclk);
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int blocks = cfg == 0 ? 1 : (cfg == 1 ? 2048 : 1024);
        const int threads = cfg == 2 ? 256 : 64;
        run<0>("dependent s_add", 1, blocks, threads);
        run<1>("independent s_add x4", 1, blocks, threads);
        run<8>("s_nop 0", 1, blocks, threads);
        run<2>("readlane(s-sel) -> s_and -> readlane", 2, blocks, threads);
        run<11>("readlane(const) -> s_add", 2, blocks, threads);
        run<3>("dependent v_add", 1, blocks, threads);
        run<4>("taken s_branch (+skipped nop)", 1, blocks, threads);
        run<9>("s_cmp + not-taken cbranch", 2, blocks, threads);
        run<5>("huff chain step, 7 instr", 7, blocks, threads);
        run<6>("ds_bpermute dependent + wait", 1, blocks, threads);
        run<7>("ds_read dependent + wait", 1, blocks, threads);
        run<10>("readfirstlane -> v_add(sgpr)", 2, blocks, threads);
        run<12>("v_cmp vcc -> s_ff1", 2, blocks, threads);
    }
    return 0;
}
