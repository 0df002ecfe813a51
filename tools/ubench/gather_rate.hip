// Developer micro-benchmark: how many dependent 4-byte gathers per second the GPU sustains from tables of the size of a
// block's packed LF table (900 000 x u32 = 3.6 MB), when every XCD works on its own table (the walk's placement) -- by
// workgroups per XCD and independent chains per lane.  Answers: is k_walk at the machine's limit, or at the limit of the
// gathers it keeps in flight?   usage: gather_rate  -> JSON lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>
#include <numeric>

constexpr uint32_t N = 900000;
constexpr uint32_t STRIDE = 1u << 20;

template<int CHAINS>
__global__ __launch_bounds__( 256 ) void
k_chase( const uint32_t* __restrict__ tables, uint32_t nTables, uint32_t steps, uint32_t* out )
{
    uint32_t xcc;
    asm volatile( "s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"( xcc ) );
    const uint32_t* const tab = tables + (size_t)( ( xcc & 7u ) % nTables ) * STRIDE;
    uint32_t p[CHAINS];
#pragma unroll
    for ( int c = 0; c < CHAINS; ++c ) p[c] = ( blockIdx.x * 256u + threadIdx.x ) * 977u * ( c + 1 ) % N;
    for ( uint32_t s = 0; s < steps; ++s ) {
#pragma unroll
        for ( int c = 0; c < CHAINS; ++c ) p[c] = tab[p[c]] >> 8;
    }
    uint32_t acc = 0;
#pragma unroll
    for ( int c = 0; c < CHAINS; ++c ) acc ^= p[c];
    if ( acc == 0xFFFFFFFFu ) out[0] = acc;
}

template<int CHAINS>
static void
run( const uint32_t* dTab, uint32_t nTables, uint32_t wgsPerXcd, uint32_t* dOut )
{
    const uint32_t steps = 2048 / CHAINS;
    hipEvent_t a, b;
    hipEventCreate( &a ); hipEventCreate( &b );
    const uint32_t grid = 8 * wgsPerXcd;
    hipLaunchKernelGGL( k_chase<CHAINS>, dim3( grid ), dim3( 256 ), 0, nullptr, dTab, nTables, steps, dOut );
    hipEventRecord( a );
    const int reps = 5;
    for ( int r = 0; r < reps; ++r ) hipLaunchKernelGGL( k_chase<CHAINS>, dim3( grid ), dim3( 256 ), 0, nullptr, dTab, nTables, steps, dOut );
    hipEventRecord( b );
    hipEventSynchronize( b );
    float ms = 0;
    hipEventElapsedTime( &ms, a, b );
    const double gathers = (double)grid * 256 * CHAINS * steps * reps;
    std::printf( "{\"tables\": %u, \"wgs_per_xcd\": %u, \"chains_per_lane\": %d, \"G_gathers_per_s\": %.1f}\n", nTables, wgsPerXcd, CHAINS,
                 gathers / ( ms * 1e-3 ) / 1e9 );
    std::fflush( stdout );
}

int
main()
{
    // one random N-cycle per table, packed like the decoder's: next << 8 | byte
    const uint32_t nTablesMax = 16;
    std::vector<uint32_t> host( (size_t)nTablesMax * STRIDE, 0 );
    std::mt19937 rng( 7 );
    for ( uint32_t t = 0; t < nTablesMax; ++t ) {
        std::vector<uint32_t> order( N );
        std::iota( order.begin(), order.end(), 0u );
        std::shuffle( order.begin(), order.end(), rng );
        for ( uint32_t i = 0; i < N; ++i ) host[(size_t)t * STRIDE + order[i]] = ( order[( i + 1 ) % N] << 8 ) | ( i & 0xFF );
    }
    uint32_t *dTab = nullptr, *dOut = nullptr;
    hipMalloc( &dTab, host.size() * 4 );
    hipMalloc( &dOut, 256 );
    hipMemcpy( dTab, host.data(), host.size() * 4, hipMemcpyHostToDevice );
    for ( uint32_t nTables : { 8u, 16u } ) {      /* one table per XCD; two per XCD (7.2 MB: beyond its 4 MB L2) */
        for ( uint32_t wgs : { 64u, 128u, 256u } ) {
            run<1>( dTab, nTables, wgs, dOut );
            run<2>( dTab, nTables, wgs, dOut );
            run<4>( dTab, nTables, wgs, dOut );
        }
    }
    return 0;
}
