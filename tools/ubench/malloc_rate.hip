// Micro-benchmark: what does device memory cost to allocate?  (the reader's scratch is 13 MB per block)
// Finding (two boxes): it depends on the state of the VRAM, not on the call.  The first allocation of a process that
// reaches pages another process has used pays for clearing them (0.6-1.8 s for 16 GB, whether through hipMalloc or
// hipMallocAsync); the same size again, or on a fresh box, is free (0.3 ms).  So the only lever is to allocate less.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }

int main()
{
    hipFree( nullptr );
    for ( int round = 0; round < 1; ++round ) {
        for ( size_t gb : { 1, 2, 3, 4, 5, 6, 8, 16 } ) {
            void* p = nullptr;
            double t0 = now();
            if ( hipMalloc( &p, gb << 30 ) != hipSuccess ) { std::printf( "hipMalloc %zu GB failed\n", gb ); continue; }
            double t1 = now();
            hipMemset( p, 0, 64 );
            hipDeviceSynchronize();
            double t2 = now();
            hipFree( p );
            double t3 = now();
            std::printf( "round %d hipMalloc %2zu GB: %.1f ms (%.1f ms/GB), first touch %.1f ms, hipFree %.1f ms\n", round, gb,
                         t1 - t0, ( t1 - t0 ) / gb, t2 - t1, t3 - t2 );
        }
    }
    {   // many pieces vs one piece
        std::vector<void*> ps( 16 );
        double t0 = now();
        for ( auto& p : ps ) hipMalloc( &p, size_t( 1 ) << 30 );
        double t1 = now();
        for ( auto& p : ps ) hipFree( p );
        std::printf( "16 x 1 GB: %.1f ms, free %.1f ms\n", t1 - t0, now() - t1 );
    }
    {
        hipStream_t s;
        hipStreamCreate( &s );
        for ( int round = 0; round < 2; ++round ) {
            void* p = nullptr;
            double t0 = now();
            hipError_t e = hipMallocAsync( &p, size_t( 16 ) << 30, s );
            hipStreamSynchronize( s );
            double t1 = now();
            if ( e == hipSuccess ) hipFreeAsync( p, s );
            hipStreamSynchronize( s );
            std::printf( "round %d hipMallocAsync 16 GB: %s %.1f ms, free %.1f ms\n", round, hipGetErrorString( e ), t1 - t0, now() - t1 );
        }
    }
    return 0;
}
