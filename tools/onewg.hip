// what does ONE workgroup cost per barrier step on an otherwise idle MI355X?  (tools/: the band form of the triangular solve)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k_bar(int iters, double* out, const int* __restrict__ idx, int mode)
{
    __shared__ double lds[512];
    double v = threadIdx.x;
    long long t0 = wall_clock64();
    long long c0 = clock64();
    for(int i = 0; i < iters; ++i)
    {
        if(mode & 1) // a dependent chain of 32 adds
            for(int k = 0; k < 32; ++k)
                v = v + 1.0;
        if(mode & 2) // a scalar load per step whose address moves on
            v += (double)idx[(i * 128) & 0xfffff];
        lds[threadIdx.x] = v;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        v += lds[(threadIdx.x + 64) & 511];
    }
    long long c1 = clock64();
    long long t1 = wall_clock64();
    if(threadIdx.x == 0)
    {
        out[0] = v;
        out[1] = (double)(c1 - c0);
        out[2] = (double)(t1 - t0);
    }
}
int main()
{
    double* d; int* idx;
    hipMalloc(&d, 64); hipMalloc(&idx, 4 << 20); hipMemset(idx, 0, 4 << 20);
    for(int threads : {64, 256, 512})
        for(int mode = 0; mode < 4; ++mode)
        {
            const int iters = 100000;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            k_bar<<<1, threads>>>(1000, d, idx, mode);
            hipEventRecord(a);
            k_bar<<<1, threads>>>(iters, d, idx, mode);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("threads %d mode %d: %.3f us per step, %.0f shader clocks per step, wall clock ticks per step %.1f\n", threads, mode,
                   ms * 1e3 / iters, h[1] / iters, h[2] / iters);
        }
    return 0;
}
