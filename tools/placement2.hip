// placement2.hip -- which address bits make concurrent streams collide?  (tools/, not product)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/placement2 tools/placement2.hip && tools/_bin/placement2
// One 9 GiB allocation; streams of 1 GiB placed at chosen byte offsets inside it.  For every offset d of a list the
// 5-stream update (3 reads + 2 writes, the shape of k_cg_update) runs on streams at k * (1 GiB + d), k = 0..4, and a
// 2-stream copy on (0, 1 GiB + d).  If the speed depends on d, the channel / bank hash uses those bits, and an allocator
// can separate the solver's work vectors by construction.  A second part repeats the test on SEPARATE hipMalloc blocks
// with the same offsets applied inside each block (what the caching allocator could do).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ __launch_bounds__(256) void k_upd(int64_t np, v2* __restrict__ r, const v2* __restrict__ q, const v2* __restrict__ d,
                                             v2* __restrict__ z, double alpha)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= np)
        return;
    v2 a = __builtin_nontemporal_load(r + i), b = __builtin_nontemporal_load(q + i), c = __builtin_nontemporal_load(d + i);
    a = a - alpha * b;
    __builtin_nontemporal_store(a, r + i);
    __builtin_nontemporal_store(c * a, z + i);
}
// the same update with a scrambled block -> chunk map (chunk = block * mul mod nblocks, nblocks a power of two, mul odd):
// the chunks in flight at any time are spread over the whole vectors instead of forming one window per stream
__global__ __launch_bounds__(256) void k_upd_scr(int64_t np, unsigned mul, v2* __restrict__ r, const v2* __restrict__ q,
                                                 const v2* __restrict__ d, v2* __restrict__ z, double alpha)
{
    const unsigned nb = gridDim.x;
    const unsigned cb = (blockIdx.x * mul) & (nb - 1);
    const int64_t  i  = (int64_t)cb * 256 + threadIdx.x;
    if(i >= np)
        return;
    v2 a = __builtin_nontemporal_load(r + i), b = __builtin_nontemporal_load(q + i), c = __builtin_nontemporal_load(d + i);
    a = a - alpha * b;
    __builtin_nontemporal_store(a, r + i);
    __builtin_nontemporal_store(c * a, z + i);
}
__global__ __launch_bounds__(256) void k_copy(int64_t np, const v2* __restrict__ a, v2* __restrict__ b)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i < np)
        __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
__global__ __launch_bounds__(256) void k_ww(int64_t np, v2* __restrict__ a, v2* __restrict__ b)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i < np)
    {
        __builtin_nontemporal_store(v2{1.0, 2.0}, a + i);
        __builtin_nontemporal_store(v2{3.0, 4.0}, b + i);
    }
}
__global__ __launch_bounds__(256) void k_rr(int64_t np, const v2* __restrict__ a, const v2* __restrict__ b, double* sink)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i < np)
    {
        const v2 x = __builtin_nontemporal_load(a + i), y = __builtin_nontemporal_load(b + i);
        if(x.x + y.y == 12345.678)
            *sink = 1.0;
    }
}
__global__ __launch_bounds__(256) void k_fill(int64_t np, v2* a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if(i < np)
        a[i] = v2{1.0, 1.0};
}
static double timed(void (*f)(void*), void* arg, int reps = 6)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(arg);
    CHECK(hipEventRecord(e0, 0));
    for(int i = 0; i < reps; ++i) f(arg);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}
struct Arg { v2 *a, *b, *c, *d; int64_t np; };
static void run_upd(void* p) { Arg* g = (Arg*)p; hipLaunchKernelGGL(k_upd, dim3((unsigned)((g->np + 255) / 256)), dim3(256), 0, 0, g->np, g->a, g->b, g->c, g->d, 1e-30); }
static unsigned g_mul = 1;
static void run_upd_scr(void* p) { Arg* g = (Arg*)p; hipLaunchKernelGGL(k_upd_scr, dim3((unsigned)((g->np + 255) / 256)), dim3(256), 0, 0, g->np, g_mul, g->a, g->b, g->c, g->d, 1e-30); }
static double* g_sink = nullptr;
static void run_ww(void* p) { Arg* g = (Arg*)p; hipLaunchKernelGGL(k_ww, dim3((unsigned)((g->np + 255) / 256)), dim3(256), 0, 0, g->np, g->a, g->b); }
static void run_rr(void* p) { Arg* g = (Arg*)p; hipLaunchKernelGGL(k_rr, dim3((unsigned)((g->np + 255) / 256)), dim3(256), 0, 0, g->np, g->a, g->b, g_sink); }
static void run_copy(void* p) { Arg* g = (Arg*)p; hipLaunchKernelGGL(k_copy, dim3((unsigned)((g->np + 255) / 256)), dim3(256), 0, 0, g->np, g->a, g->b); }

int main(int argc, char** argv)
{
    const int64_t GiB = 1ll << 30, MiB = 1ll << 20;
    const int64_t n = GiB / 8, np = n / 2;
    char* big = nullptr;
    CHECK(hipMalloc(&big, 10 * GiB));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((10 * GiB / 16 + 255) / 256)), dim3(256), 0, 0, 10 * GiB / 16, (v2*)big);
    CHECK(hipDeviceSynchronize());
    printf("big block at %p\n", (void*)big);
    const int64_t ds[] = {0, 256, 4096, 64 * 1024, MiB, 2 * MiB, 4 * MiB, 6 * MiB, 8 * MiB, 16 * MiB, 24 * MiB, 32 * MiB, 48 * MiB, 64 * MiB, 96 * MiB,
                          128 * MiB, 192 * MiB, 256 * MiB, 384 * MiB, 512 * MiB, 768 * MiB, 1000 * MiB};
    printf("# part 1: one allocation, streams at k * (1 GiB + d)\n");
    for(int64_t d : ds)
    {
        Arg g = {(v2*)(big), (v2*)(big + 1 * (GiB + d)), (v2*)(big + 2 * (GiB + d)), (v2*)(big + 3 * (GiB + d)), np};
        const double t5 = timed(run_upd, &g);
        Arg c = {(v2*)(big), (v2*)(big + GiB + d), nullptr, nullptr, np};
        const double t2 = timed(run_copy, &c);
        printf("d = %9.3f MiB : upd(3R+2W) %6.1f GB/s   copy %6.1f GB/s\n", (double)d / MiB, 5.0 * GiB / t5 / 1e6, 2.0 * GiB / t2 / 1e6);
        fflush(stdout);
    }
    CHECK(hipFree(big));
    CHECK(hipMalloc(&g_sink, 8));
    if(argc > 1 && argv[1][0] == 's') // part 5 only: scrambled block order on good and bad placements
    {
        char* ar = nullptr;
        CHECK(hipMalloc(&ar, 5 * GiB + 64 * MiB));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((5 * GiB / 16 + 255) / 256)), dim3(256), 0, 0, 5 * GiB / 16, (v2*)ar);
        const int NB = 8;
        char* blk[NB];
        for(int k = 0; k < NB; ++k)
        {
            CHECK(hipMalloc(&blk[k], GiB + 256));
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((GiB / 16 + 255) / 256)), dim3(256), 0, 0, GiB / 16, (v2*)blk[k]);
        }
        CHECK(hipDeviceSynchronize());
        const unsigned muls[] = {1u, 3u, 17u, 129u, 1025u, 4097u, 40503u, 0x9E3779B1u & 0x3FFFFu | 1u};
        const int64_t dsel[] = {0, 4 * MiB, 32 * MiB, 1000 * MiB > 250 * MiB ? 250 * MiB : 0};
        for(int64_t d : dsel)
        {
            Arg g = {(v2*)(ar), (v2*)(ar + 1 * (GiB + d)), (v2*)(ar + 2 * (GiB + d)), (v2*)(ar + 3 * (GiB + d)), np};
            printf("one allocation, d = %6.1f MiB : linear %5.0f | scrambled", (double)d / MiB, 5.0 * GiB / timed(run_upd, &g) / 1e6);
            for(unsigned m : muls)
            {
                g_mul = m;
                printf(" x%u: %5.0f", m, 5.0 * GiB / timed(run_upd_scr, &g) / 1e6);
            }
            printf("\n"); fflush(stdout);
        }
        const int grp[][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 2, 4, 6}, {1, 3, 5, 7}, {0, 1, 4, 5}, {2, 3, 6, 7}, {0, 3, 5, 6}, {1, 2, 4, 7}};
        for(auto& gr : grp)
        {
            Arg g = {(v2*)blk[gr[0]], (v2*)blk[gr[1]], (v2*)blk[gr[2]], (v2*)blk[gr[3]], np};
            printf("separate blocks (%d,%d,%d,%d) : linear %5.0f | scrambled", gr[0], gr[1], gr[2], gr[3], 5.0 * GiB / timed(run_upd, &g) / 1e6);
            for(unsigned m : muls)
            {
                g_mul = m;
                printf(" x%u: %5.0f", m, 5.0 * GiB / timed(run_upd_scr, &g) / 1e6);
            }
            printf("\n"); fflush(stdout);
        }
        return 0;
    }
    if(argc > 1 && argv[1][0] == 'w') // part 4 only: which PAIRS collide, by role -- two writes, two reads (separate blocks)
    {
        const int NB = 10;
        char* blk[NB];
        for(int k = 0; k < NB; ++k)
        {
            CHECK(hipMalloc(&blk[k], GiB + 256));
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((GiB / 16 + 255) / 256)), dim3(256), 0, 0, GiB / 16, (v2*)blk[k]);
        }
        CHECK(hipDeviceSynchronize());
        for(int mode = 0; mode < 3; ++mode)
        {
            printf("# pairwise %s GB/s between %d separate 1-GiB blocks\n", mode == 0 ? "write+write" : (mode == 1 ? "read+read" : "copy"), NB);
            for(int a = 0; a < NB; ++a)
            {
                for(int b2 = 0; b2 < NB; ++b2)
                {
                    if(a == b2) { printf("     -"); continue; }
                    Arg c = {(v2*)blk[a], (v2*)blk[b2], nullptr, nullptr, np};
                    printf(" %5.0f", 2.0 * GiB / timed(mode == 0 ? run_ww : (mode == 1 ? run_rr : run_copy), &c, 4) / 1e6);
                }
                printf("\n"); fflush(stdout);
            }
        }
        // the 5-stream update on triples of groups: is the speed explained by the write pair (a, d)?
        printf("# upd(3R+2W) r=a q=b dinv=c z=d\n");
        for(int a = 0; a < 4; ++a)
            for(int d = 4; d < 8; ++d)
            {
                Arg g = {(v2*)blk[a], (v2*)blk[8], (v2*)blk[9], (v2*)blk[d], np};
                printf("  write pair (%d,%d): %5.0f", a, d, 5.0 * GiB / timed(run_upd, &g) / 1e6);
                Arg g2 = {(v2*)blk[a], (v2*)blk[(a + 1) % 4], (v2*)blk[4 + (d + 1) % 4], (v2*)blk[d], np};
                printf("   with other readers: %5.0f\n", 5.0 * GiB / timed(run_upd, &g2) / 1e6);
            }
        return 0;
    }
    if(argc > 1) // part 3 only: 14 vectors inside ONE allocation at stride 1 GiB + D; the groups of profiles/r02_placement_probe.txt
    {
        const int NV = 14;
        const int grp[][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {8, 9, 10, 11}, {0, 4, 8, 12}, {1, 5, 9, 13}, {12, 13, 10, 11}, {0, 2, 4, 6},
                              {0, 1, 4, 5}, {0, 3, 6, 9}, {2, 3, 4, 5}, {6, 7, 8, 9}, {1, 6, 11, 12}, {3, 7, 10, 13}, {2, 5, 8, 11}};
        const int64_t Ds[] = {4 * MiB, 0, 16 * MiB, 32 * MiB, 48 * MiB};
        for(int64_t D : Ds)
        {
            char* ar = nullptr;
            CHECK(hipMalloc(&ar, NV * (GiB + D) + 64 * MiB));
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((NV * (GiB + D) / 16 + 255) / 256)), dim3(256), 0, 0, NV * (GiB + D) / 16, (v2*)ar);
            CHECK(hipDeviceSynchronize());
            printf("arena stride 1 GiB + %5.1f MiB :", (double)D / MiB);
            double lo = 1e30, hi = 0;
            for(auto& gr : grp)
            {
                Arg g = {(v2*)(ar + gr[0] * (GiB + D)), (v2*)(ar + gr[1] * (GiB + D)), (v2*)(ar + gr[2] * (GiB + D)), (v2*)(ar + gr[3] * (GiB + D)), np};
                const double v = 5.0 * GiB / timed(run_upd, &g) / 1e6;
                lo = v < lo ? v : lo; hi = v > hi ? v : hi;
                printf(" %5.0f", v);
            }
            printf("   | min %5.0f max %5.0f\n", lo, hi); fflush(stdout);
            CHECK(hipFree(ar));
        }
        return 0;
    }
    // part 2: 8 separate blocks of 1 GiB + 1 GiB slack; group tests with in-block offsets 0 vs k * S
    const int NB = 8;
    char* blk[NB];
    for(int k = 0; k < NB; ++k)
    {
        CHECK(hipMalloc(&blk[k], 2 * GiB + 64 * MiB));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)(((2 * GiB + 64 * MiB) / 16 + 255) / 256)), dim3(256), 0, 0, (2 * GiB + 64 * MiB) / 16, (v2*)blk[k]);
    }
    CHECK(hipDeviceSynchronize());
    printf("# part 2: separate blocks at");
    for(int k = 0; k < NB; ++k) printf(" %p", (void*)blk[k]);
    printf("\n");
    const int groups[][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 2, 4, 6}, {1, 3, 5, 7}, {0, 1, 4, 5}, {2, 3, 6, 7}, {0, 3, 5, 6}, {1, 2, 4, 7}};
    const int64_t Ss[] = {0, 2 * MiB, 8 * MiB, 32 * MiB, 128 * MiB, 256 * MiB};
    for(int64_t S : Ss)
    {
        printf("in-block offset k * %7.1f MiB :", (double)S / MiB);
        for(auto& gr : groups)
        {
            Arg g = {(v2*)(blk[gr[0]] + 0 * S), (v2*)(blk[gr[1]] + 1 * S), (v2*)(blk[gr[2]] + 2 * S), (v2*)(blk[gr[3]] + 3 * S), np};
            printf(" %6.0f", 5.0 * GiB / timed(run_upd, &g) / 1e6);
        }
        printf("\n"); fflush(stdout);
    }
    // pairwise copy speed between the blocks (offset 0): which pairs collide?
    printf("# pairwise copy GB/s (row = source block, column = destination block)\n");
    for(int a = 0; a < NB; ++a)
    {
        for(int b2 = 0; b2 < NB; ++b2)
        {
            if(a == b2) { printf("      -"); continue; }
            Arg c = {(v2*)blk[a], (v2*)blk[b2], nullptr, nullptr, np};
            printf(" %6.0f", 2.0 * GiB / timed(run_copy, &c, 4) / 1e6);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
