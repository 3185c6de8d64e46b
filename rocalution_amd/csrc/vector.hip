// vector.hip -- device vectors: BLAS-1 updates and single-launch reductions.
//
// Replaces src/base/hip/hip_vector.cpp + hip_kernels_vector.hpp of the reference (which call
// rocBLAS axpy/scal/dot/nrm2 and small elementwise kernels).  Arithmetic per element is the
// reference HOST expression (src/base/host/host_vector.cpp, cited per op), compiled with
// -ffp-contract=off so that updates are bit-identical to the OpenMP backend; reductions use a
// fixed-order tree (the reference's OpenMP reduction order is itself thread-count dependent).
#include "device_utils.hpp"
#include <chrono>
#include "matrix_impl.hpp"

namespace ramd
{

// ---------------------------------------------------------------- elementwise kernels
// One generic driver: 16-byte packets, grid-stride, scalar tail.  F is a per-element functor.
template <typename T, typename F>
__global__ __launch_bounds__(kBlock) void k_map(int64_t n, F f)
{
    using P            = typename Pack<T>::type;
    constexpr int NP   = Pack<T>::N;
    int64_t       np   = n / NP;
    int64_t       gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t       gsz  = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = gtid; i < np; i += gsz)
        f.packet(i);
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
        f.scalar(i);
}

template <typename T, typename F>
int launch_map(int64_t n, F f)
{
    if(n <= 0)
        return RAMD_OK;
    // one-shot grid (a packet per thread, no grid-stride loop): 5.3-6.7 TB/s against 4.6-4.9 TB/s with a resident
    // grid on these streaming patterns (tools/membench.hip, profiles/r02_membench.txt)
    const int64_t np   = (n + Pack<T>::N - 1) / Pack<T>::N;
    int64_t       g    = (np + kBlock - 1) / kBlock;
    const int     grid = (int)(g < 1 ? 1 : (g > 0x7fffffff ? 0x7fffffff : g));
    hipLaunchKernelGGL((k_map<T, F>), dim3(grid), dim3(kBlock), 0, backend().cur, n, f);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

// v = value
template <typename T>
struct FSet
{
    T* v;
    T  a;
    __device__ __forceinline__ void packet(int64_t i) const
    {
        typename Pack<T>::type p;
#pragma unroll
        for(int k = 0; k < Pack<T>::N; ++k)
            pk_elems<T>(p)[k] = a;
        reinterpret_cast<typename Pack<T>::type*>(v)[i] = p;
    }
    __device__ __forceinline__ void scalar(int64_t i) const
    {
        v[i] = a;
    }
};

// host_vector.cpp:635-651  AddScale: v = v + alpha*x
template <typename T>
struct FAddScale
{
    T*       v;
    const T* x;
    T        alpha;
    __device__ __forceinline__ void packet(int64_t i) const
    {
        using P = typename Pack<T>::type;
        P a     = reinterpret_cast<P*>(v)[i];
        P b     = reinterpret_cast<const P*>(x)[i];
#pragma unroll
        for(int k = 0; k < Pack<T>::N; ++k)
            pk_elems<T>(a)[k] = pk_elems<T>(a)[k] + alpha * pk_elems<T>(b)[k];
        reinterpret_cast<P*>(v)[i] = a;
    }
    __device__ __forceinline__ void scalar(int64_t i) const
    {
        v[i] = v[i] + alpha * x[i];
    }
};

// host_vector.cpp:654-670  ScaleAdd: v = alpha*v + x
template <typename T>
struct FScaleAdd
{
    T*       v;
    const T* x;
    T        alpha;
    __device__ __forceinline__ void packet(int64_t i) const
    {
        using P = typename Pack<T>::type;
        P a     = reinterpret_cast<P*>(v)[i];
        P b     = reinterpret_cast<const P*>(x)[i];
#pragma unroll
        for(int k = 0; k < Pack<T>::N; ++k)
            pk_elems<T>(a)[k] = alpha * pk_elems<T>(a)[k] + pk_elems<T>(b)[k];
        reinterpret_cast<P*>(v)[i] = a;
    }
    __device__ __forceinline__ void scalar(int64_t i) const
    {
        v[i] = alpha * v[i] + x[i];
    }
};

// host_vector.cpp:693-720  ScaleAddScale on a sub-range (element loop: the ranges need not be packet aligned)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_scale_add_scale_offset(int64_t n, T* __restrict__ v, const T* __restrict__ x,
                                                                   T alpha, T beta)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        v[i] = alpha * v[i] + beta * x[i];
}

// host_vector.cpp:672-690  ScaleAddScale: v = alpha*v + beta*x
template <typename T>
struct FScaleAddScale
{
    T*       v;
    const T* x;
    T        alpha, beta;
    __device__ __forceinline__ void packet(int64_t i) const
    {
        using P = typename Pack<T>::type;
        P a     = reinterpret_cast<P*>(v)[i];
        P b     = reinterpret_cast<const P*>(x)[i];
#pragma unroll
        for(int k = 0; k < Pack<T>::N; ++k)
            pk_elems<T>(a)[k] = alpha * pk_elems<T>(a)[k] + beta * pk_elems<T>(b)[k];
        reinterpret_cast<P*>(v)[i] = a;
    }
    __device__ __forceinline__ void scalar(int64_t i) const
    {
        v[i] = alpha * v[i] + beta * x[i];
    }
};

// host_vector.cpp:723-747  ScaleAdd2: v = alpha*v + beta*x + gamma*y
template <typename T>
struct FScaleAdd2
{
    T*       v;
    const T* x;
    const T* y;
    T        alpha, beta, gamma;
    __device__ __forceinline__ void packet(int64_t i) const
    {
        using P = typename Pack<T>::type;
        P a     = reinterpret_cast<P*>(v)[i];
        P b     = reinterpret_cast<const P*>(x)[i];
        P c     = reinterpret_cast<const P*>(y)[i];
#pragma unroll
        for(int k = 0; k < Pack<T>::N; ++k)
            pk_elems<T>(a)[k]
                = alpha * pk_elems<T>(a)[k] + beta * pk_elems<T>(b)[k] + gamma * pk_elems<T>(c)[k];
        reinterpret_cast<P*>(v)[i] = a;
    }
    __device__ __forceinline__ void scalar(int64_t i) const
    {
        v[i] = alpha * v[i] + beta * x[i] + gamma * y[i];
    }
};

// host_vector.cpp:750-760  Scale: v *= alpha
template <typename T>
struct FScale
{
    T* v;
    T  alpha;
    __device__ __forceinline__ void packet(int64_t i) const
    {
        using P = typename Pack<T>::type;
        P a     = reinterpret_cast<P*>(v)[i];
#pragma unroll
        for(int k = 0; k < Pack<T>::N; ++k)
            pk_elems<T>(a)[k] *= alpha;
        reinterpret_cast<P*>(v)[i] = a;
    }
    __device__ __forceinline__ void scalar(int64_t i) const
    {
        v[i] *= alpha;
    }
};

// host_vector.cpp:1231-1279  PointWiseMult: v = a*b (a may alias v)
template <typename T>
struct FPointWise
{
    T*       v;
    const T* a;
    const T* b;
    __device__ __forceinline__ void packet(int64_t i) const
    {
        using P = typename Pack<T>::type;
        P p     = reinterpret_cast<const P*>(a)[i];
        P q     = reinterpret_cast<const P*>(b)[i];
#pragma unroll
        for(int k = 0; k < Pack<T>::N; ++k)
            pk_elems<T>(p)[k] = pk_elems<T>(p)[k] * pk_elems<T>(q)[k];
        reinterpret_cast<P*>(v)[i] = p;
    }
    __device__ __forceinline__ void scalar(int64_t i) const
    {
        v[i] = a[i] * b[i];
    }
};

// host_vector.cpp:258-330  CopyFromFloat / CopyFromDouble (value cast)
template <typename D, typename S>
__global__ __launch_bounds__(kBlock) void k_cast(int64_t n, D* dst, const S* src)
{
    int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        dst[i] = static_cast<D>(src[i]);
}

// host_vector.cpp:1365-1412  CopyFromPermute (scatter) / CopyFromPermuteBackward (gather)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_permute_fwd(int64_t n, T* dst, const T* src,
                                                        const int* perm)
{
    int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        dst[perm[i]] = src[i];
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_permute_bwd(int64_t n, T* dst, const T* src,
                                                        const int* perm)
{
    int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        dst[i] = src[perm[i]];
}

// host_vector.cpp:1469-1490  GetIndexValues: out[i] = v[index[i]]   (halo pack)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather(int64_t n, T* out, const T* v, const int* index)
{
    int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        out[i] = v[index[i]];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_copy_offset(int64_t n, T* dst, const T* src)
{
    int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        dst[i] = src[i];
}

// ---------------------------------------------------------------- reductions
// mode: 0 dot(a,b)  1 sum(a)  2 asum(a)
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void k_reduce(int64_t n, const T* a, const T* b, ReduceCtx ctx,
                                                   int slot, int op)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    __shared__ double lds[8];
    int64_t np   = n / NP;
    int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    double  acc  = 0.0;
    for(int64_t i = gtid; i < np; i += gsz)
    {
        P pa = reinterpret_cast<const P*>(a)[i];
        if(MODE == 0)
        {
            P pb = reinterpret_cast<const P*>(b)[i];
#pragma unroll
            for(int k = 0; k < NP; ++k)
                acc += (double)pk_elems<T>(pa)[k] * (double)pk_elems<T>(pb)[k];
        }
        else
        {
#pragma unroll
            for(int k = 0; k < NP; ++k)
                acc += MODE == 1 ? (double)pk_elems<T>(pa)[k] : fabs((double)pk_elems<T>(pa)[k]);
        }
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
    {
        if(MODE == 0)
            acc += (double)a[i] * (double)b[i];
        else
            acc += MODE == 1 ? (double)a[i] : fabs((double)a[i]);
    }
    const double vals[1]  = {acc};
    const int    slots[1] = {slot};
    const int    ops[1]   = {op};
    grid_reduce_finish<1>(ctx, vals, slots, ops, lds);
}

// Amax: value and first index of max |a_i| (host_vector.cpp Amax).  Two-stage, tiny 2nd stage.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_amax_partial(int64_t n, const T* a, double* pval,
                                                         long long* pidx)
{
    __shared__ double    sv[kBlock];
    __shared__ long long si[kBlock];
    int64_t              gsz  = (int64_t)gridDim.x * blockDim.x;
    double               best = -1.0;
    long long            bi   = 0;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
    {
        double v = fabs((double)a[i]);
        if(v > best)
        {
            best = v;
            bi   = i;
        }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for(int s = kBlock / 2; s > 0; s >>= 1)
    {
        if((int)threadIdx.x < s)
        {
            double    ov = sv[threadIdx.x + s];
            long long oi = si[threadIdx.x + s];
            if(ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))
            {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if(threadIdx.x == 0)
    {
        pval[blockIdx.x] = sv[0];
        pidx[blockIdx.x] = si[0];
    }
}

// Sum of a few million partial sums (one per wave of a fused SpMV + dot) into a scalar slot.  With a thread per 16-byte packet
// this took 4096 workgroups and 55 microseconds at 512^3 -- 41 of them the workgroups' tickets, ~10 ns each on the one ticket
// word, all at the end of a kernel too short to spread them (profiles/r03_kernel_stats_cg.txt: k_reduce<double, 1>).  Here at most
// 256 workgroups, every thread with U independent packets in flight per round; the order of the additions is fixed by (n, grid).
template <int U>
__global__ __launch_bounds__(kBlock) void k_sum_partials(int64_t n, const double* __restrict__ a, ReduceCtx ctx, int slot)
{
    __shared__ double lds[8];
    const int64_t np   = n / 2;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    double        acc  = 0.0;
    for(int64_t i = gtid; i < np; i += U * gsz)
    {
        v2f64 v[U];
#pragma unroll
        for(int u = 0; u < U; ++u)
        {
            const int64_t j = i + u * gsz;
            v[u]            = j < np ? reinterpret_cast<const v2f64*>(a)[j] : v2f64{0.0, 0.0};
        }
#pragma unroll
        for(int u = 0; u < U; ++u)
        {
            acc += v[u].x;
            acc += v[u].y;
        }
    }
    if((n & 1) && gtid == 0)
        acc += a[n - 1];
    const double vals[1]  = {acc};
    const int    slots[1] = {slot};
    const int    ops[1]   = {(int)RED_SUM};
    grid_reduce_finish<1>(ctx, vals, slots, ops, lds);
}

int reduce_sum_to_slot(const double* a, int64_t n, int slot)
{
    if(n <= 0)
        return ramd_scalars_set(slot, 0.0);
    static const bool plain = getenv("RAMD_SUM_PARTIALS") && atoi(getenv("RAMD_SUM_PARTIALS")) == 0; // (A/B: the general kernel)
    if(plain)
    {
        hipLaunchKernelGGL((k_reduce<double, 1>), dim3(reduce_grid((n + 1) / 2)), dim3(kBlock), 0, backend().cur, n, a, a,
                           reduce_ctx(), slot, (int)RED_SUM);
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    constexpr int U = 8;
    int64_t       g = (n / 2 + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
    g               = g < 1 ? 1 : (g > 256 ? 256 : g);
    hipLaunchKernelGGL((k_sum_partials<U>), dim3((unsigned)g), dim3(kBlock), 0, backend().cur, n, a, reduce_ctx(), slot);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

} // namespace ramd

using namespace ramd;

// ---------------------------------------------------------------- helpers
#define CHECK_VEC(v)                                      \
    do                                                    \
    {                                                     \
        if(!(v))                                          \
            RAMD_FAIL(RAMD_ERR_ARG, "null vector handle"); \
    } while(0)

#define CHECK_SAME(v, x)                                                                     \
    do                                                                                       \
    {                                                                                        \
        CHECK_VEC(v);                                                                        \
        CHECK_VEC(x);                                                                        \
        if((v)->dtype != (x)->dtype)                                                         \
            RAMD_FAIL(RAMD_ERR_ARG, "vector value types differ");                            \
        if((v)->n != (x)->n)                                                                 \
            RAMD_FAIL(RAMD_ERR_ARG, "vector sizes differ (the reference asserts size_ ==)"); \
    } while(0)

#define DISPATCH_FP(v, CALL)                                         \
    do                                                               \
    {                                                                \
        if((v)->dtype == RAMD_F64)                                   \
        {                                                            \
            using T = double;                                        \
            CALL;                                                    \
        }                                                            \
        else if((v)->dtype == RAMD_F32)                              \
        {                                                            \
            using T = float;                                         \
            CALL;                                                    \
        }                                                            \
        else                                                         \
            RAMD_FAIL(RAMD_ERR_ARG, "operation needs a real vector"); \
    } while(0)

static size_t dtype_size(int dtype)
{
    return dtype == RAMD_F64 ? 8 : 4;
}

// ---- blocking scalar reductions: one launch + one stream sync (the reference: rocBLAS call +
// hipStreamSynchronize, hip_vector.cpp:785-931)
template <typename T, int MODE>
static int reduce_blocking(int64_t n, const T* a, const T* b, int op, double* result)
{
    Backend& bk = backend();
    if(n <= 0)
    {
        *result = 0.0;
        return RAMD_OK;
    }
    const int slot = kScalarSlots - 1; // scratch slot of the blocking API
    hipLaunchKernelGGL((k_reduce<T, MODE>), dim3(reduce_grid((n + Pack<T>::N - 1) / Pack<T>::N)),
                       dim3(kBlock), 0, bk.cur, n, a, b, reduce_ctx(), slot, op);
    RAMD_HIP(hipGetLastError());
    return ramd_scalars_fetch(result, slot, 1);
}

extern "C" {

int ramd_vec_create(int dtype, ramd_vec_t* out)
{
    RAMD_TRY(ensure_init());
    if(!out || (dtype != RAMD_F64 && dtype != RAMD_F32 && dtype != RAMD_I32))
        RAMD_FAIL(RAMD_ERR_ARG, "bad dtype / null output");
    ramd_vec_s* v = new ramd_vec_s;
    v->dtype      = dtype;
    *out          = v;
    return RAMD_OK;
}

int ramd_vec_clear(ramd_vec_t v)
{
    CHECK_VEC(v);
    if(v->d)
        (void)cached_free(v->d);
    v->d = nullptr;
    v->n = 0;
    return RAMD_OK;
}

int ramd_vec_destroy(ramd_vec_t v)
{
    if(!v)
        return RAMD_OK;
    ramd_vec_clear(v);
    delete v;
    return RAMD_OK;
}

int ramd_vec_allocate(ramd_vec_t v, int64_t n)
{
    CHECK_VEC(v);
    if(n < 0)
        RAMD_FAIL(RAMD_ERR_ARG, "negative size");
    RAMD_TRY(ramd_vec_clear(v));
    if(n > 0)
    {
        size_t bytes = (size_t)n * dtype_size(v->dtype);
        RAMD_HIP(cached_malloc(&v->d, bytes + kPad));
        RAMD_HIP(hipMemsetAsync(v->d, 0, bytes + kPad, backend().cur));
        v->n = n;
    }
    return RAMD_OK;
}

int ramd_vec_allocate_apart(ramd_vec_t v, int64_t n, ramd_vec_t other)
{
    CHECK_VEC(v);
    if(n < 0)
        RAMD_FAIL(RAMD_ERR_ARG, "negative size");
    if(other == v)
        other = nullptr;
    RAMD_TRY(ramd_vec_clear(v));
    if(n > 0)
    {
        size_t bytes = (size_t)n * dtype_size(v->dtype);
        void*  q     = nullptr;
        RAMD_HIP(cached_malloc_apart(&q, bytes + kPad, other ? other->d : nullptr));
        v->d = q;
        RAMD_HIP(hipMemsetAsync(v->d, 0, bytes + kPad, backend().cur));
        v->n = n;
    }
    return RAMD_OK;
}

namespace
{
double g_placement_seconds = 0.0; // wall time spent measuring placements (ramd_placement_seconds)
struct PlacementClock
{
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~PlacementClock()
    {
        g_placement_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};
} // namespace

int ramd_placement_seconds(double* seconds, int reset)
{
    if(seconds)
        *seconds = g_placement_seconds;
    if(reset)
        g_placement_seconds = 0.0;
    return RAMD_OK;
}

int ramd_placement_room(int64_t bytes, int blocks, int* ok)
{
    if(!ok || bytes < 0 || blocks < 0)
        RAMD_FAIL(RAMD_ERR_ARG, "placement_room: bad arguments");
    *ok = placement_room((size_t)bytes + kPad, blocks) ? 1 : 0;
    return RAMD_OK;
}

int ramd_vec_place_apart(ramd_vec_t v, ramd_vec_t other, int* moved)
{
    CHECK_VEC(v);
    PlacementClock clock;
    if(moved)
        *moved = 0;
    if(!other || other == v || !v->d || !other->d || v->n != other->n || v->dtype != other->dtype)
        return RAMD_OK;
    const size_t bytes = (size_t)v->n * dtype_size(v->dtype);
    static const bool off = getenv("RAMD_ALLOC_CLASSES") && atoi(getenv("RAMD_ALLOC_CLASSES")) == 0;
    if(off || bytes < ((size_t)64 << 20))
        return RAMD_OK;
    // how the two blocks get along as the outputs of one kernel is MEASURED (a write pass over both, ~0.3 ms per GiB):
    // the vector keeps its block unless one of a few fresh candidates -- drawn from the other placement class first -- is
    // clearly faster with `other`; the contents move with it.
    const size_t pb = bytes & ~(size_t)4095;
    // (scratch + keep + one candidate: placement is an optimisation -- without room for it the vector stays where it is)
    if(!placement_room(bytes + kPad, 3))
        return RAMD_OK;
    void*        scratch = nullptr; // `other` is in use: probe against a copy of nothing -- its block is written, so save it
    if(cached_malloc_bytes(&scratch, bytes + kPad) != hipSuccess)
    {
        (void)hipGetLastError();
        return RAMD_OK;
    }
    void* keep = nullptr; // ... and the vector's own contents
    if(cached_malloc_bytes(&keep, bytes + kPad) != hipSuccess)
    {
        (void)hipGetLastError();
        (void)cached_free(scratch);
        return RAMD_OK;
    }
    RAMD_HIP(hipMemcpyAsync(scratch, other->d, bytes, hipMemcpyDeviceToDevice, backend().cur));
    (void)hipMemcpyAsync(keep, v->d, bytes, hipMemcpyDeviceToDevice, backend().cur);
    (void)hipStreamSynchronize(backend().cur);
    float              best_ms = probe_write_pair_ms(v->d, other->d, pb);
    void*              best    = v->d;
    std::vector<void*> losers;
    static const bool  verbose = getenv("RAMD_ALLOC_VERBOSE") != nullptr;
    if(verbose)
        fprintf(stderr, "place apart: current pair %.4f ms", best_ms);
    float worst_ms = best_ms;
    // (pairs come in two speeds, ~8-15 % apart: the search ends as soon as both have been seen and the fast one is held)
    for(int k = 0; k < 5 && worst_ms < 1.07f * best_ms; ++k) // (each candidate is a fresh allocation: 1 ... 100 ms per GiB)
    {
        void* c = nullptr;
        if(!placement_room(bytes + kPad, 1) || cached_malloc_apart(&c, bytes + kPad, other->d) != hipSuccess)
        {
            (void)hipGetLastError();
            break;
        }
        const float ms = probe_write_pair_ms(c, other->d, pb);
        if(verbose)
            fprintf(stderr, ", candidate %.4f", ms);
        if(ms > worst_ms)
            worst_ms = ms;
        if(ms > 0.f && ms < 0.96f * best_ms)
        {
            if(best != v->d)
                losers.push_back(best);
            best    = c;
            best_ms = ms;
        }
        else
            losers.push_back(c);
    }
    if(verbose)
        fprintf(stderr, " -> %s (%.4f ms)\n", best == v->d ? "kept" : "moved", best_ms);
    // the probes wrote zeros over both blocks: restore
    (void)hipMemcpyAsync(other->d, scratch, bytes, hipMemcpyDeviceToDevice, backend().cur);
    if(best != v->d)
    {
        (void)hipMemsetAsync((char*)best + bytes, 0, kPad, backend().cur);
        losers.push_back(v->d);
        v->d = best;
        if(moved)
            *moved = 1;
    }
    (void)hipMemcpyAsync(v->d, keep, bytes, hipMemcpyDeviceToDevice, backend().cur);
    (void)hipStreamSynchronize(backend().cur);
    for(void* l : losers)
        (void)cached_free(l);
    (void)cached_free(scratch);
    (void)cached_free(keep);
    return RAMD_OK;
}

// Placement by trial: `run` launches the kernels that use the vector (on the current stream, any number of them); it is
// timed with the vector in its own block and in up to `tries` fresh ones, and the vector moves to the fastest (contents
// kept).  `run` must not contain a collective (a rank that is short of memory tries fewer candidates than the others).
// stop_ratio = 0: up to `tries` trials.  stop_ratio in (0, 1): the kernels run at one of a few discrete speeds, and
// the search ends as soon as the best time is below stop_ratio x the worst one seen (a fresh GiB costs up to 100 ms).
int ramd_vec_place_by_trial(ramd_vec_t v, ramd_trial_cb run, void* ctx, int tries, double stop_ratio, ramd_vec_t apart_from,
                            int* moved)
{
    CHECK_VEC(v);
    PlacementClock clock;
    if(moved)
        *moved = 0;
    if(!run || tries < 1 || !v->d)
        return RAMD_OK;
    const size_t      bytes = (size_t)v->n * dtype_size(v->dtype);
    static const bool off   = getenv("RAMD_ALLOC_CLASSES") && atoi(getenv("RAMD_ALLOC_CLASSES")) == 0;
    if(off || bytes < ((size_t)64 << 20))
        return RAMD_OK;
    static const bool verbose = getenv("RAMD_ALLOC_VERBOSE") != nullptr;
    hipStream_t       st      = backend().cur;
    hipEvent_t        e0, e1;
    RAMD_HIP(hipEventCreate(&e0));
    RAMD_HIP(hipEventCreate(&e1));
    void* const own  = v->d;
    void*       keep = nullptr;
    // (the saved contents + one candidate; every further candidate is asked for again below -- losers stay allocated
    //  until the end so that a draw never returns a block already seen)
    if(!placement_room(bytes + kPad, 2) || cached_malloc_bytes(&keep, bytes + kPad) != hipSuccess)
    {
        (void)hipGetLastError();
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return RAMD_OK; // no room to try: stay
    }
    (void)hipMemcpyAsync(keep, own, bytes, hipMemcpyDeviceToDevice, st);
    int  rc        = RAMD_OK;
    auto time_with = [&](void* block, float* ms) { // one warm run, then the faster of two
        v->d       = block;
        float best = 1e30f;
        for(int rep = 0; rep < 3 && rc == RAMD_OK; ++rep)
        {
            (void)hipEventRecord(e0, st);
            rc = run(ctx);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            if(rep > 0 && t < best)
                best = t;
        }
        v->d = own;
        *ms  = best;
    };
    float best_ms = 0.f;
    time_with(own, &best_ms);
    float              worst_ms = best_ms;
    void*              best     = own;
    std::vector<void*> losers;
    if(verbose)
        fprintf(stderr, "place by trial: own block %.4f ms", best_ms);
    for(int k = 0; k < tries && rc == RAMD_OK; ++k)
    {
        if(stop_ratio > 0.0 && best_ms < (float)stop_ratio * worst_ms)
            break;
        void* c = nullptr;
        // (apart_from: the first candidates come from the placement class that vector's block is NOT in -- fresh blocks of
        //  one process tend to share a class, and for two vectors a kernel writes the other class is the likely fast one)
        const hipError_t ea = !placement_room(bytes + kPad, 1)
                                  ? hipErrorOutOfMemory
                                  : ((apart_from && apart_from->d && k < 3) ? cached_malloc_apart(&c, bytes + kPad, apart_from->d)
                                                                            : cached_malloc_bytes(&c, bytes + kPad));
        if(ea != hipSuccess)
        {
            (void)hipGetLastError();
            c = nullptr;
        }
        float ms = 1e30f;
        if(c)
        {
            (void)hipMemcpyAsync(c, keep, bytes, hipMemcpyDeviceToDevice, st);
            (void)hipMemsetAsync((char*)c + bytes, 0, kPad, st);
            time_with(c, &ms);
        }
        else
            time_with(own, &ms); // (keeps the number of runs the same on every rank)
        if(verbose)
            fprintf(stderr, ", candidate %.4f", ms);
        if(ms < 1e29f && ms > worst_ms)
            worst_ms = ms;
        if(c && rc == RAMD_OK && ms < 0.995f * best_ms)
        {
            if(best != own)
                losers.push_back(best);
            best    = c;
            best_ms = ms;
        }
        else if(c)
            losers.push_back(c);
    }
    if(verbose)
        fprintf(stderr, " -> %s (%.4f ms)\n", best == own ? "kept" : "moved", best_ms);
    if(best != own)
    {
        losers.push_back(own);
        v->d = best;
        if(moved)
            *moved = 1;
    }
    (void)hipMemcpyAsync(v->d, keep, bytes, hipMemcpyDeviceToDevice, st);
    (void)hipStreamSynchronize(st);
    for(void* l : losers)
        (void)cached_free(l);
    (void)cached_free(keep);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int ramd_vec_placement_class(ramd_vec_t v, int* cls)
{
    CHECK_VEC(v);
    if(!cls)
        RAMD_FAIL(RAMD_ERR_ARG, "null output");
    *cls = v->d ? cached_block_class(v->d) : -1;
    return RAMD_OK;
}

int ramd_vec_size(ramd_vec_t v, int64_t* n)
{
    CHECK_VEC(v);
    *n = v->n;
    return RAMD_OK;
}
int ramd_vec_dtype(ramd_vec_t v, int* dtype)
{
    CHECK_VEC(v);
    *dtype = v->dtype;
    return RAMD_OK;
}
void* ramd_vec_data(ramd_vec_t v)
{
    return v ? v->d : nullptr;
}

int ramd_vec_set_values(ramd_vec_t v, double val)
{
    CHECK_VEC(v);
    if(v->dtype == RAMD_F64)
        return launch_map<double>(v->n, FSet<double>{(double*)v->d, val});
    if(v->dtype == RAMD_F32)
        return launch_map<float>(v->n, FSet<float>{(float*)v->d, (float)val});
    return launch_map<int>(v->n, FSet<int>{(int*)v->d, (int)val});
}
int ramd_vec_zeros(ramd_vec_t v)
{
    CHECK_VEC(v);
    if(v->n > 0)
        RAMD_HIP(hipMemsetAsync(v->d, 0, (size_t)v->n * dtype_size(v->dtype), backend().cur));
    return RAMD_OK;
}
int ramd_vec_ones(ramd_vec_t v)
{
    return ramd_vec_set_values(v, 1.0);
}

int ramd_vec_copy_from_host(ramd_vec_t v, const void* host)
{
    CHECK_VEC(v);
    if(v->n > 0)
    {
        RAMD_HIP(hipMemcpyAsync(v->d, host, (size_t)v->n * dtype_size(v->dtype), hipMemcpyHostToDevice,
                                backend().cur));
        RAMD_HIP(hipStreamSynchronize(backend().cur));
    }
    return RAMD_OK;
}
int ramd_vec_copy_to_host(ramd_vec_t v, void* host)
{
    CHECK_VEC(v);
    if(v->n > 0)
    {
        RAMD_HIP(hipMemcpyAsync(host, v->d, (size_t)v->n * dtype_size(v->dtype), hipMemcpyDeviceToHost,
                                backend().cur));
        RAMD_HIP(hipStreamSynchronize(backend().cur));
    }
    return RAMD_OK;
}

int ramd_vec_copy_from(ramd_vec_t v, ramd_vec_t src)
{
    CHECK_VEC(v);
    CHECK_VEC(src);
    if(v->dtype != src->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "vector value types differ");
    if(v == src)
        return RAMD_OK;
    if(v->n != src->n) // hip_vector.cpp CopyFrom: allocate when the destination is empty
    {
        if(v->n != 0)
            RAMD_FAIL(RAMD_ERR_ARG, "CopyFrom: sizes differ (the reference asserts)");
        RAMD_TRY(ramd_vec_allocate(v, src->n));
    }
    if(v->n > 0)
        RAMD_HIP(hipMemcpyAsync(v->d, src->d, (size_t)v->n * dtype_size(v->dtype),
                                hipMemcpyDeviceToDevice, backend().cur));
    return RAMD_OK;
}

int ramd_vec_copy_from_offset(ramd_vec_t v, ramd_vec_t src, int64_t so, int64_t dof, int64_t size)
{
    CHECK_VEC(v);
    CHECK_VEC(src);
    if(v->dtype != src->dtype || so < 0 || dof < 0 || size < 0 || so + size > src->n
       || dof + size > v->n)
        RAMD_FAIL(RAMD_ERR_ARG, "CopyFrom(offset): range out of bounds");
    if(size > 0)
    {
        size_t es = dtype_size(v->dtype);
        RAMD_HIP(hipMemcpyAsync((char*)v->d + dof * es, (const char*)src->d + so * es, size * es,
                                hipMemcpyDeviceToDevice, backend().cur));
    }
    return RAMD_OK;
}

int ramd_vec_copy_from_float(ramd_vec_t v, ramd_vec_t src)
{
    CHECK_VEC(v);
    CHECK_VEC(src);
    if(v->dtype != RAMD_F64 || src->dtype != RAMD_F32)
        RAMD_FAIL(RAMD_ERR_ARG, "CopyFromFloat: need f64 <- f32");
    if(v->n != src->n)
        RAMD_TRY(ramd_vec_allocate(v, src->n));
    if(v->n > 0)
    {
        hipLaunchKernelGGL((k_cast<double, float>), dim3(ew_grid(v->n)), dim3(kBlock), 0, backend().cur,
                           v->n, (double*)v->d, (const float*)src->d);
        RAMD_HIP(hipGetLastError());
    }
    return RAMD_OK;
}
int ramd_vec_copy_from_double(ramd_vec_t v, ramd_vec_t src)
{
    CHECK_VEC(v);
    CHECK_VEC(src);
    if(v->dtype != RAMD_F32 || src->dtype != RAMD_F64)
        RAMD_FAIL(RAMD_ERR_ARG, "CopyFromDouble: need f32 <- f64");
    if(v->n != src->n)
        RAMD_TRY(ramd_vec_allocate(v, src->n));
    if(v->n > 0)
    {
        hipLaunchKernelGGL((k_cast<float, double>), dim3(ew_grid(v->n)), dim3(kBlock), 0, backend().cur,
                           v->n, (float*)v->d, (const double*)src->d);
        RAMD_HIP(hipGetLastError());
    }
    return RAMD_OK;
}

static int check_perm(ramd_vec_t v, ramd_vec_t src, ramd_vec_t perm)
{
    CHECK_SAME(v, src);
    CHECK_VEC(perm);
    if(perm->dtype != RAMD_I32 || perm->n != v->n)
        RAMD_FAIL(RAMD_ERR_ARG, "permutation must be an int32 vector of the same size");
    if(v == src)
        RAMD_FAIL(RAMD_ERR_ARG, "CopyFromPermute: this == &src (the reference asserts)");
    return RAMD_OK;
}

int ramd_vec_copy_from_permute(ramd_vec_t v, ramd_vec_t src, ramd_vec_t perm)
{
    RAMD_TRY(check_perm(v, src, perm));
    if(v->n == 0)
        return RAMD_OK;
    DISPATCH_FP(v, hipLaunchKernelGGL((k_permute_fwd<T>), dim3(ew_grid(v->n)), dim3(kBlock), 0,
                                      backend().cur, v->n, (T*)v->d, (const T*)src->d,
                                      (const int*)perm->d));
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}
int ramd_vec_copy_from_permute_backward(ramd_vec_t v, ramd_vec_t src, ramd_vec_t perm)
{
    RAMD_TRY(check_perm(v, src, perm));
    if(v->n == 0)
        return RAMD_OK;
    DISPATCH_FP(v, hipLaunchKernelGGL((k_permute_bwd<T>), dim3(ew_grid(v->n)), dim3(kBlock), 0,
                                      backend().cur, v->n, (T*)v->d, (const T*)src->d,
                                      (const int*)perm->d));
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_vec_get_index_values(ramd_vec_t v, ramd_vec_t index, ramd_vec_t out)
{
    CHECK_VEC(v);
    CHECK_VEC(index);
    CHECK_VEC(out);
    if(index->dtype != RAMD_I32 || out->dtype != v->dtype || out->n != index->n)
        RAMD_FAIL(RAMD_ERR_ARG, "GetIndexValues: bad index/output vector");
    if(out->n == 0)
        return RAMD_OK;
    DISPATCH_FP(v, hipLaunchKernelGGL((k_gather<T>), dim3(ew_grid(out->n)), dim3(kBlock), 0,
                                      backend().cur, out->n, (T*)out->d, (const T*)v->d,
                                      (const int*)index->d));
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_vec_add_scale(ramd_vec_t v, ramd_vec_t x, double alpha)
{
    CHECK_SAME(v, x);
    DISPATCH_FP(v, return launch_map<T>(v->n, FAddScale<T>{(T*)v->d, (const T*)x->d, (T)alpha}));
}
int ramd_vec_scale_add(ramd_vec_t v, double alpha, ramd_vec_t x)
{
    CHECK_SAME(v, x);
    DISPATCH_FP(v, return launch_map<T>(v->n, FScaleAdd<T>{(T*)v->d, (const T*)x->d, (T)alpha}));
}
int ramd_vec_scale_add_scale(ramd_vec_t v, double alpha, ramd_vec_t x, double beta)
{
    CHECK_SAME(v, x);
    DISPATCH_FP(
        v, return launch_map<T>(v->n, FScaleAddScale<T>{(T*)v->d, (const T*)x->d, (T)alpha, (T)beta}));
}
int ramd_vec_scale_add_scale_offset(ramd_vec_t v, double alpha, ramd_vec_t x, double beta, int64_t src_offset,
                                    int64_t dst_offset, int64_t size)
{
    CHECK_VEC(v);
    CHECK_VEC(x);
    if(v->dtype != x->dtype || size < 0 || src_offset < 0 || dst_offset < 0 || src_offset + size > x->n
       || dst_offset + size > v->n)
        RAMD_FAIL(RAMD_ERR_ARG, "ScaleAddScale(offsets): range outside the vectors / type mismatch");
    if(size == 0)
        return RAMD_OK;
    const int grid = ew_grid(size);
    if(v->dtype == RAMD_F64)
        hipLaunchKernelGGL((k_scale_add_scale_offset<double>), dim3(grid), dim3(kBlock), 0, backend().cur, size,
                           (double*)v->d + dst_offset, (const double*)x->d + src_offset, alpha, beta);
    else if(v->dtype == RAMD_F32)
        hipLaunchKernelGGL((k_scale_add_scale_offset<float>), dim3(grid), dim3(kBlock), 0, backend().cur, size,
                           (float*)v->d + dst_offset, (const float*)x->d + src_offset, (float)alpha, (float)beta);
    else
        RAMD_FAIL(RAMD_ERR_ARG, "ScaleAddScale(offsets): real vectors expected");
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}
int ramd_vec_scale_add2(ramd_vec_t v, double alpha, ramd_vec_t x, double beta, ramd_vec_t y, double gamma)
{
    CHECK_SAME(v, x);
    CHECK_SAME(v, y);
    DISPATCH_FP(v, return launch_map<T>(v->n, FScaleAdd2<T>{(T*)v->d, (const T*)x->d, (const T*)y->d,
                                                            (T)alpha, (T)beta, (T)gamma}));
}
int ramd_vec_scale(ramd_vec_t v, double alpha)
{
    CHECK_VEC(v);
    DISPATCH_FP(v, return launch_map<T>(v->n, FScale<T>{(T*)v->d, (T)alpha}));
}
int ramd_vec_pointwise_mult(ramd_vec_t v, ramd_vec_t x)
{
    CHECK_SAME(v, x);
    DISPATCH_FP(v, return launch_map<T>(v->n, FPointWise<T>{(T*)v->d, (const T*)v->d, (const T*)x->d}));
}
int ramd_vec_pointwise_mult2(ramd_vec_t v, ramd_vec_t x, ramd_vec_t y)
{
    CHECK_SAME(v, x);
    CHECK_SAME(v, y);
    // host_vector.cpp:1257-1279: this = y * x
    DISPATCH_FP(v, return launch_map<T>(v->n, FPointWise<T>{(T*)v->d, (const T*)y->d, (const T*)x->d}));
}

int ramd_vec_dot(ramd_vec_t v, ramd_vec_t x, double* result)
{
    CHECK_SAME(v, x);
    DISPATCH_FP(v, return (reduce_blocking<T, 0>(v->n, (const T*)v->d, (const T*)x->d, RED_SUM, result)));
}
int ramd_vec_norm(ramd_vec_t v, double* result)
{
    CHECK_VEC(v);
    DISPATCH_FP(v, return (reduce_blocking<T, 0>(v->n, (const T*)v->d, (const T*)v->d, RED_SQRT, result)));
}
int ramd_vec_reduce(ramd_vec_t v, double* result)
{
    CHECK_VEC(v);
    DISPATCH_FP(v, return (reduce_blocking<T, 1>(v->n, (const T*)v->d, (const T*)v->d, RED_SUM, result)));
}
int ramd_vec_asum(ramd_vec_t v, double* result)
{
    CHECK_VEC(v);
    DISPATCH_FP(v, return (reduce_blocking<T, 2>(v->n, (const T*)v->d, (const T*)v->d, RED_SUM, result)));
}

int ramd_vec_amax(ramd_vec_t v, double* value, int64_t* index)
{
    CHECK_VEC(v);
    if(v->n <= 0)
    {
        *value = 0.0;
        *index = 0;
        return RAMD_OK;
    }
    Backend&   bk   = backend();
    int        grid = reduce_grid(v->n);
    double*    dval = nullptr;
    long long* didx = nullptr;
    RAMD_HIP(cached_malloc((void**)&dval, sizeof(double) * grid));
    RAMD_HIP(cached_malloc((void**)&didx, sizeof(long long) * grid));
    DISPATCH_FP(v, hipLaunchKernelGGL((k_amax_partial<T>), dim3(grid), dim3(kBlock), 0, bk.cur, v->n,
                                      (const T*)v->d, dval, didx));
    double*    hv = (double*)malloc(sizeof(double) * grid);
    long long* hi = (long long*)malloc(sizeof(long long) * grid);
    hipError_t e1 = hipMemcpyAsync(hv, dval, sizeof(double) * grid, hipMemcpyDeviceToHost, bk.cur);
    hipError_t e2 = hipMemcpyAsync(hi, didx, sizeof(long long) * grid, hipMemcpyDeviceToHost, bk.cur);
    hipError_t e3 = hipStreamSynchronize(bk.cur);
    double     best = -1.0;
    long long  bi   = 0;
    for(int b = 0; b < grid; ++b)
        if(hv[b] > best || (hv[b] == best && hi[b] < bi))
        {
            best = hv[b];
            bi   = hi[b];
        }
    free(hv);
    free(hi);
    (void)cached_free(dval);
    (void)cached_free(didx);
    RAMD_HIP(e1);
    RAMD_HIP(e2);
    RAMD_HIP(e3);
    *value = best;
    *index = bi;
    return RAMD_OK;
}

} // extern "C"
