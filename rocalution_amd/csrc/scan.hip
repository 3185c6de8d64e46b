// scan.hip -- device exclusive scan / max (own implementation; the reference uses rocPRIM for the
// same setup steps, e.g. src/base/hip/hip_conversion.cpp:584-680).  Setup-time only.
#include "device_utils.hpp"
#include "matrix_impl.hpp"

#include <cstdlib>

#include <utility>
#include <vector>

namespace ramd
{

constexpr int kScanItems = 8; // per thread
constexpr int kScanTile  = kBlock * kScanItems;

// tile-local exclusive scan; tile totals go to tsum[blockIdx.x]
__global__ __launch_bounds__(kBlock) void k_scan_tiles(const int* __restrict__ in,
                                                       int* __restrict__ out, int64_t n,
                                                       int* __restrict__ tsum)
{
    __shared__ int wsum[4];
    const int64_t  base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int            v[kScanItems];
    int            tot = 0;
#pragma unroll
    for(int k = 0; k < kScanItems; ++k)
    {
        v[k] = (base + k < n) ? in[base + k] : 0;
        tot += v[k];
    }
    // inclusive scan of per-thread totals inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int       inc  = tot;
#pragma unroll
    for(int off = 1; off < 64; off <<= 1)
    {
        int t = __shfl_up(inc, off, 64);
        if(lane >= off)
            inc += t;
    }
    if(lane == 63)
        wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for(int w = 0; w < wave; ++w)
        woff += wsum[w];
    int run = woff + inc - tot;
#pragma unroll
    for(int k = 0; k < kScanItems; ++k)
    {
        if(base + k < n)
            out[base + k] = run;
        run += v[k];
    }
    if(threadIdx.x == kBlock - 1 && tsum)
        tsum[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(kBlock) void k_scan_add(int* __restrict__ out, int64_t n,
                                                     const int* __restrict__ toff)
{
    const int     add  = toff[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    for(int k = threadIdx.x; k < kScanTile; k += kBlock)
        if(base + k < n)
            out[base + k] += add;
}

int device_exclusive_scan(const int* in, int* out, int64_t n)
{
    if(n <= 0)
        return RAMD_OK;
    Backend&      b      = backend();
    const int64_t ntiles = (n + kScanTile - 1) / kScanTile;
    if(ntiles == 1)
    {
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(kBlock), 0, b.cur, in, out, n, (int*)nullptr);
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    int* tsum = nullptr;
    RAMD_TRY(dev_alloc(&tsum, ntiles));
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(kBlock), 0, b.cur, in, out, n, tsum);
    int s = device_exclusive_scan(tsum, tsum, ntiles);
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)ntiles), dim3(kBlock), 0, b.cur, out, n, tsum);
        if(hipGetLastError() != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    // tsum is freed after the queued kernels ran (hipFree synchronises)
    dev_free(&tsum);
    return s;
}

// ---- stable LSD radix sort by key, one bit per pass, built on the scan above (setup-time only).
// Used to order rows by dependency level while keeping ascending row order inside a level, which makes
// the level-ordered triangular solve poll and gather contiguous memory.
__global__ __launch_bounds__(kBlock) void k_sort_flag(int64_t n, const int* __restrict__ keys, int bit,
                                                      int* __restrict__ flag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gsz)
        flag[i] = (i < n && ((keys[i] >> bit) & 1) == 0) ? 1 : 0;
}
__global__ __launch_bounds__(kBlock) void k_sort_scatter(int64_t n, const int* __restrict__ keys,
                                                         const int* __restrict__ vals, int bit,
                                                         const int* __restrict__ pos0,
                                                         int* __restrict__ okeys, int* __restrict__ ovals)
{
    const int64_t gsz   = (int64_t)gridDim.x * blockDim.x;
    const int     zeros = pos0[n];
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
    {
        const int k = keys[i];
        const int d = (((k >> bit) & 1) == 0) ? pos0[i] : zeros + (int)(i - pos0[i]);
        okeys[d]    = k;
        ovals[d]    = vals ? vals[i] : (int)i;
    }
}

// ---- stable LSD radix sort by key, 4 bits per pass: per-tile digit histograms (bin-major) -> one scan -> scatter with
// the in-tile rank of every element among the equal digits (thread-private LDS counters, blocked arrangement, so
// ranks follow the input order).  A pass moves ~20 bytes per element; 21-bit keys take 6 passes instead of 21.
constexpr int kRadixItems = 8;
constexpr int kRadixTile  = kBlock * kRadixItems;
__global__ __launch_bounds__(kBlock) void k_radix_hist(int64_t n, const int* __restrict__ keys, int shift,
                                                       int nblocks, int* __restrict__ hist)
{
    __shared__ int h[16];
    if(threadIdx.x < 16)
        h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kRadixTile + (int64_t)threadIdx.x * kRadixItems;
#pragma unroll
    for(int e = 0; e < kRadixItems; ++e)
        if(base + e < n)
            atomicAdd(&h[(keys[base + e] >> shift) & 15], 1);
    __syncthreads();
    if(threadIdx.x < 16)
        hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(kBlock) void k_radix_scatter(int64_t n, const int* __restrict__ keys,
                                                          const int* __restrict__ vals, int shift, int nblocks,
                                                          const int* __restrict__ hist, int* __restrict__ okeys,
                                                          int* __restrict__ ovals)
{
    __shared__ int cnt[16][kBlock];
    const int tid = threadIdx.x;
#pragma unroll
    for(int d = 0; d < 16; ++d)
        cnt[d][tid] = 0;
    const int64_t base = (int64_t)blockIdx.x * kRadixTile + (int64_t)tid * kRadixItems;
    int           k[kRadixItems];
#pragma unroll
    for(int e = 0; e < kRadixItems; ++e)
    {
        k[e] = (base + e < n) ? keys[base + e] : 0;
        if(base + e < n)
            cnt[(k[e] >> shift) & 15][tid] += 1; // my own column: no conflicts
    }
    __syncthreads();
    if(tid < 16) // exclusive prefix over the threads, one digit per thread
    {
        int run = 0;
        for(int t = 0; t < kBlock; ++t)
        {
            const int c = cnt[tid][t];
            cnt[tid][t] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for(int e = 0; e < kRadixItems; ++e)
        if(base + e < n)
        {
            const int d   = (k[e] >> shift) & 15;
            const int r   = cnt[d][tid]++;
            const int pos = hist[(int64_t)d * nblocks + blockIdx.x] + r;
            okeys[pos]    = k[e];
            ovals[pos]    = vals ? vals[base + e] : (int)(base + e);
        }
}

static int device_stable_sort_radix4(const int* keys, int64_t n, int max_key, int* order_out)
{
    Backend& b    = backend();
    int      bits = 0;
    while((1ll << bits) <= (long long)max_key)
        ++bits;
    const int64_t nblk64 = (n + kRadixTile - 1) / kRadixTile;
    if(nblk64 * 16 + 1 >= 0x7fffffffLL)
        return RAMD_ERR_UNSUPPORTED;
    const int nblocks = (int)nblk64;
    int *     ka = nullptr, *kb = nullptr, *va = nullptr, *vb = nullptr, *hist = nullptr;
    int       s = dev_alloc(&ka, n);
    if(s == RAMD_OK)
        s = dev_alloc(&kb, n);
    if(s == RAMD_OK)
        s = dev_alloc(&va, n);
    if(s == RAMD_OK)
        s = dev_alloc(&vb, n);
    if(s == RAMD_OK)
        s = dev_alloc(&hist, (int64_t)nblocks * 16 + 1);
    if(s == RAMD_OK && hipMemcpyAsync(ka, keys, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur) != hipSuccess)
        s = RAMD_ERR_HIP;
    bool first = true;
    for(int shift = 0; shift < bits && s == RAMD_OK; shift += 4)
    {
        hipLaunchKernelGGL(k_radix_hist, dim3((unsigned)nblocks), dim3(kBlock), 0, b.cur, n, (const int*)ka, shift,
                           nblocks, hist);
        s = device_exclusive_scan(hist, hist, (int64_t)nblocks * 16 + 1);
        if(s != RAMD_OK)
            break;
        hipLaunchKernelGGL(k_radix_scatter, dim3((unsigned)nblocks), dim3(kBlock), 0, b.cur, n, (const int*)ka,
                           first ? (const int*)nullptr : (const int*)va, shift, nblocks, (const int*)hist, kb, vb);
        std::swap(ka, kb);
        std::swap(va, vb);
        first = false;
    }
    if(s == RAMD_OK)
    {
        hipError_t e = hipSuccess;
        if(first) // all keys zero: identity order
        {
            std::vector<int> iota((size_t)n);
            for(int64_t i = 0; i < n; ++i)
                iota[(size_t)i] = (int)i;
            e = hipMemcpy(order_out, iota.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice);
        }
        else
            e = hipMemcpyAsync(order_out, va, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur);
        if(e == hipSuccess)
            e = hipGetLastError();
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&ka);
    dev_free(&kb);
    dev_free(&va);
    dev_free(&vb);
    dev_free(&hist);
    return s;
}

static int device_stable_sort_bitwise(const int* keys, int64_t n, int max_key, int* order_out);

int device_stable_sort_by_key(const int* keys, int64_t n, int max_key, int* order_out)
{
    if(n <= 0)
        return RAMD_OK;
    static int bitwise = -1; // RAMD_SORT_BITWISE=1: the one-bit-per-pass variant (A/B experiments)
    if(bitwise < 0)
    {
        const char* e = getenv("RAMD_SORT_BITWISE");
        bitwise       = e ? atoi(e) : 0;
    }
    if(!bitwise)
    {
        int s = device_stable_sort_radix4(keys, n, max_key, order_out);
        if(s != RAMD_ERR_UNSUPPORTED)
            return s;
    }
    return device_stable_sort_bitwise(keys, n, max_key, order_out);
}

static int device_stable_sort_bitwise(const int* keys, int64_t n, int max_key, int* order_out)
{
    if(n <= 0)
        return RAMD_OK;
    Backend& b    = backend();
    int      bits = 0;
    while((1ll << bits) <= (long long)max_key)
        ++bits;
    int *ka = nullptr, *kb = nullptr, *va = nullptr, *vb = nullptr, *fl = nullptr;
    int  s = dev_alloc(&ka, n);
    if(s == RAMD_OK)
        s = dev_alloc(&kb, n);
    if(s == RAMD_OK)
        s = dev_alloc(&va, n);
    if(s == RAMD_OK)
        s = dev_alloc(&vb, n);
    if(s == RAMD_OK)
        s = dev_alloc(&fl, n + 1);
    if(s == RAMD_OK && hipMemcpyAsync(ka, keys, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur) != hipSuccess)
        s = RAMD_ERR_HIP;
    const int grid   = ew_grid(n + 1);
    bool      first  = true;
    for(int bit = 0; bit < bits && s == RAMD_OK; ++bit)
    {
        hipLaunchKernelGGL(k_sort_flag, dim3(grid), dim3(kBlock), 0, b.cur, n, ka, bit, fl);
        s = device_exclusive_scan(fl, fl, n + 1);
        if(s != RAMD_OK)
            break;
        hipLaunchKernelGGL(k_sort_scatter, dim3(grid), dim3(kBlock), 0, b.cur, n, ka, first ? nullptr : va,
                           bit, fl, kb, vb);
        std::swap(ka, kb);
        std::swap(va, vb);
        first = false;
    }
    if(s == RAMD_OK)
    {
        if(first) // all keys equal: identity order
            hipLaunchKernelGGL(k_sort_scatter, dim3(grid), dim3(kBlock), 0, b.cur, (int64_t)0, ka, nullptr, 0,
                               fl, kb, vb);
        hipError_t e = hipSuccess;
        if(first)
        {
            std::vector<int> iota((size_t)n);
            for(int64_t i = 0; i < n; ++i)
                iota[(size_t)i] = (int)i;
            e = hipMemcpy(order_out, iota.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice);
        }
        else
            e = hipMemcpyAsync(order_out, va, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&ka);
    dev_free(&kb);
    dev_free(&va);
    dev_free(&vb);
    dev_free(&fl);
    return s;
}

__global__ __launch_bounds__(kBlock) void k_max_int(const int* __restrict__ in, int64_t n,
                                                    int* __restrict__ result)
{
    __shared__ int sm[kBlock];
    int            m   = INT_MIN;
    const int64_t  gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        m = max(m, in[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for(int s = kBlock / 2; s > 0; s >>= 1)
    {
        if((int)threadIdx.x < s)
            sm[threadIdx.x] = max(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if(threadIdx.x == 0)
        atomicMax(result, sm[0]);
}

int device_max_int(const int* in, int64_t n, int* result)
{
    Backend& b = backend();
    int*     d = nullptr;
    RAMD_TRY(dev_alloc(&d, 1));
    int init = INT_MIN;
    RAMD_HIP(hipMemcpyAsync(d, &init, sizeof(int), hipMemcpyHostToDevice, b.cur));
    if(n > 0)
        hipLaunchKernelGGL(k_max_int, dim3(reduce_grid(n)), dim3(kBlock), 0, b.cur, in, n, d);
    hipError_t e = hipMemcpyAsync(result, d, sizeof(int), hipMemcpyDeviceToHost, b.cur);
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&d);
    RAMD_HIP(e);
    return RAMD_OK;
}

} // namespace ramd
