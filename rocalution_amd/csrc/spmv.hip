// spmv.hip -- hand-written SpMV for CSR / ELL / HYB / COO on gfx950 (wave64, LDS-staged).
//
// Replaces the rocSPARSE csrmv/ellmv/coomv calls of the reference HIP backend
// (src/base/hip/hip_matrix_csr.cpp:1215-1292, hip_matrix_ell.cpp:590-670,
//  hip_matrix_hyb.cpp:695-835, hip_matrix_coo.cpp:632-700).  Results follow the HOST backend's
// arithmetic: every row is accumulated left to right in storage order
// (src/base/host/host_matrix_csr.cpp:718-734, :755-767; host_matrix_ell.cpp:296-321;
//  host_matrix_hyb.cpp:330-364; host_matrix_coo.cpp:368-376), so y is bit-identical to the
// OpenMP backend (products are formed as in the reference, no FMA contraction).
//
// CSR kernel ("stream" layout): a 256-thread workgroup owns 256 consecutive rows.  Its nnz range
// is contiguous in val/col, so the workgroup streams it with fully coalesced 16-byte loads
// (non-temporal: read once), gathers x through L1/L2 (x is the only reused operand and stays
// cache-resident thanks to the XCD-aware row mapping), parks the products in LDS, and then each
// thread adds up ITS row's products sequentially from LDS.  Bandwidth-bound: 12 B/nnz + 20 B/row.
#include "device_utils.hpp"
#include "matrix_impl.hpp"

namespace ramd
{

constexpr int kCsrRows  = 256; // rows per workgroup (one per thread)
constexpr int kCsrChunk = 2048; // products staged in LDS per pass (16 KiB fp64)

// XCD-aware persistent mapping: hardware places workgroup b on XCD b % 8 (observed, used for speed
// only).  Every XCD gets one contiguous eighth of the row blocks, and the (up to) 256 workgroups
// of an XCD walk that eighth block-cyclically -- the same order a plain launch would dispatch
// them in -- so rows that gather the same x planes are in flight on the same L2 at the same time.
// A persistent grid (<= kReduceBlocks workgroups) also gives the fused <x,y> reduction exactly one
// partial per workgroup.
constexpr int kCsrWgPerXcd = kReduceBlocks / 8;

template <typename T>
struct ValPack4; // 4 consecutive values as 16-byte packets
template <>
struct ValPack4<double>
{
    v2f64 a, b;
    __device__ __forceinline__ void load(const double* p)
    {
        a = nt_load(reinterpret_cast<const v2f64*>(p));
        b = nt_load(reinterpret_cast<const v2f64*>(p) + 1);
    }
    __device__ __forceinline__ double get(int e) const
    {
        return e == 0 ? a.x : e == 1 ? a.y : e == 2 ? b.x : b.y;
    }
};
template <>
struct ValPack4<float>
{
    v4f32 a;
    __device__ __forceinline__ void load(const float* p)
    {
        a = nt_load(reinterpret_cast<const v4f32*>(p));
    }
    __device__ __forceinline__ float get(int e) const
    {
        return e == 0 ? a.x : e == 1 ? a.y : e == 2 ? a.z : a.w;
    }
};

// MODE 0: y = A x      MODE 1: y += scalar * A x (term by term into y, as the host ApplyAdd)
// DOT: additionally reduce <x, y> into scalar slot `slot` (needs a square matrix)
template <typename T, int MODE, bool DOT>
__global__ __launch_bounds__(kBlock) void k_csr_stream(int nrow, int per_xcd, int wg_per_xcd,
                                                       const int* __restrict__ rp,
                                                       const int* __restrict__ ci,
                                                       const T* __restrict__ val,
                                                       const T* __restrict__ x, T* __restrict__ y,
                                                       T scalar, ReduceCtx ctx, int slot)
{
    __shared__ T      prod[kCsrChunk];
    __shared__ double red[8];
    const int xcd  = blockIdx.x & 7;
    const int nblk = (nrow + kCsrRows - 1) / kCsrRows;
    double    dacc = 0.0;
    for(int lb = blockIdx.x >> 3; lb < per_xcd; lb += wg_per_xcd)
    {
        const int blk = xcd * per_xcd + lb;
        if(blk >= nblk)
            break;
        const int r0   = blk * kCsrRows;
        const int rend = min(r0 + kCsrRows, nrow);
        const int row  = r0 + threadIdx.x;
        int       rs = 0, re = 0;
        if(row < nrow)
        {
            rs = rp[row];
            re = rp[row + 1];
        }
        const int start = rp[r0];
        const int end   = rp[rend];
        T         sum   = (T)0;
        if(MODE == 1 && row < nrow)
            sum = y[row];
        for(int cb = start & ~3; cb < end; cb += kCsrChunk)
        {
            // ---- stream val/col, gather x, park products
#pragma unroll
            for(int k = 0; k < kCsrChunk / (4 * kBlock); ++k)
            {
                const int g = (k * kBlock + threadIdx.x) * 4;
                const int j = cb + g;
                if(j < end)
                {
                    v4i32 c = nt_load(reinterpret_cast<const v4i32*>(ci + j));
                    ValPack4<T> v;
                    v.load(val + j);
                    const int cc[4] = {c.x, c.y, c.z, c.w};
                    T         p[4];
#pragma unroll
                    for(int e = 0; e < 4; ++e)
                    {
                        const int jj = j + e;
                        if(jj >= start && jj < end)
                        {
                            if(MODE == 0)
                                p[e] = v.get(e) * x[cc[e]];
                            else
                                p[e] = scalar * v.get(e) * x[cc[e]];
                        }
                        else
                            p[e] = (T)0;
                    }
#pragma unroll
                    for(int e = 0; e < 4; ++e)
                        prod[g + e] = p[e];
                }
            }
            __syncthreads();
            // ---- every thread adds up its own row, left to right
            const int lo = max(rs, cb);
            const int hi = min(re, cb + kCsrChunk);
            for(int j = lo; j < hi; ++j)
                sum += prod[j - cb];
            __syncthreads();
        }
        if(row < nrow)
        {
            nt_store(sum, y + row);
            if(DOT)
                dacc += (double)sum * (double)x[row];
        }
    }
    if(DOT)
    {
        const double vals[1]  = {dacc};
        const int    slots[1] = {slot};
        const int    ops[1]   = {RED_SUM};
        grid_reduce_finish<1>(ctx, vals, slots, ops, red);
    }
}

// ELL: one thread per row, column-major => every load is a perfectly coalesced wave access.
// STOP=true : ELL semantics (stop at the first negative column, host_matrix_ell.cpp:309-318)
// STOP=false: HYB-ELL semantics (skip invalid columns, host_matrix_hyb.cpp:344-352)
template <typename T, int MODE, bool STOP, bool DOT>
__global__ __launch_bounds__(kBlock) void k_ell(int nrow, int ncol, int width,
                                                const int* __restrict__ ecol,
                                                const T* __restrict__ eval,
                                                const T* __restrict__ x, T* __restrict__ y, T scalar,
                                                ReduceCtx ctx, int slot)
{
    __shared__ double red[8];
    double            dacc = 0.0;
    const int64_t     gsz  = (int64_t)gridDim.x * blockDim.x;
    for(int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrow; row += gsz)
    {
        T sum = (T)0;
        if(MODE == 1)
            sum = y[row];
        int el = 0;
        // 4-wide unrolled: issue the independent col/val loads before the dependent gathers
        for(; el + 4 <= width; el += 4)
        {
            int c[4];
            T   v[4];
#pragma unroll
            for(int e = 0; e < 4; ++e)
            {
                c[e] = nt_load(ecol + (int64_t)(el + e) * nrow + row);
                v[e] = nt_load(eval + (int64_t)(el + e) * nrow + row);
            }
            bool stop = false;
#pragma unroll
            for(int e = 0; e < 4; ++e)
            {
                if(STOP)
                {
                    if(stop || c[e] < 0)
                    {
                        stop = true;
                        continue;
                    }
                }
                else if(c[e] < 0 || c[e] >= ncol)
                    continue;
                if(MODE == 0)
                    sum += v[e] * x[c[e]];
                else
                    sum += scalar * v[e] * x[c[e]];
            }
            if(STOP && stop)
            {
                el = width;
                break;
            }
        }
        for(; el < width; ++el)
        {
            int c = nt_load(ecol + (int64_t)el * nrow + row);
            if(STOP)
            {
                if(c < 0)
                    break;
            }
            else if(c < 0 || c >= ncol)
                continue;
            T v = nt_load(eval + (int64_t)el * nrow + row);
            if(MODE == 0)
                sum += v * x[c];
            else
                sum += scalar * v * x[c];
        }
        nt_store(sum, y + row);
        if(DOT)
            dacc += (double)sum * (double)x[row];
    }
    if(DOT)
    {
        const double vals[1]  = {dacc};
        const int    slots[1] = {slot};
        const int    ops[1]   = {RED_SUM};
        grid_reduce_finish<1>(ctx, vals, slots, ops, red);
    }
}

// COO (ghost part of a GlobalMatrix, HYB tail): COO data in this library always comes from a CSR
// conversion, i.e. it is sorted by row; the non-empty rows and their ranges are compacted once, so
// one thread adds up one touched row in storage order -- the order of the reference's serial loop
// (host_matrix_coo.cpp:368-376, :395-400) restricted to that row.  No atomics, deterministic.
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void k_coo_grouped(int ngroups, const int* __restrict__ grow,
                                                        const int* __restrict__ gptr,
                                                        const int* __restrict__ ccol,
                                                        const T* __restrict__ cval,
                                                        const T* __restrict__ x, T* __restrict__ y,
                                                        T scalar)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gsz)
    {
        const int row = grow[g];
        T         sum = y[row];
        for(int i = gptr[g]; i < gptr[g + 1]; ++i)
        {
            if(MODE == 0)
                sum += cval[i] * x[ccol[i]];
            else
                sum += scalar * cval[i] * x[ccol[i]];
        }
        y[row] = sum;
    }
}

// ------------------------------------------------------------------------------------------
template <typename T>
static int launch_csr(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar, bool dot, int slot)
{
    Backend&  b       = backend();
    const int nblk    = (m->nrow + kCsrRows - 1) / kCsrRows;
    const int per_xcd = (nblk + 7) / 8;
    const int wg_xcd  = std::min(per_xcd, kCsrWgPerXcd);
    const int grid    = wg_xcd * 8;
    ReduceCtx ctx     = reduce_ctx();
#define LAUNCH(MODE, DOT)                                                                           \
    hipLaunchKernelGGL((k_csr_stream<T, MODE, DOT>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow,   \
                       per_xcd, wg_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ctx, slot)
    if(mode == 0 && !dot)
        LAUNCH(0, false);
    else if(mode == 0 && dot)
        LAUNCH(0, true);
    else
        LAUNCH(1, false);
#undef LAUNCH
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

template <typename T>
static int launch_ell(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar, bool stop)
{
    Backend&  b    = backend();
    const int grid = (int)std::min<int64_t>(((int64_t)m->nrow + kBlock - 1) / kBlock, 1 << 22);
    ReduceCtx ctx  = reduce_ctx();
#define LAUNCH(MODE, STOP)                                                                        \
    hipLaunchKernelGGL((k_ell<T, MODE, STOP, false>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, \
                       m->ncol, m->ell_width, m->ell_col, (const T*)m->ell_val, x, y, scalar, ctx, 0)
    if(mode == 0 && stop)
        LAUNCH(0, true);
    else if(mode == 0)
        LAUNCH(0, false);
    else if(stop)
        LAUNCH(1, true);
    else
        LAUNCH(1, false);
#undef LAUNCH
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

template <typename T>
static int launch_coo(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar)
{
    if(m->coo_nnz <= 0)
        return RAMD_OK;
    if(!m->coo_gptr)
        RAMD_FAIL(RAMD_ERR_STATE, "COO part has no row grouping (internal)");
    Backend&  b    = backend();
    const int grid = std::max(1, (m->coo_ngroups + kBlock - 1) / kBlock);
    if(mode == 0)
        hipLaunchKernelGGL((k_coo_grouped<T, 0>), dim3(grid), dim3(kBlock), 0, b.cur, m->coo_ngroups,
                           m->coo_grow, m->coo_gptr, m->coo_col, (const T*)m->coo_val, x,
                           y, scalar);
    else
        hipLaunchKernelGGL((k_coo_grouped<T, 1>), dim3(grid), dim3(kBlock), 0, b.cur, m->coo_ngroups,
                           m->coo_grow, m->coo_gptr, m->coo_col, (const T*)m->coo_val, x,
                           y, scalar);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

// mode 0: Apply, 1: ApplyAdd
template <typename T>
static int mat_apply_inner(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar);

template <typename T>
int mat_apply_impl(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar)
{
    prof_spmv_begin();
    int s = mat_apply_inner<T>(m, x, y, mode, scalar);
    prof_spmv_end();
    return s;
}

template <typename T>
static int mat_apply_inner(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar)
{
    Backend& b = backend();
    // LocalMatrix::Apply zero-fills when nnz == 0, ApplyAdd does nothing
    // (src/base/local_matrix.cpp:2176-2181, :2201-2209)
    if(m->nnz <= 0)
    {
        if(mode == 0 && m->nrow > 0)
            RAMD_HIP(hipMemsetAsync(y, 0, sizeof(T) * (size_t)m->nrow, b.cur));
        return RAMD_OK;
    }
    switch(m->format)
    {
    case RAMD_CSR:
        return launch_csr<T>(m, x, y, mode, scalar, false, 0);
    case RAMD_ELL:
        return launch_ell<T>(m, x, y, mode, scalar, true);
    case RAMD_HYB:
        if(m->ell_width > 0)
            RAMD_TRY(launch_ell<T>(m, x, y, mode, scalar, false));
        else if(mode == 0)
            RAMD_HIP(hipMemsetAsync(y, 0, sizeof(T) * (size_t)m->nrow, b.cur));
        return launch_coo<T>(m, x, y, mode, scalar);
    case RAMD_COO:
        if(mode == 0)
            RAMD_HIP(hipMemsetAsync(y, 0, sizeof(T) * (size_t)m->nrow, b.cur));
        return launch_coo<T>(m, x, y, mode, scalar);
    default:
        RAMD_FAIL(RAMD_ERR_UNSUPPORTED, "matrix format not provided by this backend");
    }
}

template int mat_apply_impl<double>(const ramd_mat_s*, const double*, double*, int, double);
template int mat_apply_impl<float>(const ramd_mat_s*, const float*, float*, int, float);

template <typename T>
int mat_apply_dot_impl(const ramd_mat_s* m, const T* x, T* y, int slot)
{
    if(m->format == RAMD_CSR && m->nnz > 0 && m->nrow == m->ncol)
    {
        prof_spmv_begin();
        int s = launch_csr<T>(m, x, y, 0, (T)1, true, slot);
        prof_spmv_end();
        return s;
    }
    return RAMD_ERR_UNSUPPORTED; // caller falls back to apply + dot (two launches)
}
template int mat_apply_dot_impl<double>(const ramd_mat_s*, const double*, double*, int);
template int mat_apply_dot_impl<float>(const ramd_mat_s*, const float*, float*, int);

} // namespace ramd
