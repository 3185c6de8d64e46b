// matrix_ops.hip -- small CSR utilities of the LocalMatrix API the reference's drivers and tests call next to
// the solver path: Gershgorin bounds, triangular extraction, value scaling / shifting, value update.
// Reference (host backend, which both backends follow): src/base/host/host_matrix_csr.cpp
//   Gershgorin :3465-3506 | ExtractU/UDiagonal/L/LDiagonal :919-1160 | Scale* :3509-3568 | AddScalar* :3570-3630
//   UpdateValuesCSR: src/base/local_matrix.cpp (values copied into the existing pattern)
#include "device_utils.hpp"
#include "matrix_impl.hpp"

namespace ramd
{

// per row: sum_{j != i} |a_ij| (left to right) and the LAST stored diagonal value, as the host loop;
// row results  hi = sum + diag, lo = diag - sum
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gershgorin_rows(int nrow, const int* __restrict__ rp,
                                                            const int* __restrict__ ci,
                                                            const T* __restrict__ val, T* __restrict__ lo,
                                                            T* __restrict__ hi)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += gsz)
    {
        T sum = (T)0, diag = (T)0;
        for(int j = rp[r]; j < rp[r + 1]; ++j)
        {
            if(ci[j] != (int)r)
                sum += (val[j] < (T)0) ? -val[j] : val[j];
            else
                diag = val[j];
        }
        hi[r] = sum + diag;
        lo[r] = diag - sum;
    }
}

// exact min / max of a vector, folded with the caller's start value 0 (lambda_min/max start at 0 in the host loop)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_minmax(int64_t n, const T* __restrict__ lo, const T* __restrict__ hi,
                                                   double* __restrict__ out /* [2*gridDim] */)
{
    __shared__ double smin[kBlock / 64], smax[kBlock / 64];
    double            mn = 0.0, mx = 0.0;
    const int64_t     gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
    {
        mn = fmin(mn, (double)lo[i]);
        mx = fmax(mx, (double)hi[i]);
    }
#pragma unroll
    for(int o = 32; o > 0; o >>= 1)
    {
        mn = fmin(mn, __shfl_xor(mn, o, 64));
        mx = fmax(mx, __shfl_xor(mx, o, 64));
    }
    if((threadIdx.x & 63) == 0)
    {
        smin[threadIdx.x >> 6] = mn;
        smax[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if(threadIdx.x == 0)
    {
        for(int w = 1; w < kBlock / 64; ++w)
        {
            mn = fmin(mn, smin[w]);
            mx = fmax(mx, smax[w]);
        }
        out[2 * blockIdx.x]     = mn;
        out[2 * blockIdx.x + 1] = mx;
    }
}

// KIND 0: col < row   1: col <= row   2: col > row   3: col >= row
__device__ __forceinline__ bool tri_keep(int kind, int col, int row)
{
    return kind == 0 ? col < row : kind == 1 ? col <= row : kind == 2 ? col > row : col >= row;
}
__global__ __launch_bounds__(kBlock) void k_tri_count(int nrow, int kind, const int* __restrict__ rp,
                                                      const int* __restrict__ ci, int* __restrict__ cnt)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= nrow; r += gsz)
    {
        int c = 0;
        if(r < nrow)
            for(int j = rp[r]; j < rp[r + 1]; ++j)
                if(tri_keep(kind, ci[j], (int)r))
                    ++c;
        cnt[r] = c;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_tri_fill(int nrow, int kind, const int* __restrict__ rp,
                                                     const int* __restrict__ ci, const T* __restrict__ val,
                                                     const int* __restrict__ orp, int* __restrict__ oci,
                                                     T* __restrict__ oval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += gsz)
    {
        int k = orp[r];
        for(int j = rp[r]; j < rp[r + 1]; ++j)
            if(tri_keep(kind, ci[j], (int)r))
            {
                oci[k]  = ci[j];
                oval[k] = val[j];
                ++k;
            }
    }
}

// OP 0: v *= alpha, 1: v += alpha;   WHICH 0: every entry, 1: the FIRST diagonal entry of a row, 2: off-diagonal
template <typename T, int OP>
__global__ __launch_bounds__(kBlock) void k_values_op(int nrow, int which, const int* __restrict__ rp,
                                                      const int* __restrict__ ci, T* __restrict__ val, T alpha)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += gsz)
        for(int j = rp[r]; j < rp[r + 1]; ++j)
        {
            const bool d = ci[j] == (int)r;
            if(which == 0 || (which == 1 && d) || (which == 2 && !d))
            {
                val[j] = (OP == 0) ? val[j] * alpha : val[j] + alpha;
                if(which == 1)
                    break; // the host loops stop at the first diagonal entry
            }
        }
}

} // namespace ramd

using namespace ramd;

#define NEED_CSR(m, what)                                                      \
    do                                                                         \
    {                                                                          \
        if(!(m))                                                               \
            RAMD_FAIL(RAMD_ERR_ARG, what ": null matrix handle");              \
        if((m)->format != RAMD_CSR)                                            \
            return RAMD_ERR_UNSUPPORTED;                                       \
    } while(0)

extern "C" {

#ifdef RAMD_WITH_OFFSCOPE // (Gershgorin: out of scope; built with RAMD_EXTRA_CXXFLAGS=-DRAMD_WITH_OFFSCOPE)
int ramd_mat_gershgorin(ramd_mat_t m, double* lambda_min, double* lambda_max)
{
    NEED_CSR(m, "Gershgorin");
    if(!lambda_min || !lambda_max)
        RAMD_FAIL(RAMD_ERR_ARG, "Gershgorin: null result pointer");
    *lambda_min = *lambda_max = 0.0;
    if(m->nrow == 0)
        return RAMD_OK;
    Backend&     b  = backend();
    const size_t vs = val_size(m->dtype);
    void *       lo = nullptr, *hi = nullptr;
    double*      part = nullptr;
    const int    grid = reduce_grid(m->nrow);
    int          s    = RAMD_OK;
    if(cached_malloc(&lo, vs * (size_t)m->nrow) != hipSuccess || cached_malloc(&hi, vs * (size_t)m->nrow) != hipSuccess
       || cached_malloc((void**)&part, sizeof(double) * 2 * (size_t)grid) != hipSuccess)
        s = RAMD_ERR_HIP;
    std::vector<double> h((size_t)2 * grid);
    if(s == RAMD_OK)
    {
        if(m->dtype == RAMD_F64)
        {
            hipLaunchKernelGGL((k_gershgorin_rows<double>), dim3(ew_grid(m->nrow)), dim3(kBlock), 0, b.cur, m->nrow,
                               m->rp, m->ci, (const double*)m->val, (double*)lo, (double*)hi);
            hipLaunchKernelGGL((k_minmax<double>), dim3(grid), dim3(kBlock), 0, b.cur, (int64_t)m->nrow,
                               (const double*)lo, (const double*)hi, part);
        }
        else
        {
            hipLaunchKernelGGL((k_gershgorin_rows<float>), dim3(ew_grid(m->nrow)), dim3(kBlock), 0, b.cur, m->nrow,
                               m->rp, m->ci, (const float*)m->val, (float*)lo, (float*)hi);
            hipLaunchKernelGGL((k_minmax<float>), dim3(grid), dim3(kBlock), 0, b.cur, (int64_t)m->nrow,
                               (const float*)lo, (const float*)hi, part);
        }
        if(hipMemcpyAsync(h.data(), part, sizeof(double) * h.size(), hipMemcpyDeviceToHost, b.cur) != hipSuccess
           || hipStreamSynchronize(b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    (void)cached_free(lo);
    (void)cached_free(hi);
    (void)cached_free(part);
    if(s != RAMD_OK)
        RAMD_FAIL(s, "Gershgorin: allocation / launch failed");
    double mn = 0.0, mx = 0.0;
    for(int i = 0; i < grid; ++i)
    {
        mn = std::min(mn, h[(size_t)2 * i]);
        mx = std::max(mx, h[(size_t)2 * i + 1]);
    }
    *lambda_min = mn;
    *lambda_max = mx;
    return RAMD_OK;
}
#endif // RAMD_WITH_OFFSCOPE

int ramd_mat_extract_tri(ramd_mat_t m, ramd_mat_t out, int upper, int with_diag)
{
    NEED_CSR(m, "ExtractL/U");
    if(!out || out == m || out->dtype != m->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "ExtractL/U: bad output handle");
    const int kind = (upper ? 2 : 0) + (with_diag ? 1 : 0);
    Backend&  b    = backend();
    int*      cnt  = nullptr;
    RAMD_TRY(dev_alloc(&cnt, (int64_t)m->nrow + 1));
    hipLaunchKernelGGL(k_tri_count, dim3(ew_grid((int64_t)m->nrow + 1)), dim3(kBlock), 0, b.cur, m->nrow, kind, m->rp,
                       m->ci, cnt);
    int s   = device_exclusive_scan(cnt, cnt, (int64_t)m->nrow + 1);
    int nnz = 0;
    if(s == RAMD_OK
       && (hipMemcpyAsync(&nnz, cnt + m->nrow, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
           || hipStreamSynchronize(b.cur) != hipSuccess))
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK)
        s = mat_alloc_csr(out, m->nrow, m->ncol, nnz);
    if(s == RAMD_OK)
    {
        s = (hipMemcpyAsync(out->rp, cnt, sizeof(int) * ((size_t)m->nrow + 1), hipMemcpyDeviceToDevice, b.cur)
             == hipSuccess)
                ? RAMD_OK
                : RAMD_ERR_HIP;
        if(s == RAMD_OK && nnz > 0)
        {
            if(m->dtype == RAMD_F64)
                hipLaunchKernelGGL((k_tri_fill<double>), dim3(ew_grid(m->nrow)), dim3(kBlock), 0, b.cur, m->nrow, kind,
                                   m->rp, m->ci, (const double*)m->val, cnt, out->ci, (double*)out->val);
            else
                hipLaunchKernelGGL((k_tri_fill<float>), dim3(ew_grid(m->nrow)), dim3(kBlock), 0, b.cur, m->nrow, kind,
                                   m->rp, m->ci, (const float*)m->val, cnt, out->ci, (float*)out->val);
        }
        if(s == RAMD_OK && hipStreamSynchronize(b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&cnt);
    if(s != RAMD_OK)
        RAMD_FAIL(s, "ExtractL/U failed");
    return RAMD_OK;
}

static int values_op(ramd_mat_t m, double alpha, int which, int op)
{
    NEED_CSR(m, "Scale/AddScalar");
    if(which < 0 || which > 2)
        RAMD_FAIL(RAMD_ERR_ARG, "Scale/AddScalar: which must be 0 (all), 1 (diagonal) or 2 (off-diagonal)");
    if(m->nnz <= 0)
        return RAMD_OK;
    if(m->lu_analysed || m->l_analysed || m->u_analysed)
        mat_free_analysis(m); // solve plans hold copies of the values
    Backend&  b    = backend();
    const int grid = ew_grid(m->nrow);
#define GO(T, OP)                                                                                              \
    hipLaunchKernelGGL((k_values_op<T, OP>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, which, m->rp, m->ci, \
                       (T*)m->val, (T)alpha)
    if(m->dtype == RAMD_F64)
    {
        if(op == 0)
            GO(double, 0);
        else
            GO(double, 1);
    }
    else
    {
        if(op == 0)
            GO(float, 0);
        else
            GO(float, 1);
    }
#undef GO
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}
int ramd_mat_scale_values(ramd_mat_t m, double alpha, int which)
{
    return values_op(m, alpha, which, 0);
}
int ramd_mat_add_scalar_values(ramd_mat_t m, double alpha, int which)
{
    return values_op(m, alpha, which, 1);
}

int ramd_mat_update_values(ramd_mat_t m, const void* host_val)
{
    NEED_CSR(m, "UpdateValuesCSR");
    if(!host_val && m->nnz > 0)
        RAMD_FAIL(RAMD_ERR_ARG, "UpdateValuesCSR: null value array");
    if(m->lu_analysed || m->l_analysed || m->u_analysed)
        mat_free_analysis(m);
    if(m->nnz > 0)
    {
        Backend& b = backend();
        RAMD_HIP(hipMemcpyAsync(m->val, host_val, val_size(m->dtype) * (size_t)m->nnz, hipMemcpyHostToDevice, b.cur));
        RAMD_HIP(hipStreamSynchronize(b.cur));
    }
    return RAMD_OK;
}

} // extern "C"
