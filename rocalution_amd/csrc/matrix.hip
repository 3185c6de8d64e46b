// matrix.hip -- device sparse matrices: storage, host<->device, Apply dispatch, diagonal
// extraction, device-side synthetic operators.
//
// Replaces the object plumbing of src/base/hip/hip_matrix_{csr,ell,hyb,coo}.cpp
// (CopyFromHost/CopyToHost, Clear, CopyFrom) and kernel_csr_extract_inv_diag
// (src/base/hip/hip_kernels_csr.hpp:158-188).  Format conversions live in convert.hip,
// triangular solves / ILU(0) in trisolve.hip.
#include "device_utils.hpp"
#include "matrix_impl.hpp"

namespace ramd
{

size_t val_size(int dtype)
{
    return dtype == RAMD_F64 ? 8 : 4;
}

void mat_free_csr(ramd_mat_s* m)
{
    dev_free(&m->rp);
    dev_free(&m->ci);
    if(m->val)
        (void)cached_free(m->val);
    m->val = nullptr;
}
void mat_free_ell(ramd_mat_s* m)
{
    dev_free(&m->ell_col);
    if(m->ell_val)
        (void)cached_free(m->ell_val);
    m->ell_val   = nullptr;
    m->ell_width = 0;
}
void mat_free_coo(ramd_mat_s* m)
{
    dev_free(&m->coo_row);
    dev_free(&m->coo_col);
    if(m->coo_val)
        (void)cached_free(m->coo_val);
    m->coo_val = nullptr;
    dev_free(&m->coo_grow);
    dev_free(&m->coo_gptr);
    m->coo_nnz     = 0;
    m->coo_ngroups = 0;
}
void mat_free_analysis(ramd_mat_s* m)
{
    dev_free(&m->diag_pos);
    dev_free(&m->dot_part1);
    m->dot_nblk  = 0;
    m->band_dist = -1;
    m->shift_rows = -1;
    dev_free(&m->pat_id);
    dev_free(&m->pat_dict);
    dev_free(&m->blk_rp);
    dev_free(&m->wav_rp);
    m->blk_span = 0;
    dev_free(&m->xl_dict);
    m->xl_state = 0;
    dev_free(&m->grp_lead);
    dev_free(&m->grp_need);
    m->grp_state = 0;
    m->pat_state = m->pat_n = m->pat_w = 0;
    tri_release(m);
    m->lu_analysed = m->l_analysed = m->u_analysed = false;
    m->l_diag_unit                                 = true;
    m->u_diag_unit                                 = false;
}

int mat_alloc_csr(ramd_mat_s* m, int nrow, int ncol, int64_t nnz)
{
    mat_free_csr(m);
    mat_free_ell(m);
    mat_free_coo(m);
    mat_free_analysis(m);
    m->format = RAMD_CSR;
    m->nrow   = nrow;
    m->ncol   = ncol;
    m->nnz    = nnz;
    RAMD_TRY(dev_alloc(&m->rp, (int64_t)nrow + 1));
    RAMD_TRY(dev_alloc(&m->ci, nnz));
    void* v = nullptr;
    RAMD_HIP(cached_malloc(&v, (size_t)(nnz > 0 ? nnz : 0) * val_size(m->dtype) + kPad));
    m->val = v;
    return RAMD_OK;
}

// ---- inverse / plain diagonal: host_matrix_csr.cpp:772-845 (first matching column wins; zero
// diagonal -> 1 for the inverse; rows without a stored diagonal are left untouched)
template <typename T, bool INV>
__global__ __launch_bounds__(kBlock) void k_csr_diag(int nrow, const int* __restrict__ rp,
                                                     const int* __restrict__ ci,
                                                     const T* __restrict__ val, T* __restrict__ d,
                                                     int* __restrict__ zero_flag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrow; row += gsz)
    {
        for(int j = rp[row]; j < rp[row + 1]; ++j)
        {
            if(ci[j] == row)
            {
                T a = val[j];
                if(INV)
                {
                    if(a != (T)0)
                        d[row] = (T)1 / a;
                    else
                    {
                        d[row] = (T)1;
                        if(zero_flag)
                            *zero_flag = 1;
                    }
                }
                else
                    d[row] = a;
                break;
            }
        }
    }
}

// ---- 3-D 7-point Poisson generator (SURVEY.md §8d layout): two passes, no host traffic
__device__ __forceinline__ int poisson_row_nnz(int64_t r, int N, int64_t lo, int64_t hi, bool ghost)
{
    const int     x = (int)(r % N), y = (int)((r / N) % N), z = (int)(r / ((int64_t)N * N));
    const int64_t N2   = (int64_t)N * N;
    int           c    = 0;
    const int64_t nb[7] = {r - N2, r - N, r - 1, r, r + 1, r + N, r + N2};
    const bool    ok[7] = {z > 0, y > 0, x > 0, true, x < N - 1, y < N - 1, z < N - 1};
    for(int k = 0; k < 7; ++k)
        if(ok[k])
        {
            bool local = nb[k] >= lo && nb[k] < hi;
            if(local != ghost)
                ++c;
        }
    return c;
}

__global__ __launch_bounds__(kBlock) void k_poisson_count(int N, int64_t lo, int64_t hi, int ghost,
                                                          int* __restrict__ cnt)
{
    const int64_t nloc = hi - lo;
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nloc; i += gsz)
        cnt[i] = poisson_row_nnz(lo + i, N, lo, hi, ghost != 0);
}

// ghost columns are renumbered into the halo receive buffer: [lower neighbour plane | upper plane]
template <typename T>
__global__ __launch_bounds__(kBlock) void k_poisson_fill(int N, int64_t lo, int64_t hi, int ghost,
                                                         int64_t n_lower_halo,
                                                         const int* __restrict__ rp,
                                                         int* __restrict__ ci, T* __restrict__ val)
{
    const int64_t nloc = hi - lo;
    const int64_t N2   = (int64_t)N * N;
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nloc; i += gsz)
    {
        const int64_t r = lo + i;
        const int     x = (int)(r % N), y = (int)((r / N) % N), z = (int)(r / N2);
        const int64_t nb[7] = {r - N2, r - N, r - 1, r, r + 1, r + N, r + N2};
        const bool    ok[7] = {z > 0, y > 0, x > 0, true, x < N - 1, y < N - 1, z < N - 1};
        int           p     = rp[i];
        for(int k = 0; k < 7; ++k)
            if(ok[k])
            {
                const bool local = nb[k] >= lo && nb[k] < hi;
                if(local == (ghost != 0))
                    continue;
                int c;
                if(!ghost)
                    c = (int)(nb[k] - lo);
                else if(nb[k] < lo) // received from the lower z-neighbour: its last plane
                    c = (int)(nb[k] - (lo - N2));
                else // received from the upper z-neighbour: its first plane
                    c = (int)(n_lower_halo + (nb[k] - hi));
                ci[p]  = c;
                val[p] = (k == 3) ? (T)6 : (T)-1;
                ++p;
            }
    }
}

// The reference's own 3-D operator (clients/include/utility.hpp:110-177 gen_3d_laplacian): the 27-point stencil on a lattice, row
// r = (z ny + y) nx + x, the entries of a row in the order of its three nested offset loops (sz, sy, sx = -1 .. 1: ascending
// columns), 26 on the diagonal and -1 elsewhere, a neighbour the lattice does not have left out.  (nx = ny = nz there.)
__device__ __forceinline__ int lap27_span(int i, int n)
{
    return (i > 0 ? 1 : 0) + 1 + (i < n - 1 ? 1 : 0);
}
__global__ __launch_bounds__(kBlock) void k_lap27_count(int nx, int ny, int nz, int* __restrict__ cnt)
{
    const int64_t n   = (int64_t)nx * ny * nz;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gsz)
    {
        const int x = (int)(r % nx), y = (int)((r / nx) % ny), z = (int)(r / ((int64_t)nx * ny));
        cnt[r]      = lap27_span(x, nx) * lap27_span(y, ny) * lap27_span(z, nz);
    }
}
// rows of the planes [z0, z1): the entries inside the slab (ghost = 0) or in the planes z0 - 1 / z1 (ghost = 1)
__global__ __launch_bounds__(kBlock) void k_lap27_slab_count(int nx, int ny, int nz, int z0, int z1, int ghost, int* __restrict__ cnt)
{
    const int64_t nxny = (int64_t)nx * ny, nloc = nxny * (z1 - z0);
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nloc; i += gsz)
    {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = z0 + (int)(i / nxny);
        int       planes = 0;
        for(int sz = -1; sz <= 1; ++sz)
            if(z + sz >= 0 && z + sz < nz && ((z + sz >= z0 && z + sz < z1) != (ghost != 0)))
                ++planes;
        cnt[i] = lap27_span(x, nx) * lap27_span(y, ny) * planes;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_lap27_slab_fill(int nx, int ny, int nz, int z0, int z1, int ghost, int n_lower,
                                                            const int* __restrict__ rp, int* __restrict__ ci, T* __restrict__ val)
{
    const int64_t nxny = (int64_t)nx * ny, nloc = nxny * (z1 - z0);
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nloc; i += gsz)
    {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = z0 + (int)(i / nxny);
        int       p = rp[i];
        for(int sz = -1; sz <= 1; ++sz)
        {
            if(z + sz < 0 || z + sz >= nz || ((z + sz >= z0 && z + sz < z1) == (ghost != 0)))
                continue;
            for(int sy = -1; sy <= 1; ++sy)
            {
                if(y + sy < 0 || y + sy >= ny)
                    continue;
                for(int sx = -1; sx <= 1; ++sx)
                {
                    if(x + sx < 0 || x + sx >= nx)
                        continue;
                    const int64_t inplane = (int64_t)(y + sy) * nx + (x + sx);
                    ci[p]  = !ghost ? (int)((int64_t)(z + sz - z0) * nxny + inplane) : (int)((z + sz < z0 ? 0 : n_lower) + inplane);
                    val[p] = (sz == 0 && sy == 0 && sx == 0) ? (T)26 : (T)-1;
                    ++p;
                }
            }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_lap27_fill(int nx, int ny, int nz, const int* __restrict__ rp, int* __restrict__ ci,
                                                       T* __restrict__ val)
{
    const int64_t n   = (int64_t)nx * ny * nz;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gsz)
    {
        const int x = (int)(r % nx), y = (int)((r / nx) % ny), z = (int)(r / ((int64_t)nx * ny));
        int       p = rp[r];
        for(int sz = -1; sz <= 1; ++sz)
        {
            if(z + sz < 0 || z + sz >= nz)
                continue;
            for(int sy = -1; sy <= 1; ++sy)
            {
                if(y + sy < 0 || y + sy >= ny)
                    continue;
                for(int sx = -1; sx <= 1; ++sx)
                {
                    if(x + sx < 0 || x + sx >= nx)
                        continue;
                    const int64_t col = r + ((int64_t)sz * ny + sy) * nx + sx;
                    ci[p]             = (int)col;
                    val[p]            = (col == r) ? (T)26 : (T)-1;
                    ++p;
                }
            }
        }
    }
}

} // namespace ramd

using namespace ramd;

#define CHECK_MAT(m)                                      \
    do                                                    \
    {                                                     \
        if(!(m))                                          \
            RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle"); \
    } while(0)

static int check_apply_args(ramd_mat_t m, ramd_vec_t x, ramd_vec_t y)
{
    CHECK_MAT(m);
    if(!x || !y)
        RAMD_FAIL(RAMD_ERR_ARG, "null vector handle");
    if(x->dtype != m->dtype || y->dtype != m->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "Apply: vector/matrix value types differ");
    // the reference asserts in.GetSize() == ncol && out->GetSize() == nrow
    if(x->n != m->ncol || y->n != m->nrow)
        RAMD_FAIL(RAMD_ERR_ARG, "Apply: vector sizes do not match the matrix");
    if(x == y)
        RAMD_FAIL(RAMD_ERR_ARG, "Apply: in and out must differ");
    return RAMD_OK;
}

extern "C" {

int ramd_mat_create(int dtype, ramd_mat_t* out)
{
    RAMD_TRY(ensure_init());
    if(!out || (dtype != RAMD_F64 && dtype != RAMD_F32))
        RAMD_FAIL(RAMD_ERR_ARG, "bad dtype / null output");
    ramd_mat_s* m = new ramd_mat_s;
    m->dtype      = dtype;
    *out          = m;
    return RAMD_OK;
}

int ramd_mat_clear(ramd_mat_t m)
{
    CHECK_MAT(m);
    mat_free_csr(m);
    mat_free_ell(m);
    mat_free_coo(m);
    mat_free_analysis(m);
    m->format = RAMD_CSR;
    m->nrow = m->ncol = 0;
    m->nnz            = 0;
    return RAMD_OK;
}

int ramd_mat_destroy(ramd_mat_t m)
{
    if(!m)
        return RAMD_OK;
    ramd_mat_clear(m);
    delete m;
    return RAMD_OK;
}

int ramd_mat_info(ramd_mat_t m, int* nrow, int* ncol, int64_t* nnz, int* format, int* dtype)
{
    CHECK_MAT(m);
    if(nrow)
        *nrow = m->nrow;
    if(ncol)
        *ncol = m->ncol;
    if(nnz)
        *nnz = m->nnz;
    if(format)
        *format = m->format;
    if(dtype)
        *dtype = m->dtype;
    return RAMD_OK;
}

int ramd_mat_set_csr_from_host(ramd_mat_t m, int nrow, int ncol, int64_t nnz, const int32_t* rp,
                               const int32_t* ci, const void* val)
{
    CHECK_MAT(m);
    if(nrow < 0 || ncol < 0 || nnz < 0 || (nrow > 0 && !rp) || (nnz > 0 && (!ci || !val)))
        RAMD_FAIL(RAMD_ERR_ARG, "bad CSR arguments");
    RAMD_TRY(mat_alloc_csr(m, nrow, ncol, nnz));
    Backend& b = backend();
    RAMD_HIP(hipMemcpyAsync(m->rp, rp, sizeof(int) * ((size_t)nrow + 1), hipMemcpyHostToDevice, b.cur));
    if(nnz > 0)
    {
        RAMD_HIP(hipMemcpyAsync(m->ci, ci, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, b.cur));
        RAMD_HIP(hipMemcpyAsync(m->val, val, val_size(m->dtype) * (size_t)nnz, hipMemcpyHostToDevice,
                                b.cur));
    }
    RAMD_HIP(hipStreamSynchronize(b.cur));
    return RAMD_OK;
}

int ramd_mat_copy_csr_to_host(ramd_mat_t m, int32_t* rp, int32_t* ci, void* val)
{
    CHECK_MAT(m);
    if(m->format != RAMD_CSR)
        RAMD_FAIL(RAMD_ERR_STATE, "CopyToCSR: matrix is not in CSR format");
    Backend& b = backend();
    if(rp)
        RAMD_HIP(hipMemcpyAsync(rp, m->rp, sizeof(int) * ((size_t)m->nrow + 1), hipMemcpyDeviceToHost,
                                b.cur));
    if(m->nnz > 0)
    {
        if(ci)
            RAMD_HIP(hipMemcpyAsync(ci, m->ci, sizeof(int) * (size_t)m->nnz, hipMemcpyDeviceToHost,
                                    b.cur));
        if(val)
            RAMD_HIP(hipMemcpyAsync(val, m->val, val_size(m->dtype) * (size_t)m->nnz,
                                    hipMemcpyDeviceToHost, b.cur));
    }
    RAMD_HIP(hipStreamSynchronize(b.cur));
    return RAMD_OK;
}

int ramd_mat_clone(ramd_mat_t src, ramd_mat_t* out)
{
    CHECK_MAT(src);
    ramd_mat_t m = nullptr;
    RAMD_TRY(ramd_mat_create(src->dtype, &m));
    Backend&     b  = backend();
    const size_t vs = val_size(src->dtype);
    int          s  = RAMD_OK;
    // deep copy of whatever the format holds (analysis data is not cloned: CloneFrom copies the matrix)
    auto dup_i = [&](int** dst, const int* from, int64_t n) {
        if(s != RAMD_OK || !from)
            return;
        s = dev_alloc(dst, n);
        if(s == RAMD_OK && n > 0
           && hipMemcpyAsync(*dst, from, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    };
    auto dup_v = [&](void** dst, const void* from, int64_t n) {
        if(s != RAMD_OK || !from)
            return;
        if(cached_malloc(dst, (size_t)(n > 0 ? n : 0) * vs + kPad) != hipSuccess)
        {
            s = RAMD_ERR_HIP;
            return;
        }
        if(n > 0 && hipMemcpyAsync(*dst, from, vs * (size_t)n, hipMemcpyDeviceToDevice, b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    };
    m->format = src->format;
    m->nrow   = src->nrow;
    m->ncol   = src->ncol;
    m->nnz    = src->nnz;
    if(src->format == RAMD_CSR)
    {
        dup_i(&m->rp, src->rp, (int64_t)src->nrow + 1);
        dup_i(&m->ci, src->ci, src->nnz);
        dup_v(&m->val, src->val, src->nnz);
    }
    if(src->format == RAMD_ELL || src->format == RAMD_HYB)
    {
        const int64_t ne = (int64_t)src->ell_width * src->nrow;
        m->ell_width     = src->ell_width;
        dup_i(&m->ell_col, src->ell_col, ne);
        dup_v(&m->ell_val, src->ell_val, ne);
    }
    if(src->format == RAMD_COO || src->format == RAMD_HYB)
    {
        m->coo_nnz     = src->coo_nnz;
        m->coo_ngroups = src->coo_ngroups;
        dup_i(&m->coo_row, src->coo_row, src->coo_nnz);
        dup_i(&m->coo_col, src->coo_col, src->coo_nnz);
        dup_v(&m->coo_val, src->coo_val, src->coo_nnz);
        dup_i(&m->coo_grow, src->coo_grow, src->coo_ngroups);
        dup_i(&m->coo_gptr, src->coo_gptr, (int64_t)src->coo_ngroups + 1);
    }
    m->band_dist = src->band_dist;
    m->shift_rows = src->shift_rows;
    if(s != RAMD_OK)
    {
        ramd_mat_destroy(m);
        RAMD_FAIL(s, "clone: allocation / copy failed");
    }
    *out = m;
    return RAMD_OK;
}

int ramd_mat_pattern_info(ramd_mat_t m, int* state, int* entries, int* width)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(state)
        *state = (m->pat_state != 1 && m->grp_state == 1) ? 2 : m->pat_state;
    if(entries)
        *entries = m->pat_n;
    if(width)
        *width = m->pat_w;
    return RAMD_OK;
}
int ramd_mat_pattern_use(ramd_mat_t m, int on)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    m->pat_off = (on == 0);
    return RAMD_OK;
}
int ramd_mat_apply(ramd_mat_t m, ramd_vec_t x, ramd_vec_t y)
{
    RAMD_TRY(check_apply_args(m, x, y));
    if(m->dtype == RAMD_F64)
        return mat_apply_impl<double>(m, (const double*)x->d, (double*)y->d, 0, 1.0);
    return mat_apply_impl<float>(m, (const float*)x->d, (float*)y->d, 0, 1.0f);
}

int ramd_mat_apply_add(ramd_mat_t m, ramd_vec_t x, double scalar, ramd_vec_t y)
{
    RAMD_TRY(check_apply_args(m, x, y));
    if(m->dtype == RAMD_F64)
        return mat_apply_impl<double>(m, (const double*)x->d, (double*)y->d, 1, scalar);
    return mat_apply_impl<float>(m, (const float*)x->d, (float*)y->d, 1, (float)scalar);
}

static int extract_diag_common(ramd_mat_t m, ramd_vec_t d, bool inv)
{
    CHECK_MAT(m);
    if(!d || d->dtype != m->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "diagonal vector has the wrong value type");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED; // the reference's front end converts to CSR first
    const int64_t nd = std::min(m->nrow, m->ncol);
    if(inv)
    {
        // LocalMatrix::ExtractInverseDiagonal (local_matrix.cpp:2294-2299) allocates (zero-filled)
        if(m->nnz > 0)
            RAMD_TRY(ramd_vec_allocate(d, nd));
    }
    else if(d->n < nd)
        RAMD_FAIL(RAMD_ERR_ARG, "ExtractDiagonal: vector too small");
    if(m->nnz <= 0 || nd == 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = ew_grid(nd);
    if(m->dtype == RAMD_F64)
    {
        if(inv)
            hipLaunchKernelGGL((k_csr_diag<double, true>), dim3(grid), dim3(kBlock), 0, b.cur, (int)nd,
                               m->rp, m->ci, (const double*)m->val, (double*)d->d, (int*)nullptr);
        else
            hipLaunchKernelGGL((k_csr_diag<double, false>), dim3(grid), dim3(kBlock), 0, b.cur, (int)nd,
                               m->rp, m->ci, (const double*)m->val, (double*)d->d, (int*)nullptr);
    }
    else
    {
        if(inv)
            hipLaunchKernelGGL((k_csr_diag<float, true>), dim3(grid), dim3(kBlock), 0, b.cur, (int)nd,
                               m->rp, m->ci, (const float*)m->val, (float*)d->d, (int*)nullptr);
        else
            hipLaunchKernelGGL((k_csr_diag<float, false>), dim3(grid), dim3(kBlock), 0, b.cur, (int)nd,
                               m->rp, m->ci, (const float*)m->val, (float*)d->d, (int*)nullptr);
    }
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_mat_extract_diag(ramd_mat_t m, ramd_vec_t d)
{
    return extract_diag_common(m, d, false);
}
int ramd_mat_extract_inv_diag(ramd_mat_t m, ramd_vec_t d)
{
    return extract_diag_common(m, d, true);
}

static int gen_poisson_common(ramd_mat_t m, int N, int64_t lo, int64_t hi, int ghost)
{
    Backend&      b    = backend();
    const int64_t nloc = hi - lo;
    const int64_t N2   = (int64_t)N * N;
    if(nloc <= 0 || nloc >= (1ll << 31))
        RAMD_FAIL(RAMD_ERR_ARG, "poisson7: bad row range");
    int* cnt = nullptr;
    RAMD_TRY(dev_alloc(&cnt, nloc + 1));
    const int grid = ew_grid(nloc);
    hipLaunchKernelGGL(k_poisson_count, dim3(grid), dim3(kBlock), 0, b.cur, N, lo, hi, ghost, cnt);
    int* rp = nullptr;
    int  s  = dev_alloc(&rp, nloc + 1);
    if(s == RAMD_OK)
        s = device_exclusive_scan(cnt, rp, nloc + 1);
    int64_t nnz = 0;
    if(s == RAMD_OK)
    {
        int        last = 0;
        hipError_t e    = hipMemcpyAsync(&last, rp + nloc, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
        nnz = last;
    }
    dev_free(&cnt);
    if(s != RAMD_OK)
    {
        dev_free(&rp);
        return s;
    }
    // halo buffer layout: [lower plane (if any) | upper plane (if any)]
    const int64_t n_lower = (lo > 0) ? N2 : 0;
    const int64_t n_upper = (hi < (int64_t)N * N2) ? N2 : 0;
    const int     ncol    = ghost ? (int)(n_lower + n_upper) : (int)nloc;
    s                     = mat_alloc_csr(m, (int)nloc, ncol, nnz);
    if(s != RAMD_OK)
    {
        dev_free(&rp);
        return s;
    }
    RAMD_HIP(hipMemcpyAsync(m->rp, rp, sizeof(int) * ((size_t)nloc + 1), hipMemcpyDeviceToDevice, b.cur));
    if(nnz > 0)
    {
        if(m->dtype == RAMD_F64)
            hipLaunchKernelGGL((k_poisson_fill<double>), dim3(grid), dim3(kBlock), 0, b.cur, N, lo, hi,
                               ghost, n_lower, m->rp, m->ci, (double*)m->val);
        else
            hipLaunchKernelGGL((k_poisson_fill<float>), dim3(grid), dim3(kBlock), 0, b.cur, N, lo, hi,
                               ghost, n_lower, m->rp, m->ci, (float*)m->val);
    }
    RAMD_HIP(hipGetLastError());
    RAMD_HIP(hipStreamSynchronize(b.cur));
    dev_free(&rp);
    return RAMD_OK;
}

int ramd_mat_gen_poisson7(ramd_mat_t m, int N)
{
    CHECK_MAT(m);
    if(N < 1 || (int64_t)N * N * N >= (1ll << 31) / 7)
        RAMD_FAIL(RAMD_ERR_ARG, "poisson7: N out of the int32 index range");
    return gen_poisson_common(m, N, 0, (int64_t)N * N * N, 0);
}

int ramd_mat_gen_laplace27(ramd_mat_t m, int nx, int ny, int nz)
{
    CHECK_MAT(m);
    const int64_t n = (int64_t)nx * ny * nz;
    if(nx < 1 || ny < 1 || nz < 1 || n >= (1ll << 31) / 27)
        RAMD_FAIL(RAMD_ERR_ARG, "laplace27: extents out of the int32 index range");
    Backend& b   = backend();
    int *    cnt = nullptr, *rp = nullptr;
    RAMD_TRY(dev_alloc(&cnt, n + 1));
    int s = dev_alloc(&rp, n + 1);
    if(s == RAMD_OK && hipMemsetAsync(cnt + n, 0, sizeof(int), b.cur) != hipSuccess)
        s = RAMD_ERR_HIP;
    const int grid = ew_grid(n);
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL(k_lap27_count, dim3(grid), dim3(kBlock), 0, b.cur, nx, ny, nz, cnt);
        s = device_exclusive_scan(cnt, rp, n + 1);
    }
    int last = 0;
    if(s == RAMD_OK
       && (hipMemcpyAsync(&last, rp + n, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
           || hipStreamSynchronize(b.cur) != hipSuccess))
        s = RAMD_ERR_HIP;
    dev_free(&cnt);
    if(s == RAMD_OK)
        s = mat_alloc_csr(m, (int)n, (int)n, (int64_t)last);
    if(s == RAMD_OK)
    {
        hipError_t e = hipMemcpyAsync(m->rp, rp, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToDevice, b.cur);
        if(m->dtype == RAMD_F64)
            hipLaunchKernelGGL((k_lap27_fill<double>), dim3(grid), dim3(kBlock), 0, b.cur, nx, ny, nz, m->rp, m->ci, (double*)m->val);
        else
            hipLaunchKernelGGL((k_lap27_fill<float>), dim3(grid), dim3(kBlock), 0, b.cur, nx, ny, nz, m->rp, m->ci, (float*)m->val);
        if(e != hipSuccess || hipGetLastError() != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&rp);
    return s;
}

// the z-slab [z0, z1) of the same operator, split like ramd_mat_gen_poisson7_slab: entries whose column lies in the slab (interior,
// local columns) / in the plane below or above it (ghost, columns renumbered into the halo receive buffer [lower plane | upper
// plane], in-plane index y nx + x) -- what a rank of the reference's MPI generator holds (clients/include/common.hpp:926-1249
// builds the same 27-point operator per rank)
static int gen_lap27_slab_part(ramd_mat_t m, int nx, int ny, int nz, int z0, int z1, int ghost)
{
    Backend&      b    = backend();
    const int64_t nxny = (int64_t)nx * ny, nloc = nxny * (z1 - z0);
    int *         cnt = nullptr, *rp = nullptr;
    RAMD_TRY(dev_alloc(&cnt, nloc + 1));
    int s = dev_alloc(&rp, nloc + 1);
    if(s == RAMD_OK && hipMemsetAsync(cnt + nloc, 0, sizeof(int), b.cur) != hipSuccess)
        s = RAMD_ERR_HIP;
    const int grid = ew_grid(nloc);
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL(k_lap27_slab_count, dim3(grid), dim3(kBlock), 0, b.cur, nx, ny, nz, z0, z1, ghost, cnt);
        s = device_exclusive_scan(cnt, rp, nloc + 1);
    }
    int last = 0;
    if(s == RAMD_OK
       && (hipMemcpyAsync(&last, rp + nloc, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
           || hipStreamSynchronize(b.cur) != hipSuccess))
        s = RAMD_ERR_HIP;
    dev_free(&cnt);
    const int64_t n_lower = z0 > 0 ? nxny : 0, n_upper = z1 < nz ? nxny : 0;
    if(s == RAMD_OK)
        s = mat_alloc_csr(m, (int)nloc, ghost ? (int)(n_lower + n_upper) : (int)nloc, (int64_t)last);
    if(s == RAMD_OK)
    {
        hipError_t e = hipMemcpyAsync(m->rp, rp, sizeof(int) * ((size_t)nloc + 1), hipMemcpyDeviceToDevice, b.cur);
        if(last > 0)
        {
            if(m->dtype == RAMD_F64)
                hipLaunchKernelGGL((k_lap27_slab_fill<double>), dim3(grid), dim3(kBlock), 0, b.cur, nx, ny, nz, z0, z1, ghost, (int)n_lower,
                                   m->rp, m->ci, (double*)m->val);
            else
                hipLaunchKernelGGL((k_lap27_slab_fill<float>), dim3(grid), dim3(kBlock), 0, b.cur, nx, ny, nz, z0, z1, ghost, (int)n_lower,
                                   m->rp, m->ci, (float*)m->val);
        }
        if(e != hipSuccess || hipGetLastError() != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&rp);
    return s;
}

int ramd_mat_gen_laplace27_slab(ramd_mat_t interior, ramd_mat_t ghost, int nx, int ny, int nz, int z_begin, int z_end)
{
    CHECK_MAT(interior);
    CHECK_MAT(ghost);
    if(nx < 1 || ny < 1 || nz < 1 || z_begin < 0 || z_end > nz || z_begin >= z_end
       || (int64_t)nx * ny * (z_end - z_begin) >= (1ll << 31) / 27)
        RAMD_FAIL(RAMD_ERR_ARG, "laplace27_slab: planes [z_begin, z_end) of an nx x ny x nz lattice, inside the int32 index range");
    RAMD_TRY(gen_lap27_slab_part(interior, nx, ny, nz, z_begin, z_end, 0));
    return gen_lap27_slab_part(ghost, nx, ny, nz, z_begin, z_end, 1);
}

int ramd_mat_gen_poisson7_slab(ramd_mat_t interior, ramd_mat_t ghost, int N, int64_t row_begin,
                               int64_t row_end)
{
    CHECK_MAT(interior);
    CHECK_MAT(ghost);
    const int64_t N2 = (int64_t)N * N;
    if(N < 1 || row_begin < 0 || row_end > N2 * N || row_begin >= row_end || row_begin % N2 != 0
       || row_end % N2 != 0)
        RAMD_FAIL(RAMD_ERR_ARG, "poisson7_slab: the row range must be a whole number of z-planes");
    RAMD_TRY(gen_poisson_common(interior, N, row_begin, row_end, 0));
    return gen_poisson_common(ghost, N, row_begin, row_end, 1);
}

} // extern "C"
