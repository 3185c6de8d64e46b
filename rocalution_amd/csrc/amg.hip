// amg.hip -- aggregation kernels of the unsmoothed-aggregation AMG setup (UAAMG with CoarseningStrategy PMIS).
// Reference (host backend): src/base/host/host_matrix_csr.cpp
//   AMGComputeStrongConnections :5098-5160 | hash1 :5162-5168 | AMGPMISInitializeState :5171-5222
//   AMGPMISFindMaxNeighbourNode :5340-5533 | AMGPMISInitializeAggregateGlobalIndices :5642-5660
//   AMGPMISAddUnassignedNodesToAggregations :5536-5640 | AMGUnsmoothedAggregationProlongNnz/Fill :6331-6512
// and the driver sequence of LocalMatrix::AMGPMISAggregate / AMGUnsmoothedAggregation (local_matrix.cpp:6519-6640,
// :6852-6930).  Every step of the reference is a row-parallel loop reading the previous step's arrays only, so each
// one is a kernel with the same per-row code; the only host decisions are the "any row undecided?" flag per PMIS round
// and the scan totals.  Single-process (Local) form: no ghost part.
#include "device_utils.hpp"
#include "matrix_impl.hpp"

#include <algorithm>
#include <cmath>

namespace ramd
{

__device__ __forceinline__ unsigned pmis_hash(unsigned x)
{
    x = ((x >> 16) ^ x) * 0x45d9f3bu;
    x = ((x >> 16) ^ x) * 0x45d9f3bu;
    x = (x >> 16) ^ x;
    return x / 2u;
}

// ExtractDiagonal (host_matrix_csr.cpp:772-800): first matching column; rows without a diagonal keep the zero fill
template <typename T>
__global__ __launch_bounds__(kBlock) void k_extract_diag_plain(int nrow, const int* __restrict__ rp,
                                                               const int* __restrict__ ci, const T* __restrict__ val,
                                                               T* __restrict__ d)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(ci[j] == (int)i)
            {
                d[i] = val[j];
                break;
            }
}

// conn[j] = (c != i) && (v*v > eps^2 * d_i * d_c)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_amg_connections(int nrow, const int* __restrict__ rp,
                                                            const int* __restrict__ ci, const T* __restrict__ val,
                                                            const T* __restrict__ diag, T eps2, int* __restrict__ conn)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        const T eps_dia_i = eps2 * diag[i];
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int c = ci[j];
            const T   v = val[j];
            conn[j]     = (c != (int)i) && (v * v > eps_dia_i * diag[c]);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_pmis_init(int nrow, const int* __restrict__ rp, const int* __restrict__ conn,
                                                      int* __restrict__ state, int* __restrict__ hash,
                                                      unsigned first_row = 0u) // (global number of row 0)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        int s = -2;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(conn[j])
            {
                s = 0;
                break;
            }
        state[i] = s;
        hash[i]  = (int)pmis_hash((unsigned)i + first_row);
    }
}

struct MisTuple
{
    int s, v, i;
};
// lexographical_max(&neighbour, &t_max) of the reference: t_max stays only when strictly larger in (s, v)
__device__ __forceinline__ MisTuple mis_max(const MisTuple& nb, const MisTuple& tm)
{
    if(tm.s > nb.s)
        return tm;
    if(tm.s == nb.s && tm.v > nb.v)
        return tm;
    return nb;
}

__global__ __launch_bounds__(kBlock) void k_pmis_find_max(int nrow, const int* __restrict__ rp,
                                                          const int* __restrict__ ci, const int* __restrict__ conn,
                                                          const int* __restrict__ state, const int* __restrict__ hash,
                                                          int* __restrict__ max_state, int* __restrict__ agg,
                                                          int* __restrict__ undecided)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        MisTuple t = {state[i], hash[i], (int)i};
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(conn[j])
            {
                const int      c  = ci[j];
                const MisTuple tj = {state[c], hash[c], c};
                t                 = mis_max(tj, t);
            }
        const int row = t.i; // distance two, through the distance-one maximum
        for(int j = rp[row]; j < rp[row + 1]; ++j)
            if(conn[j])
            {
                const int      c  = ci[j];
                const MisTuple tj = {state[c], hash[c], c};
                t                 = mis_max(tj, t);
            }
        if(state[i] == 0)
        {
            if(t.i == (int)i)
            {
                max_state[i] = 1;
                agg[i]       = 1;
            }
            else if(t.s == 1)
            {
                max_state[i] = -1;
                agg[i]       = 0;
            }
            else
                *undecided = 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_pmis_roots(int nrow, const int* __restrict__ agg, int* __restrict__ roots)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        roots[i] = (agg[i] == 1) ? (int)i : -1;
}

// rows in state -1 join the aggregate of their first strongly connected neighbour in state 1 (state: copy taken before
// the pass; the rows written here are in state -1, the rows read are in state 1: no overlap)
__global__ __launch_bounds__(kBlock) void k_pmis_add_unassigned(int nrow, const int* __restrict__ rp,
                                                                const int* __restrict__ ci,
                                                                const int* __restrict__ conn,
                                                                const int* __restrict__ state,
                                                                int* __restrict__ max_state, int* agg, int* roots)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        const int s = state[i];
        if(s == -1)
        {
            for(int j = rp[i]; j < rp[i + 1]; ++j)
                if(conn[j])
                {
                    const int c = ci[j];
                    if(state[c] == 1)
                    {
                        agg[i]       = agg[c];
                        max_state[i] = 1;
                        roots[i]     = roots[c];
                        break;
                    }
                }
        }
        else if(s == -2)
            agg[i] = -2;
    }
}

// ---- P of the unsmoothed aggregation: one entry 1 per aggregated row, column = rank of the aggregate's root node
__global__ __launch_bounds__(kBlock) void k_ua_count(int nrow, const int* __restrict__ agg, const int* __restrict__ roots,
                                                     int* __restrict__ prp, int* __restrict__ f2c)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
    {
        int c = 0;
        if(i < nrow && agg[i] >= 0)
        {
            c = 1;
            if(f2c)
                f2c[roots[i]] = 1; // same value from every member of the aggregate
        }
        prp[i] = c;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_ua_fill(int nrow, const int* __restrict__ agg, const int* __restrict__ roots,
                                                    const int* __restrict__ prp, const int* __restrict__ f2c,
                                                    int* __restrict__ pci, T* __restrict__ pval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        if(agg[i] >= 0)
        {
            pci[prp[i]]  = f2c ? f2c[roots[i]] : agg[i];
            pval[prp[i]] = (T)1;
        }
}

// ---- P of the smoothed aggregation (host_matrix_csr.cpp:5936-6330, local part): row i of (I - relax D_f^-1 A_f) P_tent,
// A_f the strength-filtered matrix with the weak couplings lumped into the diagonal.  Per row: the diagonal entry
// contributes 1 - relax, a strong entry -relax * (1/dia) * a_ij, keyed by the root node of the neighbour's aggregate;
// equal keys are summed in row order, keys ascend (std::map) -- here a stable insertion into the row's own scratch
// segment (its range of the operator's entries) followed by the in-order sum.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sa_row(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                   const T* __restrict__ val, const int* __restrict__ conn,
                                                   const int* __restrict__ agg, const int* __restrict__ roots, T relax,
                                                   int lumping, int* __restrict__ tkey, T* __restrict__ tval,
                                                   int* __restrict__ cnt, int* __restrict__ f2c)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
    {
        if(i == nrow)
        {
            cnt[i] = 0;
            continue;
        }
        const int rs = rp[i], re = rp[i + 1];
        T         dia = (T)0;
        for(int j = rs; j < re; ++j)
        {
            if(ci[j] == (int)i)
                dia += val[j];
            else if(!conn[j])
            {
                if(lumping == 0)
                    dia += val[j];
                else
                    dia -= val[j];
            }
        }
        dia = (T)1 / dia;
        int e = rs; // entries [rs, e) of the scratch segment are filled, sorted by key, stable
        for(int j = rs; j < re; ++j)
        {
            const int c = ci[j];
            if(c != (int)i && !conn[j])
                continue;
            if(agg[c] < 0)
                continue;
            const T   v   = (c == (int)i) ? (T)1 - relax : -relax * dia * val[j];
            const int key = f2c ? roots[c] : agg[c]; // (no table: the aggregate numbers are the coarse columns)
            if(f2c)
                f2c[key] = 1;
            int q         = e - 1;
            for(; q >= rs && tkey[q] > key; --q)
            {
                tkey[q + 1] = tkey[q];
                tval[q + 1] = tval[q];
            }
            tkey[q + 1] = key;
            tval[q + 1] = v;
            ++e;
        }
        int o = rs;
        for(int q = rs; q < e;)
        {
            const int k = tkey[q];
            T         v = tval[q];
            ++q;
            while(q < e && tkey[q] == k)
                v += tval[q++];
            tkey[o] = k;
            tval[o] = v;
            ++o;
        }
        cnt[i] = o - rs;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sa_compact(int nrow, const int* __restrict__ rp, const int* __restrict__ tkey,
                                                       const T* __restrict__ tval, const int* __restrict__ prp,
                                                       const int* __restrict__ f2c, int* __restrict__ pci,
                                                       T* __restrict__ pval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        const int n = prp[i + 1] - prp[i];
        for(int k = 0; k < n; ++k)
        {
            pci[prp[i] + k]  = f2c ? f2c[tkey[rp[i] + k]] : tkey[rp[i] + k];
            pval[prp[i] + k] = tval[rp[i] + k];
        }
    }
}

template <typename T>
static int sa_prolong_t(const ramd_mat_s* m, T relax, int lumping, const ramd_vec_s* vconn, const ramd_vec_s* vagg,
                        const ramd_vec_s* vroots, ramd_mat_s* p, int64_t global_ncol = -1)
{
    // global_ncol >= 0: the operator is the row block [interior | ghost] of a distributed matrix, the vectors cover its
    // columns and the aggregate numbers are global: they ARE the columns of P (no fine-to-coarse table)
    const bool gk = global_ncol >= 0;
    Backend&  b = backend();
    const int n = m->nrow;
    int *     tkey = nullptr, *cnt = nullptr, *f2c = nullptr;
    void*     tval = nullptr;
    RAMD_TRY(dev_alloc(&tkey, m->nnz));
    int s = dev_alloc(&cnt, (int64_t)n + 1);
    if(s == RAMD_OK && !gk)
        s = dev_alloc(&f2c, (int64_t)n + 1);
    if(s == RAMD_OK && cached_malloc(&tval, (size_t)m->nnz * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    hipError_t e = hipSuccess;
    int        tot[2] = {0, (int)global_ncol};
    if(s == RAMD_OK)
    {
        if(!gk)
            e = hipMemsetAsync(f2c, 0, sizeof(int) * ((size_t)n + 1), b.cur);
        hipLaunchKernelGGL((k_sa_row<T>), dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                           (const T*)m->val, (const int*)vconn->d, (const int*)vagg->d, (const int*)vroots->d, relax,
                           lumping, tkey, (T*)tval, cnt, f2c);
        s = device_exclusive_scan(cnt, cnt, (int64_t)n + 1);
        if(s == RAMD_OK && !gk)
            s = device_exclusive_scan(f2c, f2c, (int64_t)n + 1);
        if(s == RAMD_OK && e == hipSuccess)
            e = hipMemcpyAsync(&tot[0], cnt + n, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(s == RAMD_OK && e == hipSuccess && !gk)
            e = hipMemcpyAsync(&tot[1], f2c + n, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(s == RAMD_OK && e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
    }
    int*  pci = nullptr;
    void* pv  = nullptr;
    if(s == RAMD_OK && e == hipSuccess)
        s = dev_alloc(&pci, tot[0]);
    if(s == RAMD_OK && e == hipSuccess && cached_malloc(&pv, (size_t)tot[0] * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && e == hipSuccess)
    {
        hipLaunchKernelGGL((k_sa_compact<T>), dim3(ew_grid(std::max(n, 1))), dim3(kBlock), 0, b.cur, n, m->rp,
                           (const int*)tkey, (const T*)tval, (const int*)cnt, (const int*)f2c, pci, (T*)pv);
        e = hipGetLastError();
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
    }
    dev_free(&tkey);
    dev_free(&f2c);
    if(tval)
        (void)cached_free(tval);
    if(s != RAMD_OK || e != hipSuccess)
    {
        dev_free(&cnt);
        dev_free(&pci);
        if(pv)
            (void)cached_free(pv);
        RAMD_TRY(s);
        RAMD_HIP(e);
    }
    mat_free_csr(p);
    mat_free_ell(p);
    mat_free_coo(p);
    mat_free_analysis(p);
    p->format = RAMD_CSR;
    p->nrow   = n;
    p->ncol   = tot[1];
    p->nnz    = tot[0];
    p->rp     = cnt;
    p->ci     = pci;
    p->val    = pv;
    return RAMD_OK;
}

template <typename T>
static int pmis_aggregate_t(ramd_mat_s* m, T eps, ramd_vec_s* vconn, ramd_vec_s* vagg, ramd_vec_s* vroots)
{
    Backend&  b    = backend();
    const int n    = m->nrow;
    const int grid = ew_grid(std::max(n, 1));
    RAMD_TRY(ramd_vec_allocate(vconn, m->nnz));
    RAMD_TRY(ramd_vec_allocate(vagg, n)); // zero-filled, as LocalVector::Allocate
    RAMD_TRY(ramd_vec_allocate(vroots, n));
    int* conn  = (int*)vconn->d;
    int* agg   = (int*)vagg->d;
    int* roots = (int*)vroots->d;
    T*   diag  = nullptr;
    int *state = nullptr, *max_state = nullptr, *hash = nullptr, *flag = nullptr;
    int  s = dev_alloc(&diag, n);
    if(s == RAMD_OK)
        s = dev_alloc(&state, n);
    if(s == RAMD_OK)
        s = dev_alloc(&max_state, n);
    if(s == RAMD_OK)
        s = dev_alloc(&hash, n);
    if(s == RAMD_OK)
        s = dev_alloc(&flag, 1);
    hipError_t e = hipSuccess;
    if(s == RAMD_OK)
    {
        e = hipMemsetAsync(diag, 0, sizeof(T) * (size_t)n, b.cur); // ExtractDiagonal into a zero-filled vector
        hipLaunchKernelGGL((k_extract_diag_plain<T>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                           (const T*)m->val, diag);
        hipLaunchKernelGGL((k_amg_connections<T>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const T*)m->val,
                           (const T*)diag, eps * eps, conn);
        hipLaunchKernelGGL(k_pmis_init, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, (const int*)conn, max_state, hash);
        for(int iter = 0; e == hipSuccess; ++iter)
        {
            e = hipMemcpyAsync(state, max_state, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur);
            if(e == hipSuccess)
                e = hipMemsetAsync(flag, 0, sizeof(int), b.cur);
            hipLaunchKernelGGL(k_pmis_find_max, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const int*)conn,
                               (const int*)state, (const int*)hash, max_state, agg, flag);
            int undecided = 0;
            if(e == hipSuccess)
                e = hipMemcpyAsync(&undecided, flag, sizeof(int), hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(!undecided)
                break;
            if(iter > 10000)
            {
                s = RAMD_ERR_STATE;
                break;
            }
        }
        if(e == hipSuccess && s == RAMD_OK)
        {
            hipLaunchKernelGGL(k_pmis_roots, dim3(grid), dim3(kBlock), 0, b.cur, n, (const int*)agg, roots);
            // aggregates->ExclusiveSum(): rank of every root (the other rows are overwritten below)
            int* tmp = nullptr;
            s        = dev_alloc(&tmp, (int64_t)n + 1);
            if(s == RAMD_OK)
            {
                e = hipMemcpyAsync(tmp, agg, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur);
                if(e == hipSuccess)
                    e = hipMemsetAsync(tmp + n, 0, sizeof(int), b.cur);
                s = device_exclusive_scan(tmp, tmp, (int64_t)n + 1);
                if(s == RAMD_OK && e == hipSuccess)
                    e = hipMemcpyAsync(agg, tmp, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur);
                if(e == hipSuccess)
                    e = hipStreamSynchronize(b.cur);
            }
            dev_free(&tmp);
            for(int k = 0; k < 2 && s == RAMD_OK && e == hipSuccess; ++k)
            {
                e = hipMemcpyAsync(state, max_state, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur);
                hipLaunchKernelGGL(k_pmis_add_unassigned, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                                   (const int*)conn, (const int*)state, max_state, agg, roots);
            }
            if(e == hipSuccess)
                e = hipGetLastError();
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
        }
    }
    dev_free(&diag);
    dev_free(&state);
    dev_free(&max_state);
    dev_free(&hash);
    dev_free(&flag);
    RAMD_TRY(s);
    RAMD_HIP(e);
    return RAMD_OK;
}

// ---- Greedy aggregation (host_matrix_csr.cpp:4841-4938): the reference's sequential sweep -- row i becomes the seed of a
// new aggregate when it is still unassigned at its turn, claims its strong neighbours (overwriting earlier claims) and
// tentatively the unassigned strong neighbours of those -- restated as three parallel steps with the same result:
//  1. seeds = greedy distance-2 independent set in index order: i is a seed iff no seed s < i has i among its strong
//     neighbours or their strong neighbours.  Sync-free: a row polls the decisions of the lower-index rows of its 2-hop
//     neighbourhood (workgroups in natural order by ticket, so every awaited row is resident or done).
//  2. aggregate number = rank of the seed (device scan).
//  3. owner of a non-seed row: the LARGEST seed among its strong neighbours (direct claims overwrite in sweep order),
//     else the SMALLEST seed two strong hops away (the first tentative claim sticks).
// Needs a symmetric strong-connection graph (checked; RAMD_ERR_UNSUPPORTED otherwise).
__global__ __launch_bounds__(kBlock) void k_conn_symmetric(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                           const int* __restrict__ conn, int* __restrict__ asym)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(conn[j])
            {
                const int c     = ci[j];
                bool      found = false;
                for(int k = rp[c]; k < rp[c + 1]; ++k)
                    if(ci[k] == (int)i && conn[k])
                    {
                        found = true;
                        break;
                    }
                if(!found)
                    *asym = 1;
            }
}
// dec: 0 undecided, 1 seed, 2 covered, 3 removed (no strong connection)
__global__ __launch_bounds__(kBlock) void k_greedy_init(int nrow, const int* __restrict__ rp, const int* __restrict__ conn,
                                                        int* __restrict__ dec)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        int d = 3;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(conn[j])
            {
                d = 0;
                break;
            }
        dec[i] = d;
    }
}
// cov[h] = 1: a seed marked h as lying within two strong hops (an accelerator: the rows it covers stop walking their
// candidate lists, which are quadratic in the row length on the dense coarse operators of smoothed aggregation)
__global__ __launch_bounds__(kBlock) void k_greedy_seeds(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                         const int* __restrict__ conn, int* dec, int* cov,
                                                         unsigned* counter)
{
    const unsigned blk  = take_ticket(counter, 0u);
    const int64_t  t    = (int64_t)blk * kBlock + threadIdx.x;
    const bool     live = t < nrow;
    const int      i    = live ? (int)t : 0;
    bool           fin  = !live;
    if(live && __hip_atomic_load(dec + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3)
        fin = true;
    const int rs = live ? rp[i] : 0, re = live ? rp[i + 1] : 0;
    int       j  = rs; // cursor over my strong neighbours c
    int       k  = -1; // -1: c itself is the candidate; >= 0: cursor over the strong neighbours of c
    int       spins = 0, backoff = 1;
    do
    {
        spin_guard(spins);
        bool advanced = false;
        if(!fin && __hip_atomic_load(cov + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
        {
            __hip_atomic_store(dec + i, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fin = advanced = true;
        }
        if(!fin)
        {
            // walk the candidates until one is undecided (retry later), one is a seed (covered) or none is left (seed)
            while(true)
            {
                if(j >= re)
                {
                    __hip_atomic_store(dec + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    fin = advanced = true;
                    for(int a = rs; a < re; ++a) // mark my two-hop neighbourhood
                        if(conn[a])
                        {
                            const int c = ci[a];
                            if(c > i)
                                __hip_atomic_store(cov + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            for(int q = rp[c]; q < rp[c + 1]; ++q)
                                if(conn[q] && ci[q] > i)
                                    __hip_atomic_store(cov + ci[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    break;
                }
                if(!conn[j])
                {
                    ++j;
                    k = -1;
                    continue;
                }
                const int c = ci[j];
                int       u; // candidate
                if(k < 0)
                    u = c;
                else
                {
                    if(k >= rp[c + 1])
                    {
                        ++j;
                        k = -1;
                        continue;
                    }
                    if(!conn[k])
                    {
                        ++k;
                        continue;
                    }
                    u = ci[k];
                }
                if(u < i)
                {
                    const int d = __hip_atomic_load(dec + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if(d == 0)
                        break; // not decided yet: poll again
                    if(d == 1)
                    {
                        __hip_atomic_store(dec + i, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        fin = advanced = true;
                        break;
                    }
                }
                advanced = true;
                if(k < 0)
                    k = rp[c];
                else
                    ++k;
            }
        }
        backoff = poll_backoff(__ballot(advanced) != 0ull, backoff);
    } while(__ballot(!fin) != 0ull);
}
__global__ __launch_bounds__(kBlock) void k_greedy_flags(int nrow, const int* __restrict__ dec, int* __restrict__ flag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
        flag[i] = (i < nrow && dec[i] == 1) ? 1 : 0;
}
__global__ __launch_bounds__(kBlock) void k_greedy_assign(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                          const int* __restrict__ conn, const int* __restrict__ dec,
                                                          const int* __restrict__ rank, int* __restrict__ agg,
                                                          int* __restrict__ roots)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        const int d = dec[i];
        if(d == 3)
        {
            agg[i] = -2; // removed; the root entry keeps the zero of Allocate
            continue;
        }
        int owner = -1;
        if(d == 1)
            owner = (int)i;
        else
        {
            for(int j = rp[i]; j < rp[i + 1]; ++j) // direct claims: the last one in sweep order = the largest seed
                if(conn[j] && dec[ci[j]] == 1)
                    owner = max(owner, ci[j]);
            if(owner < 0)
            {
                int best = 0x7fffffff; // tentative claims: the first one = the smallest seed two hops away
                for(int j = rp[i]; j < rp[i + 1]; ++j)
                    if(conn[j])
                    {
                        const int c = ci[j];
                        for(int k = rp[c]; k < rp[c + 1]; ++k)
                            if(conn[k] && dec[ci[k]] == 1)
                                best = min(best, ci[k]);
                    }
                owner = best;
            }
        }
        agg[i]   = rank[owner];
        roots[i] = owner;
    }
}

template <typename T>
static int greedy_aggregate_t(ramd_mat_s* m, T eps, ramd_vec_s* vconn, ramd_vec_s* vagg, ramd_vec_s* vroots)
{
    Backend&  b    = backend();
    const int n    = m->nrow;
    const int grid = ew_grid(std::max(n, 1));
    RAMD_TRY(ramd_vec_allocate(vconn, m->nnz));
    RAMD_TRY(ramd_vec_allocate(vagg, n));
    RAMD_TRY(ramd_vec_allocate(vroots, n));
    int*      conn  = (int*)vconn->d;
    int*      agg   = (int*)vagg->d;
    int*      roots = (int*)vroots->d;
    T*        diag  = nullptr;
    int *     dec = nullptr, *rank = nullptr, *flag = nullptr, *cov = nullptr;
    unsigned* counter = nullptr;
    int       s = dev_alloc(&diag, n);
    if(s == RAMD_OK)
        s = dev_alloc(&dec, n);
    if(s == RAMD_OK)
        s = dev_alloc(&cov, n);
    if(s == RAMD_OK)
        s = dev_alloc(&rank, (int64_t)n + 1);
    if(s == RAMD_OK)
        s = dev_alloc(&flag, 1);
    if(s == RAMD_OK)
        s = dev_alloc(&counter, 1);
    hipError_t e = hipSuccess;
    int        asym = 0;
    if(s == RAMD_OK)
    {
        e = hipMemsetAsync(diag, 0, sizeof(T) * (size_t)n, b.cur);
        if(e == hipSuccess)
            e = hipMemsetAsync(flag, 0, sizeof(int), b.cur);
        if(e == hipSuccess)
            e = hipMemsetAsync(counter, 0, sizeof(unsigned), b.cur);
        hipLaunchKernelGGL((k_extract_diag_plain<T>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                           (const T*)m->val, diag);
        hipLaunchKernelGGL((k_amg_connections<T>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const T*)m->val,
                           (const T*)diag, eps * eps, conn);
        hipLaunchKernelGGL(k_conn_symmetric, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const int*)conn, flag);
        if(e == hipSuccess)
            e = hipMemcpyAsync(&asym, flag, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
    }
    if(s == RAMD_OK && e == hipSuccess && !asym)
    {
        hipLaunchKernelGGL(k_greedy_init, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, (const int*)conn, dec);
        e = hipMemsetAsync(cov, 0, sizeof(int) * (size_t)n, b.cur);
        const unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_greedy_seeds, dim3(nb), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const int*)conn, dec, cov,
                           counter);
        hipLaunchKernelGGL(k_greedy_flags, dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, (const int*)dec, rank);
        s = device_exclusive_scan(rank, rank, (int64_t)n + 1);
        if(s == RAMD_OK)
            hipLaunchKernelGGL(k_greedy_assign, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const int*)conn,
                               (const int*)dec, (const int*)rank, agg, roots);
        e = hipGetLastError();
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
    }
    dev_free(&diag);
    dev_free(&dec);
    dev_free(&cov);
    dev_free(&rank);
    dev_free(&flag);
    dev_free(&counter);
    RAMD_TRY(s);
    RAMD_HIP(e);
    if(asym)
        RAMD_FAIL(RAMD_ERR_UNSUPPORTED,
                  "AMGGreedyAggregate: the strong-connection graph is not symmetric (use CoarseningStrategy PMIS)");
    return RAMD_OK;
}

#ifdef RAMD_WITH_OFFSCOPE // (Ruge-Stueben AMG kernels: out of scope; built with RAMD_EXTRA_CXXFLAGS=-DRAMD_WITH_OFFSCOPE)
// ---- Ruge-Stueben AMG, PMIS coarsening + direct interpolation (host_matrix_csr.cpp: hash :7060-7071,
// RSPMISStrongInfluences :7074-7209, RSPMISUnassignedToCoarse :7212-7259, RSPMISCorrectCoarse :7262-7381,
// RSPMISCoarseEdgesToFine :7384-7459, RSPMISCheckUndecided :7462-7487, RSDirectProlongNnz :7501-7660,
// RSDirectProlongFill :7678-7930; drivers local_matrix.cpp RSPMISCoarsening / RSDirectInterpolation).  Every step of
// the reference is a row loop whose writes are idempotent (states only move one way inside a step), so each one is a
// kernel; omega = hash + number of strong in-edges is accumulated with float atomics of +1.0f (all increments equal:
// the sum does not depend on their order).
__device__ __forceinline__ float rs_hash(unsigned long long key)
{
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return (float)key / (float)0xffffffffffffffffULL;
}
__global__ __launch_bounds__(kBlock) void k_rs_omega_init(int nrow, float* __restrict__ omega, int* __restrict__ S,
                                                          int64_t nnz)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz || i < nrow; i += gsz)
    {
        if(i < nrow)
            omega[i] = rs_hash((unsigned long long)i);
        if(i < nnz)
            S[i] = 0;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_rs_strong(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      const T* __restrict__ val, float eps, int* __restrict__ S,
                                                      float* omega)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        T    mn = (T)0, mx = (T)0;
        bool sign = false;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const T v = val[j];
            if(ci[j] == (int)i)
                sign = v < (T)0;
            else
            {
                mn = (mn < v) ? mn : v;
                mx = (mx > v) ? mx : v;
            }
        }
        const T cond = (sign ? mx : mn) * (T)eps;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(ci[j] != (int)i && val[j] < cond)
            {
                S[j] = 1;
                atomicAdd(omega + ci[j], 1.0f);
            }
    }
}
__global__ __launch_bounds__(kBlock) void k_rs_unassigned_to_coarse(int nrow, int* __restrict__ cf,
                                                                    int* __restrict__ marked,
                                                                    const float* __restrict__ omega)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        int m = 0;
        if(cf[i] == 0)
        {
            if(omega[i] >= 1.0f)
            {
                cf[i] = 1;
                m     = 1;
            }
            else
                cf[i] = 2;
        }
        marked[i] = m;
    }
}
__global__ __launch_bounds__(kBlock) void k_rs_correct_coarse(int nrow, const int* __restrict__ rp,
                                                              const int* __restrict__ ci, const int* __restrict__ S,
                                                              const int* __restrict__ marked,
                                                              const float* __restrict__ omega, int* cf)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        if(marked[i])
        {
            const float wr = omega[i];
            for(int j = rp[i]; j < rp[i + 1]; ++j)
                if(S[j])
                {
                    const int c = ci[j];
                    if(marked[c])
                    {
                        const float wc = omega[c];
                        if(wr > wc)
                            cf[c] = 0; // only zeros are written in this step: the order of the rows does not matter
                        else if(wr < wc)
                            cf[i] = 0;
                    }
                }
        }
}
__global__ __launch_bounds__(kBlock) void k_rs_coarse_edges_to_fine(int nrow, const int* __restrict__ rp,
                                                                    const int* __restrict__ ci,
                                                                    const int* __restrict__ S, int* cf)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        if(cf[i] == 0)
            for(int j = rp[i]; j < rp[i + 1]; ++j)
                if(S[j] && cf[ci[j]] == 1) // (rows turn 0 -> 2 concurrently: the test for 1 is not affected)
                {
                    cf[i] = 2;
                    break;
                }
}
__global__ __launch_bounds__(kBlock) void k_rs_any_undecided(int nrow, const int* __restrict__ cf, int* __restrict__ flag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        if(cf[i] == 0)
            *flag = 1;
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_rs_direct_nnz(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                          const T* __restrict__ val, const int* __restrict__ cf,
                                                          const int* __restrict__ S, T* __restrict__ Amin,
                                                          T* __restrict__ Amax, int* __restrict__ f2c,
                                                          int* __restrict__ prp)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row <= nrow; row += gsz)
    {
        if(row == nrow)
        {
            f2c[row] = 0;
            prp[row] = 0;
            continue;
        }
        if(cf[row] == 1)
        {
            f2c[row] = 1;
            prp[row] = 1;
            continue;
        }
        f2c[row] = 0;
        T amin = (T)0, amax = (T)0;
        for(int j = rp[row]; j < rp[row + 1]; ++j)
        {
            if(!S[j] || cf[ci[j]] != 1)
                continue;
            amin = (amin < val[j]) ? amin : val[j];
            amax = (amax > val[j]) ? amax : val[j];
        }
        Amin[row] = amin = amin * (T)0.2f;
        Amax[row] = amax = amax * (T)0.2f;
        int nnz = 0;
        for(int j = rp[row]; j < rp[row + 1]; ++j)
            if(S[j] && cf[ci[j]] == 1)
                if(val[j] <= amin || val[j] >= amax)
                    ++nnz;
        prp[row] = nnz;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_rs_direct_fill(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                           const T* __restrict__ val, const int* __restrict__ cf,
                                                           const int* __restrict__ S, const T* __restrict__ Amin,
                                                           const T* __restrict__ Amax, const int* __restrict__ f2c,
                                                           const int* __restrict__ prp, int* __restrict__ pci,
                                                           T* __restrict__ pval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrow; row += gsz)
    {
        int row_P = prp[row];
        if(cf[row] == 1)
        {
            pci[row_P]  = f2c[row];
            pval[row_P] = (T)1;
            continue;
        }
        T diag = (T)0, a_num = (T)0, a_den = (T)0, b_num = (T)0, b_den = (T)0, d_neg = (T)0, d_pos = (T)0;
        const T amin = Amin[row], amax = Amax[row];
        for(int j = rp[row]; j < rp[row + 1]; ++j)
        {
            const int c = ci[j];
            const T   v = val[j];
            if(c == (int)row)
            {
                diag = v;
                continue;
            }
            if(v < (T)0)
            {
                a_num += v;
                if(S[j] && cf[c] == 1)
                {
                    a_den += v;
                    if(v > amin)
                        d_neg += v;
                }
            }
            else
            {
                b_num += v;
                if(S[j] && cf[c] == 1)
                {
                    b_den += v;
                    if(v < amax)
                        d_pos += v;
                }
            }
        }
        T cf_neg = (T)1, cf_pos = (T)1;
        {
            const T t1 = a_den - d_neg, t2 = b_den - d_pos;
            if((t1 < (T)0 ? -t1 : t1) > 1e-32)
                cf_neg = a_den / t1;
            if((t2 < (T)0 ? -t2 : t2) > 1e-32)
                cf_pos = b_den / t2;
        }
        if(b_num > (T)0 && (b_den < (T)0 ? -b_den : b_den) < 1e-32)
            diag += b_num;
        const T alpha = ((a_den < (T)0 ? -a_den : a_den) > 1e-32) ? -cf_neg * a_num / (diag * a_den) : (T)0;
        const T beta  = ((b_den < (T)0 ? -b_den : b_den) > 1e-32) ? -cf_pos * b_num / (diag * b_den) : (T)0;
        for(int j = rp[row]; j < rp[row + 1]; ++j)
        {
            const int c = ci[j];
            const T   v = val[j];
            if(S[j] && cf[c] == 1)
            {
                if(v > amin && v < amax)
                    continue;
                pci[row_P]  = f2c[c];
                pval[row_P] = (v < (T)0 ? alpha : beta) * v;
                ++row_P;
            }
        }
    }
}

template <typename T>
static int rs_pmis_t(ramd_mat_s* m, float eps, ramd_vec_s* vcf, ramd_vec_s* vS)
{
    Backend&  b    = backend();
    const int n    = m->nrow;
    const int grid = ew_grid(std::max(n, 1));
    RAMD_TRY(ramd_vec_allocate(vS, m->nnz));
    RAMD_TRY(ramd_vec_allocate(vcf, n)); // CFmap->Zeros()
    int*   S      = (int*)vS->d;
    int*   cf     = (int*)vcf->d;
    float* omega  = nullptr;
    int *  marked = nullptr, *flag = nullptr;
    int    s      = dev_alloc(&omega, n);
    if(s == RAMD_OK)
        s = dev_alloc(&marked, n);
    if(s == RAMD_OK)
        s = dev_alloc(&flag, 1);
    hipError_t e = hipSuccess;
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL(k_rs_omega_init, dim3(ew_grid(std::max<int64_t>(m->nnz, n))), dim3(kBlock), 0, b.cur, n, omega, S,
                           m->nnz);
        hipLaunchKernelGGL((k_rs_strong<T>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const T*)m->val, eps, S,
                           omega);
        for(int iter = 0; e == hipSuccess; ++iter)
        {
            hipLaunchKernelGGL(k_rs_unassigned_to_coarse, dim3(grid), dim3(kBlock), 0, b.cur, n, cf, marked,
                               (const float*)omega);
            hipLaunchKernelGGL(k_rs_correct_coarse, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const int*)S,
                               (const int*)marked, (const float*)omega, cf);
            hipLaunchKernelGGL(k_rs_coarse_edges_to_fine, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                               (const int*)S, cf);
            e = hipMemsetAsync(flag, 0, sizeof(int), b.cur);
            hipLaunchKernelGGL(k_rs_any_undecided, dim3(grid), dim3(kBlock), 0, b.cur, n, (const int*)cf, flag);
            int undecided = 0;
            if(e == hipSuccess)
                e = hipMemcpyAsync(&undecided, flag, sizeof(int), hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(!undecided)
                break;
            if(iter > 10000)
            {
                s = RAMD_ERR_STATE;
                break;
            }
        }
        if(e == hipSuccess)
            e = hipGetLastError();
    }
    dev_free(&omega);
    dev_free(&marked);
    dev_free(&flag);
    RAMD_TRY(s);
    RAMD_HIP(e);
    return RAMD_OK;
}

template <typename T>
static int rs_direct_t(const ramd_mat_s* m, const ramd_vec_s* vcf, const ramd_vec_s* vS, ramd_mat_s* p)
{
    Backend&   b  = backend();
    const int  n  = m->nrow;
    const int* cf = (const int*)vcf->d;
    const int* S  = (const int*)vS->d;
    T *        Amin = nullptr, *Amax = nullptr;
    int *      f2c = nullptr, *prp = nullptr;
    int        s = dev_alloc(&Amin, n);
    if(s == RAMD_OK)
        s = dev_alloc(&Amax, n);
    if(s == RAMD_OK)
        s = dev_alloc(&f2c, (int64_t)n + 1);
    if(s == RAMD_OK)
        s = dev_alloc(&prp, (int64_t)n + 1);
    int        tot[2] = {0, 0};
    int*       pci    = nullptr;
    void*      pv     = nullptr;
    hipError_t e      = hipSuccess;
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL((k_rs_direct_nnz<T>), dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                           (const T*)m->val, cf, S, Amin, Amax, f2c, prp);
        s = device_exclusive_scan(f2c, f2c, (int64_t)n + 1);
        if(s == RAMD_OK)
            s = device_exclusive_scan(prp, prp, (int64_t)n + 1);
        if(s == RAMD_OK)
            e = hipMemcpyAsync(&tot[0], prp + n, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(s == RAMD_OK && e == hipSuccess)
            e = hipMemcpyAsync(&tot[1], f2c + n, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(s == RAMD_OK && e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
    }
    if(s == RAMD_OK && e == hipSuccess)
        s = dev_alloc(&pci, tot[0]);
    if(s == RAMD_OK && e == hipSuccess && cached_malloc(&pv, (size_t)tot[0] * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && e == hipSuccess)
    {
        hipLaunchKernelGGL((k_rs_direct_fill<T>), dim3(ew_grid(std::max(n, 1))), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                           (const T*)m->val, cf, S, (const T*)Amin, (const T*)Amax, (const int*)f2c, (const int*)prp, pci,
                           (T*)pv);
        e = hipGetLastError();
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
    }
    dev_free(&Amin);
    dev_free(&Amax);
    dev_free(&f2c);
    if(s != RAMD_OK || e != hipSuccess)
    {
        dev_free(&prp);
        dev_free(&pci);
        if(pv)
            (void)cached_free(pv);
        RAMD_TRY(s);
        RAMD_HIP(e);
    }
    mat_free_csr(p);
    mat_free_ell(p);
    mat_free_coo(p);
    mat_free_analysis(p);
    p->format = RAMD_CSR;
    p->nrow   = n;
    p->ncol   = tot[1];
    p->nnz    = tot[0];
    p->rp     = prp;
    p->ci     = pci;
    p->val    = pv;
    return RAMD_OK;
}

#endif // RAMD_WITH_OFFSCOPE
template <typename T>
static int ua_prolong_t(const ramd_mat_s* m, const ramd_vec_s* vagg, const ramd_vec_s* vroots, ramd_mat_s* p,
                        int64_t global_ncol = -1)
{
    const bool gk = global_ncol >= 0; // (as in sa_prolong_t)
    Backend&   b     = backend();
    const int  n     = m->nrow;
    const int* agg   = (const int*)vagg->d;
    const int* roots = (const int*)vroots->d;
    int *      prp = nullptr, *f2c = nullptr;
    RAMD_TRY(dev_alloc(&prp, (int64_t)n + 1));
    int s = gk ? RAMD_OK : dev_alloc(&f2c, (int64_t)n + 1);
    if(s != RAMD_OK)
    {
        dev_free(&prp);
        return s;
    }
    hipError_t e = gk ? hipSuccess : hipMemsetAsync(f2c, 0, sizeof(int) * ((size_t)n + 1), b.cur);
    hipLaunchKernelGGL(k_ua_count, dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, agg, roots, prp, f2c);
    s = device_exclusive_scan(prp, prp, (int64_t)n + 1);
    if(s == RAMD_OK && !gk)
        s = device_exclusive_scan(f2c, f2c, (int64_t)n + 1);
    int tot[2] = {0, (int)global_ncol};
    if(s == RAMD_OK && e == hipSuccess)
        e = hipMemcpyAsync(&tot[0], prp + n, sizeof(int), hipMemcpyDeviceToHost, b.cur);
    if(s == RAMD_OK && e == hipSuccess && !gk)
        e = hipMemcpyAsync(&tot[1], f2c + n, sizeof(int), hipMemcpyDeviceToHost, b.cur);
    if(s == RAMD_OK && e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    int*  pci = nullptr;
    void* pv  = nullptr;
    if(s == RAMD_OK && e == hipSuccess)
        s = dev_alloc(&pci, tot[0]);
    if(s == RAMD_OK && e == hipSuccess && cached_malloc(&pv, (size_t)tot[0] * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && e == hipSuccess)
    {
        hipLaunchKernelGGL((k_ua_fill<T>), dim3(ew_grid(std::max(n, 1))), dim3(kBlock), 0, b.cur, n, agg, roots,
                           (const int*)prp, (const int*)f2c, pci, (T*)pv);
        e = hipGetLastError();
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
    }
    dev_free(&f2c);
    if(s != RAMD_OK || e != hipSuccess)
    {
        dev_free(&prp);
        dev_free(&pci);
        if(pv)
            (void)cached_free(pv);
        RAMD_TRY(s);
        RAMD_HIP(e);
    }
    mat_free_csr(p);
    mat_free_ell(p);
    mat_free_coo(p);
    mat_free_analysis(p);
    p->format = RAMD_CSR;
    p->nrow   = n;
    p->ncol   = tot[1];
    p->nnz    = tot[0];
    p->rp     = prp;
    p->ci     = pci;
    p->val    = pv;
    return RAMD_OK;
}


// ---- aggregation across the row blocks of a distributed matrix (global_matrix.cpp:2647-3121 AMGPMISAggregate; the
// ghost-aware halves of the kernels cited at the top of this file).  One rank's view: the n rows of its block as ONE
// operator [interior | ghost] with n + ng columns (k_merge_*: the interior entries of a row, then its ghost entries with
// their column moved up by n -- the order every reference loop visits them in), and every per-node array extended by the
// ng ghost nodes, whose entries arrive through the halo exchange of the matrix.  The local kernels above then serve
// unchanged wherever the reference's loops are "interior part, then ghost part" of the same body; what is new:
//   * the distance-two step through a ghost node: the reference ships the (state, hash, column) LIST of every boundary
//     row's strong neighbours and folds it at the receiver (AMGExtractBoundaryState / the bnd_* branch of
//     AMGPMISFindMaxNeighbourNode); the fold is associative with "the later entry wins a tie", so the owner folds the
//     list itself and ships ONE tuple per boundary row -- three ints instead of 3 x the row length;
//   * the tie rule of AMGPMISAddUnassignedNodesToAggregations between the first local root and a ghost root with a
//     smaller global number.
// Aggregate numbers and root nodes are GLOBAL (int: the global sizes must stay below 2^31); the numbering -- roots in
// global row order -- is the one a single rank produces, so the hierarchy does not depend on the number of ranks (equal
// hashes of two nodes within two hops aside: there the visiting order decides, and it differs at a rank boundary).
__global__ __launch_bounds__(kBlock) void k_merge_rp(int nrow, const int* __restrict__ rpi, const int* __restrict__ rpg,
                                                     int* __restrict__ rp)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
        rp[i] = rpi[i] + (rpg ? rpg[i] : 0);
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_merge_fill(int nrow, int ncol_i, const int* __restrict__ rpi,
                                                       const int* __restrict__ cii, const T* __restrict__ vi,
                                                       const int* __restrict__ rpg, const int* __restrict__ cig,
                                                       const T* __restrict__ vg, const int* __restrict__ rp,
                                                       int* __restrict__ ci, T* __restrict__ val)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        int o = rp[i];
        for(int j = rpi[i]; j < rpi[i + 1]; ++j, ++o)
        {
            ci[o]  = cii[j];
            val[o] = vi[j];
        }
        if(rpg)
            for(int j = rpg[i]; j < rpg[i + 1]; ++j, ++o)
            {
                ci[o]  = cig[j] + ncol_i;
                val[o] = vg[j];
            }
    }
}
template <typename T>
static int merge_columns_t(const ramd_mat_s* a, const ramd_mat_s* g, int ghost_ncol, ramd_mat_s* out)
{
    Backend&      b   = backend();
    const int     n   = a->nrow;
    const bool    hg  = g && g->nnz > 0;
    const int64_t nnz = a->nnz + (hg ? g->nnz : 0);
    int *         rp = nullptr, *ci = nullptr;
    void*         val = nullptr;
    RAMD_TRY(dev_alloc(&rp, (int64_t)n + 1));
    int s = dev_alloc(&ci, nnz);
    if(s == RAMD_OK && cached_malloc(&val, (size_t)nnz * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    hipError_t e = hipSuccess;
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL(k_merge_rp, dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, (const int*)a->rp,
                           hg ? (const int*)g->rp : (const int*)nullptr, rp);
        hipLaunchKernelGGL((k_merge_fill<T>), dim3(ew_grid(std::max(n, 1))), dim3(kBlock), 0, b.cur, n, a->ncol,
                           (const int*)a->rp, (const int*)a->ci, (const T*)a->val,
                           hg ? (const int*)g->rp : (const int*)nullptr, hg ? (const int*)g->ci : (const int*)nullptr,
                           hg ? (const T*)g->val : (const T*)nullptr, (const int*)rp, ci, (T*)val);
        e = hipGetLastError();
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
    }
    if(s != RAMD_OK || e != hipSuccess)
    {
        dev_free(&rp);
        dev_free(&ci);
        if(val)
            (void)cached_free(val);
        RAMD_TRY(s);
        RAMD_HIP(e);
    }
    mat_free_csr(out);
    mat_free_ell(out);
    mat_free_coo(out);
    mat_free_analysis(out);
    out->format = RAMD_CSR;
    out->nrow   = n;
    out->ncol   = a->ncol + ghost_ncol;
    out->nnz    = nnz;
    out->rp     = rp;
    out->ci     = ci;
    out->val    = val;
    return RAMD_OK;
}

template <typename U>
__global__ __launch_bounds__(kBlock) void k_pick(int64_t n, const int* __restrict__ idx, const U* __restrict__ src,
                                                 U* __restrict__ dst)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gsz)
        dst[s] = src[idx[s]];
}

// what one exchange of the operator's halo pattern needs (the arguments of ramd_comm_halo_begin_plan)
struct HaloArgs
{
    ramd_comm_t    comm;
    int            plan, npeers;
    const int*     peers;
    const int64_t *so, *ro;
    const int*     d_boundary; // device copy of the boundary index (nsend)
    int64_t        nsend, nrecv;
    void*          d_send; // 8 bytes per entry
};
// the ghost entries [n, n + ng) of a per-node array from the owners of the ghost nodes
template <typename U>
static int halo_extend(const HaloArgs& h, U* ext, int n)
{
    static_assert(sizeof(U) == 4 || sizeof(U) == 8, "4- or 8-byte entries");
    Backend& b = backend();
    if(h.nsend > 0)
        hipLaunchKernelGGL((k_pick<U>), dim3(ew_grid(h.nsend)), dim3(kBlock), 0, b.cur, h.nsend, h.d_boundary,
                           (const U*)ext, (U*)h.d_send);
    ramd_vec_s vs, vr;
    vs.dtype = vr.dtype = (sizeof(U) == 8) ? RAMD_F64 : RAMD_I32; // (the exchange moves bytes: 8 or 4 per entry)
    vs.n                = h.nsend;
    vs.d                = h.d_send;
    vr.n                = h.nrecv;
    vr.d                = ext + n; // (the receive buffer IS the ghost part of the array)
    RAMD_TRY(ramd_comm_halo_begin_plan(h.comm, h.plan, &vs, &vr, h.npeers, h.peers, h.so, h.ro));
    RAMD_TRY(ramd_comm_halo_end(h.comm));
    return RAMD_OK;
}

constexpr int kNoTuple = -3; // below every state: "this boundary row has no strong neighbour"

// per boundary row: the fold of (state, hash, global number) over its strong neighbours, in row order
__global__ __launch_bounds__(kBlock) void k_pmis_boundary_fold(int64_t nb, const int* __restrict__ boundary, int n,
                                                               const int* __restrict__ rp, const int* __restrict__ ci,
                                                               const int* __restrict__ conn,
                                                               const int* __restrict__ state,
                                                               const int* __restrict__ hash, int first_row,
                                                               const int* __restrict__ ghost_number,
                                                               int* __restrict__ out_s, int* __restrict__ out_v,
                                                               int* __restrict__ out_g)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nb; s += gsz)
    {
        const int row = boundary[s];
        MisTuple  t   = {kNoTuple, 0, -1};
        for(int j = rp[row]; j < rp[row + 1]; ++j)
            if(conn[j])
            {
                const int      c  = ci[j];
                const MisTuple tj = {state[c], hash[c], c < n ? first_row + c : ghost_number[c - n]};
                t                 = (t.s == kNoTuple) ? tj : mis_max(tj, t);
            }
        out_s[s] = t.s;
        out_v[s] = t.v;
        out_g[s] = t.i;
    }
}

// AMGPMISFindMaxNeighbourNode with the ghost branch: gs / gv / gg = the folded tuple of every ghost node
__global__ __launch_bounds__(kBlock) void k_pmis_find_max_global(int nrow, const int* __restrict__ rp,
                                                                 const int* __restrict__ ci,
                                                                 const int* __restrict__ conn,
                                                                 const int* __restrict__ state,
                                                                 const int* __restrict__ hash, int first_row,
                                                                 const int* __restrict__ gs, const int* __restrict__ gv,
                                                                 const int* __restrict__ gg, int* __restrict__ max_state,
                                                                 int* __restrict__ agg, int* __restrict__ undecided)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        MisTuple t = {state[i], hash[i], (int)i};
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(conn[j])
            {
                const int      c  = ci[j];
                const MisTuple tj = {state[c], hash[c], c};
                t                 = mis_max(tj, t);
            }
        if(t.i < nrow)
        {
            const int row = t.i;
            for(int j = rp[row]; j < rp[row + 1]; ++j)
                if(conn[j])
                {
                    const int      c  = ci[j];
                    const MisTuple tj = {state[c], hash[c], c};
                    t                 = mis_max(tj, t);
                }
        }
        else
        {
            const int g = t.i - nrow;
            if(gs[g] != kNoTuple)
            {
                const int      l  = gg[g] - first_row; // a row of this block, or "elsewhere"
                const MisTuple tj = {gs[g], gv[g], (l >= 0 && l < nrow) ? l : -1};
                t                 = mis_max(tj, t);
            }
        }
        if(state[i] == 0)
        {
            if(t.i == (int)i)
            {
                max_state[i] = 1;
                agg[i]       = 1;
            }
            else if(t.s == 1)
            {
                max_state[i] = -1;
                agg[i]       = 0;
            }
            else
                *undecided = 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_pmis_roots_global(int nrow, int first_row, const int* __restrict__ agg,
                                                              int* __restrict__ roots)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        roots[i] = (agg[i] == 1) ? first_row + (int)i : -1;
}
__global__ __launch_bounds__(kBlock) void k_iota(int64_t n, int* __restrict__ x, int first)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        x[i] = first + (int)i;
}
__global__ __launch_bounds__(kBlock) void k_add_const(int64_t n, int* __restrict__ x, int c)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        x[i] += c;
}

// AMGPMISAddUnassignedNodesToAggregations: the first strongly connected local root, unless a ghost root with a smaller
// global number follows in the row (then that one); without a local root the first ghost root
__global__ __launch_bounds__(kBlock) void k_pmis_add_unassigned_global(int nrow, const int* __restrict__ rp,
                                                                       const int* __restrict__ ci,
                                                                       const int* __restrict__ conn,
                                                                       const int* __restrict__ state, int first_row,
                                                                       const int* __restrict__ ghost_number,
                                                                       int* __restrict__ max_state, int* agg, int* roots)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        const int s = state[i];
        if(s == -1)
        {
            int  gcol  = -1;
            bool local = false;
            for(int j = rp[i]; j < rp[i + 1]; ++j)
                if(conn[j])
                {
                    const int c = ci[j];
                    if(state[c] != 1)
                        continue;
                    if(c < nrow)
                    {
                        if(local)
                            continue;
                        local = true;
                        gcol  = first_row + c;
                    }
                    else if(!(gcol == -1 || ghost_number[c - nrow] < gcol))
                        continue;
                    agg[i]       = agg[c];
                    max_state[i] = 1;
                    roots[i]     = roots[c];
                    if(c >= nrow)
                        break;
                }
        }
        else if(s == -2)
            agg[i] = -2;
    }
}

// exclusive prefix of one count per rank (an all-gather) and the total
static int ranks_prefix(ramd_comm_t comm, int64_t mine, int64_t* before, int64_t* total)
{
    int rank = 0, size = 1;
    RAMD_TRY(ramd_comm_rank(comm, &rank));
    RAMD_TRY(ramd_comm_size(comm, &size));
    std::vector<int64_t> v((size_t)size, 0);
    RAMD_TRY(ramd_comm_allgather_i64(comm, &mine, 1, v.data()));
    *before = 0;
    *total  = 0;
    for(int k = 0; k < size; ++k)
    {
        if(k < rank)
            *before += v[(size_t)k];
        *total += v[(size_t)k];
    }
    return RAMD_OK;
}
static int ranks_any(ramd_comm_t comm, int mine, int* any)
{
    const int slot = RAMD_NSCALARS - 2;
    RAMD_TRY(ramd_scalars_set(slot, mine ? 1.0 : 0.0));
    RAMD_TRY(ramd_comm_allreduce_scalars(comm, slot, 1));
    double r = 0.0;
    RAMD_TRY(ramd_scalars_fetch(&r, slot, 1));
    *any = r > 0.5;
    return RAMD_OK;
}

template <typename T>
static int pmis_aggregate_global_t(const ramd_mat_s* m, T eps, HaloArgs h, int first_row, ramd_vec_s* vnumber,
                                   ramd_vec_s* vconn, ramd_vec_s* vagg, ramd_vec_s* vroots, int64_t* agg_first,
                                   int64_t* agg_mine, int64_t* agg_total)
{
    Backend&      b    = backend();
    const int     n    = m->nrow;
    const int     ng   = m->ncol - n;
    const int64_t next = (int64_t)n + ng;
    const int     grid = ew_grid(std::max(n, 1));
    RAMD_TRY(ramd_vec_allocate(vconn, m->nnz));
    RAMD_TRY(ramd_vec_allocate(vagg, next)); // zero-filled
    RAMD_TRY(ramd_vec_allocate(vroots, next));
    RAMD_TRY(ramd_vec_allocate(vnumber, next)); // global number of every node of the extended block
    int* conn  = (int*)vconn->d;
    int* agg   = (int*)vagg->d;
    int* roots = (int*)vroots->d;
    int* const number       = (int*)vnumber->d;
    const int* ghost_number = number + n;
    T*   diag  = nullptr;
    int *state = nullptr, *max_state = nullptr, *hash = nullptr, *flag = nullptr, *bt = nullptr, *gt = nullptr,
        *tmp = nullptr;
    double* sb = nullptr;
    int     s  = dev_alloc(&diag, next);
    if(s == RAMD_OK)
        s = dev_alloc(&state, next);
    if(s == RAMD_OK)
        s = dev_alloc(&max_state, next);
    if(s == RAMD_OK)
        s = dev_alloc(&hash, next);
    if(s == RAMD_OK)
        s = dev_alloc(&flag, 1);
    if(s == RAMD_OK)
        s = dev_alloc(&bt, 3 * h.nsend);
    if(s == RAMD_OK)
        s = dev_alloc(&gt, 3 * (int64_t)ng + 3);
    if(s == RAMD_OK)
        s = dev_alloc(&sb, h.nsend);
    if(s == RAMD_OK)
        s = dev_alloc(&tmp, (int64_t)n + 1);
    h.d_send = sb;
    auto body = [&]() -> int {
        RAMD_HIP(hipMemsetAsync(diag, 0, sizeof(T) * (size_t)next, b.cur));
        RAMD_HIP(hipMemsetAsync(max_state, 0, sizeof(int) * (size_t)next, b.cur));
        hipLaunchKernelGGL(k_iota, dim3(grid), dim3(kBlock), 0, b.cur, (int64_t)n, number, first_row);
        RAMD_TRY(halo_extend<int>(h, number, n));
        hipLaunchKernelGGL((k_extract_diag_plain<T>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                           (const T*)m->val, diag);
        RAMD_TRY(halo_extend<T>(h, diag, n));
        hipLaunchKernelGGL((k_amg_connections<T>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const T*)m->val,
                           (const T*)diag, eps * eps, conn);
        hipLaunchKernelGGL(k_pmis_init, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, (const int*)conn, max_state, hash,
                           (unsigned)first_row);
        RAMD_TRY(halo_extend<int>(h, max_state, n));
        RAMD_TRY(halo_extend<int>(h, hash, n));
        int* const bs = bt, *const bv = bt + h.nsend, *const bg = bt + 2 * h.nsend;
        int* const gs = gt, *const gv = gt + ng, *const gg = gt + 2 * (int64_t)ng;
        for(int iter = 0;; ++iter)
        {
            RAMD_HIP(hipMemcpyAsync(state, max_state, sizeof(int) * (size_t)next, hipMemcpyDeviceToDevice, b.cur));
            RAMD_HIP(hipMemsetAsync(flag, 0, sizeof(int), b.cur));
            if(h.nsend > 0)
                hipLaunchKernelGGL(k_pmis_boundary_fold, dim3(ew_grid(h.nsend)), dim3(kBlock), 0, b.cur, h.nsend,
                                   h.d_boundary, n, m->rp, m->ci, (const int*)conn, (const int*)state, (const int*)hash,
                                   first_row, ghost_number, bs, bv, bg);
            // (the three fields of the tuples travel as three exchanges of the pattern: the plan fixes the entry count)
            for(int f = 0; f < 3; ++f)
            {
                ramd_vec_s vs, vr;
                vs.dtype = vr.dtype = RAMD_I32;
                vs.n                = h.nsend;
                vs.d                = bt + (int64_t)f * h.nsend;
                vr.n                = h.nrecv;
                vr.d                = gt + (int64_t)f * ng;
                RAMD_TRY(ramd_comm_halo_begin_plan(h.comm, h.plan, &vs, &vr, h.npeers, h.peers, h.so, h.ro));
                RAMD_TRY(ramd_comm_halo_end(h.comm));
            }
            hipLaunchKernelGGL(k_pmis_find_max_global, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                               (const int*)conn, (const int*)state, (const int*)hash, first_row, (const int*)gs,
                               (const int*)gv, (const int*)gg, max_state, agg, flag);
            RAMD_TRY(halo_extend<int>(h, max_state, n));
            int undecided = 0, any = 0;
            RAMD_HIP(hipMemcpyAsync(&undecided, flag, sizeof(int), hipMemcpyDeviceToHost, b.cur));
            RAMD_HIP(hipStreamSynchronize(b.cur));
            RAMD_TRY(ranks_any(h.comm, undecided, &any));
            if(!any)
                break;
            if(iter > 10000)
                return RAMD_ERR_STATE;
        }
        hipLaunchKernelGGL(k_pmis_roots_global, dim3(grid), dim3(kBlock), 0, b.cur, n, first_row, (const int*)agg, roots);
        RAMD_TRY(halo_extend<int>(h, roots, n));
        // aggregates: rank of every root among the roots of all ranks, in global row order
        RAMD_HIP(hipMemcpyAsync(tmp, agg, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur));
        RAMD_HIP(hipMemsetAsync(tmp + n, 0, sizeof(int), b.cur));
        RAMD_TRY(device_exclusive_scan(tmp, tmp, (int64_t)n + 1));
        int mine = 0;
        RAMD_HIP(hipMemcpyAsync(&mine, tmp + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
        RAMD_HIP(hipStreamSynchronize(b.cur));
        RAMD_TRY(ranks_prefix(h.comm, mine, agg_first, agg_total));
        *agg_mine = mine;
        if(*agg_total >= ((int64_t)1 << 31) - 1)
            RAMD_FAIL(RAMD_ERR_UNSUPPORTED, "AMGPMISAggregate: 2^31 aggregates and more");
        RAMD_HIP(hipMemcpyAsync(agg, tmp, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, b.cur));
        if(n > 0 && *agg_first != 0)
            hipLaunchKernelGGL(k_add_const, dim3(grid), dim3(kBlock), 0, b.cur, (int64_t)n, agg, (int)*agg_first);
        RAMD_TRY(halo_extend<int>(h, agg, n));
        for(int k = 0; k < 2; ++k)
        {
            RAMD_HIP(hipMemcpyAsync(state, max_state, sizeof(int) * (size_t)next, hipMemcpyDeviceToDevice, b.cur));
            hipLaunchKernelGGL(k_pmis_add_unassigned_global, dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                               (const int*)conn, (const int*)state, first_row, ghost_number, max_state, agg, roots);
            RAMD_TRY(halo_extend<int>(h, agg, n));
            RAMD_TRY(halo_extend<int>(h, roots, n));
            RAMD_TRY(halo_extend<int>(h, max_state, n));
        }
        RAMD_HIP(hipGetLastError());
        RAMD_HIP(hipStreamSynchronize(b.cur));
        return RAMD_OK;
    };
    if(s == RAMD_OK)
        s = body();
    dev_free(&diag);
    dev_free(&state);
    dev_free(&max_state);
    dev_free(&hash);
    dev_free(&flag);
    dev_free(&bt);
    dev_free(&gt);
    dev_free(&sb);
    dev_free(&tmp);
    return s;
}

} // namespace ramd

using namespace ramd;

extern "C" {

int ramd_mat_amg_pmis_aggregate(ramd_mat_t m, double eps, ramd_vec_t connections, ramd_vec_t aggregates,
                                ramd_vec_t aggregate_root_nodes)
{
    if(!m || !connections || !aggregates || !aggregate_root_nodes)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(connections->dtype != RAMD_I32 || aggregates->dtype != RAMD_I32 || aggregate_root_nodes->dtype != RAMD_I32)
        RAMD_FAIL(RAMD_ERR_ARG, "AMGPMISAggregate: int vectors expected");
    if(m->nrow != m->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "AMGPMISAggregate: square matrix expected");
    if(m->nnz <= 0)
        return RAMD_OK; // local_matrix.cpp:6545: nothing happens for an empty matrix
    if(m->dtype == RAMD_F64)
        return pmis_aggregate_t<double>(m, eps, connections, aggregates, aggregate_root_nodes);
    return pmis_aggregate_t<float>(m, (float)eps, connections, aggregates, aggregate_root_nodes);
}

int ramd_mat_amg_greedy_aggregate(ramd_mat_t m, double eps, ramd_vec_t connections, ramd_vec_t aggregates,
                                  ramd_vec_t aggregate_root_nodes)
{
    if(!m || !connections || !aggregates || !aggregate_root_nodes)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(connections->dtype != RAMD_I32 || aggregates->dtype != RAMD_I32 || aggregate_root_nodes->dtype != RAMD_I32)
        RAMD_FAIL(RAMD_ERR_ARG, "AMGGreedyAggregate: int vectors expected");
    if(m->nrow != m->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "AMGGreedyAggregate: square matrix expected");
    if(m->nnz <= 0)
        return RAMD_OK;
    if(m->dtype == RAMD_F64)
        return greedy_aggregate_t<double>(m, eps, connections, aggregates, aggregate_root_nodes);
    return greedy_aggregate_t<float>(m, (float)eps, connections, aggregates, aggregate_root_nodes);
}

#ifdef RAMD_WITH_OFFSCOPE // (Ruge-Stueben AMG: out of scope; built with RAMD_EXTRA_CXXFLAGS=-DRAMD_WITH_OFFSCOPE)
int ramd_mat_rs_pmis_coarsening(ramd_mat_t m, float eps, ramd_vec_t cfmap, ramd_vec_t S)
{
    if(!m || !cfmap || !S)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(cfmap->dtype != RAMD_I32 || S->dtype != RAMD_I32 || m->nrow != m->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "RSPMISCoarsening: square matrix and int vectors expected");
    if(m->nnz <= 0)
        return RAMD_OK;
    return (m->dtype == RAMD_F64) ? rs_pmis_t<double>(m, eps, cfmap, S) : rs_pmis_t<float>(m, eps, cfmap, S);
}

int ramd_mat_rs_direct_interpolation(ramd_mat_t m, ramd_vec_t cfmap, ramd_vec_t S, ramd_mat_t prolong)
{
    if(!m || !cfmap || !S || !prolong || prolong == m)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle / prolong aliases the operator");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(cfmap->dtype != RAMD_I32 || S->dtype != RAMD_I32 || cfmap->n != m->nrow || S->n != m->nnz
       || prolong->dtype != m->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "RSDirectInterpolation: int vectors of the operator's sizes, P of its value type");
    if(m->nnz <= 0)
        return RAMD_OK;
    return (m->dtype == RAMD_F64) ? rs_direct_t<double>(m, cfmap, S, prolong) : rs_direct_t<float>(m, cfmap, S, prolong);
}
#endif // RAMD_WITH_OFFSCOPE

int ramd_mat_amg_smoothed_prolong(ramd_mat_t m, double relax, int lumping_strat, ramd_vec_t connections,
                                  ramd_vec_t aggregates, ramd_vec_t aggregate_root_nodes, ramd_mat_t prolong)
{
    if(!m || !connections || !aggregates || !aggregate_root_nodes || !prolong || prolong == m)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle / prolong aliases the operator");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(connections->dtype != RAMD_I32 || aggregates->dtype != RAMD_I32 || aggregate_root_nodes->dtype != RAMD_I32
       || connections->n != m->nnz || aggregates->n != m->nrow || aggregate_root_nodes->n != m->nrow
       || prolong->dtype != m->dtype || !(relax > 0.0))
        RAMD_FAIL(RAMD_ERR_ARG, "AMGSmoothedAggregation: relax > 0, int vectors of the operator's sizes, P of its type");
    if(m->nnz <= 0)
        return RAMD_OK;
    if(m->dtype == RAMD_F64)
        return sa_prolong_t<double>(m, relax, lumping_strat, connections, aggregates, aggregate_root_nodes, prolong);
    return sa_prolong_t<float>(m, (float)relax, lumping_strat, connections, aggregates, aggregate_root_nodes, prolong);
}

int ramd_mat_amg_unsmoothed_prolong(ramd_mat_t m, ramd_vec_t aggregates, ramd_vec_t aggregate_root_nodes,
                                    ramd_mat_t prolong)
{
    if(!m || !aggregates || !aggregate_root_nodes || !prolong || prolong == m)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle / prolong aliases the operator");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(aggregates->dtype != RAMD_I32 || aggregate_root_nodes->dtype != RAMD_I32 || aggregates->n != m->nrow
       || aggregate_root_nodes->n != m->nrow || prolong->dtype != m->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "AMGUnsmoothedAggregation: int vectors of the operator's size, P of its value type");
    if(m->nnz <= 0)
        return RAMD_OK;
    if(m->dtype == RAMD_F64)
        return ua_prolong_t<double>(m, aggregates, aggregate_root_nodes, prolong);
    return ua_prolong_t<float>(m, aggregates, aggregate_root_nodes, prolong);
}

int ramd_mat_merge_columns(ramd_mat_t interior, ramd_mat_t ghost, int ghost_ncol, ramd_mat_t out)
{
    if(!interior || !out || out == interior || out == ghost)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle / the result aliases an operand");
    if(interior->format != RAMD_CSR || (ghost && ghost->nnz > 0 && ghost->format != RAMD_CSR))
        return RAMD_ERR_UNSUPPORTED;
    if(ghost_ncol < 0
       || (ghost && ghost->nnz > 0
           && (ghost->nrow != interior->nrow || ghost->dtype != interior->dtype || ghost->ncol > ghost_ncol)))
        RAMD_FAIL(RAMD_ERR_ARG, "merge_columns: the two parts of one row block expected");
    if(out->dtype != interior->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "merge_columns: result of the operands' value type expected");
    return (interior->dtype == RAMD_F64) ? merge_columns_t<double>(interior, ghost, ghost_ncol, out)
                                         : merge_columns_t<float>(interior, ghost, ghost_ncol, out);
}

int ramd_mat_amg_pmis_aggregate_global(ramd_mat_t block, double eps, ramd_comm_t comm, int plan, int npeers,
                                       const int* peers, const int64_t* send_offset, const int64_t* recv_offset,
                                       ramd_vec_t boundary, int64_t first_row, ramd_vec_t numbers,
                                       ramd_vec_t connections, ramd_vec_t aggregates, ramd_vec_t aggregate_root_nodes,
                                       int64_t* agg_first, int64_t* agg_mine, int64_t* agg_total)
{
    if(!block || !comm || !boundary || !numbers || !connections || !aggregates || !aggregate_root_nodes || !agg_first
       || !agg_mine || !agg_total)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle");
    if(block->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(boundary->dtype != RAMD_I32 || numbers->dtype != RAMD_I32 || connections->dtype != RAMD_I32
       || aggregates->dtype != RAMD_I32 || aggregate_root_nodes->dtype != RAMD_I32)
        RAMD_FAIL(RAMD_ERR_ARG, "AMGPMISAggregate: int vectors expected");
    const int64_t nsend = npeers > 0 ? send_offset[npeers] : 0, nrecv = npeers > 0 ? recv_offset[npeers] : 0;
    if(block->ncol - block->nrow != nrecv || boundary->n != nsend)
        RAMD_FAIL(RAMD_ERR_ARG, "AMGPMISAggregate: the row block [interior | ghost] and the halo pattern do not match");
    if(first_row + (int64_t)block->nrow >= ((int64_t)1 << 31) - 1)
        RAMD_FAIL(RAMD_ERR_UNSUPPORTED, "AMGPMISAggregate: global row numbers of 2^31 and more");
    HaloArgs h;
    h.comm       = comm;
    h.plan       = plan;
    h.npeers     = npeers;
    h.peers      = peers;
    h.so         = send_offset;
    h.ro         = recv_offset;
    h.d_boundary = (const int*)boundary->d;
    h.nsend      = nsend;
    h.nrecv      = nrecv;
    h.d_send = nullptr;
    if(block->dtype == RAMD_F64)
        return pmis_aggregate_global_t<double>(block, eps, h, (int)first_row, numbers, connections, aggregates,
                                               aggregate_root_nodes, agg_first, agg_mine, agg_total);
    return pmis_aggregate_global_t<float>(block, (float)eps, h, (int)first_row, numbers, connections, aggregates,
                                          aggregate_root_nodes, agg_first, agg_mine, agg_total);
}

int ramd_mat_amg_prolong_global(ramd_mat_t block, int smoothed, double relax, int lumping_strat, ramd_vec_t connections,
                                ramd_vec_t aggregates, ramd_vec_t aggregate_root_nodes, int64_t global_ncol,
                                ramd_mat_t prolong)
{
    if(!block || !connections || !aggregates || !aggregate_root_nodes || !prolong || prolong == block)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle / prolong aliases the operator");
    if(block->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(connections->dtype != RAMD_I32 || aggregates->dtype != RAMD_I32 || aggregate_root_nodes->dtype != RAMD_I32
       || (smoothed && connections->n != block->nnz) || aggregates->n != block->ncol
       || aggregate_root_nodes->n != block->ncol
       || prolong->dtype != block->dtype || global_ncol < 0 || global_ncol >= ((int64_t)1 << 31) - 1
       || (smoothed && !(relax > 0.0)))
        RAMD_FAIL(RAMD_ERR_ARG, "prolong_global: int vectors over the block's entries / columns, P of its value type");
    if(block->dtype == RAMD_F64)
        return smoothed ? sa_prolong_t<double>(block, relax, lumping_strat, connections, aggregates, aggregate_root_nodes,
                                               prolong, global_ncol)
                        : ua_prolong_t<double>(block, aggregates, aggregate_root_nodes, prolong, global_ncol);
    return smoothed ? sa_prolong_t<float>(block, (float)relax, lumping_strat, connections, aggregates,
                                          aggregate_root_nodes, prolong, global_ncol)
                    : ua_prolong_t<float>(block, aggregates, aggregate_root_nodes, prolong, global_ncol);
}

} // extern "C"
