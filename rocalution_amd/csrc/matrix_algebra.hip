// matrix_algebra.hip -- CSR matrix algebra of the LocalMatrix API (the building blocks of the reference's coarse-grid
// construction): Transpose, Sort, MatrixAdd, MatrixMult.  Reference: src/base/host/host_matrix_csr.cpp
//   Transpose :3743-3806 | Sort :3812-3846 | MatrixAdd :3324-3462 | MatMatMult :2805-2938
// Every result entry is produced by the same operations in the same order as the host loops (bit-exact values);
// rows are one thread each, the intermediate product list of MatrixMult lives in device memory.
#include "device_utils.hpp"
#include "matrix_impl.hpp"

#include <algorithm>
#include <vector>
#include <cstdlib>

namespace ramd
{

// stable insertion sort of every row by column (host: bubble sort, also stable)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sort_rows(int nrow, const int* __restrict__ rp, int* __restrict__ ci,
                                                      T* __restrict__ val)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += gsz)
        for(int a = rp[r] + 1; a < rp[r + 1]; ++a)
        {
            const int c = ci[a];
            const T   v = val[a];
            int       q = a - 1;
            for(; q >= rp[r] && ci[q] > c; --q)
            {
                ci[q + 1]  = ci[q];
                val[q + 1] = val[q];
            }
            ci[q + 1]  = c;
            val[q + 1] = v;
        }
}

// DiagonalMatrixMultR / L (host_matrix_csr.cpp:3631-3676): val[j] *= diag[col[j]] resp. diag[row]
template <typename T, bool LEFT>
__global__ __launch_bounds__(kBlock) void k_diag_mult(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      T* __restrict__ val, const T* __restrict__ diag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            val[j] *= LEFT ? diag[i] : diag[ci[j]];
}

#ifdef RAMD_WITH_OFFSCOPE // (FSAI / SPAI kernels: out of scope; built with RAMD_EXTRA_CXXFLAGS=-DRAMD_WITH_OFFSCOPE)
// ---- FSAI(1) factor (host_matrix_csr.cpp:6514-6662): for every row the dense system of the operator restricted to the
// row's lower pattern is factorised (in-place LU without pivoting, the host's loop order) and solved for the last unit
// vector; the row is then scaled by sqrt(1 / |last entry|).  One thread per row, dense scratch in device memory.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_fsai(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                 const T* __restrict__ val, const int* __restrict__ lrp,
                                                 const int* __restrict__ lci, T* __restrict__ lval,
                                                 const long long* __restrict__ soff, T* __restrict__ scratch)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t ai = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ai < nrow; ai += gsz)
    {
        const int base = lrp[ai];
        const int nr   = lrp[ai + 1] - base;
        if(nr == 1)
        {
            const int aj = rp[ai];
            if(ci[aj] == (int)ai)
                lval[base] = (T)1 / val[aj];
        }
        else if(nr > 1)
        {
            T* Asub = scratch + soff[ai];
            T* mk   = Asub + (long long)nr * nr;
            for(int q = 0; q < nr * nr; ++q)
                Asub[q] = (T)0;
            for(int k = 0; k < nr; ++k)
            {
                const int rk = lci[base + k];
                for(int aj = rp[rk]; aj < rp[rk + 1]; ++aj)
                {
                    for(int j = 0; j < nr; ++j)
                    {
                        const int ac = lci[base + j];
                        if(ci[aj] < ac)
                            break;
                        if(ci[aj] == ac)
                        {
                            Asub[j + k * nr] = val[aj];
                            break;
                        }
                    }
                    if(ci[aj] == (int)ai)
                        break;
                }
            }
            for(int q = 0; q < nr; ++q)
                mk[q] = (T)0;
            mk[nr - 1] = (T)1;
            for(int i = 0; i < nr - 1; ++i)
                for(int k = i + 1; k < nr; ++k)
                {
                    Asub[i + k * nr] /= Asub[i + i * nr];
                    for(int j = i + 1; j < nr; ++j)
                        Asub[j + k * nr] -= Asub[i + k * nr] * Asub[j + i * nr];
                }
            for(int i = nr - 1; i >= 0; --i)
            {
                mk[i] /= Asub[i + i * nr];
                for(int j = 0; j < i; ++j)
                    mk[j] -= mk[i] * Asub[i + j * nr];
            }
            for(int k = 0; k < nr; ++k)
                lval[base + k] = mk[k];
        }
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_fsai_scale(int nrow, const int* __restrict__ lrp, T* __restrict__ lval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t ai = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ai < nrow; ai += gsz)
    {
        if(lrp[ai + 1] == lrp[ai])
            continue;
        const T last = lval[lrp[ai + 1] - 1];
        const T fac  = (T)sqrt((double)((T)1 / (last < (T)0 ? -last : last)));
        for(int aj = lrp[ai]; aj < lrp[ai + 1]; ++aj)
            lval[aj] *= fac;
    }
}
__global__ __launch_bounds__(kBlock) void k_fsai_sizes(int nrow, const int* __restrict__ lrp, long long* __restrict__ sz)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
    {
        long long nr = (i < nrow) ? (lrp[i + 1] - lrp[i]) : 0;
        sz[i]        = nr > 1 ? nr * nr + nr : 0;
    }
}

// ---- SPAI (host_matrix_csr.cpp:6665-6780 with the dense QR of host_matrix_dense.cpp:361-520): on the pattern of A^T,
// row i of M^T minimises || e_i - A(I, J) m ||_2 with J = pattern of row i of A^T and I = the rows of A meeting J (in
// order of first appearance); Householder QR and back substitution in the host's loop order, one thread per row, dense
// scratch in device memory.  at* = CSR of A^T (the pattern that is filled), a* = CSR of A.
struct SpaiDims
{
    int nI, nJ;
};
// candidate k of the traversal "for idx in J: for j in row J[idx] of A^T" is new iff it did not occur earlier
__device__ __forceinline__ int spai_collect(int i, const int* __restrict__ atrp, const int* __restrict__ atci,
                                            int* __restrict__ Iout)
{
    int       nI = 0;
    const int rs = atrp[i], re = atrp[i + 1];
    for(int a = rs; a < re; ++a)
    {
        const int Ja = atci[a];
        for(int j = atrp[Ja]; j < atrp[Ja + 1]; ++j)
        {
            const int c    = atci[j];
            bool      seen = false;
            if(Iout)
            {
                for(int q = 0; q < nI; ++q)
                    if(Iout[q] == c)
                    {
                        seen = true;
                        break;
                    }
            }
            else
            {
                // no list yet: replay the traversal up to this position
                for(int a2 = rs; a2 <= a && !seen; ++a2)
                {
                    const int J2  = atci[a2];
                    const int end = (a2 == a) ? j : atrp[J2 + 1];
                    for(int j2 = atrp[J2]; j2 < end; ++j2)
                        if(atci[j2] == c)
                        {
                            seen = true;
                            break;
                        }
                }
            }
            if(!seen)
            {
                if(Iout)
                    Iout[nI] = c;
                ++nI;
            }
        }
    }
    return nI;
}
__device__ __forceinline__ long long spai_index_slots(long long nI, int tsz)
{
    return (nI * 4 + tsz - 1) / tsz + 1; // value-type slots holding the nI row indices
}
__global__ __launch_bounds__(kBlock) void k_spai_sizes(int r0, int r1, const int* __restrict__ atrp,
                                                       const int* __restrict__ atci, long long* __restrict__ sz, int tsz)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= r1; i += gsz)
    {
        long long v = 0;
        if(i < r1)
        {
            const long long nJ = atrp[i + 1] - atrp[i];
            const long long nI = spai_collect((int)i, atrp, atci, nullptr);
            v                  = nI * nJ + 2 * nI + nJ + spai_index_slots(nI, tsz); // Asub, v, ek, mk, I
        }
        sz[i - r0] = v;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_spai_rows(int r0, int r1, const int* __restrict__ atrp,
                                                      const int* __restrict__ atci, const int* __restrict__ arp,
                                                      const int* __restrict__ aci, const T* __restrict__ aval,
                                                      const long long* __restrict__ soff, T* __restrict__ scratch,
                                                      T* __restrict__ mval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r1; i += gsz)
    {
        const int rs = atrp[i];
        const int nJ = atrp[i + 1] - rs;
        if(nJ == 0)
            continue;
        T*   base = scratch + soff[i - r0];
        // layout: [I ints | Asub nI*nJ | v nI | ek nI | mk nJ]; nI known after collecting into the head of the segment
        int* I  = reinterpret_cast<int*>(base);
        const int nI = spai_collect((int)i, atrp, atci, I);
        T*   Asub = base + spai_index_slots(nI, (int)sizeof(T));
        T*   v    = Asub + (long long)nI * nJ;
        T*   ek   = v + nI;
        T*   mk   = ek + nI;
        for(int q = 0; q < nI * nJ; ++q)
            Asub[q] = (T)0;
        for(int k = 0; k < nI; ++k) // Asub(k, j) = A(I[k], J[j]), DENSE_IND(k, j) = k + j * nI
            for(int aj = arp[I[k]]; aj < arp[I[k] + 1]; ++aj)
                for(int j = 0; j < nJ; ++j)
                    if(aci[aj] == atci[rs + j])
                        Asub[k + j * nI] = aval[aj];
        const int size = nI < nJ ? nI : nJ;
        for(int q = 0; q < nI; ++q)
            v[q] = (T)0;
        // QRDecompose
        for(int c = 0; c < size; ++c)
        {
            T beta;
            T s = (T)0;
            for(int r = 1; r < nI - c; ++r)
                v[r] = Asub[(r + c) + c * nI];
            for(int r = c + 1; r < nI; ++r)
                s += v[r - c] * v[r - c];
            if(s == (T)0)
                beta = (T)0;
            else
            {
                T aii = Asub[c + c * nI];
                if(aii <= (T)0)
                    aii -= (T)sqrt((double)(aii * aii + s));
                else
                    aii += (T)sqrt((double)(aii * aii + s));
                const T squared = aii * aii;
                beta            = (T)2 * squared / (s + squared);
                aii             = (T)1 / aii;
                for(int r = 1; r < nI - c; ++r)
                    v[r] *= aii;
            }
            if(beta != (T)0)
            {
                for(int aj = c; aj < nJ; ++aj)
                {
                    T sum = Asub[c + aj * nI];
                    for(int ai = c + 1; ai < nI; ++ai)
                        sum += v[ai - c] * Asub[ai + aj * nI];
                    sum *= beta;
                    Asub[c + aj * nI] -= sum;
                    for(int ai = c + 1; ai < nI; ++ai)
                        Asub[ai + aj * nI] -= sum * v[ai - c];
                }
                for(int k = c + 1; k < nI; ++k)
                    Asub[k + c * nI] = v[k - c];
            }
        }
        // QRSolve(e_k, mk)
        for(int q = 0; q < nI; ++q)
            ek[q] = (I[q] == (int)i) ? (T)1 : (T)0;
        for(int q = 0; q < nJ; ++q)
            mk[q] = (T)0;
        for(int c = 0; c < size; ++c)
        {
            T sum = (T)1;
            for(int j = c + 1; j < nI; ++j)
                sum += Asub[j + c * nI] * Asub[j + c * nI];
            sum = (T)2 / sum;
            if(sum != (T)2)
            {
                T sum2 = ek[c];
                for(int j = c + 1; j < nI; ++j)
                    sum2 += Asub[j + c * nI] * ek[j];
                sum2 *= sum;
                ek[c] -= sum2;
                for(int j = c + 1; j < nI; ++j)
                    ek[j] -= sum2 * Asub[j + c * nI];
            }
        }
        for(int c = size - 1; c >= 0; --c)
        {
            T sum = (T)0;
            for(int j = c + 1; j < nJ; ++j)
                sum += Asub[c + j * nI] * mk[j];
            mk[c] = (ek[c] - sum) / Asub[c + c * nI];
        }
        for(int j = 0; j < nJ; ++j)
            mval[rs + j] = mk[j];
    }
}

#endif // RAMD_WITH_OFFSCOPE
// ---- MatrixAdd, pattern of `mat` a subset of this (structure == false): this = alpha*this + beta*mat on the matches
template <typename T>
__global__ __launch_bounds__(kBlock) void k_add_subset(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                       T* __restrict__ val, const int* __restrict__ brp,
                                                       const int* __restrict__ bci, const T* __restrict__ bval, T alpha,
                                                       T beta)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        int       first = brp[i];
        const int bend  = brp[i + 1];
        for(int ajj = rp[i]; ajj < rp[i + 1]; ++ajj)
            for(int aj = first; aj < bend; ++aj)
                if(bci[aj] == ci[ajj])
                {
                    val[ajj] = alpha * val[ajj] + beta * bval[aj];
                    ++first; // as the host loop: advanced by one per match (rows sorted)
                    break;
                }
    }
}

// ---- MatrixAdd with the union pattern (structure == true); rows sorted, columns unique
__global__ __launch_bounds__(kBlock) void k_union_count(int nrow, const int* __restrict__ arp,
                                                        const int* __restrict__ aci, const int* __restrict__ brp,
                                                        const int* __restrict__ bci, int* __restrict__ cnt)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
    {
        int c = 0;
        if(i < nrow)
        {
            int a = arp[i], b = brp[i];
            const int ae = arp[i + 1], be = brp[i + 1];
            while(a < ae || b < be)
            {
                const int ca = a < ae ? aci[a] : 0x7fffffff;
                const int cb = b < be ? bci[b] : 0x7fffffff;
                const int m  = min(ca, cb);
                if(ca == m)
                    ++a;
                if(cb == m)
                    ++b;
                ++c;
            }
        }
        cnt[i] = c;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_union_fill(int nrow, const int* __restrict__ arp,
                                                       const int* __restrict__ aci, const T* __restrict__ aval,
                                                       const int* __restrict__ brp, const int* __restrict__ bci,
                                                       const T* __restrict__ bval, T alpha, T beta,
                                                       const int* __restrict__ crp, int* __restrict__ cci,
                                                       T* __restrict__ cval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        int       a = arp[i], b = brp[i], o = crp[i];
        const int ae = arp[i + 1], be = brp[i + 1];
        while(a < ae || b < be)
        {
            const int ca = a < ae ? aci[a] : 0x7fffffff;
            const int cb = b < be ? bci[b] : 0x7fffffff;
            const int m  = min(ca, cb);
            T         v  = (T)0; // AllocateCSR zero-fills; then += alpha*A, += beta*B (host :3421-3447)
            if(ca == m)
                v += alpha * aval[a++];
            if(cb == m)
                v += beta * bval[b++];
            cci[o]  = m;
            cval[o] = v;
            ++o;
        }
    }
}

// ---- MatrixMult: C = A * B
// upper bound of products per row of C
__global__ __launch_bounds__(kBlock) void k_mm_bound(int nrow, const int* __restrict__ arp, const int* __restrict__ aci,
                                                     const int* __restrict__ brp, long long* __restrict__ ub)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
    {
        long long c = 0;
        if(i < nrow)
            for(int ja = arp[i]; ja < arp[i + 1]; ++ja)
                c += brp[aci[ja] + 1] - brp[aci[ja]];
        ub[i] = c;
    }
}
__global__ void k_scan_ll(int64_t n, long long* v) // exclusive scan, single thread per 1 block (n+1 small vs nnz work)
{
    // one workgroup, sequential over tiles: the array has nrow+1 entries and is touched once
    __shared__ long long carry;
    __shared__ long long tile[kBlock];
    if(threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for(int64_t base = 0; base < n; base += kBlock)
    {
        const int64_t i = base + threadIdx.x;
        const long long x = i < n ? v[i] : 0;
        tile[threadIdx.x] = x;
        __syncthreads();
        for(int o = 1; o < kBlock; o <<= 1) // Hillis-Steele inclusive scan
        {
            const long long y = threadIdx.x >= o ? tile[threadIdx.x - o] : 0;
            __syncthreads();
            tile[threadIdx.x] += y;
            __syncthreads();
        }
        if(i < n)
            v[i] = carry + tile[threadIdx.x] - x;
        __syncthreads();
        if(threadIdx.x == 0)
            carry += tile[kBlock - 1];
        __syncthreads();
    }
}
// products of row i in the host's order (ja ascending, jb ascending), then a stable insertion sort by column and the
// in-order sum of equal columns (= val[marker] = first product, += the later ones); cnt[i] = distinct columns
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mm_products(int nrow, const int* __restrict__ arp,
                                                        const int* __restrict__ aci, const T* __restrict__ aval,
                                                        const int* __restrict__ brp, const int* __restrict__ bci,
                                                        const T* __restrict__ bval, const long long* __restrict__ off,
                                                        int* __restrict__ pcol, T* __restrict__ pval,
                                                        int* __restrict__ cnt)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
    {
        if(i == nrow)
        {
            cnt[i] = 0;
            continue;
        }
        const long long s = off[i];
        long long       e = s;
        for(int ja = arp[i]; ja < arp[i + 1]; ++ja)
        {
            const int ca = aci[ja];
            const T   va = aval[ja];
            for(int jb = brp[ca]; jb < brp[ca + 1]; ++jb)
            {
                // insert (cb, va*vb) behind every entry with column <= cb: stable
                const int cb = bci[jb];
                const T   pv = va * bval[jb];
                long long q  = e - 1;
                for(; q >= s && pcol[q] > cb; --q)
                {
                    pcol[q + 1] = pcol[q];
                    pval[q + 1] = pval[q];
                }
                pcol[q + 1] = cb;
                pval[q + 1] = pv;
                ++e;
            }
        }
        // merge equal columns in place
        long long o = s;
        for(long long q = s; q < e;)
        {
            const int c = pcol[q];
            T         v = pval[q];
            ++q;
            while(q < e && pcol[q] == c)
                v += pval[q++];
            pcol[o] = c;
            pval[o] = v;
            ++o;
        }
        cnt[i] = (int)(o - s);
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mm_compact(int nrow, const long long* __restrict__ off,
                                                       const int* __restrict__ pcol, const T* __restrict__ pval,
                                                       const int* __restrict__ crp, int* __restrict__ cci,
                                                       T* __restrict__ cval, long long base = 0)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        const long long s = off[i] - base;
        const int       n = crp[i + 1] - crp[i];
        for(int k = 0; k < n; ++k)
        {
            cci[crp[i] + k]  = pcol[s + k];
            cval[crp[i] + k] = pval[s + k];
        }
    }
}

// exclusive scan of 64-bit counts: the multi-block 32-bit scan whenever the total fits (the usual case), else one block
__global__ __launch_bounds__(kBlock) void k_ll_total(int64_t n, const long long* __restrict__ v, unsigned long long* tot)
{
    unsigned long long acc = 0;
    const int64_t      gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        acc += (unsigned long long)v[i];
    for(int o = 32; o > 0; o >>= 1)
        acc += __shfl_xor(acc, o, 64);
    if((threadIdx.x & 63) == 0 && acc)
        atomicAdd(tot, acc); // integer sum: order does not matter
}
__global__ __launch_bounds__(kBlock) void k_ll_to_int(int64_t n, const long long* __restrict__ v, int* __restrict__ o)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        o[i] = (int)v[i];
}
__global__ __launch_bounds__(kBlock) void k_int_to_ll(int64_t n, const int* __restrict__ v, long long* __restrict__ o)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        o[i] = (long long)v[i];
}
static int scan_ll(long long* v, int64_t n)
{
    Backend&            b   = backend();
    unsigned long long* tot = nullptr;
    RAMD_TRY(dev_alloc(&tot, 1));
    unsigned long long h = 0;
    hipError_t         e = hipMemsetAsync(tot, 0, sizeof(unsigned long long), b.cur);
    hipLaunchKernelGGL(k_ll_total, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, (const long long*)v, tot);
    if(e == hipSuccess)
        e = hipMemcpyAsync(&h, tot, sizeof(h), hipMemcpyDeviceToHost, b.cur);
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&tot);
    RAMD_HIP(e);
    if(h >= 0x7fffffffULL)
    {
        hipLaunchKernelGGL(k_scan_ll, dim3(1), dim3(kBlock), 0, b.cur, n, v);
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    int* tmp = nullptr;
    RAMD_TRY(dev_alloc(&tmp, n));
    hipLaunchKernelGGL(k_ll_to_int, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, (const long long*)v, tmp);
    int s = device_exclusive_scan(tmp, tmp, n);
    if(s == RAMD_OK)
        hipLaunchKernelGGL(k_int_to_ll, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, (const int*)tmp, v);
    if(s == RAMD_OK && hipStreamSynchronize(b.cur) != hipSuccess)
        s = RAMD_ERR_HIP;
    dev_free(&tmp);
    return s;
}

// ---- MatrixMult, long rows: products in generation order, two stable sorts (by column, then by row) keep that order
// among equal (row, column) pairs, one thread per distinct pair sums its run left to right
__global__ __launch_bounds__(kBlock) void k_mm_any_long(int nrow, const long long* __restrict__ off, int limit,
                                                        int* __restrict__ flag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        if(off[i + 1] - off[i] > limit)
            *flag = 1;
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mm_generate(int r0, int r1, const int* __restrict__ arp,
                                                        const int* __restrict__ aci, const T* __restrict__ aval,
                                                        const int* __restrict__ brp, const int* __restrict__ bci,
                                                        const T* __restrict__ bval, const long long* __restrict__ off,
                                                        long long base, int* __restrict__ prow, int* __restrict__ pcol,
                                                        T* __restrict__ pval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r1; i += gsz)
    {
        long long q = off[i] - base;
        for(int ja = arp[i]; ja < arp[i + 1]; ++ja)
        {
            const int ca = aci[ja];
            const T   va = aval[ja];
            for(int jb = brp[ca]; jb < brp[ca + 1]; ++jb, ++q)
            {
                prow[q] = (int)(i - r0);
                pcol[q] = bci[jb];
                pval[q] = va * bval[jb];
            }
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_gather_int(int64_t n, const int* __restrict__ idx, const int* __restrict__ src,
                                                       int* __restrict__ dst)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gsz)
        dst[k] = src[idx[k]];
}
// perm[k] = o1[o2[k]]; head[k] = 1 where the (row, col) pair differs from the previous one; cnt[row] += heads
__global__ __launch_bounds__(kBlock) void k_mm_heads(int64_t n, const int* __restrict__ o1, const int* __restrict__ o2,
                                                     const int* __restrict__ prow, const int* __restrict__ pcol,
                                                     int* __restrict__ perm, int* __restrict__ head, int* __restrict__ cnt)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += gsz)
    {
        if(k == n)
        {
            head[k] = 0;
            continue;
        }
        const int p = o1[o2[k]];
        perm[k]     = p;
        bool h      = true;
        if(k > 0)
        {
            const int pp = o1[o2[k - 1]];
            h            = prow[pp] != prow[p] || pcol[pp] != pcol[p];
        }
        head[k] = h ? 1 : 0;
        if(h)
            atomicAdd(cnt + prow[p], 1);
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mm_reduce(int64_t n, const int* __restrict__ perm,
                                                      const int* __restrict__ headpos, const int* __restrict__ prow,
                                                      const int* __restrict__ pcol, const T* __restrict__ pval,
                                                      int* __restrict__ cci, T* __restrict__ cval)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gsz)
    {
        if(headpos[k + 1] == headpos[k]) // not the first product of its (row, col) pair
            continue;
        const int p = perm[k];
        const int r = prow[p], c = pcol[p];
        T         v = pval[p];
        for(int64_t q = k + 1; q < n; ++q)
        {
            const int pq = perm[q];
            if(prow[pq] != r || pcol[pq] != c)
                break;
            v += pval[pq];
        }
        cci[headpos[k]]  = c;
        cval[headpos[k]] = v;
    }
}

// ---- MatrixMult, medium rows (<= kMmCap products): one workgroup per row; products into LDS in generation order with
// the key (column, generation index), bitonic sort, one thread per distinct column sums its run in generation order.
// Writes the compacted row into the scratch segment like k_mm_products (same k_mm_compact afterwards).
constexpr int kMmCap = 2048;
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mm_row_lds(int r0, int nrow, const int* __restrict__ arp,
                                                       const int* __restrict__ aci, const T* __restrict__ aval,
                                                       const int* __restrict__ brp, const int* __restrict__ bci,
                                                       const T* __restrict__ bval, const long long* __restrict__ off,
                                                       long long base, int* __restrict__ pcol, T* __restrict__ pval,
                                                       int* __restrict__ cnt, int* __restrict__ toolong)
{
    __shared__ unsigned long long key[kMmCap];
    __shared__ T                  val[kMmCap];
    __shared__ int                aux[kMmCap + 1]; // prefix of the B-row lengths, later the output positions
    const int tid = threadIdx.x;
    // rows [r0, nrow): scratch positions relative to `base`, counts into cnt[row]
    for(int i = r0 + blockIdx.x; i < nrow; i += gridDim.x)
    {
        const long long s  = off[i] - base;
        const long long ul = off[i + 1] - off[i];
        const int       ra = arp[i], na = arp[i + 1] - ra;
        if(ul > kMmCap || na > kMmCap)
        {
            if(tid == 0)
            {
                *toolong = 1;
                cnt[i]   = 0;
            }
            continue;
        }
        const int ub = (int)ul;
        int       m  = 1;
        while(m < ub)
            m <<= 1;
        for(int t = tid; t < na; t += kBlock)
            aux[t] = brp[aci[ra + t] + 1] - brp[aci[ra + t]];
        __syncthreads();
        if(tid == 0) // exclusive prefix (rows of A are short)
        {
            int run = 0;
            for(int t = 0; t < na; ++t)
            {
                const int l = aux[t];
                aux[t]      = run;
                run += l;
            }
        }
        __syncthreads();
        for(int t = tid; t < na; t += kBlock)
        {
            const int ca = aci[ra + t];
            const T   va = aval[ra + t];
            int       q  = aux[t];
            for(int jb = brp[ca]; jb < brp[ca + 1]; ++jb, ++q)
            {
                key[q] = ((unsigned long long)(unsigned)bci[jb] << 32) | (unsigned)q;
                val[q] = va * bval[jb];
            }
        }
        for(int q = ub + tid; q < m; q += kBlock)
        {
            key[q] = ~0ull;
            val[q] = (T)0;
        }
        __syncthreads();
        for(int k = 2; k <= m; k <<= 1)
            for(int j = k >> 1; j > 0; j >>= 1)
            {
                for(int idx = tid; idx < m; idx += kBlock)
                {
                    const int ixj = idx ^ j;
                    if(ixj > idx)
                    {
                        const bool               up = (idx & k) == 0;
                        const unsigned long long a = key[idx], c = key[ixj];
                        if((a > c) == up)
                        {
                            key[idx] = c;
                            key[ixj] = a;
                            const T v = val[idx];
                            val[idx]  = val[ixj];
                            val[ixj]  = v;
                        }
                    }
                }
                __syncthreads();
            }
        // heads of the runs of equal columns -> output positions
        for(int idx = tid; idx < ub; idx += kBlock)
            aux[idx] = (idx == 0 || (unsigned)(key[idx - 1] >> 32) != (unsigned)(key[idx] >> 32)) ? 1 : 0;
        __syncthreads();
        if(tid == 0)
        {
            int run = 0;
            for(int idx = 0; idx < ub; ++idx)
            {
                const int h = aux[idx];
                aux[idx]    = h ? run : -1;
                run += h;
            }
            cnt[i] = run;
        }
        __syncthreads();
        for(int idx = tid; idx < ub; idx += kBlock)
            if(aux[idx] >= 0)
            {
                const unsigned c = (unsigned)(key[idx] >> 32);
                T              v = val[idx];
                for(int q = idx + 1; q < ub && (unsigned)(key[q] >> 32) == c; ++q)
                    v += val[q];
                pcol[s + aux[idx]] = (int)c;
                pval[s + aux[idx]] = v;
            }
        __syncthreads();
    }
}

static int scan_to_rowptr(int* rp, int nrow, int* total)
{
    Backend& b = backend();
    RAMD_TRY(device_exclusive_scan(rp, rp, (int64_t)nrow + 1));
    RAMD_HIP(hipMemcpyAsync(total, rp + nrow, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    RAMD_HIP(hipStreamSynchronize(b.cur));
    return RAMD_OK;
}

template <typename T>
static int matrix_add_t(ramd_mat_s* m, const ramd_mat_s* o, T alpha, T beta, bool structure)
{
    Backend&  b    = backend();
    const int grid = ew_grid(std::max(m->nrow, 1));
    if(!structure)
    {
        if(m->nnz > 0 && o->nnz > 0)
            hipLaunchKernelGGL((k_add_subset<T>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci, (T*)m->val,
                               o->rp, o->ci, (const T*)o->val, alpha, beta);
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    int* crp = nullptr;
    RAMD_TRY(dev_alloc(&crp, (int64_t)m->nrow + 1));
    hipLaunchKernelGGL(k_union_count, dim3(ew_grid((int64_t)m->nrow + 1)), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci,
                       o->rp, o->ci, crp);
    int nnz = 0;
    int s   = scan_to_rowptr(crp, m->nrow, &nnz);
    int*  cci = nullptr;
    void* cv  = nullptr;
    if(s == RAMD_OK)
        s = dev_alloc(&cci, nnz);
    if(s == RAMD_OK && cached_malloc(&cv, (size_t)nnz * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s != RAMD_OK)
    {
        dev_free(&crp);
        dev_free(&cci);
        return s;
    }
    hipLaunchKernelGGL((k_union_fill<T>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci, (const T*)m->val,
                       o->rp, o->ci, (const T*)o->val, alpha, beta, crp, cci, (T*)cv);
    hipError_t e = hipGetLastError();
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    const int nrow = m->nrow, ncol = m->ncol;
    mat_free_csr(m);
    mat_free_analysis(m);
    m->rp   = crp;
    m->ci   = cci;
    m->val  = cv;
    m->nnz  = nnz;
    m->nrow = nrow;
    m->ncol = ncol;
    RAMD_HIP(e);
    return RAMD_OK;
}

// one chunk of rows [r0, r1) of C through the sorted path; the rows' counts go to cnt[r0..r1), the compacted entries of
// the chunk (contiguous in C, rows being contiguous) to freshly allocated arrays
template <typename T>
static int mat_mult_sorted_chunk(const ramd_mat_s* a, const ramd_mat_s* bm, const long long* off, int r0, int r1,
                                 long long base, int64_t P, int* cnt, int** cci_out, void** cv_out, int* nnz_out)
{
    Backend&  b  = backend();
    const int nr = r1 - r0;
    *cci_out     = nullptr;
    *cv_out      = nullptr;
    *nnz_out     = 0;
    if(P <= 0 || nr <= 0)
        return RAMD_OK;
    int * prow = nullptr, *pcol = nullptr, *o1 = nullptr, *o2 = nullptr, *k2 = nullptr, *head = nullptr;
    void* pval = nullptr;
    int   s    = dev_alloc(&prow, P);
    if(s == RAMD_OK)
        s = dev_alloc(&pcol, P);
    if(s == RAMD_OK)
        s = dev_alloc(&o1, P);
    if(s == RAMD_OK)
        s = dev_alloc(&o2, P);
    if(s == RAMD_OK)
        s = dev_alloc(&k2, P);
    if(s == RAMD_OK)
        s = dev_alloc(&head, P + 1);
    if(s == RAMD_OK && cached_malloc(&pval, (size_t)P * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    int   nnz = 0;
    int*  cci = nullptr;
    void* cv  = nullptr;
    if(s == RAMD_OK)
    {
        const int gp = ew_grid(std::max<int64_t>(P, 1));
        hipLaunchKernelGGL((k_mm_generate<T>), dim3(ew_grid(nr)), dim3(kBlock), 0, b.cur, r0, r1, a->rp, a->ci,
                           (const T*)a->val, bm->rp, bm->ci, (const T*)bm->val, off, base, prow, pcol, (T*)pval);
        s = device_stable_sort_by_key(pcol, P, std::max(bm->ncol - 1, 0), o1);
        if(s == RAMD_OK)
        {
            hipLaunchKernelGGL(k_gather_int, dim3(gp), dim3(kBlock), 0, b.cur, P, (const int*)o1, (const int*)prow, k2);
            s = device_stable_sort_by_key(k2, P, std::max(nr - 1, 0), o2);
        }
        if(s == RAMD_OK)
        {
            // k2 is reused as the composed permutation
            hipLaunchKernelGGL(k_mm_heads, dim3(ew_grid(P + 1)), dim3(kBlock), 0, b.cur, P, (const int*)o1, (const int*)o2,
                               (const int*)prow, (const int*)pcol, k2, head, cnt + r0);
            s = device_exclusive_scan(head, head, P + 1);
        }
        if(s == RAMD_OK)
        {
            hipError_t e = hipMemcpyAsync(&nnz, head + P, sizeof(int), hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
        }
        if(s == RAMD_OK)
            s = dev_alloc(&cci, nnz);
        if(s == RAMD_OK && cached_malloc(&cv, (size_t)nnz * sizeof(T) + kPad) != hipSuccess)
            s = RAMD_ERR_HIP;
        if(s == RAMD_OK)
        {
            hipLaunchKernelGGL((k_mm_reduce<T>), dim3(gp), dim3(kBlock), 0, b.cur, P, (const int*)k2, (const int*)head,
                               (const int*)prow, (const int*)pcol, (const T*)pval, cci, (T*)cv);
            hipError_t e = hipGetLastError();
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
        }
    }
    dev_free(&prow);
    dev_free(&pcol);
    dev_free(&o1);
    dev_free(&o2);
    dev_free(&k2);
    dev_free(&head);
    if(pval)
        (void)cached_free(pval);
    if(s != RAMD_OK)
    {
        dev_free(&cci);
        if(cv)
            (void)cached_free(cv);
        return s;
    }
    *cci_out = cci;
    *cv_out  = cv;
    *nnz_out = nnz;
    return RAMD_OK;
}

// one chunk through the LDS kernel: compacted rows into a scratch of P entries, then into a piece; *toolong: a row of
// the chunk exceeds the LDS capacity (the caller sends the chunk through the global sort instead)
template <typename T>
static int mat_mult_lds_chunk(const ramd_mat_s* a, const ramd_mat_s* bm, const long long* off, int r0, int r1,
                              long long base, int64_t P, int* cnt, int** cci_out, void** cv_out, int* nnz_out,
                              int* toolong)
{
    Backend&  b  = backend();
    const int nr = r1 - r0;
    *cci_out     = nullptr;
    *cv_out      = nullptr;
    *nnz_out     = 0;
    *toolong     = 0;
    if(P <= 0 || nr <= 0)
        return RAMD_OK;
    int * pcol = nullptr, *flag = nullptr, *crp = nullptr;
    void* pval = nullptr;
    int   s    = dev_alloc(&pcol, P);
    if(s == RAMD_OK)
        s = dev_alloc(&flag, 1);
    if(s == RAMD_OK)
        s = dev_alloc(&crp, (int64_t)nr + 1);
    if(s == RAMD_OK && cached_malloc(&pval, (size_t)P * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    int*  cci = nullptr;
    void* cv  = nullptr;
    int   nnz = 0;
    if(s == RAMD_OK)
    {
        hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), b.cur);
        const int  g = (int)std::min<int64_t>((int64_t)nr, (int64_t)backend().num_cu * 8);
        hipLaunchKernelGGL((k_mm_row_lds<T>), dim3(g), dim3(kBlock), 0, b.cur, r0, r1, a->rp, a->ci, (const T*)a->val,
                           bm->rp, bm->ci, (const T*)bm->val, off, base, pcol, (T*)pval, cnt, flag);
        if(e == hipSuccess)
            e = hipMemcpyAsync(toolong, flag, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    if(s == RAMD_OK && !*toolong)
    {
        hipError_t e = hipMemcpyAsync(crp, cnt + r0, sizeof(int) * (size_t)nr, hipMemcpyDeviceToDevice, b.cur);
        if(e == hipSuccess)
            e = hipMemsetAsync(crp + nr, 0, sizeof(int), b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
        if(s == RAMD_OK)
            s = scan_to_rowptr(crp, nr, &nnz);
        if(s == RAMD_OK)
            s = dev_alloc(&cci, nnz);
        if(s == RAMD_OK && cached_malloc(&cv, (size_t)nnz * sizeof(T) + kPad) != hipSuccess)
            s = RAMD_ERR_HIP;
        if(s == RAMD_OK)
        {
            hipLaunchKernelGGL((k_mm_compact<T>), dim3(ew_grid(nr)), dim3(kBlock), 0, b.cur, nr, off + r0, (const int*)pcol,
                               (const T*)pval, (const int*)crp, cci, (T*)cv, base);
            e = hipGetLastError();
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
        }
    }
    dev_free(&pcol);
    dev_free(&flag);
    dev_free(&crp);
    if(pval)
        (void)cached_free(pval);
    if(s != RAMD_OK || *toolong)
    {
        dev_free(&cci);
        if(cv)
            (void)cached_free(cv);
        return s;
    }
    *cci_out = cci;
    *cv_out  = cv;
    *nnz_out = nnz;
    return RAMD_OK;
}

// long rows: row chunks of at most RAMD_MM_CHUNK products (bounded, reusable scratch; int32 sort indices); every chunk
// goes through the LDS kernel, or through the global sort when one of its rows exceeds the LDS capacity
template <typename T>
static int mat_mult_chunked_t(ramd_mat_s* c, const ramd_mat_s* a, const ramd_mat_s* bm, const long long* off)
{
    Backend&  b = backend();
    const int n = a->nrow;
    static long long chunk_products = -1;
    if(chunk_products < 0)
    {
        const char* e  = getenv("RAMD_MM_CHUNK"); // products per chunk (tests force tiny chunks)
        chunk_products = e ? atoll(e) : (1ll << 28);
        if(chunk_products < 1)
            chunk_products = 1;
    }
    static int use_lds = -1;
    if(use_lds < 0)
    {
        const char* e2 = getenv("RAMD_MM_LDS");
        use_lds        = e2 ? atoi(e2) : 1;
    }
    auto read_off = [&](int i, long long* v) -> int {
        RAMD_HIP(hipMemcpyAsync(v, off + i, sizeof(long long), hipMemcpyDeviceToHost, b.cur));
        RAMD_HIP(hipStreamSynchronize(b.cur));
        return RAMD_OK;
    };
    int* cnt = nullptr;
    RAMD_TRY(dev_alloc(&cnt, (int64_t)n + 1));
    int s = RAMD_OK;
    if(hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)n + 1), b.cur) != hipSuccess)
        s = RAMD_ERR_HIP;
    struct Piece
    {
        int*  ci;
        void* val;
        int   nnz;
    };
    std::vector<Piece> pieces;
    int                r0   = 0;
    long long          base = 0;
    long long          nnz_total = 0;
    while(s == RAMD_OK && r0 < n)
    {
        // largest r1 with off[r1] - base <= chunk_products (at least one row)
        int lo = r0 + 1, hi = n;
        long long v = 0;
        s = read_off(hi, &v);
        if(s != RAMD_OK)
            break;
        if(v - base > chunk_products)
        {
            while(lo < hi)
            {
                const int mid = lo + (hi - lo + 1) / 2;
                s             = read_off(mid, &v);
                if(s != RAMD_OK)
                    break;
                if(v - base <= chunk_products)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            if(s != RAMD_OK)
                break;
            hi = lo;
            s  = read_off(hi, &v);
            if(s != RAMD_OK)
                break;
        }
        const int       r1 = hi;
        const long long P  = v - base;
        if(P >= 0x7fffffffLL)
        {
            s = RAMD_ERR_UNSUPPORTED; // a single row with more than 2^31 products
            break;
        }
        Piece pc      = {nullptr, nullptr, 0};
        int   toolong = 1;
        if(use_lds)
            s = mat_mult_lds_chunk<T>(a, bm, off, r0, r1, base, (int64_t)P, cnt, &pc.ci, &pc.val, &pc.nnz, &toolong);
        if(s == RAMD_OK && toolong)
        {
            if(hipMemsetAsync(cnt + r0, 0, sizeof(int) * (size_t)(r1 - r0), b.cur) != hipSuccess)
                s = RAMD_ERR_HIP;
            if(s == RAMD_OK)
                s = mat_mult_sorted_chunk<T>(a, bm, off, r0, r1, base, (int64_t)P, cnt, &pc.ci, &pc.val, &pc.nnz);
        }
        if(s == RAMD_OK)
        {
            pieces.push_back(pc);
            nnz_total += pc.nnz;
        }
        r0   = r1;
        base = v;
    }
    if(s == RAMD_OK && nnz_total >= 0x7fffffffLL)
        s = RAMD_ERR_UNSUPPORTED;
    int   nnz = 0;
    int*  cci = nullptr;
    void* cv  = nullptr;
    if(s == RAMD_OK)
        s = scan_to_rowptr(cnt, n, &nnz);
    if(s == RAMD_OK)
        s = dev_alloc(&cci, nnz);
    if(s == RAMD_OK && cached_malloc(&cv, (size_t)nnz * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK)
    {
        size_t     at = 0;
        hipError_t e  = hipSuccess;
        for(size_t k = 0; k < pieces.size() && e == hipSuccess; ++k)
        {
            if(pieces[k].nnz > 0)
            {
                e = hipMemcpyAsync(cci + at, pieces[k].ci, sizeof(int) * (size_t)pieces[k].nnz, hipMemcpyDeviceToDevice, b.cur);
                if(e == hipSuccess)
                    e = hipMemcpyAsync((char*)cv + at * sizeof(T), pieces[k].val, sizeof(T) * (size_t)pieces[k].nnz,
                                       hipMemcpyDeviceToDevice, b.cur);
            }
            at += (size_t)pieces[k].nnz;
        }
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    for(size_t k = 0; k < pieces.size(); ++k)
    {
        dev_free(&pieces[k].ci);
        if(pieces[k].val)
            (void)cached_free(pieces[k].val);
    }
    if(s != RAMD_OK)
    {
        dev_free(&cnt);
        dev_free(&cci);
        if(cv)
            (void)cached_free(cv);
        if(s == RAMD_ERR_UNSUPPORTED)
            RAMD_FAIL(RAMD_ERR_UNSUPPORTED, "MatrixMult: result or a single row beyond the 32-bit index range");
        return s;
    }
    mat_free_csr(c);
    mat_free_ell(c);
    mat_free_coo(c);
    mat_free_analysis(c);
    c->format = RAMD_CSR;
    c->nrow   = a->nrow;
    c->ncol   = bm->ncol;
    c->nnz    = nnz;
    c->rp     = cnt;
    c->ci     = cci;
    c->val    = cv;
    return RAMD_OK;
}

template <typename T>
static int mat_mult_t(ramd_mat_s* c, const ramd_mat_s* a, const ramd_mat_s* bm)
{
    Backend&   b    = backend();
    const int  n    = a->nrow;
    long long* off  = nullptr;
    int*       cnt  = nullptr;
    int*       pcol = nullptr;
    void*      pval = nullptr;
    RAMD_TRY(dev_alloc(&off, (int64_t)n + 1));
    int s = dev_alloc(&cnt, (int64_t)n + 1);
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL(k_mm_bound, dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, a->rp, a->ci, bm->rp, off);
        s = scan_ll(off, (int64_t)n + 1);
    }
    long long total = 0;
    if(s == RAMD_OK)
    {
        hipError_t e = hipMemcpyAsync(&total, off + n, sizeof(long long), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    if(s == RAMD_OK)
    {
        // rows with many products: the per-thread insertion below is quadratic in the row length
        static int limit = -1;
        if(limit < 0)
        {
            const char* e = getenv("RAMD_MM_INSERT_LIMIT");
            limit         = e ? atoi(e) : 64;
        }
        int  any  = 0;
        int* flag = nullptr;
        s         = dev_alloc(&flag, 1);
        if(s == RAMD_OK)
        {
            hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), b.cur);
            hipLaunchKernelGGL(k_mm_any_long, dim3(ew_grid(std::max(n, 1))), dim3(kBlock), 0, b.cur, n, off, limit, flag);
            if(e == hipSuccess)
                e = hipMemcpyAsync(&any, flag, sizeof(int), hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
        }
        dev_free(&flag);
        if(s == RAMD_OK && any)
        {
            s = mat_mult_chunked_t<T>(c, a, bm, off);
            dev_free(&off);
            dev_free(&cnt);
            return s;
        }
    }
    if(s == RAMD_OK)
        s = dev_alloc(&pcol, total);
    if(s == RAMD_OK && cached_malloc(&pval, (size_t)total * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    int nnz = 0;
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL((k_mm_products<T>), dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, a->rp, a->ci,
                           (const T*)a->val, bm->rp, bm->ci, (const T*)bm->val, off, pcol, (T*)pval, cnt);
        s = scan_to_rowptr(cnt, n, &nnz);
    }
    int*  cci = nullptr;
    void* cv  = nullptr;
    if(s == RAMD_OK)
        s = dev_alloc(&cci, nnz);
    if(s == RAMD_OK && cached_malloc(&cv, (size_t)nnz * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL((k_mm_compact<T>), dim3(ew_grid(std::max(n, 1))), dim3(kBlock), 0, b.cur, n, off, pcol,
                           (const T*)pval, cnt, cci, (T*)cv);
        hipError_t e = hipGetLastError();
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&off);
    dev_free(&pcol);
    if(pval)
        (void)cached_free(pval);
    if(s != RAMD_OK)
    {
        dev_free(&cnt);
        dev_free(&cci);
        if(cv)
            (void)cached_free(cv);
        return s;
    }
    mat_free_csr(c);
    mat_free_ell(c);
    mat_free_coo(c);
    mat_free_analysis(c);
    c->format = RAMD_CSR;
    c->nrow   = a->nrow;
    c->ncol   = bm->ncol;
    c->nnz    = nnz;
    c->rp     = cnt;
    c->ci     = cci;
    c->val    = cv;
    return RAMD_OK;
}

// pattern of A^q with sorted rows: SymbolicPower(q) (host_matrix_csr.cpp:3073-3146 -- beyond 8 its loop multiplies once
// more).  The numeric product kernels give exactly that pattern (cancelled zeros stay); the values are not meaningful.
int mat_symbolic_power(const ramd_mat_s* a, int q, ramd_mat_s** out)
{
    ramd_mat_s* A = const_cast<ramd_mat_s*>(a);
    const int   nmul = (q > 8) ? q : q - 1;
    ramd_mat_s* S    = nullptr;
    RAMD_TRY(ramd_mat_clone(A, &S));
    for(int i = 0; i < nmul; ++i)
    {
        ramd_mat_s* nx = nullptr;
        int         s  = ramd_mat_create(a->dtype, &nx);
        if(s == RAMD_OK)
            s = ramd_mat_mat_mult(nx, S, A);
        ramd_mat_destroy(S);
        S = nx;
        if(s != RAMD_OK)
        {
            if(S)
                ramd_mat_destroy(S);
            return s;
        }
    }
    *out = S;
    return RAMD_OK;
}

} // namespace ramd

using namespace ramd;

static int need_csr(const ramd_mat_s* m, const char* what)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(m->format != RAMD_CSR)
    {
        (void)what;
        return RAMD_ERR_UNSUPPORTED; // the front end converts to CSR first (local_matrix.cpp), as for the host backend
    }
    return RAMD_OK;
}

extern "C" {

int ramd_mat_sort(ramd_mat_t m)
{
    RAMD_TRY(need_csr(m, "Sort"));
    if(m->nnz <= 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = ew_grid(std::max(m->nrow, 1));
    if(m->dtype == RAMD_F64)
        hipLaunchKernelGGL((k_sort_rows<double>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci, (double*)m->val);
    else
        hipLaunchKernelGGL((k_sort_rows<float>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci, (float*)m->val);
    RAMD_HIP(hipGetLastError());
    mat_free_analysis(m);
    return RAMD_OK;
}

#ifdef RAMD_WITH_OFFSCOPE // (FSAI / SPAI: out of scope; built with RAMD_EXTRA_CXXFLAGS=-DRAMD_WITH_OFFSCOPE)
static int fsai_impl(ramd_mat_t m, int power, ramd_mat_t pattern);

int ramd_mat_fsai(ramd_mat_t m, int power)
{
    return fsai_impl(m, power, nullptr);
}

int ramd_mat_fsai_pattern(ramd_mat_t m, ramd_mat_t pattern)
{
    RAMD_TRY(need_csr(pattern, "FSAI pattern"));
    if(!m || pattern == m || pattern->nrow != m->nrow || pattern->ncol != m->ncol || pattern->dtype != m->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "FSAI: pattern of the operator's shape and value type, distinct from it");
    return fsai_impl(m, 1, pattern);
}

static int fsai_impl(ramd_mat_t m, int power, ramd_mat_t pattern)
{
    RAMD_TRY(need_csr(m, "FSAI"));
    if(power < 1)
        RAMD_FAIL(RAMD_ERR_ARG, "FSAI: power >= 1");
    if(m->nrow != m->ncol || m->nnz <= 0)
        RAMD_FAIL(RAMD_ERR_ARG, "FSAI: square, non-empty matrix expected");
    Backend&   b = backend();
    ramd_mat_t L = nullptr;
    RAMD_TRY(ramd_mat_create(m->dtype, &L));
    int s = RAMD_OK;
    if(pattern) // host_matrix_csr.cpp:6525-6531: the lower part of the caller's pattern (its values stay where no row system is solved)
        s = ramd_mat_extract_tri(pattern, L, 0, 1);
    else if(power > 1) // host_matrix_csr.cpp:6532-6538: the lower part of the pattern of A^power, values zero
    {
        ramd_mat_s* structure = nullptr;
        s                     = ramd::mat_symbolic_power(m, power, &structure);
        if(s == RAMD_OK)
            s = ramd_mat_extract_tri(structure, L, 0, 1);
        if(structure)
            ramd_mat_destroy(structure);
        if(s == RAMD_OK
           && hipMemsetAsync(L->val, 0, (size_t)L->nnz * val_size(m->dtype), b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    else
        s = ramd_mat_extract_tri(m, L, 0, 1); // ExtractLDiagonal
    long long* soff    = nullptr;
    void*      scratch = nullptr;
    long long  total   = 0;
    const int  n       = m->nrow;
    if(s == RAMD_OK)
        s = dev_alloc(&soff, (int64_t)n + 1);
    if(s == RAMD_OK)
    {
        hipLaunchKernelGGL(k_fsai_sizes, dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, (const int*)L->rp, soff);
        s            = scan_ll(soff, (int64_t)n + 1);
        hipError_t e = (s == RAMD_OK) ? hipSuccess : hipErrorUnknown;
        if(e == hipSuccess)
            e = hipMemcpyAsync(&total, soff + n, sizeof(long long), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    if(s == RAMD_OK && cached_malloc(&scratch, (size_t)total * val_size(m->dtype) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK)
    {
        const int grid = ew_grid(std::max(n, 1));
        if(m->dtype == RAMD_F64)
        {
            hipLaunchKernelGGL((k_fsai<double>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const double*)m->val,
                               (const int*)L->rp, (const int*)L->ci, (double*)L->val, (const long long*)soff,
                               (double*)scratch);
            hipLaunchKernelGGL((k_fsai_scale<double>), dim3(grid), dim3(kBlock), 0, b.cur, n, (const int*)L->rp,
                               (double*)L->val);
        }
        else
        {
            hipLaunchKernelGGL((k_fsai<float>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const float*)m->val,
                               (const int*)L->rp, (const int*)L->ci, (float*)L->val, (const long long*)soff, (float*)scratch);
            hipLaunchKernelGGL((k_fsai_scale<float>), dim3(grid), dim3(kBlock), 0, b.cur, n, (const int*)L->rp,
                               (float*)L->val);
        }
        hipError_t e = hipGetLastError();
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&soff);
    if(scratch)
        (void)cached_free(scratch);
    if(s == RAMD_OK)
    {
        // this becomes the factor (host: Clear(); SetDataPtrCSR(L arrays))
        mat_free_csr(m);
        mat_free_analysis(m);
        m->rp  = L->rp;
        m->ci  = L->ci;
        m->val = L->val;
        m->nnz = L->nnz;
        L->rp = L->ci = nullptr;
        L->val        = nullptr;
        L->nnz        = 0;
    }
    ramd_mat_destroy(L);
    return s;
}

int ramd_mat_spai(ramd_mat_t m)
{
    RAMD_TRY(need_csr(m, "SPAI"));
    if(m->nrow != m->ncol || m->nnz <= 0)
        RAMD_FAIL(RAMD_ERR_ARG, "SPAI: square, non-empty matrix expected");
    Backend&   b  = backend();
    const int  n  = m->nrow;
    ramd_mat_t AT = nullptr;
    RAMD_TRY(ramd_mat_create(m->dtype, &AT));
    int s = mat_transpose(m, AT); // the pattern that is filled; rows sorted
    void* mval = nullptr;
    if(s == RAMD_OK && cached_malloc(&mval, (size_t)AT->nnz * val_size(m->dtype) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && hipMemsetAsync(mval, 0, (size_t)AT->nnz * val_size(m->dtype), b.cur) != hipSuccess)
        s = RAMD_ERR_HIP;
    const int chunk = 1 << 18; // rows per pass: bounds the dense scratch
    for(int r0 = 0; r0 < n && s == RAMD_OK; r0 += chunk)
    {
        const int  r1   = std::min(n, r0 + chunk);
        const int  nr   = r1 - r0;
        long long* soff = nullptr;
        void*      scr  = nullptr;
        long long  tot  = 0;
        s               = dev_alloc(&soff, (int64_t)nr + 1);
        if(s == RAMD_OK)
        {
            hipLaunchKernelGGL(k_spai_sizes, dim3(ew_grid((int64_t)nr + 1)), dim3(kBlock), 0, b.cur, r0, r1,
                               (const int*)AT->rp, (const int*)AT->ci, soff, (int)val_size(m->dtype));
            s = scan_ll(soff, (int64_t)nr + 1);
        }
        if(s == RAMD_OK)
        {
            hipError_t e = hipMemcpyAsync(&tot, soff + nr, sizeof(long long), hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
        }
        if(s == RAMD_OK && cached_malloc(&scr, (size_t)tot * val_size(m->dtype) + kPad) != hipSuccess)
            s = RAMD_ERR_HIP;
        if(s == RAMD_OK)
        {
            const int grid = ew_grid(nr);
            if(m->dtype == RAMD_F64)
                hipLaunchKernelGGL((k_spai_rows<double>), dim3(grid), dim3(kBlock), 0, b.cur, r0, r1, (const int*)AT->rp,
                                   (const int*)AT->ci, (const int*)m->rp, (const int*)m->ci, (const double*)m->val,
                                   (const long long*)soff, (double*)scr, (double*)mval);
            else
                hipLaunchKernelGGL((k_spai_rows<float>), dim3(grid), dim3(kBlock), 0, b.cur, r0, r1, (const int*)AT->rp,
                                   (const int*)AT->ci, (const int*)m->rp, (const int*)m->ci, (const float*)m->val,
                                   (const long long*)soff, (float*)scr, (float*)mval);
            hipError_t e = hipGetLastError();
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
        }
        dev_free(&soff);
        if(scr)
            (void)cached_free(scr);
    }
    if(s == RAMD_OK)
    {
        // M^T = (pattern of A^T, new values); this = its transpose
        if(AT->val)
            (void)cached_free(AT->val);
        AT->val = mval;
        mval    = nullptr;
        ramd_mat_t M = nullptr;
        s            = ramd_mat_create(m->dtype, &M);
        if(s == RAMD_OK)
            s = mat_transpose(AT, M);
        if(s == RAMD_OK)
        {
            mat_free_csr(m);
            mat_free_analysis(m);
            m->rp  = M->rp;
            m->ci  = M->ci;
            m->val = M->val;
            m->nnz = M->nnz;
            M->rp = M->ci = nullptr;
            M->val        = nullptr;
            M->nnz        = 0;
        }
        if(M)
            ramd_mat_destroy(M);
    }
    if(mval)
        (void)cached_free(mval);
    ramd_mat_destroy(AT);
    return s;
}
#endif // RAMD_WITH_OFFSCOPE

int ramd_mat_diag_mult(ramd_mat_t m, ramd_vec_t diag, int left)
{
    RAMD_TRY(need_csr(m, "DiagonalMatrixMult"));
    if(!diag || diag->dtype != m->dtype || diag->n != (left ? m->nrow : m->ncol))
        RAMD_FAIL(RAMD_ERR_ARG, "DiagonalMatrixMult: diagonal vector of the matrix' value type and size expected");
    if(m->nnz <= 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = ew_grid(std::max(m->nrow, 1));
#define GO(T, L)                                                                                                  \
    hipLaunchKernelGGL((k_diag_mult<T, L>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci, (T*)m->val, \
                       (const T*)diag->d)
    if(m->dtype == RAMD_F64)
    {
        if(left)
            GO(double, true);
        else
            GO(double, false);
    }
    else
    {
        if(left)
            GO(float, true);
        else
            GO(float, false);
    }
#undef GO
    RAMD_HIP(hipGetLastError());
    mat_free_analysis(m);
    return RAMD_OK;
}

int ramd_mat_transpose(ramd_mat_t m, ramd_mat_t out)
{
    RAMD_TRY(need_csr(m, "Transpose"));
    if(!out || out == m || out->dtype != m->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "Transpose: a distinct output matrix of the same value type expected");
    if(m->nnz <= 0) // host_matrix_csr.cpp:3765: nothing happens for an empty matrix
        return RAMD_OK;
    return mat_transpose(m, out);
}

int ramd_mat_matrix_add(ramd_mat_t m, ramd_mat_t other, double alpha, double beta, int structure)
{
    RAMD_TRY(need_csr(m, "MatrixAdd"));
    RAMD_TRY(need_csr(other, "MatrixAdd"));
    if(other == m || other->dtype != m->dtype || other->nrow != m->nrow || other->ncol != m->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "MatrixAdd: a second matrix of the same shape and value type expected");
    if(m->dtype == RAMD_F64)
        return matrix_add_t<double>(m, other, alpha, beta, structure != 0);
    return matrix_add_t<float>(m, other, (float)alpha, (float)beta, structure != 0);
}

int ramd_mat_mat_mult(ramd_mat_t c, ramd_mat_t a, ramd_mat_t b)
{
    RAMD_TRY(need_csr(a, "MatrixMult"));
    RAMD_TRY(need_csr(b, "MatrixMult"));
    if(!c || c == a || c == b || a->dtype != b->dtype || c->dtype != a->dtype || a->ncol != b->nrow)
        RAMD_FAIL(RAMD_ERR_ARG, "MatrixMult: C distinct from A and B, A.ncol == B.nrow, one value type");
    if(a->dtype == RAMD_F64)
        return mat_mult_t<double>(c, a, b);
    return mat_mult_t<float>(c, a, b);
}

} // extern "C"
