// lab.hip -- kernel experiments (NOT part of the product ABI; symbols are prefixed ramdx_ and are
// not declared in include/rocalution_amd.h).  Used by tools/spmv_lab.py to A/B kernel variants on the
// GPU box; winners are moved into spmv.hip.
#include "device_utils.hpp"
#include "matrix_impl.hpp"

namespace ramd
{

constexpr int LROWS = 256;

__device__ __forceinline__ int lab_blk(int nblk, int per_xcd)
{
    int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    return b < nblk ? b : -1;
}

// V1: rocSPARSE-like CSR-stream: strided scalar loads (coalesced 8B/4B per lane), products to LDS,
// thread-per-row reduce.  One workgroup per 256 rows, XCD-remapped, non persistent.
// GATHER=false replaces x[col] by a constant (upper bound without the gather)
template <int UNROLL, bool GATHER, bool NT>
__global__ __launch_bounds__(kBlock) void k_lab_scalar(int nrow, int nblk, int per_xcd,
                                                       const int* __restrict__ rp,
                                                       const int* __restrict__ ci,
                                                       const double* __restrict__ val,
                                                       const double* __restrict__ x,
                                                       double* __restrict__ y)
{
    __shared__ double prod[4096];
    const int blk = lab_blk(nblk, per_xcd);
    if(blk < 0)
        return;
    const int r0   = blk * LROWS;
    const int rend = min(r0 + LROWS, nrow);
    const int row  = r0 + threadIdx.x;
    int       rs = 0, re = 0;
    if(row < nrow)
    {
        rs = rp[row];
        re = rp[row + 1];
    }
    const int start = rp[r0];
    const int end   = rp[rend];
    double    sum   = 0.0;
    for(int cb = start; cb < end; cb += 4096)
    {
        const int cnt = min(4096, end - cb);
        for(int i0 = threadIdx.x; i0 < cnt; i0 += kBlock * UNROLL)
        {
            int    c[UNROLL];
            double v[UNROLL];
#pragma unroll
            for(int u = 0; u < UNROLL; ++u)
            {
                const int i = i0 + u * kBlock;
                if(i < cnt)
                {
                    c[u] = NT ? nt_load(ci + cb + i) : ci[cb + i];
                    v[u] = NT ? nt_load(val + cb + i) : val[cb + i];
                }
            }
#pragma unroll
            for(int u = 0; u < UNROLL; ++u)
            {
                const int i = i0 + u * kBlock;
                if(i < cnt)
                    prod[i] = v[u] * (GATHER ? x[c[u]] : (double)(c[u] & 1));
            }
        }
        __syncthreads();
        const int lo = max(rs, cb), hi = min(re, cb + 4096);
        for(int j = lo; j < hi; ++j)
            sum += prod[j - cb];
        __syncthreads();
    }
    if(row < nrow)
        y[row] = sum;
}

// V4: LDS transpose: stage raw col/val (coalesced 16B), then each thread walks ITS row from LDS and
// gathers x with row-consecutive lanes (ELL-like gather coalescing)
__global__ __launch_bounds__(kBlock) void k_lab_transpose(int nrow, int nblk, int per_xcd,
                                                          const int* __restrict__ rp,
                                                          const int* __restrict__ ci,
                                                          const double* __restrict__ val,
                                                          const double* __restrict__ x,
                                                          double* __restrict__ y)
{
    __shared__ double sval[2048];
    __shared__ int    scol[2048];
    const int blk = lab_blk(nblk, per_xcd);
    if(blk < 0)
        return;
    const int r0   = blk * LROWS;
    const int rend = min(r0 + LROWS, nrow);
    const int row  = r0 + threadIdx.x;
    int       rs = 0, re = 0;
    if(row < nrow)
    {
        rs = rp[row];
        re = rp[row + 1];
    }
    const int start = rp[r0];
    const int end   = rp[rend];
    double    sum   = 0.0;
    for(int cb = start & ~3; cb < end; cb += 2048)
    {
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            const int g = (k * kBlock + threadIdx.x) * 4;
            const int j = cb + g;
            if(j < end)
            {
                v4i32 c = nt_load(reinterpret_cast<const v4i32*>(ci + j));
                v2f64 a = nt_load(reinterpret_cast<const v2f64*>(val + j));
                v2f64 b = nt_load(reinterpret_cast<const v2f64*>(val + j) + 1);
                *reinterpret_cast<v4i32*>(scol + g)     = c;
                *reinterpret_cast<v2f64*>(sval + g)     = a;
                *reinterpret_cast<v2f64*>(sval + g + 2) = b;
            }
        }
        __syncthreads();
        const int lo = max(rs, cb), hi = min(re, cb + 2048);
        int       j  = lo;
        for(; j + 4 <= hi; j += 4)
        {
            int    c[4];
            double v[4], xv[4];
#pragma unroll
            for(int e = 0; e < 4; ++e)
            {
                c[e] = scol[j - cb + e];
                v[e] = sval[j - cb + e];
            }
#pragma unroll
            for(int e = 0; e < 4; ++e)
                xv[e] = x[c[e]];
#pragma unroll
            for(int e = 0; e < 4; ++e)
                sum += v[e] * xv[e];
        }
        for(; j < hi; ++j)
            sum += sval[j - cb] * x[scol[j - cb]];
        __syncthreads();
    }
    if(row < nrow)
        y[row] = sum;
}

// V5: vector-packet stream (production layout) but non-persistent, selectable nt
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_lab_packet(int nrow, int nblk, int per_xcd,
                                                       const int* __restrict__ rp,
                                                       const int* __restrict__ ci,
                                                       const double* __restrict__ val,
                                                       const double* __restrict__ x,
                                                       double* __restrict__ y)
{
    __shared__ double prod[2048];
    const int blk = lab_blk(nblk, per_xcd);
    if(blk < 0)
        return;
    const int r0   = blk * LROWS;
    const int rend = min(r0 + LROWS, nrow);
    const int row  = r0 + threadIdx.x;
    int       rs = 0, re = 0;
    if(row < nrow)
    {
        rs = rp[row];
        re = rp[row + 1];
    }
    const int start = rp[r0];
    const int end   = rp[rend];
    double    sum   = 0.0;
    for(int cb = start & ~3; cb < end; cb += 2048)
    {
        v4i32 c[2];
        v2f64 a[2], b[2];
        bool  ok[2];
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            const int j = cb + (k * kBlock + threadIdx.x) * 4;
            ok[k]       = j < end;
            if(ok[k])
            {
                if(NT)
                {
                    c[k] = nt_load(reinterpret_cast<const v4i32*>(ci + j));
                    a[k] = nt_load(reinterpret_cast<const v2f64*>(val + j));
                    b[k] = nt_load(reinterpret_cast<const v2f64*>(val + j) + 1);
                }
                else
                {
                    c[k] = *reinterpret_cast<const v4i32*>(ci + j);
                    a[k] = *reinterpret_cast<const v2f64*>(val + j);
                    b[k] = *(reinterpret_cast<const v2f64*>(val + j) + 1);
                }
            }
        }
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            const int g = (k * kBlock + threadIdx.x) * 4;
            const int j = cb + g;
            if(ok[k])
            {
                const int    cc[4] = {c[k].x, c[k].y, c[k].z, c[k].w};
                const double vv[4] = {a[k].x, a[k].y, b[k].x, b[k].y};
                double       xv[4];
#pragma unroll
                for(int e = 0; e < 4; ++e)
                    xv[e] = (j + e >= start && j + e < end) ? x[cc[e]] : 0.0;
#pragma unroll
                for(int e = 0; e < 4; ++e)
                    prod[g + e] = (j + e >= start && j + e < end) ? vv[e] * xv[e] : 0.0;
            }
        }
        __syncthreads();
        const int lo = max(rs, cb), hi = min(re, cb + 2048);
        for(int j = lo; j < hi; ++j)
            sum += prod[j - cb];
        __syncthreads();
    }
    if(row < nrow)
        y[row] = sum;
}


// V9/V10: WAVE-granular stream: every wave owns 64 consecutive rows, keeps its products in a private
// LDS slice and never meets a workgroup barrier.  PIPE: persistent over row groups with the next
// group's val/col prefetched into registers while the current one is reduced.
constexpr int WNNZ = 512; // products per wave pass

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool PIPE>
__global__ __launch_bounds__(kBlock) void k_lab_wave(int nrow, int ngrp, int per_xcd, int wg_per_xcd,
                                                     const int* __restrict__ rp,
                                                     const int* __restrict__ ci,
                                                     const double* __restrict__ val,
                                                     const double* __restrict__ x,
                                                     double* __restrict__ y)
{
    __shared__ double prod_all[4 * WNNZ];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double*   prod = prod_all + wv * WNNZ;
    const int xcd  = blockIdx.x & 7;
    // groups of 64 rows; an XCD owns a contiguous range of groups, its waves walk it cyclically
    const int wave_in_xcd = (blockIdx.x >> 3) * 4 + wv;
    const int waves_xcd   = wg_per_xcd * 4;
    int       g           = wave_in_xcd;
    constexpr int U = 8;
    int    c[U];
    double v[U];
    int    rs = 0, re = 0, start = 0, end = 0;
    auto   load_meta = [&](int gg) {
        const int grp = xcd * per_xcd + gg;
        const int row = grp * 64 + lane;
        rs = re = 0;
        if(gg < per_xcd && grp < ngrp && row < nrow)
        {
            rs = rp[row];
            re = rp[row + 1];
        }
        start = __shfl(rs, 0, 64);
        int last = min(63, nrow - 1 - grp * 64);
        end = (gg < per_xcd && grp < ngrp) ? __shfl(re, last < 0 ? 0 : last, 64) : start;
    };
    auto issue = [&](int s, int e) {
#pragma unroll
        for(int u = 0; u < U; ++u)
        {
            const int i = s + lane + u * 64;
            if(i < e)
            {
                c[u] = nt_load(ci + i);
                v[u] = nt_load(val + i);
            }
        }
    };
    load_meta(g);
    issue(start, min(end, start + WNNZ));
    while(g < per_xcd && xcd * per_xcd + g < ngrp)
    {
        const int grp = xcd * per_xcd + g;
        const int row = grp * 64 + lane;
        const int crs = rs, cre = re, cstart = start, cend = end;
        double    sum = 0.0;
        for(int cb = cstart; cb < cend; cb += WNNZ)
        {
            if(cb != cstart)
                issue(cb, min(cend, cb + WNNZ));
            // products of the current pass -> LDS
#pragma unroll
            for(int u = 0; u < U; ++u)
            {
                const int i = cb + lane + u * 64;
                if(i < min(cend, cb + WNNZ))
                    prod[i - cb] = v[u] * x[c[u]];
            }
            if(PIPE && cb + WNNZ >= cend)
            {
                // prefetch the NEXT group's first pass while this one is reduced
                load_meta(g + waves_xcd);
                issue(start, min(end, start + WNNZ));
            }
            wave_lds_sync();
            const int lo = max(crs, cb), hi = min(cre, cb + WNNZ);
            for(int j = lo; j < hi; ++j)
                sum += prod[j - cb];
            wave_lds_sync();
        }
        if(row < nrow)
            nt_store(sum, y + row);
        g += waves_xcd;
        if(!PIPE)
        {
            load_meta(g);
            issue(start, min(end, start + WNNZ));
        }
    }
}


// elimination study on the packet kernel (v6): which part costs the time?
template <bool NOSTORE, bool NOGATHER, bool NOLDS, bool NORP>
__global__ __launch_bounds__(kBlock) void k_lab_elim(int nrow, int nblk, int per_xcd,
                                                     const int* __restrict__ rp,
                                                     const int* __restrict__ ci,
                                                     const double* __restrict__ val,
                                                     const double* __restrict__ x,
                                                     double* __restrict__ y)
{
    __shared__ double prod[2048];
    const int blk = lab_blk(nblk, per_xcd);
    if(blk < 0)
        return;
    const int r0   = blk * LROWS;
    const int rend = min(r0 + LROWS, nrow);
    const int row  = r0 + threadIdx.x;
    int       rs = 0, re = 0, start, end;
    if(NORP)
    {
        // synthetic uniform rows (7 per row, minus boundary effects ignored): no dependent load
        rs    = row * 7 - 6 * 512;
        rs    = rs < 0 ? 0 : rs;
        re    = rs + 7;
        start = r0 * 7 - 6 * 512;
        start = start < 0 ? 0 : start;
        end   = start + 7 * (rend - r0);
        const int tot = 7 * nrow - 6 * 512 * 512 - 8; // stay inside the arrays
        end   = end > tot ? tot : end;
        start = start > end ? end : start;
    }
    else
    {
        if(row < nrow)
        {
            rs = rp[row];
            re = rp[row + 1];
        }
        start = rp[r0];
        end   = rp[rend];
    }
    double sum = 0.0;
    for(int cb = start & ~3; cb < end; cb += 2048)
    {
        v4i32 c[2];
        v2f64 a[2], b[2];
        bool  ok[2];
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            const int j = cb + (k * kBlock + threadIdx.x) * 4;
            ok[k]       = j < end;
            if(ok[k])
            {
                c[k] = *reinterpret_cast<const v4i32*>(ci + j);
                a[k] = *reinterpret_cast<const v2f64*>(val + j);
                b[k] = *(reinterpret_cast<const v2f64*>(val + j) + 1);
            }
        }
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            const int g = (k * kBlock + threadIdx.x) * 4;
            const int j = cb + g;
            if(ok[k])
            {
                const int    cc[4] = {c[k].x, c[k].y, c[k].z, c[k].w};
                const double vv[4] = {a[k].x, a[k].y, b[k].x, b[k].y};
                double       xv[4];
#pragma unroll
                for(int e = 0; e < 4; ++e)
                    xv[e] = NOGATHER ? (double)(cc[e] & 3)
                                     : ((j + e >= start && j + e < end)
                                            ? x[NORP ? min(max(cc[e], 0), nrow - 1) : cc[e]]
                                            : 0.0);
#pragma unroll
                for(int e = 0; e < 4; ++e)
                {
                    if(NOLDS)
                        sum += vv[e] * xv[e];
                    else
                        prod[g + e] = (j + e >= start && j + e < end) ? vv[e] * xv[e] : 0.0;
                }
            }
        }
        if(!NOLDS)
        {
            __syncthreads();
            const int lo = max(rs, cb), hi = min(re, cb + 2048);
            for(int j = lo; j < hi; ++j)
                sum += prod[j - cb];
            __syncthreads();
        }
    }
    if(row < nrow)
    {
        if(!NOSTORE || sum == 1.2345e-300)
            y[row] = sum;
    }
}


// persistent packet kernel, explicit software pipeline: while block i is reduced / stored, the val/col
// packets of block i+1 are already in flight (its row pointers were fetched during block i's gathers)
template <bool PREFETCH, bool NT>
__global__ __launch_bounds__(kBlock) void k_lab_persist(int nrow, int nblk, int per_xcd, int wg_per_xcd,
                                                        const int* __restrict__ rp,
                                                        const int* __restrict__ ci,
                                                        const double* __restrict__ val,
                                                        const double* __restrict__ x,
                                                        double* __restrict__ y)
{
    __shared__ double prod[2048];
    const int xcd = blockIdx.x & 7;
    int       lb  = blockIdx.x >> 3;
    // ---- metadata + packets of the first block
    int   rs = 0, re = 0, start = 0, end = 0;
    v4i32 c[2];
    v2f64 a[2], b[2];
    auto  meta = [&](int l, int& mrs, int& mre, int& ms, int& me) {
        const int blk = xcd * per_xcd + l;
        mrs = mre = ms = me = 0;
        if(l < per_xcd && blk < nblk)
        {
            const int r0   = blk * LROWS;
            const int rend = min(r0 + LROWS, nrow);
            const int row  = r0 + threadIdx.x;
            if(row < nrow)
            {
                mrs = rp[row];
                mre = rp[row + 1];
            }
            ms = rp[r0];
            me = rp[rend];
        }
    };
    auto issue = [&](int cb, int e) {
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            const int j = cb + (k * kBlock + threadIdx.x) * 4;
            if(j < e)
            {
                if(NT)
                {
                    c[k] = nt_load(reinterpret_cast<const v4i32*>(ci + j));
                    a[k] = nt_load(reinterpret_cast<const v2f64*>(val + j));
                    b[k] = nt_load(reinterpret_cast<const v2f64*>(val + j) + 1);
                }
                else
                {
                    c[k] = *reinterpret_cast<const v4i32*>(ci + j);
                    a[k] = *reinterpret_cast<const v2f64*>(val + j);
                    b[k] = *(reinterpret_cast<const v2f64*>(val + j) + 1);
                }
            }
        }
    };
    meta(lb, rs, re, start, end);
    issue(start & ~3, end);
    while(lb < per_xcd && xcd * per_xcd + lb < nblk)
    {
        const int blk = xcd * per_xcd + lb;
        const int row = blk * LROWS + threadIdx.x;
        int       nrs = 0, nre = 0, nstart = 0, nend = 0;
        if(PREFETCH)
            meta(lb + wg_per_xcd, nrs, nre, nstart, nend); // lands while we gather
        double sum = 0.0;
        for(int cb = start & ~3; cb < end; cb += 2048)
        {
            if(cb != (start & ~3))
                issue(cb, end);
#pragma unroll
            for(int k = 0; k < 2; ++k)
            {
                const int g = (k * kBlock + threadIdx.x) * 4;
                const int j = cb + g;
                if(j < end)
                {
                    const int    cc[4] = {c[k].x, c[k].y, c[k].z, c[k].w};
                    const double vv[4] = {a[k].x, a[k].y, b[k].x, b[k].y};
                    double       xv[4];
#pragma unroll
                    for(int e = 0; e < 4; ++e)
                        xv[e] = (j + e >= start && j + e < end) ? x[cc[e]] : 0.0;
#pragma unroll
                    for(int e = 0; e < 4; ++e)
                        prod[g + e] = (j + e >= start && j + e < end) ? vv[e] * xv[e] : 0.0;
                }
            }
            if(PREFETCH && cb + 2048 >= end)
                issue(nstart & ~3, nend); // next block's packets fly during reduce + store
            __syncthreads();
            const int lo = max(rs, cb), hi = min(re, cb + 2048);
            for(int j = lo; j < hi; ++j)
                sum += prod[j - cb];
            __syncthreads();
        }
        if(row < nrow)
            y[row] = sum;
        lb += wg_per_xcd;
        if(PREFETCH)
        {
            rs    = nrs;
            re    = nre;
            start = nstart;
            end   = nend;
        }
        else
        {
            meta(lb, rs, re, start, end);
            issue(start & ~3, end);
        }
    }
}


// V40: LDS transpose with INDEPENDENT fully-coalesced packet layouts for col (int4) and val (double2),
// thread-per-row gathers (row-consecutive lanes => 4 lines per gather instruction).
// Minimises L1 line accesses: 8 (col) + 16 (val) + ~16 (gather) per 256 nnz.
template <bool NTSTORE, bool NTLOAD>
__global__ __launch_bounds__(kBlock) void k_lab_tr2(int nrow, int nblk, int per_xcd,
                                                    const int* __restrict__ rp,
                                                    const int* __restrict__ ci,
                                                    const double* __restrict__ val,
                                                    const double* __restrict__ x,
                                                    double* __restrict__ y)
{
    __shared__ double sval[2048];
    __shared__ int    scol[2048];
    const int blk = lab_blk(nblk, per_xcd);
    if(blk < 0)
        return;
    const int r0   = blk * LROWS;
    const int rend = min(r0 + LROWS, nrow);
    const int row  = r0 + threadIdx.x;
    int       rs = 0, re = 0;
    if(row < nrow)
    {
        rs = rp[row];
        re = rp[row + 1];
    }
    const int start = rp[r0];
    const int end   = rp[rend];
    double    sum   = 0.0;
    for(int cb = start & ~3; cb < end; cb += 2048)
    {
        v4i32 c[2];
        v2f64 a[4];
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            const int j = cb + (k * kBlock + threadIdx.x) * 4;
            if(j < end)
                c[k] = NTLOAD ? nt_load(reinterpret_cast<const v4i32*>(ci + j))
                              : *reinterpret_cast<const v4i32*>(ci + j);
        }
#pragma unroll
        for(int k = 0; k < 4; ++k)
        {
            const int j = cb + (k * kBlock + threadIdx.x) * 2;
            if(j < end)
                a[k] = NTLOAD ? nt_load(reinterpret_cast<const v2f64*>(val + j))
                              : *reinterpret_cast<const v2f64*>(val + j);
        }
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            const int g = (k * kBlock + threadIdx.x) * 4;
            if(cb + g < end)
                *reinterpret_cast<v4i32*>(scol + g) = c[k];
        }
#pragma unroll
        for(int k = 0; k < 4; ++k)
        {
            const int g = (k * kBlock + threadIdx.x) * 2;
            if(cb + g < end)
                *reinterpret_cast<v2f64*>(sval + g) = a[k];
        }
        __syncthreads();
        const int lo = max(rs, cb), hi = min(re, cb + 2048);
        int       j  = lo;
        for(; j + 4 <= hi; j += 4)
        {
            int    cc[4];
            double v[4], xv[4];
#pragma unroll
            for(int e = 0; e < 4; ++e)
            {
                cc[e] = scol[j - cb + e];
                v[e]  = sval[j - cb + e];
            }
#pragma unroll
            for(int e = 0; e < 4; ++e)
                xv[e] = x[cc[e]];
#pragma unroll
            for(int e = 0; e < 4; ++e)
                sum += v[e] * xv[e];
        }
        for(; j < hi; ++j)
            sum += sval[j - cb] * x[scol[j - cb]];
        __syncthreads();
    }
    if(row < nrow)
    {
        if(NTSTORE)
            nt_store(sum, y + row);
        else
            y[row] = sum;
    }
}

// plain device copy: achievable HBM bandwidth reference (read n doubles, write n doubles)
__global__ __launch_bounds__(kBlock) void k_lab_copy(int64_t n2, const v2f64* __restrict__ a,
                                                     v2f64* __restrict__ b)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gsz)
        b[i] = a[i];
}
__global__ __launch_bounds__(kBlock) void k_lab_read(int64_t n2, const v2f64* __restrict__ a,
                                                     double* __restrict__ out)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    double        s   = 0.0;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gsz)
    {
        v2f64 v = nt_load(a + i);
        s += v.x + v.y;
    }
    if(s == 1.2345e-300)
        out[0] = s;
}

} // namespace ramd

using namespace ramd;

extern "C" int ramdx_lab_csr(ramd_mat_t m, ramd_vec_t x, ramd_vec_t y, int variant, int reps, double* ms)
{
    Backend&  b       = backend();
    const int nblk    = (m->nrow + LROWS - 1) / LROWS;
    const int per_xcd = (nblk + 7) / 8;
    const int grid    = per_xcd * 8;
    hipEvent_t e0, e1;
    RAMD_HIP(hipEventCreate(&e0));
    RAMD_HIP(hipEventCreate(&e1));
    const double* xv = (const double*)x->d;
    double*       yv = (double*)y->d;
    const double* va = (const double*)m->val;
    for(int r = -3; r < reps; ++r)
    {
        if(r == 0)
            RAMD_HIP(hipEventRecord(e0, b.cur));
#define L(K) hipLaunchKernelGGL(K, dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, per_xcd, m->rp, m->ci, va, xv, yv)
        switch(variant)
        {
        case 0:
            mat_apply_impl<double>(m, xv, yv, 0, 1.0);
            break;
        case 50:
            mat_apply_dot_impl<double>(m, xv, yv, 5);
            break;
        case 1:
            L((k_lab_scalar<4, true, true>));
            break;
        case 2:
            L((k_lab_scalar<8, true, true>));
            break;
        case 3:
            L((k_lab_scalar<8, true, false>));
            break;
        case 4:
            L(k_lab_transpose);
            break;
        case 5:
            L((k_lab_packet<true>));
            break;
        case 6:
            L((k_lab_packet<false>));
            break;
        case 7:
            L((k_lab_scalar<8, false, true>)); // no gather: streaming upper bound of this structure
            break;
        case 8:
            L((k_lab_scalar<16, true, true>));
            break;
        case 40:
            L((k_lab_tr2<false, false>));
            break;
        case 41:
            L((k_lab_tr2<true, false>));
            break;
        case 42:
            L((k_lab_tr2<false, true>));
            break;
        case 43:
            L((k_lab_tr2<true, true>));
            break;
        case 30:
        case 31:
        case 32:
        case 33:
        case 34:
        {
            // 30: persistent no prefetch   31: prefetch   32: prefetch+nt   33: prefetch, 128 WG/XCD
            // 34: prefetch, 192 WG/XCD
            int wgx = (variant == 33) ? 128 : (variant == 34 ? 192 : 256);
            if(wgx > per_xcd)
                wgx = per_xcd;
            if(variant == 30)
                hipLaunchKernelGGL((k_lab_persist<false, false>), dim3(wgx * 8), dim3(kBlock), 0, b.cur,
                                   m->nrow, nblk, per_xcd, wgx, m->rp, m->ci, va, xv, yv);
            else if(variant == 32)
                hipLaunchKernelGGL((k_lab_persist<true, true>), dim3(wgx * 8), dim3(kBlock), 0, b.cur,
                                   m->nrow, nblk, per_xcd, wgx, m->rp, m->ci, va, xv, yv);
            else
                hipLaunchKernelGGL((k_lab_persist<true, false>), dim3(wgx * 8), dim3(kBlock), 0, b.cur,
                                   m->nrow, nblk, per_xcd, wgx, m->rp, m->ci, va, xv, yv);
            break;
        }
        case 20:
            L((k_lab_elim<false, false, false, false>));
            break;
        case 21:
            L((k_lab_elim<true, false, false, false>)); // no y store
            break;
        case 22:
            L((k_lab_elim<false, true, false, false>)); // no gather
            break;
        case 23:
            L((k_lab_elim<false, false, true, false>)); // no LDS / barriers
            break;
        case 24:
            L((k_lab_elim<false, false, false, true>)); // no row-pointer dependency
            break;
        case 25:
            L((k_lab_elim<true, true, true, true>)); // pure val/col stream
            break;
        case 26:
            L((k_lab_elim<false, false, true, true>)); // no LDS, no rp
            break;
        case 9:
        case 10:
        case 11:
        case 12:
        {
            const int ngrp = (m->nrow + 63) / 64;
            const int pxg  = (ngrp + 7) / 8;
            // 9: one pass per wave (non persistent)   10: persistent, no prefetch
            // 11: persistent + prefetch, 256 WG/XCD   12: persistent + prefetch, 128 WG/XCD
            int wgx = (variant == 9) ? (pxg + 3) / 4 : (variant == 12 ? 128 : 256);
            if(wgx * 4 > pxg)
                wgx = (pxg + 3) / 4;
            if(variant == 11 || variant == 12)
                hipLaunchKernelGGL((k_lab_wave<true>), dim3(wgx * 8), dim3(kBlock), 0, b.cur, m->nrow, ngrp,
                                   pxg, wgx, m->rp, m->ci, va, xv, yv);
            else
                hipLaunchKernelGGL((k_lab_wave<false>), dim3(wgx * 8), dim3(kBlock), 0, b.cur, m->nrow, ngrp,
                                   pxg, wgx, m->rp, m->ci, va, xv, yv);
            break;
        }
        default:
            RAMD_FAIL(RAMD_ERR_ARG, "unknown lab variant");
        }
#undef L
    }
    RAMD_HIP(hipEventRecord(e1, b.cur));
    RAMD_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    RAMD_HIP(hipEventElapsedTime(&t, e0, e1));
    *ms = t / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return RAMD_OK;
}

extern "C" int ramdx_lab_copy(ramd_vec_t a, ramd_vec_t bvec, int mode, int reps, double* ms)
{
    Backend&   b = backend();
    hipEvent_t e0, e1;
    RAMD_HIP(hipEventCreate(&e0));
    RAMD_HIP(hipEventCreate(&e1));
    const int64_t n2   = a->n / 2;
    const int     grid = b.num_cu * 16;
    for(int r = -3; r < reps; ++r)
    {
        if(r == 0)
            RAMD_HIP(hipEventRecord(e0, b.cur));
        if(mode == 0)
            hipLaunchKernelGGL(k_lab_copy, dim3(grid), dim3(kBlock), 0, b.cur, n2, (const v2f64*)a->d,
                               (v2f64*)bvec->d);
        else
            hipLaunchKernelGGL(k_lab_read, dim3(grid), dim3(kBlock), 0, b.cur, n2, (const v2f64*)a->d,
                               (double*)bvec->d);
    }
    RAMD_HIP(hipEventRecord(e1, b.cur));
    RAMD_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    RAMD_HIP(hipEventElapsedTime(&t, e0, e1));
    *ms = t / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return RAMD_OK;
}
