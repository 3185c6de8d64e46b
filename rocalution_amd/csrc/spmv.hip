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
// CSR kernel: see k_csr_tr below (row-blocked, LDS-staged, XCD-aware).  Bandwidth-bound:
// 12 B/nnz + 20 B/row of algorithmic traffic (clients/samples/benchmark.cpp:213-233 accounting).
#include "device_utils.hpp"
#include "matrix_impl.hpp"

#include <algorithm>
#include <climits>
#include <cstring>
#include <vector>

namespace ramd
{

constexpr int kCsrRows  = 256; // rows per workgroup (one per thread)
#ifndef RAMD_GATHER_W
#define RAMD_GATHER_W 8
#endif
constexpr int kGatherW = RAMD_GATHER_W; // x gathers in flight per row and batch
constexpr int kCsrChunk = 2048; // entries staged in LDS per pass (16 KiB values + 8 KiB columns)
constexpr int kXlElems  = 2048; // k_csr_xl: LDS elements of the x area (16 KiB fp64)

// XCD- and band-aware order of the row blocks: BandMap / xcd_block in device_utils.hpp
template <typename T>
struct ValPk; // 16-byte packet of values
template <>
struct ValPk<double>
{
    using type             = v2f64;
    static constexpr int N = 2;
};
template <>
struct ValPk<float>
{
    using type             = v4f32;
    static constexpr int N = 4;
};

// workspace of the fused <x, y> epilogue
struct CsrDotWs
{
    double*     part1; // [4 * nblk] one partial per wave
    const void* dotv; // the dot runs against this vector (nullptr: against x)
    const void* jdinv; // MODE 2 (Jacobi sweep): inverse diagonal and right-hand side
    const void* jrhs;
};

// CSR SpMV, "LDS transpose" layout.  Measured on MI355X (round 1, a since-deleted lab of kernel variants): the kernel is bound by
// HBM *and* by L1/TA line throughput, so every global access is made as line-efficient as possible:
//   1. the workgroup's contiguous nnz range is streamed RAW into LDS with independent, fully
//      coalesced 16-byte packets (int4 columns, double2/float4 values; non-temporal: read once);
//   2. thread t then walks ITS row in LDS and gathers x[col] -- lanes = consecutive rows, so for
//      banded matrices a gather instruction touches ~4 cache lines instead of ~10, and same-row
//      neighbours hit the same lines;
//   3. the row sum runs left to right in storage order (bit-identical to the host backend);
//      y is written once, non-temporal.
// MODE 0: y = A x      MODE 1: y += scalar * A x (term by term into y, as the host ApplyAdd)
// MODE 2: one damped-Jacobi sweep  y = x + scalar * (dinv * (-(A x) + rhs))  -- the four vector kernels of the
//         FixedPoint(omega)+Jacobi smoother (Apply, ScaleAdd(-1, rhs), PointWiseMult, AddScale) as the epilogue, same operations
// DOT   : additionally reduce <x, y> into scalar slot `slot` (square matrix)
// LDS (24 KiB per workgroup) allows 6 workgroups = 6 waves per SIMD: keep the register budget inside
// 512/6 VGPRs (the fused-dot variant sat at 86 and lost a whole wave per SIMD: -4%)
// row patterns: see ramd_mat_s::pat_* and csr_analyse_pattern below
#ifndef RAMD_CSR_PAT_WAVES
#define RAMD_CSR_PAT_WAVES 6
#endif
// GRP (row groups, csr_analyse_groups): the rows of one mesh node of an FE matrix carry the SAME column list (5 unknowns per
// node on a shell mesh: five rows of ~35 entries with identical columns).  Only the first row of such a group -- its leader --
// has its column packets read; the others walk the leader's columns in LDS (lead[row] = offset from the row's entries to
// the leader's).  4 bytes per entry become ~0.8 on the af_shell10-class matrix: 12.6 -> 9.4 bytes per entry moved.  A
// follower whose leader's entries were staged in an earlier pass reads its own columns from memory (one group in ~12).
// PIPE (RAMD_CSR_PIPE=1, an experiment for rows of 25-45 entries where a row block takes several LDS passes): the packets of
// pass k + 1 are requested before the rows of pass k are walked and wait in registers meanwhile.  Measured slower (0.173
// against 0.154-0.159 ms on the config-3 surrogate): the waves of the other five workgroups of the CU already cover that
// latency, and the held packets cost registers.
// CHK: entries per LDS pass.  kCsrChunk (2048) holds the 256 rows of a block when they are short; the rows of a 27-point stencil
// (6 912 entries per block) took 3.4 passes of it, a quarter of the rows walking while the others stand at the barrier -- the
// long-row instantiations stage 4096 or (row patterns: no columns in LDS) 8192 entries per pass.
template <typename T, int MODE, bool DOT, bool PAT, bool GRP = false, bool PIPE = false, int CHK = kCsrChunk>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(CHK > kCsrChunk ? 2 : (PAT ? RAMD_CSR_PAT_WAVES : 6), 8))) void k_csr_tr(int nrow, int nblk, int per_xcd,
                                                   const int* __restrict__ rp,
                                                   const int* __restrict__ ci,
                                                   const T* __restrict__ val,
                                                   const T* __restrict__ x, T* __restrict__ y, T scalar,
                                                   CsrDotWs ws, int slot, BandMap bm, CsrPattern pat, CsrGroups grp = {},
                                                   const int* __restrict__ blk_rp = nullptr)
{
    using VP          = typename ValPk<T>::type;
    constexpr int VN  = ValPk<T>::N;
    __shared__ T      sval[CHK];
    __shared__ int    scol[PAT ? kPatMax * kPatMaxW : CHK]; // PAT: the dictionary of column offsets instead
    const int blk  = xcd_block(nblk, per_xcd, bm);
    double    dacc = 0.0;
    if(blk >= 0)
    {
        const int r0   = blk * kCsrRows;
        const int rend = min(r0 + kCsrRows, nrow);
        const int row  = r0 + threadIdx.x;
        int       rs = 0, re = 0;
        int       dbase = 0; // PAT: where this row's offsets start in the dictionary, minus rs
        int       lead  = 0; // GRP: leader's entry index minus this row's
        if(PAT)
            for(int i = threadIdx.x; i < pat.n * pat.w; i += kBlock)
                scol[i] = pat.dict[i];
        if(row < nrow)
        {
            rs = rp[row];
            re = rp[row + 1];
            if(PAT)
                dbase = (int)pat.id[row] * pat.w - rs;
            if(GRP)
                lead = grp.lead[row];
        }
        // (blk_rp: a compact, cache-resident copy of the block offsets -- the addresses of the first packets hang on them,
        //  and rp[r0] itself is a miss of a stream as long as the row count; see k_csr_pat2)
        const int start = blk_rp ? blk_rp[blk] : rp[r0];
        const int end   = blk_rp ? blk_rp[blk + 1] : rp[rend];
        T         sum   = (T)0;
        // x[row] for the epilogues (fused dot against x, Jacobi sweep): the row's own diagonal entry gathers it anyway, so
        // the row-pattern product takes it from there (one gather per row less).  What the fused dot costs on top of the
        // plain product differs from box to box by more than this saves (tools/spmv_time.py at 512^3: +0.20 ms before,
        // +0.004 and +0.12 ms after, gpurun_out/r03aq / r03ar / r03as).  Only for row patterns: with the stored columns
        // read the same test costs registers that kernel does not have (2.5 -> 2.97 ms).
        T    xrow      = (T)0;
        bool have_xrow = false;
        if(MODE == 1 && row < nrow)
            sum = y[row];
        v4i32 c[CHK / (4 * kBlock)];
        VP    a[CHK / (VN * kBlock)];
        auto  request = [&](int cb) { // the packets of the pass that starts at entry cb
#pragma unroll
            for(int k = 0; k < (PAT ? 0 : CHK / (4 * kBlock)); ++k)
            {
                const int j = cb + (k * kBlock + threadIdx.x) * 4;
                if(j < end && (!GRP || grp.need[j >> 2]))
                    c[k] = nt_load(reinterpret_cast<const v4i32*>(ci + j));
            }
#pragma unroll
            for(int k = 0; k < CHK / (VN * kBlock); ++k)
            {
                const int j = cb + (k * kBlock + threadIdx.x) * VN;
                if(j < end)
                    a[k] = nt_load(reinterpret_cast<const VP*>(val + j));
            }
        };
        if(PIPE)
            request(start & ~3);
        for(int cb = start & ~3; cb < end; cb += CHK)
        {
            if(!PIPE)
                request(cb);
#pragma unroll
            for(int k = 0; k < (PAT ? 0 : CHK / (4 * kBlock)); ++k)
            {
                const int g = (k * kBlock + threadIdx.x) * 4;
                if(cb + g < end)
                    *reinterpret_cast<v4i32*>(scol + g) = c[k];
            }
#pragma unroll
            for(int k = 0; k < CHK / (VN * kBlock); ++k)
            {
                const int g = (k * kBlock + threadIdx.x) * VN;
                if(cb + g < end)
                    *reinterpret_cast<VP*>(sval + g) = a[k];
            }
            __syncthreads();
            if(PIPE && cb + CHK < end)
                request(cb + CHK);
            const int lo = max(rs, cb), hi = min(re, cb + CHK);
            // masked batches of kGatherW entries: all gathers of a batch are issued before the first use,
            // the products are added IN ORDER.  (A row of 7 used to cost one batch of 4 plus three
            // one-by-one gathers = 4 dependent L2 round trips; now it is one batch.)
            for(int j = lo; j < hi; j += kGatherW)
            {
                int cc[kGatherW];
                T   v[kGatherW], xv[kGatherW];
#pragma unroll
                for(int e = 0; e < kGatherW; ++e)
                    if(j + e < hi)
                    {
                        if(GRP)
                        {
                            const int jl = j + e + lead; // the same entry of the group's leader
                            cc[e]        = jl >= cb ? scol[jl - cb] : ci[j + e];
                        }
                        else
                            cc[e] = PAT ? row + scol[dbase + j + e] : scol[j - cb + e];
                        v[e] = sval[j - cb + e];
                    }
#pragma unroll
                for(int e = 0; e < kGatherW; ++e)
                    if(j + e < hi)
                        xv[e] = x[cc[e]];
#pragma unroll
                for(int e = 0; e < kGatherW; ++e)
                    if(j + e < hi)
                    {
                        if(MODE != 1)
                            sum += v[e] * xv[e];
                        else
                            sum += scalar * v[e] * xv[e];
                        if(PAT && (DOT || MODE == 2) && cc[e] == row)
                        {
                            xrow      = xv[e];
                            have_xrow = true;
                        }
                    }
            }
            __syncthreads();
        }
        if(row < nrow)
        {
            if((DOT && !ws.dotv) || MODE == 2)
                if(!have_xrow) // (a row without a stored diagonal entry)
                    xrow = x[row];
            if(MODE == 2)
            {
                T t = (T)(-1) * sum + static_cast<const T*>(ws.jrhs)[row];
                t   = static_cast<const T*>(ws.jdinv)[row] * t;
                sum = xrow + scalar * t;
            }
            // non-temporal, unconditionally: a run-time switch here let the compiler merge both branches
            // into ONE plain store (the hint was lost and the kernel ran 15% slower at 256^3)
            nt_store(sum, y + row);
            if(DOT)
                dacc = (double)sum * (double)(ws.dotv ? static_cast<const T*>(ws.dotv)[row] : xrow);
        }
    }
    if(DOT)
    {
        // one partial per WAVE, fire-and-forget (no workgroup barrier in the epilogue); a tiny second
        // launch (reduce_sum_to_slot) adds the partials in a fixed order.  (A ticketed in-kernel finish
        // was measured 9% slower: every workgroup would have to drain its partial store before taking
        // the ticket, which holds its wave slots for a full memory round trip.)
        const double wsum = wave_reduce_sum(dacc);
        if((threadIdx.x & 63) == 0 && blk >= 0)
            ws.part1[blk * (kBlock / 64) + (threadIdx.x >> 6)] = wsum;
    }
}

// k_csr_wr (round 6): the row walk of k_csr_tr from WAVE-private LDS images, for rows of 16+ entries (the 27-point operator:
// 6 912 entries per 256-row block).  k_csr_tr stages a workgroup's entries in passes of 4096 behind workgroup barriers: 1.7
// passes per block, rows cut by the pass boundary, the waves of a workgroup waiting for each other 72 % of their cycles
// (profiles/r06_pmc_csr_tr_lap27.txt).  Here every wave stages the entries of ITS 64 rows (one pass where they fit kWrCap:
// 64 x 28), all packets requested at once, and walks them lane = row -- the x gathers of a step stay the 64 consecutive
// elements they are in k_csr_tr (the element-order forms k_csr_w4 / k_csr_wp pay 21+ cache lines per gather instruction on
// this operator) -- with no barrier after the dictionary's.  Same products in the same order per row, the same per-wave
// partials of the fused dot: results identical to k_csr_tr bit for bit.  Measured on the 27-point operator at 256^3
// (tools/spmv_time.py, alternating runs): row patterns 1.13 -> 0.88 ms (0.63 -> 0.81 of 8 TB/s on the CSR bytes), with the fused
// dot 1.24 -> 0.94 ms, stored columns read 1.16 -> 1.12 ms.  What the steps gave: wave-private staging 1.13 -> 1.00; the walk
// without a load under a branch (a lane past its row's end reads entry 0 and keeps its sum by a select) 0.98 -> 0.90; the compact
// copy of the 64-row offsets, gather widths of 8 / 14 / 28 and half-size passes at twice the occupancy: nothing (+-2 %).
// The same select form of the walk inside k_csr_tr at 512^3 (7-entry rows, stored columns read): 2.38-2.39 -> 2.43-2.51 ms --
// not kept there; and this kernel on the shell surrogate's 35-entry rows (RAMD_CSR_W4=0): 0.185 ms against k_csr_wp's 0.152.
// entries of a wave's LDS image per pass: with row patterns (8 bytes an entry) the rows of a wave in one pass, two workgroups per
// CU; with the stored columns (12 bytes an entry) half of that, three workgroups per CU
template <bool PAT>
constexpr int kWrCapOf = PAT ? 2048 : 1024;
template <typename T, int MODE, bool DOT, bool PAT, int GW>
__global__ __launch_bounds__(kBlock) void k_csr_wr(int nrow, int nblk, int per_xcd, const int* __restrict__ rp, const int* __restrict__ ci,
                                                   const T* __restrict__ val, const T* __restrict__ x, T* __restrict__ y, T scalar,
                                                   CsrDotWs ws, int slot, BandMap bm, CsrPattern pat, const int* __restrict__ wav_rp)
{
    using VP             = typename ValPk<T>::type;
    constexpr int VN     = ValPk<T>::N;
    constexpr int kWrCap = kWrCapOf<PAT>;
    extern __shared__ __attribute__((aligned(16))) char wr_lds[];
    T*   sval_all = reinterpret_cast<T*>(wr_lds); // [4][kWrCap]
    int* scol     = reinterpret_cast<int*>(wr_lds + sizeof(T) * 4 * kWrCap); // PAT: the dictionary; else [4][kWrCap] columns
    const int blk = xcd_block(nblk, per_xcd, bm);
    double    dacc = 0.0;
    if(blk >= 0)
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int w0   = blk * kCsrRows + 64 * wave;
        const int row  = w0 + lane;
        int       rs = 0, re = 0, dbase = 0;
        if(PAT)
        {
            for(int i = threadIdx.x; i < pat.n * pat.w; i += kBlock)
                scol[i] = pat.dict[i];
            __syncthreads();
        }
        if(row < nrow)
        {
            rs = rp[row];
            re = rp[row + 1];
            if(PAT)
                dbase = (int)pat.id[row] * pat.w - rs;
        }
        // (from the compact copy: a hit in the L2 -- rp[w0] itself is a miss of a stream as long as the row count, and the
        //  addresses of the value packets hang on it; rs / re of the rows are needed only when the walk begins)
        const int ngrp  = (nrow + 63) >> 6;
        const int start = __builtin_amdgcn_readfirstlane(wav_rp[min(w0 >> 6, ngrp)]);
        const int end   = __builtin_amdgcn_readfirstlane(wav_rp[min((w0 >> 6) + 1, ngrp)]);
        T*        sv    = sval_all + wave * kWrCap;
        int*      sc    = scol + wave * kWrCap;
        T         sum   = (T)0;
        T         xrow  = (T)0;
        bool      have_xrow = false;
        if(MODE == 1 && row < nrow)
            sum = y[row];
        // x[row] for the epilogues (fused dot against x, Jacobi sweep): requested here, one coalesced load per wave -- picking it
        // out of the row's own gathers (k_csr_tr<PAT>) is three vector instructions per entry of a 27-entry row
        if(((DOT && !ws.dotv) || MODE == 2) && row < nrow)
        {
            xrow      = x[row];
            have_xrow = true;
        }
        for(int cb = start & ~3; cb < end; cb += kWrCap)
        {
            v4i32 c[kWrCap / (4 * 64)];
            VP    a[kWrCap / (VN * 64)];
#pragma unroll
            for(int k = 0; k < (PAT ? 0 : kWrCap / (4 * 64)); ++k)
            {
                // (every lane loads: a packet behind the wave's entries re-reads the first one -- a load under a branch makes
                //  the compiler wait for each packet on its own)
                const int j = cb + (k * 64 + lane) * 4;
                c[k]        = nt_load(reinterpret_cast<const v4i32*>(ci + (j < end ? j : cb)));
            }
#pragma unroll
            for(int k = 0; k < kWrCap / (VN * 64); ++k)
            {
                const int j = cb + (k * 64 + lane) * VN;
                a[k]        = nt_load(reinterpret_cast<const VP*>(val + (j < end ? j : cb)));
            }
#pragma unroll
            for(int k = 0; k < (PAT ? 0 : kWrCap / (4 * 64)); ++k)
            {
                const int g = (k * 64 + lane) * 4;
                if(cb + g < end)
                    *reinterpret_cast<v4i32*>(sc + g) = c[k];
            }
#pragma unroll
            for(int k = 0; k < kWrCap / (VN * 64); ++k)
            {
                const int g = (k * 64 + lane) * VN;
                if(cb + g < end)
                    *reinterpret_cast<VP*>(sv + g) = a[k];
            }
            __builtin_amdgcn_wave_barrier();
            const int lo = max(rs, cb), hi = min(re, cb + kWrCap);
            // GW gathers in flight per row (two waves per SIMD leave this kernel the registers for a whole 27-entry row), and
            // no load under a branch: a lane past its row's end reads entry 0 / x[0] and keeps its sum by a select
            const int len = hi - lo;
            for(int jb = 0; __ballot(jb < len) != 0ull; jb += GW)
            {
                int  cc[GW];
                T    v[GW], xv[GW];
                bool ok[GW];
#pragma unroll
                for(int e = 0; e < GW; ++e)
                {
                    ok[e]         = jb + e < len;
                    const int idx = ok[e] ? lo - cb + jb + e : 0;
                    v[e]          = sv[idx];
                    const int col = PAT ? row + scol[ok[e] ? dbase + lo + jb + e : 0] : sc[idx];
                    cc[e]         = ok[e] ? col : 0;
                }
#pragma unroll
                for(int e = 0; e < GW; ++e)
                    xv[e] = x[cc[e]];
#pragma unroll
                for(int e = 0; e < GW; ++e)
                {
                    const T s2 = MODE != 1 ? sum + v[e] * xv[e] : sum + scalar * v[e] * xv[e];
                    sum        = ok[e] ? s2 : sum;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if(row < nrow)
        {
            if((DOT && !ws.dotv) || MODE == 2)
                if(!have_xrow)
                    xrow = x[row];
            if(MODE == 2)
            {
                T t = (T)(-1) * sum + static_cast<const T*>(ws.jrhs)[row];
                t   = static_cast<const T*>(ws.jdinv)[row] * t;
                sum = xrow + scalar * t;
            }
            nt_store(sum, y + row);
            if(DOT)
                dacc = (double)sum * (double)(ws.dotv ? static_cast<const T*>(ws.dotv)[row] : xrow);
        }
    }
    if(DOT)
    {
        const double wsum = wave_reduce_sum(dacc);
        if((threadIdx.x & 63) == 0 && blk >= 0)
            ws.part1[blk * (kBlock / 64) + (threadIdx.x >> 6)] = wsum;
    }
}

// Row-pattern product, TWO row blocks per workgroup and a shorter dependency chain (k_csr_tr<PAT> reworked).  PMC passes on
// k_csr_tr<PAT> at 512^3 (tools/pmc_spmv.sh, gpurun_out/r03bc): 82 % of the wave cycles are SQ_WAIT_ANY (parked at a
// waitcnt or the barrier), 13 % issue; a wave lives 7.2 us -- the product is bound by the latency chain of a workgroup
// times the waves a CU holds, not by HBM (10.3 GB in 2.2-2.3 ms = 4.6 TB/s where two read streams reach 7.1 TB/s).  The
// chain was: dictionary fetched and staged (an L2 round trip before anything else) -> rp[r0] (a MISS of a 0.5-GB stream:
// the addresses of the value packets hang on it) -> value packets (HBM) -> LDS, barrier -> two rounds of x gathers -> store.
// Here: the value packets of BOTH row blocks are requested at once, their addresses come from a compact copy of the block
// offsets (blk_rp: 2 MB at 512^3, cache-resident), the dictionary is requested first but staged last, and block 0 is walked
// while block 1's packets are still on their way.  Same products in the same order per row, the same per-wave partials
// of the fused dot in the same places: results identical to k_csr_tr<PAT> bit for bit.  Needs every row block's entries to
// fit one LDS pass (256 x longest pattern + 3 <= kCsrChunk).  Measured, alternating with k_csr_tr<PAT> on one box
// (gpurun_out/r03bd, r03be): plain product 2.19-2.27 -> 1.95-1.98 ms, product + dot 2.23-2.33 -> 1.93-2.08 ms (0.84-0.90 of
// 8 TB/s on algorithmic CSR bytes), inside the CG loop 2.36 -> 1.93-2.15 ms.  (A first measurement of this kernel reported
// "slower": its switch tested the dictionary stride instead of the longest pattern and the kernel never ran -- the
// difference it showed was box-to-box noise, which says how large that is.)  Three and four row blocks per workgroup:
// plain product equal (1.94-2.15 ms), product + dot clearly slower (2.26-2.41 vs 2.05 ms; gpurun_out/r03bh).
// RAMD_CSR_PAT2=0: the old kernel.
__global__ __launch_bounds__(kBlock) void k_blk_rp(int nrow, int nblk, const int* __restrict__ rp, int* __restrict__ blk_rp)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= nblk; b += gsz)
    {
        const int64_t r = b * kCsrRows;
        blk_rp[b]       = rp[r < nrow ? r : nrow];
    }
}

// (the same for groups of 64 rows: what a wave of k_csr_wr stages)
__global__ __launch_bounds__(kBlock) void k_wav_rp(int nrow, int ngrp, const int* __restrict__ rp, int* __restrict__ wav_rp)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g <= ngrp; g += gsz)
    {
        const int64_t r = g * 64;
        wav_rp[g]       = rp[r < nrow ? r : nrow];
    }
}

__device__ __forceinline__ int xcd_block_at(int i, int nblk, int per_xcd, BandMap bm)
{
    int l = i;
    if(bm.P > 0 && i < bm.Z * bm.P)
    {
        const int tile   = i / (bm.W * bm.Z);
        const int within = i - tile * (bm.W * bm.Z);
        const int z      = within / bm.W;
        const int w      = within - z * bm.W;
        l                = z * bm.P + tile * bm.W + w;
    }
    const int b = (blockIdx.x & 7) * per_xcd + l;
    return (i < per_xcd && b < nblk) ? b : -1;
}

// PAT = false: the same two-block structure with the stored columns read (their packets travel with the value packets):
// matrices whose 256-row blocks all fit one LDS pass (short rows: 5- / 7-point operators, unstructured graphs of low degree).
// NORP (row patterns only): the row offsets are not read either -- a row's length is its pattern's, its first entry the
// block's offset plus the lengths of the rows before it in the block (a scan over 256 lengths through DPP and 4 LDS words,
// done while the value packets are in flight): 4 bytes per row less, 0.54 of 10.6 GB at 512^3.
struct PatLens
{
    unsigned char len[kPatMax];
};
template <typename T, int MODE, bool DOT, int NB, bool PAT = true, bool NORP = false>
__global__ __launch_bounds__(kBlock) void k_csr_pat2(
    int nrow, int nblk, int per_xcd, const int* __restrict__ rp, const int* __restrict__ ci, const T* __restrict__ val,
    const T* __restrict__ x, T* __restrict__ y, T scalar, CsrDotWs ws, int slot, BandMap bm, CsrPattern pat,
    const int* __restrict__ blk_rp, PatLens pl = {})
{
    static_assert(!NORP || PAT, "row offsets can only be rebuilt from row patterns");
    using VP           = typename ValPk<T>::type;
    constexpr int VN   = ValPk<T>::N;
    constexpr int NPKT = kCsrChunk / (VN * kBlock);
    constexpr int NCPK = PAT ? 0 : kCsrChunk / (4 * kBlock); // column packets per thread and block
    constexpr int NDW  = PAT ? kPatMax * kPatMaxW / kBlock : 0; // dictionary words per thread
    __shared__ T   sval[kCsrChunk];
    __shared__ int scol[PAT ? kPatMax * kPatMaxW : kCsrChunk]; // PAT: the dictionary of column offsets
    // the dictionary is requested first and staged LAST, after the value packets are on their way: its round trip (an L2
    // hit) overlaps theirs instead of preceding them
    int dreg[NDW > 0 ? NDW : 1];
#pragma unroll
    for(int q = 0; q < NDW; ++q)
    {
        const int i = q * kBlock + threadIdx.x;
        dreg[q]     = i < pat.n * pat.w ? pat.dict[i] : 0;
    }
    __shared__ int slen[NORP ? kPatMax : 1];
    __shared__ int swsum[NORP ? NB * (kBlock / 64) : 1];
    int   blk[NB], rs[NB], re[NB], dbase[NB], cb[NB], end[NB], pid[NB];
    VP    a[NB][NPKT];
    v4i32 c[NB][NCPK > 0 ? NCPK : 1];
    T     sum[NB];
#pragma unroll
    for(int h = 0; h < NB; ++h)
    {
        blk[h]        = xcd_block_at(NB * (int)(blockIdx.x >> 3) + h, nblk, per_xcd, bm);
        rs[h] = re[h] = dbase[h] = cb[h] = end[h] = 0;
        pid[h]                                    = -1;
        sum[h]                                    = (T)0;
        if(blk[h] >= 0)
        {
            const int r0  = blk[h] * kCsrRows;
            const int row = r0 + threadIdx.x;
            if(row < nrow)
            {
                if(NORP)
                    pid[h] = (int)pat.id[row];
                else
                {
                    rs[h] = rp[row];
                    re[h] = rp[row + 1];
                    if(PAT)
                        dbase[h] = (int)pat.id[row] * pat.w - rs[h];
                }
                if(MODE == 1)
                    sum[h] = y[row];
            }
            // (block offsets from the compact copy: 2 MB at 512^3, cache-resident, where rp[r0] is a miss of a 0.5-GB
            //  stream -- the value packets' addresses hang on it)
            cb[h]  = blk_rp[blk[h]] & ~3;
            end[h] = blk_rp[blk[h] + 1];
#pragma unroll
            for(int k = 0; k < NCPK; ++k)
            {
                const int j = cb[h] + (k * kBlock + threadIdx.x) * 4;
                if(j < end[h])
                    c[h][k] = nt_load(reinterpret_cast<const v4i32*>(ci + j));
            }
#pragma unroll
            for(int k = 0; k < NPKT; ++k)
            {
                const int j = cb[h] + (k * kBlock + threadIdx.x) * VN;
                if(j < end[h])
                    a[h][k] = nt_load(reinterpret_cast<const VP*>(val + j));
            }
        }
    }
#pragma unroll
    for(int q = 0; q < NDW; ++q)
    {
        const int i = q * kBlock + threadIdx.x;
        if(i < pat.n * pat.w)
            scol[i] = dreg[q];
    }
    if(NORP)
    {
        if(threadIdx.x < kPatMax)
            slen[threadIdx.x] = (int)pl.len[threadIdx.x];
        __syncthreads();
        int len[NB], excl[NB];
#pragma unroll
        for(int h = 0; h < NB; ++h)
        {
            len[h] = pid[h] >= 0 ? slen[pid[h]] : 0;
            int inc = len[h]; // inclusive scan over the wave
#pragma unroll
            for(int o = 1; o < 64; o <<= 1)
            {
                const int up = __shfl_up(inc, o, 64);
                if((int)(threadIdx.x & 63) >= o)
                    inc += up;
            }
            excl[h] = inc - len[h];
            if((threadIdx.x & 63) == 63)
                swsum[h * (kBlock / 64) + (threadIdx.x >> 6)] = inc;
        }
        __syncthreads();
#pragma unroll
        for(int h = 0; h < NB; ++h)
        {
            int before = 0;
#pragma unroll
            for(int w = 0; w < kBlock / 64; ++w)
                before += (w < (int)(threadIdx.x >> 6)) ? swsum[h * (kBlock / 64) + w] : 0;
            const int first = blk[h] >= 0 ? blk_rp[blk[h]] : 0; // (the block's true offset: cb is its 4-aligned floor)
            rs[h]    = first + before + excl[h];
            re[h]    = rs[h] + len[h];
            dbase[h] = pid[h] >= 0 ? pid[h] * pat.w - rs[h] : 0;
        }
    }
    double dacc[NB];
#pragma unroll
    for(int h = 0; h < NB; ++h)
        dacc[h] = 0.0;
#pragma unroll
    for(int h = 0; h < NB; ++h)
    {
        if(h > 0)
            __syncthreads(); // (the previous block's values have been read by everybody)
        if(blk[h] >= 0)
        {
#pragma unroll
            for(int k = 0; k < NCPK; ++k)
            {
                const int g = (k * kBlock + threadIdx.x) * 4;
                if(cb[h] + g < end[h])
                    *reinterpret_cast<v4i32*>(scol + g) = c[h][k];
            }
#pragma unroll
            for(int k = 0; k < NPKT; ++k)
            {
                const int g = (k * kBlock + threadIdx.x) * VN;
                if(cb[h] + g < end[h])
                    *reinterpret_cast<VP*>(sval + g) = a[h][k];
            }
        }
        __syncthreads();
        if(blk[h] >= 0)
        {
            const int row = blk[h] * kCsrRows + threadIdx.x;
            T         sm  = sum[h];
            T         xrow      = (T)0;
            bool      have_xrow = false;
            for(int j = rs[h]; j < re[h]; j += kGatherW)
            {
                int cc[kGatherW];
                T   v[kGatherW], xv[kGatherW];
#pragma unroll
                for(int e = 0; e < kGatherW; ++e)
                    if(j + e < re[h])
                    {
                        cc[e] = PAT ? row + scol[dbase[h] + j + e] : scol[j - cb[h] + e];
                        v[e]  = sval[j - cb[h] + e];
                    }
#pragma unroll
                for(int e = 0; e < kGatherW; ++e)
                    if(j + e < re[h])
                        xv[e] = x[cc[e]];
#pragma unroll
                for(int e = 0; e < kGatherW; ++e)
                    if(j + e < re[h])
                    {
                        if(MODE != 1)
                            sm += v[e] * xv[e];
                        else
                            sm += scalar * v[e] * xv[e];
                        if(PAT && (DOT || MODE == 2) && cc[e] == row)
                        {
                            xrow      = xv[e];
                            have_xrow = true;
                        }
                    }
            }
            if(row < nrow)
            {
                if((DOT && !ws.dotv) || MODE == 2)
                    if(!have_xrow)
                        xrow = x[row];
                if(MODE == 2)
                {
                    T t = (T)(-1) * sm + static_cast<const T*>(ws.jrhs)[row];
                    t   = static_cast<const T*>(ws.jdinv)[row] * t;
                    sm  = xrow + scalar * t;
                }
                nt_store(sm, y + row);
                if(DOT)
                    dacc[h] = (double)sm * (double)(ws.dotv ? static_cast<const T*>(ws.dotv)[row] : xrow);
            }
        }
    }
    if(DOT)
    {
#pragma unroll
        for(int h = 0; h < NB; ++h)
        {
            const double wsum = wave_reduce_sum(dacc[h]);
            if((threadIdx.x & 63) == 0 && blk[h] >= 0)
                ws.part1[blk[h] * (kBlock / 64) + (threadIdx.x >> 6)] = wsum;
        }
    }
}

// Measured and removed (round 4, alternating runs at 512^3, product + dot inside the CG loop; k_csr_pat2: 2.04-2.09 ms):
//  * a PERSISTENT form -- workgroups that stay, the XCD's block pairs dealt out with stride "workgroups per XCD", the dictionary
//    staged once, the packets of pair k + 1 requested before pair k is walked (146 VGPRs, 3 workgroups per CU): 2.61-2.63 ms,
//    3.36 ms with 2 workgroups per CU (gpurun_out/r04f);
//  * MORE waves instead: the second block's packets requested into the first block's registers once those are in LDS, 7 / 8
//    waves per SIMD asked of the compiler: 2.17-2.20 ms / 3.11-3.16 ms (the walk's 40 gather registers spill); both blocks
//    requested at once with 7 waves: 2.08-2.12 ms (gpurun_out/r04g).
// Neither a longer stream per workgroup nor more resident waves moves this product: what it moves per row (79 bytes) comes at
// 5.2 TB/s like the other gather kernels (colour sweeps 5.0-5.4, ELL with patterns 5.0), whichever way it is requested.
// largest number of entries (from the 4-aligned start) a 256-row block stages: decides whether a matrix takes k_csr_pat2
__global__ __launch_bounds__(kBlock) void k_blk_span_max(int nblk, const int* __restrict__ blk_rp, int* __restrict__ out)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    int           mx  = 0;
    for(int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += gsz)
        mx = max(mx, blk_rp[b + 1] - (blk_rp[b] & ~3));
#pragma unroll
    for(int o = 32; o > 0; o >>= 1)
        mx = max(mx, __shfl_xor(mx, o, 64));
    if((threadIdx.x & 63) == 0 && mx > 0)
        atomicMax(out, mx);
}

// CSR SpMV of a STRUCTURED matrix with x tiles in LDS (row patterns, csr_analyse_xl).  k_csr_tr<PAT> no longer reads the
// columns, so what it moves per row is the values (7 x 8 B on the 7-point operator) -- and 7 x 8 B of x through the gather
// path: 28 wave-level 8-byte loads per 256 rows, each 4-5 lines (the +-1 neighbours straddle), 4.6-4.9 TB/s of the bytes
// moved.  With the dictionary known a block needs, per CLUSTER of column offsets, one contiguous piece of x (the 7-point
// operator: 5 pieces of 2 KiB instead of 7 gathers).  The pieces and the block's values go straight into LDS with 16-byte
// packets (global_load_lds: no staging registers, no ds_write pass); the row walk then only touches LDS.  Same products,
// same left-to-right sums.
// Measured (round 3, 512^3, profiles/r03_spmv_xl.txt): the plain product 2.17-2.19 ms against 2.18-2.30 ms of k_csr_tr<PAT>
// (equal once the allocations are placed by the arena allocator, 5 % ahead without it), but the form the CG loop uses --
// the product carrying <p, q> -- 2.44-2.52 ms against 2.24-2.29 ms: the x tiles do not buy what the request concurrency
// of the gather form already delivers (x is served by the L2 in both), and the two dependent memory phases per block
// (x pieces + row offsets, then the values) leave the CU idle longer than the gather form's overlap of its own blocks.
// Kept as an opt-in (RAMD_CSR_XL=1) under the same bit-exact tests; the default stays the gather form.
template <typename T, int MODE, bool DOT>
__global__ __launch_bounds__(kBlock) void k_csr_xl(int nrow, int nblk, int per_xcd, const int* __restrict__ rp,
                                                   const T* __restrict__ val, const T* __restrict__ x, T* __restrict__ y,
                                                   T scalar, CsrDotWs ws, int slot, BandMap bm, CsrPattern pat, XlSegs sg,
                                                   const int* __restrict__ blk_rp)
{
    constexpr int VN = 16 / (int)sizeof(T);
    // LDS (dynamic: what this matrix needs decides how many workgroups a CU holds): values of a pass | x pieces | dictionary
    extern __shared__ __attribute__((aligned(16))) char xl_lds[];
    T*   sval  = reinterpret_cast<T*>(xl_lds);
    T*   sx    = sval + kCsrChunk;
    int* sdict = reinterpret_cast<int*>(sx + sg.total);
    using GP = const __attribute__((address_space(1))) void*;
    using LP = __attribute__((address_space(3))) void*;
    const int blk  = xcd_block(nblk, per_xcd, bm);
    const int tid  = threadIdx.x;
    const int wtid = tid & ~63; // first thread of my wave
    double    dacc = 0.0;
    if(blk >= 0)
    {
        const int r0   = blk * kCsrRows;
        const int rend = min(r0 + kCsrRows, nrow);
        const int row  = r0 + tid;
        // x pieces: packet q of piece s holds x[r0 + omin + q * VN ...]; packets outside [0, nrow) are skipped (nothing
        // reads them: a column that exists lies inside), the vector is padded by 256 B beyond nrow
#pragma unroll 1
        for(int s = 0; s < sg.nseg; ++s)
        {
            const int64_t g0 = (int64_t)r0 + sg.omin[s];
            for(int q0 = 0; q0 < sg.npk[s]; q0 += kBlock)
            {
                const int     q = q0 + tid;
                const int64_t g = g0 + (int64_t)q * VN;
                if(q < sg.npk[s] && g >= 0 && g < nrow)
                    __builtin_amdgcn_global_load_lds((GP)(x + g), (LP)(sx + sg.base[s] + (q0 + wtid) * VN), 16, 0, 0);
            }
        }
        // (block offsets from the compact copy where there is one: the first pass's value packets then hang on nothing but
        //  an L2 hit and leave together with the x pieces; dictionary and row offsets are requested after them -- k_csr_pat2)
        const int start = blk_rp ? blk_rp[blk] : rp[r0];
        const int end   = blk_rp ? blk_rp[blk + 1] : rp[rend];
        {
            const int cb0 = start & ~3;
#pragma unroll
            for(int k = 0; k < kCsrChunk / (VN * kBlock); ++k)
            {
                const int j = cb0 + (k * kBlock + tid) * VN;
                if(j < end)
                    __builtin_amdgcn_global_load_lds((GP)(val + j), (LP)(sval + (k * kBlock + wtid) * VN), 16, 0, 2);
            }
        }
        for(int i = tid; i < pat.n * pat.w; i += kBlock)
            sdict[i] = pat.dict[i];
        int rs = 0, re = 0, dbase = 0;
        if(row < nrow)
        {
            rs    = rp[row];
            re    = rp[row + 1];
            dbase = (int)pat.id[row] * pat.w - rs;
        }
        T sum = (T)0;
        if(MODE == 1 && row < nrow)
            sum = y[row];
        for(int cb = start & ~3; cb < end; cb += kCsrChunk)
        {
            if(cb != (start & ~3)) // (the first pass is on its way already)
            {
#pragma unroll
                for(int k = 0; k < kCsrChunk / (VN * kBlock); ++k)
                {
                    const int j = cb + (k * kBlock + tid) * VN;
                    if(j < end)
                        __builtin_amdgcn_global_load_lds((GP)(val + j), (LP)(sval + (k * kBlock + wtid) * VN), 16, 0, 2);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int lo = max(rs, cb), hi = min(re, cb + kCsrChunk);
            for(int j = lo; j < hi; j += kGatherW)
            {
                T v[kGatherW], xv[kGatherW];
#pragma unroll
                for(int e = 0; e < kGatherW; ++e)
                    if(j + e < hi)
                    {
                        v[e]  = sval[j - cb + e];
                        xv[e] = sx[sdict[dbase + j + e] + tid];
                    }
#pragma unroll
                for(int e = 0; e < kGatherW; ++e)
                    if(j + e < hi)
                    {
                        if(MODE != 1)
                            sum += v[e] * xv[e];
                        else
                            sum += scalar * v[e] * xv[e];
                    }
            }
            __syncthreads();
        }
        if(row < nrow)
        {
            if(MODE == 2)
            {
                T t = (T)(-1) * sum + static_cast<const T*>(ws.jrhs)[row];
                t   = static_cast<const T*>(ws.jdinv)[row] * t;
                sum = x[row] + scalar * t;
            }
            nt_store(sum, y + row);
            if(DOT)
                dacc = (double)sum * (double)(ws.dotv ? static_cast<const T*>(ws.dotv)[row] : x[row]);
        }
    }
    if(DOT)
    {
        const double wsum = wave_reduce_sum(dacc);
        if((threadIdx.x & 63) == 0 && blk >= 0)
            ws.part1[blk * (kBlock / 64) + (threadIdx.x >> 6)] = wsum;
    }
}

// CSR SpMV for LONG rows (mean row length > 16: FE matrices with several unknowns per node, 27-point stencils).
// With one thread per row a 2048-entry LDS pass holds only 2048 / row_length rows, so 3 of 4 lanes idle in the row walk
// of 35-entry rows and every gather instruction carries 16 addresses (k_csr_tr: 4.1 TB/s = 51 % on the af_shell10-class
// matrix).  Here FOUR lanes share a row: the same raw stream of the workgroup's entries through LDS, then quad q walks the
// rows q, q+64, ... that intersect the pass; lane l gathers x for the entries l, l+4, ... of its row (consecutive lanes =
// consecutive entries, whose columns are mostly the unknowns of one node: few lines per gather instruction, all 64 lanes
// busy) and forms the rounded product; the row sum is then taken IN STORAGE ORDER by every lane of the quad from the
// quad-broadcast products (DPP quad_perm: no LDS round trip) -- the same roundings in the same order as the host loop.
// A row that straddles two passes carries its running sum in LDS.
template <int E, typename T>
__device__ __forceinline__ T quad_bcast(T v)
{
    constexpr int ctrl = E * 85; // quad_perm [E, E, E, E]
    if constexpr(sizeof(T) == 8)
    {
        const long long b  = __builtin_bit_cast(long long, v);
        const int       lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), ctrl, 0xf, 0xf, true);
        const int       hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), ctrl, 0xf, 0xf, true);
        return __builtin_bit_cast(T, ((long long)hi << 32) | (long long)(unsigned)lo);
    }
    else
        return __builtin_bit_cast(T, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}
constexpr int kQ4Batch = 4; // gathers in flight per lane (16 entries of a row per batch)
template <typename T, int MODE, bool DOT>
__global__ __launch_bounds__(kBlock) void k_csr_q4(int nrow, int nblk, int per_xcd, const int* __restrict__ rp,
                                                   const int* __restrict__ ci, const T* __restrict__ val,
                                                   const T* __restrict__ x, T* __restrict__ y, T scalar, CsrDotWs ws,
                                                   int slot)
{
    using VP         = typename ValPk<T>::type;
    constexpr int VN = ValPk<T>::N;
    __shared__ T   sval[kCsrChunk];
    __shared__ int scol[kCsrChunk];
    __shared__ int srp[kCsrRows + 1];
    __shared__ T   ssum[kCsrRows];
    const BandMap bm0  = {0, 0, 0};
    const int     blk  = xcd_block(nblk, per_xcd, bm0);
    double        dacc = 0.0;
    if(blk >= 0)
    {
        const int r0   = blk * kCsrRows;
        const int rend = min(r0 + kCsrRows, nrow);
        const int nr   = rend - r0;
        const int row  = r0 + threadIdx.x;
        if((int)threadIdx.x < nr)
        {
            srp[threadIdx.x]  = rp[row];
            ssum[threadIdx.x] = (MODE == 1) ? y[row] : (T)0;
        }
        if(threadIdx.x == 0)
            srp[nr] = rp[rend];
        __syncthreads();
        const int start = srp[0];
        const int end   = srp[nr];
        const int q = threadIdx.x >> 2, l = threadIdx.x & 3;
        for(int cb = start & ~3; cb < end; cb += kCsrChunk)
        {
            v4i32 c[kCsrChunk / (4 * kBlock)];
            VP    a[kCsrChunk / (VN * kBlock)];
#pragma unroll
            for(int k = 0; k < kCsrChunk / (4 * kBlock); ++k)
            {
                const int j = cb + (k * kBlock + threadIdx.x) * 4;
                if(j < end)
                    c[k] = nt_load(reinterpret_cast<const v4i32*>(ci + j));
            }
#pragma unroll
            for(int k = 0; k < kCsrChunk / (VN * kBlock); ++k)
            {
                const int j = cb + (k * kBlock + threadIdx.x) * VN;
                if(j < end)
                    a[k] = nt_load(reinterpret_cast<const VP*>(val + j));
            }
#pragma unroll
            for(int k = 0; k < kCsrChunk / (4 * kBlock); ++k)
            {
                const int g = (k * kBlock + threadIdx.x) * 4;
                if(cb + g < end)
                    *reinterpret_cast<v4i32*>(scol + g) = c[k];
            }
#pragma unroll
            for(int k = 0; k < kCsrChunk / (VN * kBlock); ++k)
            {
                const int g = (k * kBlock + threadIdx.x) * VN;
                if(cb + g < end)
                    *reinterpret_cast<VP*>(sval + g) = a[k];
            }
            __syncthreads();
            for(int r = q; r < nr; r += kBlock / 4)
            {
                const int lo = max(srp[r], cb), hi = min(srp[r + 1], cb + kCsrChunk);
                if(lo >= hi)
                    continue;
                T sum = ssum[r];
                for(int j = lo; j < hi; j += 4 * kQ4Batch)
                {
                    int cc[kQ4Batch];
                    T   p[kQ4Batch], xv[kQ4Batch];
#pragma unroll
                    for(int e = 0; e < kQ4Batch; ++e)
                    {
                        const int jj = j + 4 * e + l;
                        cc[e]        = -1;
                        if(jj < hi)
                        {
                            cc[e] = scol[jj - cb];
                            p[e]  = sval[jj - cb];
                        }
                    }
#pragma unroll
                    for(int e = 0; e < kQ4Batch; ++e)
                        if(cc[e] >= 0)
                            xv[e] = x[cc[e]];
#pragma unroll
                    for(int e = 0; e < kQ4Batch; ++e)
                    {
                        if(cc[e] >= 0)
                            p[e] = (MODE != 1) ? p[e] * xv[e] : scalar * p[e] * xv[e];
                        const int left = hi - (j + 4 * e); // entries of this group of four (quad-uniform)
                        if(left >= 4)
                        {
                            sum += quad_bcast<0>(p[e]);
                            sum += quad_bcast<1>(p[e]);
                            sum += quad_bcast<2>(p[e]);
                            sum += quad_bcast<3>(p[e]);
                        }
                        else if(left > 0)
                        {
                            const T b0 = quad_bcast<0>(p[e]), b1 = quad_bcast<1>(p[e]), b2 = quad_bcast<2>(p[e]);
                            sum += b0;
                            if(left > 1)
                                sum += b1;
                            if(left > 2)
                                sum += b2;
                        }
                    }
                }
                if(l == 0)
                    ssum[r] = sum;
            }
            __syncthreads();
        }
        if((int)threadIdx.x < nr)
        {
            T sum = ssum[threadIdx.x];
            if(MODE == 2)
            {
                T t = (T)(-1) * sum + static_cast<const T*>(ws.jrhs)[row];
                t   = static_cast<const T*>(ws.jdinv)[row] * t;
                sum = x[row] + scalar * t;
            }
            nt_store(sum, y + row);
            if(DOT)
                dacc = (double)sum * (double)(ws.dotv ? static_cast<const T*>(ws.dotv)[row] : x[row]);
        }
    }
    if(DOT)
    {
        const double wsum = wave_reduce_sum(dacc);
        if((threadIdx.x & 63) == 0 && blk >= 0)
            ws.part1[blk * (kBlock / 64) + (threadIdx.x >> 6)] = wsum;
    }
}

// CSR SpMV for rows of 16+ entries, WAVE-PRIVATE passes (k_csr_w4).  Counters on k_csr_tr with the 35-entry rows of the
// config-3 surrogate (tools/pmc_spmv_shell.sh, profiles/r03_pmc_spmv_shell.txt): 81 % of the wave cycles are waits, on average
// 0.08 vector memory instructions in flight per wave -- a 2048-entry LDS pass of the workgroup holds the rows of ONE of its
// four waves, the other three stand at the barrier while that wave walks its rows (five passes, ten barriers per workgroup).
// Here a wave owns its 64 rows from the first packet to the store: it stages 16 rows at a time (four lanes per row, as in
// k_csr_q4) through its own piece of LDS, with no workgroup barrier anywhere; the four waves of a workgroup and the
// workgroups of a CU are in different phases at any time.  Products and sums as in k_csr_q4: lane l of a quad takes the
// entries l, l + 4, ... of the row, the row sum runs in storage order over the quad-broadcast products -- bit-identical to
// the host loop.  The finished sums pass through LDS so that lane t ends up with row t of the wave: the epilogue (Jacobi
// sweep, fused dot with ONE partial per wave) is the one of k_csr_tr, same order of additions.
constexpr int kW4Chunk = 640; // entries per wave and pass (16 rows of up to 40 entries in one pass)
constexpr int kW4Batch = 3; // gathers in flight per lane (12 entries of a row per round)
template <typename T, int MODE, bool DOT, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_csr_w4(int nrow, int nblk, int per_xcd, const int* __restrict__ rp,
                                                   const int* __restrict__ ci, const T* __restrict__ val,
                                                   const T* __restrict__ x, T* __restrict__ y, T scalar, CsrDotWs ws,
                                                   int slot, BandMap bm)
{
    using VP           = typename ValPk<T>::type;
    constexpr int VN   = ValPk<T>::N;
    constexpr int CH   = kW4Chunk;
    constexpr int NCP  = (CH / 4 + 63) / 64; // column packets per lane and pass
    constexpr int NVP  = (CH / VN + 63) / 64; // value packets per lane and pass
    __shared__ __attribute__((aligned(16))) T   sval[NWV][CH];
    __shared__ __attribute__((aligned(16))) int scol[NWV][CH];
    __shared__ T   ssum[NWV][64];
    const int blk  = xcd_block(nblk, per_xcd, bm);
    double    dacc = 0.0;
    if(blk >= 0)
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int q = lane >> 2, l = lane & 3;
        const int wbase = (blk * NWV + wave) * 64;
        T*        sv    = sval[wave];
        int*      sc    = scol[wave];
        for(int s4 = 0; s4 < 4; ++s4)
        {
            const int first = wbase + 16 * s4;
            if(first >= nrow) // (wave-uniform)
                break;
            const int row = first + q;
            int       rs = 0, re = 0;
            if(row < nrow)
            {
                rs = rp[row];
                re = rp[row + 1];
            }
            const int lastq = min(15, nrow - 1 - first);
            const int S     = __builtin_amdgcn_readlane(rs, 0);
            const int E     = __shfl(re, 4 * lastq, 64);
            T         sum   = (MODE == 1 && row < nrow) ? y[row] : (T)0;
            for(int cb = S & ~3; cb < E; cb += CH)
            {
                v4i32 c[NCP];
                VP    a[NVP];
#pragma unroll
                for(int k = 0; k < NCP; ++k)
                {
                    const int g = (k * 64 + lane) * 4;
                    if(g < CH && cb + g < E)
                        c[k] = nt_load(reinterpret_cast<const v4i32*>(ci + cb + g));
                }
#pragma unroll
                for(int k = 0; k < NVP; ++k)
                {
                    const int g = (k * 64 + lane) * VN;
                    if(g < CH && cb + g < E)
                        a[k] = nt_load(reinterpret_cast<const VP*>(val + cb + g));
                }
#pragma unroll
                for(int k = 0; k < NCP; ++k)
                {
                    const int g = (k * 64 + lane) * 4;
                    if(g < CH && cb + g < E)
                        *reinterpret_cast<v4i32*>(sc + g) = c[k];
                }
#pragma unroll
                for(int k = 0; k < NVP; ++k)
                {
                    const int g = (k * 64 + lane) * VN;
                    if(g < CH && cb + g < E)
                        *reinterpret_cast<VP*>(sv + g) = a[k];
                }
                // (one wave: its LDS operations complete in order; the barrier only keeps the compiler from moving reads up)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int lo = max(rs, cb), hi = min(re, cb + CH);
                for(int j = lo; j < hi; j += 4 * kW4Batch)
                {
                    int cc[kW4Batch];
                    T   p[kW4Batch], xv[kW4Batch];
#pragma unroll
                    for(int e = 0; e < kW4Batch; ++e)
                    {
                        const int jj = j + 4 * e + l;
                        cc[e]        = -1;
                        if(jj < hi)
                        {
                            cc[e] = sc[jj - cb];
                            p[e]  = sv[jj - cb];
                        }
                    }
#pragma unroll
                    for(int e = 0; e < kW4Batch; ++e)
                        if(cc[e] >= 0)
                            xv[e] = x[cc[e]];
#pragma unroll
                    for(int e = 0; e < kW4Batch; ++e)
                    {
                        if(cc[e] >= 0)
                            p[e] = (MODE != 1) ? p[e] * xv[e] : scalar * p[e] * xv[e];
                        const int left = hi - (j + 4 * e); // entries of this group of four (quad-uniform)
                        if(left >= 4)
                        {
                            sum += quad_bcast<0>(p[e]);
                            sum += quad_bcast<1>(p[e]);
                            sum += quad_bcast<2>(p[e]);
                            sum += quad_bcast<3>(p[e]);
                        }
                        else if(left > 0)
                        {
                            const T b0 = quad_bcast<0>(p[e]), b1 = quad_bcast<1>(p[e]), b2 = quad_bcast<2>(p[e]);
                            sum += b0;
                            if(left > 1)
                                sum += b1;
                            if(left > 2)
                                sum += b2;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier(); // the next pass overwrites what this one read
            }
            if(l == 0)
                ssum[wave][16 * s4 + q] = sum;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int row = wbase + lane;
        if(row < nrow)
        {
            T sum = ssum[wave][lane];
            T xrow = (T)0;
            if((DOT && !ws.dotv) || MODE == 2)
                xrow = x[row];
            if(MODE == 2)
            {
                T t = (T)(-1) * sum + static_cast<const T*>(ws.jrhs)[row];
                t   = static_cast<const T*>(ws.jdinv)[row] * t;
                sum = xrow + scalar * t;
            }
            nt_store(sum, y + row);
            if(DOT)
                dacc = (double)sum * (double)(ws.dotv ? static_cast<const T*>(ws.dotv)[row] : xrow);
        }
    }
    if(DOT)
    {
        const double wsum = wave_reduce_sum(dacc);
        if((threadIdx.x & 63) == 0 && blk >= 0)
            ws.part1[blk * NWV + (threadIdx.x >> 6)] = wsum;
    }
}

// CSR SpMV for rows of 16+ entries, products formed WHERE THE PACKETS LAND (k_csr_wp, round 6).  k_csr_w4 on the reference's own
// 3-D operator (27 entries per row, 256^3): 1.42 ms = 0.50 of the roofline, behind the vendor's csrmv (1.14 ms) -- a quad walks
// its row in three dependent rounds (LDS read -> gather -> broadcast sum) and sixteen rows are all a pass holds.  Here a lane
// keeps the packets it loads: four consecutive entries (one column packet, the value packets over them), four gathers of x, four
// products -- every gather of a 1024-entry pass independent of every other, issued back to back -- and only the PRODUCTS pass
// through the wave's piece of LDS (8 bytes per entry instead of 12).  Lane t then sums row t of the wave from LDS, left to
// right in storage order: the additions of the host loop, bit-identical; the sum ends in the lane that stores it.  A wave owns
// its 64 rows from the first packet to the store, no workgroup barrier (as k_csr_w4).
constexpr int kWpChunk = 1024; // entries per wave and pass
template <typename T, int MODE, bool DOT, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_csr_wp(int nrow, int nblk, int per_xcd, const int* __restrict__ rp,
                                                   const int* __restrict__ ci, const T* __restrict__ val,
                                                   const T* __restrict__ x, T* __restrict__ y, T scalar, CsrDotWs ws,
                                                   int slot, BandMap bm)
{
    using VP          = typename ValPk<T>::type;
    constexpr int VN  = ValPk<T>::N; // values per 16-byte packet (2 or 4)
    constexpr int CH  = kWpChunk;
    constexpr int NCP = CH / 4 / 64; // column packets (4 entries each) per lane and pass
    constexpr int VPC = 4 / VN; // value packets over one column packet
    constexpr int SB  = 8; // products a lane reads ahead of its additions
    __shared__ __attribute__((aligned(16))) T sprod[NWV][CH];
    const int blk  = xcd_block(nblk, per_xcd, bm);
    double    dacc = 0.0;
    if(blk >= 0)
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int wbase = (blk * NWV + wave) * 64;
        T*        sp    = sprod[wave];
        if(wbase < nrow) // (wave-uniform)
        {
            const int row = wbase + lane;
            int       rs = 0, re = 0;
            if(row < nrow)
            {
                rs = rp[row];
                re = rp[row + 1];
            }
            const int last = min(63, nrow - 1 - wbase);
            const int S    = __builtin_amdgcn_readlane(rs, 0);
            const int E    = __shfl(re, last, 64);
            if(row >= nrow)
                rs = re = E;
            T sum = (MODE == 1 && row < nrow) ? y[row] : (T)0;
            for(int cb = S & ~3; cb < E; cb += CH)
            {
                // (no test around a request: a packet behind the wave's last entry asks for the pass's first packet instead, an
                //  entry behind it gathers x[0] -- requests in a straight line, nothing waits for anything before the products)
                v4i32 c[NCP];
                VP    a[NCP][VPC];
                T     xv[NCP][4];
#pragma unroll
                for(int k = 0; k < NCP; ++k)
                {
                    const int g  = cb + (k * 64 + lane) * 4;
                    const int gl = g < E ? g : cb;
                    c[k]         = nt_load(reinterpret_cast<const v4i32*>(ci + gl));
#pragma unroll
                    for(int h = 0; h < VPC; ++h)
                        a[k][h] = nt_load(reinterpret_cast<const VP*>(val + gl + h * VN));
                }
#pragma unroll
                for(int k = 0; k < NCP; ++k)
                {
                    const int g = cb + (k * 64 + lane) * 4;
#pragma unroll
                    for(int i = 0; i < 4; ++i)
                        xv[k][i] = x[g + i < E ? c[k][i] : 0]; // (an entry behind the wave's last one may lie behind the matrix' last one)
                }
#pragma unroll
                for(int k = 0; k < NCP; ++k)
                {
                    const int g = cb + (k * 64 + lane) * 4;
                    if(g < E)
                    {
#pragma unroll
                        for(int h = 0; h < VPC; ++h)
                        {
                            VP pr;
#pragma unroll
                            for(int i = 0; i < VN; ++i)
                                pr[i] = (MODE != 1) ? a[k][h][i] * xv[k][h * VN + i] : scalar * a[k][h][i] * xv[k][h * VN + i];
                            *reinterpret_cast<VP*>(sp + (g - cb) + h * VN) = pr;
                        }
                    }
                }
                // (one wave: its LDS operations complete in order; the barrier only keeps the compiler from moving reads up)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int lo = max(rs, cb), hi = min(re, cb + CH);
                for(int j = lo; j < hi; j += SB)
                {
                    T p[SB];
#pragma unroll
                    for(int e = 0; e < SB; ++e)
                        p[e] = sp[min(j + e, hi - 1) - cb];
#pragma unroll
                    for(int e = 0; e < SB; ++e)
                    {
                        const T t = sum + p[e];
                        sum       = (j + e < hi) ? t : sum; // (a select, not an addition of zero: -0 + +0 is +0)
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier(); // the next pass overwrites what this one read
            }
            if(row < nrow)
            {
                T xrow = (T)0;
                if((DOT && !ws.dotv) || MODE == 2)
                    xrow = x[row];
                if(MODE == 2)
                {
                    T t = (T)(-1) * sum + static_cast<const T*>(ws.jrhs)[row];
                    t   = static_cast<const T*>(ws.jdinv)[row] * t;
                    sum = xrow + scalar * t;
                }
                nt_store(sum, y + row);
                if(DOT)
                    dacc = (double)sum * (double)(ws.dotv ? static_cast<const T*>(ws.dotv)[row] : xrow);
            }
        }
    }
    if(DOT)
    {
        const double wsum = wave_reduce_sum(dacc);
        if((threadIdx.x & 63) == 0 && blk >= 0)
            ws.part1[blk * NWV + (threadIdx.x >> 6)] = wsum;
    }
}

// Measured and removed (round 4): k_csr_w4 with look-ahead -- the row offsets of a wave's four 16-row pieces requested at once,
// the first pass of piece k + 1 on its way while piece k is walked (two packet sets used alternately, pieces unrolled with
// compile-time numbers; 142 VGPRs, 3 waves per SIMD, or 128 with 4): 0.1655 / 0.1707 ms against 0.1604 ms of k_csr_w4 on the
// config-3 surrogate in alternating runs (gpurun_out/r04i; a first version that indexed the sets at run time kept them in
// scratch memory: 0.58 ms).  With k_csr_tr, k_csr_q4, the PIPE form and one wave per workgroup that makes SIX row walks at
// 0.155-0.17 ms: the wave's dependency chain is not what bounds this product either.

// ELL: one thread per row, column-major => every load is a perfectly coalesced wave access.
// STOP=true : ELL semantics (stop at the first negative column, host_matrix_ell.cpp:309-318)
// STOP=false: HYB-ELL semantics (skip invalid columns, host_matrix_hyb.cpp:344-352)
// PAT: the slot tuple of a row comes from the row-pattern dictionary (one byte per row instead of four per slot)
template <typename T, int MODE, bool STOP, bool DOT, bool PAT>
__global__ __launch_bounds__(kBlock) void k_ell(int nrow, int ncol, int width,
                                                const int* __restrict__ ecol,
                                                const T* __restrict__ eval,
                                                const T* __restrict__ x, T* __restrict__ y, T scalar,
                                                double* __restrict__ part1, const T* __restrict__ dotv,
                                                int nblk, int per_xcd, BandMap bm, CsrPattern pat)
{
    __shared__ int sdict[PAT ? kPatMax * kPatMaxW : 1];
    double dacc = 0.0;
    // one workgroup per 256 rows, XCD- and band-aware order (same mapping as the CSR kernel)
    const int     blk  = xcd_block(nblk, per_xcd, bm);
    const int64_t row  = (int64_t)blk * kCsrRows + threadIdx.x;
    const bool    live = blk >= 0 && row < nrow;
    // PAT: the dictionary is requested first and staged last; the row's pattern number, its y (ApplyAdd) and the values of
    // its first batch of slots -- none of which hangs on the dictionary -- are on their way before the barrier (round 4:
    // the chain of a workgroup was dictionary -> barrier -> values -> gathers)
    constexpr int NDW = PAT ? kPatMax * kPatMaxW / kBlock : 0;
    int           dreg[NDW > 0 ? NDW : 1];
    T             v0[kGatherW];
    int           pid  = 0;
    T             sum0 = (T)0;
    if(PAT)
    {
#pragma unroll
        for(int q = 0; q < NDW; ++q)
        {
            const int i = q * kBlock + threadIdx.x;
            dreg[q]     = i < pat.n * kPatMaxW ? pat.dict[i] : 0;
        }
        if(live)
        {
            pid = (int)pat.id[row];
            if(MODE == 1)
                sum0 = y[row];
#pragma unroll
            for(int e = 0; e < kGatherW; ++e)
                if(e < width)
                    v0[e] = nt_load(eval + (int64_t)e * nrow + row);
        }
#pragma unroll
        for(int q = 0; q < NDW; ++q)
        {
            const int i = q * kBlock + threadIdx.x;
            if(i < pat.n * kPatMaxW)
                sdict[i] = dreg[q];
        }
        __syncthreads();
    }
    if(live)
    {
        const int dbase = PAT ? pid * kPatMaxW : 0;
        T sum = PAT ? sum0 : (T)0;
        if(!PAT && MODE == 1)
            sum = y[row];
        // masked batches of kGatherW slots: the independent col/val loads of a batch first, then all its
        // gathers, then the products IN ORDER (a width-7 row is one batch, not 4 + three dependent steps)
        bool stopped = false;
        for(int el = 0; el < width && !stopped; el += kGatherW)
        {
            int c[kGatherW];
            T   v[kGatherW], xv[kGatherW];
#pragma unroll
            for(int e = 0; e < kGatherW; ++e)
            {
                c[e] = -1;
                if(el + e < width)
                {
                    if(PAT)
                    {
                        const int o = sdict[dbase + el + e];
                        c[e]        = o == kPatEnd ? -1 : (int)row + o;
                        v[e]        = el == 0 ? v0[e] : nt_load(eval + (int64_t)(el + e) * nrow + row);
                    }
                    else
                    {
                        c[e] = nt_load(ecol + (int64_t)(el + e) * nrow + row);
                        v[e] = nt_load(eval + (int64_t)(el + e) * nrow + row);
                    }
                }
            }
            bool use[kGatherW];
#pragma unroll
            for(int e = 0; e < kGatherW; ++e)
            {
                if(STOP) // ELL: everything after the first negative column is padding
                {
                    if(el + e < width && c[e] < 0)
                        stopped = true;
                    use[e] = (el + e < width) && !stopped;
                }
                else // HYB-ELL: skip invalid columns
                    use[e] = (el + e < width) && c[e] >= 0 && c[e] < ncol;
                if(use[e])
                    xv[e] = x[c[e]];
            }
#pragma unroll
            for(int e = 0; e < kGatherW; ++e)
                if(use[e])
                {
                    if(MODE == 0)
                        sum += v[e] * xv[e];
                    else
                        sum += scalar * v[e] * xv[e];
                }
        }
        nt_store(sum, y + row);
        if(DOT)
            dacc += (double)sum * (double)(dotv ? dotv[row] : x[row]);
    }
    if(DOT) // one partial per wave, summed in fixed order by a second tiny launch (as the CSR kernel)
    {
        const double wsum = wave_reduce_sum(dacc);
        if((threadIdx.x & 63) == 0 && blk >= 0)
            part1[blk * (kBlock / 64) + (threadIdx.x >> 6)] = wsum;
    }
}



// ELL, TWO ROWS PER THREAD (k_ell2; operators with an even row count of 2^16 rows and more, RAMD_ELL2=0/1: off / any even size;
// GW slots per batch: 8 with row patterns, 4 with the columns read).
// The reference's layout puts slot el of neighbouring rows side by side (ELL_IND = el * nrow + row), so a thread that owns
// rows 2t and 2t + 1 reads a slot's two values with ONE 16-byte access and its two columns with one 8-byte access, y and the
// pattern numbers in pairs as well: a third fewer memory instructions per row, and the streamed part of the product in
// 16-byte accesses (8-byte accesses stream at 0.54-0.70 of that rate, MI355X_MICROARCH.md).  The gathers stay per row.  Each
// row's products are added in slot order as in k_ell: y is bit-identical; the fused <x, y> sums two rows per lane before
// the wave reduction (one partial per 128 rows), i.e. differs from k_ell's in the last bits like any other summation order.
template <typename T, int MODE, bool STOP, bool DOT, bool PAT, int GW = kGatherW>
__global__ __launch_bounds__(kBlock) void k_ell2(int nrow, int ncol, int width, const int* __restrict__ ecol,
                                                 const T* __restrict__ eval, const T* __restrict__ x, T* __restrict__ y,
                                                 T scalar, double* __restrict__ part1, const T* __restrict__ dotv, int nblk,
                                                 int per_xcd, BandMap bm, CsrPattern pat)
{
    using P2 = T __attribute__((ext_vector_type(2)));
    using I2 = int __attribute__((ext_vector_type(2)));
    __shared__ int sdict[PAT ? kPatMax * kPatMaxW : 1];
    double         dacc = 0.0;
    const int      blk  = xcd_block(nblk, per_xcd, bm); // (a block = 2 * kBlock rows)
    const int64_t  r0   = ((int64_t)blk * kBlock + threadIdx.x) * 2; // rows r0 and r0 + 1 (nrow is even)
    const bool     live = blk >= 0 && r0 < nrow;
    constexpr int  NDW  = PAT ? kPatMax * kPatMaxW / kBlock : 0;
    int            dreg[NDW > 0 ? NDW : 1];
    P2             v0[GW];
    int            pidA = 0, pidB = 0;
    P2             sum  = {(T)0, (T)0};
    if(PAT)
    {
#pragma unroll
        for(int q = 0; q < NDW; ++q)
        {
            const int i = q * kBlock + threadIdx.x;
            dreg[q]     = i < pat.n * kPatMaxW ? pat.dict[i] : 0;
        }
    }
    if(live)
    {
        if(PAT)
        {
            const unsigned short pp = *reinterpret_cast<const unsigned short*>(pat.id + r0);
            pidA                    = pp & 0xff;
            pidB                    = pp >> 8;
        }
        if(MODE == 1)
            sum = *reinterpret_cast<const P2*>(y + r0);
#pragma unroll
        for(int e = 0; e < GW; ++e)
            if(e < width)
                v0[e] = nt_load(reinterpret_cast<const P2*>(eval + (int64_t)e * nrow + r0));
    }
    if(PAT)
    {
#pragma unroll
        for(int q = 0; q < NDW; ++q)
        {
            const int i = q * kBlock + threadIdx.x;
            if(i < pat.n * kPatMaxW)
                sdict[i] = dreg[q];
        }
        __syncthreads();
    }
    if(live)
    {
        T    sA = sum.x, sB = sum.y;
        bool stopA = false, stopB = false;
        for(int el = 0; el < width && !(stopA && stopB); el += GW)
        {
            int cA[GW], cB[GW];
            P2  v[GW];
            T   xA[GW], xB[GW];
#pragma unroll
            for(int e = 0; e < GW; ++e)
            {
                cA[e] = cB[e] = -1;
                if(el + e < width)
                {
                    if(PAT)
                    {
                        const int oA = sdict[pidA * kPatMaxW + el + e], oB = sdict[pidB * kPatMaxW + el + e];
                        cA[e]        = oA == kPatEnd ? -1 : (int)r0 + oA;
                        cB[e]        = oB == kPatEnd ? -1 : (int)r0 + 1 + oB;
                    }
                    else
                    {
                        const I2 cc = nt_load(reinterpret_cast<const I2*>(ecol + (int64_t)(el + e) * nrow + r0));
                        cA[e]       = cc.x;
                        cB[e]       = cc.y;
                    }
                    v[e] = el == 0 ? v0[e] : nt_load(reinterpret_cast<const P2*>(eval + (int64_t)(el + e) * nrow + r0));
                }
            }
            bool useA[GW], useB[GW];
#pragma unroll
            for(int e = 0; e < GW; ++e)
            {
                if(STOP) // ELL: everything after a row's first negative column is padding
                {
                    if(el + e < width && cA[e] < 0)
                        stopA = true;
                    if(el + e < width && cB[e] < 0)
                        stopB = true;
                    useA[e] = (el + e < width) && !stopA;
                    useB[e] = (el + e < width) && !stopB;
                }
                else // HYB-ELL: skip invalid columns
                {
                    useA[e] = (el + e < width) && cA[e] >= 0 && cA[e] < ncol;
                    useB[e] = (el + e < width) && cB[e] >= 0 && cB[e] < ncol;
                }
                if(useA[e])
                    xA[e] = x[cA[e]];
                if(useB[e])
                    xB[e] = x[cB[e]];
            }
#pragma unroll
            for(int e = 0; e < GW; ++e)
            {
                if(useA[e])
                    sA += (MODE == 0) ? v[e].x * xA[e] : scalar * v[e].x * xA[e];
                if(useB[e])
                    sB += (MODE == 0) ? v[e].y * xB[e] : scalar * v[e].y * xB[e];
            }
        }
        P2 out;
        out.x = sA;
        out.y = sB;
        nt_store(out, reinterpret_cast<P2*>(y + r0));
        if(DOT)
        {
            const P2 w2 = *reinterpret_cast<const P2*>((dotv ? dotv : x) + r0);
            dacc += (double)sA * (double)w2.x;
            dacc += (double)sB * (double)w2.y;
        }
    }
    if(DOT) // one partial per wave (128 rows), summed in fixed order by a second tiny launch
    {
        const double wsum = wave_reduce_sum(dacc);
        if((threadIdx.x & 63) == 0 && blk >= 0)
            part1[blk * (kBlock / 64) + (threadIdx.x >> 6)] = wsum;
    }
}

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

// ApplyAdd + correction of a dot product that already holds <p, y_old>: the touched rows add
// p_i * y_new_i - p_i * y_old_i (each product rounded as the full dot would round it)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_coo_grouped_dot(int ngroups, const int* __restrict__ grow,
                                                            const int* __restrict__ gptr,
                                                            const int* __restrict__ ccol,
                                                            const T* __restrict__ cval,
                                                            const T* __restrict__ x, T* __restrict__ y,
                                                            T scalar, const T* __restrict__ p,
                                                            ReduceCtx ctx, int slot)
{
    __shared__ double red[8];
    const int64_t     gsz  = (int64_t)gridDim.x * blockDim.x;
    double            dacc = 0.0;
    for(int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gsz)
    {
        const int row = grow[g];
        const T   old = y[row];
        T         sum = old;
        for(int i = gptr[g]; i < gptr[g + 1]; ++i)
            sum += scalar * cval[i] * x[ccol[i]];
        y[row] = sum;
        const double pr = (double)p[row];
        dacc += pr * (double)sum - pr * (double)old;
    }
    const double vals[1]  = {dacc};
    const int    slots[1] = {slot};
    const int    ops[1]   = {RED_ACC};
    grid_reduce_finish<1>(ctx, vals, slots, ops, red);
}

// ------------------------------------------------------------------------------------------
// sample ~2048 rows: the farthest column of a row, in rows.  A band is accepted when at least half
// of the samples agree on the same distance D, D is a multiple of the row-block size, and one "plane"
// of x (8 D bytes) is big enough to fall out of an XCD's L2 when two of them have to stay resident.
__global__ __launch_bounds__(kBlock) void k_band_sample(int nrow, int stride, const int* __restrict__ rp,
                                                        const int* __restrict__ ci, int* __restrict__ out)
{
    const int s   = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(((int64_t)s * stride + stride / 2) % nrow);
    int       far = 0;
    for(int j = rp[row]; j < rp[row + 1]; ++j)
        far = max(far, abs(ci[j] - row));
    out[s] = far;
}

int csr_analyse_band(ramd_mat_s* m)
{
    m->band_dist = 0;
    if(m->format != RAMD_CSR || m->nrow < (1 << 20) || m->nrow != m->ncol)
        return RAMD_OK;
    Backend&  b       = backend();
    const int samples = 2048;
    int*      d       = nullptr;
    RAMD_TRY(dev_alloc(&d, samples));
    hipLaunchKernelGGL(k_band_sample, dim3(samples / kBlock), dim3(kBlock), 0, b.cur, m->nrow,
                       std::max(1, m->nrow / samples), m->rp, m->ci, d);
    std::vector<int> h((size_t)samples);
    hipError_t       e = hipMemcpyAsync(h.data(), d, sizeof(int) * samples, hipMemcpyDeviceToHost, b.cur);
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&d);
    RAMD_HIP(e);
    std::sort(h.begin(), h.end());
    const int med = h[samples / 2];
    const int cnt = (int)(std::upper_bound(h.begin(), h.end(), med) - std::lower_bound(h.begin(), h.end(), med));
    if(cnt * 2 >= samples && med % kCsrRows == 0 && (int64_t)med * 8 >= (1 << 20) && med < m->nrow / 16)
        m->band_dist = med;
    return RAMD_OK;
}

// Do consecutive rows gather consecutive elements of x?  (A stencil on a lattice: row r + 1 has the columns of row r, each + 1.)
// Then a gather in which lane t serves row t -- the row walk of k_csr_tr -- touches a handful of lines per instruction, and
// that kernel beats the wave-private forms whose lanes serve the entries of a few rows (k_csr_w4 / k_csr_wp: counters on the
// 27-point operator at 256^3 show the data-return path of the L1 busy 95 % of the time, 63 accesses per gather instruction:
// 1.12 ms against 1.22 / 1.34).  2048 sampled rows; where most are shifts of their predecessor the row walk is taken.
__global__ __launch_bounds__(kBlock) void k_shift_sample(int nrow, int stride, const int* __restrict__ rp, const int* __restrict__ ci,
                                                         int* __restrict__ out)
{
    const int s   = blockIdx.x * blockDim.x + threadIdx.x;
    // (scattered rows, not every stride-th one: a stride that is a multiple of the lattice's line length samples one x only)
    const int row = (int)(((int64_t)s * stride + (int64_t)((unsigned)s * 2654435761u % (unsigned)stride)) % nrow);
    int       ok  = 0;
    if(row + 1 < nrow)
    {
        const int a = rp[row], b = rp[row + 1], e = rp[row + 2];
        ok          = (b - a == e - b) ? 1 : 0;
        for(int k = 0; ok && k < b - a; ++k)
            ok = ci[b + k] == ci[a + k] + 1 ? 1 : 0;
    }
    out[s] = ok;
}
int csr_analyse_shift(ramd_mat_s* m)
{
    m->shift_rows = 0;
    if(m->format != RAMD_CSR || m->nrow < 4096)
        return RAMD_OK;
    Backend&  b       = backend();
    const int samples = 2048;
    int*      d       = nullptr;
    RAMD_TRY(dev_alloc(&d, samples));
    hipLaunchKernelGGL(k_shift_sample, dim3(samples / kBlock), dim3(kBlock), 0, b.cur, m->nrow, std::max(1, m->nrow / samples), m->rp,
                       m->ci, d);
    std::vector<int> h((size_t)samples);
    hipError_t       e = hipMemcpyAsync(h.data(), d, sizeof(int) * samples, hipMemcpyDeviceToHost, b.cur);
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&d);
    RAMD_HIP(e);
    int cnt = 0;
    for(int v : h)
        cnt += v;
    m->shift_rows = cnt * 2 >= samples ? 1 : 0;
    return RAMD_OK;
}

// ------------------------------------------------------------------------------------------ row patterns
// Structured operators repeat a handful of rows: every interior row of a stencil has the same column offsets col - row.
// The analysis hashes (length, offsets) of every row into a 256-slot table, gives up beyond kPatMax distinct rows or rows
// longer than kPatMaxW, builds the dictionary from one representative row per slot and then VERIFIES every row against its
// dictionary entry entry by entry (a hash collision makes the matrix "not usable", never a wrong column).  The SpMV then
// reads one byte per row instead of four per entry: 9.9 instead of 13.9 GB per launch at 512^3, same values, same order.
constexpr int kPatTable = 256;
// where the columns of row r live: CSR rows, or the column-major slots of an ELL block (negative column = no entry)
struct PatCsr
{
    const int* rp;
    const int* ci;
    __device__ int len(int r) const
    {
        return rp[r + 1] - rp[r];
    }
    __device__ int off(int r, int k) const
    {
        return ci[rp[r] + k] - r;
    }
};
struct PatEll
{
    const int* ecol;
    int        nrow, width;
    __device__ int len(int) const
    {
        return width;
    }
    __device__ int off(int r, int k) const
    {
        const int c = ecol[(int64_t)k * nrow + r];
        return c < 0 ? kPatEnd : c - r;
    }
};
struct PatSell // slices of 64 rows, column-major inside the slice, width = slice length / 64
{
    const int* slice_off;
    const int* ecol;
    __device__ int len(int r) const
    {
        return (slice_off[(r >> 6) + 1] - slice_off[r >> 6]) >> 6;
    }
    __device__ int off(int r, int k) const
    {
        const int c = ecol[slice_off[r >> 6] + k * 64 + (r & 63)];
        return c < 0 ? kPatEnd : c - r;
    }
};
template <class Acc>
__device__ __forceinline__ unsigned long long pat_hash(const Acc& a, int r, int len)
{
    unsigned long long h = 0x9E3779B97F4A7C15ull ^ (unsigned long long)(unsigned)len;
    for(int k = 0; k < len; ++k)
    {
        const unsigned long long o = (unsigned long long)(unsigned)a.off(r, k);
        h ^= o + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xD6E8FEB86659FD93ull;
    }
    return h | 1ull; // (0 marks an empty slot)
}
// fail[0]: not usable; fail[1]: distinct rows so far (pass 1) / a row differs from its dictionary entry (pass 2)
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_pat_insert(int nrow, Acc a, unsigned long long* table, int* rep, int* fail)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += gsz)
    {
        // (an unstructured matrix is recognised after a few thousand rows: the rest of the sweep only looks at the flag --
        // one lane per wave: a load of ONE word by every row is served by one L2 channel, 0.5 s at 512^3)
        int stop = 0;
        if((threadIdx.x & 63) == 0)
            stop = __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(__shfl(stop, 0) != 0)
            return;
        const int len = a.len((int)r);
        if(len > kPatMaxW)
        {
            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const unsigned long long h = pat_hash(a, (int)r, len);
        int                      s = (int)(h % kPatTable);
        int                      probes = 0;
        for(; probes < kPatTable; ++probes, s = (s + 1) % kPatTable)
        {
            unsigned long long cur = __hip_atomic_load(table + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(cur == 0)
            {
                cur = atomicCAS(table + s, 0ull, h);
                if(cur == 0)
                {
                    rep[s] = (int)r;
                    if(atomicAdd(fail + 1, 1) >= kPatMax)
                        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            if(cur == h)
                break;
        }
        if(probes == kPatTable)
            __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the dictionary: entry p = the offsets of its representative row, kPatMaxW slots each
template <class Acc>
__global__ void k_pat_dict(int np, Acc a, const int* __restrict__ rows, int* __restrict__ dict, int* __restrict__ dlen)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= np)
        return;
    const int r = rows[p], len = a.len(r);
    dlen[p] = len;
    for(int k = 0; k < kPatMaxW; ++k)
        dict[p * kPatMaxW + k] = k < len ? a.off(r, k) : 0;
}
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_pat_assign(int nrow, Acc a, const unsigned long long* __restrict__ table,
                                                       const int* __restrict__ slot_id, const int* __restrict__ dict,
                                                       const int* __restrict__ dlen, unsigned char* __restrict__ id, int* fail)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += gsz)
    {
        const int                len = a.len((int)r);
        const unsigned long long h = pat_hash(a, (int)r, len);
        int                      s = (int)(h % kPatTable), probes = 0;
        while(probes < kPatTable && table[s] != h)
        {
            s = (s + 1) % kPatTable;
            ++probes;
        }
        bool ok = probes < kPatTable;
        int  p  = 0;
        if(ok)
        {
            p  = slot_id[s];
            ok = dlen[p] == len;
            for(int k = 0; ok && k < len; ++k)
                ok = dict[p * kPatMaxW + k] == a.off((int)r, k);
        }
        if(!ok)
            *fail = 1;
        id[r] = (unsigned char)p;
    }
}

template <class Acc>
static int analyse_pattern(int nrow, Acc acc, int* out_state, int* out_n, unsigned char** out_id, int** out_dict,
                           int* out_len_host = nullptr)
{
    *out_state = -1;
    Backend&            b = backend();
    unsigned long long* table = nullptr;
    int *               rep = nullptr, *flag = nullptr, *d_slot = nullptr, *d_len = nullptr, *d_rows = nullptr;
    auto                cleanup = [&]() {
        dev_free(&table);
        dev_free(&rep);
        dev_free(&flag);
        dev_free(&d_slot);
        dev_free(&d_len);
        dev_free(&d_rows);
    };
#define PAT_TRY(expr)          \
    do                         \
    {                          \
        const int s_ = (expr); \
        if(s_ != RAMD_OK)      \
        {                      \
            cleanup();         \
            return s_;         \
        }                      \
    } while(0)
#define PAT_HIP(expr)                                    \
    do                                                   \
    {                                                    \
        if((expr) != hipSuccess)                         \
        {                                                \
            cleanup();                                   \
            RAMD_FAIL(RAMD_ERR_HIP, "pattern analysis"); \
        }                                                \
    } while(0)
    PAT_TRY(dev_alloc(&table, kPatTable));
    PAT_TRY(dev_alloc(&rep, kPatTable));
    PAT_TRY(dev_alloc(&flag, 2));
    PAT_HIP(hipMemsetAsync(table, 0, sizeof(unsigned long long) * kPatTable, b.cur));
    PAT_HIP(hipMemsetAsync(rep, 0, sizeof(int) * kPatTable, b.cur));
    PAT_HIP(hipMemsetAsync(flag, 0, sizeof(int) * 2, b.cur));
    const int grid = ew_grid(nrow);
    hipLaunchKernelGGL((k_pat_insert<Acc>), dim3(grid), dim3(kBlock), 0, b.cur, nrow, acc, table, rep, flag);
    unsigned long long h_table[kPatTable];
    int                h_rep[kPatTable], h_flag[2] = {0, 0};
    PAT_HIP(hipMemcpyAsync(h_table, table, sizeof(h_table), hipMemcpyDeviceToHost, b.cur));
    PAT_HIP(hipMemcpyAsync(h_rep, rep, sizeof(h_rep), hipMemcpyDeviceToHost, b.cur));
    PAT_HIP(hipMemcpyAsync(h_flag, flag, sizeof(int) * 2, hipMemcpyDeviceToHost, b.cur));
    PAT_HIP(hipStreamSynchronize(b.cur));
    int np = 0, h_slot[kPatTable], h_rows[kPatMax];
    for(int s = 0; s < kPatTable; ++s)
    {
        h_slot[s] = -1;
        if(h_table[s] != 0)
        {
            if(np < kPatMax)
                h_rows[np] = h_rep[s];
            h_slot[s] = np++;
        }
    }
    if(h_flag[0] != 0 || np == 0 || np > kPatMax)
    {
        cleanup();
        return RAMD_OK; // not usable: stays -1
    }
    dev_free(out_id);
    dev_free(out_dict);
    PAT_TRY(dev_alloc(out_id, nrow));
    PAT_TRY(dev_alloc(out_dict, (int64_t)np * kPatMaxW));
    PAT_TRY(dev_alloc(&d_slot, kPatTable));
    PAT_TRY(dev_alloc(&d_len, np));
    PAT_TRY(dev_alloc(&d_rows, np));
    PAT_HIP(hipMemcpyAsync(d_slot, h_slot, sizeof(int) * kPatTable, hipMemcpyHostToDevice, b.cur));
    PAT_HIP(hipMemcpyAsync(d_rows, h_rows, sizeof(int) * (size_t)np, hipMemcpyHostToDevice, b.cur));
    PAT_HIP(hipMemsetAsync(flag + 1, 0, sizeof(int), b.cur));
    hipLaunchKernelGGL((k_pat_dict<Acc>), dim3(1), dim3(kPatMax), 0, b.cur, np, acc, d_rows, *out_dict, d_len);
    hipLaunchKernelGGL((k_pat_assign<Acc>), dim3(grid), dim3(kBlock), 0, b.cur, nrow, acc, table, d_slot, *out_dict, d_len,
                       *out_id, flag + 1);
    PAT_HIP(hipMemcpyAsync(h_flag, flag, sizeof(int) * 2, hipMemcpyDeviceToHost, b.cur));
    if(out_len_host)
        PAT_HIP(hipMemcpyAsync(out_len_host, d_len, sizeof(int) * (size_t)np, hipMemcpyDeviceToHost, b.cur));
    PAT_HIP(hipStreamSynchronize(b.cur)); // (also: the host arrays above were read by the copies)
    cleanup();
#undef PAT_TRY
#undef PAT_HIP
    if(h_flag[1] != 0) // two different rows behind one hash
    {
        dev_free(out_id);
        dev_free(out_dict);
        return RAMD_OK;
    }
    *out_n     = np;
    *out_state = 1;
    return RAMD_OK;
}

int csr_analyse_pattern(ramd_mat_s* m)
{
    m->pat_state = -1;
    if(m->format != RAMD_CSR || m->nrow <= 0 || m->nnz <= 0)
        return RAMD_OK;
    m->pat_w = kPatMaxW;
    m->xl_state = 0;
    return analyse_pattern(m->nrow, PatCsr{m->rp, m->ci}, &m->pat_state, &m->pat_n, &m->pat_id, &m->pat_dict, m->pat_len);
}

// ------------------------------------------------------------------------------------------ row groups
// same[i] = 1: row i has exactly the columns of row i-1 and does not start a 256-row block
__global__ __launch_bounds__(kBlock) void k_grp_same(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                     unsigned char* __restrict__ same)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        bool s = false;
        if(i % kCsrRows != 0)
        {
            const int a = rp[i - 1], b = rp[i], e = rp[i + 1];
            s           = (e - b) == (b - a) && e > b;
            for(int k = 0; s && k < e - b; ++k)
                s = ci[a + k] == ci[b + k];
        }
        same[i] = s ? 1 : 0;
    }
}
constexpr int kGrpRows = 8; // rows of a group at most (a longer run starts over)
__global__ __launch_bounds__(kBlock) void k_grp_lead(int nrow, const int* __restrict__ rp, const unsigned char* __restrict__ same,
                                                     int* __restrict__ lead, unsigned long long* __restrict__ followers)
{
    const int64_t      gsz = (int64_t)gridDim.x * blockDim.x;
    unsigned long long cnt = 0;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        int k = 0;
        while(k < kGrpRows && same[i - k]) // (same[] is 0 at the first row of every block: the walk stays inside)
            ++k;
        if(k == kGrpRows)
            k = 0; // deep inside a long run: on its own
        lead[i] = rp[i - k] - rp[i];
        cnt += k > 0 ? 1 : 0;
    }
    for(int off = 32; off > 0; off >>= 1)
        cnt += __shfl_xor(cnt, off, 64);
    if((threadIdx.x & 63) == 0 && cnt)
        atomicAdd(followers, cnt);
}
__global__ __launch_bounds__(kBlock) void k_grp_need(int nrow, const int* __restrict__ rp, const int* __restrict__ lead,
                                                     unsigned char* __restrict__ need)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        if(lead[i] == 0) // a leader (or a row on its own): its packets are read
            for(int q = rp[i] >> 2; q <= (rp[i + 1] - 1) >> 2 && rp[i + 1] > rp[i]; ++q)
                need[q] = 1;
}
int csr_analyse_groups(ramd_mat_s* m)
{
    m->grp_state = -1;
    if(m->format != RAMD_CSR || m->nrow <= 0 || m->nnz <= 0)
        return RAMD_OK;
    Backend&            b    = backend();
    unsigned char*      same = nullptr;
    unsigned long long* dcnt = nullptr;
    RAMD_TRY(dev_alloc(&same, m->nrow));
    int s = dev_alloc(&dcnt, 1);
    dev_free(&m->grp_lead);
    dev_free(&m->grp_need);
    if(s == RAMD_OK)
        s = dev_alloc(&m->grp_lead, m->nrow);
    const int64_t npk = (m->nnz + 3) / 4 + 1;
    if(s == RAMD_OK)
        s = dev_alloc(&m->grp_need, npk);
    unsigned long long followers = 0;
    if(s == RAMD_OK)
    {
        (void)hipMemsetAsync(dcnt, 0, sizeof(unsigned long long), b.cur);
        (void)hipMemsetAsync(m->grp_need, 0, (size_t)npk, b.cur);
        const int grid = ew_grid(m->nrow);
        hipLaunchKernelGGL(k_grp_same, dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci, same);
        hipLaunchKernelGGL(k_grp_lead, dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, same, m->grp_lead, dcnt);
        hipLaunchKernelGGL(k_grp_need, dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->grp_lead, m->grp_need);
        hipError_t e = hipMemcpyAsync(&followers, dcnt, sizeof(followers), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&same);
    dev_free(&dcnt);
    if(s != RAMD_OK)
    {
        dev_free(&m->grp_lead);
        dev_free(&m->grp_need);
        RAMD_FAIL(s, "row-group analysis");
    }
    // worth it when at least half of the rows ride on another row's columns
    if(followers * 2 >= (unsigned long long)m->nrow)
        m->grp_state = 1;
    else
    {
        dev_free(&m->grp_lead);
        dev_free(&m->grp_need);
    }
    return RAMD_OK;
}

// x tiles (k_csr_xl): clusters of the dictionary's distinct column offsets -> pieces of x a 256-row block stages in LDS
template <typename T>
static int csr_analyse_xl(ramd_mat_s* m)
{
    m->xl_state = -1;
    if(m->pat_state != 1 || m->pat_n <= 0 || m->nrow != m->ncol)
        return RAMD_OK;
    constexpr int EP = 16 / (int)sizeof(T); // elements per packet
    const int     np = m->pat_n;
    std::vector<int> dict((size_t)np * kPatMaxW);
    RAMD_HIP(hipMemcpy(dict.data(), m->pat_dict, sizeof(int) * dict.size(), hipMemcpyDeviceToHost));
    std::vector<int> offs;
    for(int p = 0; p < np; ++p)
        for(int k = 0; k < m->pat_len[p] && k < kPatMaxW; ++k)
            offs.push_back(dict[(size_t)p * kPatMaxW + k]);
    if(offs.empty())
        return RAMD_OK;
    std::sort(offs.begin(), offs.end());
    offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
    auto floor_to = [](int v, int q) { return v >= 0 ? v / q * q : -((-v + q - 1) / q * q); };
    XlSegs sg   = {};
    int    lo   = offs[0], hi = offs[0];
    int    used = 0;
    auto   close_seg = [&]() -> bool {
        if(sg.nseg >= kXlSegs)
            return false;
        const int al  = floor_to(lo, EP);
        const int len = hi - al + kCsrRows; // elements al .. hi + 255 relative to the block's first row
        const int npk = (len + EP - 1) / EP;
        if(used + npk * EP > kXlElems)
            return false;
        sg.omin[sg.nseg] = al;
        sg.npk[sg.nseg]  = npk;
        sg.base[sg.nseg] = used;
        used += npk * EP;
        ++sg.nseg;
        return true;
    };
    for(size_t i = 1; i < offs.size(); ++i)
    {
        if((int64_t)offs[i] - floor_to(lo, EP) > kCsrRows) // (a gap this wide: two pieces are cheaper than one)
        {
            if(!close_seg())
                return RAMD_OK;
            lo = offs[i];
        }
        hi = offs[i];
    }
    if(!close_seg())
        return RAMD_OK;
    sg.total = used;
    std::vector<int> xd((size_t)np * kPatMaxW, 0);
    for(int p = 0; p < np; ++p)
        for(int k = 0; k < m->pat_len[p] && k < kPatMaxW; ++k)
        {
            const int o = dict[(size_t)p * kPatMaxW + k];
            int       s = sg.nseg - 1;
            while(s > 0 && o < sg.omin[s])
                --s;
            xd[(size_t)p * kPatMaxW + k] = sg.base[s] + (o - sg.omin[s]);
        }
    dev_free(&m->xl_dict);
    RAMD_TRY(dev_alloc(&m->xl_dict, (int64_t)xd.size()));
    RAMD_HIP(hipMemcpy(m->xl_dict, xd.data(), sizeof(int) * xd.size(), hipMemcpyHostToDevice));
    static_assert(sizeof(XlSegs) <= sizeof(m->xl_segs), "ramd_mat_s::xl_segs holds an XlSegs");
    memcpy(m->xl_segs, &sg, sizeof(sg));
    m->xl_state = 1;
    return RAMD_OK;
}
// the ELL block of an ELL / HYB matrix: a pattern is the whole slot tuple, empty slots (col < 0) included
int ell_analyse_pattern(ramd_mat_s* m)
{
    m->pat_state = -1;
    if((m->format != RAMD_ELL && m->format != RAMD_HYB) || m->nrow <= 0 || m->ell_width <= 0 || m->ell_width > kPatMaxW)
        return RAMD_OK;
    m->pat_w = kPatMaxW;
    return analyse_pattern(m->nrow, PatEll{m->ell_col, m->nrow, m->ell_width}, &m->pat_state, &m->pat_n, &m->pat_id,
                           &m->pat_dict);
}
int sell_analyse_pattern(int nrow, const int* slice_off, const int* ecol, int* state, int* n, unsigned char** id, int** dict)
{
    *state = -1;
    if(nrow <= 0)
        return RAMD_OK;
    return analyse_pattern(nrow, PatSell{slice_off, ecol}, state, n, id, dict);
}

static BandMap band_map_for(const ramd_mat_s* m, int per_xcd);

template <typename T>
static int launch_csr(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar, bool dot, int slot,
                      const T* dotv = nullptr, const T* jdinv = nullptr, const T* jrhs = nullptr)
{
    Backend&  b = backend();
    // long rows (mean > 16 entries): four lanes per row (see k_csr_q4)
    static const int q4_env = getenv("RAMD_CSR_Q4") ? atoi(getenv("RAMD_CSR_Q4")) : -1; // (0 / 1: force, A/B experiments)
    const int        rows_per_wg = kCsrRows;
    // structured operators: columns from a row-pattern dictionary (csr_analyse_pattern); analysed once, on the first
    // product of a matrix with >= 2^20 entries (RAMD_CSR_PAT=1: every matrix, =0: never)
    static const int pat_env = getenv("RAMD_CSR_PAT") ? atoi(getenv("RAMD_CSR_PAT")) : -1;
    if(m->pat_state == 0 && pat_env != 0 && (pat_env > 0 || m->nnz >= (1 << 20)))
        RAMD_TRY(csr_analyse_pattern(const_cast<ramd_mat_s*>(m)));
    const bool       use_pat = pat_env != 0 && m->pat_state == 1 && !m->pat_off;
    // ... and, opt-in, their x tiles in LDS (RAMD_CSR_XL=1: k_csr_xl; default: the gather form k_csr_tr<PAT>)
    static const int xl_env = getenv("RAMD_CSR_XL") ? atoi(getenv("RAMD_CSR_XL")) : 0;
    if(use_pat && xl_env != 0 && m->xl_state == 0)
        RAMD_TRY(csr_analyse_xl<T>(const_cast<ramd_mat_s*>(m)));
    const bool       use_xl  = use_pat && xl_env != 0 && m->xl_state == 1;
    const CsrPattern pat     = {use_pat ? m->pat_id : nullptr, use_pat ? (use_xl ? m->xl_dict : m->pat_dict) : nullptr, m->pat_n,
                                m->pat_w};
    XlSegs           xsg     = {};
    if(use_xl)
        memcpy(&xsg, m->xl_segs, sizeof(xsg));
    // rows sharing the column list of their predecessor (FE matrices), opt-in: RAMD_CSR_GRP=1.  Measured on the af_shell10-class
    // matrix (profiles/r03_spmv_shell_variants.txt): 9.4 instead of 12.6 bytes per entry moved, and 0.235 ms instead of
    // 0.159 ms -- the product is bound by the dependent gather rounds of the row walk (five per 35-entry row, one row in four
    // lanes active per 2048-entry pass), not by bytes; the per-entry choice between the leader's staged columns and the
    // row's own costs more than the column packets saved.  Kept under the same bit-exact tests as an experiment.
    static const int grp_env = getenv("RAMD_CSR_GRP") ? atoi(getenv("RAMD_CSR_GRP")) : 0;
    if(!use_pat && grp_env > 0 && m->grp_state == 0 && m->pat_state != 1)
        RAMD_TRY(csr_analyse_groups(const_cast<ramd_mat_s*>(m)));
    const bool      use_grp = !use_pat && grp_env != 0 && m->grp_state == 1 && !m->pat_off;
    const CsrGroups cgr     = {use_grp ? m->grp_lead : nullptr, use_grp ? m->grp_need : nullptr};
    // two row blocks per workgroup with their memory phases overlapped (k_csr_pat2), where a row block fits one LDS pass
    static const int pat2_env = getenv("RAMD_CSR_PAT2") ? atoi(getenv("RAMD_CSR_PAT2")) : 1; // (0: k_csr_tr<PAT>; A/B experiments)
    int              pat_maxlen = 0;
    for(int p2 = 0; p2 < m->pat_n && p2 < 64; ++p2)
        pat_maxlen = m->pat_len[p2] > pat_maxlen ? m->pat_len[p2] : pat_maxlen;
    const bool use_pat2 = use_pat && !use_xl && pat2_env != 0 && pat_maxlen > 0 && kCsrRows * pat_maxlen + 3 <= kCsrChunk;
    // (measured without gain, 1.80-1.85 vs 1.73-1.92 ms plain and 2.03-2.10 vs 2.02-2.04 ms with the dot in alternating runs,
    //  gpurun_out/r03bq: 5 % fewer bytes do nothing for a product bound by its latency chain, the scan adds two barriers --
    //  opt-in: RAMD_CSR_NORP=1)
    static const int norp_env = getenv("RAMD_CSR_NORP") ? atoi(getenv("RAMD_CSR_NORP")) : 0;
    const bool       use_norp = use_pat2 && norp_env != 0;
    PatLens          plens    = {};
    for(int p2 = 0; p2 < m->pat_n && p2 < kPatMax; ++p2)
        plens.len[p2] = (unsigned char)m->pat_len[p2];
    // (RAMD_CSR_BLKRP=1: the compact block offsets for the general kernel too -- measured without effect there, 2.36-2.52 vs
    //  2.45-2.52 ms with the columns read at 512^3 and 0.157 vs 0.157 ms on the shell surrogate, gpurun_out/r03bg: that kernel
    //  runs at the stream rate of its bytes already)
    static const int blkrp_env = getenv("RAMD_CSR_BLKRP") ? atoi(getenv("RAMD_CSR_BLKRP")) : 0;
    // ... and the same structure with the columns read, for matrices of short rows (every row block fits one LDS pass)
    // (measured SLOWER than k_csr_tr, 2.52-2.55 vs 2.38-2.40 ms at 512^3 in alternating runs, gpurun_out/r03bm: with the columns
    //  read the product runs at the stream rate of its 14.2 GB and the extra registers only cost occupancy -- opt-in:
    //  RAMD_CSR_COL2=1; 2: also small matrices, for the tests)
    static const int col2_env = getenv("RAMD_CSR_COL2") ? atoi(getenv("RAMD_CSR_COL2")) : 0;
    const bool col2_candidate = !use_pat && !use_grp && !(q4_env > 0) && col2_env != 0 && (m->nrow >= (1 << 16) || col2_env == 2)
                                && m->nnz <= (int64_t)8 * m->nrow && m->blk_span >= 0;
    if((use_pat2 || use_xl || col2_candidate || (blkrp_env != 0 && !(!use_pat && q4_env > 0) && m->nrow >= (1 << 16))) && !m->blk_rp)
    {
        ramd_mat_s* mm  = const_cast<ramd_mat_s*>(m);
        const int   nb2 = (m->nrow + kCsrRows - 1) / kCsrRows;
        RAMD_TRY(dev_alloc(&mm->blk_rp, (int64_t)nb2 + 1));
        hipLaunchKernelGGL(k_blk_rp, dim3(ew_grid((int64_t)nb2 + 1)), dim3(kBlock), 0, b.cur, m->nrow, nb2, m->rp, mm->blk_rp);
    }
    if(col2_candidate && m->blk_span == 0) // measured once per matrix
    {
        ramd_mat_s* mm  = const_cast<ramd_mat_s*>(m);
        const int   nb2 = (m->nrow + kCsrRows - 1) / kCsrRows;
        int*        d   = nullptr;
        RAMD_TRY(dev_alloc(&d, 1));
        hipError_t e = hipMemsetAsync(d, 0, sizeof(int), b.cur);
        hipLaunchKernelGGL(k_blk_span_max, dim3(ew_grid(nb2)), dim3(kBlock), 0, b.cur, nb2, m->blk_rp, d);
        int span = 0;
        if(e == hipSuccess)
            e = hipMemcpyAsync(&span, d, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        dev_free(&d);
        RAMD_HIP(e);
        mm->blk_span = span > 0 ? span : -1;
    }
    const bool use_col2 = col2_candidate && m->blk_span > 0 && m->blk_span <= kCsrChunk;
    // (measured SLOWER on the config-3 surrogate: 0.173 against 0.154-0.159 ms in alternating runs, gpurun_out/r03cb -- opt-in)
    static const int pipe_env = getenv("RAMD_CSR_PIPE") ? atoi(getenv("RAMD_CSR_PIPE")) : 0;
    const bool use_pipe = pipe_env != 0;
    static const int w4_env = getenv("RAMD_CSR_W4") ? atoi(getenv("RAMD_CSR_W4")) : -1; // (0 / 1: force; default: rows of 16+ entries)
    const bool w4_rows = !use_pat && !use_grp && !(q4_env > 0) && !use_col2 && (int64_t)m->nnz >= (int64_t)16 * m->nrow;
    if(w4_rows && w4_env < 0 && m->shift_rows < 0)
        RAMD_TRY(csr_analyse_shift(const_cast<ramd_mat_s*>(m)));
    // (rows of 16+ entries take a wave-private form -- unless they are the rows of a stencil: see csr_analyse_shift)
    const bool use_w4  = !use_pat && !use_grp && !(q4_env > 0) && !use_col2
                         && (w4_env >= 0 ? w4_env != 0 : (w4_rows && m->shift_rows != 1));
    // rows of 16+ entries that walk their rows in k_csr_tr (stencils; row patterns of long rows): entries per LDS pass
    static const int lchunk_env = getenv("RAMD_CSR_LCHUNK") ? atoi(getenv("RAMD_CSR_LCHUNK")) : -1; // (2048 / 4096 / 8192; 4096 measured best on the 27-point operator: 1.13 ms against 1.66 ms at 8192 with row patterns)
    const bool long_rows  = (int64_t)m->nnz >= (int64_t)16 * m->nrow && !use_grp && !(q4_env > 0) && !use_col2 && !use_xl && !use_pat2;
    const int  long_chunk = !long_rows ? 0 : (lchunk_env >= 0 ? lchunk_env : 4096);
    // rows of 16+ entries that would walk their rows in k_csr_tr's 4096-entry passes: the wave-private row walk (k_csr_wr)
    static const int wr_env = getenv("RAMD_CSR_WR") ? atoi(getenv("RAMD_CSR_WR")) : 1; // (0: k_csr_tr in long chunks)
    const bool use_wr = wr_env != 0 && long_rows && long_chunk >= 4096 && !use_w4 && !use_pipe;
    static const int wr_gw = getenv("RAMD_CSR_WR_GW") ? atoi(getenv("RAMD_CSR_WR_GW")) : 14; // (gathers in flight per row: 8 / 14 / 28)
    auto wr_lds = [&](bool with_pat) -> size_t {
        return with_pat ? sizeof(T) * 4 * kWrCapOf<true> + sizeof(int) * (size_t)kPatMax * kPatMaxW
                        : (sizeof(T) + sizeof(int)) * 4 * kWrCapOf<false>;
    };
    if(use_wr && !m->wav_rp)
    {
        ramd_mat_s* mm   = const_cast<ramd_mat_s*>(m);
        const int   ngrp = (m->nrow + 63) / 64;
        RAMD_TRY(dev_alloc(&mm->wav_rp, (int64_t)ngrp + 1));
        hipLaunchKernelGGL(k_wav_rp, dim3(ew_grid((int64_t)ngrp + 1)), dim3(kBlock), 0, b.cur, m->nrow, ngrp, m->rp, mm->wav_rp);
    }
    if(use_wr)
    {
        // (more than 64 KB of LDS per workgroup: opt in once per instantiation)
        static bool raised = false;
        if(!raised)
        {
            const int big = (int)std::max(wr_lds(true), wr_lds(false));
#define WR_RAISE1(MODE, DOT, PATB, G) RAMD_HIP(hipFuncSetAttribute((const void*)k_csr_wr<T, MODE, DOT, PATB, G>, hipFuncAttributeMaxDynamicSharedMemorySize, big))
#define WR_RAISE(MODE, DOT, PATB) WR_RAISE1(MODE, DOT, PATB, 8); WR_RAISE1(MODE, DOT, PATB, 14); WR_RAISE1(MODE, DOT, PATB, 28)
            WR_RAISE(0, false, true); WR_RAISE(0, true, true); WR_RAISE(1, false, true); WR_RAISE(2, false, true);
            WR_RAISE(0, false, false); WR_RAISE(0, true, false); WR_RAISE(1, false, false); WR_RAISE(2, false, false);
#undef WR_RAISE
#undef WR_RAISE1
            raised = true;
        }
    }
    static const int w4_waves = getenv("RAMD_CSR_W4_WAVES") ? atoi(getenv("RAMD_CSR_W4_WAVES")) : 4; // (1: one wave per workgroup)
    static const int wp_env   = getenv("RAMD_CSR_WP") ? atoi(getenv("RAMD_CSR_WP")) : 1; // (0: k_csr_w4 where k_csr_wp would run)
    const bool q4      = !use_pat && q4_env > 0; // measured EQUAL to k_csr_tr on the shell surrogate (0.162 vs 0.160 ms): opt-in
    const int  nblk    = (m->nrow + rows_per_wg - 1) / rows_per_wg;
    const int  per_xcd = (nblk + 7) / 8;
    const int  grid    = per_xcd * 8;
    CsrDotWs   ws      = {};
    if(!q4 && m->band_dist < 0)
        RAMD_TRY(csr_analyse_band(const_cast<ramd_mat_s*>(m)));
    const BandMap bm = q4 ? BandMap{0, 0, 0} : band_map_for(m, per_xcd);
    if(dot)
    {
        ramd_mat_s* mm = const_cast<ramd_mat_s*>(m);
        if(!mm->dot_part1 || mm->dot_nblk != nblk)
        {
            dev_free(&mm->dot_part1);
            RAMD_TRY(dev_alloc(&mm->dot_part1, (int64_t)nblk * (kBlock / 64)));
            mm->dot_nblk = nblk;
        }
        ws.part1 = mm->dot_part1;
        ws.dotv  = dotv;
    }
    if(dot)
        prof_spmv_begin();
#define WR_LAUNCH(MODE, DOT, PATB)                                                                         \
    do                                                                                                     \
    {                                                                                                      \
        if(wr_gw == 8)                                                                                     \
            hipLaunchKernelGGL((k_csr_wr<T, MODE, DOT, PATB, 8>), dim3(grid), dim3(kBlock), wr_lds(PATB), b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, m->wav_rp);  \
        else if(wr_gw == 14)                                                                               \
            hipLaunchKernelGGL((k_csr_wr<T, MODE, DOT, PATB, 14>), dim3(grid), dim3(kBlock), wr_lds(PATB), b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, m->wav_rp);  \
        else                                                                                               \
            hipLaunchKernelGGL((k_csr_wr<T, MODE, DOT, PATB, 28>), dim3(grid), dim3(kBlock), wr_lds(PATB), b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, m->wav_rp);  \
    } while(0)
#define LAUNCH(MODE, DOT)                                                                                  \
    do                                                                                                     \
    {                                                                                                      \
        if(q4)                                                                                             \
            hipLaunchKernelGGL((k_csr_q4<T, MODE, DOT>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot);           \
        else if(use_xl)                                                                                    \
            hipLaunchKernelGGL((k_csr_xl<T, MODE, DOT>), dim3(grid), dim3(kBlock),                         \
                               sizeof(T) * (size_t)(kCsrChunk + xsg.total) + sizeof(int) * (size_t)(pat.n * pat.w), b.cur, \
                               m->nrow, nblk, per_xcd, \
                               m->rp, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, xsg, m->blk_rp); \
        else if(use_pat2 && use_norp)                                                                      \
            hipLaunchKernelGGL((k_csr_pat2<T, MODE, DOT, 2, true, true>), dim3(((per_xcd + 1) / 2) * 8), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, m->blk_rp, plens); \
        else if(use_pat2)                                                                                  \
            hipLaunchKernelGGL((k_csr_pat2<T, MODE, DOT, 2, true>), dim3(((per_xcd + 1) / 2) * 8), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, m->blk_rp); \
        else if(use_col2)                                                                                  \
            hipLaunchKernelGGL((k_csr_pat2<T, MODE, DOT, 2, false>), dim3(((per_xcd + 1) / 2) * 8), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, m->blk_rp); \
        else if(use_wr && use_pat)                                                                         \
            WR_LAUNCH(MODE, DOT, true);                                                                     \
        else if(use_wr)                                                                                    \
            WR_LAUNCH(MODE, DOT, false);                                                                    \
        else if(use_pat && long_chunk == 8192)                                                             \
            hipLaunchKernelGGL((k_csr_tr<T, MODE, DOT, true, false, false, 8192>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, CsrGroups{}, m->blk_rp);  \
        else if(use_pat && long_chunk == 4096)                                                             \
            hipLaunchKernelGGL((k_csr_tr<T, MODE, DOT, true, false, false, 4096>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, CsrGroups{}, m->blk_rp);  \
        else if(use_pat)                                                                                   \
            hipLaunchKernelGGL((k_csr_tr<T, MODE, DOT, true>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, CsrGroups{}, m->blk_rp);  \
        else if(use_grp)                                                                                   \
            hipLaunchKernelGGL((k_csr_tr<T, MODE, DOT, false, true>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, cgr, m->blk_rp); \
        else if(use_w4 && wp_env != 0 && w4_waves != 1)                                                    \
            hipLaunchKernelGGL((k_csr_wp<T, MODE, DOT, 4>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk,  \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm);        \
        else if(use_w4 && w4_waves == 1)                                                                   \
            hipLaunchKernelGGL((k_csr_w4<T, MODE, DOT, 1>), dim3(((nblk * 4 + 7) / 8) * 8), dim3(64), 0, b.cur, m->nrow,     \
                               nblk * 4, (nblk * 4 + 7) / 8, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot,        \
                               BandMap{0, 0, 0});                                                         \
        else if(use_w4)                                                                                    \
            hipLaunchKernelGGL((k_csr_w4<T, MODE, DOT, 4>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk,  \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm);        \
        else if(long_chunk >= 4096 && !use_w4)                                                             \
            hipLaunchKernelGGL((k_csr_tr<T, MODE, DOT, false, false, false, 4096>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, CsrGroups{}, m->blk_rp);  \
        else if(use_pipe)                                                                                  \
            hipLaunchKernelGGL((k_csr_tr<T, MODE, DOT, false, false, true>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, CsrGroups{}, m->blk_rp);  \
        else                                                                                               \
            hipLaunchKernelGGL((k_csr_tr<T, MODE, DOT, false>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, nblk, \
                               per_xcd, m->rp, m->ci, (const T*)m->val, x, y, scalar, ws, slot, bm, pat, CsrGroups{}, m->blk_rp);  \
    } while(0)
    ws.jdinv = jdinv;
    ws.jrhs  = jrhs;
    if(mode == 2)
        LAUNCH(2, false);
    else if(mode == 0 && !dot)
        LAUNCH(0, false);
    else if(mode == 0 && dot)
        LAUNCH(0, true);
    else
        LAUNCH(1, false);
#undef LAUNCH
#undef WR_LAUNCH
    RAMD_HIP(hipGetLastError());
    if(dot)
    {
        prof_spmv_end();
        return reduce_sum_to_slot(ws.part1, (int64_t)nblk * (kBlock / 64), slot);
    }
    return RAMD_OK;
}

static BandMap band_map_for(const ramd_mat_s* m, int per_xcd)
{
    BandMap bm = {0, 0, 0};
    if(m->band_dist > 0)
    {
        bm.P = m->band_dist / kCsrRows;
        bm.Z = per_xcd / bm.P;
        bm.W = 32;
        while(bm.W > 1 && bm.P % bm.W != 0)
            bm.W >>= 1;
        if(bm.Z < 3)
            bm.P = 0;
    }
    return bm;
}

template <typename T>
static int launch_ell(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar, bool stop, bool dot = false,
                      int slot = 0, const T* dotv = nullptr)
{
    Backend&      b       = backend();
    const int     nblk    = (m->nrow + kCsrRows - 1) / kCsrRows;
    const int     per_xcd = (nblk + 7) / 8;
    const int     grid    = per_xcd * 8;
    const BandMap bm      = band_map_for(m, per_xcd);
    double*       part1   = nullptr;
    if(dot) // Apply + <x,y>: per-wave partials in the matrix' workspace (shared with the CSR kernel)
    {
        ramd_mat_s* mm = const_cast<ramd_mat_s*>(m);
        if(!mm->dot_part1 || mm->dot_nblk != nblk)
        {
            dev_free(&mm->dot_part1);
            RAMD_TRY(dev_alloc(&mm->dot_part1, (int64_t)nblk * (kBlock / 64)));
            mm->dot_nblk = nblk;
        }
        part1 = mm->dot_part1;
    }
    // structured operators: the slot tuples from a row-pattern dictionary (see csr_analyse_pattern / launch_csr)
    static const int pat_env = getenv("RAMD_CSR_PAT") ? atoi(getenv("RAMD_CSR_PAT")) : -1;
    if(m->pat_state == 0 && pat_env != 0 && (pat_env > 0 || (int64_t)m->nrow * m->ell_width >= (1 << 20)))
        RAMD_TRY(ell_analyse_pattern(const_cast<ramd_mat_s*>(m)));
    const bool       use_pat = pat_env != 0 && m->pat_state == 1 && !m->pat_off;
    const CsrPattern pat     = {use_pat ? m->pat_id : nullptr, use_pat ? m->pat_dict : nullptr, m->pat_n, m->pat_w};
    // two rows per thread (k_ell2) where the pairs are aligned: an even number of rows
    static const int ell2_env = getenv("RAMD_ELL2") ? atoi(getenv("RAMD_ELL2")) : -1;
    // (with patterns eight slots per batch: 126 registers, 1.81-1.88 against 2.12-2.19 ms at 512^3, gpurun_out/r04u; eight slots
    //  per batch with the columns read need 206 registers -- 2 waves per SIMD -- and ran at 3.8-4.0 ms against k_ell's 2.4 ms)
    const bool       use2     = use_pat && (m->nrow % 2 == 0) && m->nrow > 0 && (ell2_env >= 0 ? ell2_env != 0 : m->nrow >= (1 << 16));
    // ... with the columns read: four slots per batch (76 registers, 6 waves per SIMD) -- 2.27-2.48 ms against k_ell's
    // 2.42-2.55 ms in alternating runs (gpurun_out/r04x): a few per cent, taken
    const bool       use2c    = !use_pat && (m->nrow % 2 == 0) && m->nrow > 0 && (ell2_env >= 0 ? ell2_env != 0 : m->nrow >= (1 << 16));
    const int        nblk2    = (m->nrow + 2 * kCsrRows - 1) / (2 * kCsrRows);
    const int        per_xcd2 = (nblk2 + 7) / 8;
    BandMap          bm2      = {0, 0, 0};
    if((use2 || use2c) && m->band_dist > 0 && m->band_dist % (2 * kCsrRows) == 0)
    {
        bm2.P = m->band_dist / (2 * kCsrRows);
        bm2.Z = per_xcd2 / bm2.P;
        bm2.W = 16;
        while(bm2.W > 1 && bm2.P % bm2.W != 0)
            bm2.W >>= 1;
        if(bm2.Z < 3)
            bm2.P = 0;
    }
#define LAUNCH(MODE, STOP, DOT)                                                                          \
    do                                                                                                   \
    {                                                                                                    \
        if(use2)                                                                                         \
            hipLaunchKernelGGL((k_ell2<T, MODE, STOP, DOT, true>), dim3(per_xcd2 * 8), dim3(kBlock), 0, b.cur, m->nrow, \
                               m->ncol, m->ell_width, m->ell_col, (const T*)m->ell_val, x, y, scalar, part1, \
                               dotv, nblk2, per_xcd2, bm2, pat);                                         \
        else if(use2c)                                                                                   \
            hipLaunchKernelGGL((k_ell2<T, MODE, STOP, DOT, false, 4>), dim3(per_xcd2 * 8), dim3(kBlock), 0, b.cur, m->nrow, \
                               m->ncol, m->ell_width, m->ell_col, (const T*)m->ell_val, x, y, scalar, part1, \
                               dotv, nblk2, per_xcd2, bm2, pat);                                         \
        else if(use_pat)                                                                                 \
            hipLaunchKernelGGL((k_ell<T, MODE, STOP, DOT, true>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, \
                               m->ncol, m->ell_width, m->ell_col, (const T*)m->ell_val, x, y, scalar, part1, \
                               dotv, nblk, per_xcd, bm, pat);                                            \
        else                                                                                             \
            hipLaunchKernelGGL((k_ell<T, MODE, STOP, DOT, false>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, \
                               m->ncol, m->ell_width, m->ell_col, (const T*)m->ell_val, x, y, scalar, part1, \
                               dotv, nblk, per_xcd, bm, pat);                                            \
    } while(0)
    if(dot && stop)
        LAUNCH(0, true, true);
    else if(dot)
        LAUNCH(0, false, true);
    else if(mode == 0 && stop)
        LAUNCH(0, true, false);
    else if(mode == 0)
        LAUNCH(0, false, false);
    else if(stop)
        LAUNCH(1, true, false);
    else
        LAUNCH(1, false, false);
#undef LAUNCH
    RAMD_HIP(hipGetLastError());
    if(dot)
        return reduce_sum_to_slot(part1, (int64_t)((use2 || use2c) ? nblk2 : nblk) * (kBlock / 64), slot);
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
int mat_apply_dot_impl(const ramd_mat_s* m, const T* x, T* y, int slot, const T* dotv)
{
    if(m->format == RAMD_CSR && m->nnz > 0 && m->nrow == m->ncol)
        return launch_csr<T>(m, x, y, 0, (T)1, true, slot, dotv); // bracketed inside (SpMV kernel only)
    if((m->format == RAMD_ELL || m->format == RAMD_HYB) && m->ell_width > 0 && m->nrow == m->ncol && m->nrow > 0)
    {
        prof_spmv_begin();
        int s = launch_ell<T>(m, x, y, 0, (T)1, m->format == RAMD_ELL, true, slot, dotv);
        prof_spmv_end();
        RAMD_TRY(s);
        if(m->format == RAMD_HYB && m->coo_nnz > 0) // tail: y += COO x, the dot corrected on the touched rows
        {
            Backend&  b    = backend();
            const int grid = reduce_grid(m->coo_ngroups);
            hipLaunchKernelGGL((k_coo_grouped_dot<T>), dim3(grid), dim3(kBlock), 0, b.cur, m->coo_ngroups,
                               m->coo_grow, m->coo_gptr, m->coo_col, (const T*)m->coo_val, x, y, (T)1,
                               dotv ? dotv : x, reduce_ctx(), slot);
            RAMD_HIP(hipGetLastError());
        }
        return RAMD_OK;
    }
    return RAMD_ERR_UNSUPPORTED; // caller falls back to apply + dot (two launches)
}
// y += scalar A x ; slot (holding <p, y_old>) corrected to <p, y>: grouped COO only
template <typename T>
int mat_apply_add_dot_impl(const ramd_mat_s* m, const T* x, T* y, T scalar, const T* p, int slot)
{
    if(m->format != RAMD_COO || !m->coo_gptr)
        return RAMD_ERR_UNSUPPORTED;
    if(m->coo_nnz <= 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = reduce_grid(m->coo_ngroups);
    hipLaunchKernelGGL((k_coo_grouped_dot<T>), dim3(grid), dim3(kBlock), 0, b.cur, m->coo_ngroups, m->coo_grow,
                       m->coo_gptr, m->coo_col, (const T*)m->coo_val, x, y, scalar, p, reduce_ctx(), slot);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}
template int mat_apply_add_dot_impl<double>(const ramd_mat_s*, const double*, double*, double, const double*, int);
template int mat_apply_add_dot_impl<float>(const ramd_mat_s*, const float*, float*, float, const float*, int);
// one damped-Jacobi sweep xnew = x + omega * dinv * (rhs - A x), CSR only
template <typename T>
int mat_jacobi_sweep_impl(const ramd_mat_s* m, const T* dinv, const T* rhs, const T* x, T* xnew, T omega)
{
    if(m->format != RAMD_CSR || m->nnz <= 0 || m->nrow != m->ncol)
        return RAMD_ERR_UNSUPPORTED;
    prof_spmv_begin();
    int s = launch_csr<T>(m, x, xnew, 2, omega, false, 0, nullptr, dinv, rhs);
    prof_spmv_end();
    return s;
}
template int mat_jacobi_sweep_impl<double>(const ramd_mat_s*, const double*, const double*, const double*, double*, double);
template int mat_jacobi_sweep_impl<float>(const ramd_mat_s*, const float*, const float*, const float*, float*, float);
template int mat_apply_dot_impl<double>(const ramd_mat_s*, const double*, double*, int, const double*);
template int mat_apply_dot_impl<float>(const ramd_mat_s*, const float*, float*, int, const float*);

} // namespace ramd
