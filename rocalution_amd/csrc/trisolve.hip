// trisolve.hip -- ILU(0) factorisation and sparse triangular solves on gfx950, sync-free.
//
// Replaces the rocSPARSE csrilu0 / csrsv_analysis / csrsv_solve calls of the reference HIP backend
// (src/base/hip/hip_matrix_csr.cpp:1295-1359 ILU0Factorize, :1594-1698 LUAnalyse, :1756-1821 LUSolve,
//  :2566-2870 L/U Analyse+Solve).  Arithmetic follows the HOST backend row by row:
//   ILU0Factorize  src/base/host/host_matrix_csr.cpp:2096-2171 (IKJ, ascending k per entry)
//   LUSolve        :1163-1221,  LSolve :1357-1404,  USolve :1420-1466
// (each row subtracts its dependencies in ascending column order, then divides by the stored
//  diagonal), so factors and solutions are bit-identical to the OpenMP backend.
//
// Design (MI355X): a triangular solve is a DAG; one kernel launch walks it without grid barriers.
//   * analysis (once): dependency LEVELS of every row are computed on the device, rows are grouped
//     by level into a position order, and the strictly-triangular part is re-packed in that order as
//     wave-sliced ELL (64 positions per slice, column-major inside the slice) -> every matrix load
//     of the solve is a fully coalesced wave access, and rows sharing a wave are (almost always)
//     independent.
//   * solve: thread t owns position t.  Workgroups take a ticket (so a running workgroup only ever
//     waits on positions owned by workgroups that already started), and every dependency is
//     consumed by polling the solution array itself: it is pre-filled with a NaN sentinel and each
//     finished value is published with ONE 8-byte (4-byte for fp32) agent-scope store -- the
//     "data-tagged granule" hand-off of MI355X_MICROARCH.md, one L2 round trip per DAG edge,
//     no flags, no fences.
#include "device_utils.hpp"
#include "matrix_impl.hpp"
#include "trsv_handoff.hpp"
#include "trsv_lattice.hpp"
#include "trsv_syncfree.hpp"
#include "trsv_box27.hpp"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace ramd
{

__device__ __forceinline__ bool lane_bit(unsigned long long mask, int b) // (b wave-uniform: 32-bit shifts)
{
    const unsigned half = b < 32 ? (unsigned)mask : (unsigned)(mask >> 32);
    return (half >> (b & 31)) & 1u;
}

// workgroup size of the sweeps that take their rows in 64-row wave units (k_levels, k_ct_glevels, k_ct_coords, k_ilu0_rows):
// one ticket per workgroup -- ~10 ns per atomic on the one counter, 1.3 ms for the 131 072 workgroups of 512^3 (a ticket per
// wave: 21 ms, per 256 rows: 5 ms, both more than the level sweep itself).  The 16 waves of workgroup k take the units
// order[16 k .. 16 k + 15]: every unit a unit waits for sits at an earlier place of that order, i.e. in a workgroup that
// started earlier or in this one.
constexpr int kSweepBlock = 1024;

// ---------------------------------------------------------------- levels (natural order, once)
// level[i] = 1 + max(level[dep]); 0 means "not computed yet" and doubles as the poll flag.
// LOWER: deps are columns < i, rows taken in ascending order; else columns > i, descending order.
//
// A wave holds 64 consecutive rows of the sweep, so a dependency is either a row of an EARLIER wave (polled in memory) or a
// row of a lower lane of this wave.  The two kinds are separated: (A) every lane consumes its out-of-wave dependencies and
// notes the in-wave ones in a 64-bit lane mask; (B) the in-wave part -- a longest path over the lanes in ascending order --
// is resolved in registers: at step b lane b is final (its in-wave dependencies are lower lanes), its value is broadcast
// with one v_readlane and the lanes that hold bit b take it.  A chain link inside the wave costs a handful of ALU
// instructions instead of a trip through the L2 (~0.4 us: the i-1 chain of a 512-row grid line used to take ~100 us per
// 256-row block, 0.11 s per sweep at 512^3); the wave publishes its 64 levels together.
template <bool LOWER>
__global__ __launch_bounds__(kSweepBlock) void k_levels(int nrow, const int* __restrict__ rp,
                                                   const int* __restrict__ ci, int* level,
                                                   unsigned* counter, unsigned base,
                                                   UnitView uv, int poll_cap)
{
    const unsigned slot = take_ticket(counter, base) * (blockDim.x >> 6) + (threadIdx.x >> 6); // (one ticket per workgroup)
    const int      lane = threadIdx.x & 63;
    const bool     have = slot < (unsigned)uv.nunits; // unit_schedule: the rows of my unit
    const int      unit = have ? (uv.order ? uv.order[slot] : (int)slot) : 0;
    const int      ufirst = have ? uv.ustart[unit] : 0;
    const int64_t  t    = (int64_t)ufirst + lane;
    const bool     live = have && t < uv.ustart[unit + 1];
    const int      i    = live ? (LOWER ? (int)t : (int)(nrow - 1 - t)) : 0;
    const int      i0   = LOWER ? i - lane : i + lane; // the row of lane 0
    const int      dir  = LOWER ? 1 : -1;
    int            j    = live ? (LOWER ? rp[i] : rp[i + 1] - 1) : 0;
    const int      end  = live ? (LOWER ? rp[i + 1] : rp[i] - 1) : 0; // one past the last entry in scan direction
    int            lev  = 0;
    unsigned long long inwave = 0ull;
    // (A) SIMT hazard: a lane that has LEFT a loop cannot execute anything until the whole wave leaves it, so the loop
    // exit is made wave-uniform with a ballot
    bool fin     = !live;
    int  spins   = 0;
    int  backoff = 1;
    bool stalled = false; // nobody of the wave advanced last turn
    do
    {
        spin_guard(spins);
        const int  j_start = j;
        const bool was_fin = fin;
        const bool my_turn = !stalled || lane == (int)__ffsll((long long)__ballot(!fin)) - 1; // (see poll_turn)
        if(!fin && my_turn)
        {
            while(j != end)
            {
                const int c = ci[j];
                if(LOWER ? (c >= i) : (c <= i))
                {
                    j = end; // sorted rows: no dependency follows in scan direction
                    continue;
                }
                const int rel = LOWER ? c - i0 : i0 - c; // lane that owns row c, if it is one of mine
                if(rel >= 0)
                {
                    inwave |= 1ull << rel; // (rel < lane)
                    j += dir;
                    continue;
                }
                const int lc = __hip_atomic_load(level + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if(lc == 0)
                    break;
                lev = max(lev, lc);
                j += dir;
            }
            fin = (j == end);
        }
        // nobody in the wave advanced (all waiting on other waves): back off instead of flooding the L2 with polls
        const bool advanced = __ballot(!was_fin && (fin || j != j_start)) != 0ull;
        stalled             = !advanced;
        backoff             = poll_backoff(advanced, backoff, poll_cap);
    } while(__ballot(!fin) != 0ull);
    // (B)
    if(__ballot(inwave != 0ull) != 0ull)
        for(int b = 0; b < 63; ++b)
        {
            const int lb = __builtin_amdgcn_readlane(lev, b) + 1;
            if(lane_bit(inwave, b))
                lev = max(lev, lb);
        }
    if(live)
        __hip_atomic_store(level + i, lev + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kBlock) void k_invert_perm(int n, const int* __restrict__ order,
                                                        int* __restrict__ pos)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        pos[order[t]] = (int)t;
}

// ---------------------------------------------------------------- sliced-ELL packing
// per position: number of strictly-triangular entries; per 64-slice: max -> width
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_tri_width(int nrow, const int* __restrict__ rp,
                                                      const int* __restrict__ ci,
                                                      const int* __restrict__ order,
                                                      int* __restrict__ slice_w)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int           c = 0;
    if(t < nrow)
    {
        const int i = order[t];
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(LOWER ? (ci[j] < i) : (ci[j] > i))
                ++c;
    }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
        c = max(c, __shfl_xor(c, off, 64));
    if((threadIdx.x & 63) == 0 && t < nrow + 64)
    {
        const int64_t s = t >> 6;
        if(s * 64 < nrow)
            slice_w[s] = c * 64; // entries occupied by the slice
    }
}

// fill: dependency positions + values in ORIGINAL (ascending column) order; diagonal value aside.
// host LUSolve / USolve locate the diagonal by equality scan and otherwise reuse the previous
// position (host_matrix_csr.cpp:1199-1218); a missing diagonal is reported by analysis instead.
template <typename T, bool LOWER>
__global__ __launch_bounds__(kBlock) void k_tri_fill(int nrow, const int* __restrict__ rp,
                                                     const int* __restrict__ ci,
                                                     const T* __restrict__ val,
                                                     const int* __restrict__ order,
                                                     const int* __restrict__ pos,
                                                     const int* __restrict__ slice_off,
                                                     int* __restrict__ ecol, T* __restrict__ eval,
                                                     T* __restrict__ diag, int* __restrict__ nodiag,
                                                     int reverse)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nrow)
        return;
    const int     i    = order[t];
    const int64_t s    = t >> 6;
    const int     lane = (int)(t & 63);
    const int     base = slice_off[s];
    const int     w    = (slice_off[s + 1] - base) >> 6;
    int           k    = 0;
    bool          have = false;
    const int     rs = rp[i], re = rp[i + 1];
    for(int q = rs; q < re; ++q)
    {
        const int j = reverse ? (re - 1 - (q - rs)) : q; // reverse: entries in DESCENDING storage order
        const int c = ci[j];
        if(LOWER ? (c < i) : (c > i))
        {
            ecol[base + k * 64 + lane] = pos[c];
            eval[base + k * 64 + lane] = val[j];
            ++k;
        }
        else if(c == i)
        {
            diag[t] = val[j];
            have    = true;
        }
    }
    for(; k < w; ++k)
    {
        ecol[base + k * 64 + lane] = -1;
        eval[base + k * 64 + lane] = (T)0;
    }
    if(!have)
    {
        diag[t] = (T)1;
        *nodiag = 1;
    }
}

// ---------------------------------------------------------------- the solve kernel
//   sum = rhs ; for each dependency (ascending original column): sum -= val * w[dep] ; [sum /= diag]
// rhs  = rhs_src[rhs_idx[t]]          (gather: natural-order input, or the L stage's positions)
// w    = sentinel-initialised scratch in position order (polled + published)
// out  = optional natural-order output, out[order[t]] = sum
// DMODE 0: unit diagonal   1: sum /= diag (stored diagonal)   2: sum *= diag (an inverse diagonal: LLSolve)
template <typename T, int DMODE>
__global__ __launch_bounds__(kBlock) void k_trsv(int nrow, const int* __restrict__ slice_off,
                                                 const int* __restrict__ ecol,
                                                 const T* __restrict__ eval,
                                                 const T* __restrict__ diag,
                                                 const T* __restrict__ rhs_src,
                                                 const int* __restrict__ rhs_idx, T* w,
                                                 T* __restrict__ out, const int* __restrict__ order,
                                                 unsigned* counter, unsigned base, int sleep_cycles)
{
    extern __shared__ __attribute__((aligned(16))) char occupancy_pad[]; // launch-time occupancy limiter
    (void)occupancy_pad;
    const unsigned blk = take_ticket(counter, base);
    const int64_t  t   = (int64_t)blk * kBlock + threadIdx.x;
    if(t >= nrow)
        return;
    const int64_t s    = t >> 6;
    const int     lane = (int)(t & 63);
    const int     b0   = slice_off[s];
    const int     wd   = (slice_off[s + 1] - b0) >> 6;
    T             sum  = rhs_src[rhs_idx[t]];
    const T       dg   = (DMODE == 0) ? (T)1 : diag[t]; // fetched before the wait, not after it
    const int     onat = out ? order[t] : 0;
    // A row's dependencies mostly sit in the previous level and become ready together, so they are
    // polled TOGETHER (one L2 round trip per attempt, not one per dependency) and the row's entries are
    // fetched up front in chunks of kDepChunk; the subtraction itself still runs in storage order.
    constexpr int kDepChunk = 8;
    int           c[kDepChunk];
    T             a[kDepChunk];
    auto          load_chunk = [&](int k0) {
#pragma unroll
        for(int e = 0; e < kDepChunk; ++e)
        {
            const int k = k0 + e;
            c[e]        = (k < wd) ? nt_load(ecol + b0 + k * 64 + lane) : -1;
            a[e]        = (k < wd) ? nt_load(eval + b0 + k * 64 + lane) : (T)0;
        }
    };
    int k0 = 0;
    load_chunk(0);
    // publish inside a wave-uniform-exit loop (see k_levels): small levels can put dependent rows
    // into one wave
    bool fin   = false;
    int  spins = 0;
    do
    {
        spin_guard(spins);
        bool progress = true;
        if(!fin)
        {
            // (Round 6 tried to issue every request of the attempt before the first answer is looked at, as the sync-free grouped
            //  form now does -- a slot without an entry asking for the row's own position: min 69 instead of 86 ms per triangle
            //  on the RCM shell, but single launches of SECONDS: 64 x 8 requests per attempt from every waiting wave of a grid
            //  that is as large as the matrix starve the few rows that can advance.  The test per request stays.)
            typename Sentinel<T>::bits bits[kDepChunk];
            bool                       all = true;
#pragma unroll
            for(int e = 0; e < kDepChunk; ++e)
                if(c[e] >= 0)
                {
                    bits[e] = poll_load(w + c[e]);
                    all     = all && (bits[e] != Sentinel<T>::value);
                }
            if(all)
            {
#pragma unroll
                for(int e = 0; e < kDepChunk; ++e)
                    if(c[e] >= 0)
                        sum -= a[e] * Sentinel<T>::from_bits(bits[e]);
                k0 += kDepChunk;
                if(k0 < wd)
                    load_chunk(k0);
                else
                {
                    if(DMODE == 1)
                        sum /= dg;
                    else if(DMODE == 2)
                        sum = sum * dg;
                    publish(w + t, sum);
                    if(out)
                        out[onat] = sum;
                    fin = true;
                }
            }
            else
                progress = false;
        }
        // nobody in the wave could advance: back off instead of hammering the L2 with polls
        if(__ballot(progress && !fin) == 0ull && __ballot(!fin) != 0ull)
            for(int z = 0; z < sleep_cycles; ++z)
                __builtin_amdgcn_s_sleep(1);
    } while(__ballot(!fin) != 0ull);
}

// (The band form of round 5 -- k_trsv_band: ONE workgroup walking the levels of a deep, narrow dependency graph behind workgroup
//  barriers, the last 16 384 values in an LDS window, the row data requested a round ahead by hand-counted loads -- stood here:
//  46 ms per triangle on the RCM-numbered shell against 95 ms of the level-scheduled rows, bound by the load path of its one CU.
//  The sync-free grouped form further down (k_trsv_sf) solves the same triangles in 3.4 / 4.7 ms and is asked first for every
//  matrix the band form took; it was removed with its forced parity test at the end of the round.  DESIGN.md section 9.1a.)

// ---------------------------------------------------------------- ILU(0), natural order, sync-free
// host_matrix_csr.cpp:2096-2171.  Thread per row; a row waits for every pivot row k < i of its
// pattern (flag done[k]), scales a_ik, and subtracts a_ik * a_kj from its own entries that exist
// (sorted merge instead of the host's scatter map; same entries, same ascending-k order).
// Finished rows publish their values with agent-scope stores and then raise done[i].
template <typename T>
__global__ __launch_bounds__(kBlock) void k_ilu0(int nrow, const int* __restrict__ rp,
                                                 const int* __restrict__ ci, T* val, int* done,
                                                 int* diag_pos, unsigned* counter,
                                                 unsigned base, const int* __restrict__ order)
{
    using B            = typename Sentinel<T>::bits;
    const unsigned blk = take_ticket(counter, base);
    const int64_t  t   = (int64_t)blk * kBlock + threadIdx.x;
    if(t >= nrow)
        return;
    // rows are taken in (level, row) order: the rows of a wave are (almost always) independent and every
    // pivot row belongs to an earlier level, i.e. an earlier or the same workgroup ticket
    const int i   = order[t];
    const int rs  = rp[i];
    const int re  = rp[i + 1];
    int       j   = rs;
    // position of the first entry with col >= i ("diag_offset", host :2162)
    int dj = rs;
    while(dj < re && ci[dj] < i)
        ++dj;
    bool fin = false; // wave-uniform loop exit, see k_levels
    int  spins = 0;
    int  backoff = 1;
    do
    {
        spin_guard(spins);
        const int  j_before   = j;
        const bool fin_before = fin;
        if(!fin)
        {
            if(j < dj)
            {
                const int k = ci[j];
                if(__hip_atomic_load(done + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                {
                    const int kd
                        = __hip_atomic_load(diag_pos + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int kre   = rp[k + 1];
                    const T   pivot = Sentinel<T>::from_bits(
                        __hip_atomic_load(reinterpret_cast<const B*>(val + kd), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT));
                    if(pivot != (T)0)
                    {
                        const T f = val[j] / pivot;
                        val[j]    = f;
                        int m     = j + 1; // own entries right of (i,k), ascending
                        for(int q = kd + 1; q < kre; ++q)
                        {
                            const int cq = ci[q];
                            while(m < re && ci[m] < cq)
                                ++m;
                            if(m >= re)
                                break;
                            if(ci[m] == cq)
                            {
                                const T akq = Sentinel<T>::from_bits(__hip_atomic_load(
                                    reinterpret_cast<const B*>(val + q), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT));
                                val[m] -= f * akq;
                            }
                        }
                    }
                    ++j;
                }
            }
            else
            {
                // publish the finished upper part (incl. diagonal) write-through, then the flag
                for(int q = dj; q < re; ++q)
                    __hip_atomic_store(reinterpret_cast<B*>(val + q), Sentinel<T>::as_bits(val[q]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(diag_pos + i, dj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(done + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fin = true;
            }
        }
        // nobody in the wave advanced: back off (exponentially) instead of flooding the fabric with polls -- without
        // it a few thousand resident waves polling at full rate can slow the producers down by orders of magnitude
        backoff = poll_backoff(__ballot(!fin_before && (fin || j != j_before)) != 0ull, backoff);
    } while(__ballot(!fin) != 0ull);
}

// ... the same factorisation for rows of at most W entries (W = 8: 5- / 7-point operators; W = 16: 9-point ones), in NATURAL order (blocks in hyperplane order, blocksched.hip):
// a lane keeps its whole row in registers; pivot rows of earlier waves are read from memory as above, pivot rows held by a
// lower lane of the same wave -- they are the LAST pivots of the row, the columns being sorted -- are taken from that
// lane's registers, lane by lane in ascending order (at step b lane b is final; see k_levels).  The i-1 chain of a grid
// line never leaves the wave: a link costs ~200 ALU instructions instead of a flag, a pivot and a row fetched through the
// L2, and the sweep reads and writes the matrix in storage order (the level order above touches ~25 scattered 64-byte
// lines per row: 0.26 s at 512^3).  Same operations per entry in the same ascending-pivot order: bit-identical factors.
constexpr int kIluWMax = 16;

template <typename T>
__device__ __forceinline__ T bcast_lane(T v, int lane_id);
template <>
__device__ __forceinline__ float bcast_lane<float>(float v, int lane_id)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_id));
}
template <>
__device__ __forceinline__ double bcast_lane<double>(double v, int lane_id)
{
    const long long bits = __double_as_longlong(v);
    const int       lo   = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), lane_id);
    const int       hi   = __builtin_amdgcn_readlane((int)(bits >> 32), lane_id);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

template <typename T, int BLOCK, int kIluW>
__global__ __launch_bounds__(BLOCK) void k_ilu0_rows(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      T* val, int* done, int* diag_pos, unsigned* counter, unsigned base,
                                                      UnitView uv, int poll_cap)
{
    using B             = typename Sentinel<T>::bits;
    constexpr int  NO   = kIluW - 1; // entries besides the pivot entry
    const unsigned slot = take_ticket(counter, base) * (BLOCK / 64) + (threadIdx.x >> 6); // (one ticket per workgroup)
    const int      lane = threadIdx.x & 63;
    const bool     have = slot < (unsigned)uv.nunits; // unit_schedule: the rows of my unit
    const int      unit = have ? (uv.order ? uv.order[slot] : (int)slot) : 0;
    const int      ufirst = have ? uv.ustart[unit] : 0;
    const int64_t  t    = (int64_t)ufirst + lane;
    const bool     live = have && t < uv.ustart[unit + 1];
    const int      i    = live ? (int)t : 0;
    const int      i0   = (int)t - lane; // the row of lane 0
    const int      rs   = live ? rp[i] : 0;
    const int      len  = live ? rp[i + 1] - rs : 0; // <= kIluW (caller)
    // the row in registers: the pivot entry -- the first one at or right of the diagonal ("diag_offset", host :2162) -- apart
    // (dcol, dg), the other entries in storage order (oc, ov): [0, dp) left of it, [dp, len - 1) right of it
    int dp = 0, dcol = 0x7fffffff;
    T   dg = (T)0;
    int oc[NO];
    T   ov[NO];
    {
        int cols[kIluW];
        T   vals[kIluW];
#pragma unroll
        for(int m = 0; m < kIluW; ++m)
        {
            cols[m] = m < len ? ci[rs + m] : 0x7fffffff;
            vals[m] = m < len ? val[rs + m] : (T)0;
            dp += (cols[m] < i) ? 1 : 0;
        }
#pragma unroll
        for(int m = 0; m < kIluW; ++m)
        {
            dcol = (m == dp) ? cols[m] : dcol;
            dg   = (m == dp) ? vals[m] : dg;
        }
#pragma unroll
        for(int m = 0; m < NO; ++m)
        {
            oc[m] = m < dp ? cols[m] : cols[m + 1];
            ov[m] = m < dp ? vals[m] : vals[m + 1];
        }
    }
    const int          nup     = len - 1 - dp; // entries right of the pivot entry (< 0: there is none)
    unsigned long long inwave  = 0ull;
    int                a       = 0; // next pivot of the row
    bool               fin     = !live;
    int                spins   = 0;
    int                backoff = 1;
    bool stalled = false; // nobody of the wave advanced last turn
    do // (A) pivot rows held by earlier waves; wave-uniform exit as in k_levels
    {
        spin_guard(spins);
        const int  a_before   = a;
        const bool fin_before = fin;
        const bool my_turn = !stalled || lane == (int)__ffsll((long long)__ballot(!fin)) - 1; // (see poll_turn)
        if(!fin && my_turn)
        {
            if(a < dp)
            {
                int k = 0;
#pragma unroll
                for(int m = 0; m < NO; ++m)
                    k = (m == a) ? oc[m] : k;
                if(k >= i0)
                {
                    inwave |= 1ull << (k - i0);
                    ++a;
                }
                else if(__hip_atomic_load(done + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                {
                    const int kd    = __hip_atomic_load(diag_pos + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int kre   = rp[k + 1];
                    const T   pivot = Sentinel<T>::from_bits(__hip_atomic_load(
                        reinterpret_cast<const B*>(val + kd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if(pivot != (T)0)
                    {
                        T x = (T)0;
#pragma unroll
                        for(int m = 0; m < NO; ++m)
                            x = (m == a) ? ov[m] : x;
                        const T f = x / pivot;
#pragma unroll
                        for(int m = 0; m < NO; ++m)
                            ov[m] = (m == a) ? f : ov[m];
                        for(int q = kd + 1; q < kre; ++q)
                        {
                            const int cq  = ci[q];
                            const T   akq = Sentinel<T>::from_bits(__hip_atomic_load(
                                reinterpret_cast<const B*>(val + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            if(cq == dcol)
                                dg -= f * akq;
#pragma unroll
                            for(int m = 0; m < NO; ++m) // (cq > k: a match lies right of (i, k))
                                if(oc[m] == cq)
                                    ov[m] -= f * akq;
                        }
                    }
                    ++a;
                }
            }
            else
                fin = true;
        }
        const bool advanced = __ballot(!fin_before && (fin || a != a_before)) != 0ull;
        stalled             = !advanced;
        backoff             = poll_backoff(advanced, backoff, poll_cap);
    } while(__ballot(!fin) != 0ull);
    if(__ballot(inwave != 0ull) != 0ull) // (B) pivot rows held by lower lanes
    {
        const int meta = (dp & 0xff) | ((nup & 0xff) << 8); // (nup = -1 -> 0xff)
        for(int b = 0; b < 63; ++b)
        {
            const bool mine = lane_bit(inwave, b);
            if(__ballot(mine) == 0ull)
                continue;
            const int metab = __builtin_amdgcn_readlane(meta, b);
            const int dpb = metab & 0xff, nupb = (metab >> 8) & 0xff;
            const T   pivot = bcast_lane<T>(dg, b);
            if(nupb == 0xff || pivot == (T)0) // (no entry at or right of the diagonal: nothing to eliminate with)
                continue;
            const int kb = i0 + b;
            T         x  = (T)0;
#pragma unroll
            for(int m = 0; m < NO; ++m)
                x = (oc[m] == kb) ? ov[m] : x;
            const T f = x / pivot;
#pragma unroll
            for(int m = 0; m < NO; ++m)
                ov[m] = (mine && oc[m] == kb) ? f : ov[m];
#pragma unroll
            for(int mb = 0; mb < NO; ++mb)
                if(mb >= dpb && mb < dpb + nupb) // (wave-uniform)
                {
                    const int cb = __builtin_amdgcn_readlane(oc[mb], b);
                    const T   vb = bcast_lane<T>(ov[mb], b);
                    const int cm = mine ? cb : -1; // (no column is -1)
                    // (most entries of a pivot row meet nothing in the rows that use it -- on a 7-point grid two of three:
                    //  the column tests alone, 8 of the ~35 instructions, tell)
                    bool hit = (dcol == cm);
#pragma unroll
                    for(int m = 0; m < NO; ++m)
                        hit = hit || (oc[m] == cm);
                    if(__ballot(hit) == 0ull)
                        continue;
                    if(dcol == cm)
                        dg -= f * vb;
#pragma unroll
                    for(int m = 0; m < NO; ++m)
                        if(oc[m] == cm)
                            ov[m] -= f * vb;
                }
        }
    }
    if(live)
    {
        // publish the finished row write-through, then the flag
#pragma unroll
        for(int m = 0; m < kIluW; ++m)
            if(m < len)
            {
                const T v = m == dp ? dg : (m < dp ? ov[m < NO ? m : 0] : ov[m > 0 ? m - 1 : 0]);
                __hip_atomic_store(reinterpret_cast<B*>(val + rs + m), Sentinel<T>::as_bits(v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        __hip_atomic_store(diag_pos + i, rs + dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(done + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// register slots a row needs in k_ilu0_rows: its entries, plus one where NO entry sits at or right of the diagonal (columns
// are sorted: the last one is left of it) -- such a row has its "pivot entry" one past its end, and a row of exactly W
// entries of that kind would not fit W slots (its last column would be dropped and a foreign value slot written)
__global__ __launch_bounds__(kBlock) void k_max_row_len(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                        int* __restrict__ out)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    int           mx  = 0;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gsz)
    {
        const int rs = rp[r], re = rp[r + 1];
        mx = max(mx, re - rs + ((re > rs && ci[re - 1] < (int)r) ? 1 : 0));
    }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
        mx = max(mx, __shfl_xor(mx, off, 64));
    if((threadIdx.x & 63) == 0 && mx > 0)
        atomicMax(out, mx);
}

// ---------------------------------------------------------------- plans
struct TriPlan
{
    int   n         = 0;
    int   nslices   = 0;
    int*  order     = nullptr; // [n] position -> row
    int*  pos       = nullptr; // [n] row -> position
    int*  slice_off = nullptr; // [nslices+1]
    int*  ecol      = nullptr;
    void* eval      = nullptr;
    void* diag      = nullptr; // [n] diagonal value per position
    void* w         = nullptr; // [n] polled scratch
    int   nlevels   = 0;
    bool  nodiag    = false;
    bool  filled_once = false;
    // box-tile form (build_ct_plan): tiles = boxes in three monotone dependency coordinates, one workgroup per tile
    bool ct        = false;
    bool ct_rec    = false; // rows of <= 8 entries: record form (k_trsv_rec), `eval` holds the quads
    int  ct_ntiles = 0, ct_nsteps = 0, ct_wmax = 0;
    bool ct_grp     = false; // grouped form: row groups (supernodes) are one step each, kGrpLPR lanes per row
    bool ct_infirst = false; // ... whose in-group entries come first in the order of the host loop (upper solve)
    int  ct_dims[4] = {0, 0, 0, 0}; // max rows / steps / packed entries / external dependencies of a tile
    int* ct_tile_step = nullptr; // [ntiles+1] first step of a tile
    int* ct_step_pos  = nullptr; // [nsteps+1] first position of a step (a step = the rows of one level inside a tile)
    int* ct_step_ent  = nullptr; // [nsteps+1] first packed entry of a step
    int* ct_ext_start = nullptr; // [n+1] first external dependency of a position (running count in position order)
    int* ct_ext_idx   = nullptr; // [next] positions the external dependencies refer to
    int* ct_tile_desc = nullptr; // [8 * ntiles] k_ct_tile_desc
    int* ct_step_rec  = nullptr; // [4 * (nsteps + 1)] k_ct_step_rec
    // record form: the rows of every tile sorted by where their right-hand side comes from (source index, row number inside
    // the tile), so that the fetch wave reads a tile's right-hand side with line-sized runs; valid for the index array `ct_in_key`
    int*       ct_in_pairs  = nullptr; // [2 n] {source index, row number inside the tile}
    const int* ct_in_key    = nullptr;
    int*       ct_out_pairs = nullptr; // [2 n] the same for the natural-order output: {row of the matrix, row number inside the tile}
    bool       ct_in_packed = false, ct_out_packed = false; // the lists in 4 bytes per row (see CtDims)
    bool       prefilled_next = false; // the last run filled the next stage's w with sentinels (run_plan)
    bool       w_sentinel     = false; // w holds sentinels everywhere (the stage that read it as its right-hand side left them)
    // lattice form (trsv_lattice.hip): the triangle of a 5- / 7-point lattice operator, pencils marched along x; such a plan
    // has no order / pos / w -- it reads and writes natural-order vectors
    LatPlan* lat = nullptr;
    // sync-free grouped form (k_trsv_sf): a deep, narrow dependency graph of long rows, one row group per hand-off
    SfPlan* sf = nullptr;
    // 27-point form (trsv_box27.hip): the triangle of the full 3 x 3 x 3 stencil on a lattice, pencils marched along x; like
    // the lattice form it reads and writes natural-order vectors, unlike it the output vector is its hand-off medium
    BoxPlan* box = nullptr;
    // what the analysis found (ramd_tri_plan_stats): chains, external values of all tiles, box edges
    int       st_chains = 0, st_box[3] = {0, 0, 0};
    int       st_why    = 0; // why this plan is not in box-tile form (ct_why_text)
    long long st_ext    = 0;
    void  release()
    {
        lat_release(&lat);
        sf_release(&sf);
        box_release(&box);
        dev_free(&ct_tile_step);
        dev_free(&ct_step_pos);
        dev_free(&ct_step_ent);
        dev_free(&ct_ext_start);
        dev_free(&ct_ext_idx);
        dev_free(&ct_tile_desc);
        dev_free(&ct_step_rec);
        dev_free(&ct_in_pairs);
        dev_free(&ct_out_pairs);
        ct_in_key = nullptr;
        ct = ct_rec = ct_grp = ct_infirst = ct_in_packed = ct_out_packed = false;
        dev_free(&order);
        dev_free(&pos);
        dev_free(&slice_off);
        dev_free(&ecol);
        if(eval)
            (void)cached_free(eval);
        if(diag)
            (void)cached_free(diag);
        if(w)
            (void)cached_free(w);
        eval = diag = w = nullptr;
        n               = 0;
        w_sentinel = prefilled_next = filled_once = false;
    }
};

struct TriState
{
    // LUAnalyse owns L and U; LAnalyse / UAnalyse own Ls / Us: independent plans, as the analyses are independent calls with
    // their own diagonal flag (a lattice plan fixes the flag when it is built, and LUSolve chains L into U through lu_rhs_idx
    // and the shared scratch: interleaved LUAnalyse / LAnalyse / UAnalyse / *Clear must not see each other's plans)
    TriPlan   L, U, Ls, Us;
    bool      haveL = false, haveU = false;
    int*      lu_rhs_idx = nullptr; // [n]: U position -> L position of the same row
    unsigned* counter    = nullptr; // shared workgroup ticket
    unsigned  ticket     = 0; // host copy of the counter value
    // ticket streams of the persistent box-tile solve: one counter word serves ~88 tickets per microsecond (262144 tiles at
    // 512^3 = 3 ms of tickets alone); tile k belongs to stream k % streams, every stream has its own word (own 4 KB page)
    static constexpr int kStreams = 16, kStreamStride = 1024;
    unsigned* stream_counter = nullptr; // [kStreams * kStreamStride]
    unsigned  stream_ticket[kStreams] = {0}; // host copies
    // rows ordered by (lower-dependency level, row), computed once per pattern: ILU(0) runs in this order
    // and LAnalyse / LUAnalyse take it over
    int* l_order_cache = nullptr;
    int* l_level_cache = nullptr; // per-row levels of the same sweep (the box-tile plan of L takes them over)
    int  l_nlev_cache  = 0;
    // LLSolve (incomplete Cholesky): forward plan on L, backward plan on L^T, both scaled by an inverse diagonal
    TriPlan     LLf, LLb;
    bool        haveLL      = false;
    int*        ll_rhs_idx  = nullptr; // [n]: L^T position -> L position of the same row
    const void* ll_diag_src = nullptr; // inverse-diagonal vector the plans' diag arrays were gathered from
    // iterative (Jacobi-sweep) triangular solves, TriSolverAlg_Iterative: natural-order sliced-ELL triangles
    struct ItSlot
    {
        TriPlan A, B; // LU: L, U    LL: L, L^T    L / U alone: A
        int     kind      = 0; // 0 none, 1 LU, 2 LL, 3 L, 4 U
        bool    unit      = false; // L / U alone: unit diagonal
        bool    zero_diag = false; // a non-unit stage has a zero pivot: the reference leaves the output untouched
    } it[3]; // [0] LU or LL, [1] L alone, [2] U alone (SGS analyses L and U of one matrix)
    void*   it_tmp       = nullptr; // [n] persistent intermediate vector (tmp_vec_), zero at analysis
    void*   it_buf       = nullptr; // [n] previous-iterate buffer of the sweeps
    void*   it_ctl       = nullptr; // ItCtl
};

static int it_slot_of(int kind)
{
    return kind <= 2 ? 0 : kind - 2;
}
// idx < 0: everything
static void it_release(TriState* st, int idx = -1)
{
    for(int i = 0; i < 3; ++i)
        if(idx < 0 || idx == i)
        {
            st->it[i].A.release();
            st->it[i].B.release();
            st->it[i].kind = 0;
        }
    if(st->it[0].kind == 0 && st->it_tmp)
    {
        (void)cached_free(st->it_tmp);
        st->it_tmp = nullptr;
    }
    if(st->it[0].kind == 0 && st->it[1].kind == 0 && st->it[2].kind == 0)
    {
        if(st->it_buf)
            (void)cached_free(st->it_buf);
        if(st->it_ctl)
            (void)hipFree(st->it_ctl);
        st->it_buf = st->it_ctl = nullptr;
    }
}

static TriState* tri_state(ramd_mat_s* m)
{
    return reinterpret_cast<TriState*>(m->tri);
}

void tri_release(ramd_mat_s* m)
{
    TriState* st = tri_state(m);
    if(!st)
        return;
    st->L.release();
    st->U.release();
    st->Ls.release();
    st->Us.release();
    dev_free(&st->lu_rhs_idx);
    dev_free(&st->counter);
    dev_free(&st->stream_counter);
    dev_free(&st->l_order_cache);
    dev_free(&st->l_level_cache);
    st->LLf.release();
    st->LLb.release();
    dev_free(&st->ll_rhs_idx);
    it_release(st);
    delete st;
    m->tri = nullptr;
}

static int tri_get(ramd_mat_s* m, TriState** out)
{
    if(!m->tri)
    {
        TriState* st = new TriState;
        int       s  = dev_alloc(&st->counter, 4);
        if(s != RAMD_OK)
        {
            delete st;
            return s;
        }
        hipError_t e = hipMemsetAsync(st->counter, 0, sizeof(unsigned) * 4, backend().cur);
        if(e != hipSuccess)
        {
            dev_free(&st->counter);
            delete st;
            RAMD_HIP(e);
        }
        m->tri = st;
    }
    *out = tri_state(m);
    return RAMD_OK;
}

// after a failed launch of a ticketed kernel the device counter and its host copy may disagree (every later sync-free
// kernel on this matrix would then map its workgroups to the wrong blocks): bring both back to zero
static void tri_resync(TriState* st)
{
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    if(st && st->counter)
        (void)hipMemset(st->counter, 0, sizeof(unsigned) * 4);
    if(st && st->stream_counter)
        (void)hipMemset(st->stream_counter, 0, sizeof(unsigned) * TriState::kStreams * TriState::kStreamStride);
    if(st)
    {
        st->ticket = 0;
        for(int i = 0; i < TriState::kStreams; ++i)
            st->stream_ticket[i] = 0;
    }
}

static unsigned nblocks_of(int n)
{
    return (unsigned)((n + kBlock - 1) / kBlock);
}
static int sweep_poll_cap()
{
    // longest sleep between two polls of a stalled wave, in units of 64 cycles (device_utils.hpp poll_backoff)
    static const int v = getenv("RAMD_SWEEP_POLLCAP") ? atoi(getenv("RAMD_SWEEP_POLLCAP")) : 64;
    return v;
}
static int sweep_block_size()
{
    static const int v = getenv("RAMD_SWEEP_BLOCK") ? atoi(getenv("RAMD_SWEEP_BLOCK")) : kSweepBlock; // (64 .. 1024, experiments)
    return v;
}
static unsigned sweep_blocks(int nunits) // workgroups for that many wave units
{
    const int per = sweep_block_size() / 64;
    return (unsigned)((nunits + per - 1) / per);
}

// dependency levels (sync-free sweep in natural order) and the rows ordered by (level, row): a stable sort,
// so rows of one level keep ascending row order and neighbouring positions poll / gather neighbouring memory
// (order_out == nullptr: only the levels are wanted; units: a unit order the caller holds already, see unit_schedule)
static int level_order(ramd_mat_s* m, TriState* st, bool lower, int** order_out, int* nlev_out, int** level_out = nullptr,
                       const UnitPlan* units = nullptr)
{
    Backend&       b     = backend();
    const int      n     = m->nrow;
    int*           level = nullptr;
    const unsigned nb    = nblocks_of(n);
    RAMD_TRY(dev_alloc(&level, n));
    hipError_t e = hipMemsetAsync(level, 0, sizeof(int) * (size_t)n, b.cur);
    UnitPlan own;
    build_mark(nullptr);
    if(!units)
    {
        int su = unit_schedule(m, lower, &own);
        if(su != RAMD_OK)
        {
            dev_free(&level);
            return su;
        }
        units = &own;
    }
    build_mark("levels: unit schedule");
    const unsigned nbs = sweep_blocks(units->nunits);
    if(lower)
        hipLaunchKernelGGL((k_levels<true>), dim3(nbs), dim3(sweep_block_size()), 0, b.cur, n, m->rp, m->ci, level,
                           st->counter, st->ticket, unit_view(*units), sweep_poll_cap());
    else
        hipLaunchKernelGGL((k_levels<false>), dim3(nbs), dim3(sweep_block_size()), 0, b.cur, n, m->rp, m->ci, level,
                           st->counter, st->ticket, unit_view(*units), sweep_poll_cap());
    st->ticket += nbs;
    int nlev = 0;
    int s    = (e == hipSuccess) ? device_max_int(level, n, &nlev) : RAMD_ERR_HIP; // (synchronises)
    build_mark("levels: sweep");
    own.release();
    int* order = nullptr;
    if(s == RAMD_OK && order_out)
        s = dev_alloc(&order, n);
    if(s == RAMD_OK && order_out)
        s = device_stable_sort_by_key(level, n, nlev, order);
    build_mark("levels: sort by level");
    if(s == RAMD_OK && level_out)
        *level_out = level; // the caller keeps (and frees) the per-row levels
    else
        dev_free(&level);
    if(s != RAMD_OK)
    {
        dev_free(&order);
        return s;
    }
    if(order_out)
        *order_out = order;
    *nlev_out = nlev;
    return RAMD_OK;
}

__global__ __launch_bounds__(kBlock) void k_natural_order(int n, int* __restrict__ v)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        v[t] = (int)t;
}

static bool ct_enabled();
// why the last build_ct_plan returned RAMD_ERR_UNSUPPORTED (ramd_tri_plan_stats reports it: no matrix falls back silently)
static int         g_ct_gave_up = 0;
static const char* ct_why_text(int why)
{
    switch(why)
    {
    case 1: return "no chains: fewer than 8 consecutively numbered dependent rows per chain on average";
    case 2: return "no dependencies at all";
    case 3: return "triangular rows longer than 32 entries that do not form row groups";
    case 4: return "too many tiles for the 30-bit tile keys";
    case 5: return "more than 2^31 packed entries";
    case 6: return "the tiles cannot be made to fit the LDS";
    case 7: return "fewer rows than the box-tile form is worth (RAMD_TRSV_CT_MINROWS)";
    case 8: return "switched off (RAMD_TRSV_CT=0)";
    case 9: return "row groups with more than 24 entries outside the group: the sync-free grouped form takes these";
    default: return "";
    }
}
template <typename T>
static int build_ct_plan(ramd_mat_s* m, TriState* st, TriPlan* P, bool lower, bool reverse);
template <typename T>
static int build_sf_plan(ramd_mat_s* m, TriState* st, TriPlan* P, bool lower, bool reverse);

// natural = true: rows stay in matrix order (no level analysis) -- the packing of the iterative (Jacobi-sweep) solves
template <typename T>
static int build_plan(ramd_mat_s* m, TriState* st, TriPlan* P, bool lower, bool reverse = false, bool natural = false)
{
    Backend&  b = backend();
    const int n = m->nrow;
    P->release();
    P->n       = n;
    P->nslices = (n + 63) / 64;
    if(n == 0)
        return RAMD_OK;
    P->st_why = (!natural && !ct_enabled()) ? 8 : 0;
    if(!natural && ct_enabled())
    {
        const int sc = build_ct_plan<T>(m, st, P, lower, reverse);
        P->st_why    = sc == RAMD_ERR_UNSUPPORTED ? g_ct_gave_up : 0;
        if(sc == RAMD_OK)
        {
            if(lower && st->l_order_cache) // the level order of ILU0Factorize is not needed by this form
                dev_free(&st->l_order_cache);
            if(lower)
                dev_free(&st->l_level_cache);
            return RAMD_OK;
        }
        if(sc != RAMD_ERR_UNSUPPORTED)
            return sc;
        P->release();
        P->n       = n;
        P->nslices = (n + 63) / 64;
    }
    // deep, narrow dependency graphs the tiles could not take: one row group per hand-off (k_trsv_sf)
    if(!natural && !(reverse && lower))
    {
        const int why = P->st_why;
        const int sc  = build_sf_plan<T>(m, st, P, lower, reverse);
        if(sc == RAMD_OK)
        {
            P->st_why = why;
            if(lower && st->l_order_cache)
                dev_free(&st->l_order_cache);
            if(lower)
                dev_free(&st->l_level_cache);
            return RAMD_OK;
        }
        if(sc != RAMD_ERR_UNSUPPORTED)
            return sc;
        P->release();
        P->n       = n;
        P->nslices = (n + 63) / 64;
        P->st_why  = why;
    }
    const unsigned nb   = nblocks_of(n);
    int            nlev = 0;
    int            s    = RAMD_OK;
    if(natural)
    {
        s = dev_alloc(&P->order, n);
        if(s == RAMD_OK)
            hipLaunchKernelGGL(k_natural_order, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, P->order);
    }
    else if(lower && st->l_order_cache) // computed by ILU0Factorize on the same pattern
    {
        P->order          = st->l_order_cache;
        nlev              = st->l_nlev_cache;
        st->l_order_cache = nullptr;
    }
    else
        s = level_order(m, st, lower, &P->order, &nlev);
    const int grid = ew_grid(n);
    if(s == RAMD_OK)
        s = dev_alloc(&P->pos, n);
    if(s == RAMD_OK)
        hipLaunchKernelGGL(k_invert_perm, dim3(grid), dim3(kBlock), 0, b.cur, n, P->order, P->pos);
    // slice widths -> offsets
    if(s == RAMD_OK)
        s = dev_alloc(&P->slice_off, (int64_t)P->nslices + 1);
    if(s == RAMD_OK)
    {
        hipError_t e = hipMemsetAsync(P->slice_off, 0, sizeof(int) * ((size_t)P->nslices + 1), b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    int total = 0;
    if(s == RAMD_OK)
    {
        const unsigned g64 = (unsigned)(((int64_t)P->nslices * 64 + kBlock - 1) / kBlock);
        if(lower)
            hipLaunchKernelGGL((k_tri_width<true>), dim3(g64), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                               P->order, P->slice_off);
        else
            hipLaunchKernelGGL((k_tri_width<false>), dim3(g64), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                               P->order, P->slice_off);
        s = device_exclusive_scan(P->slice_off, P->slice_off, (int64_t)P->nslices + 1);
        if(s == RAMD_OK)
        {
            hipError_t e = hipMemcpyAsync(&total, P->slice_off + P->nslices, sizeof(int),
                                          hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
        }
    }
    int* nodiag = nullptr;
    if(s == RAMD_OK)
        s = dev_alloc(&P->ecol, total);
    if(s == RAMD_OK && cached_malloc(&P->eval, (size_t)total * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && cached_malloc(&P->diag, (size_t)n * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && cached_malloc(&P->w, (size_t)n * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK)
        s = dev_alloc(&nodiag, 1);
    if(s == RAMD_OK)
    {
        hipError_t e = hipMemsetAsync(nodiag, 0, sizeof(int), b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    if(s == RAMD_OK)
    {
        if(lower)
            hipLaunchKernelGGL((k_tri_fill<T, true>), dim3(nb), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                               (const T*)m->val, P->order, P->pos, P->slice_off, P->ecol, (T*)P->eval,
                               (T*)P->diag, nodiag, reverse ? 1 : 0);
        else
            hipLaunchKernelGGL((k_tri_fill<T, false>), dim3(nb), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,
                               (const T*)m->val, P->order, P->pos, P->slice_off, P->ecol, (T*)P->eval,
                               (T*)P->diag, nodiag, reverse ? 1 : 0);
        int        nd = 0;
        hipError_t e  = hipMemcpyAsync(&nd, nodiag, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
        P->nodiag = nd != 0;
    }
    P->nlevels = nlev;
    dev_free(&nodiag);
    if(s != RAMD_OK)
        P->release();
    return s;
}

// ======================================================================= box-tile triangular solve
// The level-scheduled kernel above pays one cross-CU hand-off (~5 us) per dependency level: 1534 levels at 512^3, ~8200 on
// a 2-D shell mesh with 5 unknowns per node (deep, narrow DAG: GMRES+ILU(0) there was SLOWER than one CPU core).  Making
// the hand-off itself cheaper does not help: tiles that stay one level apart run in lockstep at that latency whatever the
// tile shape (measured: chains x level bands with per-step polling 5.4 ms, with a run-ahead fetcher wave 55 ms at 512^3).
// What helps is a tile that is SMALL IN EVERY DEPENDENCY DIRECTION: its own dependencies are resolved in LDS, and a
// successor tile only trails it by the few steps its first rows need.
//
// Tiles for any matrix -- three MONOTONE coordinates per row, propagated along the dependency DAG:
//      c0(v) = max over dependencies u of c0(u) + [u is v's chain predecessor]   (chain: v depends on the row right before it)
//      c1(v) = max ... c1(u) + [u lies in the chain right before v's chain]
//      c2(v) = max ... c2(u) + [u lies in an earlier chain than that]
// Every coordinate is non-decreasing along EVERY edge, so boxes (c0 / b0, c1 / b1, c2 / b2) form an acyclic tile graph for
// any matrix, and tiles taken in order of (t0 + t1 + t2, t2, t1, t0) -- the ticket order -- only wait on tiles that
// already started.  On a lexicographic 3-D stencil the coordinates are the grid coordinates (boxes = cubes), on a
// banded 2-D FE matrix they are (skewed position along the mesh line, mesh line): parallelograms.
// Inside a tile the rows run in order of their global dependency level (one step per level present in the tile, at most
// 64 / lanes-per-row rows per step).  The solve kernel is k_trsv_rec further down ("record form").
// The arithmetic per row is unchanged (ascending columns, divide by the stored diagonal): bit-exact with the host.

// sweep space: t = row (lower solve) or n-1-row (upper solve).  start[t] = 1 if row t does not depend on row t-1.
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_chain_start(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                           int* __restrict__ start)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(t > n)
        return;
    if(t == n)
    {
        start[t] = 0; // scan sentinel
        return;
    }
    const int i    = LOWER ? (int)t : (int)(n - 1 - t);
    const int prev = LOWER ? i - 1 : i + 1;
    bool      cont = false;
    if(t > 0)
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(ci[j] == prev)
            {
                cont = true;
                break;
            }
    start[t] = cont ? 0 : 1;
}

// ---- row groups (supernodes).  The unknowns of one mesh node of an FE matrix form a dense diagonal block: row t of the
// sweep depends on row t-1 AND shares every other dependency with it (lower part of t = lower part of t-1 plus {t-1}).  Such
// rows are a serial chain -- one dependency level each, 5 levels per node of a shell mesh, ~8200 levels on the af_shell10-class
// matrix -- but their in-group entries are the LAST entries of the row in the order of the host loop (lower solve; the FIRST
// ones in the upper solve).  A group (<= kGrpMax consecutive rows of one supernode) is therefore ONE vertex of the tile DAG and
// ONE step of the solve: the step sums the out-of-group entries of all its rows at once and finishes the in-group ones in
// rounds through lane permutes (k_trsv_rec, grouped form) -- the same subtractions in the same order as the host loop.
// brk[t] = 1: sweep row t does not continue the supernode of row t-1.
constexpr int kGrpLPR = 4, kGrpWL = 6; // (kGrpMax: trsv_syncfree.hpp)
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_sn_breaks(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                         int* __restrict__ brk)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(t > n)
        return;
    if(t == n)
    {
        brk[t] = 0; // scan sentinel
        return;
    }
    bool same = false;
    if(t > 0)
    {
        const int i = LOWER ? (int)t : (int)(n - 1 - t);
        const int p = LOWER ? i - 1 : i + 1; // the row before i in the sweep
        if(LOWER)
        {
            // lower(i) == lower(p) followed by p  (sorted rows)
            int a = rp[p], b = rp[i];
            const int ae = rp[p + 1], be = rp[i + 1];
            same = true;
            while(a < ae && ci[a] < p)
            {
                if(b >= be || ci[b] != ci[a])
                {
                    same = false;
                    break;
                }
                ++a;
                ++b;
            }
            if(same)
                same = (b < be && ci[b] == p) && (b + 1 >= be || ci[b + 1] >= i);
        }
        else
        {
            // upper(i) == p followed by upper(p)
            int a = rp[p], b = rp[i];
            const int ae = rp[p + 1], be = rp[i + 1];
            while(a < ae && ci[a] <= p)
                ++a;
            while(b < be && ci[b] <= i)
                ++b;
            same = (b < be && ci[b] == p);
            if(same)
            {
                ++b;
                same = (ae - a) == (be - b);
                for(; same && a < ae; ++a, ++b)
                    same = ci[a] == ci[b];
            }
        }
    }
    brk[t] = same ? 0 : 1;
}

// supernode runs are cut into groups of at most kGrpMax rows: gs[t] = 1 where a group starts (n + 1 entries, the last 0)
__global__ __launch_bounds__(kBlock) void k_ct_group_starts(int n, const int* __restrict__ brk, const int* __restrict__ bscan,
                                                            const int* __restrict__ run_start, int* __restrict__ gs)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += gsz)
    {
        if(t == n)
        {
            gs[t] = 0;
            continue;
        }
        const int run = bscan[t] + brk[t] - 1;
        gs[t]         = ((t - run_start[run]) % kGrpMax == 0) ? 1 : 0;
    }
}

// first / last sweep row of the group of every sweep row
__global__ __launch_bounds__(kBlock) void k_ct_group_bounds(int n, const int* __restrict__ gs, const int* __restrict__ gscan,
                                                            const int* __restrict__ gstart, int* __restrict__ gf,
                                                            int* __restrict__ gl)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
    {
        const int g = gscan[t] + gs[t] - 1;
        gf[t]       = gstart[g];
        gl[t]       = gstart[g + 1] - 1; // (gstart[ngroups] = n)
    }
}

// group levels, sync-free sweep in sweep order: level of a group = 1 + max level of the groups its rows depend on.  A row
// publishes max(own external dependencies, the row before it in the group) + [it is the group's first row]; readers take
// the word of the LAST row of a dependency's group.  0 = not computed yet.  In-wave dependencies are resolved in registers
// (see k_levels): `near` notes the lanes whose published value is taken as it is, `prev` the row before this one in its group.
template <bool LOWER>
__global__ __launch_bounds__(kSweepBlock) void k_ct_glevels(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                       const int* __restrict__ gf, const int* __restrict__ gl, int* glev,
                                                       unsigned* counter, unsigned base, UnitView uv, int poll_cap)
{
    const unsigned slot = take_ticket(counter, base) * (blockDim.x >> 6) + (threadIdx.x >> 6); // (one ticket per workgroup)
    const int      lane = threadIdx.x & 63;
    const bool     have = slot < (unsigned)uv.nunits; // unit_schedule: the rows of my unit
    const int      unit = have ? (uv.order ? uv.order[slot] : (int)slot) : 0;
    const int      ufirst = have ? uv.ustart[unit] : 0;
    const int64_t  t    = (int64_t)ufirst + lane;
    const bool     live = have && t < uv.ustart[unit + 1];
    const int      t0   = (int)t - lane; // sweep index of lane 0
    const int      i    = live ? (LOWER ? (int)t : (int)(n - 1 - t)) : 0;
    const int      dir  = LOWER ? 1 : -1; // far-to-near in sweep order, as k_levels
    int            j    = live ? (LOWER ? rp[i] : rp[i + 1] - 1) : 0;
    const int      end  = live ? (LOWER ? rp[i + 1] : rp[i] - 1) : 0;
    const int      myf  = live ? gf[t] : 0;
    int            lev  = 0;
    unsigned long long near = 0ull;
    bool           prev    = false;
    bool           fin     = !live;
    int            spins   = 0;
    int            backoff = 1;
    bool stalled = false; // nobody of the wave advanced last turn
    do
    {
        spin_guard(spins);
        const int  j_start = j;
        const bool was_fin = fin;
        const bool my_turn = !stalled || lane == (int)__ffsll((long long)__ballot(!fin)) - 1; // (see poll_turn)
        if(!fin && my_turn)
        {
            while(j != end)
            {
                const int c = ci[j];
                if(LOWER ? (c >= i) : (c <= i))
                {
                    j = end;
                    continue;
                }
                const int tc = LOWER ? c : n - 1 - c;
                if(tc >= myf) // in-group: only the row right before this one carries the group's running maximum
                {
                    if(tc == (int)t - 1)
                    {
                        if(lane > 0)
                            prev = true;
                        else
                        {
                            const int lc = __hip_atomic_load(glev + tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if(lc == 0)
                                break;
                            lev = max(lev, lc - 1);
                        }
                    }
                    j += dir;
                    continue;
                }
                const int tg = gl[tc];
                if(tg >= t0)
                {
                    near |= 1ull << (tg - t0);
                    j += dir;
                    continue;
                }
                const int lc = __hip_atomic_load(glev + tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if(lc == 0)
                    break;
                lev = max(lev, lc);
                j += dir;
            }
            fin = (j == end);
        }
        const bool advanced = __ballot(!was_fin && (fin || j != j_start)) != 0ull;
        stalled             = !advanced;
        backoff             = poll_backoff(advanced, backoff, poll_cap);
    } while(__ballot(!fin) != 0ull);
    if(__ballot(near != 0ull || prev) != 0ull)
        for(int b = 0; b < 63; ++b)
        {
            const int pb = __builtin_amdgcn_readlane(lev, b) + 1; // what lane b publishes
            if(lane_bit(near, b))
                lev = max(lev, pb);
            if(prev && lane == b + 1)
                lev = max(lev, pb - 1);
        }
    if(live)
        __hip_atomic_store(glev + t, lev + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// packed coordinates: bit 63 = computed, c0 24 bits, c1 20 bits, c2 19 bits, each saturating (a clamped monotone
// coordinate is still monotone: far-out tiles just get coarser)
constexpr unsigned long long kCtReady = 1ull << 63;
constexpr int                kCtC0Bits = 24, kCtC1Bits = 20, kCtC2Bits = 19;
__device__ __forceinline__ unsigned long long ct_pack(int c0, int c1, int c2)
{
    c0 = min(c0, (1 << kCtC0Bits) - 1);
    c1 = min(c1, (1 << kCtC1Bits) - 1);
    c2 = min(c2, (1 << kCtC2Bits) - 1);
    return kCtReady | (unsigned long long)c0 | ((unsigned long long)c1 << kCtC0Bits)
           | ((unsigned long long)c2 << (kCtC0Bits + kCtC1Bits));
}
__device__ __forceinline__ int ct_c0(unsigned long long w)
{
    return (int)(w & ((1ull << kCtC0Bits) - 1));
}
__device__ __forceinline__ int ct_c1(unsigned long long w)
{
    return (int)((w >> kCtC0Bits) & ((1ull << kCtC1Bits) - 1));
}
__device__ __forceinline__ int ct_c2(unsigned long long w)
{
    return (int)((w >> (kCtC0Bits + kCtC1Bits)) & ((1ull << kCtC2Bits) - 1));
}

// sync-free sweep in sweep order (as k_levels): a row polls the packed words of its out-of-wave dependencies; the word is the
// flag (one 8-byte agent-scope store per row); in-wave dependencies are resolved in registers, lane by lane (k_levels): a
// lane mask per coordinate notes where the +1 applies.  ext[0..2] = maxima of the three coordinates.
template <bool LOWER>
__global__ __launch_bounds__(kSweepBlock) void k_ct_coords(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      const int* __restrict__ escan, unsigned long long* word,
                                                      int* ext, unsigned* counter, unsigned base,
                                                      UnitView uv, const int* __restrict__ gf,
                                                      const int* __restrict__ gl, int poll_cap)
{
    // gf / gl != nullptr: row groups (k_ct_sn_breaks) -- the coordinates are those of the QUOTIENT graph (a group is one
    // vertex: contiguous pieces of a topological order, so the quotient is acyclic): in-group dependencies add nothing, the
    // word of a dependency is the word of the last row of its group (the maximum over the group, handed from row to row)
    const unsigned slot = take_ticket(counter, base) * (blockDim.x >> 6) + (threadIdx.x >> 6); // (one ticket per workgroup)
    const int      lane = threadIdx.x & 63;
    const bool     have = slot < (unsigned)uv.nunits; // unit_schedule: the rows of my unit
    const int      unit = have ? (uv.order ? uv.order[slot] : (int)slot) : 0;
    const int      ufirst = have ? uv.ustart[unit] : 0;
    const int64_t  t    = (int64_t)ufirst + lane;
    const bool     live = have && t < uv.ustart[unit + 1];
    const int      t0   = (int)t - lane; // sweep index of lane 0
    const int      i    = live ? (LOWER ? (int)t : (int)(n - 1 - t)) : 0;
    const int      dir  = LOWER ? 1 : -1; // far-to-near in sweep order, as k_levels
    int            j    = live ? (LOWER ? rp[i] : rp[i + 1] - 1) : 0;
    const int      end  = live ? (LOWER ? rp[i + 1] : rp[i] - 1) : 0;
    const int      mych = live ? escan[t + 1] - 1 : 0;
    const int      myf  = (live && gf) ? gf[t] : (int)t; // first row of my group
    int            c0 = 0, c1 = 0, c2 = 0;
    unsigned long long near = 0ull, up0 = 0ull, up1 = 0ull, up2 = 0ull; // in-wave dependencies; where a coordinate steps
    bool           fin     = !live;
    int            spins   = 0;
    int            backoff = 1;
    bool stalled = false; // nobody of the wave advanced last turn
    do
    {
        spin_guard(spins);
        const int  j_start = j;
        const bool was_fin = fin;
        const bool my_turn = !stalled || lane == (int)__ffsll((long long)__ballot(!fin)) - 1; // (see poll_turn)
        if(!fin && my_turn)
        {
            while(j != end)
            {
                const int c = ci[j];
                if(LOWER ? (c >= i) : (c <= i))
                {
                    j = end; // sorted rows: no dependency follows in scan direction
                    continue;
                }
                const int tc = LOWER ? c : n - 1 - c;
                if(tc >= myf) // in-group (grouped form only): the row before this one carries the group's maxima so far
                {
                    if(tc == (int)t - 1)
                    {
                        if(lane > 0)
                            near |= 1ull << (lane - 1);
                        else
                        {
                            const unsigned long long wp
                                = __hip_atomic_load(word + tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if(!(wp & kCtReady))
                                break;
                            c0 = max(c0, ct_c0(wp));
                            c1 = max(c1, ct_c1(wp));
                            c2 = max(c2, ct_c2(wp));
                        }
                    }
                    j += dir;
                    continue;
                }
                const int  tg  = gl ? gl[tc] : tc;
                const int  chc = escan[tc + 1] - 1;
                const bool s0 = (tc == myf - 1 && chc == mych), s1 = (chc == mych - 1), s2 = (chc < mych - 1);
                if(tg >= t0)
                {
                    const unsigned long long bit = 1ull << (tg - t0);
                    near |= bit;
                    up0 |= s0 ? bit : 0ull; // (several rows of one group: the maximum of the steps)
                    up1 |= s1 ? bit : 0ull;
                    up2 |= s2 ? bit : 0ull;
                    j += dir;
                    continue;
                }
                const unsigned long long w = __hip_atomic_load(word + tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if(!(w & kCtReady))
                    break;
                c0 = max(c0, ct_c0(w) + (s0 ? 1 : 0));
                c1 = max(c1, ct_c1(w) + (s1 ? 1 : 0));
                c2 = max(c2, ct_c2(w) + (s2 ? 1 : 0));
                j += dir;
            }
            fin = (j == end);
        }
        const bool advanced = __ballot(!was_fin && (fin || j != j_start)) != 0ull;
        stalled             = !advanced;
        backoff             = poll_backoff(advanced, backoff, poll_cap);
    } while(__ballot(!fin) != 0ull);
    if(__ballot(near != 0ull) != 0ull)
        for(int b = 0; b < 63; ++b)
        {
            // what lane b publishes (saturated like the packed word)
            const int p0 = min(__builtin_amdgcn_readlane(c0, b), (1 << kCtC0Bits) - 1);
            const int p1 = min(__builtin_amdgcn_readlane(c1, b), (1 << kCtC1Bits) - 1);
            const int p2 = min(__builtin_amdgcn_readlane(c2, b), (1 << kCtC2Bits) - 1);
            if(lane_bit(near, b))
            {
                c0 = max(c0, p0 + (lane_bit(up0, b) ? 1 : 0));
                c1 = max(c1, p1 + (lane_bit(up1, b) ? 1 : 0));
                c2 = max(c2, p2 + (lane_bit(up2, b) ? 1 : 0));
            }
        }
    if(live)
        __hip_atomic_store(word + t, ct_pack(c0, c1, c2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // maxima: one atomic per wave and coordinate
    int m0 = live ? min(c0, (1 << kCtC0Bits) - 1) : 0, m1 = live ? min(c1, (1 << kCtC1Bits) - 1) : 0,
        m2 = live ? min(c2, (1 << kCtC2Bits) - 1) : 0;
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
    {
        m0 = max(m0, __shfl_xor(m0, off, 64));
        m1 = max(m1, __shfl_xor(m1, off, 64));
        m2 = max(m2, __shfl_xor(m2, off, 64));
    }
    // (one address for every wave: an RMW costs ~10 ns there, 63 ms for the 6.3 M of a 512^3 sweep -- so only where it raises
    //  the maximum, which a plain agent-scope load tells)
    if((threadIdx.x & 63) == 0)
    {
        if(m0 > __hip_atomic_load(ext + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(ext + 0, m0);
        if(m1 > __hip_atomic_load(ext + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(ext + 1, m1);
        if(m2 > __hip_atomic_load(ext + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(ext + 2, m2);
    }
}

// tile key of every row (sweep space): boxes of b0 x b1 x b2 in coordinate space, ordered by (t0+t1+t2, t2, t1, t0)
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_keys(int n, const int* __restrict__ level,
                                                    const unsigned long long* __restrict__ word, int b0, int b1, int b2,
                                                    int T0, int T1, int T2, int* __restrict__ lev_t, int* __restrict__ tkey,
                                                    const int* __restrict__ gl)
{
    // gl != nullptr (row groups): `level` holds the group levels in SWEEP order; every row takes the word and the level of
    // the last row of its group, so a group stays together in one tile and one step
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
    {
        const int                i = LOWER ? (int)t : (int)(n - 1 - t);
        const unsigned long long w = word[gl ? gl[t] : t];
        const int                t0 = ct_c0(w) / b0, t1 = ct_c1(w) / b1, t2 = ct_c2(w) / b2;
        lev_t[t] = gl ? level[gl[t]] : level[i];
        tkey[t]  = (((t0 + t1 + t2) * T2 + t2) * T1 + t1) * T0 + t0;
    }
}

// Rows of one level of a tile may come in any order (they do not depend on each other): the rows another tile reads come
// first, grouped by the direction in which that tile lies (the first box coordinate in which the two tiles differ), then
// the rows nobody else reads.  The fetch waves read other tiles' values with agent-scope loads that pay whole 64-byte lines
// per instruction; in plain sweep order a cube's face row sits alone in its line (one line per 8-byte value), this way the
// face rows of a level are neighbours in w.  cls[t] in sweep space: 0..2 = direction, 3 = not read by another tile.
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_export_class(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                            const unsigned long long* __restrict__ word,
                                                            const int* __restrict__ tkey, int b0, int b1, int b2,
                                                            int* __restrict__ cls)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
    {
        const int i  = LOWER ? (int)t : (int)(n - 1 - t);
        const int tk = tkey[t];
        const unsigned long long w = word[t];
        const int                a0 = ct_c0(w) / b0, a1 = ct_c1(w) / b1;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int c = ci[j];
            if(LOWER ? (c < i) : (c > i))
            {
                const int tc = LOWER ? c : n - 1 - c;
                if(tkey[tc] != tk)
                {
                    const unsigned long long wc = word[tc];
                    const int d = (ct_c0(wc) / b0 != a0) ? 0 : ((ct_c1(wc) / b1 != a1) ? 1 : 2);
                    if(cls[tc] > d)
                        atomicMin(cls + tc, d);
                }
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_ct_class_key(int n, const int* __restrict__ lev_t, int* __restrict__ cls)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        cls[t] = lev_t[t] * 4 + cls[t];
}

__global__ __launch_bounds__(kBlock) void k_ct_gather_int(int64_t n, const int* __restrict__ src, const int* __restrict__ idx,
                                                          int* __restrict__ dst)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        dst[t] = src[idx[t]];
}

// sorted sequences -> row order, tile / step start flags (n+1 entries, the last one 0 for the scans)
__global__ __launch_bounds__(kBlock) void k_ct_flags(int n, int lower, const int* __restrict__ o1, const int* __restrict__ o2,
                                                     const int* __restrict__ k2, const int* __restrict__ lev_t,
                                                     int* __restrict__ order, int* __restrict__ tflag,
                                                     int* __restrict__ sflag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= n; p += gsz)
    {
        if(p == n)
        {
            tflag[p] = sflag[p] = 0;
            continue;
        }
        const int t  = o1[o2[p]];
        order[p]     = lower ? t : n - 1 - t;
        const int tk = k2[o2[p]];
        const int lv = lev_t[t];
        bool      tf = (p == 0), sf = (p == 0);
        if(p > 0)
        {
            const int tq = o1[o2[p - 1]];
            tf           = k2[o2[p - 1]] != tk;
            sf           = tf || lev_t[tq] != lv;
        }
        tflag[p] = tf ? 1 : 0;
        sflag[p] = sf ? 1 : 0;
    }
}

// tile_of / step_of per position (from the exclusive scans of the flags), first step of a tile, first position of a step,
// and the number of strictly-triangular entries of the widest row of every step
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_bounds(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      const int* __restrict__ order, const int* __restrict__ tflag,
                                                      const int* __restrict__ sflag, const int* __restrict__ tscan,
                                                      const int* __restrict__ sscan, int* __restrict__ tile_of,
                                                      int* __restrict__ step_of, int* __restrict__ tile_step,
                                                      int* __restrict__ step_pos, int* __restrict__ step_w)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
    {
        const int tl = tscan[p] + tflag[p] - 1;
        const int st = sscan[p] + sflag[p] - 1;
        tile_of[p]   = tl;
        step_of[p]   = st;
        if(tflag[p])
            tile_step[tl] = st;
        if(sflag[p])
            step_pos[st] = (int)p;
        const int i = order[p];
        int       c = 0;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(LOWER ? (ci[j] < i) : (ci[j] > i))
                ++c;
        if(c > 0)
            atomicMax(step_w + st, c);
    }
}

__global__ __launch_bounds__(kBlock) void k_ct_ent_sizes(int nsteps, const int* __restrict__ step_pos,
                                                         const int* __restrict__ step_w, int* __restrict__ size)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g <= nsteps; g += gsz)
        size[g] = (g < nsteps) ? step_w[g] * (step_pos[g + 1] - step_pos[g]) : 0;
}

// external dependencies (columns owned by another tile) per position -> scanned into ct_ext_start
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_count_ext(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                         const int* __restrict__ order, const int* __restrict__ pos,
                                                         const int* __restrict__ tile_of, int* __restrict__ cnt)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= n; p += gsz)
    {
        int c = 0;
        if(p < n)
        {
            const int i  = order[p];
            const int tl = tile_of[p];
            for(int j = rp[i]; j < rp[i + 1]; ++j)
            {
                const int col = ci[j];
                if((LOWER ? (col < i) : (col > i)) && tile_of[pos[col]] != tl)
                    ++c;
            }
        }
        cnt[p] = c;
    }
}

// ---- where a tile's values sit in w (rows of <= 3 entries, one lane per row; RAMD_TRSV_WSLOT=1 -- measured: 7 bytes per row
// less traffic at 512^3, 2-4 % slower, because a step's publication store becomes four short runs; off by default).  The fetch waves read other tiles' values with
// agent-scope loads -- nothing of them is kept in a cache, every load instruction pays whole 64-byte lines -- and in
// position (= level) order the rows another tile needs are spread over the producer's whole piece of w: one line per
// 8-byte value (profiles/r02_traffic.json: 19-25 bytes per row at 512^3).  So the rows of a tile take their places in w in
// the order {rows some other tile reads, by the smallest such tile; rows nobody else reads}, position order inside each
// class: what one consumer fetches from one producer is a contiguous piece (the face of an 8 x 8 x 8 cube: 8 lines instead
// of ~50), and a step still stores one short run per class.  The place travels in the record's spare 16-bit code.
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_min_consumer(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                            const int* __restrict__ order, const int* __restrict__ pos,
                                                            const int* __restrict__ tile_of, int* __restrict__ cons)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
    {
        const int i  = order[p];
        const int tl = tile_of[p];
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int col = ci[j];
            if(LOWER ? (col < i) : (col > i))
            {
                const int pc = pos[col];
                if(tile_of[pc] != tl && cons[pc] > tl) // (plain read first: most references find the minimum already there)
                    atomicMin(cons + pc, tl);
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_fill_int(int64_t n, int v, int* __restrict__ out)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        out[t] = v;
}

// wmap[o1[o2[k]]] = k: the k-th place of w goes to the position that comes k-th in (tile, consumer class, position) order
__global__ __launch_bounds__(kBlock) void k_ct_wmap(int n, const int* __restrict__ o1, const int* __restrict__ o2,
                                                    int* __restrict__ wmap)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gsz)
        wmap[o1[o2[k]]] = (int)k;
}

// idx[j] = map[idx[j]]
__global__ __launch_bounds__(kBlock) void k_ct_map_idx(int64_t n, const int* __restrict__ map, int* __restrict__ idx)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        idx[t] = map[idx[t]];
}

// ---- external values of a tile, each ONCE.  Rows of FE matrices share their external columns (the 5 unknowns of a mesh
// node are referenced by ~10 rows of the neighbouring tile): listed per reference, a 350-row tile of the shell surrogate
// carried 1300 external values -- LDS, and through it the tile size, was set by the duplicates.  A reference r (numbered in
// use order: position, then entry) OWNS its value if no earlier reference of the same tile names the same position; slots
// are numbered over the owners in use order, every other reference takes the slot of its owner.
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_enum_refs(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                         const int* __restrict__ order, const int* __restrict__ pos,
                                                         const int* __restrict__ tile_of, const int* __restrict__ ref_start,
                                                         int* __restrict__ ref_pc, int* __restrict__ ref_tile)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
    {
        const int i  = order[p];
        const int tl = tile_of[p];
        int       r  = ref_start[p];
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int col = ci[j];
            if(LOWER ? (col < i) : (col > i))
            {
                const int pc = pos[col];
                if(tile_of[pc] != tl)
                {
                    ref_pc[r]   = pc;
                    ref_tile[r] = tl;
                    ++r;
                }
            }
        }
    }
}

// sorted by (tile, position), references in use order inside a group: the first of a group owns the value
__global__ __launch_bounds__(kBlock) void k_ct_ref_heads(int nr, const int* __restrict__ o1, const int* __restrict__ o2,
                                                         const int* __restrict__ ref_pc, const int* __restrict__ ref_tile,
                                                         int* __restrict__ head, int* __restrict__ is_owner)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= nr; q += gsz)
    {
        if(q == nr)
        {
            head[q]     = 0;
            is_owner[q] = 0;
            continue;
        }
        const int r = o1[o2[q]];
        int       h = 1;
        if(q > 0)
        {
            const int rb = o1[o2[q - 1]];
            h            = (ref_pc[r] != ref_pc[rb] || ref_tile[r] != ref_tile[rb]) ? 1 : 0;
        }
        head[q]     = h;
        is_owner[r] = h;
    }
}

__global__ __launch_bounds__(kBlock) void k_ct_group_slots(int nr, const int* __restrict__ o1, const int* __restrict__ o2,
                                                           const int* __restrict__ head, const int* __restrict__ hscan,
                                                           const int* __restrict__ uniq, const int* __restrict__ ref_pc,
                                                           int* __restrict__ group_slot, int* __restrict__ ext_idx)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nr; q += gsz)
        if(head[q])
        {
            const int r          = o1[o2[q]];
            group_slot[hscan[q]] = uniq[r];
            ext_idx[uniq[r]]     = ref_pc[r];
        }
}

__global__ __launch_bounds__(kBlock) void k_ct_ref_slots(int nr, const int* __restrict__ o1, const int* __restrict__ o2,
                                                         const int* __restrict__ head, const int* __restrict__ hscan,
                                                         const int* __restrict__ group_slot, int* __restrict__ slot_of_ref)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nr; q += gsz)
        slot_of_ref[o1[o2[q]]] = group_slot[hscan[q] + head[q] - 1];
}

__global__ __launch_bounds__(kBlock) void k_ct_ext_start_unique(int n, const int* __restrict__ ref_start,
                                                                const int* __restrict__ uniq, int* __restrict__ ext_start)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= n; p += gsz)
        ext_start[p] = uniq[ref_start[p]];
}

// per tile: rows, steps, packed entries, external dependencies (sizes of the LDS areas)
__global__ __launch_bounds__(kBlock) void k_ct_tile_sizes(int ntiles, const int* __restrict__ tile_step,
                                                          const int* __restrict__ step_pos, const int* __restrict__ step_ent,
                                                          const int* __restrict__ ext_start, int* __restrict__ rows,
                                                          int* __restrict__ steps, int* __restrict__ ents,
                                                          int* __restrict__ exts)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gsz)
    {
        const int s0 = tile_step[t], s1 = tile_step[t + 1];
        rows[t]  = step_pos[s1] - step_pos[s0];
        steps[t] = s1 - s0;
        ents[t]  = step_ent[s1] - step_ent[s0];
        exts[t]  = ext_start[step_pos[s1]] - ext_start[step_pos[s0]];
    }
}

// everything a workgroup needs to know about its tile in one 32-byte record:
// {first step, steps, first position, rows | first packed entry, packed entries, first external slot, external slots}
__global__ __launch_bounds__(kBlock) void k_ct_tile_desc(int ntiles, const int* __restrict__ tile_step,
                                                         const int* __restrict__ step_pos, const int* __restrict__ step_ent,
                                                         const int* __restrict__ ext_start, int* __restrict__ desc)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gsz)
    {
        const int s0 = tile_step[t], s1 = tile_step[t + 1];
        const int p0 = step_pos[s0], p1 = step_pos[s1];
        desc[8 * t + 0] = s0;
        desc[8 * t + 1] = s1 - s0;
        desc[8 * t + 2] = p0;
        desc[8 * t + 3] = p1 - p0;
        desc[8 * t + 4] = step_ent[s0];
        desc[8 * t + 5] = step_ent[s1] - step_ent[s0];
        desc[8 * t + 6] = ext_start[p0];
        desc[8 * t + 7] = ext_start[p1] - ext_start[p0];
    }
}

// longest strictly-triangular row part
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_row_wmax(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                        int* __restrict__ out)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    int           m   = 0;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
    {
        int c = 0;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
            if(LOWER ? (ci[j] < i) : (ci[j] > i))
                ++c;
        m = max(m, c);
    }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
        m = max(m, __shfl_xor(m, off, 64));
    if((threadIdx.x & 63) == 0)
        atomicMax(out, m);
}

// a (tile, level) group of more than `rpp` rows becomes several steps (its rows are independent of each other)
__global__ __launch_bounds__(kBlock) void k_ct_split(int n, int rpp, const int* __restrict__ gflag, const int* __restrict__ gscan,
                                                     const int* __restrict__ gpos, int* __restrict__ sflag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= n; p += gsz)
    {
        if(p == n)
        {
            sflag[p] = 0;
            continue;
        }
        const int gid = gscan[p] + gflag[p] - 1;
        sflag[p]      = ((p - gpos[gid]) % rpp == 0) ? 1 : 0;
    }
}

// row groups: a step holds whole groups, at most rpp rows -- one thread walks the rows of a tile (a few hundred) and packs
// the groups of every (tile, level) piece into steps in position order
__global__ __launch_bounds__(kBlock) void k_ct_split_groups(int ntiles, int n, int lower, int rpp, const int* __restrict__ tpos,
                                                            const int* __restrict__ sflag, const int* __restrict__ order,
                                                            const int* __restrict__ gf, const int* __restrict__ gl,
                                                            int* __restrict__ sflag2)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t tl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tl < ntiles; tl += gsz)
    {
        int fill = 0;
        for(int p = tpos[tl]; p < tpos[tl + 1]; ++p)
        {
            const int t  = lower ? order[p] : n - 1 - order[p];
            int       st = 0;
            if(sflag[p])
            {
                st   = 1;
                fill = 0;
            }
            if(gf[t] == t) // first row of its group
            {
                const int gsize = gl[t] - t + 1;
                if(fill + gsize > rpp)
                {
                    st   = 1;
                    fill = 0;
                }
                fill += gsize;
            }
            sflag2[p] = st;
        }
        if(tl == ntiles - 1)
            sflag2[n] = 0;
    }
}

// largest group of every step (rows)
__global__ __launch_bounds__(kBlock) void k_ct_step_maxg(int n, int lower, const int* __restrict__ order,
                                                         const int* __restrict__ step_of, const int* __restrict__ gf,
                                                         const int* __restrict__ gl, int* __restrict__ step_maxg)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
    {
        const int t = lower ? order[p] : n - 1 - order[p];
        if(gf[t] == t && gl[t] > t)
            atomicMax(step_maxg + step_of[p], gl[t] - t + 1);
    }
}

// longest strictly-triangular row part OUTSIDE the row's group
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_row_wmax_out(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                            const int* __restrict__ gf, const int* __restrict__ gl,
                                                            int* __restrict__ out, int* __restrict__ out_gsize)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    int           m = 0, mg = 0;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
    {
        const int i = LOWER ? (int)t : (int)(n - 1 - t);
        int       c = 0;
        mg          = max(mg, gl[t] - gf[t] + 1);
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int col = ci[j];
            if(LOWER ? (col < i) : (col > i))
                if((LOWER ? col : n - 1 - col) < gf[t])
                    ++c;
        }
        m = max(m, c);
    }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
    {
        m  = max(m, __shfl_xor(m, off, 64));
        mg = max(mg, __shfl_xor(mg, off, 64));
    }
    if((threadIdx.x & 63) == 0)
    {
        atomicMax(out, m);
        atomicMax(out_gsize, mg);
    }
}

__global__ __launch_bounds__(kBlock) void k_ct_scatter_starts(int n, const int* __restrict__ flag, const int* __restrict__ scan,
                                                              int* __restrict__ start)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
        if(flag[p])
            start[scan[p]] = (int)p;
}

// LDS areas of a workgroup (element counts, the plan's maxima over tiles)
struct CtDims
{
    int rows, steps, ents, exts;
    int infirst; // grouped form: the in-group entries of a row come FIRST in the order of the host loop (upper solve)
    // sorted index lists in 4 bytes per row: (index - the tile's smallest index) << sbits | row number inside the tile
    int in_packed, out_packed, sbits;
    // stages whose result leaves through `out`: w only hands values over to other tiles, and a step publishes just its rows
    // that some other tile reads (they come first in the step; their number travels in the step record)
    int mask_pub;
};
constexpr int kCtFetchDepth = 4; // batches of 64 external values the fetch wave keeps in flight

// ======================================================================= record form of the box-tile solve
// What the counters and the phase timers of round 2 said about the first box-tile kernels (one workgroup per tile, narrow
// per-array loads, everything of a step gathered per step; profiles/r02_trsv_*):
//   * ONE ticket word serves ~88 atomics per microsecond: 262144 tiles at 512^3 cost 3 ms of tickets alone, whatever the
//     kernel did otherwise;
//   * below that, the solve is bound by the NUMBER of vector memory instructions a CU issues (~45 cycles each, 23 rows of 64
//     lanes or not), and by the start-up chain of a tile (ticket -> descriptor -> step records -> first loads);
//   * with dependencies on, the critical path is (tile hops) x (steps a successor trails its predecessor) x (time of a step).
// The record form:
//   * everything static a row needs is ONE record of 16-byte quads -- {column codes as 16-bit LDS indices | values |
//     diagonal} -- stored quad-major per step: 2-3 fully coalesced 16-byte loads per step instead of 8-11 narrow ones
//     (a unit-diagonal solve skips the quad holding the diagonal); rows of 9-32 entries take 8 lanes, each with a record
//     of 4 consecutive entries, and the running value of the row passes from lane to lane through DPP in storage order;
//   * workgroups are PERSISTENT, tickets come from 16 streams (tile k belongs to stream k % 16, every stream its own
//     word on its own page).  The second wave of the workgroup takes the tickets, posts the tile descriptor into an LDS
//     ring and works up to two tiles ahead: it reads the tile's right-hand side -- rows sorted by source index, so that
//     neighbouring lanes read neighbouring addresses -- into the tile's LDS slot, parks the values of other tiles in use
//     order (handing over every value up to the first missing one), and writes the natural-order output of a finished
//     tile back in destination order.  Ticket order still guarantees progress: a tile only waits on lower tickets, all
//     held by running workgroups whose earlier tiles are finished first;
//   * the compute wave walks the steps of tile after tile as one stream whose prefetch never drains: quads DEPTH steps ahead
//     into registers (hand-issued loads, hand-counted vmcnt), step records through the scalar cache, a step itself only
//     touches LDS (right-hand side and solution share a slot per row) and issues one store: the agent-scope publication of
//     its values, which doubles as the position-order result the next stage reads.
// Measured and dropped: a compact EXPORT array (only rows other tiles wait for are published, slots in position order, the
// position-order result written per tile by the fetch wave; external traffic 25 -> ~10 bytes per row, sentinel fill three
// times smaller) -- 2.66 -> 2.76 ms per triangle at 512^3, 5.8 -> 6.5 ms on the shell surrogate: the per-step rank
// computation and the scratch stores of lanes without an exported row cost more than the bytes saved.
// Column codes: 0 = padding (LDS slot 0 holds 0.0 and the padded value is 0: subtracts +0), 1 + q = row q of the tile,
// 1 + rows_max + j = external value j of the tile.
// (compile-time tuning hooks; measured: prefetch depth 3 .. 8 equal on 512^3 / 256^3 / the shell surrogate, 12 slower (registers);
//  ring 2 / 3 equal, 4 slower (LDS))
#ifndef RAMD_CT_RING
#define RAMD_CT_RING 3
#endif
#ifndef RAMD_CT_PRIO
#define RAMD_CT_PRIO 3
#endif
#ifndef RAMD_CT_POLL_CAP
#define RAMD_CT_POLL_CAP 8 // (measured: 64 -> 8 gives 2.5 % at 512^3, 6 % at 256^3, 2.4 % on the shell; 2 is no better)
#endif
#ifndef RAMD_CT_DEPTH3
#define RAMD_CT_DEPTH3 8
#endif
#ifndef RAMD_CT_DEPTH8L
#define RAMD_CT_DEPTH8L 8 // (eight lanes per row)
#endif
#ifndef RAMD_CT_DEPTHG
#define RAMD_CT_DEPTHG 4 // (grouped form: a step is several times longer, the records twice as wide)
#endif
constexpr int kCtRing = RAMD_CT_RING; // tiles the ticket/fetch wave may be ahead of the compute wave (+1)

// records with a spare 16-bit code carry the row's place in its tile's piece of w (see k_ct_min_consumer)
template <int WL, int LPR>
constexpr bool kCtWSlot = (WL == 3 && LPR == 1);

template <typename T, int WL>
struct CtRec
{
    static constexpr int WLC      = (WL + 3) / 4 * 4; // 16-bit codes, padded to whole 8-byte slots
    static constexpr int off_val  = WLC * 2;
    static constexpr int off_diag = off_val + WL * (int)sizeof(T);
    static constexpr int NQ       = (off_diag + (int)sizeof(T) + 15) / 16; // quads of a record with the diagonal in it
    static constexpr int NQL      = (off_diag + 15) / 16; // quads without the diagonal
    // codes + values fill whole quads and the diagonal would sit alone in a half-used one (fp64 rows of <= 3 entries: 48
    // bytes stored for 40): the diagonal then lives in its own position-order array and a step loads it with one 8-byte access
    static constexpr bool DSEP = (off_diag % 16 == 0) && sizeof(T) == 8;
    static constexpr int  NQS  = DSEP ? NQL : NQ; // quads stored per record
};

template <typename T>
static size_t ct_rec_lds_bytes(const CtDims& d)
{
    return (size_t)kCtRing * ((size_t)1 + d.rows + d.exts) * sizeof(T) + (size_t)(2 + 6 * kCtRing) * sizeof(int) + 64;
}

// per step: {first position, rows (| largest row group << 8), external values of the tile used up to and including this step,
// first row's index in the tile}
__global__ __launch_bounds__(kBlock) void k_ct_step_rec2(int n, int nsteps, const int* __restrict__ step_pos,
                                                         const int* __restrict__ ext_start, const int* __restrict__ tile_of,
                                                         const int* __restrict__ tile_step, int* __restrict__ rec,
                                                         const int* __restrict__ step_maxg, const int* __restrict__ step_nexp)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g <= nsteps; g += gsz)
    {
        const int p = step_pos[g];
        if(g == nsteps)
        {
            rec[4 * g + 0] = p;
            rec[4 * g + 1] = 1;
            rec[4 * g + 2] = 0;
            rec[4 * g + 3] = 0;
            continue;
        }
        const int p1   = step_pos[g + 1];
        const int tpos = step_pos[tile_step[tile_of[p]]];
        rec[4 * g + 0] = p;
        // rows | largest group of the step << 8 | rows of the step, counted from its first, up to the last one another tile reads << 16
        rec[4 * g + 1] = (p1 - p) | ((step_maxg ? max(step_maxg[g], 1) : 1) << 8) | ((step_nexp ? step_nexp[g] : (p1 - p)) << 16);
        rec[4 * g + 2] = ext_start[p1] - ext_start[tpos];
        rec[4 * g + 3] = p - tpos;
    }
}

// rows another tile reads (they are some tile's external values), and per step how far they reach into the step: inside a
// level the exported rows come first (k_ct_export_class), so this is normally their count
__global__ __launch_bounds__(kBlock) void k_ct_mark_exported(int64_t next, const int* __restrict__ ext_idx, int* __restrict__ mark)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < next; j += gsz)
        mark[ext_idx[j]] = 1;
}
__global__ __launch_bounds__(kBlock) void k_ct_step_nexp(int nsteps, const int* __restrict__ step_pos, const int* __restrict__ mark,
                                                         int* __restrict__ nexp)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nsteps; g += gsz)
    {
        const int p0 = step_pos[g], p1 = step_pos[g + 1];
        int       last = 0;
        for(int p = p0; p < p1; ++p)
            if(mark[p])
                last = p - p0 + 1;
        // (never 0: the compute wave counts its vector memory operations per step by hand -- one store, then the record
        //  loads -- and a step whose store instruction is skipped because no lane has anything to publish would let the
        //  wait for a record through one operation early; row 0 of a step without exported rows is published for that reason)
        nexp[g] = last > 0 ? last : 1;
    }
}

// LPR lanes share a row (LPR = 8: rows of up to 8 * WL entries): the row's record is LPR lane records of WL consecutive
// entries each; the diagonal sits in the last lane's record (the lane that finishes the row)
template <typename T, bool LOWER, int WL, int LPR>
__global__ __launch_bounds__(kBlock) void k_ct_fill_rec(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                        const T* __restrict__ val, const int* __restrict__ order,
                                                        const int* __restrict__ pos, const int* __restrict__ tile_of,
                                                        const int* __restrict__ step_of, const int* __restrict__ tile_step,
                                                        const int* __restrict__ step_pos, const int* __restrict__ ext_start,
                                                        int* __restrict__ ext_idx, char* __restrict__ erec,
                                                        int* __restrict__ nodiag, int reverse, int rows_max,
                                                        const int* __restrict__ ref_start,
                                                        const int* __restrict__ slot_of_ref, T* __restrict__ diag_sep,
                                                        const int* __restrict__ wmap)
{
    using L         = CtRec<T, WL>;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= n)
        return;
    const int i    = order[p];
    const int gs   = step_of[p];
    const int p0   = step_pos[gs];
    const int cnt  = step_pos[gs + 1] - p0;
    const int rank = (int)p - p0;
    const int tl   = tile_of[p];
    const int tpos = step_pos[tile_step[tl]];
    int       e    = ext_start[p];
    const int e0   = ext_start[tpos];
    // (de-duplicated external values: the j-th external reference of the row, in ascending column order, has the slot
    //  slot_of_ref[ref_start[p] + j]; otherwise every reference has its own slot, numbered in storage order)
    const int rbase = slot_of_ref ? ref_start[p] : 0;
    int       nref  = 0;
    for(int q = rp[i]; slot_of_ref && reverse && q < rp[i + 1]; ++q) // storage order is descending: count first
    {
        const int c = ci[q];
        if((LOWER ? (c < i) : (c > i)) && tile_of[pos[c]] != tl)
            ++nref;
    }
    int jref = 0;
    // byte `off` of the record of lane `sub` of this row: quad-major over the step's cnt * LPR lane records
    auto field = [&](int sub, int off) -> char* {
        constexpr int stride = (L::DSEP && LPR == 1) ? L::NQS : L::NQ; // quads stored per lane record
        return erec + ((size_t)stride * LPR * p0 + (size_t)(off / 16) * (cnt * LPR) + (rank * LPR + sub)) * 16 + (off % 16);
    };
    int       k    = 0;
    bool      have = false;
    const int rs = rp[i], re = rp[i + 1];
    for(int q = rs; q < re; ++q)
    {
        const int j = reverse ? (re - 1 - (q - rs)) : q;
        const int c = ci[j];
        if(LOWER ? (c < i) : (c > i))
        {
            const int pc = pos[c];
            int       code;
            if(tile_of[pc] == tl)
                code = 1 + (pc - tpos);
            else if(slot_of_ref)
            {
                const int r = rbase + (reverse ? (nref - 1 - jref) : jref); // (references are numbered in ascending columns)
                code        = 1 + rows_max + (slot_of_ref[r] - e0);
                ++jref;
            }
            else
            {
                ext_idx[e] = pc;
                code       = 1 + rows_max + (e - e0);
                ++e;
            }
            if(k < WL * LPR)
            {
                const int sub = k / WL, kk = k % WL;
                *reinterpret_cast<unsigned short*>(field(sub, 2 * kk))                 = (unsigned short)code;
                *reinterpret_cast<T*>(field(sub, L::off_val + kk * (int)sizeof(T))) = val[j];
            }
            ++k;
        }
        else if(c == i)
        {
            if(L::DSEP && LPR == 1)
                diag_sep[p] = val[j];
            else
                *reinterpret_cast<T*>(field(LPR - 1, L::off_diag)) = val[j];
            have = true;
        }
    }
    // (padding codes / values stay 0: the array is zeroed before the fill)
    if(kCtWSlot<WL, LPR>) // the row's place in its tile's piece of w (k_ct_min_consumer), in the spare fourth code
        *reinterpret_cast<unsigned short*>(field(0, 2 * WL)) = (unsigned short)((wmap ? wmap[p] : (int)p) - tpos);
    if(!have)
    {
        if(L::DSEP && LPR == 1)
            diag_sep[p] = (T)1;
        else
            *reinterpret_cast<T*>(field(LPR - 1, L::off_diag)) = (T)1;
        *nodiag = 1;
    }
}

// Grouped form (row groups, k_ct_sn_breaks): kGrpLPR lanes per row with kGrpWL out-of-group entries each.
//   lane record  : {kGrpWL 16-bit column codes | meta (4 bytes: index of the row in its group, << 8: which in-group
//                  coefficients exist) | kGrpWL values}, quad-major per step like CtRec
//   row record   : {kGrpMax - 1 in-group coefficients, indexed by the group row they multiply | diagonal}, position order
template <typename T>
struct CtGRec
{
    static constexpr int WLC      = kGrpWL;
    static constexpr int off_meta = 12;
    static constexpr int off_val  = 16;
    static constexpr int off_diag = 0; // (not in the lane record)
    static constexpr int NQ       = (off_val + kGrpWL * (int)sizeof(T) + 15) / 16;
    static constexpr int NQL      = NQ;
    static constexpr int NQS      = NQ;
    static constexpr bool DSEP    = false;
    static constexpr int  NGQ     = kGrpMax * (int)sizeof(T) / 16; // quads of a row record
};

template <typename T, bool LOWER>
__global__ __launch_bounds__(kBlock) void k_ct_fill_grec(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                         const T* __restrict__ val, const int* __restrict__ order,
                                                         const int* __restrict__ pos, const int* __restrict__ tile_of,
                                                         const int* __restrict__ step_of, const int* __restrict__ tile_step,
                                                         const int* __restrict__ step_pos, const int* __restrict__ ext_start,
                                                         int* __restrict__ ext_idx, char* __restrict__ erec,
                                                         int* __restrict__ nodiag, int reverse, int rows_max,
                                                         const int* __restrict__ ref_start,
                                                         const int* __restrict__ slot_of_ref, T* __restrict__ grec,
                                                         const int* __restrict__ gf)
{
    using L           = CtGRec<T>;
    constexpr int LPR = kGrpLPR, WL = kGrpWL;
    const int64_t p   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= n)
        return;
    const int i    = order[p];
    const int t    = LOWER ? i : n - 1 - i;
    const int tf   = gf[t]; // first sweep row of the group
    const int gi   = t - tf;
    const int gs   = step_of[p];
    const int p0   = step_pos[gs];
    const int cnt  = step_pos[gs + 1] - p0;
    const int rank = (int)p - p0;
    const int tl   = tile_of[p];
    const int tpos = step_pos[tile_step[tl]];
    int       e    = ext_start[p];
    const int e0   = ext_start[tpos];
    const int rbase = slot_of_ref ? ref_start[p] : 0;
    int       nref  = 0;
    for(int q = rp[i]; slot_of_ref && reverse && q < rp[i + 1]; ++q) // storage order is descending: count first
    {
        const int c = ci[q];
        if((LOWER ? (c < i) : (c > i)) && tile_of[pos[c]] != tl)
            ++nref;
    }
    int  jref  = 0;
    auto field = [&](int sub, int off) -> char* {
        return erec + ((size_t)L::NQ * LPR * p0 + (size_t)(off / 16) * (cnt * LPR) + (rank * LPR + sub)) * 16 + (off % 16);
    };
    T*        rr   = grec + (size_t)kGrpMax * p;
    int       k    = 0;
    int       mask = 0;
    bool      have = false;
    const int rs = rp[i], re = rp[i + 1];
    for(int q = rs; q < re; ++q)
    {
        const int j = reverse ? (re - 1 - (q - rs)) : q;
        const int c = ci[j];
        if(LOWER ? (c < i) : (c > i))
        {
            const int tc = LOWER ? c : n - 1 - c;
            if(tc >= tf) // in-group: coefficient of group row tc - tf
            {
                rr[tc - tf] = val[j];
                mask |= 1 << (tc - tf);
                continue;
            }
            const int pc = pos[c];
            int       code;
            if(tile_of[pc] == tl)
                code = 1 + (pc - tpos);
            else if(slot_of_ref)
            {
                const int r = rbase + (reverse ? (nref - 1 - jref) : jref);
                code        = 1 + rows_max + (slot_of_ref[r] - e0);
                ++jref;
            }
            else
            {
                ext_idx[e] = pc;
                code       = 1 + rows_max + (e - e0);
                ++e;
            }
            if(k < WL * LPR)
            {
                const int sub = k / WL, kk = k % WL;
                *reinterpret_cast<unsigned short*>(field(sub, 2 * kk))            = (unsigned short)code;
                *reinterpret_cast<T*>(field(sub, L::off_val + kk * (int)sizeof(T))) = val[j];
            }
            ++k;
        }
        else if(c == i)
        {
            rr[kGrpMax - 1] = val[j];
            have            = true;
        }
    }
    for(int sub = 0; sub < LPR; ++sub)
        *reinterpret_cast<int*>(field(sub, L::off_meta)) = gi | (mask << 8);
    if(!have)
    {
        rr[kGrpMax - 1] = (T)1;
        *nodiag         = 1;
    }
}

// The compute wave's loads are issued by hand (inline asm) and waited for by hand: left to the compiler, the counter waits
// of a software pipeline with scalar control flow in the loop body come out as drains (vmcnt(0) at the loop header, the
// state of the prologue merged into every trip).  The hardware returns LOADS in the order they were issued, so with a fixed
// number of loads per step "the record issued D steps ago has arrived" is exactly s_waitcnt vmcnt(D * loads per step);
// tying the registers through an empty asm after the wait keeps the compiler from using them earlier.
__device__ __forceinline__ const void* ct_uniform(const void* p) // (a wave-uniform pointer the compiler may not recognise as one)
{
    const unsigned long long v  = (unsigned long long)p;
    const unsigned           lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned           hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ v4i32 ct_load_quad(const void* sbase, unsigned voff)
{
    sbase = ct_uniform(sbase);
    v4i32 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
__device__ __forceinline__ double ct_load_T(const double* sbase, unsigned voff)
{
    const void* b = ct_uniform(sbase);
    double      r;
    asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(r) : "v"(voff), "s"(b) : "memory");
    return r;
}
__device__ __forceinline__ float ct_load_T(const float* sbase, unsigned voff)
{
    const void* b = ct_uniform(sbase);
    float       r;
    asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(r) : "v"(voff), "s"(b) : "memory");
    return r;
}
__device__ __forceinline__ int ct_load_int(const void* sbase, unsigned voff)
{
    sbase = ct_uniform(sbase);
    int r;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
__device__ __forceinline__ double ct_load_val(const double* p)
{
    double r;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ float ct_load_val(const float* p)
{
    float r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
// value of the lane below (row_shr:1 inside a row of 16 lanes; lane 0 of a row gets 0)
// (mov_dpp with bound_ctrl: no "old" value to set up before every move -- 3 instructions less per round of the lane chain)
__device__ __forceinline__ double ct_from_lane_below(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float ct_from_lane_below(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x111, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ void ct_wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct CtBases // counter values of the ticket streams before the launch
{
    unsigned v[16];
};

template <typename T, int NQ, int NGQ = 0>
struct CtStage
{
    v4i32 q[NQ]; // the row's record
    v4i32 gq[NGQ > 0 ? NGQ : 1]; // grouped form: the row record (in-group coefficients, diagonal)
    T     dg; // the diagonal, where it is kept outside the record
    int   g, tf; // uniform: index of the step's record; its tile number * 4 + flags (1 = a new step, 2 = last step of its tile)
};

template <typename T, int DMODE, bool HAS_OUT, int LPR, int WL, int DEPTH, bool PROF>
__global__ __launch_bounds__(128) void k_trsv_rec(int ntiles, CtDims dims, const v4i32* __restrict__ tile_desc,
                                                  const v4i32* __restrict__ step_rec, const int* __restrict__ ext_idx,
                                                  const v4i32* __restrict__ erec, const T* __restrict__ diag_sep,
                                                  const T* rhs_src,
                                                  const int* __restrict__ in_pairs,
                                                  const int* __restrict__ out_pairs, T* w, T* __restrict__ out,
                                                  unsigned* counter, CtBases bases, int nstreams, unsigned long long* prof_arg,
                                                  T* __restrict__ prefill, T* refill)
{
    constexpr bool GRP = (LPR == kGrpLPR); // grouped form (row groups; CtGRec): `diag_sep` holds the row records
    using L           = typename std::conditional<GRP, CtGRec<T>, CtRec<T, WL>>::type;
    using B           = typename Sentinel<T>::bits;
    static_assert(!GRP || WL == kGrpWL, "the grouped form has one lane-record shape");
    constexpr bool DSEP = L::DSEP && LPR == 1; // the diagonal in its own array (CtRec)
    constexpr int  NQ   = (DMODE != 0 && !DSEP) ? L::NQ : L::NQL;
    constexpr int  NGQ  = GRP ? CtGRec<T>::NGQ : 0;
    constexpr int R   = kCtRing;
    unsigned long long* const prof = PROF ? prof_arg : nullptr; // (diagnostic instantiation only: the counters cost scalar registers)
    extern __shared__ __attribute__((aligned(16))) char ct_lds[];
    // One region per ring slot: xs[slot * S] = 0 (the padding column), then row q of the tile at [1 + q] -- its right-hand side
    // value until the row's step has run (parked by the fetch wave), its solution afterwards --, then the tile's external
    // values at [1 + rows + j].  A column code c is therefore the element slot * S + c, whatever its kind: one add.
    T*        xs     = reinterpret_cast<T*>(ct_lds);
    const int S      = 1 + dims.rows + dims.exts;
    int*      posted = reinterpret_cast<int*>(xs + R * S); // tiles whose descriptor is in the ring
    int* tdone   = posted + 1; // tiles the compute wave has finished
    int* fetched = tdone + 1; // [R] external values parked so far
    int* tdesc   = fetched + R; // [R][4] {first step, steps, first position, rows}
    int* obase   = tdesc + 4 * R; // [R] smallest destination index of the tile (packed output list)
    const int smask = (1 << dims.sbits) - 1;
    const int tid = threadIdx.x;
    auto      uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    if(tid == 0)
    {
        *posted = 0;
        *tdone  = 0;
        for(int r = 0; r < R; ++r)
            xs[r * S] = (T)0;
    }
    __syncthreads();
    if(tid >= 64)
    {
        // ---------------- ticket / fetch wave
        const int lane = tid - 64;
        // ticket stream of this workgroup: tiles stream, stream + nstreams, ... in this order
        const int      stream = (int)(blockIdx.x % (unsigned)nstreams);
        unsigned*      cword  = counter + (size_t)stream * 1024;
        const unsigned base   = bases.v[stream];
        // (prof != nullptr: cycle counts per phase, tools/ diagnostics)
        unsigned long long pf_t0 = prof ? __builtin_amdgcn_s_memtime() : 0, pf_ring = 0, pf_ticket = 0, pf_idx = 0, pf_poll = 0;
        auto pf_flush = [&]() {
            if(prof && lane == 0)
            {
                atomicAdd(prof + 20 + (__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)) & 3), 1ull);
                atomicAdd(prof + 8, __builtin_amdgcn_s_memtime() - pf_t0);
                atomicAdd(prof + 9, pf_ring);
                atomicAdd(prof + 10, pf_ticket);
                atomicAdd(prof + 11, pf_idx);
                atomicAdd(prof + 12, pf_poll);
            }
        };
        // natural-order output of a finished tile: rows sorted by destination -> neighbouring lanes write neighbouring addresses
        auto write_back = [&](int sl) {
            const int p0 = uni(tdesc[4 * sl + 2]), nr = uni(tdesc[4 * sl + 3]);
            const T*  xb = xs + sl * S + 1;
            const v2i32* pr = reinterpret_cast<const v2i32*>(out_pairs) + p0;
            const int*   pk = out_pairs + p0;
            const int    ob = uni(obase[sl]);
            for(int q0 = 0; q0 < nr; q0 += 4 * 64)
            {
                v2i32 d[4];
                if(dims.out_packed)
                {
#pragma unroll
                    for(int u = 0; u < 4; ++u)
                    {
                        const int q = q0 + u * 64 + lane;
                        const int x = (q < nr) ? nt_load(pk + q) : 0;
                        d[u]        = (q < nr) ? v2i32{ob + (int)((unsigned)x >> dims.sbits), x & smask} : v2i32{-1, 0};
                    }
                }
                else
#pragma unroll
                for(int u = 0; u < 4; ++u)
                {
                    const int q = q0 + u * 64 + lane;
                    d[u]        = (q < nr) ? nt_load(pr + q) : v2i32{-1, 0};
                }
#pragma unroll
                for(int u = 0; u < 4; ++u)
                    if(d[u].x >= 0)
                        out[d[u].x] = xb[d[u].y];
            }
        };
        // Finished tiles are written back when this wave has nothing better to do -- while it waits for other tiles' values --
        // and at the latest before their ring slot is reused: done at that point only, the write-back of tile n - R sat in
        // front of the right-hand side and the external values of tile n, which the compute wave was about to wait for.
        int  wb_next = 0; // tiles [0, wb_next) of this workgroup are written back
        auto wait_done = [&](int want) {
            int spins = 0;
            while(uni(__hip_atomic_load(tdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want)
            {
                spin_guard(spins);
                __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
        };
        auto try_write_back = [&](int posted_tiles) -> bool { // one finished tile, if there is one
            if(!HAS_OUT || wb_next >= posted_tiles
               || uni(__hip_atomic_load(tdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= wb_next)
                return false;
            asm volatile("" ::: "memory");
            write_back(wb_next % R);
            wb_next = uni(wb_next + 1);
            return true;
        };
        for(int n = 0;; ++n)
        {
            const int slot = n % R;
            unsigned long long pf_a = prof ? __builtin_amdgcn_s_memtime() : 0;
            if(n >= R) // the slot was used by tile n - R
                wait_done(n - R + 1);
            if(HAS_OUT)
                while(wb_next <= n - R) // (finished: ring wait above) its values leave before the slot is reused
                {
                    write_back(wb_next % R);
                    wb_next = uni(wb_next + 1);
                }
            unsigned long long pf_b = prof ? __builtin_amdgcn_s_memtime() : 0;
            pf_ring += pf_b - pf_a;
            unsigned tk = 0;
            if(lane == 0)
                tk = __hip_atomic_fetch_add(cword, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
            const long long t64 = (long long)(unsigned)uni((int)tk) * nstreams + stream;
            const bool      end = t64 >= (long long)ntiles;
            const int       t   = end ? 0 : (int)t64;
            v4i32      d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
            if(!end)
            {
                d0 = tile_desc[2 * (size_t)t];
                d1 = tile_desc[2 * (size_t)t + 1];
            }
            if(lane == 0)
            {
                tdesc[4 * slot + 0] = d0.x;
                tdesc[4 * slot + 1] = end ? 0 : d0.y;
                if(!end)
                {
                    tdesc[4 * slot + 2] = d0.z;
                    tdesc[4 * slot + 3] = d0.w;
                    obase[slot]         = d1.y;
                }
                fetched[slot]       = -1; // (-1: the tile's right-hand side is not in LDS yet)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if(lane == 0)
                __hip_atomic_store(posted, n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if(end)
            {
                if(HAS_OUT) // the last tiles of this workgroup (their slots were never reused)
                    for(; wb_next < n; wb_next = uni(wb_next + 1))
                    {
                        wait_done(wb_next + 1);
                        write_back(wb_next % R);
                    }
                pf_flush();
                return;
            }
            const int e0 = uni(d1.z), e1 = e0 + uni(d1.w);
            if(prof)
                pf_ticket += __builtin_amdgcn_s_memtime() - pf_b;
            T*        exb = xs + slot * S + 1 + dims.rows;
            {
                // the tile's right-hand side: rows sorted by source index -> neighbouring lanes read neighbouring addresses
                // (whole lines where the tile covers contiguous pieces of the source vector); parked in row order
                const int p0 = uni(d0.z), nr = uni(d0.w);
                T*        rbb = xs + slot * S + 1;
                const v2i32* pr = reinterpret_cast<const v2i32*>(in_pairs) + p0;
                const int*   pk = in_pairs + p0;
                const int    ib = uni(d1.x);
                for(int q0 = 0; q0 < nr; q0 += 4 * 64)
                {
                    v2i32 d[4];
                    T     v[4];
                    if(dims.in_packed)
                    {
#pragma unroll
                        for(int u = 0; u < 4; ++u)
                        {
                            const int q = q0 + u * 64 + lane;
                            const int x = (q < nr) ? nt_load(pk + q) : 0;
                            d[u]        = (q < nr) ? v2i32{ib + (int)((unsigned)x >> dims.sbits), x & smask} : v2i32{-1, 0};
                        }
                    }
                    else
#pragma unroll
                    for(int u = 0; u < 4; ++u)
                    {
                        const int q = q0 + u * 64 + lane;
                        d[u]        = (q < nr) ? nt_load(pr + q) : v2i32{-1, 0};
                    }
#pragma unroll
                    for(int u = 0; u < 4; ++u)
                        if(d[u].x >= 0)
                            v[u] = rhs_src[d[u].x];
#pragma unroll
                    for(int u = 0; u < 4; ++u)
                        if(d[u].x >= 0)
                            rbb[d[u].y] = v[u];
                    // the right-hand side is the previous stage's w, and this is the one read of the element: the sentinel
                    // that stage's NEXT run needs goes in behind it (was k_fill_sentinel, a kernel of its own per solve)
                    if(refill)
                    {
                        const T sv = Sentinel<T>::from_bits(Sentinel<T>::value);
#pragma unroll
                        for(int u = 0; u < 4; ++u)
                            if(d[u].x >= 0)
                                nt_store(sv, refill + d[u].x);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if(lane == 0)
                    __hip_atomic_store(fetched + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // the sentinels of the NEXT stage's w (the upper solve after this lower one: same positions, another array):
                // this kernel waits on its tile wavefront a quarter of its time and has the bandwidth to spare, the fill as a
                // kernel of its own (0.19 ms at 512^3) has not
                if(prefill)
                {
                    const T sv = Sentinel<T>::from_bits(Sentinel<T>::value);
                    for(int q = lane; q < nr; q += 64)
                        nt_store(sv, prefill + p0 + q);
                }
            }
            for(int e = e0; e < e1; e += 64 * kCtFetchDepth)
            {
                int idx[kCtFetchDepth];
                B   bits[kCtFetchDepth];
#pragma unroll
                for(int u = 0; u < kCtFetchDepth; ++u)
                {
                    const int j = e + u * 64 + lane;
                    idx[u]      = (j < e1) ? nt_load(ext_idx + j) : -1;
                }
                const int nbatch = min(kCtFetchDepth, (e1 - e + 63) / 64);
                int       next = 0, spins = 0, backoff = 1;
                unsigned long long pf_c = 0;
                if(prof)
                {
                    pf_c = __builtin_amdgcn_s_memtime();
                    int keep = idx[0];
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(keep)::"memory");
                    idx[0] = keep;
                    pf_idx += __builtin_amdgcn_s_memtime() - pf_c;
                    pf_c = __builtin_amdgcn_s_memtime();
                }
                int done = 0; // values of batch `next` already handed over
                // Every poll is a read from memory (agent scope: nothing of it stays in a cache), whole lines per load
                // instruction: all batches are requested once, afterwards only the lanes of the batch being handed over whose
                // value has not arrived ask again (re-reading everything on every round was 2/3 of the external traffic).
#pragma unroll
                for(int u = 0; u < kCtFetchDepth; ++u)
                    bits[u] = (idx[u] >= 0) ? poll_load(w + idx[u]) : (B)0;
                bool first_round = true;
                while(next < nbatch)
                {
                    if(!first_round)
                    {
#pragma unroll
                        for(int u = 0; u < kCtFetchDepth; ++u)
                            if(u == next && idx[u] >= 0 && bits[u] == Sentinel<T>::value)
                                bits[u] = poll_load(w + idx[u]);
                    }
                    first_round   = false;
                    bool advanced = false;
#pragma unroll
                    for(int u = 0; u < kCtFetchDepth; ++u)
                        if(u == next && u < nbatch)
                        {
                            // hand over every value up to the first missing one (use order): the tile starts as soon as what its
                            // first steps need is there, not when 64 values -- several steps' worth -- are complete
                            const bool               missing = idx[u] >= 0 && bits[u] == Sentinel<T>::value;
                            const unsigned long long mm      = __ballot(missing);
                            const int                nb      = min(64, e1 - (e + u * 64));
                            const int                ready   = min(nb, mm ? (int)__builtin_ctzll(mm) : 64);
                            if(ready > done)
                            {
                                if(lane >= done && lane < ready)
                                {
                                    exb[e - e0 + u * 64 + lane] = Sentinel<T>::from_bits(bits[u]);
                                    idx[u]                      = -1; // parked: not polled again
                                }
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                if(lane == 0)
                                    __hip_atomic_store(fetched + slot, e - e0 + u * 64 + ready, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                                done     = ready;
                                advanced = true;
                            }
                            if(ready == nb)
                            {
                                ++next;
                                done = 0;
                            }
                        }
                    if(!advanced && try_write_back(n)) // (tiles 0 .. n - 1 of this workgroup have their slots filled in)
                        advanced = true;
                    if(!advanced)
                    {
                        spin_guard(spins);
                        // (sleep between polls: doubling, capped low -- the wait for a neighbour tile's values is on the critical
                        //  path of the solve, half a back-off period of it on average per tile hop)
                        for(int z = 0; z < backoff; ++z)
                            __builtin_amdgcn_s_sleep(1);
                        backoff = backoff < RAMD_CT_POLL_CAP ? backoff * 2 : RAMD_CT_POLL_CAP;
                    }
                }
                if(prof)
                    pf_poll += __builtin_amdgcn_s_memtime() - pf_c;
            }
        }
    }
    // ---------------- compute wave: one stream of steps over all the tiles this workgroup gets
    // (its steps are the critical path of the solve: issue priority over the polling waves that share the SIMD)
    __builtin_amdgcn_s_setprio(RAMD_CT_PRIO);
    const int lane = tid;
    // iterator over the steps: `cur` is the record of the next step to fetch (loaded one fetch ahead through the scalar cache)
    unsigned long long pc_t0 = prof ? __builtin_amdgcn_s_memtime() : 0, pc_ext = 0, pc_post = 0, pc_steps = 0, pc_dups = 0;
    int   g = 0, gend = 0, tn = 0, pending = 0, done_tiles = 0;
    bool  ending = false, cur_fresh = false, cur_last = false;
    v4i32 cur = {0, 1, 0, 0};
    auto  wait_posted = [&](int want) {
        int spins = 0;
        while(uni(__hip_atomic_load(posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want)
        {
            spin_guard(spins);
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    auto load_cur = [&]() {
        const v4i32 r = step_rec[g];
        cur           = v4i32{uni(r.x), uni(r.y), uni(r.z), uni(r.w)};
        cur_last      = (g + 1 == gend);
    };
    // moves to the next step; false: nothing new (the next tile is not posted yet and real steps are still in flight, or
    // there are no more tiles) -- the caller then repeats the current step as a harmless duplicate
    auto try_advance = [&]() -> bool {
        if(ending)
            return false;
        if(g + 1 < gend)
        {
            g = uni(g + 1);
            load_cur();
            return true;
        }
        if(uni(__hip_atomic_load(posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < tn + 2)
        {
            if(pending > 0)
                return false;
            const unsigned long long a = prof ? __builtin_amdgcn_s_memtime() : 0;
            wait_posted(tn + 2);
            if(prof)
                pc_post += __builtin_amdgcn_s_memtime() - a;
        }
        asm volatile("" ::: "memory");
        const int s   = (tn + 1) % R;
        const int S0  = uni(tdesc[4 * s + 0]);
        const int nst = uni(tdesc[4 * s + 1]);
        if(nst == 0)
        {
            ending = true;
            return false;
        }
        tn   = uni(tn + 1);
        g    = S0;
        gend = S0 + nst;
        load_cur();
        return true;
    };
    // vector memory operations of one step of the stream, in issue order: the store of the step's values, then the quads of
    // the record DEPTH steps ahead
    constexpr int OPS = 1 + NQ + NGQ + ((DSEP && DMODE != 0) ? 1 : 0);
    static_assert(DEPTH * OPS < 64, "the whole prefetch window has to fit the 6-bit counter");
    auto fetch_rec = [&](CtStage<T, NQ, NGQ>& st) {
        // (the record itself comes back through the scalar cache when the step runs: scalar registers are scarce.  Carrying
        //  it with the stage in four vector registers instead -- no scalar load left in the step, whose LDS waits are waits on
        //  lgkmcnt(0) and so wait for that load too -- was measured: 512 x 512 x 64 slab 3.27 vs 3.34 ms per GMRES iteration,
        //  full cube 60.2 vs 62.1 it/s, FE surrogate 199 vs 200.5: not the step's bottleneck, and the registers cost more
        //  where the solve is bandwidth-bound; tools/r04_runs/zp.sh.  Neither does it help to request both of a step's records
        //  -- this one and the next step's -- in the MIDDLE of the step, behind its LDS reads, so that no scalar load is in
        //  flight at the step's first LDS wait: slab 3.35 vs 3.33 ms, cube 63.1 vs 63.4 it/s, surrogate 201.4 vs 200.5;
        //  tools/r04_runs/zx.sh.  The scalar-cache round trips are not what a step waits for.)
        st.g  = g;
        st.tf = tn * 4 + ((cur_fresh ? 1 : 0) | (cur_last ? 2 : 0));
        const int      nl  = (cur.y & 0xff) * LPR; // lane records of the step
        const int      row = min(lane, nl - 1);
        const v4i32*   qb  = erec + (size_t)((DSEP ? L::NQS : L::NQ) * LPR) * (size_t)cur.x; // scalar base of the step + lane offsets
        const unsigned ro  = (unsigned)row * 16u;
        // (lanes beyond the step's rows mirror its last row -- same addresses, same values: every operation is issued, and
        //  counted, in every step; switching those lanes off instead was measured no faster)
#pragma unroll
        for(int q = 0; q < NQ; ++q)
            st.q[q] = ct_load_quad(qb, ro + (unsigned)(q * nl) * 16u);
        if(DSEP && DMODE != 0)
            st.dg = ct_load_T(diag_sep + cur.x, (unsigned)row * (unsigned)sizeof(T));
        if(GRP) // the row record: the lanes of a row read the same 16-byte words
        {
            const unsigned go = ((unsigned)row / (unsigned)LPR) * (unsigned)(kGrpMax * sizeof(T));
#pragma unroll
            for(int q = 0; q < NGQ; ++q)
                st.gq[q] = ct_load_quad(diag_sep + (size_t)kGrpMax * (size_t)cur.x, go + (unsigned)q * 16u);
        }
        if(cur_fresh)
            pending = uni(pending + 1);
        cur_fresh = try_advance();
    };
    auto arrived = [&](CtStage<T, NQ, NGQ>& st) { // (after the wait) the registers of the stage may be used from here on
#pragma unroll
        for(int q = 0; q < NQ; ++q)
            asm volatile("" : "+v"(st.q[q]));
#pragma unroll
        for(int q = 0; q < NGQ; ++q)
            asm volatile("" : "+v"(st.gq[q]));
        if(DSEP && DMODE != 0)
            asm volatile("" : "+v"(st.dg));
    };
    int  have = -1, have_tn = -1;
    auto step = [&](const CtStage<T, NQ, NGQ>& stq, const v4i32 rec) {
        struct
        {
            int pos, cnt, need, lbase, tn, flags;
        } st = {uni(rec.x), uni(rec.y) & 0xff, uni(rec.z), uni(rec.w), uni(stq.tf) >> 2, uni(stq.tf) & 3};
        const int maxg = (uni(rec.y) >> 8) & 0xff; // grouped form: rows of the step's largest group
        // rows of the step whose value another tile reads: [0, nexp) -- all of them where w is also the stage's result
        // (at least 1: the publication store is one of the step's hand-counted vector memory operations -- it has to be ISSUED
        //  in every step, i.e. with at least one active lane; see k_ct_step_nexp)
        const int nexp = dims.mask_pub ? max(1, (uni(rec.y) >> 16) & 0xff) : 64;
        const int nl   = st.cnt * LPR; // lane records of the step
        const int lrec = min(lane, nl - 1);
        const int row  = (int)((unsigned)lrec / (unsigned)LPR); // row of the step this lane works for
        const int sub  = (int)((unsigned)lrec % (unsigned)LPR); // its place among the row's lanes
        const int slot = st.tn % R;
        have    = (st.tn != have_tn) ? -1 : have; // (-1: until the fetch wave has parked the tile's right-hand side)
        have_tn = st.tn;
        if(prof)
        {
            if(st.flags & 1)
                ++pc_steps;
            else
                ++pc_dups;
        }
        if(__builtin_expect(have < st.need, 0)) // wave-uniform: wait for the fetch wave (an LDS count, no memory round trip)
        {
            const unsigned long long a = prof ? __builtin_amdgcn_s_memtime() : 0;
            int spins = 0;
            while((have = uni(__hip_atomic_load(fetched + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) < st.need)
            {
                spin_guard(spins);
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            if(prof)
                pc_ext += __builtin_amdgcn_s_memtime() - a;
        }
        {
        // every dword of the record counts as used here: a dword nobody reads would be handed out as a temporary while
        // its load is still in flight, and the write-after-write hazard costs a drain of the prefetch
#pragma unroll
        for(int q = 0; q < NQ; ++q)
            asm volatile("" ::"v"(stq.q[q]));
#pragma unroll
        for(int q = 0; q < NGQ; ++q)
            asm volatile("" ::"v"(stq.gq[q]));
        const int  sbase = slot * S; // this tile's region
        const int  own   = sbase + 1 + st.lbase + row;
        const T    bval = xs[own]; // right-hand side (a repeated step finds its result there)
        T          v[WL], a[WL];
#pragma unroll
        for(int k = 0; k < WL; ++k)
        {
            const int word = stq.q[(k / 2) / 4][(k / 2) % 4];
            const int c    = (k & 1) ? (int)((unsigned)word >> 16) : (word & 0xffff);
            const int at   = sbase + c;
            v[k]           = xs[at];
            if(sizeof(T) == 8)
            {
                const int wi = (L::off_val + 8 * k) / 4;
                a[k]         = (T)__hiloint2double(stq.q[(wi + 1) / 4][(wi + 1) % 4], stq.q[wi / 4][wi % 4]);
            }
            else
            {
                const int wi = (L::off_val + 4 * k) / 4;
                a[k]         = (T)__int_as_float(stq.q[wi / 4][wi % 4]);
            }
        }
        T sum = bval;
        {
        if constexpr(GRP)
        {
            // One step = whole row groups.  The out-of-group entries of a row are summed by its LPR lanes exactly as in the
            // ungrouped form (lane chain in storage order); the in-group entries -- coefficients gq[r] of the group's row r --
            // are subtracted in the order of the host loop with the FINAL value of row r, fetched from that row's last lane by
            // a lane permute.  The value that counts is the one of the row's last lane (in-last) / first lane (in-first).
            const int meta = stq.q[0][3];
            const int gi   = meta & 0xff; // this row's index in its group
            const int gm   = meta >> 8; // bit r: the row has a coefficient for group row r
            auto coef = [&](int r) -> T {
                if(sizeof(T) == 8)
                    return (T)__hiloint2double(stq.gq[(2 * r + 1) / 4][(2 * r + 1) % 4], stq.gq[(2 * r) / 4][(2 * r) % 4]);
                return (T)__int_as_float(stq.gq[r / 4][r % 4]);
            };
            const T dg = coef(kGrpMax - 1);
            auto finalize = [&](T s) -> T { return DMODE == 0 ? s : (DMODE == 1 ? s / dg : s * dg); };
            // last lane of group row r of this lane's group
            auto from_group_row = [&](int r, T v) -> T {
                const int src = ((row - gi + r) * LPR + (LPR - 1)) * 4;
                if(sizeof(T) == 8)
                {
                    const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint((double)v));
                    const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint((double)v));
                    return (T)__hiloint2double(hi, lo);
                }
                return (T)__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int((float)v)));
            };
            T pr[WL];
#pragma unroll
            for(int k = 0; k < WL; ++k)
                pr[k] = a[k] * v[k];
            auto chain = [&](T first) -> T { // the row's out-of-group entries in storage order, starting from `first` in lane 0
                T s = first;
#pragma unroll
                for(int t = 0; t < LPR; ++t)
                {
                    T acc = (t == 0) ? first : ct_from_lane_below(s);
#pragma unroll
                    for(int k = 0; k < WL; ++k)
                        acc -= pr[k];
                    s = acc;
                }
                return s;
            };
            if(!dims.infirst)
            {
                // lower solve: out-of-group entries first (all rows of the step at once), then round r hands the finished
                // row r of every group to the rows behind it
                sum = chain(bval);
#pragma unroll
                for(int r = 0; r < kGrpMax - 1; ++r)
                {
                    if(r + 1 < maxg) // (wave-uniform)
                    {
                        const T y = from_group_row(r, finalize(sum));
                        const T d = sum - coef(r) * y;
                        sum       = (gi > r && ((gm >> r) & 1)) ? d : sum;
                    }
                }
                sum = finalize(sum);
            }
            else
            {
                // upper solve: a row starts with its in-group entries (nearest row first), so the rows of a group run one
                // after the other; the groups of the step side by side
                T yr[kGrpMax - 1];
                T fin = bval;
#pragma unroll
                for(int g2 = 0; g2 < kGrpMax; ++g2)
                {
                    if(g2 < maxg) // (wave-uniform)
                    {
                        if(g2 > 0)
                            yr[g2 - 1] = from_group_row(g2 - 1, fin);
                        T acc0 = bval;
#pragma unroll
                        for(int r = g2 - 1; r >= 0; --r)
                        {
                            const T d = acc0 - coef(r) * yr[r];
                            acc0      = ((gm >> r) & 1) ? d : acc0;
                        }
                        const T f = finalize(chain(acc0));
                        fin       = (gi == g2) ? f : fin;
                    }
                }
                sum = fin;
            }
        }
        else if(LPR == 1)
        {
#pragma unroll
            for(int k = 0; k < WL; ++k)
                sum -= a[k] * v[k]; // (padding: value 0 * the slot's zero element)
        }
        else
        {
            // the row's entries are subtracted in storage order, as the host loop does: lane 0 of the row starts from the
            // right-hand side and takes its WL entries, hands the running value to lane 1 (DPP, no LDS), ... -- every lane
            // runs every round, the value that counts is the one of lane `t` in round t; the last lane ends up with the row
            T pr[WL];
#pragma unroll
            for(int k = 0; k < WL; ++k)
                pr[k] = a[k] * v[k];
#pragma unroll
            for(int t = 0; t < LPR; ++t)
            {
                T acc = (t == 0) ? bval : ct_from_lane_below(sum);
#pragma unroll
                for(int k = 0; k < WL; ++k)
                    acc -= pr[k];
                sum = acc;
            }
        }
        if(DMODE != 0 && !GRP)
        {
            T dg;
            if(DSEP)
                dg = stq.dg;
            else if(sizeof(T) == 8)
            {
                const int wi = L::off_diag / 4;
                dg           = (T)__hiloint2double(stq.q[(wi + 1) / 4][(wi + 1) % 4], stq.q[wi / 4][wi % 4]);
            }
            else
                dg = (T)__int_as_float(stq.q[(L::off_diag / 4) / 4][(L::off_diag / 4) % 4]);
            if(DMODE == 1)
                sum /= dg;
            else
                sum = sum * dg;
        }
        sum = (st.flags & 1) ? sum : bval; // (a repeated step finds its result in place of the right-hand side: re-store it)
        if(LPR == 1 || (sub == LPR - 1 && lane < nl))
            xs[own] = sum;
        }
        // (LPR = 1: lanes beyond the step's rows repeat its last row; LPR > 1: only the last lane of a row holds it)
        if(kCtWSlot<WL, LPR>)
        {
            if(row < nexp)
                publish(w + (st.pos - st.lbase) + (int)((unsigned)stq.q[0][1] >> 16), sum);
        }
        else if((LPR == 1 || (sub == LPR - 1 && lane < nl)) && row < nexp)
            publish(w + st.pos + row, sum);
        }
        // this step's LDS traffic before the next step's: one wave, in-order LDS queue
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pending = uni(pending - (st.flags & 1));
        {
            if(__builtin_expect((st.flags & 3) == 3, 0)) // the tile is finished: its ring slot may be reused
            {
                done_tiles = uni(done_tiles + 1);
                if(lane == 0)
                    __hip_atomic_store(tdone, done_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    // first tile
    wait_posted(1);
    {
        const int S0  = uni(tdesc[0]);
        const int nst = uni(tdesc[1]);
        if(nst == 0)
            return;
        g    = S0;
        gend = S0 + nst;
        load_cur();
        cur_fresh = true;
    }
    CtStage<T, NQ, NGQ> st[DEPTH];
#pragma unroll
    for(int j = 0; j < DEPTH; ++j)
        fetch_rec(st[j]);
    ct_wait_vm<0>(); // (the stream starts with nothing in flight: every steady-state wait below is then sufficient)
    v4i32 rec_next = step_rec[uni(st[0].g)];
    for(;;)
    {
#pragma unroll
        for(int j = 0; j < DEPTH; ++j)
        {
            const v4i32 rec = rec_next; // (requested during the step before)
            rec_next        = step_rec[uni(st[(j + 1) % DEPTH].g)];
            // this step's quads were loaded DEPTH steps ago: the LOADS of DEPTH - 1 steps were issued after them.  (Not their
            // stores: a store may be acknowledged before an older load's data is there -- only loads return in issue order;
            // counting the step's store as well, as rounds 3 to 5 did, waits for too little when loads are slow.  See
            // trsv_lattice.hip LatSched, where a shared device showed it.)
            ct_wait_vm<(DEPTH - 1) * (OPS - 1)>();
            arrived(st[j]);
            step(st[j], rec);
            fetch_rec(st[j]);
        }
        if(ending && pending == 0)
            break;
    }
    if(prof && lane == 0)
    {
        atomicAdd(prof + 16 + (__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)) & 3), 1ull); // SIMD of the compute wave
        atomicAdd(prof + 0, __builtin_amdgcn_s_memtime() - pc_t0);
        atomicAdd(prof + 1, pc_ext);
        atomicAdd(prof + 2, pc_post);
        atomicAdd(prof + 3, pc_steps);
        atomicAdd(prof + 4, pc_dups);
        atomicAdd(prof + 5, 1ull);
    }
}

static bool ct_enabled()
{
    static int on = -1;
    if(on < 0)
    {
        const char* e = getenv("RAMD_TRSV_CT"); // 0: always the level-scheduled kernel (A/B experiments)
        on            = e ? atoi(e) : 1;
    }
    return on != 0;
}

// builds the box-tile form of a plan; RAMD_ERR_UNSUPPORTED: the matrix has no chains worth it, or its tiles do not fit
// the LDS budget (caller falls back to the level-scheduled form)
template <typename T>
static int build_ct_plan(ramd_mat_s* m, TriState* st, TriPlan* P, bool lower, bool reverse)
{
    Backend&  b = backend();
    const int n = m->nrow;
    static int min_rows = -1, min_len = -1, rows_target = -1; // RAMD_TRSV_CT_MINROWS / _MINLEN = 0: force this form (tests)
    if(min_rows < 0)
    {
        min_rows    = getenv("RAMD_TRSV_CT_MINROWS") ? atoi(getenv("RAMD_TRSV_CT_MINROWS")) : 4096;
        min_len     = getenv("RAMD_TRSV_CT_MINLEN") ? atoi(getenv("RAMD_TRSV_CT_MINLEN")) : 8;
        rows_target = getenv("RAMD_TRSV_CT_ROWS") ? atoi(getenv("RAMD_TRSV_CT_ROWS")) : 512;
    }
    static const bool verbose    = getenv("RAMD_TRSV_CT_VERBOSE") != nullptr;
    static const int  lds_budget = getenv("RAMD_TRSV_CT_LDS") ? atoi(getenv("RAMD_TRSV_CT_LDS")) : 40 * 1024;
    g_ct_gave_up = 0;
    if(n < min_rows || n < 1)
    {
        g_ct_gave_up = 7;
        return RAMD_ERR_UNSUPPORTED;
    }
    int *level = nullptr, *lorder = nullptr, *start = nullptr, *lev_t = nullptr, *tkey = nullptr, *o1 = nullptr,
        *k2 = nullptr, *o2 = nullptr, *tflag = nullptr, *sflag = nullptr, *tscan = nullptr, *sscan = nullptr,
        *tile_of = nullptr, *step_of = nullptr, *step_w = nullptr, *nodiag = nullptr, *cext = nullptr, *tsz = nullptr,
        *ref_start = nullptr, *slot_of_ref = nullptr, *gf = nullptr, *gl = nullptr, *tposv = nullptr, *step_maxg = nullptr,
        *wmap = nullptr;
    UnitPlan units;
    unsigned long long* word = nullptr;
    int  nlev = 0;
    int  s    = RAMD_OK;
    auto cleanup = [&]() {
        dev_free(&level);
        dev_free(&lorder);
        dev_free(&start);
        dev_free(&lev_t);
        dev_free(&tkey);
        dev_free(&o1);
        dev_free(&k2);
        dev_free(&o2);
        dev_free(&tflag);
        dev_free(&sflag);
        dev_free(&tscan);
        dev_free(&sscan);
        dev_free(&tile_of);
        dev_free(&step_of);
        dev_free(&step_w);
        dev_free(&nodiag);
        dev_free(&cext);
        dev_free(&tsz);
        dev_free(&word);
        dev_free(&ref_start);
        dev_free(&slot_of_ref);
        dev_free(&gf);
        dev_free(&gl);
        dev_free(&tposv);
        dev_free(&step_maxg);
        dev_free(&wmap);
        units.release();
    };
#define CT_TRY(expr)     \
    do                   \
    {                    \
        s = (expr);      \
        if(s != RAMD_OK) \
        {                \
            cleanup();   \
            P->release(); \
            return s;    \
        }                \
    } while(0)
#define CT_HIP(expr)                 \
    do                               \
    {                                \
        if((expr) != hipSuccess)     \
        {                            \
            cleanup();               \
            P->release();            \
            RAMD_FAIL(RAMD_ERR_HIP, #expr); \
        }                            \
    } while(0)
#define CT_GIVE_UP(why)                                                                    \
    do                                                                                     \
    {                                                                                      \
        g_ct_gave_up = (why);                                                              \
        if(verbose)                                                                        \
            fprintf(stderr, "box-tile plan: not used for this matrix (%s; trisolve.hip:%d)\n", ct_why_text(why), __LINE__); \
        cleanup();                                                                         \
        P->release();                                                                      \
        return RAMD_ERR_UNSUPPORTED;                                                       \
    } while(0)
    const int      grid = ew_grid(n + 1);
    const unsigned nb1  = (unsigned)(((int64_t)n + 1 + kBlock - 1) / kBlock);
    const unsigned nb   = nblocks_of(n);
    build_mark(nullptr);
    // chains
    CT_TRY(dev_alloc(&start, (int64_t)n + 1));
    if(lower)
        hipLaunchKernelGGL((k_ct_chain_start<true>), dim3(nb1), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, start);
    else
        hipLaunchKernelGGL((k_ct_chain_start<false>), dim3(nb1), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, start);
    CT_TRY(device_exclusive_scan(start, start, (int64_t)n + 1)); // start[t] = chains begun before t
    int nchains = 0;
    CT_HIP(hipMemcpyAsync(&nchains, start + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    CT_HIP(hipStreamSynchronize(b.cur));
    if(nchains <= 0 || (int64_t)n < (int64_t)min_len * nchains)
        CT_GIVE_UP(1);
    // longest strictly-triangular row: decides how many lanes share a row (and so how many rows a step may hold)
    int wmax = 0;
    CT_TRY(dev_alloc(&cext, 4));
    CT_HIP(hipMemsetAsync(cext, 0, sizeof(int) * 4, b.cur));
    if(lower)
        hipLaunchKernelGGL((k_ct_row_wmax<true>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, cext + 3);
    else
        hipLaunchKernelGGL((k_ct_row_wmax<false>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, cext + 3);
    CT_HIP(hipMemcpyAsync(&wmax, cext + 3, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    CT_HIP(hipStreamSynchronize(b.cur));
    // row groups (supernodes): only where several lanes share a row anyway (long rows), and not for the descending-order
    // sweep over the lower part (no caller)
    static const int grp_env = getenv("RAMD_TRSV_CT_GROUPS") ? atoi(getenv("RAMD_TRSV_CT_GROUPS")) : 1; // (0: off; A/B experiments)
    bool grp = false;
    int  grp_count = 0, grp_maxsize = 1;
    if(grp_env != 0 && wmax > 8 && wmax <= kGrpLPR * kGrpWL + kGrpMax - 1 && !(lower && reverse))
    {
        int *brk = nullptr, *bscan = nullptr, *rstart = nullptr, *gs = nullptr, *gscan = nullptr, *gstart = nullptr;
        auto drop = [&]() {
            dev_free(&brk);
            dev_free(&bscan);
            dev_free(&rstart);
            dev_free(&gs);
            dev_free(&gscan);
            dev_free(&gstart);
        };
#define CT_TRYG(expr)    \
    do                   \
    {                    \
        s = (expr);      \
        if(s != RAMD_OK) \
        {                \
            drop();      \
            CT_TRY(s);   \
        }                \
    } while(0)
        CT_TRYG(dev_alloc(&brk, (int64_t)n + 1));
        CT_TRYG(dev_alloc(&bscan, (int64_t)n + 1));
        if(lower)
            hipLaunchKernelGGL((k_ct_sn_breaks<true>), dim3(nb1), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, brk);
        else
            hipLaunchKernelGGL((k_ct_sn_breaks<false>), dim3(nb1), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, brk);
        CT_TRYG(device_exclusive_scan(brk, bscan, (int64_t)n + 1));
        int nruns = 0;
        if(hipMemcpyAsync(&nruns, bscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
           || hipStreamSynchronize(b.cur) != hipSuccess)
            CT_TRYG(RAMD_ERR_HIP);
        if(nruns > 0 && (int64_t)nruns * 3 <= (int64_t)n * 2) // (mean run of at least 1.5 rows: otherwise not worth the wider records)
        {
            CT_TRYG(dev_alloc(&rstart, (int64_t)nruns + 1));
            hipLaunchKernelGGL(k_ct_scatter_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, brk, bscan, rstart);
            CT_TRYG(dev_alloc(&gs, (int64_t)n + 1));
            CT_TRYG(dev_alloc(&gscan, (int64_t)n + 1));
            hipLaunchKernelGGL(k_ct_group_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, brk, bscan, rstart, gs);
            CT_TRYG(device_exclusive_scan(gs, gscan, (int64_t)n + 1));
            int ngroups = 0;
            if(hipMemcpyAsync(&ngroups, gscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
               || hipStreamSynchronize(b.cur) != hipSuccess)
                CT_TRYG(RAMD_ERR_HIP);
            CT_TRYG(dev_alloc(&gstart, (int64_t)ngroups + 1));
            hipLaunchKernelGGL(k_ct_scatter_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, gs, gscan, gstart);
            if(hipMemcpyAsync(gstart + ngroups, &n, sizeof(int), hipMemcpyHostToDevice, b.cur) != hipSuccess)
                CT_TRYG(RAMD_ERR_HIP);
            CT_TRYG(dev_alloc(&gf, n));
            CT_TRYG(dev_alloc(&gl, n));
            hipLaunchKernelGGL(k_ct_group_bounds, dim3(grid), dim3(kBlock), 0, b.cur, n, gs, gscan, gstart, gf, gl);
            // the out-of-group part of every row has to fit kGrpLPR x kGrpWL entries
            int wout = 0;
            CT_HIP(hipMemsetAsync(cext + 2, 0, 2 * sizeof(int), b.cur));
            if(lower)
                hipLaunchKernelGGL((k_ct_row_wmax_out<true>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, gf, gl,
                                   cext + 3, cext + 2);
            else
                hipLaunchKernelGGL((k_ct_row_wmax_out<false>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, gf, gl,
                                   cext + 3, cext + 2);
            int two[2] = {0, 0};
            if(hipMemcpyAsync(two, cext + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
               || hipStreamSynchronize(b.cur) != hipSuccess)
                CT_TRYG(RAMD_ERR_HIP);
            wout        = two[1];
            grp_maxsize = two[0] < 1 ? 1 : two[0];
            grp_count   = ngroups;
            grp = wout <= kGrpLPR * kGrpWL;
            if(verbose)
                fprintf(stderr, "box-tile plan: %d supernode runs, %d row groups of <= %d rows, longest out-of-group part %d%s\n",
                        nruns, ngroups, kGrpMax, wout, grp ? "" : " (too long: ungrouped form)");
            if(!grp)
            {
                dev_free(&gf);
                dev_free(&gl);
            }
            // supernodes whose rows are too long for the grouped record: as single rows they are five times as many levels (the
            // state of round 2), and the analysis sweeps of that form cost 0.5 s on a front-ordered shell before its tiles turn out
            // not to fit -- the sync-free grouped form keeps the groups and takes rows of up to 48 entries outside them
            static const int sf_on = getenv("RAMD_TRSV_SF") ? atoi(getenv("RAMD_TRSV_SF")) : 1;
            if(!grp && sf_on != 0 && !reverse && wout <= 8 * 6 && n >= 4096)
            {
                drop();
                CT_GIVE_UP(9);
            }
        }
        drop();
#undef CT_TRYG
    }
    build_mark("plan: chains, row widths, groups");
    if(wmax > 32 && !grp) // (8 lanes x 4 entries per row and step) -- known before the coordinate sweep, which such a matrix is spared
        CT_GIVE_UP(3);
    // monotone coordinates (sync-free sweep) and their extents
    CT_TRY(dev_alloc(&word, n));
    CT_HIP(hipMemsetAsync(word, 0, sizeof(unsigned long long) * (size_t)n, b.cur));
    CT_HIP(hipMemsetAsync(cext, 0, sizeof(int) * 4, b.cur));
    {
        // same block graph as the level sweep: blocks in hyperplane order fill the machine (natural order: only the ~2048
        // resident blocks -- two grid planes at 512^3 -- are in flight, the lines of a plane a serial chain inside that window)
        CT_TRY(unit_schedule(m, lower, &units)); // (kept for the level sweep below)
        build_mark("plan: unit schedule");
        const unsigned nbs = sweep_blocks(units.nunits);
        if(lower)
            hipLaunchKernelGGL((k_ct_coords<true>), dim3(nbs), dim3(sweep_block_size()), 0, b.cur, n, m->rp, m->ci, start, word,
                               cext, st->counter, st->ticket, unit_view(units), gf, gl, sweep_poll_cap());
        else
            hipLaunchKernelGGL((k_ct_coords<false>), dim3(nbs), dim3(sweep_block_size()), 0, b.cur, n, m->rp, m->ci, start, word,
                               cext, st->counter, st->ticket, unit_view(units), gf, gl, sweep_poll_cap());
        st->ticket += nbs;
    }
    int hext[3] = {0, 0, 0};
    CT_HIP(hipMemcpyAsync(hext, cext, sizeof(int) * 3, hipMemcpyDeviceToHost, b.cur));
    CT_HIP(hipStreamSynchronize(b.cur));
    build_mark("plan: coordinates sweep");
    dev_free(&start);
    // box sizes: b_k proportional to the extent E_k of every non-trivial coordinate (equal depth of the tile DAG in
    // every direction), about `rows` rows per tile; rows = what the LDS budget (~40 KB) allows for this row length
    int64_t ntri = 0;
    {
        // strictly-triangular entries: (nnz - diagonal) / 2 for a symmetric pattern; nnz as the safe estimate otherwise
        ntri = (m->nnz > n) ? (m->nnz - n) / 2 : m->nnz;
    }
    (void)ntri;
    int rows = rows_target < 32 ? 32 : rows_target; // (LDS holds only the tile's own values and its external ones)
    const int64_t E[3] = {(int64_t)hext[0] + 1, (int64_t)hext[1] + 1, (int64_t)hext[2] + 1};
    int           dnz  = 0;
    for(int k = 0; k < 3; ++k)
        dnz += E[k] > 1 ? 1 : 0;
    if(dnz == 0)
        CT_GIVE_UP(2);
    const int lpr = grp ? kGrpLPR : (wmax > 8 ? 8 : 1);
    const int wl  = grp ? kGrpWL : (lpr == 1 ? (wmax <= 3 ? 3 : (wmax <= 4 ? 4 : 8)) : 4);
    const int rpp = 64 / lpr;
    int       bs[3] = {1, 1, 1}, Ts[3] = {1, 1, 1};
    int       ntiles = 0, nsteps = 0, total = 0, next = 0;
    int64_t keymax = 0;
    // levels (natural row index): once
    if(grp)
    {
        // levels of the quotient graph, in sweep order
        CT_TRY(dev_alloc(&level, n));
        CT_HIP(hipMemsetAsync(level, 0, sizeof(int) * (size_t)n, b.cur));
        const unsigned nbs = sweep_blocks(units.nunits);
        if(lower)
            hipLaunchKernelGGL((k_ct_glevels<true>), dim3(nbs), dim3(sweep_block_size()), 0, b.cur, n, m->rp, m->ci, gf, gl, level,
                               st->counter, st->ticket, unit_view(units), sweep_poll_cap());
        else
            hipLaunchKernelGGL((k_ct_glevels<false>), dim3(nbs), dim3(sweep_block_size()), 0, b.cur, n, m->rp, m->ci, gf, gl, level,
                               st->counter, st->ticket, unit_view(units), sweep_poll_cap());
        st->ticket += nbs;
        s = device_max_int(level, n, &nlev); // (synchronises)
        CT_TRY(s);
    }
    else if(lower && st->l_level_cache) // the sweep ILU0Factorize ran on the same pattern
    {
        level             = st->l_level_cache;
        st->l_level_cache = nullptr;
        nlev              = st->l_nlev_cache;
    }
    else
    {
        CT_TRY(level_order(m, st, lower, nullptr, &nlev, &level, &units));
    }
    CT_HIP(hipStreamSynchronize(b.cur));
    units.release();
    build_mark("plan: levels");
    bool fits = false;
    for(int attempt = 0; attempt < 5 && !fits; ++attempt)
    {
    build_mark(nullptr);
    // boxes as cubic as the extents allow, `rows` rows each: a box of b0 x b1 x b2 has b0 + b1 + b2 - 2 dependency levels, and
    // every level costs the tile's wave at least one step -- the flatter the box, the emptier its steps.  (Boxes in the
    // proportion of the extents are the same thing on a cube; on the 512 x 512 x 64 slab of an 8-way row split they were
    // 16 x 16 x 2 = 34 levels for 512 rows against 22 for 8 x 8 x 8: GMRES(30)+ILU(0) there 3.52 -> 3.30 ms per iteration;
    // larger cubes lose again -- 10^3: 3.64, 12^3: 4.43, 16^3: 4.85 ms, on the full cube 62.7 / 54.7 / 49.5 / 42.6 it/s for
    // 8^3 / 10^3 / 12^3 / 16^3; tools/r04_runs/zn.sh.)  An extent shorter than the edge keeps its length and the others share
    // the rest.
    {
        double vol  = (double)rows * ((double)E[0] * (double)E[1] * (double)E[2] / (double)n); // (coordinate cells per box)
        bool   done[3] = {E[0] <= 1, E[1] <= 1, E[2] <= 1};
        for(int k = 0; k < 3; ++k)
            bs[k] = 1;
        for(int round = 0; round < 3; ++round)
        {
            int free_dims = 0;
            for(int k = 0; k < 3; ++k)
                free_dims += done[k] ? 0 : 1;
            if(free_dims == 0)
                break;
            const double edge  = pow(vol < 1.0 ? 1.0 : vol, 1.0 / free_dims);
            bool         again = false;
            for(int k = 0; k < 3; ++k)
                if(!done[k] && (double)E[k] <= edge) // the whole extent: the other edges grow
                {
                    bs[k]   = (int)E[k];
                    vol    /= (double)E[k];
                    done[k] = true;
                    again   = true;
                }
            if(!again)
            {
                for(int k = 0; k < 3; ++k)
                    if(!done[k])
                        bs[k] = edge < 1.0 ? 1 : (int)(edge + 0.5);
                break;
            }
        }
    }
    if(const char* e = getenv("RAMD_TRSV_CT_BOX")) // "b0,b1,b2": the box edges as given (tools/ experiments; first attempt only)
    {
        int v[3] = {0, 0, 0};
        if(attempt == 0 && sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]) == 3 && v[0] > 0 && v[1] > 0 && v[2] > 0)
            for(int k = 0; k < 3; ++k)
                bs[k] = v[k];
    }
    static const int grp_shape = getenv("RAMD_TRSV_CT_GSHAPE") ? atoi(getenv("RAMD_TRSV_CT_GSHAPE")) : 1; // (0: proportional boxes)
    if(grp && grp_shape != 0 && grp_count > 0)
    {
        // grouped form: the rows of one dependency level inside a tile are its cross-section over the non-chain coordinates;
        // a cross-section that fits ONE step (64 / lanes-per-row rows = rpp / largest group whole groups) lets a tile advance
        // one level per step -- the critical path of the solve -- instead of one level per several steps.  The box is as long
        // along the chain coordinate as the row budget allows (hops along it trail the predecessor by one step).
        // (the typical group: the mean, rounded -- a few longer runs, e.g. the first mesh nodes whose rows have nothing below
        //  them, only split a step now and then)
        const double gavg  = (double)n / (double)grp_count;
        int          gtyp  = (int)(gavg + 0.5);
        gtyp               = gtyp < 1 ? 1 : (gtyp > grp_maxsize ? grp_maxsize : gtyp);
        static const int cross_env = getenv("RAMD_TRSV_CT_GCROSS") ? atoi(getenv("RAMD_TRSV_CT_GCROSS")) : 0; // (experiments)
        const int    cross = cross_env > 0 ? cross_env : (rpp / gtyp < 1 ? 1 : rpp / gtyp);
        bs[1]              = E[1] > 1 ? (int)(E[1] < cross ? E[1] : cross) : 1;
        bs[2]              = E[2] > 1 ? (cross / bs[1] < 1 ? 1 : cross / bs[1]) : 1;
        double b0          = (double)rows / (gavg * bs[1] * bs[2]);
        bs[0]              = E[0] > 1 ? (b0 < 1.0 ? 1 : (int)(b0 + 0.5)) : 1;
    }
    for(int k = 0; k < 3; ++k)
        Ts[k] = (int)((E[k] + bs[k] - 1) / bs[k]);
    keymax = (int64_t)(Ts[0] + Ts[1] + Ts[2]) * Ts[2] * Ts[1] * Ts[0];
    if(keymax >= (1ll << 30))
        CT_GIVE_UP(4);
    // keys, sort by (tile, level, sweep index)
    CT_TRY(dev_alloc(&lev_t, n));
    CT_TRY(dev_alloc(&tkey, n));
    if(lower)
        hipLaunchKernelGGL((k_ct_keys<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, level, word, bs[0], bs[1], bs[2], Ts[0],
                           Ts[1], Ts[2], lev_t, tkey, gl);
    else
        hipLaunchKernelGGL((k_ct_keys<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, level, word, bs[0], bs[1], bs[2], Ts[0],
                           Ts[1], Ts[2], lev_t, tkey, gl);
    CT_TRY(dev_alloc(&o1, n));
    static const int cls_env = getenv("RAMD_TRSV_CLASSES") ? atoi(getenv("RAMD_TRSV_CLASSES")) : 1; // (0: sweep order inside a level; A/B)
    if(cls_env != 0 && !grp && (int64_t)nlev * 4 + 4 < (1ll << 31))
    {
        int* cls = nullptr;
        CT_TRY(dev_alloc(&cls, n));
        hipLaunchKernelGGL(k_fill_int, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, (int64_t)n, 3, cls);
        if(lower)
            hipLaunchKernelGGL((k_ct_export_class<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, word, tkey, bs[0],
                               bs[1], bs[2], cls);
        else
            hipLaunchKernelGGL((k_ct_export_class<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, word, tkey, bs[0],
                               bs[1], bs[2], cls);
        hipLaunchKernelGGL(k_ct_class_key, dim3(grid), dim3(kBlock), 0, b.cur, n, lev_t, cls);
        s = device_stable_sort_by_key(cls, n, nlev * 4 + 4, o1);
        dev_free(&cls);
        CT_TRY(s);
    }
    else
        CT_TRY(device_stable_sort_by_key(lev_t, n, nlev, o1));
    CT_TRY(dev_alloc(&k2, n));
    hipLaunchKernelGGL(k_ct_gather_int, dim3(grid), dim3(kBlock), 0, b.cur, (int64_t)n, tkey, o1, k2);
    dev_free(&tkey);
    CT_TRY(dev_alloc(&o2, n));
    CT_TRY(device_stable_sort_by_key(k2, n, (int)keymax, o2));
    build_mark("plan: keys and two sorts");
    P->release();
    P->n       = n;
    P->nslices = (n + 63) / 64;
    P->nlevels = nlev;
    CT_TRY(dev_alloc(&P->order, n));
    CT_TRY(dev_alloc(&tflag, (int64_t)n + 1));
    CT_TRY(dev_alloc(&sflag, (int64_t)n + 1));
    hipLaunchKernelGGL(k_ct_flags, dim3(grid), dim3(kBlock), 0, b.cur, n, lower ? 1 : 0, o1, o2, k2, lev_t, P->order, tflag,
                       sflag);
    dev_free(&o1);
    dev_free(&o2);
    dev_free(&k2);
    dev_free(&lev_t);
    CT_TRY(dev_alloc(&P->pos, n));
    hipLaunchKernelGGL(k_invert_perm, dim3(grid), dim3(kBlock), 0, b.cur, n, P->order, P->pos);
    CT_TRY(dev_alloc(&tscan, (int64_t)n + 1));
    CT_TRY(dev_alloc(&sscan, (int64_t)n + 1));
    CT_TRY(device_exclusive_scan(tflag, tscan, (int64_t)n + 1));
    CT_TRY(device_exclusive_scan(sflag, sscan, (int64_t)n + 1));
    {
        // a step holds at most rpp rows: cut the (tile, level) groups (sflag so far) into pieces of rpp rows
        int ngroups = 0;
        CT_HIP(hipMemcpyAsync(&ngroups, sscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
        CT_HIP(hipStreamSynchronize(b.cur));
        int* gpos = nullptr;
        CT_TRY(dev_alloc(&gpos, (int64_t)ngroups + 1));
        hipLaunchKernelGGL(k_ct_scatter_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, sflag, sscan, gpos);
        int* sflag2 = nullptr;
        s           = dev_alloc(&sflag2, (int64_t)n + 1);
        if(s == RAMD_OK && grp)
        {
            // whole groups per step: one thread per tile packs them
            int nt = 0;
            if(hipMemcpyAsync(&nt, tscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
               || hipStreamSynchronize(b.cur) != hipSuccess)
                s = RAMD_ERR_HIP;
            dev_free(&tposv);
            if(s == RAMD_OK)
                s = dev_alloc(&tposv, (int64_t)nt + 1);
            if(s == RAMD_OK)
            {
                hipLaunchKernelGGL(k_ct_scatter_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, tflag, tscan, tposv);
                if(hipMemcpyAsync(tposv + nt, &n, sizeof(int), hipMemcpyHostToDevice, b.cur) != hipSuccess)
                    s = RAMD_ERR_HIP;
                hipLaunchKernelGGL(k_ct_split_groups, dim3(ew_grid(nt)), dim3(kBlock), 0, b.cur, nt, n, lower ? 1 : 0, rpp, tposv,
                                   sflag, P->order, gf, gl, sflag2);
                if(s == RAMD_OK)
                    s = device_exclusive_scan(sflag2, sscan, (int64_t)n + 1);
            }
        }
        else if(s == RAMD_OK)
        {
            hipLaunchKernelGGL(k_ct_split, dim3(grid), dim3(kBlock), 0, b.cur, n, rpp, sflag, sscan, gpos, sflag2);
            s = device_exclusive_scan(sflag2, sscan, (int64_t)n + 1);
        }
        dev_free(&gpos);
        dev_free(&sflag);
        sflag = sflag2;
        CT_TRY(s);
    }
    int cnts[2] = {0, 0};
    CT_HIP(hipMemcpyAsync(&cnts[0], tscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    CT_HIP(hipMemcpyAsync(&cnts[1], sscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    CT_HIP(hipStreamSynchronize(b.cur));
    ntiles = cnts[0];
    nsteps = cnts[1];
    build_mark("plan: flags, scans, step split");
    P->ct_ntiles     = ntiles;
    P->ct_nsteps     = nsteps;
    CT_TRY(dev_alloc(&P->ct_tile_step, (int64_t)ntiles + 1));
    CT_TRY(dev_alloc(&P->ct_step_pos, (int64_t)nsteps + 1));
    CT_TRY(dev_alloc(&P->ct_step_ent, (int64_t)nsteps + 1));
    CT_TRY(dev_alloc(&step_w, (int64_t)nsteps + 1));
    CT_TRY(dev_alloc(&tile_of, n));
    CT_TRY(dev_alloc(&step_of, n));
    CT_HIP(hipMemsetAsync(step_w, 0, sizeof(int) * ((size_t)nsteps + 1), b.cur));
    CT_HIP(hipMemcpyAsync(P->ct_tile_step + ntiles, &nsteps, sizeof(int), hipMemcpyHostToDevice, b.cur));
    CT_HIP(hipMemcpyAsync(P->ct_step_pos + nsteps, &n, sizeof(int), hipMemcpyHostToDevice, b.cur));
    if(lower)
        hipLaunchKernelGGL((k_ct_bounds<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->order, tflag, sflag,
                           tscan, sscan, tile_of, step_of, P->ct_tile_step, P->ct_step_pos, step_w);
    else
        hipLaunchKernelGGL((k_ct_bounds<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->order, tflag, sflag,
                           tscan, sscan, tile_of, step_of, P->ct_tile_step, P->ct_step_pos, step_w);
    dev_free(&tflag);
    dev_free(&sflag);
    dev_free(&tscan);
    dev_free(&sscan);
    if((int64_t)wmax * n >= (1ll << 31) - 65536) // packed entries are addressed with 32-bit offsets
        CT_GIVE_UP(5);
    P->ct_wmax = wmax;
    hipLaunchKernelGGL(k_ct_ent_sizes, dim3(ew_grid(nsteps + 1)), dim3(kBlock), 0, b.cur, nsteps, P->ct_step_pos, step_w,
                       P->ct_step_ent);
    CT_TRY(device_exclusive_scan(P->ct_step_ent, P->ct_step_ent, (int64_t)nsteps + 1));
    CT_HIP(hipMemcpyAsync(&total, P->ct_step_ent + nsteps, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    CT_HIP(hipStreamSynchronize(b.cur));
    // external dependencies: running count per position (position order = use order) and the positions they refer to
    CT_TRY(dev_alloc(&P->ct_ext_start, (int64_t)n + 1));
    if(lower)
        hipLaunchKernelGGL((k_ct_count_ext<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->order, P->pos,
                           tile_of, P->ct_ext_start);
    else
        hipLaunchKernelGGL((k_ct_count_ext<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->order, P->pos,
                           tile_of, P->ct_ext_start);
    CT_TRY(device_exclusive_scan(P->ct_ext_start, P->ct_ext_start, (int64_t)n + 1));
    next = 0;
    CT_HIP(hipMemcpyAsync(&next, P->ct_ext_start + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    CT_HIP(hipStreamSynchronize(b.cur));
    build_mark("plan: bounds, entry sizes, externals count");
    dev_free(&ref_start);
    dev_free(&slot_of_ref);
    static const int dedup_env = getenv("RAMD_TRSV_CT_DEDUP") ? atoi(getenv("RAMD_TRSV_CT_DEDUP")) : -1; // (0 / 1: force)
    if(next > 0 && (dedup_env >= 0 ? dedup_env != 0 : next > n)) // (more references than rows: values are shared)
    {
        // every external value of a tile once (see k_ct_enum_refs)
        const int nr = next;
        int *ref_pc = nullptr, *ref_tile = nullptr, *o1r = nullptr, *o2r = nullptr, *k2r = nullptr, *head = nullptr,
            *is_owner = nullptr, *hscan = nullptr, *uniq = nullptr, *group_slot = nullptr;
        auto drop = [&]() {
            dev_free(&ref_pc);
            dev_free(&ref_tile);
            dev_free(&o1r);
            dev_free(&o2r);
            dev_free(&k2r);
            dev_free(&head);
            dev_free(&is_owner);
            dev_free(&hscan);
            dev_free(&uniq);
            dev_free(&group_slot);
        };
#define CT_TRY2(expr)    \
    do                   \
    {                    \
        s = (expr);      \
        if(s != RAMD_OK) \
        {                \
            drop();      \
            CT_TRY(s);   \
        }                \
    } while(0)
        ref_start        = P->ct_ext_start; // (the running count per reference stays: the fill kernel numbers references with it)
        P->ct_ext_start  = nullptr;
        const int gridr = ew_grid(nr + 1);
        CT_TRY2(dev_alloc(&ref_pc, nr));
        CT_TRY2(dev_alloc(&ref_tile, nr));
        if(lower)
            hipLaunchKernelGGL((k_ct_enum_refs<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->order, P->pos,
                               tile_of, ref_start, ref_pc, ref_tile);
        else
            hipLaunchKernelGGL((k_ct_enum_refs<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->order, P->pos,
                               tile_of, ref_start, ref_pc, ref_tile);
        CT_TRY2(dev_alloc(&o1r, nr));
        CT_TRY2(device_stable_sort_by_key(ref_pc, nr, n, o1r));
        CT_TRY2(dev_alloc(&k2r, nr));
        hipLaunchKernelGGL(k_ct_gather_int, dim3(gridr), dim3(kBlock), 0, b.cur, (int64_t)nr, ref_tile, o1r, k2r);
        CT_TRY2(dev_alloc(&o2r, nr));
        CT_TRY2(device_stable_sort_by_key(k2r, nr, ntiles, o2r));
        CT_TRY2(dev_alloc(&head, (int64_t)nr + 1));
        CT_TRY2(dev_alloc(&is_owner, (int64_t)nr + 1));
        hipLaunchKernelGGL(k_ct_ref_heads, dim3(gridr), dim3(kBlock), 0, b.cur, nr, o1r, o2r, ref_pc, ref_tile, head, is_owner);
        CT_TRY2(dev_alloc(&hscan, (int64_t)nr + 1));
        CT_TRY2(dev_alloc(&uniq, (int64_t)nr + 1));
        CT_TRY2(device_exclusive_scan(head, hscan, (int64_t)nr + 1));
        CT_TRY2(device_exclusive_scan(is_owner, uniq, (int64_t)nr + 1));
        CT_HIP(hipMemcpyAsync(&next, uniq + nr, sizeof(int), hipMemcpyDeviceToHost, b.cur));
        CT_HIP(hipStreamSynchronize(b.cur));
        CT_TRY2(dev_alloc(&group_slot, next));
        CT_TRY2(dev_alloc(&P->ct_ext_idx, next));
        CT_TRY2(dev_alloc(&slot_of_ref, nr));
        hipLaunchKernelGGL(k_ct_group_slots, dim3(gridr), dim3(kBlock), 0, b.cur, nr, o1r, o2r, head, hscan, uniq, ref_pc,
                           group_slot, P->ct_ext_idx);
        hipLaunchKernelGGL(k_ct_ref_slots, dim3(gridr), dim3(kBlock), 0, b.cur, nr, o1r, o2r, head, hscan, group_slot,
                           slot_of_ref);
        CT_TRY2(dev_alloc(&P->ct_ext_start, (int64_t)n + 1));
        hipLaunchKernelGGL(k_ct_ext_start_unique, dim3(grid), dim3(kBlock), 0, b.cur, n, ref_start, uniq, P->ct_ext_start);
        if(verbose)
            fprintf(stderr, "box-tile plan: %d external references -> %d distinct values\n", nr, next);
        drop();
#undef CT_TRY2
    }
    else
        CT_TRY(dev_alloc(&P->ct_ext_idx, next));
    // the largest tile in every respect sizes the LDS areas
    CT_TRY(dev_alloc(&tsz, (int64_t)4 * ntiles));
    hipLaunchKernelGGL(k_ct_tile_sizes, dim3(ew_grid(ntiles)), dim3(kBlock), 0, b.cur, ntiles, P->ct_tile_step, P->ct_step_pos,
                       P->ct_step_ent, P->ct_ext_start, tsz, tsz + ntiles, tsz + 2 * (size_t)ntiles, tsz + 3 * (size_t)ntiles);
    CT_TRY(device_max_int(tsz, ntiles, &P->ct_dims[0]));
    CT_TRY(device_max_int(tsz + ntiles, ntiles, &P->ct_dims[1]));
    CT_TRY(device_max_int(tsz + 2 * (size_t)ntiles, ntiles, &P->ct_dims[2]));
    CT_TRY(device_max_int(tsz + 3 * (size_t)ntiles, ntiles, &P->ct_dims[3]));
    {
        const CtDims d    = {P->ct_dims[0], P->ct_dims[1], P->ct_dims[2], P->ct_dims[3]};
        const size_t need = ct_rec_lds_bytes<T>(d);
        // the occupied part of the coordinate lattice is denser than its bounding box on skewed meshes: the largest tile
        // may hold several times the rows asked for, and big tiles cost more hops' worth of waiting than they save
        const bool too_many_rows = (int64_t)d.rows * 2 > (int64_t)rows_target * 3 && attempt < 3;
        if(need <= (size_t)lds_budget && 1 + d.rows + d.exts < 65536 && !too_many_rows) // (16-bit column codes)
            fits = true;
        else
        {
            if(verbose)
                fprintf(stderr, "box-tile plan: box=(%d,%d,%d) needs %zu B of LDS per tile (max rows %d): shrinking\n", bs[0],
                        bs[1], bs[2], need, d.rows);
            double shrink = (double)lds_budget / (double)need * 0.85;
            if(too_many_rows && (double)rows_target / (double)d.rows < shrink)
                shrink = (double)rows_target / (double)d.rows;
            const int nrows_next = (int)((double)rows * shrink);
            rows                 = nrows_next < rows - 1 ? nrows_next : rows - 1;
            if(rows < 16)
                CT_GIVE_UP(6);
            dev_free(&P->order);
            dev_free(&P->pos);
            dev_free(&P->ct_tile_step);
            dev_free(&P->ct_step_pos);
            dev_free(&P->ct_step_ent);
            dev_free(&P->ct_ext_start);
            dev_free(&P->ct_ext_idx);
            dev_free(&tile_of);
            dev_free(&step_of);
            dev_free(&step_w);
            dev_free(&tsz);
        }
    }
    } // attempts
    build_mark("plan: externals, dedup, fit check");
    dev_free(&level);
    dev_free(&word);
    if(!fits)
        CT_GIVE_UP(6);
    CT_HIP(cached_malloc(&P->w, (size_t)n * sizeof(T) + kPad));
    static const int wslot_env = getenv("RAMD_TRSV_WSLOT") ? atoi(getenv("RAMD_TRSV_WSLOT")) : 0; // (1: the places below; measured, see DESIGN)
    if(wslot_env != 0 && !grp && lpr == 1 && wl == 3 && next > 0 && ntiles > 1)
    {
        // places in w: exported rows of a tile first, grouped by the tile that reads them (k_ct_min_consumer)
        int *cons = nullptr, *w1 = nullptr, *w2 = nullptr, *wk = nullptr;
        auto drop = [&]() {
            dev_free(&cons);
            dev_free(&w1);
            dev_free(&w2);
            dev_free(&wk);
        };
#define CT_TRYW(expr)    \
    do                   \
    {                    \
        s = (expr);      \
        if(s != RAMD_OK) \
        {                \
            drop();      \
            CT_TRY(s);   \
        }                \
    } while(0)
        CT_TRYW(dev_alloc(&cons, n));
        hipLaunchKernelGGL(k_fill_int, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, (int64_t)n, ntiles, cons);
        if(lower)
            hipLaunchKernelGGL((k_ct_min_consumer<true>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->order,
                               P->pos, tile_of, cons);
        else
            hipLaunchKernelGGL((k_ct_min_consumer<false>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->order,
                               P->pos, tile_of, cons);
        CT_TRYW(dev_alloc(&w1, n));
        CT_TRYW(device_stable_sort_by_key(cons, n, ntiles + 1, w1));
        dev_free(&cons);
        CT_TRYW(dev_alloc(&wk, n));
        hipLaunchKernelGGL(k_ct_gather_int, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, (int64_t)n, tile_of, w1, wk);
        CT_TRYW(dev_alloc(&w2, n));
        CT_TRYW(device_stable_sort_by_key(wk, n, ntiles, w2));
        dev_free(&wk);
        CT_TRYW(dev_alloc(&wmap, n));
        hipLaunchKernelGGL(k_ct_wmap, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, w1, w2, wmap);
        drop();
#undef CT_TRYW
        build_mark("plan: places in w");
    }
    CT_TRY(dev_alloc(&nodiag, 1));
    CT_HIP(hipMemsetAsync(nodiag, 0, sizeof(int), b.cur));
    CT_TRY(dev_alloc(&P->ct_tile_desc, (int64_t)8 * ntiles));
    hipLaunchKernelGGL(k_ct_tile_desc, dim3(ew_grid(ntiles)), dim3(kBlock), 0, b.cur, ntiles, P->ct_tile_step, P->ct_step_pos,
                       P->ct_step_ent, P->ct_ext_start, P->ct_tile_desc);
    CT_TRY(dev_alloc(&P->ct_step_rec, (int64_t)4 * ((int64_t)nsteps + 1)));
    P->ct_rec = true;
    {
        // record form: one array of 16-byte quads (CtRec), lpr lane records per row, zeroed = padded
        const size_t nq    = grp ? CtGRec<T>::NQ
                                 : (lpr == 8 ? CtRec<T, 4>::NQ
                                             : (wl == 3 ? CtRec<T, 3>::NQS : (wl == 4 ? CtRec<T, 4>::NQS : CtRec<T, 8>::NQS)));
        // (the diagonal where the record keeps none: CtRec::DSEP; grouped form: the row records)
        const size_t dbytes = (size_t)n * sizeof(T) * (grp ? (size_t)kGrpMax : (size_t)1) + kPad;
        CT_HIP(cached_malloc(&P->diag, dbytes));
        if(grp)
            CT_HIP(hipMemsetAsync(P->diag, 0, dbytes, b.cur));
        const size_t bytes = nq * 16 * (size_t)lpr * (size_t)n + kPad;
        CT_HIP(cached_malloc(&P->eval, bytes));
        CT_HIP(hipMemsetAsync(P->eval, 0, bytes, b.cur));
        if(grp)
        {
            CT_TRY(dev_alloc(&step_maxg, (int64_t)nsteps + 1));
            CT_HIP(hipMemsetAsync(step_maxg, 0, sizeof(int) * ((size_t)nsteps + 1), b.cur));
            hipLaunchKernelGGL(k_ct_step_maxg, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, lower ? 1 : 0, P->order, step_of, gf,
                               gl, step_maxg);
        }
#define CT_FILL_GREC(LOW)                                                                                                    \
    hipLaunchKernelGGL((k_ct_fill_grec<T, LOW>), dim3(nb), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (const T*)m->val,        \
                       P->order, P->pos, tile_of, step_of, P->ct_tile_step, P->ct_step_pos, P->ct_ext_start, P->ct_ext_idx, \
                       (char*)P->eval, nodiag, reverse ? 1 : 0, P->ct_dims[0], ref_start, slot_of_ref, (T*)P->diag, gf)
#define CT_FILL_REC(LOW, WLL, LP)                                                                                            \
    hipLaunchKernelGGL((k_ct_fill_rec<T, LOW, WLL, LP>), dim3(nb), dim3(kBlock), 0, b.cur, n, m->rp, m->ci,                  \
                       (const T*)m->val, P->order, P->pos, tile_of, step_of, P->ct_tile_step, P->ct_step_pos,               \
                       P->ct_ext_start, P->ct_ext_idx, (char*)P->eval, nodiag, reverse ? 1 : 0, P->ct_dims[0], ref_start, \
                       slot_of_ref, (T*)P->diag, wmap)
#define CT_FILL_REC_W(LOW)           \
    do                               \
    {                                \
        if(lpr == 8)                 \
            CT_FILL_REC(LOW, 4, 8);  \
        else if(wl == 3)             \
            CT_FILL_REC(LOW, 3, 1);  \
        else if(wl == 4)             \
            CT_FILL_REC(LOW, 4, 1);  \
        else                         \
            CT_FILL_REC(LOW, 8, 1);  \
    } while(0)
        if(grp && lower)
            CT_FILL_GREC(true);
        else if(grp)
            CT_FILL_GREC(false);
        else if(lower)
            CT_FILL_REC_W(true);
        else
            CT_FILL_REC_W(false);
#undef CT_FILL_REC_W
#undef CT_FILL_REC
#undef CT_FILL_GREC
        {
            // which rows leave their tile (positions named in the external lists, before those are re-mapped to places)
            int *mark = nullptr, *snexp = nullptr;
            s = dev_alloc(&mark, n);
            if(s == RAMD_OK)
                s = dev_alloc(&snexp, (int64_t)nsteps + 1);
            if(s == RAMD_OK)
            {
                hipError_t e = hipMemsetAsync(mark, 0, sizeof(int) * (size_t)n, b.cur);
                if(next > 0)
                    hipLaunchKernelGGL(k_ct_mark_exported, dim3(ew_grid(next)), dim3(kBlock), 0, b.cur, (int64_t)next, P->ct_ext_idx,
                                       mark);
                hipLaunchKernelGGL(k_ct_step_nexp, dim3(ew_grid(nsteps)), dim3(kBlock), 0, b.cur, nsteps, P->ct_step_pos, mark, snexp);
                hipLaunchKernelGGL(k_ct_step_rec2, dim3(ew_grid(nsteps + 1)), dim3(kBlock), 0, b.cur, n, nsteps, P->ct_step_pos,
                                   P->ct_ext_start, tile_of, P->ct_tile_step, P->ct_step_rec, step_maxg, snexp);
                if(e != hipSuccess)
                    s = RAMD_ERR_HIP;
            }
            dev_free(&mark);
            dev_free(&snexp);
            CT_TRY(s);
        }
        if(wmap)
        {
            // everything that names a value of w by its position now names its place: the external values of the tiles,
            // and -- through P->pos, which the next stage's index list is composed with -- the rows themselves
            if(next > 0)
                hipLaunchKernelGGL(k_ct_map_idx, dim3(ew_grid(next)), dim3(kBlock), 0, b.cur, (int64_t)next, wmap, P->ct_ext_idx);
            hipLaunchKernelGGL(k_ct_map_idx, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, (int64_t)n, wmap, P->pos);
        }
        P->ct_grp     = grp;
        P->ct_infirst = grp && (lower == reverse);
    }
    int nd = 0;
    CT_HIP(hipMemcpyAsync(&nd, nodiag, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    CT_HIP(hipStreamSynchronize(b.cur));
    CT_HIP(hipGetLastError());
    P->nodiag = nd != 0;
    P->ct     = true;
    P->nlevels   = nlev;
    P->st_chains = nchains;
    P->st_ext    = next;
    P->st_box[0] = bs[0], P->st_box[1] = bs[1], P->st_box[2] = bs[2];
    build_mark("plan: records");
    if(verbose)
        fprintf(stderr,
                "box-tile plan (%s): n=%d chains=%d levels=%d extents=(%lld,%lld,%lld) box=(%d,%d,%d) tiles=%d steps=%d "
                "wmax=%d max rows/steps/entries/ext per tile = %d/%d/%d/%d%s\n",
                lower ? "lower" : "upper", n, nchains, nlev, (long long)E[0], (long long)E[1], (long long)E[2], bs[0], bs[1],
                bs[2], ntiles, nsteps, wmax, P->ct_dims[0], P->ct_dims[1], P->ct_dims[2], P->ct_dims[3],
                grp ? " [row groups]" : "");
    cleanup();
#undef CT_TRY
#undef CT_HIP
#undef CT_GIVE_UP
    return RAMD_OK;
}

__global__ __launch_bounds__(kBlock) void k_compose_idx(int n, const int* __restrict__ orderU,
                                                        const int* __restrict__ posL,
                                                        int* __restrict__ out)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        out[t] = posL[orderU[t]];
}

// ---- rows of every tile sorted by the source index of their right-hand side (record form; once per index array)
__global__ __launch_bounds__(kBlock) void k_ct_tile_of_pos(int n, int ntiles, const int* __restrict__ tile_desc,
                                                           int* __restrict__ tile_of)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
    {
        int lo = 0, hi = ntiles - 1; // last tile whose first position is <= p
        while(lo < hi)
        {
            const int mid = (lo + hi + 1) >> 1;
            if(tile_desc[8 * (size_t)mid + 2] <= (int)p)
                lo = mid;
            else
                hi = mid - 1;
        }
        tile_of[p] = lo;
    }
}

__global__ __launch_bounds__(kBlock) void k_ct_pair_lists(int n, const int* __restrict__ o1, const int* __restrict__ o2,
                                                          const int* __restrict__ key, const int* __restrict__ tile_of,
                                                          const int* __restrict__ tile_desc, int* __restrict__ pairs)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gsz)
    {
        const int p      = o1[o2[q]]; // position: q-th of its tile in key order
        pairs[2 * q]     = key[p];
        pairs[2 * q + 1] = p - tile_desc[8 * (size_t)tile_of[p] + 2];
    }
}

// The lists in 4 bytes per row where the indices of every tile span little enough (rows of a box tile of a grid in natural
// order: 7 planes; the rows of one tile of the stage before): range[t] = largest - smallest index of tile t, then
// packed[q] = (index - smallest of the tile) << sbits | row number, the smallest index into the tile descriptor.
__global__ __launch_bounds__(kBlock) void k_ct_pair_range(int ntiles, const int* __restrict__ tile_desc,
                                                          const int* __restrict__ pairs, int* __restrict__ range)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gsz)
    {
        const int p0 = tile_desc[8 * t + 2], nr = tile_desc[8 * t + 3];
        range[t]     = nr > 0 ? pairs[2 * (size_t)(p0 + nr - 1)] - pairs[2 * (size_t)p0] : 0;
    }
}

__global__ __launch_bounds__(kBlock) void k_ct_pack_pairs(int n, int sbits, int which, const int* __restrict__ tile_of,
                                                          int* __restrict__ tile_desc, const int* __restrict__ pairs,
                                                          int* __restrict__ packed)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gsz)
    {
        const int t    = tile_of[q];
        const int p0   = tile_desc[8 * (size_t)t + 2];
        const int base = pairs[2 * (size_t)p0];
        packed[q]      = (int)(((unsigned)(pairs[2 * q] - base) << sbits) | (unsigned)pairs[2 * q + 1]);
        if(q == p0)
            tile_desc[8 * (size_t)t + 4 + which] = base;
    }
}

static int ct_sbits(int rows)
{
    int sb = 1;
    while((1 << sb) < rows)
        ++sb;
    return sb;
}

// pairs[q] = {key[p], row number of p inside its tile}, the positions p of every tile sorted by key[p]
// (which: 0 = the right-hand side list, 1 = the output list; *packed_out: the 4-byte form was possible)
static int ct_build_pair_lists(TriPlan* P, const int* key, int** pairs_out, int which, bool* packed_out)
{
    Backend&  b = backend();
    const int n = P->n;
    dev_free(pairs_out);
    int *o1 = nullptr, *o2 = nullptr, *tile_of = nullptr, *k2 = nullptr;
    int  s  = RAMD_OK;
    auto done = [&](int rc) {
        dev_free(&o1);
        dev_free(&o2);
        dev_free(&tile_of);
        dev_free(&k2);
        return rc;
    };
    const int grid = ew_grid(n);
    if((s = dev_alloc(&o1, n)) != RAMD_OK)
        return done(s);
    if((s = device_stable_sort_by_key(key, n, n, o1)) != RAMD_OK) // (keys are positions / rows: < n)
        return done(s);
    if((s = dev_alloc(&tile_of, n)) != RAMD_OK)
        return done(s);
    hipLaunchKernelGGL(k_ct_tile_of_pos, dim3(grid), dim3(kBlock), 0, b.cur, n, P->ct_ntiles, P->ct_tile_desc, tile_of);
    if((s = dev_alloc(&k2, n)) != RAMD_OK)
        return done(s);
    hipLaunchKernelGGL(k_ct_gather_int, dim3(grid), dim3(kBlock), 0, b.cur, (int64_t)n, tile_of, o1, k2);
    if((s = dev_alloc(&o2, n)) != RAMD_OK)
        return done(s);
    if((s = device_stable_sort_by_key(k2, n, P->ct_ntiles, o2)) != RAMD_OK)
        return done(s);
    if((s = dev_alloc(pairs_out, (int64_t)2 * n)) != RAMD_OK)
        return done(s);
    hipLaunchKernelGGL(k_ct_pair_lists, dim3(grid), dim3(kBlock), 0, b.cur, n, o1, o2, key, tile_of, P->ct_tile_desc,
                       *pairs_out);
    *packed_out = false;
    static const int pack_env = getenv("RAMD_TRSV_PACK") ? atoi(getenv("RAMD_TRSV_PACK")) : 1; // (0: 8-byte pairs; A/B)
    const int sbits = ct_sbits(P->ct_dims[0]);
    if(pack_env != 0 && sbits < 24)
    {
        int maxrange = 0;
        dev_free(&k2);
        if((s = dev_alloc(&k2, P->ct_ntiles)) != RAMD_OK)
            return done(s);
        hipLaunchKernelGGL(k_ct_pair_range, dim3(ew_grid(P->ct_ntiles)), dim3(kBlock), 0, b.cur, P->ct_ntiles, P->ct_tile_desc,
                           *pairs_out, k2);
        if((s = device_max_int(k2, P->ct_ntiles, &maxrange)) != RAMD_OK)
            return done(s);
        if(maxrange >= 0 && (int64_t)maxrange < (1ll << (32 - sbits)))
        {
            int* packed = nullptr;
            if((s = dev_alloc(&packed, n)) != RAMD_OK)
                return done(s);
            hipLaunchKernelGGL(k_ct_pack_pairs, dim3(grid), dim3(kBlock), 0, b.cur, n, sbits, which, tile_of, P->ct_tile_desc,
                               *pairs_out, packed);
            if(hipStreamSynchronize(b.cur) != hipSuccess)
            {
                dev_free(&packed);
                done(RAMD_OK);
                RAMD_FAIL(RAMD_ERR_HIP, "packed index lists of the box-tile plan");
            }
            dev_free(pairs_out);
            *pairs_out  = packed;
            *packed_out = true;
        }
    }
    if(hipStreamSynchronize(b.cur) != hipSuccess || hipGetLastError() != hipSuccess)
    {
        done(RAMD_OK);
        RAMD_FAIL(RAMD_ERR_HIP, "sorted index lists of the box-tile plan");
    }
    return done(RAMD_OK);
}

// (the sync-free grouped triangular solve -- k_trsv_sf, its plan fill and its launch -- lives in trsv_syncfree.hip; the analysis
//  that decides for it and lays out its positions and units is build_sf_plan below)
// order[p] = row of position p, plev[p] = its group level, pinfo[p] = row number inside the group | rows of the group << 4 |
// out-of-group entries << 8   (sorted[p] = sweep index of position p)
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_sf_positions(int n, const int* __restrict__ sorted, const int* __restrict__ key,
                                                         const int* __restrict__ gf, const int* __restrict__ gl,
                                                         const int* __restrict__ rp, const int* __restrict__ ci,
                                                         int* __restrict__ order, int* __restrict__ plev, int* __restrict__ pinfo)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
    {
        const int t = sorted[p];
        const int i = LOWER ? t : n - 1 - t;
        int       c = 0;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int col = ci[j];
            if(LOWER ? (col < i) : (col > i))
                if((LOWER ? col : n - 1 - col) < gf[t])
                    ++c;
        }
        order[p] = i;
        plev[p]  = key[t];
        pinfo[p] = (t - gf[t]) | ((gl[t] - gf[t] + 1) << 4) | (c << 8);
    }
}

template <typename T>
static int build_sf_plan(ramd_mat_s* m, TriState* st, TriPlan* P, bool lower, bool reverse)
{
    // reverse (the second stage of LLSolve: the upper part of L^T with a row's entries taken in DESCENDING column order,
    // host_matrix_csr.cpp:1294-1341): the entries outside the group come first then and the in-group ones last, nearest group row
    // last -- the arithmetic of the lower solve on the plan of an upper one
    if(reverse && lower)
        return RAMD_ERR_UNSUPPORTED;
    Backend&  b = backend();
    const int n = m->nrow;
    // RAMD_TRSV_SF = 0: off, 1 (default): deep and narrow graphs, 2: whatever the shape (tests)
    static const int  sf_env  = getenv("RAMD_TRSV_SF") ? atoi(getenv("RAMD_TRSV_SF")) : 1;
    static const bool verbose = getenv("RAMD_TRSV_CT_VERBOSE") != nullptr;
    static const int  sf_kw_min = getenv("RAMD_TRSV_SF_KW") ? std::min(kSfKW, std::max(1, atoi(getenv("RAMD_TRSV_SF_KW")))) : 2;
    if(sf_env == 0 || n < 1 || (sf_env == 1 && n < 4096))
        return RAMD_ERR_UNSUPPORTED;
    int *brk = nullptr, *bscan = nullptr, *rstart = nullptr, *gs = nullptr, *gscan = nullptr, *gstart = nullptr, *gf = nullptr,
        *gl = nullptr, *level = nullptr, *key = nullptr, *sorted = nullptr, *plev = nullptr, *nodiag = nullptr, *cext = nullptr,
        *punit = nullptr;
    UnitPlan units;
    SfPlan*  S = nullptr;
    int      s = RAMD_OK;
    auto cleanup = [&]() {
        dev_free(&brk);
        dev_free(&bscan);
        dev_free(&rstart);
        dev_free(&gs);
        dev_free(&gscan);
        dev_free(&gstart);
        dev_free(&gf);
        dev_free(&gl);
        dev_free(&level);
        dev_free(&key);
        dev_free(&sorted);
        dev_free(&plev);
        dev_free(&nodiag);
        dev_free(&cext);
        punit = nullptr; // (owned by the plan)
        units.release();
    };
#define SF_TRY(expr)       \
    do                     \
    {                      \
        s = (expr);        \
        if(s != RAMD_OK)   \
        {                  \
            cleanup();     \
            sf_release(&S); \
            P->release();  \
            return s;      \
        }                  \
    } while(0)
#define SF_HIP(expr) SF_TRY(((expr) == hipSuccess) ? RAMD_OK : RAMD_ERR_HIP)
#define SF_GIVE_UP(text)                                                                              \
    do                                                                                                \
    {                                                                                                 \
        if(verbose)                                                                                   \
            fprintf(stderr, "sync-free grouped plan (%s): not used (%s)\n", lower ? "lower" : "upper", text); \
        cleanup();                                                                                    \
        sf_release(&S);                                                                               \
        P->release();                                                                                 \
        return RAMD_ERR_UNSUPPORTED;                                                                  \
    } while(0)
    const int      grid = ew_grid(n + 1);
    const unsigned nb1  = (unsigned)(((int64_t)n + 1 + kBlock - 1) / kBlock);
    build_mark(nullptr);
    // row groups: supernode runs cut into pieces of at most kGrpMax rows (every other row is a group of its own)
    SF_TRY(dev_alloc(&brk, (int64_t)n + 1));
    SF_TRY(dev_alloc(&bscan, (int64_t)n + 1));
    if(lower)
        hipLaunchKernelGGL((k_ct_sn_breaks<true>), dim3(nb1), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, brk);
    else
        hipLaunchKernelGGL((k_ct_sn_breaks<false>), dim3(nb1), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, brk);
    SF_TRY(device_exclusive_scan(brk, bscan, (int64_t)n + 1));
    int nruns = 0;
    SF_HIP(hipMemcpyAsync(&nruns, bscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    SF_HIP(hipStreamSynchronize(b.cur));
    if(nruns <= 0)
        SF_GIVE_UP("no rows");
    SF_TRY(dev_alloc(&rstart, (int64_t)nruns + 1));
    hipLaunchKernelGGL(k_ct_scatter_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, brk, bscan, rstart);
    SF_TRY(dev_alloc(&gs, (int64_t)n + 1));
    SF_TRY(dev_alloc(&gscan, (int64_t)n + 1));
    hipLaunchKernelGGL(k_ct_group_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, brk, bscan, rstart, gs);
    SF_TRY(device_exclusive_scan(gs, gscan, (int64_t)n + 1));
    int ngroups = 0;
    SF_HIP(hipMemcpyAsync(&ngroups, gscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    SF_HIP(hipStreamSynchronize(b.cur));
    SF_TRY(dev_alloc(&gstart, (int64_t)ngroups + 1));
    hipLaunchKernelGGL(k_ct_scatter_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, gs, gscan, gstart);
    SF_HIP(hipMemcpyAsync(gstart + ngroups, &n, sizeof(int), hipMemcpyHostToDevice, b.cur));
    SF_TRY(dev_alloc(&gf, n));
    SF_TRY(dev_alloc(&gl, n));
    hipLaunchKernelGGL(k_ct_group_bounds, dim3(grid), dim3(kBlock), 0, b.cur, n, gs, gscan, gstart, gf, gl);
    SF_TRY(dev_alloc(&cext, 4));
    SF_HIP(hipMemsetAsync(cext, 0, sizeof(int) * 4, b.cur));
    if(lower)
        hipLaunchKernelGGL((k_ct_row_wmax_out<true>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, gf, gl, cext + 3,
                           cext + 2);
    else
        hipLaunchKernelGGL((k_ct_row_wmax_out<false>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, gf, gl, cext + 3,
                           cext + 2);
    int two[2] = {0, 0};
    SF_HIP(hipMemcpyAsync(two, cext + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, b.cur));
    SF_HIP(hipStreamSynchronize(b.cur));
    const int maxm = two[0] < 1 ? 1 : two[0], wout = two[1];
    dev_free(&brk);
    dev_free(&bscan);
    dev_free(&rstart);
    dev_free(&gs);
    dev_free(&gscan);
    dev_free(&gstart);
    if(wout > 8 * kSfKW)
        SF_GIVE_UP("rows with more than 48 entries outside their group");
    const int lpr = wout <= 4 * kSfKW ? 4 : 8, rpw = 64 / lpr;
    // levels of the quotient graph (sweep order)
    SF_TRY(unit_schedule(m, lower, &units));
    SF_TRY(dev_alloc(&level, n));
    SF_HIP(hipMemsetAsync(level, 0, sizeof(int) * (size_t)n, b.cur));
    {
        const unsigned nbs = sweep_blocks(units.nunits);
        if(lower)
            hipLaunchKernelGGL((k_ct_glevels<true>), dim3(nbs), dim3(sweep_block_size()), 0, b.cur, n, m->rp, m->ci, gf, gl, level,
                               st->counter, st->ticket, unit_view(units), sweep_poll_cap());
        else
            hipLaunchKernelGGL((k_ct_glevels<false>), dim3(nbs), dim3(sweep_block_size()), 0, b.cur, n, m->rp, m->ci, gf, gl, level,
                               st->counter, st->ticket, unit_view(units), sweep_poll_cap());
        st->ticket += nbs;
    }
    int nglev = 0;
    SF_TRY(device_max_int(level, n, &nglev)); // (synchronises)
    units.release();
    build_mark("sync-free plan: groups, group levels");
    // by default: deep and narrow graphs (fewer than 2048 rows per group level on average), and graphs of any shape whose rows are
    // long enough to keep four lanes busy (more than 8 entries outside their group) -- the random numbering of the config-3 class,
    // 18 group levels of 16 700 groups: 0.49 / 0.86 ms per triangle against 1.17 / 3.1 ms of the level-scheduled rows, whose lanes
    // fetch a row's entries in dependent chunks of eight.  Short rows in wide graphs keep the level-scheduled form (64 rows per wave).
    if(sf_env == 1 && (int64_t)n >= (int64_t)2048 * nglev && wout <= 8)
        SF_GIVE_UP("a wide dependency graph of short rows: level-scheduled rows");
    // positions: sweep rows sorted by the level of their group (stable: groups stay together, rows in sweep order)
    SF_TRY(dev_alloc(&key, n));
    hipLaunchKernelGGL(k_ct_gather_int, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, (int64_t)n, level, gl, key);
    SF_TRY(dev_alloc(&sorted, n));
    SF_TRY(device_stable_sort_by_key(key, n, nglev, sorted));
    S = new SfPlan;
    S->lpr = lpr, S->maxm = maxm, S->wout = wout, S->ngroups = ngroups, S->nglev = nglev, S->infirst = !lower && !reverse;
    SF_TRY(dev_alloc(&P->order, n));
    SF_TRY(dev_alloc(&P->pos, n));
    SF_TRY(dev_alloc(&plev, n));
    SF_TRY(dev_alloc(&S->pinfo, n));
    if(lower)
        hipLaunchKernelGGL((k_sf_positions<true>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, sorted, key, gf, gl, m->rp, m->ci,
                           P->order, plev, S->pinfo);
    else
        hipLaunchKernelGGL((k_sf_positions<false>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, sorted, key, gf, gl, m->rp, m->ci,
                           P->order, plev, S->pinfo);
    hipLaunchKernelGGL(k_invert_perm, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, P->order, P->pos);
    // units: whole groups of one level, at most rpw rows -- a short pass over two int arrays on the host
    std::vector<int> h_lev((size_t)n), h_info((size_t)n), h_u;
    SF_HIP(hipMemcpyAsync(h_lev.data(), plev, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, b.cur));
    SF_HIP(hipMemcpyAsync(h_info.data(), S->pinfo, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, b.cur));
    SF_HIP(hipStreamSynchronize(b.cur));
    h_u.reserve((size_t)ngroups * 4 / (rpw / maxm > 0 ? rpw / maxm : 1) + 16);
    int64_t nentries = 0;
    for(int p = 0; p < n;)
    {
        const int lev = h_lev[p], first = p;
        int       rows = 0, wmax = 0, ngr = 0, gmax = 1;
        while(p < n && h_lev[p] == lev)
        {
            const int gm = (h_info[p] >> 4) & 15;
            if((h_info[p] & 15) != 0 || gm < 1 || p + gm > n)
            {
                cleanup();
                sf_release(&S);
                P->release();
                RAMD_FAIL(RAMD_ERR_STATE, "sync-free plan: a position that does not start its group");
            }
            if(gm > rpw)
            {
                cleanup();
                sf_release(&S);
                P->release();
                RAMD_FAIL(RAMD_ERR_STATE, "sync-free plan: a group longer than a unit");
            }
            if(rows + gm > rpw)
                break;
            for(int q = 0; q < gm; ++q)
                wmax = std::max(wmax, (h_info[p + q] >> 8) & 255);
            rows += gm;
            p += gm;
            ++ngr;
            gmax = std::max(gmax, gm);
        }
        // entries per lane: at least sf_kw_min where the rows are that long, more where the longest row asks for it.  (Few per lane:
        // few planes to fetch and few requests per polling turn, but more lanes in the chain of subtractions -- measured on the
        // RCM shell, lower / upper ms per triangle: 2 per lane 3.46 / 5.12, 3: 3.71 / 5.19, 4: 4.05 / 5.34, 6: 4.4 / 5.7)
        const int kw = std::max(std::min(std::max(wmax, 1), sf_kw_min), (wmax + lpr - 1) / lpr);
        const int nlanes = std::max(1, (wmax + kw - 1) / kw); // lanes per row in use
        if(nentries + sf_unit_slots(rows, kw, nlanes) > (int64_t)0x7fffffff)
        {
            cleanup();
            sf_release(&S);
            P->release();
            return RAMD_ERR_UNSUPPORTED;
        }
        h_u.push_back(first);
        h_u.push_back(rows | (kw << 8) | (nlanes << 12) | ((ngr == 1 ? 1 : 0) << 16) | (gmax << 17));
        h_u.push_back((int)nentries);
        h_u.push_back(-1);
        nentries += sf_unit_slots(rows, kw, nlanes); // (only the lanes that hold entries are stored: round 5 kept 64 per plane)
    }
    S->nunits   = (int)(h_u.size() / 4);
    S->nentries = nentries;
    SF_TRY(dev_alloc(&S->uinfo, (int64_t)h_u.size()));
    SF_HIP(hipMemcpyAsync(S->uinfo, h_u.data(), sizeof(int) * h_u.size(), hipMemcpyHostToDevice, b.cur));
    SF_TRY(dev_alloc(&S->ecol, nentries + 64));
    SF_HIP(cached_malloc(&S->eval, (size_t)(nentries + 64) * sizeof(T) + kPad));
    if(sizeof(T) == 8)
        SF_HIP(cached_malloc(&S->rdiag, (size_t)n * sizeof(T) + kPad));
    if(maxm > 1)
    {
        SF_HIP(cached_malloc(&S->gcoef, (size_t)n * 8 * sizeof(T) + kPad));
        SF_HIP(hipMemsetAsync(S->gcoef, 0, (size_t)n * 8 * sizeof(T), b.cur));
    }
    SF_HIP(cached_malloc(&P->diag, (size_t)n * sizeof(T) + kPad));
    SF_HIP(cached_malloc(&P->w, (size_t)n * sizeof(T) + kPad));
    SF_TRY(dev_alloc(&S->punit, n));
    punit = S->punit;
    SF_TRY(dev_alloc(&S->ufar, S->nunits));
    bool nd = false;
    SF_TRY(sf_fill<T>(S, n, lower, reverse, P->order, P->pos, m->rp, m->ci, (const T*)m->val, (T*)P->diag, &nd));
    P->nodiag  = nd;
    P->nlevels = nglev;
    P->sf      = S;
    S          = nullptr;
    build_mark("sync-free plan: positions, units, fill");
    if(verbose)
        fprintf(stderr,
                "sync-free grouped plan (%s): n=%d, %d row groups of <= %d rows, %d group levels (%.0f rows per level), longest "
                "out-of-group part %d -> %d lanes per row, %d units, %.1f MB of planes\n",
                lower ? "lower" : "upper", n, ngroups, maxm, nglev, (double)n / nglev, wout, lpr, P->sf->nunits,
                (double)nentries * (4 + sizeof(T)) / 1e6);
    cleanup();
    return RAMD_OK;
#undef SF_GIVE_UP
#undef SF_HIP
#undef SF_TRY
}

template <typename T>
static int run_plan(TriState* st, TriPlan* P, bool unit, const T* rhs_src, const int* rhs_idx, T* out,
                    bool mul_inv_diag = false, TriPlan* next_stage = nullptr, bool own_fill_done = false,
                    TriPlan* prev_stage = nullptr)
{
    // prev_stage: the plan whose w is this run's right-hand side (rhs_src); a box-tile kernel leaves sentinels behind every
    // element it has read, so that plan's next run needs no fill (TriPlan::w_sentinel).
    // next_stage: the plan that runs right after this one on the same stream; a box-tile kernel fills its w with sentinels
    // on the way (own_fill_done tells that plan so).  Returns with P->prefilled_next set if it did.
    Backend& b = backend();
    if(P->n == 0)
        return RAMD_OK;
    if(P->sf)
    {
        P->w_sentinel = P->prefilled_next = false;
        return sf_run<T>(P->sf, P->n, mul_inv_diag ? 2 : (unit ? 0 : 1), (const T*)P->diag, (T*)P->w, P->order, rhs_src, rhs_idx, out);
    }
    const unsigned nb = nblocks_of(P->n);
    static const bool nofill = getenv("RAMD_TRSV_NOFILL") != nullptr; // diagnostic only (tools/): no dependency waits
    if((!nofill || !P->filled_once) && !own_fill_done && !P->w_sentinel)
        hipLaunchKernelGGL((k_fill_sentinel<T>), dim3(ew_grid(P->n)), dim3(kBlock), 0, b.cur, (int64_t)P->n,
                           (T*)P->w);
    P->filled_once     = true;
    P->w_sentinel      = false; // (from here on w holds this run's values)
    static const int refill_env = getenv("RAMD_TRSV_REFILL") ? atoi(getenv("RAMD_TRSV_REFILL")) : 1; // (0: off; A/B)
    T* refill = nullptr;
    if(refill_env != 0 && !nofill && prev_stage && prev_stage->n == P->n && prev_stage->w == (void*)rhs_src && P->ct && P->ct_rec)
        refill = (T*)prev_stage->w;
    P->prefilled_next  = false;
    static const int prefill_env = getenv("RAMD_TRSV_PREFILL") ? atoi(getenv("RAMD_TRSV_PREFILL")) : 1; // (0: every plan fills its own w)
    T* prefill = nullptr;
    if(prefill_env != 0 && !nofill && next_stage && next_stage->n == P->n && next_stage->w && P->ct && P->ct_rec)
    {
        prefill           = (T*)next_stage->w;
        P->prefilled_next = true;
    }
    if(P->ct)
    {
        const int    dm    = mul_inv_diag ? 2 : (unit ? 0 : 1);
        const int    lpr   = P->ct_grp ? kGrpLPR : (P->ct_wmax > 8 ? 8 : 1);
        const int    wl    = P->ct_grp ? kGrpWL : (lpr == 1 ? (P->ct_wmax <= 3 ? 3 : (P->ct_wmax <= 4 ? 4 : 8)) : 4);
        CtDims dims = {P->ct_dims[0], P->ct_dims[1], P->ct_dims[2], P->ct_dims[3], P->ct_infirst ? 1 : 0, 0, 0, ct_sbits(P->ct_dims[0])};
        if(P->ct_rec)
        {
            if(P->ct_in_key != rhs_idx)
            {
                P->ct_in_key = nullptr;
                RAMD_TRY(ct_build_pair_lists(P, rhs_idx, &P->ct_in_pairs, 0, &P->ct_in_packed));
                P->ct_in_key = rhs_idx;
            }
            if(out && !P->ct_out_pairs)
                RAMD_TRY(ct_build_pair_lists(P, P->order, &P->ct_out_pairs, 1, &P->ct_out_packed));
            dims.in_packed  = P->ct_in_packed ? 1 : 0;
            dims.out_packed = (out && P->ct_out_packed) ? 1 : 0;
            static const int maskpub_env = getenv("RAMD_TRSV_MASKPUB") ? atoi(getenv("RAMD_TRSV_MASKPUB")) : 1; // (0: every row, A/B)
            dims.mask_pub   = (out && maskpub_env != 0) ? 1 : 0;
            const size_t lds = ct_rec_lds_bytes<T>(dims);
            unsigned     nwg = 0;
            int          nstreams = 1;
            CtBases      bases;
            if(!st->stream_counter)
            {
                RAMD_TRY(dev_alloc(&st->stream_counter, (int64_t)TriState::kStreams * TriState::kStreamStride));
                RAMD_HIP(hipMemsetAsync(st->stream_counter, 0, sizeof(unsigned) * TriState::kStreams * TriState::kStreamStride,
                                        b.cur));
            }
            // RAMD_TRSV_PROF=1 (tools/ diagnostics): cycle counts per phase of the two waves, printed after every solve
            static const bool      pf_on  = getenv("RAMD_TRSV_PROF") != nullptr;
            static unsigned long long* pf_buf = nullptr;
            if(pf_on)
            {
                if(!pf_buf)
                    RAMD_HIP(hipMalloc(&pf_buf, 32 * sizeof(unsigned long long)));
                RAMD_HIP(hipMemsetAsync(pf_buf, 0, 32 * sizeof(unsigned long long), b.cur));
            }
// persistent workgroups: as many as the device holds at once (more would only queue), never more than tiles.  (Fewer -- less
// polling ahead of the tile wavefront -- is no faster: RAMD_TRSV_WGS_PER_CU = 2 / 3 / 4 / 6 / all (9) on the 512 x 512 x 64 slab
// 3.69 / 3.44 / 3.32 / 3.30 / 3.32 ms per GMRES iteration, on the cube 3 / 4 / 6 / all: 50.2 / 55.5 / 60.9 / 61.9 it/s.)
#define TRSV_RC(DM, HO, LP, WLL, DP)                                                                                        \
    do                                                                                                                      \
    {                                                                                                                       \
        static int occ = 0;                                                                                                 \
        static size_t occ_lds = (size_t)-1;                                                                                 \
        if(occ_lds != lds)                                                                                                  \
        {                                                                                                                   \
            int nb_cu = 0;                                                                                                  \
            RAMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_cu, k_trsv_rec<T, DM, HO, LP, WLL, DP, false>, 128, lds));       \
            occ     = nb_cu < 1 ? 1 : nb_cu;                                                                                \
            static const int occ_env = getenv("RAMD_TRSV_WGS_PER_CU") ? atoi(getenv("RAMD_TRSV_WGS_PER_CU")) : 0; /* (experiments) */ \
            if(occ_env > 0 && occ_env < occ)                                                                                \
                occ = occ_env;                                                                                              \
            occ_lds = lds;                                                                                                  \
        }                                                                                                                   \
        const int64_t cap = (int64_t)occ * b.num_cu;                                                                        \
        nwg               = (unsigned)(P->ct_ntiles < cap ? P->ct_ntiles : cap);                                            \
        nstreams          = nwg < (unsigned)TriState::kStreams ? (int)nwg : TriState::kStreams;                             \
        for(int i = 0; i < TriState::kStreams; ++i)                                                                         \
            bases.v[i] = st->stream_ticket[i];                                                                              \
        if(pf_on && WLL == 3 && sizeof(T) == 8)                                                                             \
            hipLaunchKernelGGL((k_trsv_rec<T, DM, HO, 1, 3, 8, true>), dim3(nwg), dim3(128), lds, b.cur,                        \
                               P->ct_ntiles, dims, (const v4i32*)P->ct_tile_desc, (const v4i32*)P->ct_step_rec,             \
                               P->ct_ext_idx, (const v4i32*)P->eval, (const T*)P->diag, rhs_src, P->ct_in_pairs,            \
                               P->ct_out_pairs, (T*)P->w, out,                                                              \
                               st->stream_counter, bases, nstreams, pf_buf, prefill, refill);                               \
        else                                                                                                                \
            hipLaunchKernelGGL((k_trsv_rec<T, DM, HO, LP, WLL, DP, false>), dim3(nwg), dim3(128), lds, b.cur, P->ct_ntiles, \
                               dims, (const v4i32*)P->ct_tile_desc, (const v4i32*)P->ct_step_rec, P->ct_ext_idx,            \
                               (const v4i32*)P->eval, (const T*)P->diag, rhs_src, P->ct_in_pairs, P->ct_out_pairs,           \
                               (T*)P->w, out,                                                                               \
                               st->stream_counter, bases, nstreams, pf_buf, prefill, refill);                               \
    } while(0)
#define TRSV_RC_L(DM, HO)            \
    do                               \
    {                                \
        if(lpr == kGrpLPR)              \
            TRSV_RC(DM, HO, kGrpLPR, kGrpWL, RAMD_CT_DEPTHG); \
        else if(lpr == 8)               \
            TRSV_RC(DM, HO, 8, 4, RAMD_CT_DEPTH8L);   \
        else if(wl == 3)                \
            TRSV_RC(DM, HO, 1, 3, RAMD_CT_DEPTH3);   \
        else if(wl == 4)                \
            TRSV_RC(DM, HO, 1, 4, 8);   \
        else                            \
            TRSV_RC(DM, HO, 1, 8, 6);   \
    } while(0)
#define TRSV_RC_O(DM)                \
    do                               \
    {                                \
        if(out)                      \
            TRSV_RC_L(DM, true);     \
        else                         \
            TRSV_RC_L(DM, false);    \
    } while(0)
            prof_begin(RAMD_PROF_TRSV, b.cur);
            if(dm == 0)
                TRSV_RC_O(0);
            else if(dm == 1)
                TRSV_RC_O(1);
            else
                TRSV_RC_O(2);
            prof_end(RAMD_PROF_TRSV, b.cur);
#undef TRSV_RC_O
#undef TRSV_RC_L
#undef TRSV_RC
            // per stream: its tiles, plus one ticket beyond the last tile for every workgroup bound to it
            for(int i = 0; i < nstreams; ++i)
            {
                const unsigned tiles_i = (unsigned)P->ct_ntiles > (unsigned)i ? ((unsigned)P->ct_ntiles - i + nstreams - 1) / nstreams : 0u;
                const unsigned wgs_i   = nwg > (unsigned)i ? (nwg - i + nstreams - 1) / nstreams : 0u;
                st->stream_ticket[i] += tiles_i + wgs_i;
            }
            RAMD_HIP(hipGetLastError());
            if(refill)
                prev_stage->w_sentinel = true;
            if(pf_on)
            {
                unsigned long long h[32];
                RAMD_HIP(hipMemcpyAsync(h, pf_buf, sizeof(h), hipMemcpyDeviceToHost, b.cur));
                RAMD_HIP(hipStreamSynchronize(b.cur));
                const double wg = h[5] ? (double)h[5] : 1.0;
                fprintf(stderr,
                        "trsv prof (dm=%d out=%d): wgs=%u tiles=%d | compute wave: %.0f cyc/wg, ext-wait %.1f%%, posted-wait %.1f%%, "
                        "steps %llu dups %llu | fetch wave: %.0f cyc/wg, ring-wait %.1f%%, ticket+desc %.1f%%, idx %.1f%%, poll %.1f%%\n",
                        dm, out ? 1 : 0, nwg, P->ct_ntiles, h[0] / wg, 100.0 * h[1] / (h[0] + 1.0), 100.0 * h[2] / (h[0] + 1.0),
                        h[3], h[4], h[8] / wg, 100.0 * h[9] / (h[8] + 1.0), 100.0 * h[10] / (h[8] + 1.0),
                        100.0 * h[11] / (h[8] + 1.0), 100.0 * h[12] / (h[8] + 1.0));
                fprintf(stderr, "trsv prof: SIMD of the compute waves %llu/%llu/%llu/%llu, of the fetch waves %llu/%llu/%llu/%llu\n", h[16],
                        h[17], h[18], h[19], h[20], h[21], h[22], h[23]);
            }
            return RAMD_OK;
        }
    }
    // tuning knobs (measured defaults; the env overrides are for tools/ experiments only)
    static int lds_pad = -1, sleep_cycles = -1;
    if(lds_pad < 0)
    {
        const char* e1 = getenv("RAMD_TRSV_LDS");
        const char* e2 = getenv("RAMD_TRSV_SLEEP");
        lds_pad        = e1 ? atoi(e1) : 0;
        sleep_cycles   = e2 ? atoi(e2) : 4;
    }
#define TRSV(DM)                                                                                          \
    hipLaunchKernelGGL((k_trsv<T, DM>), dim3(nb), dim3(kBlock), (size_t)lds_pad, b.cur, P->n, P->slice_off, \
                       P->ecol, (const T*)P->eval, (const T*)P->diag, rhs_src, rhs_idx, (T*)P->w, out,     \
                       P->order, st->counter, st->ticket, sleep_cycles)
    prof_begin(RAMD_PROF_TRSV, b.cur);
    if(mul_inv_diag)
        TRSV(2);
    else if(unit)
        TRSV(0);
    else
        TRSV(1);
    prof_end(RAMD_PROF_TRSV, b.cur);
#undef TRSV
    st->ticket += nb;
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

template <typename T>
static int ilu0_t(ramd_mat_s* m)
{
    Backend&  b  = backend();
    TriState* st = nullptr;
    RAMD_TRY(tri_get(m, &st));
    const int n    = m->nrow;
    int*      done = nullptr;
    RAMD_TRY(dev_alloc(&done, n));
    if(!m->diag_pos)
    {
        int s = dev_alloc(&m->diag_pos, n);
        if(s != RAMD_OK)
        {
            dev_free(&done);
            return s;
        }
    }
    RAMD_HIP(hipMemsetAsync(done, 0, sizeof(int) * (size_t)n, b.cur));
    const unsigned nb = nblocks_of(n);
    // rows that fit a lane's registers: the natural-order sweep with in-wave pivots (k_ilu0_rows)
    static const int rows_env = getenv("RAMD_ILU0_ROWS") ? atoi(getenv("RAMD_ILU0_ROWS")) : 1; // (0: off; A/B experiments)
    if(rows_env != 0)
    {
        int  maxlen = 0;
        int* dmax   = done; // (borrowed: zeroed above, zeroed again below)
        hipLaunchKernelGGL(k_max_row_len, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, dmax);
        hipError_t e = hipMemcpyAsync(&maxlen, dmax, sizeof(int), hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipMemsetAsync(dmax, 0, sizeof(int), b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
        {
            dev_free(&done);
            RAMD_HIP(e);
        }
        if(maxlen <= kIluWMax)
        {
            UnitPlan up;
            int      su = unit_schedule(m, true, &up);
            if(su != RAMD_OK)
            {
                dev_free(&done);
                return su;
            }
            // (workgroups of 256: the waves of this sweep run for tens of microseconds, and a workgroup of 16 holds its CU
            //  slots until the last of them is done -- measured 101 ms against 77 ms at 512^3)
            const unsigned nbi = (unsigned)((up.nunits + 3) / 4);
            if(maxlen <= 8)
                hipLaunchKernelGGL((k_ilu0_rows<T, 256, 8>), dim3(nbi), dim3(256), 0, b.cur, n, m->rp, m->ci, (T*)m->val, done,
                                   m->diag_pos, st->counter, st->ticket, unit_view(up), sweep_poll_cap());
            else
                hipLaunchKernelGGL((k_ilu0_rows<T, 256, kIluWMax>), dim3(nbi), dim3(256), 0, b.cur, n, m->rp, m->ci, (T*)m->val,
                                   done, m->diag_pos, st->counter, st->ticket, unit_view(up), sweep_poll_cap());
            st->ticket += nbi;
            e = hipGetLastError();
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            up.release();
            build_mark("ilu0: factorisation sweep (rows in registers)");
            dev_free(&done);
            RAMD_HIP(e);
            return RAMD_OK;
        }
    }
    if(!st->l_order_cache)
    {
        dev_free(&st->l_level_cache);
        int s = level_order(m, st, true, &st->l_order_cache, &st->l_nlev_cache, &st->l_level_cache);
        if(s != RAMD_OK)
        {
            dev_free(&done);
            return s;
        }
    }
    hipLaunchKernelGGL((k_ilu0<T>), dim3(nb), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (T*)m->val, done,
                       m->diag_pos, st->counter, st->ticket, st->l_order_cache);
    st->ticket += nb;
    hipError_t e = hipGetLastError();
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    build_mark("ilu0: factorisation sweep");
    dev_free(&done);
    RAMD_HIP(e);
    return RAMD_OK;
}

// ---------------------------------------------------------------- ILU(p) with fill levels, sync-free
// host_matrix_csr.cpp:3149-3312 (ILUpFactorizeNumeric) on the pattern S = pattern(A^(p+1)) (local_matrix.cpp:3929-3935).
// Entries of S start with A's value and level 0, or 0 and level "infinite".
constexpr int kIlupInf = 99999; // host :3171

template <typename T>
__global__ __launch_bounds__(kBlock) void k_ilup_init(int nrow, const int* __restrict__ srp, const int* __restrict__ sci,
                                                      const int* __restrict__ arp, const int* __restrict__ aci,
                                                      const T* __restrict__ aval, T* __restrict__ val,
                                                      int* __restrict__ lev)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrow; row += gsz)
    {
        int       a  = arp[row];
        const int ae = arp[row + 1];
        for(int j = srp[row]; j < srp[row + 1]; ++j)
        {
            const int c = sci[j];
            while(a < ae && aci[a] < c)
                ++a;
            const bool hit = (a < ae && aci[a] == c);
            val[j]         = hit ? aval[a] : (T)0;
            if(lev)
                lev[j] = hit ? 0 : kIlupInf;
        }
    }
}

// Thread per row, rows in (level, row) order of S's lower part.  Row i walks its lower entries a_ik in ascending k; an
// entry whose level (as updated so far) exceeds p is skipped, otherwise the row waits for row k, scales by its pivot
// and visits EVERY own entry right of it: level = min(level, lev_kj + lev_ik + 1), a_ij -= a_ik * a_kj (entries of row
// k that were dropped are published as 0 / infinite).  Afterwards the entries above level p are zeroed and the kept
// ones counted.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_ilup(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                 T* val, int* lev, int* done, int* diag_pos, int* __restrict__ cnt,
                                                 int p, unsigned* counter, unsigned base,
                                                 const int* __restrict__ order)
{
    using B            = typename Sentinel<T>::bits;
    const unsigned blk = take_ticket(counter, base);
    const int64_t  t   = (int64_t)blk * kBlock + threadIdx.x;
    if(t >= nrow)
        return;
    const int i  = order[t];
    const int rs = rp[i], re = rp[i + 1];
    int       j  = rs;
    int       dj = rs;
    while(dj < re && ci[dj] < i)
        ++dj;
    bool fin = false;
    int  spins = 0, backoff = 1;
    do
    {
        spin_guard(spins);
        const int  j_before   = j;
        const bool fin_before = fin;
        if(!fin)
        {
            if(j < dj)
            {
                const int lj = lev[j];
                if(lj > p)
                    ++j;
                else
                {
                    const int k = ci[j];
                    if(__hip_atomic_load(done + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                    {
                        const int kd  = __hip_atomic_load(diag_pos + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int kre = rp[k + 1];
                        const T   pivot = Sentinel<T>::from_bits(__hip_atomic_load(
                            reinterpret_cast<const B*>(val + kd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        const T f = val[j] / pivot;
                        val[j]    = f;
                        int q     = kd + 1;
                        for(int m = j + 1; m < re; ++m)
                        {
                            const int cm = ci[m];
                            while(q < kre && ci[q] < cm)
                                ++q;
                            if(q >= kre)
                                break;
                            if(ci[q] == cm)
                            {
                                const int lkq = __hip_atomic_load(lev + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const T   akq = Sentinel<T>::from_bits(__hip_atomic_load(
                                    reinterpret_cast<const B*>(val + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                const int l   = lkq + lj + 1;
                                if(lev[m] > l)
                                    lev[m] = l;
                                val[m] -= f * akq;
                            }
                        }
                        ++j;
                    }
                }
            }
            else
            {
                int c = 0;
                for(int q = rs; q < re; ++q)
                {
                    const bool keep = lev[q] <= p;
                    c += keep ? 1 : 0;
                    __hip_atomic_store(reinterpret_cast<B*>(val + q), Sentinel<T>::as_bits(keep ? val[q] : (T)0),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(lev + q, keep ? lev[q] : kIlupInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                cnt[i] = c;
                __hip_atomic_store(diag_pos + i, dj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(done + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fin = true;
            }
        }
        backoff = poll_backoff(__ballot(!fin_before && (fin || j != j_before)) != 0ull, backoff);
    } while(__ballot(!fin) != 0ull);
}

// The same sweep with one WAVE per row (rows of S up to 64*K entries): lane e % 64 keeps entry e of the row (column,
// value, level) in registers, the lower entries are taken in ascending order (their value / level broadcast from the
// owning lane), and every lane right of the pivot entry finds its column in row k by bisection -- per lower entry one
// parallel step instead of a serial merge over the whole row.  Same operations per entry in the same order.
constexpr unsigned kIlupWaveChunk = 1u << 23; // workgroups per launch of the wave-per-row sweeps (2^31 threads)
__global__ __launch_bounds__(kBlock) void k_row_lengths(int nrow, const int* __restrict__ rp, int* __restrict__ len)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrow; row += gsz)
        len[row] = rp[row + 1] - rp[row];
}

template <typename T, int K, bool ILU0 = false> // ILU0: no levels (lev == nullptr), zero pivots skipped (host :2132)
__global__ __launch_bounds__(kBlock) void k_ilup_wave(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      T* val, int* lev, int* done, int* diag_pos,
                                                      int* __restrict__ cnt, int p, unsigned* counter, unsigned base,
                                                      const int* __restrict__ order, int64_t row_base)
{
    using B              = typename Sentinel<T>::bits;
    constexpr int  WPB   = kBlock / 64;
    const unsigned blk   = take_ticket(counter, base);
    const int      lane  = threadIdx.x & 63;
    const int64_t  t     = row_base + (int64_t)blk * WPB + (threadIdx.x >> 6); // launches of <= 2^23 workgroups
    if(t >= nrow)
        return;
    const int i  = order[t];
    const int rs = rp[i], re = rp[i + 1];
    const int len = re - rs;
    int c[K], l[K];
    T   v[K];
    int nlow = 0;
#pragma unroll
    for(int k = 0; k < K; ++k)
    {
        const int  e  = k * 64 + lane;
        const bool ok = e < len;
        c[k]          = ok ? ci[rs + e] : 0x7fffffff;
        v[k]          = ok ? val[rs + e] : (T)0;
        l[k]          = ILU0 ? 0 : (ok ? lev[rs + e] : kIlupInf);
        nlow += __popcll(__ballot(ok && c[k] < i));
    }
    for(int a = 0; a < nlow; ++a)
    {
        const int ak = a >> 6, al = a & 63;
        int       la = 0, krow = 0;
        T         va = (T)0;
#pragma unroll
        for(int k = 0; k < K; ++k)
            if(k == ak)
            {
                la   = __shfl(l[k], al);
                krow = __shfl(c[k], al);
                va   = __shfl(v[k], al);
            }
        if(la > p)
            continue;
        int spins = 0, backoff = 1;
        while(__hip_atomic_load(done + krow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        {
            spin_guard(spins);
            backoff = poll_backoff(false, backoff);
        }
        const int kd    = __hip_atomic_load(diag_pos + krow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int kre   = rp[krow + 1];
        const T   pivot = Sentinel<T>::from_bits(
            __hip_atomic_load(reinterpret_cast<const B*>(val + kd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if(ILU0 && pivot == (T)0)
            continue;
        const T f = va / pivot;
#pragma unroll
        for(int k = 0; k < K; ++k)
        {
            const int e = k * 64 + lane;
            if(e == a)
                v[k] = f;
            if(e > a && e < len)
            {
                int lo = kd + 1, hi = kre; // first position with ci >= c[k]
                while(lo < hi)
                {
                    const int mid = (lo + hi) >> 1;
                    if(ci[mid] < c[k])
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                if(lo < kre && ci[lo] == c[k])
                {
                    const T akq = Sentinel<T>::from_bits(__hip_atomic_load(
                        reinterpret_cast<const B*>(val + lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if(!ILU0)
                    {
                        const int ln = __hip_atomic_load(lev + lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + la + 1;
                        if(l[k] > ln)
                            l[k] = ln;
                    }
                    v[k] -= f * akq;
                }
            }
        }
    }
    int kept = 0;
#pragma unroll
    for(int k = 0; k < K; ++k)
    {
        const int  e    = k * 64 + lane;
        const bool keep = e < len && l[k] <= p;
        kept += __popcll(__ballot(keep));
        if(e < len)
        {
            __hip_atomic_store(reinterpret_cast<B*>(val + rs + e), Sentinel<T>::as_bits(keep ? v[k] : (T)0),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(!ILU0)
                __hip_atomic_store(lev + rs + e, keep ? l[k] : kIlupInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if(lane == 0)
    {
        if(!ILU0)
            cnt[i] = kept;
        __hip_atomic_store(diag_pos + i, rs + nlow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(done + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_ilup_compact(int nrow, const int* __restrict__ srp,
                                                         const int* __restrict__ sci, const T* __restrict__ sval,
                                                         const int* __restrict__ lev, int p,
                                                         const int* __restrict__ rp, int* __restrict__ ci,
                                                         T* __restrict__ val)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrow; row += gsz)
    {
        int o = rp[row];
        for(int j = srp[row]; j < srp[row + 1]; ++j)
            if(lev[j] <= p)
            {
                ci[o]  = sci[j];
                val[o] = sval[j];
                ++o;
            }
    }
}

// ILU(0) on a pattern with long rows (the power pattern of ILU(p, level = false)): the wave-per-row sweep without
// levels; rows beyond 256 entries -> the thread-per-row kernel
template <typename T>
static int ilu0_long_rows_t(ramd_mat_s* m)
{
    Backend&       b   = backend();
    const int      n   = m->nrow;
    const unsigned nb  = nblocks_of(n);
    int*           len = nullptr;
    int            maxlen = 0;
    RAMD_TRY(dev_alloc(&len, n));
    hipLaunchKernelGGL(k_row_lengths, dim3(nb), dim3(kBlock), 0, b.cur, n, m->rp, len);
    int s = device_max_int(len, n, &maxlen);
    static const bool wave_rows = !(getenv("RAMD_ILUP_WAVE") && atoi(getenv("RAMD_ILUP_WAVE")) == 0);
    if(s != RAMD_OK || !wave_rows || maxlen > 256)
    {
        dev_free(&len);
        return s != RAMD_OK ? s : ilu0_t<T>(m);
    }
    int*      done = len; // reused as the flags
    TriState* st   = nullptr;
    if((s = tri_get(m, &st)) != RAMD_OK || (!m->diag_pos && (s = dev_alloc(&m->diag_pos, n)) != RAMD_OK)
       || (!st->l_order_cache
           && (dev_free(&st->l_level_cache),
               (s = level_order(m, st, true, &st->l_order_cache, &st->l_nlev_cache, &st->l_level_cache)) != RAMD_OK)))
    {
        dev_free(&done);
        return s;
    }
    hipError_t     e   = hipMemsetAsync(done, 0, sizeof(int) * (size_t)n, b.cur);
    const unsigned nbw = (unsigned)(((int64_t)n + kBlock / 64 - 1) / (kBlock / 64));
#define ILU0_WAVE(K)                                                                                               \
    hipLaunchKernelGGL((k_ilup_wave<T, K, true>), dim3(nbc), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (T*)m->val,  \
                       (int*)nullptr, done, m->diag_pos, (int*)nullptr, 0, st->counter, st->ticket,                \
                       st->l_order_cache, (int64_t)b0 * (kBlock / 64))
    for(unsigned b0 = 0; b0 < nbw && e == hipSuccess; b0 += kIlupWaveChunk) // a launch holds < 2^32 threads
    {
        const unsigned nbc = std::min(kIlupWaveChunk, nbw - b0);
        if(maxlen <= 64)
            ILU0_WAVE(1);
        else if(maxlen <= 128)
            ILU0_WAVE(2);
        else
            ILU0_WAVE(4);
        e = hipGetLastError(); // per chunk: the ticket only advances past launches that went out
        if(e == hipSuccess)
            st->ticket += nbc;
    }
#undef ILU0_WAVE
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&done);
    if(e != hipSuccess)
        tri_resync(st);
    RAMD_HIP(e);
    return RAMD_OK;
}

template <typename T>
static int ilup_t(ramd_mat_s* m, int p, bool level)
{
    Backend&  b = backend();
    const int n = m->nrow;
    // S = pattern(A^(p+1)), sorted rows (local_matrix.cpp:3931-3933)
    ramd_mat_s* S = nullptr;
    RAMD_TRY(mat_symbolic_power(m, p + 1, &S));
    int*           lev  = nullptr;
    int*           done = nullptr;
    int*           cnt  = nullptr;
    int            s    = RAMD_OK;
    const unsigned nb   = nblocks_of(n);
    auto           fail = [&](int code) {
        dev_free(&lev);
        dev_free(&done);
        dev_free(&cnt);
        ramd_mat_destroy(S);
        return code;
    };
    if(level && (s = dev_alloc(&lev, S->nnz)) != RAMD_OK)
        return fail(s);
    hipLaunchKernelGGL((k_ilup_init<T>), dim3(nb), dim3(kBlock), 0, b.cur, n, S->rp, S->ci, m->rp, m->ci,
                       (const T*)m->val, (T*)S->val, lev);
    if(hipGetLastError() != hipSuccess)
        return fail(RAMD_ERR_HIP);
    if(!level)
    {
        // local_matrix.cpp:3985-3993: A's values on the power pattern, then ILU(0)
        std::swap(m->rp, S->rp);
        std::swap(m->ci, S->ci);
        std::swap(m->val, S->val);
        std::swap(m->nnz, S->nnz);
        mat_free_analysis(m);
        m->band_dist = -1;
        m->shift_rows = -1;
        (void)fail(RAMD_OK);
        return ilu0_long_rows_t<T>(m);
    }
    TriState* st = nullptr;
    if((s = tri_get(S, &st)) != RAMD_OK || (s = dev_alloc(&done, n)) != RAMD_OK
       || (s = dev_alloc(&cnt, (int64_t)n + 1)) != RAMD_OK
       || (!S->diag_pos && (s = dev_alloc(&S->diag_pos, n)) != RAMD_OK))
        return fail(s);
    if(hipMemsetAsync(done, 0, sizeof(int) * (size_t)n, b.cur) != hipSuccess
       || hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)n + 1), b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    if(!st->l_order_cache && (s = level_order(S, st, true, &st->l_order_cache, &st->l_nlev_cache)) != RAMD_OK)
        return fail(s);
    // longest row of S decides: wave per row (<= 256 entries) or thread per row
    int maxlen = 0;
    hipLaunchKernelGGL(k_row_lengths, dim3(nb), dim3(kBlock), 0, b.cur, n, S->rp, cnt);
    if((s = device_max_int(cnt, n, &maxlen)) != RAMD_OK)
        return fail(s);
    static const bool wave_rows = !(getenv("RAMD_ILUP_WAVE") && atoi(getenv("RAMD_ILUP_WAVE")) == 0);
    if(wave_rows && maxlen <= 256)
    {
        const unsigned nbw = (unsigned)(((int64_t)n + kBlock / 64 - 1) / (kBlock / 64));
#define ILUP_WAVE(K)                                                                                             \
    hipLaunchKernelGGL((k_ilup_wave<T, K>), dim3(nbc), dim3(kBlock), 0, b.cur, n, S->rp, S->ci, (T*)S->val, lev, \
                       done, S->diag_pos, cnt, p, st->counter, st->ticket, st->l_order_cache,                    \
                       (int64_t)b0 * (kBlock / 64))
        hipError_t le = hipSuccess;
        for(unsigned b0 = 0; b0 < nbw && le == hipSuccess; b0 += kIlupWaveChunk)
        {
            const unsigned nbc = std::min(kIlupWaveChunk, nbw - b0);
            if(maxlen <= 64)
                ILUP_WAVE(1);
            else if(maxlen <= 128)
                ILUP_WAVE(2);
            else
                ILUP_WAVE(4);
            le = hipGetLastError(); // per chunk: the ticket only advances past launches that went out
            if(le == hipSuccess)
                st->ticket += nbc;
        }
#undef ILUP_WAVE
        if(le != hipSuccess)
        {
            tri_resync(st);
            return fail(RAMD_ERR_HIP);
        }
    }
    else
    {
        hipLaunchKernelGGL((k_ilup<T>), dim3(nb), dim3(kBlock), 0, b.cur, n, S->rp, S->ci, (T*)S->val, lev, done,
                           S->diag_pos, cnt, p, st->counter, st->ticket, st->l_order_cache);
        if(hipGetLastError() != hipSuccess)
        {
            tri_resync(st);
            return fail(RAMD_ERR_HIP);
        }
        st->ticket += nb;
    }
    if(hipStreamSynchronize(b.cur) != hipSuccess)
    {
        tri_resync(st);
        return fail(RAMD_ERR_HIP);
    }
    if((s = device_exclusive_scan(cnt, cnt, (int64_t)n + 1)) != RAMD_OK)
        return fail(s);
    int total = 0;
    if(hipMemcpyAsync(&total, cnt + n, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
       || hipStreamSynchronize(b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    if((s = mat_alloc_csr(m, n, n, total)) != RAMD_OK) // A's arrays are no longer needed
        return fail(s);
    if(hipMemcpyAsync(m->rp, cnt, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToDevice, b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    hipLaunchKernelGGL((k_ilup_compact<T>), dim3(nb), dim3(kBlock), 0, b.cur, n, S->rp, S->ci, (const T*)S->val, lev,
                       p, m->rp, m->ci, (T*)m->val);
    if(hipGetLastError() != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    return fail(RAMD_OK);
}

// ---------------------------------------------------------------- IC(0), level order, sync-free
// host_matrix_csr.cpp:2344-2466 on L = lower part incl. diagonal (sorted rows => the diagonal is the last entry of
// every row).  Thread per row, rows in (level, row) order; a row waits for every row col_j < i of its pattern:
//   l_ij = (a_ij - sum_k l_jk l_ik) / l_jj   (k ascending over row j, products added when row i has column k)
//   l_ii = sqrt(|a_ii - sum_j l_ij^2|),  inv_diag_i = 1 / l_ii
// err: 1 structural zero diagonal, 2 numerical zero (the reference aborts with "IC breakdown")
template <typename T>
__global__ __launch_bounds__(kBlock) void k_ic0(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                T* val, T* __restrict__ inv_diag, int* done, int* err,
                                                unsigned* counter, unsigned base, const int* __restrict__ order)
{
    using B            = typename Sentinel<T>::bits;
    const unsigned blk = take_ticket(counter, base);
    const int64_t  t   = (int64_t)blk * kBlock + threadIdx.x;
    if(t >= nrow)
        return;
    const int i  = order[t];
    const int rs = rp[i], re = rp[i + 1];
    int       j  = rs;
    T         sum = (T)0;
    bool      fin = false;
    int       spins = 0;
    int       backoff = 1;
    do
    {
        spin_guard(spins);
        const int  j_before   = j;
        const bool fin_before = fin;
        if(!fin)
        {
            const int cj = (j < re) ? ci[j] : nrow;
            if(j < re && cj < i)
            {
                if(__hip_atomic_load(done + cj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                {
                    const int rbj = rp[cj], rdj = rp[cj + 1] - 1; // diagonal of row cj: its last entry
                    const T   dj  = Sentinel<T>::from_bits(__hip_atomic_load(
                        reinterpret_cast<const B*>(val + rdj), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    T local_sum = (T)0;
                    int m = rs; // own entries left of (i, cj), ascending
                    for(int k = rbj; k < rdj; ++k)
                    {
                        const int ck = ci[k];
                        while(m < j && ci[m] < ck)
                            ++m;
                        if(m < j && ci[m] == ck)
                        {
                            const T ljk = Sentinel<T>::from_bits(__hip_atomic_load(
                                reinterpret_cast<const B*>(val + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            local_sum += ljk * val[m];
                        }
                    }
                    if(dj == (T)0)
                        *err = 2;
                    const T inv = (T)1 / dj;
                    const T lij = (val[j] - local_sum) * inv;
                    sum += lij * lij;
                    val[j] = lij;
                    ++j;
                }
            }
            else
            {
                // diagonal (or its absence)
                T dinv = (T)1;
                if(j < re && cj == i)
                {
                    const T a  = val[j] - sum;
                    const T de = (T)sqrt((double)(a < (T)0 ? -a : a));
                    val[j]     = de;
                    if(de == (T)0)
                        *err = 2;
                    dinv = (T)1 / de;
                }
                else
                    *err = 1;
                inv_diag[i] = dinv;
                for(int q = rs; q < re && q <= j; ++q) // publish the row write-through, then the flag
                    __hip_atomic_store(reinterpret_cast<B*>(val + q), Sentinel<T>::as_bits(val[q]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(done + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fin = true;
            }
        }
        backoff = poll_backoff(__ballot(!fin_before && (fin || j != j_before)) != 0ull, backoff);
    } while(__ballot(!fin) != 0ull);
}

template <typename T>
static int ic0_t(ramd_mat_s* m, ramd_vec_s* inv_diag)
{
    Backend&  b  = backend();
    TriState* st = nullptr;
    RAMD_TRY(tri_get(m, &st));
    const int n    = m->nrow;
    int*      done = nullptr;
    int*      err  = nullptr;
    RAMD_TRY(dev_alloc(&done, (int64_t)n + 1));
    err = done + n;
    hipError_t e = hipMemsetAsync(done, 0, sizeof(int) * ((size_t)n + 1), b.cur);
    if(!st->l_order_cache)
    {
        int s = level_order(m, st, true, &st->l_order_cache, &st->l_nlev_cache);
        if(s != RAMD_OK)
        {
            dev_free(&done);
            return s;
        }
    }
    const unsigned nb = nblocks_of(n);
    hipLaunchKernelGGL((k_ic0<T>), dim3(nb), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, (T*)m->val, (T*)inv_diag->d, done,
                       err, st->counter, st->ticket, st->l_order_cache);
    st->ticket += nb;
    int herr = 0;
    if(e == hipSuccess)
        e = hipGetLastError();
    if(e == hipSuccess)
        e = hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, b.cur);
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&done);
    RAMD_HIP(e);
    if(herr == 1)
        RAMD_FAIL(RAMD_ERR_STATE, "IC breakdown: structural zero diagonal");
    if(herr == 2)
        RAMD_FAIL(RAMD_ERR_STATE, "IC breakdown: division by zero");
    return RAMD_OK;
}

// ---------------------------------------------------------------- transpose (pattern + values) for L^T
// lower_only: entries right of the diagonal are left out (L^T of a matrix that also stores an upper part)
__global__ __launch_bounds__(kBlock) void k_tr_count(int nrow, const int* __restrict__ rp, const int* __restrict__ ci,
                                                     int* __restrict__ cnt, int lower_only)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += gsz)
        for(int j = rp[r]; j < rp[r + 1]; ++j)
            if(!lower_only || ci[j] <= r)
                atomicAdd(cnt + ci[j], 1);
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_tr_scatter(int nrow, const int* __restrict__ rp,
                                                       const int* __restrict__ ci, const T* __restrict__ val,
                                                       int* __restrict__ cursor, int* __restrict__ tci,
                                                       T* __restrict__ tval, int lower_only)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += gsz)
        for(int j = rp[r]; j < rp[r + 1]; ++j)
        {
            if(lower_only && ci[j] > r)
                continue;
            const int p = atomicAdd(cursor + ci[j], 1);
            tci[p]      = (int)r;
            tval[p]     = val[j];
        }
}
// the scatter order inside a row is arbitrary: sort every (short) row by column
template <typename T>
__global__ __launch_bounds__(kBlock) void k_tr_sort_rows(int nrow, const int* __restrict__ rp, int* __restrict__ ci,
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

template <typename T>
static int transpose_into(const ramd_mat_s* m, ramd_mat_s* t, bool lower_only = false)
{
    Backend& b = backend();
    RAMD_TRY(mat_alloc_csr(t, m->ncol, m->nrow, m->nnz)); // lower_only: an upper bound, the tail stays unused
    RAMD_HIP(hipMemsetAsync(t->rp, 0, sizeof(int) * ((size_t)t->nrow + 1), b.cur));
    const int grid = ew_grid(std::max(m->nrow, 1));
    hipLaunchKernelGGL(k_tr_count, dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci, t->rp, lower_only ? 1 : 0);
    RAMD_TRY(device_exclusive_scan(t->rp, t->rp, (int64_t)t->nrow + 1));
    int* cursor = nullptr;
    RAMD_TRY(dev_alloc(&cursor, (int64_t)t->nrow + 1));
    hipError_t e = hipMemcpyAsync(cursor, t->rp, sizeof(int) * ((size_t)t->nrow + 1), hipMemcpyDeviceToDevice, b.cur);
    hipLaunchKernelGGL((k_tr_scatter<T>), dim3(grid), dim3(kBlock), 0, b.cur, m->nrow, m->rp, m->ci, (const T*)m->val,
                       cursor, t->ci, (T*)t->val, lower_only ? 1 : 0);
    hipLaunchKernelGGL((k_tr_sort_rows<T>), dim3(ew_grid(std::max(t->nrow, 1))), dim3(kBlock), 0, b.cur, t->nrow, t->rp,
                       t->ci, (T*)t->val);
    if(e == hipSuccess)
        e = hipGetLastError();
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&cursor);
    RAMD_HIP(e);
    return RAMD_OK;
}

template <typename T>
static int transpose_lower_into(const ramd_mat_s* m, ramd_mat_s* t)
{
    return transpose_into<T>(m, t, true);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_diag(int n, const int* __restrict__ order, const T* __restrict__ src,
                                                        T* __restrict__ dst)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        dst[t] = src[order[t]];
}

// record form: the diagonal lives inside the rows' records (last lane record of the row); one wave per step
template <typename T, int WL, int LPR>
__global__ __launch_bounds__(kBlock) void k_ct_rec_set_diag(int nsteps, const int* __restrict__ step_rec,
                                                            const int* __restrict__ order, const T* __restrict__ src,
                                                            char* __restrict__ erec)
{
    using L            = CtRec<T, WL>;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int     lane = threadIdx.x & 63;
    if(wave >= nsteps)
        return;
    const int p0 = step_rec[4 * wave], cnt = step_rec[4 * wave + 1] & 0xff; // (rows | largest group << 8)
    if(lane < cnt)
        *reinterpret_cast<T*>(erec
                              + ((size_t)L::NQ * LPR * p0 + (size_t)(L::off_diag / 16) * (cnt * LPR) + (lane * LPR + LPR - 1)) * 16
                              + (L::off_diag % 16))
            = src[order[p0 + lane]];
}

// grouped form: the diagonal is the last element of the row record
template <typename T>
__global__ __launch_bounds__(kBlock) void k_ct_grec_set_diag(int n, const int* __restrict__ order, const T* __restrict__ src,
                                                             T* __restrict__ grec)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gsz)
        grec[(size_t)kGrpMax * p + kGrpMax - 1] = src[order[p]];
}

// diagonal of a plan <- src (natural order): LLSolve keeps the inverse diagonal of the factor there
template <typename T>
static int plan_set_diag(TriPlan* P, const T* src)
{
    Backend& b = backend();
    if(P->n == 0)
        return RAMD_OK;
    if(P->ct && P->ct_rec && P->ct_grp)
    {
        hipLaunchKernelGGL((k_ct_grec_set_diag<T>), dim3(ew_grid(P->n)), dim3(kBlock), 0, b.cur, P->n, P->order, src, (T*)P->diag);
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    if(P->ct && P->ct_rec)
    {
        const int      lpr = P->ct_wmax > 8 ? 8 : 1;
        const int      wl  = lpr == 8 ? 4 : (P->ct_wmax <= 3 ? 3 : (P->ct_wmax <= 4 ? 4 : 8));
        const unsigned nb  = (unsigned)(((int64_t)P->ct_nsteps * 64 + kBlock - 1) / kBlock);
        const bool     dsep = lpr == 1 && (wl == 3 ? CtRec<T, 3>::DSEP : (wl == 4 ? CtRec<T, 4>::DSEP : CtRec<T, 8>::DSEP));
        if(dsep) // the diagonal has its own position-order array
        {
            hipLaunchKernelGGL((k_gather_diag<T>), dim3(ew_grid(P->n)), dim3(kBlock), 0, b.cur, P->n, P->order, src, (T*)P->diag);
            RAMD_HIP(hipGetLastError());
            return RAMD_OK;
        }
#define CT_SET_DIAG(WLL, LP)                                                                                                \
    hipLaunchKernelGGL((k_ct_rec_set_diag<T, WLL, LP>), dim3(nb), dim3(kBlock), 0, b.cur, P->ct_nsteps, P->ct_step_rec,     \
                       P->order, src, (char*)P->eval)
        if(lpr == 8)
            CT_SET_DIAG(4, 8);
        else if(wl == 3)
            CT_SET_DIAG(3, 1);
        else if(wl == 4)
            CT_SET_DIAG(4, 1);
        else
            CT_SET_DIAG(8, 1);
#undef CT_SET_DIAG
    }
    else
        hipLaunchKernelGGL((k_gather_diag<T>), dim3(ew_grid(P->n)), dim3(kBlock), 0, b.cur, P->n, P->order, src, (T*)P->diag);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

template <typename T>
static int ll_analyse_t(ramd_mat_s* m)
{
    TriState* st = nullptr;
    RAMD_TRY(tri_get(m, &st));
    st->haveLL = false;
    RAMD_TRY(build_plan<T>(m, st, &st->LLf, true)); // forward: row sweep on L
    ramd_mat_s* lt = new ramd_mat_s;
    lt->dtype      = m->dtype;
    int s          = transpose_into<T>(m, lt);
    if(s == RAMD_OK)
    {
        TriState* stt = nullptr;
        // the backward plan lives in THIS matrix' state; the transposed copy is only its input (own ticket counter)
        s = tri_get(lt, &stt);
        if(s == RAMD_OK)
            s = build_plan<T>(lt, stt, &st->LLb, false, true); // entries of a row of L^T in DESCENDING row order
    }
    tri_release(lt);
    mat_free_csr(lt);
    delete lt;
    RAMD_TRY(s);
    dev_free(&st->ll_rhs_idx);
    RAMD_TRY(dev_alloc(&st->ll_rhs_idx, m->nrow));
    if(m->nrow > 0)
        hipLaunchKernelGGL(k_compose_idx, dim3(ew_grid(m->nrow)), dim3(kBlock), 0, backend().cur, m->nrow, st->LLb.order,
                           st->LLf.pos, st->ll_rhs_idx);
    RAMD_HIP(hipGetLastError());
    st->ll_diag_src = nullptr;
    st->haveLL      = true;
    return RAMD_OK;
}

template <typename T>
static int ll_solve_t(ramd_mat_s* m, const T* in, const T* inv_diag, T* out)
{
    TriState* st = tri_state(m);
    if(!st || !st->haveLL)
        RAMD_FAIL(RAMD_ERR_STATE, "LLSolve before LLAnalyse");
    Backend& b = backend();
    if(m->nrow == 0)
        return RAMD_OK;
    if(st->ll_diag_src != (const void*)inv_diag) // the plans keep the inverse diagonal in position order
    {
        RAMD_TRY(plan_set_diag<T>(&st->LLf, inv_diag));
        RAMD_TRY(plan_set_diag<T>(&st->LLb, inv_diag));
        st->ll_diag_src = (const void*)inv_diag;
    }
    // L y = b with y_i scaled by inv_diag_i, y kept in position order; then L^T x = y, scaled, natural order out
    RAMD_TRY(run_plan<T>(st, &st->LLf, false, in, st->LLf.order, nullptr, true, &st->LLb));
    return run_plan<T>(st, &st->LLb, false, (const T*)st->LLf.w, st->ll_rhs_idx, out, true, nullptr, st->LLf.prefilled_next,
                       &st->LLf);
}

// ---------------------------------------------------------------- iterative triangular solves
// TriSolverAlg_Iterative (solver.hpp:33-64): host_sparse.cpp:195-530 csritsv, Jacobi sweeps
//     y <- y + D^-1 (x - (D + T) y)
// on one triangle, started from the content of y, until `cap` sweeps are done or the sweep's max-norm figure is
// <= tol.  Every sweep is one bandwidth-bound pass (lane per row over a natural-order sliced-ELL triangle); the
// stopping test runs on the device (last workgroup of a sweep), so a solve is a fixed sequence of launches without
// host round trips: sweeps after the stop return at once.
struct ItCtl
{
    int                cap; // sweeps allowed ("max_iter", shared by both stages as in the reference)
    int                sweeps; // sweeps done in the current stage
    unsigned           blocks_done;
    int                pad;
    unsigned long long mx; // bits of the running max (as double, >= 0)
};

__global__ void k_it_begin(ItCtl* c, int cap, int first_stage)
{
    if(first_stage)
        c->cap = cap;
    c->sweeps      = 0;
    c->blocks_done = 0u;
    c->mx          = 0ull;
}

// MODE 0 unit diagonal                       y = x - sum                                  figure: max |y - y_p|
// MODE 1 non-unit, diagonal LAST  (lower)    r = x - sum, y = y_p + r/d ... (see below)   figure: max |r|
// MODE 2 non-unit, diagonal FIRST (upper)
// MODE 3 transposed lower (its rows = columns of L, diagonal first): no single-entry branch, and the figure is
//        max(max_{i<n-1} |h_i|, |r_{n-1}|) exactly as the reference computes it (host_sparse.cpp:487)
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void k_itsv(int n, const int* __restrict__ slice_off,
                                                 const int* __restrict__ ecol, const T* __restrict__ eval,
                                                 const T* __restrict__ diag, const T* __restrict__ x,
                                                 const T* __restrict__ yp, T* __restrict__ y, ItCtl* ctl, int sweep,
                                                 int use_tol, double tol)
{
    if(sweep >= __hip_atomic_load(&ctl->cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        return;
    double fig = 0.0;
    // a bounded grid walks the row chunks: one closing atomic pair per workgroup, not per 256 rows
    for(int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (int64_t)gridDim.x * kBlock)
    {
        const int64_t s    = t >> 6;
        const int     lane = (int)(t & 63);
        const int     b0   = slice_off[s];
        const int     wd   = (slice_off[s + 1] - b0) >> 6;
        const T       xi   = x[t];
        const T       ypi  = yp[t];
        T             sum  = (T)0;
        T             idg  = (T)1;
        bool          off  = false;
        if(MODE != 0)
        {
            const T d = diag[t];
            idg       = (T)1 / d;
            if(MODE != 1)
                sum += d * ypi;
        }
        constexpr int W = 4;
        for(int k0 = 0; k0 < wd; k0 += W)
        {
            int c[W];
            T   a[W];
#pragma unroll
            for(int e = 0; e < W; ++e)
            {
                const bool in = k0 + e < wd;
                c[e]          = in ? nt_load(ecol + b0 + (k0 + e) * 64 + lane) : -1;
                a[e]          = in ? nt_load(eval + b0 + (k0 + e) * 64 + lane) : (T)0;
            }
            T g[W];
#pragma unroll
            for(int e = 0; e < W; ++e)
                g[e] = (c[e] >= 0) ? yp[c[e]] : (T)0;
#pragma unroll
            for(int e = 0; e < W; ++e)
                if(c[e] >= 0)
                {
                    sum += a[e] * g[e];
                    off = true;
                }
        }
        T yi;
        T f;
        if(MODE == 0)
        {
            yi = xi - sum;
            f  = yi - ypi;
        }
        else if(MODE == 3)
        {
            const T r = xi - sum;
            const T h = idg * r;
            yi        = h + ypi;
            f         = (t < n - 1) ? h : r;
        }
        else
        {
            if(MODE == 1)
                sum += diag[t] * ypi;
            if(off)
            {
                const T r = xi - sum;
                yi        = ypi + idg * r;
                f         = r;
            }
            else
            {
                yi = idg * xi;
                f  = xi - ypi / idg;
            }
        }
        nt_store(yi, y + t);
        f = f < (T)0 ? -f : f;
        if(f == f) // std::max keeps the old value against a NaN
            fig = fmax(fig, (double)f);
    }
    // workgroup max -> device max -> the last workgroup closes the sweep
#pragma unroll
    for(int o = 32; o > 0; o >>= 1)
        fig = fmax(fig, __shfl_xor(fig, o, 64));
    __shared__ double wmax[kBlock / 64];
    if((threadIdx.x & 63) == 0)
        wmax[threadIdx.x >> 6] = fig;
    __syncthreads();
    if(threadIdx.x == 0)
    {
#pragma unroll
        for(int w = 1; w < kBlock / 64; ++w)
            fig = fmax(fig, wmax[w]);
        atomicMax(&ctl->mx, (unsigned long long)__double_as_longlong(fig));
        __threadfence();
        const unsigned done = atomicAdd(&ctl->blocks_done, 1u);
        if(done == gridDim.x - 1)
        {
            const double mx = __longlong_as_double(
                (long long)__hip_atomic_load(&ctl->mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if(use_tol && mx <= tol)
                __hip_atomic_store(&ctl->cap, sweep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctl->sweeps      = sweep + 1;
            ctl->blocks_done = 0u;
            ctl->mx          = 0ull;
        }
    }
}

// an odd number of sweeps leaves the result in the scratch buffer
template <typename T>
__global__ __launch_bounds__(kBlock) void k_it_finish(int n, const ItCtl* ctl, const T* __restrict__ buf,
                                                      T* __restrict__ y)
{
    if((ctl->sweeps & 1) == 0)
        return;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        y[t] = buf[t];
}

__global__ __launch_bounds__(kBlock) void k_any_zero(int n, const double* d64, const float* d32, int* flag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        if(d64 ? (d64[t] == 0.0) : (d32[t] == 0.0f))
            *flag = 1;
}

template <typename T>
static int it_stage(TriState* st, TriPlan* P, int mode, int max_iter, double tol, bool use_tol, bool first, const T* x,
                    T* y)
{
    Backend&  b   = backend();
    const int n   = P->n;
    ItCtl*    ctl = (ItCtl*)st->it_ctl;
    T*        buf = (T*)st->it_buf;
    hipLaunchKernelGGL(k_it_begin, dim3(1), dim3(1), 0, b.cur, ctl, max_iter, first ? 1 : 0);
    const unsigned nb = std::min(nblocks_of(n), 256u * 16u);
    // the tolerance is compared in the value type (numeric_traits_t<T>), the figure travels as double
    const double tol_t = (double)(T)tol;
    for(int k = 0; k < max_iter; ++k)
    {
        const T* src = (k & 1) ? buf : y;
        T*       dst = (k & 1) ? y : buf;
#define ITSV(M)                                                                                                  \
    hipLaunchKernelGGL((k_itsv<T, M>), dim3(nb), dim3(kBlock), 0, b.cur, n, P->slice_off, P->ecol, (const T*)P->eval, \
                       (const T*)P->diag, x, src, dst, ctl, k, use_tol ? 1 : 0, tol_t)
        switch(mode)
        {
        case 0: ITSV(0); break;
        case 1: ITSV(1); break;
        case 2: ITSV(2); break;
        default: ITSV(3); break;
        }
#undef ITSV
    }
    hipLaunchKernelGGL((k_it_finish<T>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, ctl, (const T*)buf, y);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

template <typename T>
static int it_zero_check(TriState::ItSlot* sl, TriPlan* P)
{
    Backend& b = backend();
    if(P->nodiag)
    {
        sl->zero_diag = true;
        return RAMD_OK;
    }
    int* flag = nullptr;
    RAMD_TRY(dev_alloc(&flag, 1));
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), b.cur);
    hipLaunchKernelGGL(k_any_zero, dim3(ew_grid(P->n)), dim3(kBlock), 0, b.cur, P->n,
                       sizeof(T) == 8 ? (const double*)P->diag : nullptr, sizeof(T) == 4 ? (const float*)P->diag : nullptr,
                       flag);
    int h = 0;
    if(e == hipSuccess)
        e = hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, b.cur);
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&flag);
    RAMD_HIP(e);
    if(h)
        sl->zero_diag = true;
    return RAMD_OK;
}

// kind 1 LU (L unit, U non-unit) | 2 LL (L non-unit, L^T) | 3 L | 4 U     (host_matrix_csr.cpp:1469-1543, :1652-1726,
// :1846-1894, :1970-2018: the reference only sizes a buffer and zero-fills tmp_vec_; the triangles are packed here)
template <typename T>
static int it_analyse_t(ramd_mat_s* m, int kind, bool unit)
{
    Backend&  b  = backend();
    TriState* st = nullptr;
    RAMD_TRY(tri_get(m, &st));
    const int idx = it_slot_of(kind);
    it_release(st, idx);
    TriState::ItSlot* sl = &st->it[idx];
    const int         n  = m->nrow;
    sl->unit             = unit;
    sl->zero_diag        = false;
    if(n == 0 || m->nnz == 0)
    {
        sl->kind = kind;
        return RAMD_OK;
    }
    int s = RAMD_OK;
    if(kind == 1)
    {
        s = build_plan<T>(m, st, &sl->A, true, false, true);
        if(s == RAMD_OK)
            s = build_plan<T>(m, st, &sl->B, false, false, true);
        if(s == RAMD_OK)
            s = it_zero_check<T>(sl, &sl->B);
    }
    else if(kind == 2)
    {
        s = build_plan<T>(m, st, &sl->A, true, false, true);
        if(s == RAMD_OK)
            s = it_zero_check<T>(sl, &sl->A);
        if(s == RAMD_OK)
        {
            // L^T from the lower triangle of m only (the reference's transposed sweep runs over [row begin, diagonal])
            ramd_mat_s* lt = new ramd_mat_s;
            lt->dtype      = m->dtype;
            s              = transpose_lower_into<T>(m, lt);
            if(s == RAMD_OK)
            {
                TriState* stt = nullptr;
                s             = tri_get(lt, &stt);
                if(s == RAMD_OK)
                    s = build_plan<T>(lt, stt, &sl->B, false, false, true);
            }
            tri_release(lt);
            mat_free_csr(lt);
            delete lt;
        }
    }
    else
    {
        s = build_plan<T>(m, st, &sl->A, kind == 3, false, true);
        if(s == RAMD_OK && !unit)
            s = it_zero_check<T>(sl, &sl->A);
    }
    if(s == RAMD_OK && !st->it_ctl && hipMalloc(&st->it_ctl, sizeof(ItCtl)) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && !st->it_buf && cached_malloc(&st->it_buf, (size_t)n * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && (kind == 1 || kind == 2))
    {
        if(cached_malloc(&st->it_tmp, (size_t)n * sizeof(T) + kPad) != hipSuccess)
            s = RAMD_ERR_HIP;
        else if(hipMemsetAsync(st->it_tmp, 0, (size_t)n * sizeof(T), b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    if(s != RAMD_OK)
    {
        it_release(st, idx);
        return s;
    }
    sl->kind = kind;
    return RAMD_OK;
}

template <typename T>
static int it_solve_t(ramd_mat_s* m, int kind, int max_iter, double tol, bool use_tol, const T* in, T* out)
{
    TriState*         st = tri_state(m);
    TriState::ItSlot* sl = st ? &st->it[it_slot_of(kind)] : nullptr;
    if(!sl || sl->kind != kind)
        RAMD_FAIL(RAMD_ERR_STATE, "iterative triangular solve before its analysis");
    if(m->nnz <= 0 || m->nrow == 0 || max_iter <= 0)
        return RAMD_OK; // nnz == 0: the reference leaves out untouched (host_matrix_csr.cpp:1571)
    if(sl->zero_diag)
        return RAMD_OK; // zero pivot: csritsv returns without touching y (host_sparse.cpp:388-391, :425-429)
    T* tmp = (T*)st->it_tmp;
    switch(kind)
    {
    case 1:
        RAMD_TRY(it_stage<T>(st, &sl->A, 0, max_iter, tol, use_tol, true, in, tmp));
        return it_stage<T>(st, &sl->B, 2, max_iter, tol, use_tol, false, (const T*)tmp, out);
    case 2:
        RAMD_TRY(it_stage<T>(st, &sl->A, 1, max_iter, tol, use_tol, true, in, tmp));
        return it_stage<T>(st, &sl->B, 3, max_iter, tol, use_tol, false, (const T*)tmp, out);
    case 3: return it_stage<T>(st, &sl->A, sl->unit ? 0 : 1, max_iter, tol, use_tol, true, in, out);
    default: return it_stage<T>(st, &sl->A, sl->unit ? 0 : 2, max_iter, tol, use_tol, true, in, out);
    }
}

// matrix_algebra.hip: LocalMatrix::Transpose
int mat_transpose(const ramd_mat_s* m, ramd_mat_s* t)
{
    return (m->dtype == RAMD_F64) ? transpose_into<double>(m, t) : transpose_into<float>(m, t);
}

} // namespace ramd

using namespace ramd;

static int check_tri(ramd_mat_t m, ramd_vec_t in, ramd_vec_t out)
{
    if(!m || !in || !out)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(in->dtype != m->dtype || out->dtype != m->dtype || in->n != m->ncol || out->n != m->nrow
       || m->nrow != m->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "triangular solve: sizes/types do not match a square matrix");
    if(in == out)
        RAMD_FAIL(RAMD_ERR_ARG, "triangular solve: in and out must differ");
    return RAMD_OK;
}

extern "C" {

int ramd_mat_ilu0_factorize(ramd_mat_t m)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(m->nrow != m->ncol || m->nnz <= 0)
        RAMD_FAIL(RAMD_ERR_ARG, "ILU0Factorize: need a square, non-empty matrix (the reference asserts)");
    // one wave per row (every lane keeps an entry of the row, pivot rows are searched by bisection) pays for rows of FE
    // size: measured on the shell surrogate (35 entries per row) 2.9 s -> 0.1 s per factorisation, on the 7-point operator
    // 8 % slower than the thread-per-row sweep.  Same operations per entry in the same order: bit-identical factors.
    static const int wave_env = getenv("RAMD_ILU0_WAVE") ? atoi(getenv("RAMD_ILU0_WAVE")) : -1; // (0 / 1: force, A/B experiments)
    const bool       wave_rows = wave_env >= 0 ? wave_env != 0 : (m->nnz >= (int64_t)12 * m->nrow);
    if(wave_rows)
        return (m->dtype == RAMD_F64) ? ilu0_long_rows_t<double>(m) : ilu0_long_rows_t<float>(m);
    return (m->dtype == RAMD_F64) ? ilu0_t<double>(m) : ilu0_t<float>(m);
}

int ramd_mat_ilup_factorize(ramd_mat_t m, int p, int level)
{
    if(!m || p < 0)
        RAMD_FAIL(RAMD_ERR_ARG, "ILUpFactorize: matrix handle, p >= 0");
    if(p == 0)
        return ramd_mat_ilu0_factorize(m); // local_matrix.cpp:3920-3923
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(m->nrow != m->ncol || m->nnz <= 0)
        RAMD_FAIL(RAMD_ERR_ARG, "ILUpFactorize: need a square, non-empty matrix");
    return (m->dtype == RAMD_F64) ? ilup_t<double>(m, p, level != 0) : ilup_t<float>(m, p, level != 0);
}

int ramd_mat_ic_factorize(ramd_mat_t m, ramd_vec_t inv_diag)
{
    if(!m || !inv_diag)
        RAMD_FAIL(RAMD_ERR_ARG, "null handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(m->nrow != m->ncol || m->nnz <= 0 || inv_diag->dtype != m->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "ICFactorize: need a square, non-empty lower-triangular matrix and a vector of its type");
    RAMD_TRY(ramd_vec_allocate(inv_diag, m->nrow));
    return (m->dtype == RAMD_F64) ? ic0_t<double>(m, inv_diag) : ic0_t<float>(m, inv_diag);
}

int ramd_mat_ll_analyse(ramd_mat_t m)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(m->nrow != m->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "LLAnalyse: square matrix expected");
    return (m->dtype == RAMD_F64) ? ll_analyse_t<double>(m) : ll_analyse_t<float>(m);
}

int ramd_mat_ll_analyse_clear(ramd_mat_t m)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    TriState* st = tri_state(m);
    if(st)
    {
        st->LLf.release();
        st->LLb.release();
        dev_free(&st->ll_rhs_idx);
        st->haveLL = false;
    }
    return RAMD_OK;
}

int ramd_mat_ll_solve(ramd_mat_t m, ramd_vec_t in, ramd_vec_t inv_diag, ramd_vec_t out)
{
    RAMD_TRY(check_tri(m, in, out));
    if(!inv_diag || inv_diag->dtype != m->dtype || inv_diag->n != m->nrow)
        RAMD_FAIL(RAMD_ERR_ARG, "LLSolve: inverse diagonal vector of the matrix' size and type expected");
    if(m->dtype == RAMD_F64)
        return ll_solve_t<double>(m, (const double*)in->d, (const double*)inv_diag->d, (double*)out->d);
    return ll_solve_t<float>(m, (const float*)in->d, (const float*)inv_diag->d, (float*)out->d);
}

// ---- TriSolverAlg_Iterative entry points (one per backend virtual of the reference)
static int it_analyse(ramd_mat_t m, int kind, int unit)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(m->nrow != m->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "iterative triangular analysis: square matrix expected");
    return (m->dtype == RAMD_F64) ? it_analyse_t<double>(m, kind, unit != 0) : it_analyse_t<float>(m, kind, unit != 0);
}
static int it_clear(ramd_mat_t m, int kind)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    TriState* st = tri_state(m);
    if(st && st->it[it_slot_of(kind)].kind == kind)
        it_release(st, it_slot_of(kind));
    return RAMD_OK;
}
static int it_solve(ramd_mat_t m, int kind, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out)
{
    RAMD_TRY(check_tri(m, in, out));
    if(in == out)
        RAMD_FAIL(RAMD_ERR_ARG, "iterative triangular solve: in and out must differ");
    if(m->dtype == RAMD_F64)
        return it_solve_t<double>(m, kind, max_iter, tol, use_tol != 0, (const double*)in->d, (double*)out->d);
    return it_solve_t<float>(m, kind, max_iter, tol, use_tol != 0, (const float*)in->d, (float*)out->d);
}
int ramd_mat_it_lu_analyse(ramd_mat_t m) { return it_analyse(m, 1, 0); }
int ramd_mat_it_lu_analyse_clear(ramd_mat_t m) { return it_clear(m, 1); }
int ramd_mat_it_lu_solve(ramd_mat_t m, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out)
{
    return it_solve(m, 1, max_iter, tol, use_tol, in, out);
}
int ramd_mat_it_ll_analyse(ramd_mat_t m) { return it_analyse(m, 2, 0); }
int ramd_mat_it_ll_analyse_clear(ramd_mat_t m) { return it_clear(m, 2); }
int ramd_mat_it_ll_solve(ramd_mat_t m, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out)
{
    return it_solve(m, 2, max_iter, tol, use_tol, in, out);
}
int ramd_mat_it_l_analyse(ramd_mat_t m, int diag_unit) { return it_analyse(m, 3, diag_unit); }
int ramd_mat_it_l_analyse_clear(ramd_mat_t m) { return it_clear(m, 3); }
int ramd_mat_it_l_solve(ramd_mat_t m, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out)
{
    return it_solve(m, 3, max_iter, tol, use_tol, in, out);
}
int ramd_mat_it_u_analyse(ramd_mat_t m, int diag_unit) { return it_analyse(m, 4, diag_unit); }
int ramd_mat_it_u_analyse_clear(ramd_mat_t m) { return it_clear(m, 4); }
int ramd_mat_it_u_solve(ramd_mat_t m, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out)
{
    return it_solve(m, 4, max_iter, tol, use_tol, in, out);
}

// statistics of the plans of the most recent LUAnalyse / LAnalyse / UAnalyse of this process (bench.py prints them, so that a run
// on a matrix nobody here has seen comes back with a diagnosis of its triangular solves, not just a rate)
static long long g_tri_stats[2][16];
static void tri_note_stats(const ramd_mat_s* m, const TriPlan* P, int which)
{
    long long* o = g_tri_stats[which];
    for(int i = 0; i < 16; ++i)
        o[i] = 0;
    o[1] = m->nrow;
    if(P->lat)
    {
        LatInfo li;
        lat_info(P->lat, &li);
        o[0] = 4;
        o[2] = (long long)li.nx + li.ny + li.nz - 2;
        o[3] = li.npencil;
        o[4] = (long long)li.npencil * li.nsteps;
        o[5] = (long long)(li.face_bytes / val_size(m->dtype));
        o[6] = 64LL * li.nsteps;
        o[7] = 3;
        o[8] = 1;
        o[9] = li.nx, o[10] = li.ny, o[11] = li.nz;
        o[12] = (long long)(li.coef_bytes + li.face_bytes);
        return;
    }
    if(P->box)
    {
        BoxInfo bi;
        box_info(P->box, &bi);
        o[0] = 7;
        o[2] = (long long)bi.nx + 2LL * bi.ny + 4LL * bi.nz - 6; // dependency levels: planes x + 2 y + 4 z
        o[3] = bi.ntiles;
        o[4] = (long long)bi.ntiles * bi.nsteps;
        o[6] = 64;
        o[7] = 13;
        o[8] = 1;
        o[9] = bi.nx, o[10] = bi.ny, o[11] = bi.nz;
        o[12] = (long long)bi.coef_bytes;
        return;
    }
    o[0] = !P->ct ? (P->sf ? 6 : 1) : (P->ct_grp ? 3 : 2); // (5 was the band form of round 5)
    o[2] = P->nlevels;
    if(P->sf)
    {
        o[3] = P->sf->nunits;  // units = whole row groups of one group level, one wave each
        o[4] = P->sf->ngroups; // row groups (steps of the dependency graph)
        o[6] = 64 / P->sf->lpr;
        o[7] = P->sf->wout;
        o[8] = P->sf->lpr;
        o[9] = P->sf->maxm;
        o[12] = P->st_why;
        o[14] = (long long)(P->sf->nentries * (4 + (long long)val_size(m->dtype)));
        return;
    }
    o[12] = P->ct ? 0 : P->st_why;
    o[7] = P->ct_wmax;
    if(P->ct)
    {
        o[3] = P->ct_ntiles;
        o[4] = P->ct_nsteps;
        o[5] = P->st_ext;
        o[6] = P->ct_dims[0];
        o[8] = P->ct_grp ? kGrpLPR : (P->ct_wmax > 8 ? 8 : 1);
        o[9] = P->st_box[0], o[10] = P->st_box[1], o[11] = P->st_box[2];
        o[13] = P->st_chains;
        o[14] = P->ct_dims[1];
        o[15] = P->ct_dims[3];
    }
}

int ramd_tri_plan_stats(int which, long long* out16)
{
    if(which < 0 || which > 1 || !out16)
        RAMD_FAIL(RAMD_ERR_ARG, "ramd_tri_plan_stats(which = 0 lower / 1 upper, out[16])");
    for(int i = 0; i < 16; ++i)
        out16[i] = g_tri_stats[which][i];
    return RAMD_OK;
}

int ramd_mat_lu_analyse(ramd_mat_t m)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    TriState* st = nullptr;
    RAMD_TRY(tri_get(m, &st));
    m->lu_analysed = false;
    st->haveL = st->haveU = false;
    // both triangles on a lattice: the pencil form, L then U in place on the output vector (no position order, no scratch)
    {
        st->L.release();
        st->U.release();
        int sl = m->dtype == RAMD_F64 ? lat_build<double>(m, true, true, &st->L.lat) : lat_build<float>(m, true, true, &st->L.lat);
        if(sl == RAMD_OK)
            sl = m->dtype == RAMD_F64 ? lat_build<double>(m, false, false, &st->U.lat) : lat_build<float>(m, false, false, &st->U.lat);
        if(sl == RAMD_OK)
        {
            st->L.n = st->U.n = m->nrow;
            st->haveL = st->haveU = true;
            dev_free(&st->lu_rhs_idx);
            dev_free(&st->l_order_cache);
            dev_free(&st->l_level_cache);
            m->lu_analysed = true;
            tri_note_stats(m, &st->L, 0);
            tri_note_stats(m, &st->U, 1);
            return RAMD_OK;
        }
        st->L.release();
        st->U.release();
        if(sl != RAMD_ERR_UNSUPPORTED)
            return sl;
        // ... or on the 27-point stencil: L into the plan's own vector, U from there into the output vector
        sl = m->dtype == RAMD_F64 ? box_build<double>(m, true, true, &st->L.box) : box_build<float>(m, true, true, &st->L.box);
        if(sl == RAMD_OK)
            sl = m->dtype == RAMD_F64 ? box_build<double>(m, false, false, &st->U.box) : box_build<float>(m, false, false, &st->U.box);
        if(sl == RAMD_OK)
        {
            st->L.n = st->U.n = m->nrow;
            st->haveL = st->haveU = true;
            dev_free(&st->lu_rhs_idx);
            dev_free(&st->l_order_cache);
            dev_free(&st->l_level_cache);
            m->lu_analysed = true;
            tri_note_stats(m, &st->L, 0);
            tri_note_stats(m, &st->U, 1);
            return RAMD_OK;
        }
        st->L.release();
        st->U.release();
        if(sl != RAMD_ERR_UNSUPPORTED)
            return sl;
    }
    if(m->dtype == RAMD_F64)
    {
        RAMD_TRY(build_plan<double>(m, st, &st->L, true));
        RAMD_TRY(build_plan<double>(m, st, &st->U, false));
    }
    else
    {
        RAMD_TRY(build_plan<float>(m, st, &st->L, true));
        RAMD_TRY(build_plan<float>(m, st, &st->U, false));
    }
    st->haveL = st->haveU = true;
    dev_free(&st->lu_rhs_idx);
    RAMD_TRY(dev_alloc(&st->lu_rhs_idx, m->nrow));
    if(m->nrow > 0)
        hipLaunchKernelGGL(k_compose_idx, dim3(ew_grid(m->nrow)), dim3(kBlock), 0, backend().cur, m->nrow,
                           st->U.order, st->L.pos, st->lu_rhs_idx);
    RAMD_HIP(hipGetLastError());
    m->lu_analysed = true;
    tri_note_stats(m, &st->L, 0);
    tri_note_stats(m, &st->U, 1);
    return RAMD_OK;
}

int ramd_mat_lu_analyse_clear(ramd_mat_t m)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    tri_release(m);
    m->lu_analysed = m->l_analysed = m->u_analysed = false;
    return RAMD_OK;
}

int ramd_mat_lu_solve(ramd_mat_t m, ramd_vec_t in, ramd_vec_t out)
{
    RAMD_TRY(check_tri(m, in, out));
    if(!m->lu_analysed)
        RAMD_FAIL(RAMD_ERR_STATE, "LUSolve before LUAnalyse");
    TriState* st = tri_state(m);
    if(!st || !st->haveL || !st->haveU || (st->L.lat != nullptr) != (st->U.lat != nullptr)
       || (st->L.box != nullptr) != (st->U.box != nullptr)
       || (!st->L.lat && !st->L.box && (!st->lu_rhs_idx || !st->L.order || !st->U.order)))
        RAMD_FAIL(RAMD_ERR_STATE, "LUSolve: the plans of LUAnalyse are incomplete");
    if(st->L.box && st->U.box)
    {
        if(!box_is_unit(st->L.box) || box_is_unit(st->U.box))
            RAMD_FAIL(RAMD_ERR_STATE, "LUSolve: the 27-point plans of LUAnalyse carry the wrong diagonal flags");
        if(m->dtype == RAMD_F64)
        {
            RAMD_TRY(box_run<double>(st->L.box, (const double*)in->d, (double*)box_scratch(st->L.box)));
            return box_run<double>(st->U.box, (const double*)box_scratch(st->L.box), (double*)out->d);
        }
        RAMD_TRY(box_run<float>(st->L.box, (const float*)in->d, (float*)box_scratch(st->L.box)));
        return box_run<float>(st->U.box, (const float*)box_scratch(st->L.box), (float*)out->d);
    }
    if(st->L.lat && st->U.lat)
    {
        if(!lat_is_unit(st->L.lat) || lat_is_unit(st->U.lat))
            RAMD_FAIL(RAMD_ERR_STATE, "LUSolve: the lattice plans of LUAnalyse carry the wrong diagonal flags");
        if(m->dtype == RAMD_F64)
        {
            RAMD_TRY(lat_run<double>(st->L.lat, (const double*)in->d, (double*)out->d));
            return lat_run<double>(st->U.lat, (const double*)out->d, (double*)out->d);
        }
        RAMD_TRY(lat_run<float>(st->L.lat, (const float*)in->d, (float*)out->d));
        return lat_run<float>(st->U.lat, (const float*)out->d, (float*)out->d);
    }
    if(m->dtype == RAMD_F64)
    {
        // L y = b (unit diagonal), y kept in L-position order inside the plan's scratch
        RAMD_TRY(run_plan<double>(st, &st->L, true, (const double*)in->d, st->L.order, nullptr, false, &st->U));
        // U x = y (stored diagonal), x written back in natural order
        return run_plan<double>(st, &st->U, false, (const double*)st->L.w, st->lu_rhs_idx,
                                (double*)out->d, false, nullptr, st->L.prefilled_next, &st->L);
    }
    RAMD_TRY(run_plan<float>(st, &st->L, true, (const float*)in->d, st->L.order, nullptr, false, &st->U));
    return run_plan<float>(st, &st->U, false, (const float*)st->L.w, st->lu_rhs_idx, (float*)out->d, false, nullptr,
                           st->L.prefilled_next, &st->L);
}

int ramd_mat_l_analyse(ramd_mat_t m, int diag_unit)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    TriState* st = nullptr;
    RAMD_TRY(tri_get(m, &st));
    m->l_analysed = false;
    st->Ls.release();
    int sl = m->dtype == RAMD_F64 ? lat_build<double>(m, true, diag_unit != 0, &st->Ls.lat)
                                  : lat_build<float>(m, true, diag_unit != 0, &st->Ls.lat);
    if(sl == RAMD_ERR_UNSUPPORTED)
        sl = m->dtype == RAMD_F64 ? box_build<double>(m, true, diag_unit != 0, &st->Ls.box)
                                  : box_build<float>(m, true, diag_unit != 0, &st->Ls.box);
    if(sl == RAMD_OK)
        st->Ls.n = m->nrow;
    else if(sl != RAMD_ERR_UNSUPPORTED)
        return sl;
    else if(m->dtype == RAMD_F64)
        RAMD_TRY(build_plan<double>(m, st, &st->Ls, true));
    else
        RAMD_TRY(build_plan<float>(m, st, &st->Ls, true));
    m->l_analysed  = true;
    m->l_diag_unit = diag_unit != 0;
    tri_note_stats(m, &st->Ls, 0);
    return RAMD_OK;
}

int ramd_mat_l_analyse_clear(ramd_mat_t m)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(TriState* st = tri_state(m))
        st->Ls.release();
    m->l_analysed  = false;
    m->l_diag_unit = true; // host LAnalyseClear resets to unit (host_matrix_csr.cpp:1350-1354)
    return RAMD_OK;
}

int ramd_mat_l_solve(ramd_mat_t m, ramd_vec_t in, ramd_vec_t out)
{
    RAMD_TRY(check_tri(m, in, out));
    if(!m->l_analysed)
        RAMD_FAIL(RAMD_ERR_STATE, "LSolve before LAnalyse");
    TriState* st = tri_state(m);
    if(st->Ls.lat)
        return m->dtype == RAMD_F64 ? lat_run<double>(st->Ls.lat, (const double*)in->d, (double*)out->d)
                                    : lat_run<float>(st->Ls.lat, (const float*)in->d, (float*)out->d);
    if(st->Ls.box)
        return m->dtype == RAMD_F64 ? box_run<double>(st->Ls.box, (const double*)in->d, (double*)out->d)
                                    : box_run<float>(st->Ls.box, (const float*)in->d, (float*)out->d);
    if(m->dtype == RAMD_F64)
        return run_plan<double>(st, &st->Ls, m->l_diag_unit, (const double*)in->d, st->Ls.order,
                                (double*)out->d);
    return run_plan<float>(st, &st->Ls, m->l_diag_unit, (const float*)in->d, st->Ls.order, (float*)out->d);
}

int ramd_mat_u_analyse(ramd_mat_t m, int diag_unit)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    TriState* st = nullptr;
    RAMD_TRY(tri_get(m, &st));
    m->u_analysed = false;
    st->Us.release();
    int sl = m->dtype == RAMD_F64 ? lat_build<double>(m, false, diag_unit != 0, &st->Us.lat)
                                  : lat_build<float>(m, false, diag_unit != 0, &st->Us.lat);
    if(sl == RAMD_ERR_UNSUPPORTED)
        sl = m->dtype == RAMD_F64 ? box_build<double>(m, false, diag_unit != 0, &st->Us.box)
                                  : box_build<float>(m, false, diag_unit != 0, &st->Us.box);
    if(sl == RAMD_OK)
        st->Us.n = m->nrow;
    else if(sl != RAMD_ERR_UNSUPPORTED)
        return sl;
    else if(m->dtype == RAMD_F64)
        RAMD_TRY(build_plan<double>(m, st, &st->Us, false));
    else
        RAMD_TRY(build_plan<float>(m, st, &st->Us, false));
    m->u_analysed  = true;
    m->u_diag_unit = diag_unit != 0;
    tri_note_stats(m, &st->Us, 1);
    return RAMD_OK;
}

int ramd_mat_u_analyse_clear(ramd_mat_t m)
{
    if(!m)
        RAMD_FAIL(RAMD_ERR_ARG, "null matrix handle");
    if(TriState* st = tri_state(m))
        st->Us.release();
    m->u_analysed  = false;
    m->u_diag_unit = false;
    return RAMD_OK;
}

int ramd_mat_u_solve(ramd_mat_t m, ramd_vec_t in, ramd_vec_t out)
{
    RAMD_TRY(check_tri(m, in, out));
    if(!m->u_analysed)
        RAMD_FAIL(RAMD_ERR_STATE, "USolve before UAnalyse");
    TriState* st = tri_state(m);
    if(st->Us.lat)
        return m->dtype == RAMD_F64 ? lat_run<double>(st->Us.lat, (const double*)in->d, (double*)out->d)
                                    : lat_run<float>(st->Us.lat, (const float*)in->d, (float*)out->d);
    if(st->Us.box)
        return m->dtype == RAMD_F64 ? box_run<double>(st->Us.box, (const double*)in->d, (double*)out->d)
                                    : box_run<float>(st->Us.box, (const float*)in->d, (float*)out->d);
    if(m->dtype == RAMD_F64)
        return run_plan<double>(st, &st->Us, m->u_diag_unit, (const double*)in->d, st->Us.order,
                                (double*)out->d);
    return run_plan<float>(st, &st->Us, m->u_diag_unit, (const float*)in->d, st->Us.order, (float*)out->d);
}

} // extern "C"
