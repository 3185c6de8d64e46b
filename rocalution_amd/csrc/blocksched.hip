// blocksched.hip -- hyperplane order of the 256-row blocks for the natural-order sync-free sweeps
// (dependency levels, greedy colouring).
//
// Those sweeps hand row blocks to workgroups in ticket order.  With the natural order only the ~2048 resident
// workgroups (512 Ki rows = two grid planes at 512^3) are in flight and, inside such a window, the lines of a
// plane form a serial chain -- measured 0.3-0.4 s per sweep at 512^3.  The sweep itself only needs the blocks a
// block depends on to have STARTED earlier, so any topological order of the block graph is admissible.  Here the
// blocks are ordered by their level in that graph (wavefront / hyperplane order): all blocks of one level are
// independent and fill the machine.
//   1. device: per block, the distinct blocks its rows depend on (up to kMaxDeps, LDS set; overflow flag)
//   2. host:   L[b] = 1 + max L[dep] in one ascending pass (natural order is topological), counting sort
//   3. device: ticket t works on block order[t]
// A block whose dependency set overflows is pinned behind everything before it (natural-order fallback).
#include "device_utils.hpp"
#include "matrix_impl.hpp"

#include <algorithm>
#include <vector>

namespace ramd
{

constexpr int kMaxDeps = 8;

// LOWER: dependencies are columns < row, blocks counted from the front; else columns > row and the sweep walks
// the rows backwards: block id b stands for rows [n-1-256b-255, n-1-256b] (the mapping of k_levels<false>)
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_block_deps(int nrow, const int* __restrict__ rp,
                                                       const int* __restrict__ ci, int* __restrict__ deps,
                                                       int* __restrict__ overflow)
{
    __shared__ int s_dep[kMaxDeps];
    __shared__ int s_over;
    if(threadIdx.x == 0)
        s_over = 0;
    if(threadIdx.x < kMaxDeps)
        s_dep[threadIdx.x] = -1; // free slot
    __syncthreads();
    const int     me = blockIdx.x;
    const int64_t t  = (int64_t)me * kBlock + threadIdx.x;
    if(t < nrow)
    {
        const int i    = LOWER ? (int)t : (int)(nrow - 1 - t);
        int       last = -1;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int c = ci[j];
            if(LOWER ? (c >= i) : (c <= i))
                continue;
            const int b = LOWER ? (c / kBlock) : ((nrow - 1 - c) / kBlock);
            if(b == me || b == last)
                continue;
            last = b;
            // insert into the small set: claim the first free slot or find the value -- wait-free (a lane never
            // waits for another lane: lanes of one wave would deadlock on that)
            bool placed = false;
            for(int sl = 0; sl < kMaxDeps && !placed; ++sl)
            {
                int expect = -1;
                if(__hip_atomic_compare_exchange_strong(&s_dep[sl], &expect, b, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_WORKGROUP)
                   || expect == b)
                    placed = true;
            }
            if(!placed)
                s_over = 1;
        }
    }
    __syncthreads();
    if(threadIdx.x < kMaxDeps)
        deps[(int64_t)me * kMaxDeps + threadIdx.x] = s_dep[threadIdx.x];
    if(threadIdx.x == 0)
        overflow[me] = s_over;
}

// -> device array order[nblk] (caller frees with dev_free), or nullptr in *out when the natural order is kept
// (small problems, failures: the sweeps work with any admissible order, natural included)
int block_schedule(const ramd_mat_s* m, bool lower, int** out)
{
    *out           = nullptr;
    const int n    = m->nrow;
    const int nblk = (n + kBlock - 1) / kBlock;
    static const int min_blocks = getenv("RAMD_BLOCKSCHED_MIN") ? atoi(getenv("RAMD_BLOCKSCHED_MIN")) : 4096;
    if(nblk < min_blocks) // everything is resident at once anyway (RAMD_BLOCKSCHED_MIN=1: tests)
        return RAMD_OK;
    Backend& b    = backend();
    int*     deps = nullptr;
    int*     over = nullptr;
    int      s    = dev_alloc(&deps, (int64_t)nblk * kMaxDeps);
    if(s == RAMD_OK)
        s = dev_alloc(&over, nblk);
    std::vector<int> hdeps((size_t)nblk * kMaxDeps), hover((size_t)nblk);
    if(s == RAMD_OK)
    {
        if(lower)
            hipLaunchKernelGGL((k_block_deps<true>), dim3(nblk), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, deps, over);
        else
            hipLaunchKernelGGL((k_block_deps<false>), dim3(nblk), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, deps, over);
        if(hipMemcpyAsync(hdeps.data(), deps, sizeof(int) * hdeps.size(), hipMemcpyDeviceToHost, b.cur) != hipSuccess
           || hipMemcpyAsync(hover.data(), over, sizeof(int) * hover.size(), hipMemcpyDeviceToHost, b.cur) != hipSuccess
           || hipStreamSynchronize(b.cur) != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&deps);
    dev_free(&over);
    if(s != RAMD_OK)
        return s;
    std::vector<int> lev((size_t)nblk);
    int              run_max = 0, nlev = 0;
    for(int k = 0; k < nblk; ++k)
    {
        int l = 0;
        if(hover[(size_t)k])
            l = run_max; // unknown dependency set: behind everything seen so far
        else
            for(int e = 0; e < kMaxDeps; ++e)
            {
                const int d = hdeps[(size_t)k * kMaxDeps + e];
                if(d >= 0 && d < k)
                    l = std::max(l, lev[(size_t)d]);
                else if(d >= k) // cannot happen for a triangular dependency; keep the natural order if it does
                    return RAMD_OK;
            }
        lev[(size_t)k] = l + 1;
        run_max        = std::max(run_max, l + 1);
        nlev           = run_max;
    }
    std::vector<int> start((size_t)nlev + 2, 0), order((size_t)nblk);
    for(int k = 0; k < nblk; ++k)
        ++start[(size_t)lev[(size_t)k] + 1];
    for(int l = 1; l <= nlev + 1; ++l)
        start[(size_t)l] += start[(size_t)l - 1];
    for(int k = 0; k < nblk; ++k)
        order[(size_t)start[(size_t)lev[(size_t)k]]++] = k;
    int* d_order = nullptr;
    RAMD_TRY(dev_alloc(&d_order, nblk));
    if(hipMemcpyAsync(d_order, order.data(), sizeof(int) * (size_t)nblk, hipMemcpyHostToDevice, b.cur) != hipSuccess
       || hipStreamSynchronize(b.cur) != hipSuccess)
    {
        dev_free(&d_order);
        return RAMD_ERR_HIP;
    }
    *out = d_order;
    return RAMD_OK;
}

// ---------------------------------------------------------------- wave units (<= 64 rows), scheduled on the device
// The sweeps that resolve in-wave dependencies in registers (trisolve.hip: k_levels, k_ct_coords, k_ct_glevels, k_ilu0_rows)
// take their rows in units of one wave, and a unit publishes its rows together.  So a unit must not tie rows together that
// have nothing to do with each other: a unit holding the END of one grid line (or mesh line, or plane) and the START of the
// next makes the next line wait for the whole previous one -- on the 549 x 549-node shell 23 500 units became one serial
// chain (measured 0.19 s for a level sweep whose unit graph is 1 141 levels deep), and any grid whose lines are not a
// multiple of 64 rows would do the same.  Units therefore start afresh at every row that has NO dependency among the 64 rows
// before it (a new line, a new plane), 64 rows apiece from there; such breaks closer than 32 rows to an earlier one are
// ignored (matrices without any locality would fall apart into single rows).
//   The unit graph is small (<= kUnitDeps edges per unit), so its own longest-path sweep -- same register resolution, one
// lane per unit, natural order -- and the sort by level run on the device in a few milliseconds: along a grid line the
// chain costs one wave-time per 64 rows, and the critical path of a sweep is the depth of the UNIT graph times that (at
// 512^3: 8 + 511 + 511 units).
constexpr int kUnitDeps = 12;

void UnitPlan::release()
{
    dev_free(&ustart);
    dev_free(&order);
    nunits = 0;
}

// brk[t] = 1: sweep row t has no dependency among the 64 rows before it (n + 1 entries, the last one 0 for the scans)
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_unit_breaks(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                        int* __restrict__ brk)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += gsz)
    {
        if(t == n)
        {
            brk[t] = 0;
            continue;
        }
        const int i    = LOWER ? (int)t : (int)(n - 1 - t);
        bool      near = false;
        for(int j = rp[i]; j < rp[i + 1] && !near; ++j)
        {
            const int c = ci[j];
            near        = LOWER ? (c < i && c >= i - 64) : (c > i && c <= i + 64);
        }
        brk[t] = (t == 0 || !near) ? 1 : 0;
    }
}

// ... of which those count that have no other break among the 31 rows before them (bscan = exclusive scan of brk)
__global__ __launch_bounds__(kBlock) void k_unit_hard_breaks(int n, const int* __restrict__ brk, const int* __restrict__ bscan,
                                                             int* __restrict__ hard)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += gsz)
    {
        if(t == n)
        {
            hard[t] = 0;
            continue;
        }
        const int64_t lo = t - 31 < 0 ? 0 : t - 31;
        hard[t]          = (t == 0 || (brk[t] && bscan[t] - bscan[lo] == 0)) ? 1 : 0;
    }
}

__global__ __launch_bounds__(kBlock) void k_unit_scatter(int n, const int* __restrict__ flag, const int* __restrict__ scan,
                                                         int* __restrict__ start)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
        if(flag[t])
            start[scan[t]] = (int)t;
}

// a unit starts every 64 rows of a segment (hscan = exclusive scan of hard, seg_start = first row of every segment)
__global__ __launch_bounds__(kBlock) void k_unit_starts(int n, const int* __restrict__ hard, const int* __restrict__ hscan,
                                                        const int* __restrict__ seg_start, int* __restrict__ uflag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += gsz)
    {
        if(t == n)
        {
            uflag[t] = 0;
            continue;
        }
        const int seg = hscan[t] + hard[t] - 1;
        uflag[t]      = (((int)t - seg_start[seg]) & 63) == 0 ? 1 : 0;
    }
}

// distinct units the rows of a unit depend on (uscan = exclusive scan of uflag: unit of row t = uscan[t] + uflag[t] - 1)
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_unit_deps(int nrow, int nunits, const int* __restrict__ rp,
                                                      const int* __restrict__ ci, const int* __restrict__ ustart,
                                                      const int* __restrict__ uflag, const int* __restrict__ uscan,
                                                      int* __restrict__ deps, int* __restrict__ overflow)
{
    __shared__ int s_dep[kBlock / 64][kUnitDeps];
    const int      wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if(lane < kUnitDeps)
        s_dep[wave][lane] = -1; // free slot
    __syncthreads();
    const int  me   = blockIdx.x * (kBlock / 64) + wave;
    const bool unit = me < nunits;
    const int  t0   = unit ? ustart[me] : 0;
    const int  cnt  = unit ? ustart[me + 1] - t0 : 0;
    bool       over = false;
    if(lane < cnt)
    {
        const int t    = t0 + lane;
        const int i    = LOWER ? t : nrow - 1 - t;
        int       last = -1;
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int c = ci[j];
            if(LOWER ? (c >= i) : (c <= i))
                continue;
            const int tc = LOWER ? c : nrow - 1 - c;
            if(tc >= t0)
                continue;
            const int u = uscan[tc] + uflag[tc] - 1;
            if(u == last)
                continue;
            last        = u;
            bool placed = false; // wait-free insertion (k_block_deps)
            for(int sl = 0; sl < kUnitDeps && !placed; ++sl)
            {
                int expect = -1;
                if(__hip_atomic_compare_exchange_strong(&s_dep[wave][sl], &expect, u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_WORKGROUP)
                   || expect == u)
                    placed = true;
            }
            over = over || !placed;
        }
    }
    const bool any_over = __ballot(over) != 0ull;
    __syncthreads();
    if(unit)
    {
        if(lane < kUnitDeps)
            deps[(int64_t)me * kUnitDeps + lane] = s_dep[wave][lane];
        if(lane == 0 && any_over)
            *overflow = 1;
    }
}

// level of a unit = 1 + max level of its dependencies (all of them earlier units); 0 = not computed yet.  One lane per unit,
// natural order.  Dependencies held by lower lanes of the same wave are noted in a lane mask; once every lane of the wave
// has its out-of-wave dependencies, the mask is resolved in registers in one ascending pass (k_levels) and the wave
// publishes.  As long as some lane still waits, the others do not wait with it (64 units that published together would tie
// the end of one grid plane to the start of the next, the very thing the units avoid): a lane whose dependencies are all
// there publishes, and the lanes that hold its bit take its level through a shuffle, one per turn.
// Units go to workgroups by TICKET (the k-th workgroup to start takes units [k kBlock, (k + 1) kBlock)): a unit only waits
// for earlier units, i.e. for workgroups that have started, whatever order the hardware dispatches the grid in.
__global__ __launch_bounds__(kBlock) void k_unit_levels(int nunits, const int* __restrict__ deps, int* level, unsigned* counter)
{
    const int64_t u    = (int64_t)take_ticket(counter, 0u) * kBlock + threadIdx.x;
    const bool    live = u < nunits;
    const int     lane = threadIdx.x & 63;
    const int     u0   = (int)u - lane;
    int           lev = 0, mine = 0;
    unsigned long long inwave = 0ull;
    if(live)
        for(int d = 0; d < kUnitDeps; ++d)
        {
            const int v = deps[u * kUnitDeps + d];
            if(v >= u0 && v < (int)u)
                inwave |= 1ull << (v - u0);
        }
    int  d       = 0;
    bool fin     = !live;
    int  backoff = 1;
    do
    {
        const int                d_start = d;
        const bool               was_fin = fin;
        const unsigned long long m_start = inwave;
        if(!fin)
            while(d < kUnitDeps)
            {
                const int v = deps[u * kUnitDeps + d];
                if(v < 0 || v >= u0) // (free slot, in-wave; v > u cannot happen for a triangular dependency)
                {
                    ++d;
                    continue;
                }
                const int lv = __hip_atomic_load(level + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if(lv == 0)
                    break;
                lev = max(lev, lv);
                ++d;
            }
        if(__ballot(!fin && d < kUnitDeps) == 0ull)
        {
            // nobody waits for another wave any more: the rest in registers
            for(int b = 0; b < 63; ++b)
            {
                const int src = fin ? mine : lev + 1;
                const int lb  = __builtin_amdgcn_readlane(src, b);
                if((inwave >> b) & 1ull)
                    lev = max(lev, lb);
            }
            if(!fin)
                __hip_atomic_store(level + u, lev + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if(!fin && d == kUnitDeps && inwave == 0ull)
        {
            mine = lev + 1;
            __hip_atomic_store(level + u, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fin = true;
        }
        const int want = inwave != 0ull ? (int)__ffsll((long long)inwave) - 1 : lane;
        const int got  = __shfl(mine, want, 64);
        if(!fin && want != lane && got != 0)
        {
            lev = max(lev, got);
            inwave &= ~(1ull << want);
        }
        backoff = poll_backoff(__ballot(!was_fin && (fin || d != d_start || inwave != m_start)) != 0ull, backoff);
    } while(__ballot(!fin) != 0ull);
}

// out->ustart [nunits + 1]: first sweep row of every unit (always, n >= 1); out->order [nunits]: units by level of the unit
// graph, or nullptr where the natural order is kept (everything resident at once anyway, more than kUnitDeps distinct units
// under one unit).  The caller releases the plan.
int unit_schedule(const ramd_mat_s* m, bool lower, UnitPlan* out)
{
    out->release();
    const int n = m->nrow;
    if(n < 1)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = ew_grid(n + 1);
    int *     brk = nullptr, *bscan = nullptr, *hard = nullptr, *hscan = nullptr, *seg = nullptr, *uflag = nullptr,
        *uscan = nullptr, *deps = nullptr, *over = nullptr, *level = nullptr, *tick = nullptr;
    auto drop = [&]() {
        dev_free(&tick);
        dev_free(&brk);
        dev_free(&bscan);
        dev_free(&hard);
        dev_free(&hscan);
        dev_free(&seg);
        dev_free(&uflag);
        dev_free(&uscan);
        dev_free(&deps);
        dev_free(&over);
        dev_free(&level);
    };
#define US_TRY(expr)     \
    do                   \
    {                    \
        int s_ = (expr); \
        if(s_ != RAMD_OK) \
        {                \
            drop();      \
            out->release(); \
            return s_;   \
        }                \
    } while(0)
#define US_HIP(expr) US_TRY((expr) == hipSuccess ? RAMD_OK : RAMD_ERR_HIP)
    US_TRY(dev_alloc(&brk, (int64_t)n + 1));
    US_TRY(dev_alloc(&bscan, (int64_t)n + 1));
    if(lower)
        hipLaunchKernelGGL((k_unit_breaks<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, brk);
    else
        hipLaunchKernelGGL((k_unit_breaks<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, brk);
    US_TRY(device_exclusive_scan(brk, bscan, (int64_t)n + 1));
    US_TRY(dev_alloc(&hard, (int64_t)n + 1));
    US_TRY(dev_alloc(&hscan, (int64_t)n + 1));
    hipLaunchKernelGGL(k_unit_hard_breaks, dim3(grid), dim3(kBlock), 0, b.cur, n, brk, bscan, hard);
    US_TRY(device_exclusive_scan(hard, hscan, (int64_t)n + 1));
    dev_free(&brk);
    dev_free(&bscan);
    int nseg = 0;
    US_HIP(hipMemcpyAsync(&nseg, hscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    US_HIP(hipStreamSynchronize(b.cur));
    US_TRY(dev_alloc(&seg, (int64_t)nseg + 1));
    hipLaunchKernelGGL(k_unit_scatter, dim3(grid), dim3(kBlock), 0, b.cur, n, hard, hscan, seg);
    US_TRY(dev_alloc(&uflag, (int64_t)n + 1));
    US_TRY(dev_alloc(&uscan, (int64_t)n + 1));
    hipLaunchKernelGGL(k_unit_starts, dim3(grid), dim3(kBlock), 0, b.cur, n, hard, hscan, seg, uflag);
    US_TRY(device_exclusive_scan(uflag, uscan, (int64_t)n + 1));
    dev_free(&hard);
    dev_free(&hscan);
    dev_free(&seg);
    int nunits = 0;
    US_HIP(hipMemcpyAsync(&nunits, uscan + n, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    US_HIP(hipStreamSynchronize(b.cur));
    out->nunits = nunits;
    US_TRY(dev_alloc(&out->ustart, (int64_t)nunits + 1));
    hipLaunchKernelGGL(k_unit_scatter, dim3(grid), dim3(kBlock), 0, b.cur, n, uflag, uscan, out->ustart);
    US_HIP(hipMemcpyAsync(out->ustart + nunits, &n, sizeof(int), hipMemcpyHostToDevice, b.cur));
    US_HIP(hipStreamSynchronize(b.cur)); // (&n is a stack address)
    static const int min_blocks = getenv("RAMD_BLOCKSCHED_MIN") ? atoi(getenv("RAMD_BLOCKSCHED_MIN")) : 4096;
    if(nunits < min_blocks * (kBlock / 64)) // everything is resident at once anyway (RAMD_BLOCKSCHED_MIN=1: tests)
    {
        drop();
        return RAMD_OK;
    }
    US_TRY(dev_alloc(&deps, (int64_t)nunits * kUnitDeps));
    US_TRY(dev_alloc(&over, 1));
    US_TRY(dev_alloc(&level, nunits));
    US_HIP(hipMemsetAsync(over, 0, sizeof(int), b.cur));
    US_HIP(hipMemsetAsync(level, 0, sizeof(int) * (size_t)nunits, b.cur));
    const unsigned nbu = (unsigned)((nunits + kBlock / 64 - 1) / (kBlock / 64));
    if(lower)
        hipLaunchKernelGGL((k_unit_deps<true>), dim3(nbu), dim3(kBlock), 0, b.cur, n, nunits, m->rp, m->ci, out->ustart, uflag,
                           uscan, deps, over);
    else
        hipLaunchKernelGGL((k_unit_deps<false>), dim3(nbu), dim3(kBlock), 0, b.cur, n, nunits, m->rp, m->ci, out->ustart,
                           uflag, uscan, deps, over);
    int hover = 0;
    US_HIP(hipMemcpyAsync(&hover, over, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    US_HIP(hipStreamSynchronize(b.cur));
    if(hover != 0)
    {
        drop();
        return RAMD_OK;
    }
    US_TRY(dev_alloc(&tick, 1));
    US_HIP(hipMemsetAsync(tick, 0, sizeof(int), b.cur));
    hipLaunchKernelGGL(k_unit_levels, dim3((unsigned)((nunits + kBlock - 1) / kBlock)), dim3(kBlock), 0, b.cur, nunits, deps,
                       level, (unsigned*)tick);
    int nlev = 0;
    US_TRY(device_max_int(level, nunits, &nlev)); // (synchronises)
    US_TRY(dev_alloc(&out->order, nunits));
    US_TRY(device_stable_sort_by_key(level, nunits, nlev, out->order));
    drop();
#undef US_TRY
#undef US_HIP
    return RAMD_OK;
}

} // namespace ramd
