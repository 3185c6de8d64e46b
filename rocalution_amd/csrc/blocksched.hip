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

} // namespace ramd
