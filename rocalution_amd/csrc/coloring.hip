// coloring.hip -- greedy first-fit multi-colouring on the device, same colours as the serial sweep.
//
// Reference: HostMatrixCSR::MultiColoring (src/base/host/host_matrix_csr.cpp:2469-2599; the HIP
// backend copies the pattern to the host and runs the same loop, hip_matrix_csr.cpp:3915-4060):
//   rows in natural order; colour(i) = smallest colour >= 1 not carried by an already coloured
//   neighbour; neighbours = entries of row i AND of column i; perm[i] = offset[colour(i)]++.
// colour(i) depends only on the neighbours j < i, so the sweep is a DAG like a triangular solve.  For a
// structurally symmetric pattern (checked here) the row entries are all the neighbours and the sweep
// runs sync-free: thread per row, workgroups in ticket order, a row polls the colour array itself
// (0 = not yet) for neighbours in other waves and takes the colours of its own wave through shuffles
// (the x-1 chain of a stencil never leaves the registers).  Unsymmetric patterns, or more than 64
// colours, return RAMD_ERR_UNSUPPORTED and the caller runs the host sweep (host_analysis.hip).
// At 512^3 the serial host sweep takes 6.3 s (+ 4.3 GB over PCIe); this one is bandwidth/latency bound.
#include "device_utils.hpp"
#include "matrix_impl.hpp"

namespace ramd
{

// every off-diagonal (i,j) must have its mirror (j,i)
__global__ __launch_bounds__(kBlock) void k_pattern_symmetric(int nrow, const int* __restrict__ rp,
                                                              const int* __restrict__ ci,
                                                              int* __restrict__ bad)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
    {
        for(int j = rp[i]; j < rp[i + 1]; ++j)
        {
            const int c = ci[j];
            if(c == (int)i)
                continue;
            bool found = false;
            for(int k = rp[c]; k < rp[c + 1]; ++k)
                if(ci[k] == (int)i)
                {
                    found = true;
                    break;
                }
            if(!found)
            {
                *bad = 1;
                return;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_greedy_color(int nrow, const int* __restrict__ rp,
                                                         const int* __restrict__ ci, int* color,
                                                         unsigned* counter, int* overflow,
                                                         const int* __restrict__ block_order)
{
    const unsigned tick = take_ticket(counter, 0u);
    const unsigned blk  = block_order ? (unsigned)block_order[tick] : tick; // blocksched.hip
    const int64_t  t    = (int64_t)blk * kBlock + threadIdx.x;
    const int      lane = threadIdx.x & 63;
    const bool     live = t < nrow;
    const int      row  = live ? (int)t : 0;
    const int      w0   = (int)(t - lane); // first row of my wave
    int            j    = live ? rp[row] : 0;
    const int      end  = live ? rp[row + 1] : 0;
    unsigned long long used = 0ull; // bit c-1 <-> colour c
    int  mine  = 0;
    bool fin   = !live;
    int  spins = 0;
    int  backoff = 1;
    // wave-uniform loop (SIMT rule of trisolve.hip): publish inside, leave together
    do
    {
        spin_guard(spins);
        const int  j_before   = j;
        const bool fin_before = fin;
        // (1) neighbours coloured by other waves: consume every one that is ready
        int want = lane; // (2) at most one neighbour inside my wave per turn, through a shuffle
        if(!fin)
        {
            while(j < end)
            {
                const int c = ci[j];
                if(c >= row) // diagonal / not coloured yet when my turn comes
                {
                    ++j;
                    continue;
                }
                if(c >= w0)
                {
                    want = c - w0;
                    break;
                }
                const int cc = __hip_atomic_load(color + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if(cc == 0)
                    break;
                used |= 1ull << (cc - 1);
                ++j;
            }
        }
        const int got = __shfl(mine, want, 64);
        if(!fin)
        {
            if(want != lane && got != 0)
            {
                used |= 1ull << (got - 1);
                ++j;
            }
            if(j >= end)
            {
                if(used == ~0ull)
                {
                    *overflow = 1; // > 64 colours: result discarded by the caller
                    mine      = 64;
                }
                else
                    mine = __ffsll((long long)~used);
                __hip_atomic_store(color + row, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fin = true;
            }
        }
        // nobody in the wave advanced: exponential back-off (device_utils.hpp) instead of polling at full rate
        backoff = poll_backoff(__ballot(!fin_before && (fin || j != j_before)) != 0ull, backoff);
    } while(__ballot(!fin) != 0ull);
}

__global__ __launch_bounds__(kBlock) void k_color_flag(int nrow, const int* __restrict__ color, int c,
                                                       int* __restrict__ flag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrow; i += gsz)
        flag[i] = (i < nrow && color[i] == c) ? 1 : 0;
}
__global__ __launch_bounds__(kBlock) void k_color_perm(int nrow, const int* __restrict__ color, int c,
                                                       const int* __restrict__ pos, int offset,
                                                       int* __restrict__ perm)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gsz)
        if(color[i] == c)
            perm[i] = offset + pos[i];
}

// -> RAMD_OK, or RAMD_ERR_UNSUPPORTED when the device sweep does not apply (caller: host sweep)
int multicoloring_device(const ramd_mat_s* m, int* num_colors, int* size_colors, ramd_vec_s* perm)
{
    Backend&  b = backend();
    const int n = m->nrow;
    if(n <= 0 || m->nnz <= 0)
        return RAMD_ERR_UNSUPPORTED;
    if(m->nnz / n > 256) // the symmetry check walks the mirror row per entry
        return RAMD_ERR_UNSUPPORTED;
    int* color = nullptr;
    int* work  = nullptr; // [0] bad pattern, [1] overflow, [2] ticket
    int* pos   = nullptr;
    int  s     = dev_alloc(&work, 4);
    if(s == RAMD_OK)
        s = dev_alloc(&color, (int64_t)n + 1);
    auto cleanup = [&](int code) {
        dev_free(&color);
        dev_free(&work);
        dev_free(&pos);
        return code;
    };
    if(s != RAMD_OK)
        return cleanup(s);
    int h[4] = {0, 0, 0, 0};
    if(hipMemsetAsync(work, 0, sizeof(int) * 4, b.cur) != hipSuccess)
        return cleanup(RAMD_ERR_HIP);
    hipLaunchKernelGGL(k_pattern_symmetric, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, work);
    if(hipMemcpyAsync(h, work, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
       || hipStreamSynchronize(b.cur) != hipSuccess)
        return cleanup(RAMD_ERR_HIP);
    if(h[0] != 0)
        return cleanup(RAMD_ERR_UNSUPPORTED);
    if(hipMemsetAsync(color, 0, sizeof(int) * ((size_t)n + 1), b.cur) != hipSuccess)
        return cleanup(RAMD_ERR_HIP);
    const int nblk = (n + kBlock - 1) / kBlock;
    int* border = nullptr;
    (void)block_schedule(m, true, &border); // hyperplane order of the row blocks (nullptr: natural order)
    hipLaunchKernelGGL(k_greedy_color, dim3(nblk), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, color,
                       reinterpret_cast<unsigned*>(work + 2), work + 1, border);
    if(hipGetLastError() != hipSuccess)
    {
        dev_free(&border);
        return cleanup(RAMD_ERR_HIP);
    }
    int nc = 0;
    s      = device_max_int(color, n, &nc); // (synchronises)
    dev_free(&border);
    if(s != RAMD_OK)
        return cleanup(s);
    if(hipMemcpyAsync(h, work, sizeof(int) * 2, hipMemcpyDeviceToHost, b.cur) != hipSuccess
       || hipStreamSynchronize(b.cur) != hipSuccess)
        return cleanup(RAMD_ERR_HIP);
    if(h[1] != 0 || nc < 1 || nc > 64)
        return cleanup(RAMD_ERR_UNSUPPORTED);
    // perm[i] = offset[colour] + rank of i among the rows of its colour (stable)
    s = ramd_vec_allocate(perm, n);
    if(s == RAMD_OK)
        s = dev_alloc(&pos, (int64_t)n + 1);
    int offset = 0;
    for(int c = 1; c <= nc && s == RAMD_OK; ++c)
    {
        hipLaunchKernelGGL(k_color_flag, dim3(ew_grid((int64_t)n + 1)), dim3(kBlock), 0, b.cur, n, color, c, pos);
        s = device_exclusive_scan(pos, pos, (int64_t)n + 1);
        if(s != RAMD_OK)
            break;
        int cnt = 0;
        if(hipMemcpyAsync(&cnt, pos + n, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess
           || hipStreamSynchronize(b.cur) != hipSuccess)
        {
            s = RAMD_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(k_color_perm, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, color, c, pos, offset,
                           (int*)perm->d);
        size_colors[c - 1] = cnt;
        offset += cnt;
    }
    if(s == RAMD_OK && hipStreamSynchronize(b.cur) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK)
        *num_colors = nc;
    return cleanup(s);
}

} // namespace ramd
