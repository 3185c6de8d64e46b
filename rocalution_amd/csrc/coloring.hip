// coloring.hip -- greedy first-fit multi-colouring on the device, same colours as the serial sweep.
//
// Reference: HostMatrixCSR::MultiColoring (src/base/host/host_matrix_csr.cpp:2469-2599; the HIP
// backend copies the pattern to the host and runs the same loop, hip_matrix_csr.cpp:3915-4060):
//   rows in natural order; colour(i) = smallest colour >= 1 not carried by an already coloured
//   neighbour; neighbours = entries of row i AND of column i; perm[i] = offset[colour(i)]++.
// colour(i) depends only on the neighbours j < i, so the sweep is a DAG like a triangular solve.  For a
// structurally symmetric pattern (checked here) the row entries are all the neighbours and the sweep
// runs sync-free: thread per row, workgroups in ticket order, a row polls the colour array itself
// (0 = not yet) for neighbours in other waves and takes the colours of its own wave in registers
// (the x-1 chain of a stencil never leaves them; see k_greedy_color).  Unsymmetric patterns, or more than 64
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

// The sweep takes its rows in the wave units of blocksched.hip (unit_schedule: <= 64 consecutive rows that belong
// together, units in hyperplane order): a neighbour is a row of an EARLIER wave -- its colour is polled in memory -- or of
// a lower lane of this wave, noted in a lane mask and resolved in registers in one ascending pass (at step b lane b has
// seen all its neighbours, picks its colour, v_readlane broadcasts it to the lanes that hold bit b).  The x-1 chain of a
// grid line costs a handful of ALU instructions per link instead of a turn of a divergent loop with a memory poll in it:
// 186 -> 13 ms at 512^3 (the 256-row blocks of round 2 were half an x-pencil each, 256 serial links).
constexpr int kColorBlock = 1024;
__global__ __launch_bounds__(kColorBlock) void k_greedy_color(int nrow, const int* __restrict__ rp,
                                                              const int* __restrict__ ci, int* color, unsigned* counter,
                                                              int* overflow, UnitView uv)
{
    const unsigned slot = take_ticket(counter, 0u) * (kColorBlock / 64) + (threadIdx.x >> 6); // (one ticket per workgroup)
    const int      lane = threadIdx.x & 63;
    const bool     have = slot < (unsigned)uv.nunits;
    const int      unit = have ? (uv.order ? uv.order[slot] : (int)slot) : 0;
    const int      w0   = have ? uv.ustart[unit] : 0; // first row of my wave
    const int64_t  t    = (int64_t)w0 + lane;
    const bool     live = have && t < uv.ustart[unit + 1];
    const int      row  = live ? (int)t : 0;
    int            j    = live ? rp[row] : 0;
    const int      end  = live ? rp[row + 1] : 0;
    unsigned long long used = 0ull; // bit c-1 <-> colour c
    unsigned long long inwave = 0ull; // neighbours held by lower lanes
    bool fin     = !live;
    int  spins   = 0;
    int  backoff = 1;
    bool stalled = false;
    // (A) neighbours coloured by other waves; wave-uniform loop (SIMT rule of trisolve.hip); a stalled wave polls with one lane
    do
    {
        spin_guard(spins);
        const int  j_before   = j;
        const bool fin_before = fin;
        const bool my_turn    = !stalled || lane == (int)__ffsll((long long)__ballot(!fin)) - 1;
        if(!fin && my_turn)
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
                    inwave |= 1ull << (c - w0);
                    ++j;
                    continue;
                }
                const int cc = __hip_atomic_load(color + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if(cc == 0)
                    break;
                used |= 1ull << (cc - 1);
                ++j;
            }
            fin = (j >= end);
        }
        const bool advanced = __ballot(!fin_before && (fin || j != j_before)) != 0ull;
        stalled             = !advanced;
        backoff             = poll_backoff(advanced, backoff);
    } while(__ballot(!fin) != 0ull);
    // (B) neighbours held by lower lanes
    if(__ballot(inwave != 0ull) != 0ull)
        for(int b = 0; b < 63; ++b)
        {
            const int pick = used == ~0ull ? 64 : __ffsll((long long)~used); // what lane b takes (it is final at step b)
            const int cb   = __builtin_amdgcn_readlane(pick, b);
            const unsigned half = b < 32 ? (unsigned)inwave : (unsigned)(inwave >> 32);
            if((half >> (b & 31)) & 1u)
                used |= 1ull << (cb - 1);
        }
    if(live)
    {
        int mine;
        if(used == ~0ull)
        {
            *overflow = 1; // > 64 colours: result discarded by the caller
            mine      = 64;
        }
        else
            mine = __ffsll((long long)~used);
        __hip_atomic_store(color + row, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
    build_mark(nullptr);
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
    build_mark("colouring: symmetry check");
    UnitPlan units; // wave units of the lower-neighbour graph, in hyperplane order (blocksched.hip)
    s = unit_schedule(m, true, &units);
    build_mark("colouring: unit schedule");
    if(s != RAMD_OK)
        return cleanup(s);
    const int per = kColorBlock / 64;
    hipLaunchKernelGGL(k_greedy_color, dim3((unsigned)((units.nunits + per - 1) / per)), dim3(kColorBlock), 0, b.cur, n, m->rp,
                       m->ci, color, reinterpret_cast<unsigned*>(work + 2), work + 1, unit_view(units));
    if(hipGetLastError() != hipSuccess)
    {
        units.release();
        return cleanup(RAMD_ERR_HIP);
    }
    int nc = 0;
    s      = device_max_int(color, n, &nc); // (synchronises)
    build_mark("colouring: greedy sweep");
    units.release();
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
    build_mark("colouring: permutation");
    if(s == RAMD_OK)
        *num_colors = nc;
    return cleanup(s);
}

} // namespace ramd
