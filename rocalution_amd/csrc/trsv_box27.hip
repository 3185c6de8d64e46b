// trsv_box27.hip -- sparse triangular solve on the 27-point stencil: pencils marched along x, one wave per pencil (round 6).
//
// The reference's own 3-D test operator is the 27-point Laplacian (clients/include/utility.hpp:110-177), and ILU(0) keeps its
// pattern: the lower / upper triangle of such a matrix has 13 entries per row.  The record-form box tiles of trisolve.hip take
// 6.2 ms per triangle of it at 256^3 (0.06 of the roofline: eight lanes per row, 74 steps per 512-row tile), the vendor's
// csrsv 1.4 s per iteration.  Where every row has exactly the entries its lattice position allows, nothing has to be decoded:
//   * a PENCIL is an 8 x 8 (y, z) cross-section marched along x; lane (j, k) of the wave takes, at step t, the row
//     x = t - 2 j - 4 k of its grid line.  With that skew all 13 dependencies of a row were computed at earlier steps: (x-1, j, k)
//     one step ago by the lane itself, (x-1 .. x+1, j-1, k) three to one steps ago, the nine of the plane below seven to one
//     steps ago -- the dependency levels of this triangle ARE the planes x + 2 y + 4 z = const;
//   * every lane keeps the last 16 values of its line in an LDS ring (element x in column x mod 16); a step is 12 LDS reads,
//     13 multiplies and subtractions in the order of the host loop (+ the division), one LDS write, one store of the result;
//   * the coefficients are packed once per analysis in exactly the order a wave consumes them (pencil, step, dependency, lane:
//     8 bytes per lane and load, fully coalesced, no column indices -- 13 x 8 instead of 13 x 12 bytes per row) and run four
//     steps ahead of their use in a register queue;
//   * the lines next to a pencil (18 of them: one in y, ten below in z incl. the corners) belong to the pencils (J-1, K),
//     (J-1 .. J+1, K-1); they are read from the OUTPUT vector itself, which the solve fills with a NaN sentinel first and into
//     which every row of an outflow face is published with one agent-scope store (data-tagged values, as everywhere in
//     trisolve.hip): lanes 0-17 keep their halo line's ring filled a block of four steps ahead, polling only what is missing;
//   * pencils are taken by ticket in order of J + 2 K: a pencil only waits for pencils with lower tickets, i.e. for waves
//     that are running.
// The upper solve is the same sweep on the mirrored lattice (x, y, z counted from their far ends): its dependency list is the
// lower one reversed.  Arithmetic per row: the subtractions in ascending column order, unfused multiply and subtract, then the
// division by the stored diagonal -- src/base/host/host_matrix_csr.cpp:1163-1221 (LUSolve), :1357-1404 (LSolve), :1420-1466
// (USolve).  A neighbour the lattice does not have is no entry of the row: its coefficient slot holds +0 and the value read for
// it is +0 (rings start at zero, a stale column is masked), so the term is (+0)(+0) = +0 and s - (+0) is s bit for bit.
#include "trsv_box27.hpp"

#include "device_utils.hpp"
#include "matrix_impl.hpp"
#include "trsv_handoff.hpp"

#include <algorithm>
#include <vector>

namespace ramd
{

namespace
{

constexpr int kBJ = 8, kBK = 8; // lanes of a pencil's cross-section in y and z
constexpr int kSkJ = 2, kSkK = 4; // steps a line starts after its neighbour below in y / z
constexpr int kSkewMax = kSkJ * (kBJ - 1) + kSkK * (kBK - 1); // 42
constexpr int kRing = 16, kRP = kRing + 1; // columns of a line's ring, elements per line (padding: bank spread)
constexpr int kLW = kBJ + 2; // lines per z-row in LDS: j = -1 .. 8
constexpr int kLines = kLW * (kBK + 1); // k = -1 .. 7
constexpr int kNDep = 13;
constexpr int kPF = 4; // steps per block = steps the coefficient queue runs ahead
constexpr int kNHalo = 18;

struct BoxDims
{
    int nx, ny, nz, ntj, ntk, ntiles, T;
};

__host__ __device__ constexpr int box_line(int j, int k) // j in [-1, 8], k in [-1, 7]
{
    return ((k + 1) * kLW + (j + 1)) * kRP;
}
// dependency i (0 .. 12) of the LOWER triangle in ascending column order: the first 13 of the 27 offsets (dk, dj, dx) in
// lexicographic order
__host__ __device__ constexpr int box_dk(int i)
{
    return i / 9 - 1;
}
__host__ __device__ constexpr int box_dj(int i)
{
    return (i / 3) % 3 - 1;
}
__host__ __device__ constexpr int box_dx(int i)
{
    return i % 3 - 1;
}

// ---------------------------------------------------------------- detection
// smallest offset |col - row| > thr over the triangle
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_box_min_offset(int n, const int* __restrict__ rp, const int* __restrict__ ci, int thr,
                                                           int* __restrict__ out_min)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    int           mn  = 0x7fffffff;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gsz)
        for(int a = rp[r]; a < rp[r + 1]; ++a)
        {
            const int d = LOWER ? (int)r - ci[a] : ci[a] - (int)r;
            if(d > thr)
                mn = min(mn, d);
        }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
        mn = min(mn, __shfl_xor(mn, off, 64));
    if((threadIdx.x & 63) == 0 && mn != 0x7fffffff)
        atomicMin(out_min, mn);
}

// every row holds, in its triangle, exactly the neighbours of the 3 x 3 x 3 box its lattice position allows, in ascending
// columns (+ the diagonal where it is needed); what the other triangle holds plays no role (the LU factors keep both)
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_box_verify(int n, const int* __restrict__ rp, const int* __restrict__ ci, int nx, int ny,
                                                       int nz, int need_diag, int* __restrict__ flag)
{
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    const int     nxny = nx * ny;
    bool          bad  = false;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gsz)
    {
        const int x = (int)(r % nx), y = (int)((r / nx) % ny), z = (int)(r / nxny);
        int       i = LOWER ? 0 : 14; // next of the 27 offsets (lexicographic in (dk, dj, dx) = ascending columns) to be matched
        const int iend = LOWER ? 13 : 27;
        bool      diag = false;
        for(int a = rp[r]; a < rp[r + 1]; ++a)
        {
            const int c = ci[a];
            if(c == (int)r)
            {
                diag = true;
                continue;
            }
            if(LOWER ? c > (int)r : c < (int)r)
                continue;
            // the next neighbour the lattice has
            bool found = false;
            for(; i < iend && !found; ++i)
            {
                const int dk = i / 9 - 1, dj = (i / 3) % 3 - 1, dx = i % 3 - 1;
                if(x + dx >= 0 && x + dx < nx && y + dj >= 0 && y + dj < ny && z + dk >= 0 && z + dk < nz)
                {
                    found = true;
                    bad   = bad || c != (int)r + dk * nxny + dj * nx + dx;
                }
            }
            bad = bad || !found;
        }
        for(; i < iend; ++i) // a neighbour the lattice has and the row does not
        {
            const int dk = i / 9 - 1, dj = (i / 3) % 3 - 1, dx = i % 3 - 1;
            bad = bad || (x + dx >= 0 && x + dx < nx && y + dj >= 0 && y + dj < ny && z + dk >= 0 && z + dk < nz);
        }
        bad = bad || (need_diag && !diag);
    }
    if(bad)
        *flag = 1;
}

// the coefficients of row r into the slot (pencil rank, step, dependency, lane) a wave reads them from; NCO = 13 (+ 1: diagonal)
template <typename T, bool LOWER>
__global__ __launch_bounds__(kBlock) void k_box_fill(int n, BoxDims g, int nco, const int* __restrict__ rp, const int* __restrict__ ci,
                                                     const T* __restrict__ val, const int* __restrict__ trank, T* __restrict__ coef)
{
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    const int     nxny = g.nx * g.ny;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gsz)
    {
        const int x0 = (int)(r % g.nx), y0 = (int)((r / g.nx) % g.ny), z0 = (int)(r / nxny);
        // the sweep's coordinates: the lattice itself (lower) or its mirror image (upper)
        const int x = LOWER ? x0 : g.nx - 1 - x0, y = LOWER ? y0 : g.ny - 1 - y0, z = LOWER ? z0 : g.nz - 1 - z0;
        const int J = y / kBJ, K = z / kBK, j = y % kBJ, k = z % kBK;
        const int lane = j + kBJ * k, t = x + kSkJ * j + kSkK * k;
        T*        dst = coef + (((int64_t)trank[J + g.ntj * K] * g.T + t) * nco) * 64 + lane;
        for(int a = rp[r]; a < rp[r + 1]; ++a)
        {
            const int c = ci[a];
            if(c == (int)r)
            {
                if(nco > kNDep)
                    dst[(int64_t)kNDep * 64] = val[a];
                continue;
            }
            if(LOWER ? c > (int)r : c < (int)r)
                continue;
            // offset -> index in the lexicographic order of (dk, dj, dx); the verified pattern guarantees a box neighbour
            const int d  = c - (int)r;
            const int dk = d < -(nxny / 2) ? -1 : (d > nxny / 2 ? 1 : 0);
            const int d2 = d - dk * nxny;
            const int dj = d2 < -(g.nx / 2) ? -1 : (d2 > g.nx / 2 ? 1 : 0);
            const int dx = d2 - dj * g.nx;
            const int i  = (dk + 1) * 9 + (dj + 1) * 3 + (dx + 1);
            // slot in the order of the host loop: lower: entry i (0 .. 12); upper: entry i - 14
            dst[(int64_t)(LOWER ? i : i - 14) * 64] = val[a];
        }
    }
}

// ---------------------------------------------------------------- the solve
template <typename T, bool LOWER, bool UNIT>
__global__ __launch_bounds__(64) void k_trsv_box(BoxDims g, const int* __restrict__ ptab, const T* __restrict__ coef, const T* __restrict__ in,
                                                 T* out, unsigned* counter)
{
    using B              = typename Sentinel<T>::bits;
    constexpr int NCO    = kNDep + (UNIT ? 0 : 1);
    extern __shared__ __attribute__((aligned(16))) char box_lds[];
    T*        ring = reinterpret_cast<T*>(box_lds);
    const int lane = threadIdx.x;
    const int cj = lane & 7, ck = lane >> 3;
    const int s_own = kSkJ * cj + kSkK * ck;
    const int a_own = box_line(cj, ck);
    // the lines of the 13 dependencies, in the order of the host loop (the upper solve runs on the mirrored lattice: the lower
    // list reversed)
    int a_dep[kNDep], dxs[kNDep];
#pragma unroll
    for(int d = 0; d < kNDep; ++d)
    {
        const int i = LOWER ? d : kNDep - 1 - d;
        a_dep[d]    = box_line(cj + box_dj(i), ck + box_dk(i));
        dxs[d]      = box_dx(i);
    }
    // the halo line of this lane (lanes 0 .. 17): h < 10: (j = h - 1, k = -1), else (j = -1, k = h - 10)
    const int hj = lane < 10 ? lane - 1 : -1, hk = lane < 10 ? -1 : lane - 10;
    const int hs = kSkJ * hj + kSkK * hk;
    const int a_halo = lane < kNHalo ? box_line(hj, hk) : 0;
    const int64_t nxny = (int64_t)g.nx * g.ny;
    for(;;)
    {
        unsigned tk = 0;
        if(lane == 0)
            tk = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int q = __builtin_amdgcn_readfirstlane((int)tk);
        if(q >= g.ntiles)
            break;
        const int pt = __builtin_amdgcn_readfirstlane(ptab[q]);
        const int J = pt & 0xffff, K = pt >> 16;
        const int yy = kBJ * J + cj, zz = kBK * K + ck;
        const bool valid = yy < g.ny && zz < g.nz;
        // element x of the lane's line in the natural-order vectors: base + x (lower) / base - x (upper, mirrored)
        const int64_t gb = LOWER ? ((int64_t)zz * g.ny + yy) * g.nx
                                 : ((int64_t)(g.nz - 1 - zz) * g.ny + (g.ny - 1 - yy)) * g.nx + (g.nx - 1);
        const int hy = kBJ * J + hj, hz = kBK * K + hk;
        const bool hvalid = lane < kNHalo && hy >= 0 && hy < g.ny && hz >= 0 && hz < g.nz;
        const int64_t hgb = !hvalid ? 0
                            : LOWER ? ((int64_t)hz * g.ny + hy) * g.nx
                                    : ((int64_t)(g.nz - 1 - hz) * g.ny + (g.ny - 1 - hy)) * g.nx + (g.nx - 1);
        auto gidx = [&](int64_t base, int x) -> int64_t { return LOWER ? base + x : base - x; };
        // rings start at +0
        for(int i = lane; i < kLines * kRP; i += 64)
            ring[i] = (T)0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // an outflow line: read by the pencils (J + 1, K), (J - 1 .. J + 1, K + 1)
        const bool face = cj == kBJ - 1 || ck == kBK - 1;
        const T*   cb   = coef + ((int64_t)q * g.T) * NCO * 64 + lane;
        // queue: coefficients and right-hand side of the next four steps
        T cq[kPF][NCO], rq[kPF];
#pragma unroll
        for(int i = 0; i < kPF; ++i)
        {
#pragma unroll
            for(int d = 0; d < NCO; ++d)
                cq[i][d] = nt_load(cb + ((int64_t)i * NCO + d) * 64);
            const int x = i - s_own;
            rq[i]       = in[gidx(gb, valid ? min(max(x, 0), g.nx - 1) : 0) * (valid ? 1 : 0)];
        }
        int xf = hvalid ? -1 : 0x3fffffff; // the halo line's ring holds every element up to xf
        B   hv[kPF];
        int hx = 0, hn = 0;
#pragma unroll
        for(int e = 0; e < kPF; ++e)
            hv[e] = Sentinel<T>::value;
        for(int t0 = 0; t0 < g.T; t0 += kPF)
        {
            // ---- halo: what the last block's requests brought, then whatever this block still needs, then the next requests
            const int need = min(g.nx - 1, t0 + kPF - 1 - hs - 1);
#pragma unroll
            for(int e = 0; e < kPF; ++e)
                if(e < hn && hx + e == xf + 1 && hv[e] != Sentinel<T>::value)
                {
                    ring[a_halo + ((xf + 1) & (kRing - 1))] = Sentinel<T>::from_bits(hv[e]);
                    ++xf;
                }
            int spins = 0;
            while(__ballot(xf < need) != 0ull)
            {
                spin_guard(spins);
                bool got = false;
                if(xf < need)
                {
                    const B v = poll_load(out + gidx(hgb, xf + 1));
                    if(v != Sentinel<T>::value)
                    {
                        ring[a_halo + ((xf + 1) & (kRing - 1))] = Sentinel<T>::from_bits(v);
                        ++xf;
                        got = true;
                    }
                }
                if(__ballot(got) == 0ull)
                    __builtin_amdgcn_s_sleep(1);
            }
            hx = xf + 1;
            hn = hvalid ? min(kPF, g.nx - hx) : 0;
#pragma unroll
            for(int e = 0; e < kPF; ++e)
                hv[e] = poll_load(out + gidx(hgb, hvalid ? min(hx + e, g.nx - 1) : 0) * (hvalid ? 1 : 0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // ---- four steps
#pragma unroll
            for(int i = 0; i < kPF; ++i)
            {
                const int  t = t0 + i, x = t - s_own;
                const bool act = valid && x >= 0 && x < g.nx;
                T          v[kNDep];
#pragma unroll
                for(int d = 0; d < kNDep; ++d)
                    v[d] = ring[a_dep[d] + ((x + dxs[d]) & (kRing - 1))];
                T sum = rq[i];
#pragma unroll
                for(int d = 0; d < kNDep; ++d)
                {
                    // (column x + 1 of a line still holds element x - 15 when x + 1 is behind the line's end)
                    const T vv = (dxs[d] > 0 && x + 1 >= g.nx) ? (T)0 : v[d];
                    sum -= cq[i][d] * vv;
                }
                if(!UNIT)
                    sum = sum / cq[i][NCO - 1];
                if(act)
                {
                    ring[a_own + (x & (kRing - 1))] = sum;
                    if(face)
                        publish(out + gidx(gb, x), sum);
                    else
                        nt_store(sum, out + gidx(gb, x));
                }
                // refill the queue slot for step t + 4
                const int tn = min(t + kPF, g.T - 1);
#pragma unroll
                for(int d = 0; d < NCO; ++d)
                    cq[i][d] = nt_load(cb + ((int64_t)tn * NCO + d) * 64);
                const int xn = x + kPF;
                rq[i]        = in[gidx(gb, valid ? min(max(xn, 0), g.nx - 1) : 0) * (valid ? 1 : 0)];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // (requests of the last block that nobody looked at: their registers stay reserved until here)
#pragma unroll
        for(int e = 0; e < kPF; ++e)
            asm volatile("" ::"v"(hv[e]));
    }
}

} // namespace

struct BoxPlan
{
    BoxDims  g{};
    bool     lower = true, unit = true;
    int      dtype = RAMD_F64;
    int      n     = 0;
    int*     ptab  = nullptr; // [ntiles] J | K << 16, in ticket order (J + 2 K ascending)
    void*    coef  = nullptr; // [ntiles][T][NCO][64]
    void*    scratch = nullptr; // [n]
    unsigned* counter = nullptr;
    size_t   coef_bytes = 0;
};

void box_release(BoxPlan** pp)
{
    BoxPlan* P = *pp;
    if(!P)
        return;
    dev_free(&P->ptab);
    dev_free(&P->counter);
    if(P->coef)
        (void)cached_free(P->coef);
    if(P->scratch)
        (void)cached_free(P->scratch);
    delete P;
    *pp = nullptr;
}
bool box_is_unit(const BoxPlan* P)
{
    return P->unit;
}
void* box_scratch(BoxPlan* P)
{
    return P->scratch;
}
void box_info(const BoxPlan* P, BoxInfo* info)
{
    *info = BoxInfo{P->g.nx, P->g.ny, P->g.nz, P->g.ntiles, P->g.T, P->coef_bytes};
}

static int box_mode()
{
    // 0: off; 1: lattices of 4096 rows and more (default); 2: every recognised lattice (tests)
    // (read at every analysis: the tests switch it inside one process)
    return getenv("RAMD_TRSV_BOX") ? atoi(getenv("RAMD_TRSV_BOX")) : 1;
}

template <typename T>
int box_build(const ramd_mat_s* m, bool lower, bool unit, BoxPlan** out)
{
    *out = nullptr;
    const int mode = box_mode();
    const int n    = m->nrow;
    if(mode == 0 || m->format != RAMD_CSR || n != m->ncol || n < 64)
        return RAMD_ERR_UNSUPPORTED;
    if(mode == 1 && n < 4096)
        return RAMD_ERR_UNSUPPORTED;
    // (13 entries per triangular row on average, or nearly: a cheap test before the passes over the matrix)
    if((int64_t)m->nnz < (int64_t)16 * n || (int64_t)m->nnz > (int64_t)27 * n)
        return RAMD_ERR_UNSUPPORTED;
    Backend& b = backend();
    int*     d = nullptr; // [0] min offset, [1] flag
    RAMD_TRY(dev_alloc(&d, 4));
    auto fail = [&](int code) {
        dev_free(&d);
        return code;
    };
    int       h[2];
    const int grid = ew_grid(n);
    auto      pass = [&](int thr) -> int {
        h[0] = 0x7fffffff;
        h[1] = 0;
        if(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, b.cur) != hipSuccess)
            return RAMD_ERR_HIP;
        if(lower)
            hipLaunchKernelGGL((k_box_min_offset<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, thr, d);
        else
            hipLaunchKernelGGL((k_box_min_offset<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, thr, d);
        if(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, b.cur) != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess)
            return RAMD_ERR_HIP;
        return RAMD_OK;
    };
    // offsets of the triangle: 1, nx - 1, nx, nx + 1, nx ny - nx - 1, ...
    int s = pass(1);
    if(s != RAMD_OK)
        return fail(s);
    if(h[0] == 0x7fffffff || h[0] < 3)
        return fail(RAMD_ERR_UNSUPPORTED);
    const int nx = h[0] + 1;
    s            = pass(nx + 1);
    if(s != RAMD_OK)
        return fail(s);
    if(h[0] == 0x7fffffff)
        return fail(RAMD_ERR_UNSUPPORTED);
    const int64_t nxny = (int64_t)h[0] + nx + 1;
    if(nx < 4 || nxny % nx != 0 || n % nxny != 0 || nxny / nx < 3 || n / nxny < 2)
        return fail(RAMD_ERR_UNSUPPORTED);
    const int ny = (int)(nxny / nx), nz = (int)(n / nxny);
    h[1]         = 0;
    if(hipMemcpyAsync(d + 1, h + 1, sizeof(int), hipMemcpyHostToDevice, b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    if(lower)
        hipLaunchKernelGGL((k_box_verify<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, nx, ny, nz, unit ? 0 : 1, d + 1);
    else
        hipLaunchKernelGGL((k_box_verify<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, nx, ny, nz, unit ? 0 : 1, d + 1);
    if(hipMemcpyAsync(h + 1, d + 1, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    if(h[1] != 0)
        return fail(RAMD_ERR_UNSUPPORTED);
    dev_free(&d);
    BoxPlan* P = new BoxPlan;
    P->lower = lower, P->unit = unit, P->dtype = m->dtype, P->n = n;
    BoxDims& g = P->g;
    g.nx = nx, g.ny = ny, g.nz = nz;
    g.ntj = (ny + kBJ - 1) / kBJ, g.ntk = (nz + kBK - 1) / kBK, g.ntiles = g.ntj * g.ntk;
    g.T = ((nx + kSkewMax + kPF - 1) / kPF) * kPF;
    auto bail = [&](int code) {
        box_release(&P);
        return code;
    };
    if(g.ntj > 0xffff || g.ntk > 0x7fff)
        return bail(RAMD_ERR_UNSUPPORTED);
    // tickets in order of J + 2 K
    std::vector<int> order((size_t)g.ntiles), rank((size_t)g.ntiles);
    for(int i = 0; i < g.ntiles; ++i)
        order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
        return (a % g.ntj) + 2 * (a / g.ntj) < (c % g.ntj) + 2 * (c / g.ntj);
    });
    std::vector<int> ptab((size_t)g.ntiles);
    for(int q = 0; q < g.ntiles; ++q)
    {
        const int tile    = order[(size_t)q];
        ptab[(size_t)q]   = (tile % g.ntj) | ((tile / g.ntj) << 16);
        rank[(size_t)tile] = q;
    }
    int* trank = nullptr;
    s          = dev_alloc(&P->ptab, g.ntiles);
    if(s == RAMD_OK)
        s = dev_alloc(&trank, g.ntiles);
    if(s == RAMD_OK)
        s = dev_alloc(&P->counter, 4);
    const int nco = kNDep + (unit ? 0 : 1);
    P->coef_bytes = (size_t)g.ntiles * g.T * nco * 64 * sizeof(T);
    if(s == RAMD_OK && cached_malloc(&P->coef, P->coef_bytes + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && cached_malloc(&P->scratch, (size_t)n * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK
       && (hipMemcpyAsync(P->ptab, ptab.data(), sizeof(int) * ptab.size(), hipMemcpyHostToDevice, b.cur) != hipSuccess
           || hipMemcpyAsync(trank, rank.data(), sizeof(int) * rank.size(), hipMemcpyHostToDevice, b.cur) != hipSuccess
           || hipMemsetAsync(P->coef, 0, P->coef_bytes, b.cur) != hipSuccess))
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK)
    {
        if(lower)
            hipLaunchKernelGGL((k_box_fill<T, true>), dim3(grid), dim3(kBlock), 0, b.cur, n, g, nco, m->rp, m->ci, (const T*)m->val, trank,
                               (T*)P->coef);
        else
            hipLaunchKernelGGL((k_box_fill<T, false>), dim3(grid), dim3(kBlock), 0, b.cur, n, g, nco, m->rp, m->ci, (const T*)m->val, trank,
                               (T*)P->coef);
        if(hipStreamSynchronize(b.cur) != hipSuccess || hipGetLastError() != hipSuccess)
            s = RAMD_ERR_HIP;
    }
    dev_free(&trank);
    if(s != RAMD_OK)
        return bail(s);
    if(getenv("RAMD_TRSV_CT_VERBOSE"))
        fprintf(stderr, "27-point pencil plan (%s): %d x %d x %d, %d pencils of %d steps, %.1f MB of coefficients\n", lower ? "lower" : "upper",
                nx, ny, nz, g.ntiles, g.T, (double)P->coef_bytes / 1e6);
    *out = P;
    return RAMD_OK;
}
template int box_build<double>(const ramd_mat_s*, bool, bool, BoxPlan**);
template int box_build<float>(const ramd_mat_s*, bool, bool, BoxPlan**);

template <typename T>
int box_run(BoxPlan* P, const T* in, T* out)
{
    if(in == out)
        RAMD_FAIL(RAMD_ERR_ARG, "27-point pencil solve: in and out must differ");
    Backend& b = backend();
    hipLaunchKernelGGL((k_fill_sentinel<T>), dim3(ew_grid(P->n)), dim3(kBlock), 0, b.cur, (int64_t)P->n, out);
    RAMD_HIP(hipMemsetAsync(P->counter, 0, sizeof(unsigned), b.cur));
    const size_t lds = (size_t)kLines * kRP * sizeof(T);
    static const int waves_env = getenv("RAMD_TRSV_BOX_WAVES") ? atoi(getenv("RAMD_TRSV_BOX_WAVES")) : 0; // (per CU; experiments)
#define BOX_LAUNCH(LO, UN)                                                                                              \
    do                                                                                                                  \
    {                                                                                                                   \
        static int occ = 0;                                                                                             \
        if(occ == 0)                                                                                                    \
        {                                                                                                               \
            RAMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_trsv_box<T, LO, UN>, 64, lds));                \
            occ = occ < 1 ? 1 : occ;                                                                                    \
        }                                                                                                               \
        const int64_t cap = (int64_t)((waves_env > 0 && waves_env < occ) ? waves_env : occ) * b.num_cu;                 \
        const unsigned nwg = (unsigned)(P->g.ntiles < cap ? P->g.ntiles : cap);                                         \
        hipLaunchKernelGGL((k_trsv_box<T, LO, UN>), dim3(nwg), dim3(64), lds, b.cur, P->g, P->ptab, (const T*)P->coef, in, out, \
                           P->counter);                                                                                \
    } while(0)
    prof_begin(RAMD_PROF_TRSV, b.cur);
    if(P->lower && P->unit)
        BOX_LAUNCH(true, true);
    else if(P->lower)
        BOX_LAUNCH(true, false);
    else if(P->unit)
        BOX_LAUNCH(false, true);
    else
        BOX_LAUNCH(false, false);
    prof_end(RAMD_PROF_TRSV, b.cur);
#undef BOX_LAUNCH
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}
template int box_run<double>(BoxPlan*, const double*, double*);
template int box_run<float>(BoxPlan*, const float*, float*);

} // namespace ramd
