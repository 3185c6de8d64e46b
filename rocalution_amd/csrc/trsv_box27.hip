// trsv_box27.hip -- sparse triangular solve on the 27-point stencil: pencils marched along x, one wave per pencil (round 6).
//
// The reference's own 3-D test operator is the 27-point Laplacian (clients/include/utility.hpp:110-177), and ILU(0) keeps its
// pattern: the lower / upper triangle of such a matrix has 13 entries per row.  The record-form box tiles of trisolve.hip take
// 6.2 ms per triangle of it at 256^3 (0.06 of the roofline: eight lanes per row, 74 steps per 512-row tile), the vendor's
// csrsv 1.4 s per iteration.  Where every row has exactly the entries its lattice position allows, nothing has to be decoded:
//   * a PENCIL is an 8 x 8 cross-section marched along x -- a SHEARED one: lane (j, k) of pencil (J, K) owns the grid line
//     y = 8 J + j - k, z = 8 K + k.  In the sheared coordinate y' = y + k the 13 dependencies of a row lie at (dy', dz) =
//     (0, 0), (-1, 0), (0 .. -2, -1): nothing points to a larger y' or z, so a pencil needs only the pencils (J - 1, K) and
//     (J, K - 1), (J + 1, K - 1) -- on the straight lattice the entries (y + 1, z - 1) would tie pencil J to pencil J + 1 and
//     back, within a window of two steps;
//   * lane (j, k) takes, at step t, the row x = t - 2 j - 2 k of its line.  With that skew all 13 dependencies were computed at
//     earlier steps: (x-1, j, k) one step ago by the lane itself, (x-1 .. x+1, j-1, k) three to one steps ago, the nine of the
//     plane below seven to one steps ago;
//   * every lane keeps the last 16 values of its line in an LDS ring (element x in column x mod 16); a step is 13 LDS reads,
//     13 multiplies and subtractions in the order of the host loop (+ the division) and one LDS write;
//   * the coefficients are packed once per analysis in exactly the order a wave consumes them (pencil, step, dependency, lane:
//     8 bytes per lane and load, fully coalesced, no column indices -- 13 x 8 instead of 13 x 12 bytes per row) and run four
//     steps ahead of their use in a register queue, with hand-counted waits (see box_wait);
//   * results leave through the ring: every second step the 16 lines whose x has just reached 7 mod 8 are written out, 64
//     contiguous bytes per line, four lanes a line.  (One 8-byte store per lane and step -- 64 cache lines per instruction --
//     cost 0.8 ms of the 1.4 ms a free-running sweep of 256^3 took, 1.9 of 2.5 ms as write-through stores.)
//   * the 22 lines of a pencil that other pencils read (j = 6, 7 of every plane; plane 7) are ALSO stored, step by step, into
//     the pencil's outflow records: 24 elements per step, contiguous, one agent-scope store per step.  The records are the
//     hand-off medium: pre-filled with a NaN sentinel before every solve (data-tagged values, as everywhere in trisolve.hip);
//     the 26 lines next to a pencil (j = -2, -1 of every plane, ten below) are rows of the records of the pencils (J-1, K),
//     (J, K-1), (J+1, K-1), and lanes 0-25 keep their halo line's ring filled a block of four steps ahead, polling only
//     what is missing;
//   * pencils are taken by ticket in order of J + 2 K: a pencil only waits for pencils with lower tickets, i.e. for waves
//     that are running.  One pencil per CU is in flight (box_run).
// Measured at 256^3 (16.8 M rows, 1056 pencils of 292 steps): 1.02 ms per triangle = 0.38 of the HBM roofline on the CSR
// bytes of the triangle; the timeline (RAMD_TRSV_BOX_DBG) shows 0.45 us per step and 11 us from a pencil's first block to
// its successor's (18 steps of skew and block granularity + the visibility of an agent-scope store), 95 such hops deep.
// Where a step's time goes (s_memtime stamps in one pencil; 0.35 us per step alone, 0.45 on the busy chip): the waits for
// loads nothing, the LDS reads and the chain of 13 multiply-subtract pairs a quarter, the halo bookkeeping of a block an eighth,
// the rest the serial issue of a lone wave.  Tried on top and not kept: the ten products of the next row formed beside the
// current row's chain (0.352 -> 0.336 us alone, nothing at 256^3), masks and address arithmetic taken out (nothing).
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
#include <type_traits>
#include <vector>

namespace ramd
{

namespace
{

constexpr int kBJ = 8, kBK = 8; // lanes of a pencil's cross-section in y and z
constexpr int kSkJ = 2, kSkK = 2; // steps a line starts after its neighbour below in y' / z
constexpr int kSkewMax = kSkJ * (kBJ - 1) + kSkK * (kBK - 1); // 28
constexpr int kRing = 16, kRP = kRing + 1; // columns of a line's ring, elements per line (padding: bank spread)
constexpr int kLW = kBJ + 2; // lines per z-row in LDS: j = -2 .. 7
constexpr int kLines = kLW * (kBK + 1); // k = -1 .. 7
constexpr int kNDep = 13;
constexpr int kNPair = 7; // coefficient PAIRS of a step: 13 dependencies + the diagonal (or a spare), 16 bytes per lane and load in fp64
constexpr int kPF = 4; // steps per block = steps the coefficient queue runs ahead
constexpr int kNHalo = 26;
constexpr int kFaceW = 24; // elements of a pencil's outflow record per step: 22 lines read by other pencils (+ 2: 64-byte multiples)
constexpr int kLead = 4; // elements a halo line's ring may hold beyond what the current block reads (< 7: see the halo phase)

struct BoxDims
{
    int nx, ny, nz, ntj, ntk, ntiles, T;
};

__host__ __device__ constexpr int box_line(int j, int k) // j in [-2, 7], k in [-1, 7]
{
    return ((k + 1) * kLW + (j + 2)) * kRP;
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
        const int K = z / kBK, k = z % kBK, J = (y + k) / kBJ, j = (y + k) % kBJ; // (sheared: y' = y + k)
        const int lane = j + kBJ * k, t = x + kSkJ * j + kSkK * k;
        // slot s of the step: pair s / 2, lanes side by side, element s % 2 of the lane's pair
        T* dst = coef + (((int64_t)trank[J + g.ntj * K] * g.T + t) * kNPair) * 128 + 2 * lane;
        auto slot = [](int sl) { return (int64_t)(sl >> 1) * 128 + (sl & 1); };
        for(int a = rp[r]; a < rp[r + 1]; ++a)
        {
            const int c = ci[a];
            if(c == (int)r)
            {
                if(nco > kNDep)
                    dst[slot(kNDep)] = val[a];
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
            dst[slot(LOWER ? i : i - 14)] = val[a];
        }
    }
}

// ---------------------------------------------------------------- the solve
// Loads and stores of the sweep are written out, with their waits counted by hand: a wave issues, per step, one store and
// NCO + 1 loads whose values it needs four steps later, and per block of four steps four polls of its halo line.  (Left to the
// compiler, the first two steps of every block waited for all but the newest 4 / 16 operations -- its bookkeeping across the
// polling loop and the back edge is conservative -- and a step cost one full memory latency: 0.37 us alone, 1.1 us on a busy
// chip.)  vmcnt counts loads and stores alike and retires them in order, so "at most N outstanding" = "everything issued before
// the newest N has landed"; every step issues the same number of operations whatever its lanes do (idle lanes store into a
// dump slot, read a clamped address), which is what makes N a constant.
template <int N>
__device__ __forceinline__ void box_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}
// (after a wait: the value is what the load brought -- nothing that uses it may be scheduled before this point)
template <typename X>
__device__ __forceinline__ void box_tie(X& v)
{
    asm volatile("" : "+v"(v));
}
// a lane's pair of coefficients (OFF: bytes, an instruction offset -- one address register pair serves four loads of a step)
template <typename T>
struct BoxPair;
template <>
struct BoxPair<double>
{
    typedef double type __attribute__((ext_vector_type(2)));
};
template <>
struct BoxPair<float>
{
    typedef float type __attribute__((ext_vector_type(2)));
};
template <int OFF>
__device__ __forceinline__ void box_ld_pair_nt(BoxPair<double>::type& r, const double* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void box_ld_pair_nt(BoxPair<float>::type& r, const float* p)
{
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2 nt" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
}
// the kNPair coefficient pairs of one step: 128 elements apart, from p.  (A step used to issue 14 loads of 8 bytes per lane; the
// ISSUE of its 17 memory instructions was 45 % of the step -- measured with s_memtime stamps --, the waits for their data nothing.)
template <typename T, int D = 0>
__device__ __forceinline__ void box_ld_step(typename BoxPair<T>::type (&c)[kNPair], const T* p)
{
    if constexpr(D < kNPair)
    {
        constexpr int kPer = 4096 / (128 * (int)sizeof(T)); // loads per 4 KB of instruction offset
        box_ld_pair_nt<(D % kPer) * 128 * (int)sizeof(T)>(c[D], p + (D / kPer) * kPer * 128);
        box_ld_step<T, D + 1>(c, p);
    }
}
__device__ __forceinline__ void box_ld(double& r, const double* p)
{
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
}
__device__ __forceinline__ void box_ld(float& r, const float* p)
{
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
}
template <int OFF = 0>
__device__ __forceinline__ void box_poll(unsigned long long& r, const double* p)
{
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2 sc1" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
}
template <int OFF = 0>
__device__ __forceinline__ void box_poll(unsigned int& r, const float* p)
{
    asm volatile("global_load_dword %0, %1, off offset:%2 sc1" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void box_st_nt(double* p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void box_st_nt(float* p, float v)
{
    asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
// agent-scope (write-through) store: what another pencil's poll sees
__device__ __forceinline__ void box_st(double* p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void box_st(float* p, float v)
{
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// a pair of consecutive elements (16 bytes in fp64, 8 in fp32), not cached on the way out.  (Two wait states after a store
// of more than 8 bytes before its data registers may change: the compiler's hazard pass does not look inside an asm statement.)
__device__ __forceinline__ void box_st_pair_nt(double* p, double a, double b)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2 v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void box_st_pair_nt(float* p, float a, float b)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    const v2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}

// slot of line (j, k) in its pencil's outflow record, -1: no other pencil reads this line
__host__ __device__ constexpr int box_face_slot(int j, int k)
{
    return j >= kBJ - 2 ? (j - (kBJ - 2)) + 2 * k : (k == kBK - 1 ? 2 * kBK + j : -1);
}

template <typename T, bool LOWER, bool UNIT>
__global__ __launch_bounds__(64) void k_trsv_box(BoxDims g, const int* __restrict__ ptab, const int* __restrict__ trank,
                                                 const T* __restrict__ coef, const T* __restrict__ in, T* __restrict__ out, T* face, T* dump,
                                                 unsigned* counter, long long* dbg)
{
    using B                = typename Sentinel<T>::bits;
    constexpr int NCO      = kNDep + (UNIT ? 0 : 1);
    constexpr int kStepOps = kNPair + 2; // a step: the store into the outflow record, the coefficient pairs, one right-hand side
    using P2               = typename BoxPair<T>::type;
    constexpr int kFlushOps = 2; // ... an odd step also the two stores of a flush
    constexpr int kBlockOps = kPF + kPF * kStepOps + 2 * kFlushOps; // polls, four steps, two flushes
    extern __shared__ __attribute__((aligned(16))) char box_lds[];
    T*        ring = reinterpret_cast<T*>(box_lds);
    const int lane = threadIdx.x;
    const int cj = lane & 7, ck = lane >> 3;
    const int s_own = kSkJ * cj + kSkK * ck;
    const int a_own = box_line(cj, ck);
    const int fs_own = box_face_slot(cj, ck);
    // the lines of the 13 dependencies, in the order of the host loop (the upper solve runs on the mirrored lattice: the lower
    // list reversed)
    int a_dep[kNDep], dxs[kNDep];
#pragma unroll
    for(int d = 0; d < kNDep; ++d)
    {
        const int i = LOWER ? d : kNDep - 1 - d;
        a_dep[d]    = box_line(cj + box_dj(i) + box_dk(i), ck + box_dk(i)); // (y' = y + k)
        dxs[d]      = box_dx(i);
    }
    // the halo line of this lane (lanes 0 .. 25): h < 10: (j = h - 2, k = -1), else (j = -2 / -1, k = (h - 10) / 2) ...
    const int hj = lane < 10 ? lane - 2 : -2 + ((lane - 10) & 1), hk = lane < 10 ? -1 : (lane - 10) >> 1;
    const int hs = kSkJ * hj + kSkK * hk;
    const int a_halo = lane < kNHalo ? box_line(hj, hk) : 0;
    // ... is line (pj, pk) of the pencil (J + dJ, K + dK): j = 6, 7 of the pencil before in y', or plane 7 of the slab below --
    // whose y' runs 7 ahead of this slab's at its plane -1
    const int dJ = hk >= 0 ? -1 : (hj < 0 ? 0 : 1), dK = hk >= 0 ? 0 : -1;
    const int pj = hk >= 0 || hj < 0 ? hj + kBJ : hj, pk = hk >= 0 ? hk : kBK - 1;
    const int p_first = (kSkJ * pj + kSkK * pk) * kFaceW + box_face_slot(pj, pk); // element 0 of that line in its pencil's records
    // the four lines this lane helps to write out: a flush takes the 16 lines of one class (j + k) mod 4 = c, four lanes a line
    const int fr = lane >> 2, fc = lane & 3;
    for(;;)
    {
        unsigned tk = 0;
        if(lane == 0)
            tk = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int q = __builtin_amdgcn_readfirstlane((int)tk);
        if(q >= g.ntiles)
            break;
        if(dbg && lane == 0)
            dbg[4 * q] = (long long)wall_clock64();
        const int pt = __builtin_amdgcn_readfirstlane(ptab[q]);
        const int J = pt & 0xffff, K = pt >> 16;
        // element x of line (y, z) in the natural-order vectors: base + x (lower) / base - x (upper, mirrored)
        auto line_base = [&](int y, int z) -> int64_t {
            return LOWER ? ((int64_t)z * g.ny + y) * g.nx : ((int64_t)(g.nz - 1 - z) * g.ny + (g.ny - 1 - y)) * g.nx + (g.nx - 1);
        };
        auto gidx = [&](int64_t base, int x) -> int64_t { return LOWER ? base + x : base - x; };
        const int  yy = kBJ * J + cj - ck, zz = kBK * K + ck;
        const bool valid = yy >= 0 && yy < g.ny && zz < g.nz;
        const int64_t gb = valid ? line_base(yy, zz) : 0;
        const int  hy = kBJ * J + hj - hk, hz = kBK * K + hk;
        const bool hvalid = lane < kNHalo && hy >= 0 && hy < g.ny && hz >= 0 && hz < g.nz;
        // (a line of the lattice next to this pencil lies in a pencil of the table)
        // (a lane without a halo line polls its own pencil's first record: every idle lane of the chip on ONE address was the
        //  busiest line of the L2 once already -- see the dump slots)
        const int pq = hvalid ? trank[(J + dJ) + g.ntj * (K + dK)] : q;
        const T*  hp = face + ((int64_t)pq * g.T * kFaceW + (hvalid ? p_first : 0)); // element x: hp[x * kFaceW]
        T*        fp = face + ((int64_t)q * g.T * kFaceW + (fs_own >= 0 ? fs_own : 0)); // step t: fp[t * kFaceW]
        const int fs_spare = kFaceW - 2 + (lane & 1) - (fs_own >= 0 ? fs_own : 0);
        T*        dmp = dump + (int64_t)q * 128; // where the idle lanes of this pencil's flushes store
        int64_t fgb[4];
        int     fla[4], fsk[4];
        bool    fok[4];
#pragma unroll
        for(int c = 0; c < 4; ++c)
        {
            const int fk = fr >> 1, fj = ((c - fk) & 3) + 4 * (fr & 1);
            const int fy = kBJ * J + fj - fk, fz = kBK * K + fk;
            fok[c] = fy >= 0 && fy < g.ny && fz < g.nz;
            fgb[c] = fok[c] ? line_base(fy, fz) : 0;
            fla[c] = box_line(fj, fk);
            fsk[c] = kSkJ * fj + kSkK * fk;
        }
        // rings start at +0
        for(int i = lane; i < kLines * kRP; i += 64)
            ring[i] = (T)0;
        const T* cb = coef + ((int64_t)q * g.T) * kNPair * 128 + 2 * lane;
        // queue: coefficients and right-hand side of the next four steps
        P2 cq[kPF][kNPair];
        T  rq[kPF];
#pragma unroll
        for(int i = 0; i < kPF; ++i)
        {
            box_ld_step<T>(cq[i], cb + (int64_t)i * kNPair * 128);
            const int x = i - s_own;
            box_ld(rq[i], in + gidx(gb, valid ? min(max(x, 0), g.nx - 1) : 0) * (valid ? 1 : 0));
        }
        // (the first block finds everything landed: its steps have fewer operations behind their loads than the later ones')
        box_wait<0>();
#pragma unroll
        for(int i = 0; i < kPF; ++i)
        {
#pragma unroll
            for(int d = 0; d < kNPair; ++d)
                box_tie(cq[i][d]);
            box_tie(rq[i]);
        }
        int  xf    = hvalid ? -1 : 0x3fffffff; // the halo line's ring holds every element up to xf
        bool hdone = false;
        B   hv[kPF];
        int hx = 0, hn = 0;
#pragma unroll
        for(int e = 0; e < kPF; ++e)
            hv[e] = Sentinel<T>::value;
        for(int t0 = 0; t0 < g.T; t0 += kPF)
        {
            // ---- halo: what the last block's requests brought, then whatever this block still needs, then the next requests.
            // A halo line may run ahead of what its readers need by kLead elements and no more: the reader furthest behind
            // (six steps after the line's own skew) still reads element need - 9 in this block, and element e lands in the
            // column of element e - 16.  (Without the cap a line whose pencil had finished long ago kept the lead it had at
            // step 0 -- 7 elements for the lines of the largest skew -- and overwrote what lane (1, 7) was about to read.)
            const int need = min(g.nx - 1, t0 + kPF - 1 - hs - 1);
            box_wait<kBlockOps - kPF>();
#pragma unroll
            for(int e = 0; e < kPF; ++e)
                box_tie(hv[e]);
#pragma unroll
            for(int e = 0; e < kPF; ++e)
                if(e < hn && hx + e == xf + 1 && hx + e <= need + kLead && hv[e] != Sentinel<T>::value)
                {
                    ring[a_halo + ((xf + 1) & (kRing - 1))] = Sentinel<T>::from_bits(hv[e]);
                    ++xf;
                }
            int spins = 0;
            while(__ballot(xf < need) != 0ull)
            {
                spin_guard(spins);
                bool got = false;
                // (every lane loads -- a finished or idle lane its line's first element -- so that the wait below is the wave's)
                B v;
                box_poll(v, hp + (int64_t)(xf < need ? xf + 1 : 0) * kFaceW);
                box_wait<0>();
                box_tie(v);
                if(xf < need)
                {
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
            if(dbg && lane == 0)
            {
                if(t0 == 0)
                    dbg[4 * q + 1] = (long long)wall_clock64();
                dbg[4 * q + 3] += spins;
            }
            // (a complete halo line: the same +0 behind its end -- its readers are at element nx - 13 or later)
            if(hvalid && xf == g.nx - 1 && !hdone)
            {
                ring[a_halo + (g.nx & (kRing - 1))] = (T)0;
                hdone = true;
            }
            hx = xf + 1;
            hn = hvalid ? min(kPF, g.nx - hx) : 0;
            // (one address, four instruction offsets: an element behind the line's end is a record of a later step -- never
            //  accepted, inside the allocation by its slack)
            {
                const T* php = hp + (int64_t)(hvalid ? hx : 0) * kFaceW;
                box_poll<0>(hv[0], php);
                box_poll<kFaceW * (int)sizeof(T)>(hv[1], php);
                box_poll<2 * kFaceW * (int)sizeof(T)>(hv[2], php);
                box_poll<3 * kFaceW * (int)sizeof(T)>(hv[3], php);
            }
            // ---- four steps
            auto step = [&](auto step_index) {
                constexpr int i = decltype(step_index)::value;
                const int  t = t0 + i, x = t - s_own;
                const bool act = valid && x >= 0 && x < g.nx;
                // (everything issued since this step's loads, one block ago: a block less the step's own operations)
                box_wait<kBlockOps - kStepOps - (i & 1 ? kFlushOps : 0)>();
#pragma unroll
                for(int d = 0; d < kNPair; ++d)
                    box_tie(cq[i][d]);
                box_tie(rq[i]);
                T v[kNDep];
#pragma unroll
                for(int d = 0; d < kNDep; ++d)
                    v[d] = ring[a_dep[d] + ((x + dxs[d]) & (kRing - 1))];
                T sum = rq[i];
#pragma unroll
                for(int d = 0; d < kNDep; ++d)
                {
                    // (x + 1 behind the line's end: the column holds +0 by then -- see the write below and the halo phase)
                    sum -= cq[i][d >> 1][d & 1] * v[d];
                }
                if(!UNIT)
                    sum = sum / cq[i][kNDep >> 1][kNDep & 1];
                if(act)
                    ring[a_own + (x & (kRing - 1))] = sum;
                // the column element nx would take still holds element nx - 16; the readers of this line are past it (the one
                // furthest behind reads element x - 7 now) and will read it as "x + 1" at the line's end: +0, as the host loop's
                // row has no such entry (a line shorter than 16 never wrote that column)
                if(valid && x == g.nx - 8)
                    ring[a_own + (g.nx & (kRing - 1))] = (T)0;
                // the outflow lines' rows of this step: one record, contiguous.  (The other lanes store into the record's two
                // spare slots: a line this wave writes anyway -- one dump for all waves would be the busiest address of the chip.)
                box_st(fp + ((int64_t)t * kFaceW + ((act && fs_own >= 0) ? 0 : fs_spare)), sum);
                if(i & 1)
                {
                    // every second step the 16 lines whose x has just reached 7 mod 8 are written out, 64 bytes a line: the
                    // lines of class (j + k) mod 4 = (t - 7) / 2 mod 4, four lanes a line, two elements a lane
                    const bool hi = (i == 1) ? ((t0 >> 1) & 2) != 0 : ((t0 >> 1) & 2) == 0; // classes 1 | 3 at i = 1, 2 | 0 at i = 3
                    const int     ca = (i == 1) ? 1 : 0, cb2 = (i == 1) ? 3 : 2;
                    const int64_t b  = hi ? fgb[cb2] : fgb[ca];
                    const int     la = hi ? fla[cb2] : fla[ca];
                    const int     xe = t - (hi ? fsk[cb2] : fsk[ca]) - 7 + 2 * fc;
                    const bool    ok = (hi ? fok[cb2] : fok[ca]) && xe >= 0;
                    const T e0 = ring[la + (xe & (kRing - 1))], e1 = ring[la + (xe & (kRing - 1)) + 1];
                    const bool both = ok && xe + 1 < g.nx, one = ok && xe + 1 == g.nx;
                    if(LOWER)
                        box_st_pair_nt(both ? out + b + xe : dmp + 2 * lane, e0, e1);
                    else
                        box_st_pair_nt(both ? out + b - xe - 1 : dmp + 2 * lane, e1, e0);
                    box_st_nt(one ? out + gidx(b, xe) : dmp + lane, e0);
                }
                // refill the queue slot for step t + 4
                const int tn = min(t + kPF, g.T - 1);
                box_ld_step<T>(cq[i], cb + (int64_t)tn * kNPair * 128);
                const int xn = x + kPF;
                box_ld(rq[i], in + gidx(gb, valid ? min(max(xn, 0), g.nx - 1) : 0) * (valid ? 1 : 0));
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            static_assert(kPF == 4, "the block is written out as four steps");
        }
        if(dbg && lane == 0)
            dbg[4 * q + 2] = (long long)wall_clock64();
        // (loads of the last block that nobody looks at: they land before their registers get another use)
        box_wait<0>();
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
    void*    coef  = nullptr; // [ntiles][T][kNPair][64][2]
    void*    scratch = nullptr; // [n]
    void*    dump    = nullptr; // [ntiles][128] where the idle lanes of a flush store
    int*     trank   = nullptr; // [ntj * ntk] pencil -> ticket
    void*    face    = nullptr; // [ntiles][T][kFaceW] the rows other pencils read, in the order they are computed (the hand-off medium)
    size_t   face_elems = 0;
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
    if(P->dump)
        (void)cached_free(P->dump);
    if(P->face)
        (void)cached_free(P->face);
    dev_free(&P->trank);
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
    g.ntj = (ny + std::min(nz, kBK) - 1 + kBJ - 1) / kBJ, g.ntk = (nz + kBK - 1) / kBK, g.ntiles = g.ntj * g.ntk; // (sheared pencils)
    // (steps: the last line's last flush comes 7 steps after the first multiple of 8 at or above nx)
    g.T = (((nx + 7) / 8 * 8 + kSkewMax + kPF - 1) / kPF) * kPF;
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
    int*& trank = P->trank;
    s           = dev_alloc(&P->ptab, g.ntiles);
    if(s == RAMD_OK)
        s = dev_alloc(&trank, g.ntiles);
    if(s == RAMD_OK)
        s = dev_alloc(&P->counter, 4);
    const int nco = kNDep + (unit ? 0 : 1);
    P->coef_bytes = (size_t)g.ntiles * g.T * kNPair * 128 * sizeof(T);
    if(s == RAMD_OK && cached_malloc(&P->coef, P->coef_bytes + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && cached_malloc(&P->scratch, (size_t)n * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    if(s == RAMD_OK && cached_malloc(&P->dump, (size_t)g.ntiles * 128 * sizeof(T) + kPad) != hipSuccess)
        s = RAMD_ERR_HIP;
    P->face_elems = (size_t)g.ntiles * g.T * kFaceW;
    // (+ 8 records: the polls of a line's last block reach up to four steps behind the pencil's last record)
    if(s == RAMD_OK && cached_malloc(&P->face, (P->face_elems + 8 * kFaceW) * sizeof(T) + kPad) != hipSuccess)
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
    hipLaunchKernelGGL((k_fill_sentinel<T>), dim3(ew_grid((int64_t)P->face_elems)), dim3(kBlock), 0, b.cur, (int64_t)P->face_elems,
                       (T*)P->face);
    RAMD_HIP(hipMemsetAsync(P->counter, 0, sizeof(unsigned), b.cur));
    const size_t lds = (size_t)kLines * kRP * sizeof(T);
    // pencils in flight per CU: ONE by default.  A pencil that waits polls, and what it polls for is produced by a pencil with a
    // lower ticket -- more resident pencils are more pollers, not more work in flight (256^3: 1.04 ms per triangle at one per
    // CU, 1.50 at two, 1.98 at as many as fit; 0: as many as fit)
    static const int waves_env = getenv("RAMD_TRSV_BOX_WAVES") ? atoi(getenv("RAMD_TRSV_BOX_WAVES")) : 1;
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
        hipLaunchKernelGGL((k_trsv_box<T, LO, UN>), dim3(nwg), dim3(64), lds, b.cur, P->g, P->ptab, P->trank, (const T*)P->coef, in,   \
                           out, (T*)P->face, (T*)P->dump, P->counter, dbg);                                                                           \
    } while(0)
    // RAMD_TRSV_BOX_DBG=<file>: per pencil (ticket order) the 100 MHz clock at its ticket, at its first block, at its end, and
    // the polls that found nothing (a diagnostic: the run is synchronous and writes a text file)
    const char* dbg_path = getenv("RAMD_TRSV_BOX_DBG");
    long long*  dbg      = nullptr;
    if(dbg_path)
    {
        RAMD_TRY(dev_alloc(&dbg, (size_t)4 * P->g.ntiles));
        RAMD_HIP(hipMemsetAsync(dbg, 0, sizeof(long long) * 4 * P->g.ntiles, b.cur));
    }
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
    if(dbg)
    {
        std::vector<long long> h((size_t)4 * P->g.ntiles);
        std::vector<int>       pt((size_t)P->g.ntiles);
        RAMD_HIP(hipMemcpyAsync(h.data(), dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost, b.cur));
        RAMD_HIP(hipMemcpyAsync(pt.data(), P->ptab, sizeof(int) * pt.size(), hipMemcpyDeviceToHost, b.cur));
        RAMD_HIP(hipStreamSynchronize(b.cur));
        dev_free(&dbg);
        if(FILE* f = fopen(dbg_path, "w"))
        {
            fprintf(f, "# %s %d x %d x %d: ticket J K ticket_us first_block_us end_us empty_polls (100 MHz clock, from the first ticket)\n",
                    P->lower ? "lower" : "upper", P->g.nx, P->g.ny, P->g.nz);
            long long t0 = h[0];
            for(int q = 0; q < P->g.ntiles; ++q)
                t0 = std::min(t0, h[(size_t)4 * q]);
            for(int q = 0; q < P->g.ntiles; ++q)
                fprintf(f, "%d %d %d %.2f %.2f %.2f %lld\n", q, pt[(size_t)q] & 0xffff, pt[(size_t)q] >> 16, (h[(size_t)4 * q] - t0) / 100.0,
                        (h[(size_t)4 * q + 1] - t0) / 100.0, (h[(size_t)4 * q + 2] - t0) / 100.0, h[(size_t)4 * q + 3]);
            fclose(f);
        }
    }
    return RAMD_OK;
}
template int box_run<double>(BoxPlan*, const double*, double*);
template int box_run<float>(BoxPlan*, const float*, float*);

} // namespace ramd
