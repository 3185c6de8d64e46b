// trsv_lattice.hip -- sparse triangular solve on lattice operators: pencils marched along x, one wave per pencil.
//
// Replaces, for the triangles it recognises, the record-form box tiles of trisolve.hip (k_trsv_rec): that kernel decodes a
// record per row and step (column codes -> LDS addresses, ~85 vector + ~100 scalar instructions per step) and is bound by that
// instruction chain, 0.8 us per dependency level.  When the triangle is the lower / upper part of a
// 5- / 7-point operator on an nx x ny x nz lattice in lexicographic numbering -- every row r has exactly the entries
// r -+ 1, r -+ nx, r -+ nx ny its position on the lattice allows: ILU(0) of the Poisson operator, north_star's second target --
// nothing has to be decoded at all:
//   * a PENCIL is an 8 x 8 (y, z) cross-section marched along x.  Lane (j, k) of the wave takes, at step t, the row
//     i = t - j - k of its grid line.  All three dependencies of that row were computed one step earlier: (i-1, j, k) by the
//     lane itself (a register), (i, j-1, k) and (i, j, k-1) by the lanes one below in y / z, which left them in the LDS ring a
//     step ago.  The ring is SKEWED: element i of the line of lane (j, k) sits in column (i + j + k) mod 32, so at step t every
//     lane reads and writes column t mod 32 (t - 1 for the neighbours): per-lane base + compile-time offset, no address
//     arithmetic in a step.  A step is 3 LDS reads, 3 multiplies, 3 subtractions (+ the division), 1 LDS write.
//   * right-hand side and result pass through the same ring in blocks of 16 x-positions: the wave loads the natural-order
//     vector with whole 128-byte runs (8 lines x 16 elements per instruction), drops them into the skewed columns one block
//     ahead, and writes finished blocks back the same way -- the natural-order vectors are read and written exactly once, in
//     line-sized pieces, without index lists, sentinels or a position-order scratch vector (in and out may be one vector).
//   * the coefficients are re-packed once per analysis in exactly the order the wave consumes them (pencil, step pair,
//     coefficient, lane): 16-byte, fully coalesced loads straight into a rolling register queue a few steps ahead.
//   * only the two outflow faces of a pencil (2/8 of its rows) leave through memory: per block, as data-tagged 8-byte
//     granules (NaN sentinel, agent scope) into a face buffer the successor pencil polls one block ahead; the reader puts
//     the sentinel back behind its read, so a solve needs no fill pass.
//   * pencils are taken by ticket in order of (J + K): a pencil only waits for pencils with lower tickets, i.e. for waves that
//     are running -- the progress argument of the box-tile kernel.  4096 pencils at 512^3 instead of 262144 tiles.
// Arithmetic per row: the subtractions in ascending column order (lower: z, y, x neighbour; upper: x, y, z), unfused multiply
// and subtract, then the division by the stored diagonal -- src/base/host/host_matrix_csr.cpp:1163-1221 (LUSolve),
// :1357-1404 (LSolve), :1420-1466 (USolve).  Rows the lattice does not have (the skew's fill and drain, lines beyond ny / nz
// in the last pencils) run with coefficients +0, right-hand side +0 and diagonal 1: they stay +0 and feed +0 * c = +0 into the
// first real rows, which leaves those bit for bit what the host loop computes without the term.
#include "trsv_lattice.hpp"
#include "device_utils.hpp"
#include "matrix_impl.hpp"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace ramd
{

namespace
{

constexpr int kLJ = 8, kLK = 8; // lanes of a pencil's cross-section in y and z
constexpr int kXB    = 16; // x-positions per staging block
constexpr int kRing  = 32; // columns of the ring = steps of one loop body
constexpr int kPitch = kRing + 1; // elements per line: [0] mirrors column 31 (what step 0 of a body reads as "t - 1")
constexpr int kLines = kLJ * kLK, kFaceLines = kLJ + kLK;
constexpr int kLdsElems = (kLines + kFaceLines) * kPitch;

template <typename T>
struct LatT;
template <>
struct LatT<double>
{
    using V2   = v2f64;
    using bits = unsigned long long;
    static constexpr bits sentinel = 0x7FF8DEADBEEF0001ull; // the quiet NaN trisolve.hip uses as "not there yet"
    __device__ static __forceinline__ bits   to_bits(double v) { return (bits)__double_as_longlong(v); }
    __device__ static __forceinline__ double from_bits(bits b) { return __longlong_as_double((long long)b); }
};
template <>
struct LatT<float>
{
    typedef float V2 __attribute__((ext_vector_type(2)));
    using bits = unsigned int;
    static constexpr bits sentinel = 0x7FDEAD01u;
    __device__ static __forceinline__ bits  to_bits(float v) { return __float_as_uint(v); }
    __device__ static __forceinline__ float from_bits(bits b) { return __uint_as_float(b); }
};

struct LatDims
{
    int nx, ny, nz;
    int nj, nk, npencil; // pencils in y and z
    int nsb; // staging blocks of a pencil: ceil(nx / 16)
    int nbody; // loop bodies (32 steps each) of a pencil: 2 nbody >= nsb + 2
    int nfb; // face batches (4 columns of both faces = 64 elements) a pencil publishes: columns 0 .. nx + 6
};

// ---------------------------------------------------------------- detection
// smallest offset |col - row| > thr over the triangle; *flag raised by a row with more than 3 entries in the triangle
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_lat_min_offset(int n, const int* __restrict__ rp, const int* __restrict__ ci, int thr,
                                                           int* __restrict__ out_min, int* __restrict__ flag)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    int           mn  = 0x7fffffff;
    bool          bad = false;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gsz)
    {
        int cnt = 0;
        for(int a = rp[r]; a < rp[r + 1]; ++a)
        {
            const int d = LOWER ? (int)r - ci[a] : ci[a] - (int)r;
            if(d > 0)
            {
                ++cnt;
                if(d > thr)
                    mn = min(mn, d);
            }
        }
        bad = bad || cnt > 3;
    }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
        mn = min(mn, __shfl_xor(mn, off, 64));
    if((threadIdx.x & 63) == 0 && mn != 0x7fffffff)
        atomicMin(out_min, mn);
    if(bad)
        *flag = 1;
}

// every row holds exactly the entries its lattice position allows, in ascending columns (+ the diagonal where it is needed)
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_lat_verify(int n, const int* __restrict__ rp, const int* __restrict__ ci, int nx, int ny,
                                                       int nz, int need_diag, int* __restrict__ flag)
{
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    const int     nxny = nx * ny;
    bool          bad  = false;
    for(int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gsz)
    {
        const int  x  = (int)(r % nx), y = (int)((r / nx) % ny), z = (int)(r / nxny);
        const bool hx = LOWER ? x > 0 : x < nx - 1, hy = LOWER ? y > 0 : y < ny - 1, hz = LOWER ? z > 0 : z < nz - 1;
        int        seen = 0, cnt = 0, prev = -1;
        bool       diag = false;
        for(int a = rp[r]; a < rp[r + 1]; ++a)
        {
            const int c = ci[a];
            bad         = bad || c <= prev; // (sorted, no duplicates: the order of the host loop is the storage order)
            prev        = c;
            const int d = LOWER ? (int)r - c : c - (int)r;
            if(d == 0)
                diag = true;
            if(d > 0)
            {
                ++cnt;
                if(d == 1 && hx)
                    seen |= 1;
                else if(d == nx && hy)
                    seen |= 2;
                else if(d == nxny && hz)
                    seen |= 4;
                else
                    bad = true;
            }
        }
        const int want = (hx ? 1 : 0) | (hy ? 2 : 0) | (hz ? 4 : 0);
        bad            = bad || seen != want || cnt != __popc(want) || (need_diag && !diag);
    }
    if(bad)
        *flag = 1;
}

// ---------------------------------------------------------------- coefficients in sweep order
// element ((q * npair + pair) * NC + c) * 128 + lane * 2 + e  =  coefficient c of the row lane `lane` of pencil q takes at step
// 2 pair + e.  c runs in the order of the subtractions: lower (z, y, x[, diagonal]), upper (x, y, z[, diagonal]).
template <typename T, bool LOWER>
__global__ __launch_bounds__(kBlock) void k_lat_fill(LatDims g, int npair, int nc, const int* __restrict__ ptab,
                                                     const int* __restrict__ rp, const int* __restrict__ ci, const T* __restrict__ val,
                                                     T* __restrict__ coef)
{
    const int64_t total = (int64_t)g.npencil * npair * 128;
    const int64_t gsz   = (int64_t)gridDim.x * blockDim.x;
    const int     nxny  = g.nx * g.ny;
    for(int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gsz)
    {
        const int     e    = (int)(id & 1), lane = (int)((id >> 1) & 63);
        const int64_t qp   = id >> 7;
        const int     pair = (int)(qp % npair), q = (int)(qp / npair);
        const int     pj = ptab[q] & 0xffff, pk = ptab[q] >> 16;
        const int     j = lane & 7, k = lane >> 3;
        const int     i = 2 * pair + e - j - k, sy = pj * kLJ + j, sz = pk * kLK + k;
        T             cf[4] = {(T)0, (T)0, (T)0, (T)1};
        if(i >= 0 && i < g.nx && sy < g.ny && sz < g.nz)
        {
            const int x = LOWER ? i : g.nx - 1 - i, y = LOWER ? sy : g.ny - 1 - sy, z = LOWER ? sz : g.nz - 1 - sz;
            const int r = x + g.nx * (y + g.ny * z);
            for(int a = rp[r]; a < rp[r + 1]; ++a)
            {
                const int d = LOWER ? r - ci[a] : ci[a] - r;
                const T   v = val[a];
                if(d == 0)
                    cf[3] = v;
                else if(d == 1)
                    cf[LOWER ? 2 : 0] = v;
                else if(d == g.nx)
                    cf[1] = v;
                else if(d == nxny)
                    cf[LOWER ? 0 : 2] = v;
            }
        }
        T* dst = coef + (qp * nc) * 128 + lane * 2 + e;
        for(int c = 0; c < nc; ++c)
            dst[(int64_t)c * 128] = cf[c];
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_lat_fill_sentinel(int64_t n, T* __restrict__ w)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const T       s   = LatT<T>::from_bits(LatT<T>::sentinel);
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        w[i] = s;
}

// ---------------------------------------------------------------- the solve
// Every vector memory operation of the pencil loop is issued by hand, unconditionally, in a fixed order -- lanes and phases
// that have nothing to move read a block of zeros / write a per-workgroup dump line instead of being masked off -- so that
// "the loads issued at the previous block boundary have arrived" is an exact operation count (the hardware returns LOADS in
// order -- see LatSched: the stores in between are not counted): s_waitcnt vmcnt(N) with N computed from the template
// parameters below, never a drain in the steady state.  (Left to the compiler the boundary code came out with vmcnt(0) in front of every store and 326 VGPRs.)
// A count may only err towards waiting longer: N has to be <= the operations really issued after the one waited for.
template <int I, int N, typename F>
__device__ __forceinline__ void lat_for(F&& f)
{
    if constexpr(I < N)
    {
        f(std::integral_constant<int, I>{});
        lat_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void lat_wait()
{
    static_assert(N >= 0, "vmcnt");
#ifdef RAMD_LAT_DRAIN // (diagnostic build: every counted wait is a drain)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
#endif
}
// (s_nop after a store: its data registers may be rewritten by the next instruction the compiler places, and the compiler's hazard
// pass does not look inside an asm statement.  gfx950 wants TWO wait states after a store of more than 8 bytes: with one, the first
// data register was seen holding the next instruction's result -- but only when other processes kept the memory pipeline busy.)
__device__ __forceinline__ v2f64 lat_ld_pair(const double* p)
{
    v2f64 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ LatT<float>::V2 lat_ld_pair(const float* p)
{
    LatT<float>::V2 r;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ double lat_ld_elem(const double* p)
{
    double r;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ float lat_ld_elem(const float* p)
{
    float r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ double lat_ld_elem_sc1(const double* p)
{
    double r;
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ float lat_ld_elem_sc1(const float* p)
{
    float r;
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void lat_st_elem_sc1(double* p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lat_st_elem_sc1(float* p, float v)
{
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ v2f64 lat_ld_pair_sc1(const double* p)
{
    v2f64 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ LatT<float>::V2 lat_ld_pair_sc1(const float* p)
{
    LatT<float>::V2 r;
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void lat_st_pair_nt(double* p, v2f64 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lat_st_pair_nt(float* p, LatT<float>::V2 v)
{
    asm volatile("global_store_dwordx2 %0, %1, off nt\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lat_st_elem_nt(double* p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off nt\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lat_st_elem_nt(float* p, float v)
{
    asm volatile("global_store_dword %0, %1, off nt\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lat_st_pair_sc1(double* p, v2f64 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lat_st_pair_sc1(float* p, LatT<float>::V2 v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
// coefficient c of a step pair: uniform base + lane offset + c KiB (fp64; half that for fp32)
template <int OFF>
__device__ __forceinline__ v2f64 lat_ld_coef(const v2f64* sbase, unsigned voff)
{
    v2f64 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    return r;
}
template <int OFF>
__device__ __forceinline__ LatT<float>::V2 lat_ld_coef(const LatT<float>::V2* sbase, unsigned voff)
{
    LatT<float>::V2 r;
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3 nt" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    return r;
}
template <typename X>
__device__ __forceinline__ void lat_tie(X& v) // (nothing that reads v may move above this point)
{
    asm volatile("" : "+v"(v));
}

// The order in which a loop body (32 steps) issues its vector memory operations.  Before step u:
//   stage 0  u % 16 == 0: the natural-order stores of block ph - 2                       (IO operations)
//   stage 1  u % 16 == 0: the natural-order loads of block ph + 1                        (IO)
//   stage 2  u % 4 == 3:  the publication of face batch (t - 11) / 4                     (1)
//   stage 3  u % 4 == 0:  the sentinel reset of face batch t / 4, then the poll of batch t / 4 + A / 4   (2)
// and after step u:
//   stage 4  u odd: the coefficient loads of step pair u / 2 + D                         (NC)
// A wait names the operation it needs by (step, stage) and sits in front of a (step, stage); what lies between is counted here.
template <int NC, int IO>
struct LatSched
{
    // LOADS only.  vmcnt counts loads and stores alike, but only the loads return in the order they were issued: a store that
    // was issued after a load may be acknowledged before the load's data is there (the compiler's own wait insertion treats a
    // mix of pending loads and stores on this target as "out of order" for that reason).  A load with M younger loads has
    // arrived once vmcnt <= M, whatever the stores in between do; counting the stores as well (rounds 5 and before) waits for
    // too little as soon as loads are slow and stores are not: found in round 6 when six processes shared the device -- a
    // quarter of the solves used the coefficients of eight steps earlier (errors of 1e-7: the ILU(0) factors of the Poisson
    // operator change that little along a grid line), alone on the device none did.
    static constexpr int ops(int u, int st)
    {
        return st == 0 ? 0 // (the natural-order stores of block ph - 2)
               : st == 1 ? (u % 16 == 0 ? IO : 0)
               : st == 2 ? 0 // (the publication of a face batch)
               : st == 3 ? (u % 4 == 0 ? 1 : 0) // (the sentinel reset is a store; the poll of the next batch)
                         : (u % 2 == 1 ? NC : 0);
    }
    static constexpr int upto(int u, int st) // operations of the body up to and including stage st of step u
    {
        int n = 0;
        for(int v = 0; v < 32; ++v)
            for(int t = 0; t < 5; ++t)
                if(v < u || (v == u && t <= st))
                    n += ops(v, t);
        return n;
    }
    static constexpr int total = upto(31, 4);
    // operations issued after the last one of (ut, st) and before stage sw of step uw (ut may be negative: the body before)
    static constexpr int younger(int ut, int st, int uw, int sw)
    {
        const int before_wait = sw == 0 ? (uw == 0 ? 0 : upto(uw - 1, 4)) : upto(uw, sw - 1);
        const int target      = ut >= 0 ? upto(ut, st) : upto(ut + 32, st) - total;
        return before_wait - target;
    }
};

// NC = 3 (unit diagonal) or 4; D = step pairs the coefficient queue runs ahead (divides 8); W = waves per SIMD the register
// budget is cut for (2: 256 registers -- the LDS ring allows 7 waves per CU).  W = 1 is NOT an option: beyond 256 registers
// the compiler parks values in accumulation registers, and a copy of a register whose load is still in flight copies garbage.
template <typename T, bool LOWER, int NC, bool A16, int D, int W>
__global__ __launch_bounds__(64, W) void k_trsv_lat(LatDims g, int npair, const int* __restrict__ ptab,
                                                 const typename LatT<T>::V2* __restrict__ coef, const T* in, T* out, T* face,
                                                 const T* __restrict__ zeros, T* __restrict__ dump_all, unsigned* counter,
                                                 unsigned base, int nodep)
{
    using V2 = typename LatT<T>::V2;
    using B  = typename LatT<T>::bits;
    static_assert(8 % D == 0, "the queue slot of a step pair has to be a compile-time constant");
    static_assert(W == 2, "see above");
    constexpr bool DIV   = NC == 4;
    constexpr int  kIO   = A16 ? kLK : 2 * kLK; // natural-order loads (= stores) of a block
    constexpr int  kPoll = 8; // steps a face batch is polled ahead of its use (a multiple of 4)
    using Sched          = LatSched<NC, kIO>;
    extern __shared__ __attribute__((aligned(16))) char lat_lds[];
    T*        ring = reinterpret_cast<T*>(lat_lds);
    const int lane = threadIdx.x;
    auto      uni  = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    // the compute view: lane (cj, ck) owns line `lane`
    const int cj = lane & 7, ck = lane >> 3;
    const int a_own = lane * kPitch + 1;
    const int a_zn  = ck > 0 ? (lane - kLJ) * kPitch : (kLines + kLK + cj) * kPitch + 1; // (the face lines hold column t, not t - 1)
    const int a_yn  = cj > 0 ? (lane - 1) * kPitch : (kLines + ck) * kPitch + 1;
    // the staging view: instruction q of a block moves lines (sj, q), the lane takes the element pair sm of its line
    const int sj0 = lane >> 3, sm0 = lane & 7;
    // the face view: a batch is columns 4 c .. 4 c + 3 of the 8 lines of the y face (lanes 0 - 31) and of the z face (32 - 63),
    // a column being the step at which the successor's lane (0, k) / (j, 0) uses the value: tau = i + k resp. i + j
    const int fl = (lane >> 2) & 7, fm = lane & 3;
    const int a_fin  = (kLines + (lane >> 2)) * kPitch + 1 + fm; // + column: where a polled value goes
    const int a_fout = (lane < 32 ? (kLJ * fl + kLJ - 1) : (kLJ * (kLK - 1) + fl)) * kPitch + 1 + fm; // + (column + 7): where a face value is
    const B   sent = LatT<T>::sentinel;
    const T*  zsrc = zeros + 2 * lane; // what lanes and phases without data read
    T*        dump = dump_all + (size_t)blockIdx.x * 128 + 2 * lane; // ... and write
    const unsigned cvoff = (unsigned)(lane * sizeof(V2));

    for(;;)
    {
        unsigned tk = 0;
        if(lane == 0)
            tk = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
        const int q = uni((int)tk);
        if(q >= g.npencil)
            break;
        const int  pt = uni(ptab[q]);
        const int  pj = pt & 0xffff, pk = pt >> 16;
        // (nodep: diagnostic runs without the hand-offs -- wrong results, the time of the streams alone)
        const bool has_yin = pj > 0 && !nodep, has_zin = pk > 0 && !nodep;
        const bool has_out = (pj + 1 < g.nj || pk + 1 < g.nk) && !nodep;
        const int  sj = sj0, sm = sm0;
        const int  sy  = pj * kLJ + sj; // sweep y of the staging line
        const bool yok = sy < g.ny;
        const int  gy  = LOWER ? sy : g.ny - 1 - sy;
        // first element of the staging line of instruction qq in the natural-order vectors; -1: the lattice has no such line
        int lineoff[kLK];
#pragma unroll
        for(int qq = 0; qq < kLK; ++qq)
        {
            const int sz = pk * kLK + qq;
            const int gz = LOWER ? sz : g.nz - 1 - sz;
            lineoff[qq]  = (yok && sz < g.nz) ? (gz * g.ny + gy) * g.nx : -1;
        }
        // face records: [pencil][batch][64]; the y half comes from pencil (pj - 1, pk), the z half from (pj, pk - 1)
        const bool fvalid = lane < 32 ? has_yin : has_zin;
        const int  fpen   = lane < 32 ? pk * g.nj + (pj > 0 ? pj - 1 : 0) : (pk > 0 ? pk - 1 : 0) * g.nj + pj;
        const T*   fin_p  = face + (size_t)fpen * g.nfb * 64 + lane;
        T*         fout_p = face + (size_t)(pk * g.nj + pj) * g.nfb * 64 + lane;
        const V2*  cpen   = coef + (size_t)q * npair * NC * 64;

        V2 rin[kLK]; // right-hand side of the next block (memory order)
        T  fin[kPoll / 4]; // polled face batches
        V2 cq[D][NC];
        T  vprev = (T)0;

        auto load_rhs = [&](int b) __attribute__((always_inline)) { // block b -> rin
            const int i0 = kXB * b + 2 * sm;
#pragma unroll
            for(int qq = 0; qq < kLK; ++qq)
            {
                const bool lv = lineoff[qq] >= 0;
                if(A16)
                {
                    const bool ok = lv && (unsigned)i0 < (unsigned)g.nx;
                    const T*   p  = in + ((size_t)lineoff[qq] + (LOWER ? i0 : g.nx - 2 - i0));
                    rin[qq]       = lat_ld_pair(ok ? p : zsrc);
                }
                else
                {
                    const bool ok0 = lv && (unsigned)i0 < (unsigned)g.nx, ok1 = lv && (unsigned)(i0 + 1) < (unsigned)g.nx;
                    const T*   p0  = in + ((size_t)lineoff[qq] + (LOWER ? i0 : g.nx - 1 - i0));
                    const T*   p1  = in + ((size_t)lineoff[qq] + (LOWER ? i0 + 1 : g.nx - 2 - i0));
                    rin[qq].x      = lat_ld_elem(ok0 ? p0 : zsrc);
                    rin[qq].y      = lat_ld_elem(ok1 ? p1 : zsrc);
                }
            }
        };
        auto poll_face = [&](auto slot, int c) __attribute__((always_inline)) { // face batch c -> fin[slot]
            const bool ok              = fvalid && (unsigned)c < (unsigned)g.nfb;
            fin[decltype(slot)::value] = lat_ld_elem_sc1(ok ? fin_p + (size_t)c * 64 : zsrc);
        };
        auto load_coef = [&](auto slot, int pair) __attribute__((always_inline)) {
            const int pp = pair < npair ? pair : npair - 1;
            const V2* sb = cpen + (size_t)pp * NC * 64;
            lat_for<0, NC>([&](auto c) __attribute__((always_inline)) {
                cq[decltype(slot)::value][decltype(c)::value] = lat_ld_coef<decltype(c)::value * 64 * (int)sizeof(V2)>(sb, cvoff);
            });
        };

        // ---- prologue: the first block's loads go out, the ring is zeroed under them, then ONE drain per pencil
        load_rhs(0);
        lat_for<0, kPoll / 4>([&](auto c) __attribute__((always_inline)) { poll_face(c, decltype(c)::value); });
        lat_for<0, D>([&](auto d) __attribute__((always_inline)) { load_coef(d, decltype(d)::value); });
        for(int e = lane; e < kLdsElems; e += 64)
            ring[e] = (T)0; // rows the lattice does not have read +0 from the ring (fill of the skew, column "-1")
        lat_wait<0>();

        for(int body = 0; body < g.nbody; ++body)
        {
            lat_for<0, kRing>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value; // column of this step
                if constexpr(u % kXB == 0)
                {
                    constexpr int half = u / kXB;
                    const int     ph   = 2 * body + half; // the block whose steps come next
                    // ---- the loads of the previous boundary (right-hand side of block ph) have arrived
                    lat_wait<Sched::younger(u - kXB, 1, u, 0)>();
#pragma unroll
                    for(int qq = 0; qq < kLK; ++qq)
                        lat_tie(rin[qq]);
                    // (the staging addresses are recomputed at every boundary from these: hoisted out of the pencil loop they
                    //  would cost ~100 registers, and a spill is a vector memory operation the counts do not know)
                    int sj = sj0, sm = sm0;
                    asm volatile("" : "+v"(sj), "+v"(sm));
                    const int cl = 2 * sm + sj;
#pragma unroll
                    for(int qq = 0; qq < kLK; ++qq)
                        asm volatile("" : "+v"(lineoff[qq]));
                    // ---- results of block ph - 2 leave the ring (every lane is past it since step 14 of block ph - 1)
                    {
                        const int  i0 = kXB * (ph - 2) + 2 * sm;
                        const bool x0 = (unsigned)i0 < (unsigned)g.nx, x1 = (unsigned)(i0 + 1) < (unsigned)g.nx;
#pragma unroll
                        for(int qq = 0; qq < kLK; ++qq)
                        {
                            const int  c0 = (cl + 16 * half + qq) & 31, c1 = (cl + 16 * half + qq + 1) & 31;
                            const T    r0 = ring[(qq * kLJ + sj) * kPitch + 1 + c0];
                            const T    r1 = ring[(qq * kLJ + sj) * kPitch + 1 + c1];
                            const bool lv = lineoff[qq] >= 0;
                            if(A16)
                            {
                                T* p = out + ((size_t)lineoff[qq] + (LOWER ? i0 : g.nx - 2 - i0));
                                lat_st_pair_nt((lv && x0) ? p : dump, LOWER ? V2{r0, r1} : V2{r1, r0});
                            }
                            else
                            {
                                T* p0 = out + ((size_t)lineoff[qq] + (LOWER ? i0 : g.nx - 1 - i0));
                                T* p1 = out + ((size_t)lineoff[qq] + (LOWER ? i0 + 1 : g.nx - 2 - i0));
                                lat_st_elem_nt((lv && x0) ? p0 : dump, r0);
                                lat_st_elem_nt((lv && x1) ? p1 : dump + 1, r1);
                            }
                        }
                    }
                    // ---- the right-hand side of block ph enters; then the requests of block ph + 1
#pragma unroll
                    for(int qq = 0; qq < kLK; ++qq)
                    {
                        const int c0 = (cl + 16 * half + qq) & 31, c1 = (cl + 16 * half + qq + 1) & 31;
                        const V2  v  = (LOWER || !A16) ? rin[qq] : V2{rin[qq].y, rin[qq].x};
                        ring[(qq * kLJ + sj) * kPitch + 1 + c0] = v.x;
                        ring[(qq * kLJ + sj) * kPitch + 1 + c1] = v.y;
                    }
                    load_rhs(ph + 1);
                }
                if constexpr(u % 4 == 3)
                {
                    // ---- face batch (t - 11) / 4 is complete: its last column was written by lanes (7, .) / (., 7) at step u - 1
                    const int  c  = 8 * body + (u - 11) / 4; // (u - 11 is a multiple of 4)
                    const T    v  = ring[a_fout + (u - 4)];
                    const bool ok = has_out && (unsigned)c < (unsigned)g.nfb;
                    lat_st_elem_sc1(ok ? fout_p + (size_t)c * 64 : dump, v);
                }
                if constexpr(u % 4 == 0)
                {
                    // ---- face batch t / 4 of the predecessors, polled kPoll steps ago (and again until it is there)
                    constexpr int slot = (u / 4) % (kPoll / 4);
                    const int     c    = 8 * body + u / 4;
                    lat_wait<Sched::younger(u - kPoll, 3, u, 3)>();
                    lat_tie(fin[slot]);
                    if(__any(LatT<T>::to_bits(fin[slot]) == sent))
                    {
                        int spins = 0, backoff = 1;
                        do
                        {
                            spin_guard(spins);
                            backoff = poll_backoff(false, backoff, 4);
                            poll_face(std::integral_constant<int, slot>{}, c);
                            lat_wait<0>();
                            lat_tie(fin[slot]);
                        } while(__any(LatT<T>::to_bits(fin[slot]) == sent));
                    }
                    ring[a_fin + u] = fin[slot];
                    const bool ok   = fvalid && (unsigned)c < (unsigned)g.nfb;
                    lat_st_elem_sc1(ok ? const_cast<T*>(fin_p) + (size_t)c * 64 : dump, LatT<T>::from_bits(sent)); // the sentinel goes back behind the read
                    poll_face(std::integral_constant<int, slot>{}, c + kPoll / 4);
                }
                constexpr int p = u >> 1, e = u & 1; // step pair inside the body, element of the pair
                constexpr int slot = p % D;
                if constexpr(e == 0)
                {
                    // this pair's coefficients were requested D pairs ago
                    lat_wait<Sched::younger(u - 2 * D + 1, 4, u, 4)>();
                    lat_for<0, NC>([&](auto c) __attribute__((always_inline)) { lat_tie(cq[slot][decltype(c)::value]); });
                }
                const T b  = ring[a_own + u];
                const T vz = ring[a_zn + u];
                const T vy = ring[a_yn + u];
                T       acc;
                if(LOWER)
                {
                    acc = b - (e ? cq[slot][0].y : cq[slot][0].x) * vz;
                    acc = acc - (e ? cq[slot][1].y : cq[slot][1].x) * vy;
                    acc = acc - (e ? cq[slot][2].y : cq[slot][2].x) * vprev;
                }
                else
                {
                    acc = b - (e ? cq[slot][0].y : cq[slot][0].x) * vprev;
                    acc = acc - (e ? cq[slot][1].y : cq[slot][1].x) * vy;
                    acc = acc - (e ? cq[slot][2].y : cq[slot][2].x) * vz;
                }
                if(DIV)
                    acc = acc / (e ? cq[slot][NC - 1].y : cq[slot][NC - 1].x);
                vprev           = acc;
                ring[a_own + u] = acc;
                if(u == kRing - 1)
                    ring[a_own - 1] = acc;
                if constexpr(e == 1)
                {
                    load_coef(std::integral_constant<int, slot>{}, body * 16 + p + D);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        lat_wait<0>(); // (the registers of the last requests are reused by the next pencil)
    }
}

} // namespace

// ---------------------------------------------------------------- host side
struct LatPlan
{
    LatDims   g{};
    int       npair = 0, nc = 0;
    bool      lower = true, unit = true;
    int       dtype = RAMD_F64;
    int*      ptab  = nullptr;
    void*     coef  = nullptr;
    void*     face  = nullptr;
    void*     zeros = nullptr; // what lanes / phases without data read
    void*     dump  = nullptr; // ... and write: 128 elements per workgroup
    int       dump_wgs = 0;
    unsigned* counter = nullptr;
    unsigned  ticket  = 0;
    size_t    coef_bytes = 0, face_bytes = 0;
};

void lat_release(LatPlan** pp)
{
    LatPlan* P = pp ? *pp : nullptr;
    if(!P)
        return;
    dev_free(&P->ptab);
    if(P->coef)
        (void)cached_free(P->coef);
    if(P->face)
        (void)cached_free(P->face);
    if(P->zeros)
        (void)cached_free(P->zeros);
    if(P->dump)
        (void)cached_free(P->dump);
    dev_free(&P->counter);
    delete P;
    *pp = nullptr;
}
bool lat_is_unit(const LatPlan* P)
{
    return P->unit;
}
void lat_info(const LatPlan* P, LatInfo* info)
{
    *info = LatInfo{P->g.nx, P->g.ny, P->g.nz, P->g.npencil, 32 * P->g.nbody, P->coef_bytes, P->face_bytes};
}

static int lat_mode()
{
    // 0: off; 1: lattices the 8 x 8 cross-section fits (default); 2: every recognised lattice (tests)
    // (read at every analysis: the tests switch it inside one process)
    return getenv("RAMD_TRSV_LAT") ? atoi(getenv("RAMD_TRSV_LAT")) : 1;
}

template <typename T>
int lat_build(const ramd_mat_s* m, bool lower, bool unit, LatPlan** out)
{
    *out = nullptr;
    const int mode = lat_mode();
    const int n    = m->nrow;
    if(mode == 0 || m->format != RAMD_CSR || n != m->ncol || n < 8)
        return RAMD_ERR_UNSUPPORTED;
    if(mode == 1 && n < 4096)
        return RAMD_ERR_UNSUPPORTED;
    Backend& b = backend();
    int*     d = nullptr; // [0] min offset, [1] flag
    RAMD_TRY(dev_alloc(&d, 4));
    auto fail = [&](int code) {
        dev_free(&d);
        return code;
    };
    int        h[2];
    const int  grid = ew_grid(n);
    auto       pass = [&](int thr) -> int {
        h[0] = 0x7fffffff;
        h[1] = 0;
        if(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, b.cur) != hipSuccess)
            return RAMD_ERR_HIP;
        if(lower)
            hipLaunchKernelGGL((k_lat_min_offset<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, thr, d, d + 1);
        else
            hipLaunchKernelGGL((k_lat_min_offset<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, thr, d, d + 1);
        if(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, b.cur) != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess)
            return RAMD_ERR_HIP;
        return RAMD_OK;
    };
    if(pass(1) != RAMD_OK)
        return fail(RAMD_ERR_HIP);
    if(h[1] || h[0] == 0x7fffffff)
        return fail(RAMD_ERR_UNSUPPORTED);
    const int nx = h[0];
    if(nx < 2 || n % nx != 0)
        return fail(RAMD_ERR_UNSUPPORTED);
    if(pass(nx) != RAMD_OK)
        return fail(RAMD_ERR_HIP);
    const int64_t nxny = h[0] == 0x7fffffff ? n : h[0];
    if(nxny % nx != 0 || n % nxny != 0 || nxny / nx < 2)
        return fail(RAMD_ERR_UNSUPPORTED);
    const int ny = (int)(nxny / nx), nz = (int)(n / nxny);
    if(mode == 1 && (ny < 4 || nz < 4))
        return fail(RAMD_ERR_UNSUPPORTED);
    h[0] = h[1] = 0;
    if(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    if(lower)
        hipLaunchKernelGGL((k_lat_verify<true>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, nx, ny, nz, unit ? 0 : 1, d + 1);
    else
        hipLaunchKernelGGL((k_lat_verify<false>), dim3(grid), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, nx, ny, nz, unit ? 0 : 1, d + 1);
    if(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, b.cur) != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    dev_free(&d);
    if(h[1])
        return RAMD_ERR_UNSUPPORTED;

    LatPlan* P = new LatPlan;
    P->lower   = lower;
    P->unit    = unit;
    P->dtype   = m->dtype;
    P->nc      = unit ? 3 : 4;
    LatDims& g = P->g;
    g.nx = nx, g.ny = ny, g.nz = nz;
    g.nj      = (ny + kLJ - 1) / kLJ;
    g.nk      = (nz + kLK - 1) / kLK;
    g.npencil = g.nj * g.nk;
    g.nsb     = (nx + kXB - 1) / kXB;
    g.nbody   = (g.nsb + 2 + 1) / 2;
    g.nfb     = (nx + 6) / 4 + 1;
    P->npair  = 16 * g.nbody;
    if(g.nj > 0xffff || g.nk > 0x7fff)
    {
        delete P;
        return RAMD_ERR_UNSUPPORTED;
    }
    auto bail = [&](int code) {
        lat_release(&P);
        return code;
    };
    // pencils in ticket order: by J + K, so that a pencil only waits for pencils with lower tickets
    std::vector<int> tab((size_t)g.npencil);
    {
        size_t at = 0;
        for(int s = 0; s <= g.nj + g.nk - 2; ++s)
            for(int k = std::max(0, s - (g.nj - 1)); k <= std::min(s, g.nk - 1); ++k)
                tab[at++] = (s - k) | (k << 16);
    }
    if(dev_alloc(&P->ptab, g.npencil) != RAMD_OK)
        return bail(RAMD_ERR_HIP);
    if(hipMemcpyAsync(P->ptab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, b.cur) != hipSuccess
       || hipStreamSynchronize(b.cur) != hipSuccess)
        return bail(RAMD_ERR_HIP);
    P->coef_bytes = (size_t)g.npencil * P->npair * P->nc * 128 * sizeof(T);
    P->face_bytes = (size_t)g.npencil * g.nfb * 64 * sizeof(T);
    if(cached_malloc(&P->coef, P->coef_bytes + kPad) != hipSuccess || cached_malloc(&P->face, P->face_bytes + kPad) != hipSuccess)
        return bail(RAMD_ERR_HIP);
    if(dev_alloc(&P->counter, 4) != RAMD_OK || hipMemsetAsync(P->counter, 0, 16, b.cur) != hipSuccess)
        return bail(RAMD_ERR_HIP);
    P->dump_wgs = 16 * b.num_cu;
    if(cached_malloc(&P->zeros, 128 * sizeof(T) + kPad) != hipSuccess
       || cached_malloc(&P->dump, (size_t)P->dump_wgs * 128 * sizeof(T) + kPad) != hipSuccess
       || hipMemsetAsync(P->zeros, 0, 128 * sizeof(T), b.cur) != hipSuccess
       || hipMemsetAsync(P->dump, 0, (size_t)P->dump_wgs * 128 * sizeof(T), b.cur) != hipSuccess)
        return bail(RAMD_ERR_HIP);
    const int64_t total = (int64_t)g.npencil * P->npair * 128;
    if(lower)
        hipLaunchKernelGGL((k_lat_fill<T, true>), dim3(ew_grid(total)), dim3(kBlock), 0, b.cur, g, P->npair, P->nc, P->ptab, m->rp,
                           m->ci, (const T*)m->val, (T*)P->coef);
    else
        hipLaunchKernelGGL((k_lat_fill<T, false>), dim3(ew_grid(total)), dim3(kBlock), 0, b.cur, g, P->npair, P->nc, P->ptab, m->rp,
                           m->ci, (const T*)m->val, (T*)P->coef);
    const int64_t nface = (int64_t)(P->face_bytes / sizeof(T));
    hipLaunchKernelGGL((k_lat_fill_sentinel<T>), dim3(ew_grid(nface)), dim3(kBlock), 0, b.cur, nface, (T*)P->face);
    if(hipGetLastError() != hipSuccess)
        return bail(RAMD_ERR_HIP);
    build_mark("lattice plan: coefficients in sweep order, faces");
    static const bool verbose = getenv("RAMD_TRSV_CT_VERBOSE") != nullptr;
    if(verbose)
        fprintf(stderr, "lattice plan (%s): %d x %d x %d, %d pencils of %d steps, coefficients %.1f MB, faces %.1f MB\n",
                lower ? "lower" : "upper", nx, ny, nz, g.npencil, 32 * g.nbody, P->coef_bytes / 1e6, P->face_bytes / 1e6);
    *out = P;
    return RAMD_OK;
}

template <typename T>
int lat_run(LatPlan* P, const T* in, T* out)
{
    Backend&      b   = backend();
    const LatDims g   = P->g;
    const size_t  lds = (size_t)kLdsElems * sizeof(T);
    const bool    a16 = (g.nx % 2 == 0) && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % (2 * sizeof(T)) == 0);
    using V2          = typename LatT<T>::V2;
    unsigned nwg      = 0;
    static const int nodep = getenv("RAMD_LAT_NODEP") ? atoi(getenv("RAMD_LAT_NODEP")) : 0; // (tools/: timing without hand-offs)
#define LAT_GO(LO, NCV, A, DV, WV)                                                                                           \
    do                                                                                                                     \
    {                                                                                                                      \
        static int occ = 0;                                                                                                \
        if(occ == 0)                                                                                                       \
        {                                                                                                                  \
            int nb_cu = 0;                                                                                                 \
            RAMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_cu, k_trsv_lat<T, LO, NCV, A, DV, WV>, 64, lds));         \
            occ = nb_cu < 1 ? 1 : nb_cu;                                                                                   \
            /* two pencils per CU: measured at 512^3 (LUSolve, ms) 1 / 2 / 3 / 4 / 5 / 7 per CU: 2.74 / 2.38 / 2.49 / 2.53 / 2.65 / \
               2.78 -- 512 waves with their queues move what the memory system delivers, and every further wave only slows the   \
               steps of the pencils the others wait for (256^3: 0.57 / 0.56 / 0.60 / 0.65 / 0.65; gpurun_out/r05d.log) */        \
            static const int occ_env = getenv("RAMD_LAT_WGS_PER_CU") ? atoi(getenv("RAMD_LAT_WGS_PER_CU")) : 2;            \
            if(occ_env > 0 && occ_env < occ)                                                                               \
                occ = occ_env;                                                                                             \
        }                                                                                                                  \
        int64_t cap = (int64_t)occ * b.num_cu;                                                                             \
        cap         = cap < P->dump_wgs ? cap : P->dump_wgs;                                                               \
        nwg         = (unsigned)(g.npencil < cap ? g.npencil : cap);                                                       \
        hipLaunchKernelGGL((k_trsv_lat<T, LO, NCV, A, DV, WV>), dim3(nwg), dim3(64), lds, b.cur, g, P->npair, P->ptab,          \
                           (const V2*)P->coef, in, out, (T*)P->face, (const T*)P->zeros, (T*)P->dump, P->counter,          \
                           P->ticket, nodep);                                                                              \
    } while(0)
#define LAT_GO_A(LO, NCV)           \
    do                              \
    {                               \
        if(a16)                     \
            LAT_GO(LO, NCV, true, 4, 2);  \
        else                        \
            LAT_GO(LO, NCV, false, 4, 2); \
    } while(0)
    prof_begin(RAMD_PROF_TRSV, b.cur);
    if(P->lower)
    {
        if(P->unit)
            LAT_GO_A(true, 3);
        else
            LAT_GO_A(true, 4);
    }
    else
    {
        if(P->unit)
            LAT_GO_A(false, 3);
        else
            LAT_GO_A(false, 4);
    }
    prof_end(RAMD_PROF_TRSV, b.cur);
#undef LAT_GO_A
#undef LAT_GO
    P->ticket += (unsigned)g.npencil + nwg;
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

template int lat_build<double>(const ramd_mat_s*, bool, bool, LatPlan**);
template int lat_build<float>(const ramd_mat_s*, bool, bool, LatPlan**);
template int lat_run<double>(LatPlan*, const double*, double*);
template int lat_run<float>(LatPlan*, const float*, float*);

} // namespace ramd
