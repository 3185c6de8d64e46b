// fused.hip -- single-launch fusions of the Krylov BLAS-1 sequences (new entry points; the
// reference runs each update / dot / norm as its own rocBLAS call or kernel with a blocking
// device->host scalar read after every reduction, src/base/hip/hip_vector.cpp:569-931).
//
// Scalars (rho, p.q, ||r||^2, ...) stay in a device record; kernels read their coefficients from
// it, so a whole CG iteration needs no host round trip except the convergence test.
// Per element the arithmetic is exactly the reference's host expression for the fused ops
// (host_vector.cpp AddScale :635, ScaleAdd :654, PointWiseMult :1257), so vectors are bit-identical
// to the unfused sequence; only the summation order of the reductions differs.
#include "device_utils.hpp"
#include "matrix_impl.hpp"

namespace ramd
{

// Streaming shape of every kernel here (tools/membench.hip, profiles/r02_membench.txt, 512^3 fp64 vectors):
//   * kStreamU independent 16-byte packets per operand and thread are loaded before the first use (one packet per
//     turn: 4.6-4.9 TB/s on the 3-reads-2-writes kernels, four: 5.4-5.7 TB/s);
//   * operands read once / written once go through non-temporal loads / stores;
//   * kernels without a reduction launch ONE SHOT grids (a workgroup per 256 * kStreamU packets, no grid-stride loop:
//     copy 6.7 TB/s against 4.8-5.9 TB/s with any resident grid); kernels with a reduction keep a grid-stride loop over
//     at most kReduceBlocks workgroups (their partial sums live in a fixed table).
constexpr int kStreamU = 4;

#define RAMD_STREAM_LOOP(np_)                                                   \
    const int64_t stream_stride_ = (int64_t)gridDim.x * kBlock * kStreamU;      \
    for(int64_t stream_base_ = (int64_t)blockIdx.x * kBlock * kStreamU + threadIdx.x; stream_base_ < (np_); \
        stream_base_ += stream_stride_)
#define RAMD_STREAM_EACH(np_, i_)                                               \
    _Pragma("unroll") for(int u = 0; u < kStreamU; ++u)                         \
        for(int64_t i_ = stream_base_ + (int64_t)u * kBlock, once_ = 1; once_ && i_ < (np_); once_ = 0)

// (a compile-time switch: with a run-time flag the compiler once merged both branches into one plain store)
template <bool NTS_, typename P_>
__device__ __forceinline__ void st_pack(P_* p, P_ v)
{
    if constexpr(NTS_)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// CG residual update (src/solvers/krylov/cg.cpp:418-438):
//   alpha = rho / (p.q) ; r = r + (-alpha)*q ; rr = <r,r>
//   PRECOND: z = dinv * r ; rz = <r,z>        else rz = rr
template <typename T, bool PRECOND, bool NTS>
__global__ __launch_bounds__(kBlock) void k_cg_update(int64_t n, T* __restrict__ r,
                                                      const T* __restrict__ q,
                                                      const T* __restrict__ dinv, T* __restrict__ z,
                                                      ReduceCtx ctx, int slot_rho, int slot_pq,
                                                      int slot_rr, int slot_rz)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    __shared__ double lds[12];
    const T alpha  = (T)ctx.scalars[slot_rho] / (T)ctx.scalars[slot_pq];
    const T malpha = -alpha;
    int64_t np     = n / NP;
    int64_t gtid   = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t gsz    = (int64_t)gridDim.x * blockDim.x;
    double  rr = 0.0, rz = 0.0;
    RAMD_STREAM_LOOP(np)
    {
        P pr[kStreamU], pq[kStreamU], pd[kStreamU];
        RAMD_STREAM_EACH(np, i)
        {
            pr[u] = reinterpret_cast<P*>(r)[i];
            pq[u] = nt_load(reinterpret_cast<const P*>(q) + i);
            if(PRECOND)
                pd[u] = nt_load(reinterpret_cast<const P*>(dinv) + i);
        }
        RAMD_STREAM_EACH(np, i)
        {
            P pz;
#pragma unroll
            for(int k = 0; k < NP; ++k)
            {
                T rn                  = pk_elems<T>(pr[u])[k] + malpha * pk_elems<T>(pq[u])[k];
                pk_elems<T>(pr[u])[k] = rn;
                rr += (double)rn * (double)rn;
                if(PRECOND)
                {
                    T zn               = pk_elems<T>(pd[u])[k] * rn;
                    pk_elems<T>(pz)[k] = zn;
                    rz += (double)rn * (double)zn;
                }
            }
            st_pack<NTS>(reinterpret_cast<P*>(r) + i, pr[u]);
            if(PRECOND)
                st_pack<NTS>(reinterpret_cast<P*>(z) + i, pz);
        }
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
    {
        T rn = r[i] + malpha * q[i];
        r[i] = rn;
        rr += (double)rn * (double)rn;
        if(PRECOND)
        {
            T zn = dinv[i] * rn;
            z[i] = zn;
            rz += (double)rn * (double)zn;
        }
    }
    if(!PRECOND)
        rz = rr;
    const double vals[2]  = {rr, rz};
    const int    slots[2] = {slot_rr, slot_rz};
    const int    ops[2]   = {RED_SUM, RED_SUM};
    grid_reduce_finish<2>(ctx, vals, slots, ops, lds);
}

// x = x + alpha*p (old p) ; p = beta*p + z       (cg.cpp:421 AddScale, :441-442 ScaleAdd)
template <typename T, bool NTS>
__global__ __launch_bounds__(kBlock) void k_cg_direction(int64_t n, T* __restrict__ x, T* __restrict__ p,
                                                         const T* __restrict__ z,
                                                         const double* __restrict__ scalars,
                                                         int slot_rho, int slot_pq, int slot_new)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    const T       alpha = (T)scalars[slot_rho] / (T)scalars[slot_pq];
    const T       beta  = (T)scalars[slot_new] / (T)scalars[slot_rho];
    int64_t       np    = n / NP;
    int64_t       gtid  = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t       gsz   = (int64_t)gridDim.x * blockDim.x;
    RAMD_STREAM_LOOP(np)
    {
        P px[kStreamU], pp[kStreamU], pz[kStreamU];
        RAMD_STREAM_EACH(np, i)
        {
            px[u] = nt_load(reinterpret_cast<const P*>(x) + i);
            pp[u] = reinterpret_cast<P*>(p)[i];
            pz[u] = nt_load(reinterpret_cast<const P*>(z) + i);
        }
        RAMD_STREAM_EACH(np, i)
        {
#pragma unroll
            for(int k = 0; k < NP; ++k)
            {
                pk_elems<T>(px[u])[k] = pk_elems<T>(px[u])[k] + alpha * pk_elems<T>(pp[u])[k];
                pk_elems<T>(pp[u])[k] = beta * pk_elems<T>(pp[u])[k] + pk_elems<T>(pz[u])[k];
            }
            st_pack<NTS>(reinterpret_cast<P*>(x) + i, px[u]);
            st_pack<NTS>(reinterpret_cast<P*>(p) + i, pp[u]);
        }
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
    {
        x[i] = x[i] + alpha * p[i];
        p[i] = beta * p[i] + z[i];
    }
}

// up to 8 dots against one vector in one pass over w: s[slot0+k] = <v_k, w>
constexpr int kMaxMultiDot = 8;
template <typename T>
struct MultiDotArgs
{
    const T* v[kMaxMultiDot];
};

template <typename T, int NV>
__global__ __launch_bounds__(kBlock) void k_multi_dot(int64_t n, MultiDotArgs<T> a,
                                                      const T* __restrict__ w, ReduceCtx ctx, int slot0)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    __shared__ double lds[4 * NV + 4];
    int64_t np   = n / NP;
    int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    double  acc[NV];
#pragma unroll
    for(int j = 0; j < NV; ++j)
        acc[j] = 0.0;
    RAMD_STREAM_LOOP(np)
    {
        P pw[kStreamU];
        RAMD_STREAM_EACH(np, i)
            pw[u] = reinterpret_cast<const P*>(w)[i];
#pragma unroll
        for(int j = 0; j < NV; ++j)
        {
            P pv[kStreamU];
            RAMD_STREAM_EACH(np, i)
                pv[u] = nt_load(reinterpret_cast<const P*>(a.v[j]) + i);
            RAMD_STREAM_EACH(np, i)
            {
#pragma unroll
                for(int k = 0; k < NP; ++k)
                    acc[j] += (double)pk_elems<T>(pv[u])[k] * (double)pk_elems<T>(pw[u])[k];
            }
        }
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
#pragma unroll
        for(int j = 0; j < NV; ++j)
            acc[j] += (double)a.v[j][i] * (double)w[i];
    int slots[NV], ops[NV];
#pragma unroll
    for(int j = 0; j < NV; ++j)
    {
        slots[j] = slot0 + j;
        ops[j]   = RED_SUM;
    }
    grid_reduce_finish<NV>(ctx, acc, slots, ops, lds);
}

// x = x + c_0 v_0 ; x = x + c_1 v_1 ; ... in THIS order per element (the solution update of a GMRES cycle,
// gmres.cpp:522-532: one AddScale per basis vector), up to kMaxMultiDot vectors per launch: x is read and written once
// instead of once per vector (24 n bytes per AddScale -> 8 n + 16 n / count)
template <typename T>
struct MultiAxpyArgs
{
    const T* v[kMaxMultiDot];
    T        c[kMaxMultiDot];
};
template <typename T, int NV>
__global__ __launch_bounds__(kBlock) void k_multi_axpy(int64_t n, T* __restrict__ x, MultiAxpyArgs<T> a)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    int64_t       np   = n / NP;
    int64_t       gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t       gsz  = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = gtid; i < np; i += gsz) // (one-shot grid: one turn)
    {
        P px = reinterpret_cast<P*>(x)[i];
        P pv[NV];
#pragma unroll
        for(int j = 0; j < NV; ++j)
            pv[j] = nt_load(reinterpret_cast<const P*>(a.v[j]) + i);
#pragma unroll
        for(int j = 0; j < NV; ++j)
#pragma unroll
            for(int k = 0; k < NP; ++k)
                pk_elems<T>(px)[k] = pk_elems<T>(px)[k] + a.c[j] * pk_elems<T>(pv[j])[k];
        reinterpret_cast<P*>(x)[i] = px;
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
    {
        T xv = x[i];
#pragma unroll
        for(int j = 0; j < NV; ++j)
            xv = xv + a.c[j] * a.v[j][i];
        x[i] = xv;
    }
}

// one modified-Gram-Schmidt step fused with the NEXT projection's dot (gmres.cpp:480-486):
//   w = w + (-h)*v ; s[slot_dot] = <u, w>   (u == nullptr: <w, w>)
// NTW: w streams (non-temporal load and store) -- vectors far beyond the caches, where keeping w's lines only evicts others
template <typename T, bool HAVE_U, bool NTW>
__global__ __launch_bounds__(kBlock) void k_mgs_step(int64_t n, T* __restrict__ w,
                                                     const T* __restrict__ v,
                                                     const T* __restrict__ u_vec, ReduceCtx ctx,
                                                     int slot_h, int slot_dot)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    __shared__ double lds[8];
    const T mh   = -(T)ctx.scalars[slot_h];
    int64_t np   = n / NP;
    int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    double  acc  = 0.0;
    RAMD_STREAM_LOOP(np)
    {
        P pw[kStreamU], pv[kStreamU], pu[kStreamU];
        RAMD_STREAM_EACH(np, i)
        {
            pw[u] = NTW ? nt_load(reinterpret_cast<const P*>(w) + i) : reinterpret_cast<P*>(w)[i];
            pv[u] = nt_load(reinterpret_cast<const P*>(v) + i);
            if(HAVE_U)
                pu[u] = nt_load(reinterpret_cast<const P*>(u_vec) + i);
        }
        RAMD_STREAM_EACH(np, i)
        {
#pragma unroll
            for(int k = 0; k < NP; ++k)
            {
                T wn                  = pk_elems<T>(pw[u])[k] + mh * pk_elems<T>(pv[u])[k];
                pk_elems<T>(pw[u])[k] = wn;
                acc += (double)(HAVE_U ? pk_elems<T>(pu[u])[k] : wn) * (double)wn;
            }
            st_pack<NTW>(reinterpret_cast<P*>(w) + i, pw[u]); // (w is read again by the next projection: cached while it fits)
        }
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
    {
        T wn = w[i] + mh * v[i];
        w[i] = wn;
        acc += (double)(HAVE_U ? u_vec[i] : wn) * (double)wn;
    }
    const double vals[1]  = {acc};
    const int    slots[1] = {slot_dot};
    const int    ops[1]   = {RED_SUM};
    grid_reduce_finish<1>(ctx, vals, slots, ops, lds);
}

// Modified Gram-Schmidt in blocks of up to kMgsBlock basis vectors (gmres.cpp:480-486, the same recurrence):
//   h_m = <v_m, w - sum_{k<m} h_k v_k> = <v_m, w> - sum_{k<m} h_k <v_k, v_m>
// so ONE pass yields the block's e_m = <v_m, w> and its Gram entries g_km = <v_k, v_m> (measured, not assumed 0), the h_m
// follow by forward substitution in the prologue of the NEXT pass, which applies  w -= h_0 v_0; w -= h_1 v_1; ...  (the
// sequential per-element arithmetic of the MGS loop) while it gathers the next block's sums: 2 + 2K vector streams per K
// projections instead of 4K.  Slots: e_c at slot_e + c, g_cd (c < d) behind them in row-major order of the strict upper
// triangle; NC == 0: s[slot_e] = <w, w> of the updated w (the norm of the new basis vector).  The sums are accumulated with
// fused multiply-adds (the 44 products per element of a block of 8 are otherwise 88 fp64 operations); the update of w is
// NOT fused: it is the AddScale expression of the host loop, rounded as there.
// Block size: measured at 512^3 (GMRES(30), iterations/s) 4: 56.5, 5: 55.2, 6: 55.9, 8: 56.3 -- longer blocks move fewer
// bytes (38.8 instead of 42.5 vector streams per Arnoldi step on average at 8) but their passes run at 4.9 instead of
// 5.3 TB/s (17 read streams at once, 164 VGPRs), which cancels the saving; 4 keeps the kernels small.
#ifndef RAMD_MGS_K
#define RAMD_MGS_K 4
#endif
constexpr int kMgsBlock = RAMD_MGS_K; // (<= 8; 36 sums per pass at 8)
template <typename T>
struct MgsBlockArgs
{
    const T* vp[kMgsBlock]; // previous block: applied
    const T* vc[kMgsBlock]; // current block: projected on
};
template <typename T, int NPV, int NC, bool NTW, int UO = 0>
__global__ __launch_bounds__(kBlock) void k_mgs_block(int64_t n, T* __restrict__ w, MgsBlockArgs<T> a, ReduceCtx ctx,
                                                      int slot_h, int slot_eprev, int slot_e)
{
    using P           = typename Pack<T>::type;
    constexpr int NPK = Pack<T>::N;
    constexpr int NG  = NC * (NC - 1) / 2;
    constexpr int NS  = NC == 0 ? 1 : NC + NG;
#ifndef RAMD_MGS_U
#define RAMD_MGS_U 4 // (packets per thread and pass; full block at 512^3: 1.71 / 1.53 / 1.53 / 1.49 ms with 1 / 2 / 3 / 4, gpurun_out/r03bk)
#endif
    constexpr int U   = UO > 0 ? UO : ((NPV + NC > 9) ? 1 : (NPV + NC > 5) ? RAMD_MGS_U : 4); // (UO: tools/ experiments)
    __shared__ double lds[4 * NS + 4];
    T mh[NPV > 0 ? NPV : 1];
    if constexpr(NPV > 0)
    {
        double h[NPV];
        int    g = slot_eprev + NPV;
#pragma unroll
        for(int m = 0; m < NPV; ++m)
            h[m] = ctx.scalars[slot_eprev + m];
#pragma unroll
        for(int k = 0; k < NPV; ++k) // column sweep of the forward substitution: row k of the triangle is contiguous
#pragma unroll
            for(int m = k + 1; m < NPV; ++m)
                h[m] -= h[k] * ctx.scalars[g++];
#pragma unroll
        for(int m = 0; m < NPV; ++m)
            mh[m] = -(T)h[m];
        if(blockIdx.x == 0 && threadIdx.x == 0)
#pragma unroll
            for(int m = 0; m < NPV; ++m)
                ctx.scalars[slot_h + m] = h[m];
    }
    double acc[NS];
#pragma unroll
    for(int j = 0; j < NS; ++j)
        acc[j] = 0.0;
    const int64_t np     = n / NPK;
    const int64_t stride = (int64_t)gridDim.x * kBlock * U;
    for(int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; base < np; base += stride)
    {
        P pw[U], pp[NPV > 0 ? NPV : 1][U], pc[NC > 0 ? NC : 1][U];
#pragma unroll
        for(int u = 0; u < U; ++u)
        {
            const int64_t i = base + (int64_t)u * kBlock;
            if(i < np)
            {
                pw[u] = (NTW || NPV == 0) ? nt_load(reinterpret_cast<const P*>(w) + i) : reinterpret_cast<P*>(w)[i];
#pragma unroll
                for(int q = 0; q < NPV; ++q)
                    pp[q][u] = nt_load(reinterpret_cast<const P*>(a.vp[q]) + i);
#pragma unroll
                for(int c = 0; c < NC; ++c)
                    pc[c][u] = nt_load(reinterpret_cast<const P*>(a.vc[c]) + i);
            }
        }
#pragma unroll
        for(int u = 0; u < U; ++u)
        {
            const int64_t i = base + (int64_t)u * kBlock;
            if(i < np)
            {
#pragma unroll
                for(int k = 0; k < NPK; ++k)
                {
                    T wn = pk_elems<T>(pw[u])[k];
#pragma unroll
                    for(int q = 0; q < NPV; ++q)
                        wn = wn + mh[q] * pk_elems<T>(pp[q][u])[k];
                    pk_elems<T>(pw[u])[k] = wn;
                    if constexpr(NC == 0)
                        acc[0] = __builtin_fma((double)wn, (double)wn, acc[0]);
                    int g = NC;
#pragma unroll
                    for(int c = 0; c < NC; ++c)
                    {
                        acc[c] = __builtin_fma((double)pk_elems<T>(pc[c][u])[k], (double)wn, acc[c]);
#pragma unroll
                        for(int d = c + 1; d < NC; ++d)
                            {
                            acc[g] = __builtin_fma((double)pk_elems<T>(pc[c][u])[k], (double)pk_elems<T>(pc[d][u])[k], acc[g]);
                            ++g;
                        }
                    }
                }
                if constexpr(NPV > 0)
                    st_pack<NTW>(reinterpret_cast<P*>(w) + i, pw[u]);
            }
        }
    }
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = np * NPK + gtid; i < n; i += gsz)
    {
        T wn = w[i];
#pragma unroll
        for(int q = 0; q < NPV; ++q)
            wn = wn + mh[q] * a.vp[q][i];
        if constexpr(NPV > 0)
            w[i] = wn;
        if constexpr(NC == 0)
            acc[0] = __builtin_fma((double)wn, (double)wn, acc[0]);
        int g = NC;
#pragma unroll
        for(int c = 0; c < NC; ++c)
        {
            acc[c] += (double)a.vc[c][i] * (double)wn;
#pragma unroll
            for(int d = c + 1; d < NC; ++d)
                acc[g++] += (double)a.vc[c][i] * (double)a.vc[d][i];
        }
    }
    int slots[NS], ops[NS];
#pragma unroll
    for(int j = 0; j < NS; ++j)
    {
        slots[j] = slot_e + j;
        ops[j]   = RED_SUM;
    }
    grid_reduce_finish<NS>(ctx, acc, slots, ops, lds);
}

// v *= 1/sqrt(s[slot_sq]); the norm itself is left in s[slot_norm]   (gmres.cpp:493-496)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_normalize(int64_t n, T* __restrict__ v,
                                                      double* __restrict__ scalars, int slot_sq,
                                                      int slot_norm)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    const T       nrm = (T)sqrt(scalars[slot_sq]);
    const T       inv = (T)1 / nrm;
    int64_t       np   = n / NP;
    int64_t       gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t       gsz  = (int64_t)gridDim.x * blockDim.x;
    RAMD_STREAM_LOOP(np)
    {
        P pv[kStreamU];
        RAMD_STREAM_EACH(np, i)
            pv[u] = reinterpret_cast<P*>(v)[i];
        RAMD_STREAM_EACH(np, i)
        {
#pragma unroll
            for(int k = 0; k < NP; ++k)
                pk_elems<T>(pv[u])[k] *= inv;
            reinterpret_cast<P*>(v)[i] = pv[u];
        }
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
        v[i] *= inv;
    if(gtid == 0)
        scalars[slot_norm] = (double)nrm;
}

// ---- BiCGStab (src/solvers/krylov/bicgstab.cpp:245-361 / :365-489), three fused updates.
// Scalars come from the device record: alpha = rho / <r0,q>, omega = <t,r> / <t,t>,
// beta = (rho_new / rho) * (alpha / omega) -- the host's expressions, evaluated in T.
//   r = r + (-alpha) q                                          (r->AddScale(*q, -alpha))
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bicg_r_update(int64_t n, T* __restrict__ r, const T* __restrict__ q,
                                                          const double* __restrict__ scalars, int slot_rho,
                                                          int slot_r0q)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    const T       malpha = -((T)scalars[slot_rho] / (T)scalars[slot_r0q]);
    int64_t       np   = n / NP;
    int64_t       gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t       gsz  = (int64_t)gridDim.x * blockDim.x;
    RAMD_STREAM_LOOP(np)
    {
        P pr[kStreamU], pq[kStreamU];
        RAMD_STREAM_EACH(np, i)
        {
            pr[u] = reinterpret_cast<P*>(r)[i];
            pq[u] = nt_load(reinterpret_cast<const P*>(q) + i);
        }
        RAMD_STREAM_EACH(np, i)
        {
#pragma unroll
            for(int k = 0; k < NP; ++k)
                pk_elems<T>(pr[u])[k] = pk_elems<T>(pr[u])[k] + malpha * pk_elems<T>(pq[u])[k];
            reinterpret_cast<P*>(r)[i] = pr[u];
        }
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
        r[i] = r[i] + malpha * q[i];
}

//   omega valid:  x = one*x + alpha*dir + omega*sv   (x->ScaleAdd2(one, dir, alpha, sv, omega))
//                 r = r + (-omega) t ; s[rr] = <r,r> ; s[rho_new] = <r0,r>
//   omega == 0 / NaN / Inf:  x = x + alpha*p only, s[flag] = 1 (the host then runs the reference's
//                 breakdown branch, bicgstab.cpp:430-447)
//   PRECOND false: dir == p and sv == r (the old r), so they are not passed separately
template <typename T, bool PRECOND>
__global__ __launch_bounds__(kBlock) void k_bicg_xr_update(int64_t n, T* __restrict__ x, const T* __restrict__ dir,
                                                           const T* __restrict__ sv, T* __restrict__ r,
                                                           const T* __restrict__ t, const T* __restrict__ r0,
                                                           const T* __restrict__ p, ReduceCtx ctx, int slot_rho,
                                                           int slot_r0q, int slot_tr, int slot_rr,
                                                           int slot_new, int slot_flag)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    __shared__ double lds[12];
    const T    alpha = (T)ctx.scalars[slot_rho] / (T)ctx.scalars[slot_r0q];
    const T    omega = (T)ctx.scalars[slot_tr] / (T)ctx.scalars[slot_tr + 1];
    const T    one   = (T)1;
    const bool bad   = (fabs((double)omega) == INFINITY) || (omega != omega) || (omega == (T)0);
    const T    mo    = -omega;
    int64_t    np    = n / NP;
    int64_t    gtid  = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t    gsz   = (int64_t)gridDim.x * blockDim.x;
    double     rr = 0.0, rn = 0.0;
    if(bad)
    {
        for(int64_t i = gtid; i < n; i += gsz)
            x[i] = x[i] + alpha * p[i];
    }
    else
    {
        RAMD_STREAM_LOOP(np)
        {
            P px[kStreamU], pr[kStreamU], pt[kStreamU], p0[kStreamU], pd[kStreamU], ps[kStreamU];
            RAMD_STREAM_EACH(np, i)
            {
                px[u] = nt_load(reinterpret_cast<const P*>(x) + i);
                pr[u] = reinterpret_cast<P*>(r)[i];
                pt[u] = nt_load(reinterpret_cast<const P*>(t) + i);
                p0[u] = reinterpret_cast<const P*>(r0)[i];
                pd[u] = PRECOND ? nt_load(reinterpret_cast<const P*>(dir) + i) : reinterpret_cast<const P*>(p)[i];
                if(PRECOND)
                    ps[u] = nt_load(reinterpret_cast<const P*>(sv) + i);
            }
            RAMD_STREAM_EACH(np, i)
            {
                if(!PRECOND)
                    ps[u] = pr[u];
#pragma unroll
                for(int k = 0; k < NP; ++k)
                {
                    pk_elems<T>(px[u])[k]
                        = one * pk_elems<T>(px[u])[k] + alpha * pk_elems<T>(pd[u])[k] + omega * pk_elems<T>(ps[u])[k];
                    const T rnew          = pk_elems<T>(pr[u])[k] + mo * pk_elems<T>(pt[u])[k];
                    pk_elems<T>(pr[u])[k] = rnew;
                    rr += (double)rnew * (double)rnew;
                    rn += (double)pk_elems<T>(p0[u])[k] * (double)rnew;
                }
                __builtin_nontemporal_store(px[u], reinterpret_cast<P*>(x) + i);
                reinterpret_cast<P*>(r)[i] = pr[u];
            }
        }
        for(int64_t i = np * NP + gtid; i < n; i += gsz)
        {
            const T dv = PRECOND ? dir[i] : p[i];
            const T s0 = PRECOND ? sv[i] : r[i];
            x[i]         = one * x[i] + alpha * dv + omega * s0;
            const T rnew = r[i] + mo * t[i];
            r[i]         = rnew;
            rr += (double)rnew * (double)rnew;
            rn += (double)r0[i] * (double)rnew;
        }
    }
    if(gtid == 0)
        ctx.scalars[slot_flag] = bad ? 1.0 : 0.0;
    const double vals[2]  = {rr, rn};
    const int    slots[2] = {slot_rr, slot_new};
    const int    ops[2]   = {RED_SUM, RED_SUM};
    grid_reduce_finish<2>(ctx, vals, slots, ops, lds);
}

//   p = beta*p + (-beta*omega)*q + one*r                 (p->ScaleAdd2(beta, *q, -beta * omega, *r, one))
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bicg_direction(int64_t n, T* __restrict__ p, const T* __restrict__ q,
                                                           const T* __restrict__ r,
                                                           const double* __restrict__ scalars, int slot_rho,
                                                           int slot_r0q, int slot_tr, int slot_new)
{
    using P          = typename Pack<T>::type;
    constexpr int NP = Pack<T>::N;
    const T       alpha = (T)scalars[slot_rho] / (T)scalars[slot_r0q];
    const T       omega = (T)scalars[slot_tr] / (T)scalars[slot_tr + 1];
    const T       beta  = ((T)scalars[slot_new] / (T)scalars[slot_rho]) * (alpha / omega);
    const T       mbo   = -beta * omega;
    const T       one   = (T)1;
    int64_t       np    = n / NP;
    int64_t       gtid  = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t       gsz   = (int64_t)gridDim.x * blockDim.x;
    RAMD_STREAM_LOOP(np)
    {
        P pp[kStreamU], pq[kStreamU], pr[kStreamU];
        RAMD_STREAM_EACH(np, i)
        {
            pp[u] = reinterpret_cast<P*>(p)[i];
            pq[u] = nt_load(reinterpret_cast<const P*>(q) + i);
            pr[u] = reinterpret_cast<const P*>(r)[i];
        }
        RAMD_STREAM_EACH(np, i)
        {
#pragma unroll
            for(int k = 0; k < NP; ++k)
                pk_elems<T>(pp[u])[k]
                    = beta * pk_elems<T>(pp[u])[k] + mbo * pk_elems<T>(pq[u])[k] + one * pk_elems<T>(pr[u])[k];
            reinterpret_cast<P*>(p)[i] = pp[u];
        }
    }
    for(int64_t i = np * NP + gtid; i < n; i += gsz)
        p[i] = beta * p[i] + mbo * q[i] + one * r[i];
}

} // namespace ramd

using namespace ramd;

// launch geometry of the streaming kernels: one workgroup per kBlock * kStreamU packets; reductions cap the grid at
// the size of the partial-sum table and stride over the rest
static int grid_oneshot(int64_t n, int dtype)
{
    const int64_t np = n / (dtype == RAMD_F64 ? 2 : 4);
    const int64_t g  = (np + (int64_t)kBlock * kStreamU - 1) / ((int64_t)kBlock * kStreamU);
    return (int)(g < 1 ? 1 : g);
}
static int grid_reduce(int64_t n, int dtype)
{
    const int g = grid_oneshot(n, dtype);
    return g > kReduceBlocks ? kReduceBlocks : g;
}

static bool slot_ok(int s)
{
    return s >= 0 && s < kScalarSlots;
}

#define CHECK_SAMEV(a, b)                                                     \
    do                                                                        \
    {                                                                         \
        if(!(a) || !(b) || (a)->dtype != (b)->dtype || (a)->n != (b)->n)      \
            RAMD_FAIL(RAMD_ERR_ARG, "fused op: vector handles/sizes/types mismatch"); \
    } while(0)

extern "C" {

int ramd_fused_apply_dot(ramd_mat_t m, ramd_vec_t x, ramd_vec_t y, int slot_dot)
{
    if(!m || !x || !y || !slot_ok(slot_dot))
        RAMD_FAIL(RAMD_ERR_ARG, "fused_apply_dot: bad arguments");
    if(x->dtype != m->dtype || y->dtype != m->dtype || x->n != m->ncol || y->n != m->nrow || x == y)
        RAMD_FAIL(RAMD_ERR_ARG, "fused_apply_dot: vector sizes/types do not match the matrix");
    int s = (m->dtype == RAMD_F64)
                ? mat_apply_dot_impl<double>(m, (const double*)x->d, (double*)y->d, slot_dot)
                : mat_apply_dot_impl<float>(m, (const float*)x->d, (float*)y->d, slot_dot);
    if(s != RAMD_ERR_UNSUPPORTED)
        return s;
    // formats without a fused epilogue: SpMV, then a one-launch dot into the same slot
    RAMD_TRY(ramd_mat_apply(m, x, y));
    const ramd_vec_t vs[1] = {x};
    return ramd_fused_multi_dot(vs, 1, y, slot_dot);
}

int ramd_fused_apply_add_dot(ramd_mat_t m, ramd_vec_t x, double scalar, ramd_vec_t y, ramd_vec_t p,
                             int slot_dot)
{
    if(!m || !x || !y || !p || !slot_ok(slot_dot))
        RAMD_FAIL(RAMD_ERR_ARG, "fused_apply_add_dot: bad arguments");
    if(x->dtype != m->dtype || y->dtype != m->dtype || p->dtype != m->dtype || x->n != m->ncol
       || y->n != m->nrow || p->n != y->n || x == y)
        RAMD_FAIL(RAMD_ERR_ARG, "fused_apply_add_dot: vector sizes/types do not match the matrix");
    if(m->nnz <= 0)
        return RAMD_OK; // ApplyAdd with an empty matrix does nothing; the dot stays valid
    int s = (m->dtype == RAMD_F64)
                ? mat_apply_add_dot_impl<double>(m, (const double*)x->d, (double*)y->d, scalar,
                                                 (const double*)p->d, slot_dot)
                : mat_apply_add_dot_impl<float>(m, (const float*)x->d, (float*)y->d, (float)scalar,
                                                (const float*)p->d, slot_dot);
    if(s != RAMD_ERR_UNSUPPORTED)
        return s;
    // other formats: ApplyAdd, then the dot once more over the whole vectors
    RAMD_TRY(ramd_mat_apply_add(m, x, scalar, y));
    const ramd_vec_t vs[1] = {p};
    return ramd_fused_multi_dot(vs, 1, y, slot_dot);
}

int ramd_fused_apply_dotv(ramd_mat_t m, ramd_vec_t x, ramd_vec_t y, ramd_vec_t w, int slot_dot)
{
    if(!m || !x || !y || !w || !slot_ok(slot_dot))
        RAMD_FAIL(RAMD_ERR_ARG, "fused_apply_dotv: bad arguments");
    if(x->dtype != m->dtype || y->dtype != m->dtype || w->dtype != m->dtype || x->n != m->ncol || y->n != m->nrow
       || w->n != m->nrow || x == y || w == y)
        RAMD_FAIL(RAMD_ERR_ARG, "fused_apply_dotv: vector sizes/types do not match the matrix");
    int s = (m->dtype == RAMD_F64) ? mat_apply_dot_impl<double>(m, (const double*)x->d, (double*)y->d, slot_dot,
                                                                (const double*)w->d)
                                   : mat_apply_dot_impl<float>(m, (const float*)x->d, (float*)y->d, slot_dot,
                                                               (const float*)w->d);
    if(s != RAMD_ERR_UNSUPPORTED)
        return s;
    RAMD_TRY(ramd_mat_apply(m, x, y));
    const ramd_vec_t vs[1] = {w};
    return ramd_fused_multi_dot(vs, 1, y, slot_dot);
}

int ramd_fused_jacobi_sweep(ramd_mat_t m, ramd_vec_t dinv, ramd_vec_t rhs, ramd_vec_t x, ramd_vec_t xnew, double omega)
{
    if(!m || !dinv || !rhs || !x || !xnew || x == xnew || rhs == xnew || dinv == xnew)
        RAMD_FAIL(RAMD_ERR_ARG, "fused_jacobi_sweep: bad arguments");
    if(dinv->dtype != m->dtype || rhs->dtype != m->dtype || x->dtype != m->dtype || xnew->dtype != m->dtype
       || dinv->n != m->nrow || rhs->n != m->nrow || x->n != m->ncol || xnew->n != m->nrow)
        RAMD_FAIL(RAMD_ERR_ARG, "fused_jacobi_sweep: vector sizes/types do not match the matrix");
    if(m->dtype == RAMD_F64)
        return mat_jacobi_sweep_impl<double>(m, (const double*)dinv->d, (const double*)rhs->d, (const double*)x->d,
                                             (double*)xnew->d, omega);
    return mat_jacobi_sweep_impl<float>(m, (const float*)dinv->d, (const float*)rhs->d, (const float*)x->d,
                                        (float*)xnew->d, (float)omega);
}

int ramd_fused_bicg_r_update(ramd_vec_t r, ramd_vec_t q, int slot_rho, int slot_r0q)
{
    CHECK_SAMEV(r, q);
    if(!slot_ok(slot_rho) || !slot_ok(slot_r0q) || r == q)
        RAMD_FAIL(RAMD_ERR_ARG, "bicg_r_update: bad arguments");
    if(r->n == 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = grid_oneshot(r->n, r->dtype);
    if(r->dtype == RAMD_F64)
        hipLaunchKernelGGL((k_bicg_r_update<double>), dim3(grid), dim3(kBlock), 0, b.cur, r->n, (double*)r->d,
                           (const double*)q->d, b.d_scalars, slot_rho, slot_r0q);
    else
        hipLaunchKernelGGL((k_bicg_r_update<float>), dim3(grid), dim3(kBlock), 0, b.cur, r->n, (float*)r->d,
                           (const float*)q->d, b.d_scalars, slot_rho, slot_r0q);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_fused_bicg_xr_update(ramd_vec_t x, ramd_vec_t dir, ramd_vec_t sv, ramd_vec_t r, ramd_vec_t t,
                              ramd_vec_t r0, ramd_vec_t p, int slot_rho, int slot_r0q, int slot_tr,
                              int slot_rr, int slot_new, int slot_flag)
{
    CHECK_SAMEV(x, r);
    CHECK_SAMEV(x, t);
    CHECK_SAMEV(x, r0);
    CHECK_SAMEV(x, p);
    const bool precond = (dir != NULL) || (sv != NULL);
    if(precond)
    {
        CHECK_SAMEV(x, dir);
        CHECK_SAMEV(x, sv);
    }
    if(!slot_ok(slot_rho) || !slot_ok(slot_r0q) || !slot_ok(slot_tr) || !slot_ok(slot_tr + 1) || !slot_ok(slot_rr)
       || !slot_ok(slot_new) || !slot_ok(slot_flag))
        RAMD_FAIL(RAMD_ERR_ARG, "scalar slot out of range");
    if(x->n == 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = grid_reduce(x->n, x->dtype);
    ReduceCtx ctx  = reduce_ctx();
#define GO(T, PC)                                                                                              \
    hipLaunchKernelGGL((k_bicg_xr_update<T, PC>), dim3(grid), dim3(kBlock), 0, b.cur, x->n, (T*)x->d,          \
                       (const T*)(dir ? dir->d : NULL), (const T*)(sv ? sv->d : NULL), (T*)r->d, (const T*)t->d, \
                       (const T*)r0->d, (const T*)p->d, ctx, slot_rho, slot_r0q, slot_tr, slot_rr, slot_new,   \
                       slot_flag)
    if(x->dtype == RAMD_F64)
    {
        if(precond)
            GO(double, true);
        else
            GO(double, false);
    }
    else
    {
        if(precond)
            GO(float, true);
        else
            GO(float, false);
    }
#undef GO
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_fused_bicg_direction(ramd_vec_t p, ramd_vec_t q, ramd_vec_t r, int slot_rho, int slot_r0q, int slot_tr,
                              int slot_new)
{
    CHECK_SAMEV(p, q);
    CHECK_SAMEV(p, r);
    if(!slot_ok(slot_rho) || !slot_ok(slot_r0q) || !slot_ok(slot_tr) || !slot_ok(slot_tr + 1) || !slot_ok(slot_new))
        RAMD_FAIL(RAMD_ERR_ARG, "scalar slot out of range");
    if(p->n == 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = grid_oneshot(p->n, p->dtype);
    if(p->dtype == RAMD_F64)
        hipLaunchKernelGGL((k_bicg_direction<double>), dim3(grid), dim3(kBlock), 0, b.cur, p->n, (double*)p->d,
                           (const double*)q->d, (const double*)r->d, b.d_scalars, slot_rho, slot_r0q, slot_tr,
                           slot_new);
    else
        hipLaunchKernelGGL((k_bicg_direction<float>), dim3(grid), dim3(kBlock), 0, b.cur, p->n, (float*)p->d,
                           (const float*)q->d, (const float*)r->d, b.d_scalars, slot_rho, slot_r0q, slot_tr,
                           slot_new);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_fused_cg_update(ramd_vec_t r, ramd_vec_t q, ramd_vec_t dinv, ramd_vec_t z, int slot_rho,
                         int slot_pq, int slot_rr, int slot_rz)
{
    CHECK_SAMEV(r, q);
    if(dinv)
    {
        CHECK_SAMEV(r, dinv);
        CHECK_SAMEV(r, z);
    }
    if(!slot_ok(slot_rho) || !slot_ok(slot_pq) || !slot_ok(slot_rr) || !slot_ok(slot_rz))
        RAMD_FAIL(RAMD_ERR_ARG, "scalar slot out of range");
    if(r->n == 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = grid_reduce(r->n, r->dtype);
    ReduceCtx ctx  = reduce_ctx();
    static int nts = -1;
    if(nts < 0)
        nts = getenv("RAMD_NT_STORES") ? atoi(getenv("RAMD_NT_STORES")) : 1;
#define GO(T)                                                                                          \
    do                                                                                                 \
    {                                                                                                  \
        if(dinv && nts)                                                                                \
            hipLaunchKernelGGL((k_cg_update<T, true, true>), dim3(grid), dim3(kBlock), 0, b.cur, r->n, \
                               (T*)r->d, (const T*)q->d, (const T*)dinv->d, (T*)z->d, ctx, slot_rho,   \
                               slot_pq, slot_rr, slot_rz);                                             \
        else if(dinv)                                                                                  \
            hipLaunchKernelGGL((k_cg_update<T, true, false>), dim3(grid), dim3(kBlock), 0, b.cur, r->n, \
                               (T*)r->d, (const T*)q->d, (const T*)dinv->d, (T*)z->d, ctx, slot_rho,   \
                               slot_pq, slot_rr, slot_rz);                                             \
        else                                                                                           \
            hipLaunchKernelGGL((k_cg_update<T, false, false>), dim3(grid), dim3(kBlock), 0, b.cur, r->n, \
                               (T*)r->d, (const T*)q->d, (const T*)nullptr, (T*)nullptr, ctx,          \
                               slot_rho, slot_pq, slot_rr, slot_rz);                                   \
    } while(0)
    prof_begin(RAMD_PROF_VEC, b.cur);
    if(r->dtype == RAMD_F64)
        GO(double);
    else if(r->dtype == RAMD_F32)
        GO(float);
    else
        RAMD_FAIL(RAMD_ERR_ARG, "fused_cg_update needs real vectors");
    prof_end(RAMD_PROF_VEC, b.cur);
#undef GO
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_fused_cg_direction(ramd_vec_t x, ramd_vec_t p, ramd_vec_t z, int slot_rho, int slot_pq,
                            int slot_new)
{
    CHECK_SAMEV(x, p);
    CHECK_SAMEV(p, z);
    if(!slot_ok(slot_rho) || !slot_ok(slot_pq) || !slot_ok(slot_new))
        RAMD_FAIL(RAMD_ERR_ARG, "scalar slot out of range");
    if(p->n == 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = grid_oneshot(p->n, p->dtype);
    static int nts = -1;
    if(nts < 0)
        nts = getenv("RAMD_NT_STORES") ? atoi(getenv("RAMD_NT_STORES")) : 1;
    prof_begin(RAMD_PROF_VEC, b.cur);
    if(p->dtype == RAMD_F64 && nts)
        hipLaunchKernelGGL((k_cg_direction<double, true>), dim3(grid), dim3(kBlock), 0, b.cur, p->n,
                           (double*)x->d, (double*)p->d, (const double*)z->d, b.d_scalars, slot_rho,
                           slot_pq, slot_new);
    else if(p->dtype == RAMD_F64)
        hipLaunchKernelGGL((k_cg_direction<double, false>), dim3(grid), dim3(kBlock), 0, b.cur, p->n,
                           (double*)x->d, (double*)p->d, (const double*)z->d, b.d_scalars, slot_rho,
                           slot_pq, slot_new);
    else if(p->dtype == RAMD_F32)
        hipLaunchKernelGGL((k_cg_direction<float, false>), dim3(grid), dim3(kBlock), 0, b.cur, p->n,
                           (float*)x->d, (float*)p->d, (const float*)z->d, b.d_scalars, slot_rho, slot_pq,
                           slot_new);
    else
        RAMD_FAIL(RAMD_ERR_ARG, "fused_cg_direction needs real vectors");
    prof_end(RAMD_PROF_VEC, b.cur);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

} // extern "C"

template <typename T>
static int multi_dot_t(const ramd_vec_t* vs, int count, ramd_vec_t w, int slot0)
{
    Backend&  b    = backend();
    const int grid = grid_reduce(w->n, w->dtype);
    ReduceCtx ctx  = reduce_ctx();
    int       done = 0;
    while(done < count)
    {
        const int        nv = std::min(count - done, kMaxMultiDot);
        MultiDotArgs<T> a;
        for(int j = 0; j < kMaxMultiDot; ++j)
            a.v[j] = (const T*)vs[done + std::min(j, nv - 1)]->d;
#define GO(NV)                                                                                     \
    case NV:                                                                                       \
        hipLaunchKernelGGL((k_multi_dot<T, NV>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, a,      \
                           (const T*)w->d, ctx, slot0 + done);                                     \
        break;
        switch(nv)
        {
            GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
        }
#undef GO
        RAMD_HIP(hipGetLastError());
        done += nv;
    }
    return RAMD_OK;
}

template <typename T>
static int mgs_block_t(ramd_vec_t w, const ramd_vec_t* vprev, int nprev, int slot_h, int slot_eprev, const ramd_vec_t* vcur,
                       int ncur, int slot_e, bool ntw)
{
    Backend&        b    = backend();
    const int       grid = grid_reduce(w->n, w->dtype);
    ReduceCtx       ctx  = reduce_ctx();
    MgsBlockArgs<T> a;
    for(int j = 0; j < kMgsBlock; ++j)
    {
        a.vp[j] = nprev > 0 ? (const T*)vprev[std::min(j, nprev - 1)]->d : nullptr;
        a.vc[j] = ncur > 0 ? (const T*)vcur[std::min(j, ncur - 1)]->d : nullptr;
    }
    prof_begin(RAMD_PROF_VEC, b.cur);
    bool launched = false;
    auto go       = [&](auto npv, auto nc) {
        constexpr int NPV = decltype(npv)::value, NC = decltype(nc)::value;
        if constexpr(NPV <= kMgsBlock && NC <= kMgsBlock)
        {
            static const int uo = getenv("RAMD_MGS_UO") ? atoi(getenv("RAMD_MGS_UO")) : 0; // (packets per thread and pass of the full block, experiments)
            if constexpr(NPV == 4 && NC == 4)
            {
                if(uo == 1 || uo == 3 || uo == 4)
                {
                    if(uo == 1)
                        hipLaunchKernelGGL((k_mgs_block<T, NPV, NC, true, 1>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, (T*)w->d, a, ctx,
                                           slot_h, slot_eprev, slot_e);
                    else if(uo == 3)
                        hipLaunchKernelGGL((k_mgs_block<T, NPV, NC, true, 3>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, (T*)w->d, a, ctx,
                                           slot_h, slot_eprev, slot_e);
                    else
                        hipLaunchKernelGGL((k_mgs_block<T, NPV, NC, true, 4>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, (T*)w->d, a, ctx,
                                           slot_h, slot_eprev, slot_e);
                    launched = true;
                    return;
                }
            }
            if(ntw)
                hipLaunchKernelGGL((k_mgs_block<T, NPV, NC, true>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, (T*)w->d, a, ctx,
                                   slot_h, slot_eprev, slot_e);
            else
                hipLaunchKernelGGL((k_mgs_block<T, NPV, NC, false>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, (T*)w->d, a, ctx,
                                   slot_h, slot_eprev, slot_e);
            launched = true;
        }
    };
    using std::integral_constant;
#define MGS_EACH(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8)
    if(ncur > 0)
    {
        // first block (w only read) or a full block applied while the next one is projected on
#define M(NC)                                                                        \
    if(ncur == NC)                                                                   \
    {                                                                                \
        if(nprev == 0)                                                               \
            go(integral_constant<int, 0>{}, integral_constant<int, NC>{});           \
        else                                                                         \
            go(integral_constant<int, kMgsBlock>{}, integral_constant<int, NC>{});   \
    }
        MGS_EACH(M)
#undef M
    }
    else
    {
#define M(NP) \
    if(nprev == NP) \
        go(integral_constant<int, NP>{}, integral_constant<int, 0>{});
        MGS_EACH(M)
#undef M
    }
#undef MGS_EACH
    prof_end(RAMD_PROF_VEC, b.cur);
    if(!launched)
        RAMD_FAIL(RAMD_ERR_ARG, "fused_mgs_block: block longer than this build's block size");
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}


// ---------------------------------------------------------------- device-resident scalar algebra
struct SopList
{
    ramd_sop_t op[RAMD_SOP_MAX];
};
__global__ void k_scalar_prog(SopList pr, int count, int single, double* __restrict__ s)
{
    if(threadIdx.x != 0 || blockIdx.x != 0)
        return;
    for(int i = 0; i < count; ++i)
    {
        const ramd_sop_t o = pr.op[i];
        const double     a = o.a >= 0 ? s[o.a] : 0.0, b = o.b >= 0 ? s[o.b] : 0.0;
        double           r = 0.0;
        bool             store = true;
        switch(o.op)
        {
        case RAMD_SOP_SET: r = o.imm; break;
        case RAMD_SOP_MOV: r = a; break;
        case RAMD_SOP_ADD: r = single ? (double)((float)a + (float)b) : a + b; break;
        case RAMD_SOP_SUB: r = single ? (double)((float)a - (float)b) : a - b; break;
        case RAMD_SOP_MUL: r = single ? (double)((float)a * (float)b) : a * b; break;
        case RAMD_SOP_DIV: r = single ? (double)((float)a / (float)b) : a / b; break;
        case RAMD_SOP_NEG: r = -a; break;
        case RAMD_SOP_SQRT: r = single ? (double)sqrtf((float)a) : sqrt(a); break;
        case RAMD_SOP_ABS: r = fabs(a); break;
        case RAMD_SOP_ZFLAG:
            store = (a == 0.0);
            r     = 1.0;
            break;
        case RAMD_SOP_BADFLAG:
            store = (a == 0.0) || (a != a) || (fabs(a) == INFINITY);
            r     = 1.0;
            break;
        case RAMD_SOP_CMOVLT:
            store = a < b;
            r     = s[(int)o.imm];
            break;
        default: store = false; break;
        }
        if(store)
            s[o.dst] = single ? (double)(float)r : r;
    }
}

struct CombineArgs
{
    const void* v[3];
    int         slot[3];
    double      factor[3];
};
// x = c0 v0 + c1 v1 + c2 v2 (left to right, no contraction), one 16-byte packet per operand and thread
template <typename T, int NT>
__global__ __launch_bounds__(kBlock) void k_combine_s(int64_t n, T* __restrict__ x, CombineArgs a, int guard,
                                                      const double* __restrict__ s)
{
    using PK         = typename Pack<T>::type;
    constexpr int PN = Pack<T>::N;
    if(guard >= 0 && s[guard] != 0.0)
        return;
    T c[NT];
#pragma unroll
    for(int k = 0; k < NT; ++k)
        c[k] = (T)(a.slot[k] >= 0 ? a.factor[k] * s[a.slot[k]] : a.factor[k]);
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * PN;
    if(i >= n)
        return;
    PK p[NT];
#pragma unroll
    for(int k = 0; k < NT; ++k)
        p[k] = *reinterpret_cast<const PK*>(static_cast<const T*>(a.v[k]) + i); // (vectors are padded by 256 B)
    PK r;
#pragma unroll
    for(int e = 0; e < PN; ++e)
    {
        T acc = c[0] * p[0][e];
#pragma unroll
        for(int k = 1; k < NT; ++k)
            acc = acc + c[k] * p[k][e];
        r[e] = acc;
    }
    if(i + PN <= n)
        *reinterpret_cast<PK*>(x + i) = r;
    else
        for(int e = 0; e < PN && i + e < n; ++e)
            x[i + e] = r[e];
}

extern "C" {

int ramd_scalars_eval(const ramd_sop_t* ops, int count, int single)
{
    RAMD_TRY(ensure_init());
    if(count == 0)
        return RAMD_OK;
    if(!ops || count < 0 || count > RAMD_SOP_MAX)
        RAMD_FAIL(RAMD_ERR_ARG, "scalars_eval: bad program");
    SopList pr;
    for(int i = 0; i < count; ++i)
    {
        const ramd_sop_t& o = ops[i];
        if(o.op == RAMD_SOP_CMOVLT && (!slot_ok(o.b) || !slot_ok((int)o.imm)))
            RAMD_FAIL(RAMD_ERR_ARG, "scalars_eval: operation or slot out of range");
        if(o.op < RAMD_SOP_SET || o.op > RAMD_SOP_CMOVLT || !slot_ok(o.dst) || (o.op != RAMD_SOP_SET && !slot_ok(o.a))
           || (o.op >= RAMD_SOP_ADD && o.op <= RAMD_SOP_DIV && !slot_ok(o.b)))
            RAMD_FAIL(RAMD_ERR_ARG, "scalars_eval: operation or slot out of range");
        pr.op[i] = o;
        if(o.op == RAMD_SOP_SET)
            pr.op[i].a = pr.op[i].b = -1;
        else if(!(o.op >= RAMD_SOP_ADD && o.op <= RAMD_SOP_DIV) && o.op != RAMD_SOP_CMOVLT)
            pr.op[i].b = -1;
    }
    Backend& b = backend();
    hipLaunchKernelGGL(k_scalar_prog, dim3(1), dim3(64), 0, b.cur, pr, count, single, b.d_scalars);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_vec_combine_s(ramd_vec_t x, int nterms, const ramd_vec_t* vs, const int* slots, const double* factors, int guard)
{
    if(!x || !vs || !slots || !factors || nterms < 1 || nterms > 3 || (guard >= 0 && !slot_ok(guard)))
        RAMD_FAIL(RAMD_ERR_ARG, "vec_combine_s: bad arguments");
    CombineArgs a = {};
    for(int k = 0; k < nterms; ++k)
    {
        CHECK_SAMEV(vs[k], x);
        if(slots[k] >= 0 && !slot_ok(slots[k]))
            RAMD_FAIL(RAMD_ERR_ARG, "vec_combine_s: slot out of range");
        a.v[k]      = vs[k]->d;
        a.slot[k]   = slots[k];
        a.factor[k] = factors[k];
    }
    if(x->n == 0)
        return RAMD_OK;
    Backend& b = backend();
#define COMBINE(T, NTT)                                                                                                       \
    hipLaunchKernelGGL((k_combine_s<T, NTT>), dim3((unsigned)((x->n + (int64_t)kBlock * Pack<T>::N - 1) / ((int64_t)kBlock * Pack<T>::N))), \
                       dim3(kBlock), 0, b.cur, x->n, (T*)x->d, a, guard, b.d_scalars)
    prof_begin(RAMD_PROF_VEC, b.cur);
    if(x->dtype == RAMD_F64)
    {
        if(nterms == 1)
            COMBINE(double, 1);
        else if(nterms == 2)
            COMBINE(double, 2);
        else
            COMBINE(double, 3);
    }
    else if(x->dtype == RAMD_F32)
    {
        if(nterms == 1)
            COMBINE(float, 1);
        else if(nterms == 2)
            COMBINE(float, 2);
        else
            COMBINE(float, 3);
    }
    else
    {
        prof_end(RAMD_PROF_VEC, b.cur);
        RAMD_FAIL(RAMD_ERR_ARG, "vec_combine_s needs real vectors");
    }
    prof_end(RAMD_PROF_VEC, b.cur);
#undef COMBINE
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_fused_multi_dot(const ramd_vec_t* vs, int count, ramd_vec_t w, int slot0)
{
    if(!vs || !w || count < 1 || !slot_ok(slot0) || !slot_ok(slot0 + count - 1))
        RAMD_FAIL(RAMD_ERR_ARG, "fused_multi_dot: bad arguments");
    for(int j = 0; j < count; ++j)
        CHECK_SAMEV(vs[j], w);
    if(w->n == 0)
    {
        for(int j = 0; j < count; ++j)
            RAMD_TRY(ramd_scalars_set(slot0 + j, 0.0));
        return RAMD_OK;
    }
    if(w->dtype == RAMD_F64)
        return multi_dot_t<double>(vs, count, w, slot0);
    if(w->dtype == RAMD_F32)
        return multi_dot_t<float>(vs, count, w, slot0);
    RAMD_FAIL(RAMD_ERR_ARG, "fused_multi_dot needs real vectors");
}

int ramd_fused_mgs_step(ramd_vec_t w, ramd_vec_t v, int slot_h, ramd_vec_t u, int slot_dot)
{
    CHECK_SAMEV(w, v);
    if(u)
        CHECK_SAMEV(w, u);
    if(!slot_ok(slot_h) || !slot_ok(slot_dot))
        RAMD_FAIL(RAMD_ERR_ARG, "scalar slot out of range");
    if(w->n == 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = grid_reduce(w->n, w->dtype);
    ReduceCtx ctx  = reduce_ctx();
    // w streams when the working set of a projection (w, v, u) is far beyond the 256 MB last-level cache
    static const int ntw_env = getenv("RAMD_MGS_NT") ? atoi(getenv("RAMD_MGS_NT")) : -1; // (0 / 1: force, A/B experiments)
    const bool       ntw     = ntw_env >= 0 ? ntw_env != 0 : (int64_t)w->n * 8 > (int64_t)256 * 1024 * 1024;
#define GO(T)                                                                                      \
    do                                                                                             \
    {                                                                                              \
        if(u && ntw)                                                                               \
            hipLaunchKernelGGL((k_mgs_step<T, true, true>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, \
                               (T*)w->d, (const T*)v->d, (const T*)u->d, ctx, slot_h, slot_dot);   \
        else if(u)                                                                                 \
            hipLaunchKernelGGL((k_mgs_step<T, true, false>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, \
                               (T*)w->d, (const T*)v->d, (const T*)u->d, ctx, slot_h, slot_dot);   \
        else if(ntw)                                                                               \
            hipLaunchKernelGGL((k_mgs_step<T, false, true>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, \
                               (T*)w->d, (const T*)v->d, (const T*)nullptr, ctx, slot_h, slot_dot); \
        else                                                                                       \
            hipLaunchKernelGGL((k_mgs_step<T, false, false>), dim3(grid), dim3(kBlock), 0, b.cur, w->n, \
                               (T*)w->d, (const T*)v->d, (const T*)nullptr, ctx, slot_h, slot_dot); \
    } while(0)
    prof_begin(RAMD_PROF_VEC, b.cur);
    if(w->dtype == RAMD_F64)
        GO(double);
    else if(w->dtype == RAMD_F32)
        GO(float);
    else
        RAMD_FAIL(RAMD_ERR_ARG, "fused_mgs_step needs real vectors");
    prof_end(RAMD_PROF_VEC, b.cur);
#undef GO
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

int ramd_fused_mgs_block_max(void)
{
    return kMgsBlock;
}

int ramd_fused_mgs_block(ramd_vec_t w, const ramd_vec_t* vprev, int nprev, int slot_h, int slot_eprev,
                         const ramd_vec_t* vcur, int ncur, int slot_e)
{
    if(!w || nprev < 0 || ncur < 0 || nprev > kMgsBlock || ncur > kMgsBlock || (nprev == 0 && ncur == 0)
       || (nprev > 0 && !vprev) || (ncur > 0 && !vcur) || (nprev > 0 && ncur > 0 && nprev != kMgsBlock))
        RAMD_FAIL(RAMD_ERR_ARG, "fused_mgs_block: bad arguments (blocks of 1..ramd_fused_mgs_block_max() vectors; a block followed by another is full)");
    const int nsum = ncur == 0 ? 1 : ncur + ncur * (ncur - 1) / 2;
    if(!slot_ok(slot_e) || !slot_ok(slot_e + nsum - 1))
        RAMD_FAIL(RAMD_ERR_ARG, "scalar slot out of range");
    if(nprev > 0)
    {
        const int nprevsum = nprev + nprev * (nprev - 1) / 2;
        if(!slot_ok(slot_h) || !slot_ok(slot_h + nprev - 1) || !slot_ok(slot_eprev) || !slot_ok(slot_eprev + nprevsum - 1))
            RAMD_FAIL(RAMD_ERR_ARG, "scalar slot out of range");
        // (the sums of this pass must not land on what its prologue still reads)
        if(slot_e < slot_eprev + nprevsum && slot_eprev < slot_e + nsum)
            RAMD_FAIL(RAMD_ERR_ARG, "fused_mgs_block: the two slot areas overlap");
    }
    for(int j = 0; j < nprev; ++j)
        CHECK_SAMEV(vprev[j], w);
    for(int j = 0; j < ncur; ++j)
        CHECK_SAMEV(vcur[j], w);
    if(w->n == 0)
    {
        for(int j = 0; j < nsum; ++j)
            RAMD_TRY(ramd_scalars_set(slot_e + j, 0.0));
        return RAMD_OK;
    }
    static const int ntw_env = getenv("RAMD_MGS_NT") ? atoi(getenv("RAMD_MGS_NT")) : -1; // (0 / 1: force, A/B experiments)
    const bool       ntw     = ntw_env >= 0 ? ntw_env != 0 : (int64_t)w->n * 8 > (int64_t)256 * 1024 * 1024;
    if(w->dtype == RAMD_F64)
        return mgs_block_t<double>(w, vprev, nprev, slot_h, slot_eprev, vcur, ncur, slot_e, ntw);
    if(w->dtype == RAMD_F32)
        return mgs_block_t<float>(w, vprev, nprev, slot_h, slot_eprev, vcur, ncur, slot_e, ntw);
    RAMD_FAIL(RAMD_ERR_ARG, "fused_mgs_block needs real vectors");
}

int ramd_fused_normalize(ramd_vec_t v, int slot_sq, int slot_norm)
{
    if(!v || !slot_ok(slot_sq) || !slot_ok(slot_norm) || slot_sq == slot_norm)
        RAMD_FAIL(RAMD_ERR_ARG, "fused_normalize: bad arguments (slots must differ)");
    if(v->n == 0)
        return RAMD_OK;
    Backend&  b    = backend();
    const int grid = grid_oneshot(v->n, v->dtype);
    if(v->dtype == RAMD_F64)
        hipLaunchKernelGGL((k_normalize<double>), dim3(grid), dim3(kBlock), 0, b.cur, v->n,
                           (double*)v->d, b.d_scalars, slot_sq, slot_norm);
    else if(v->dtype == RAMD_F32)
        hipLaunchKernelGGL((k_normalize<float>), dim3(grid), dim3(kBlock), 0, b.cur, v->n, (float*)v->d,
                           b.d_scalars, slot_sq, slot_norm);
    else
        RAMD_FAIL(RAMD_ERR_ARG, "fused_normalize needs a real vector");
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

} // extern "C"

template <typename T>
static int multi_axpy_t(ramd_vec_t x, const ramd_vec_t* vs, const double* coef, int count)
{
    Backend& b = backend();
    for(int j0 = 0; j0 < count; j0 += kMaxMultiDot)
    {
        const int        nv = (count - j0 < kMaxMultiDot) ? count - j0 : kMaxMultiDot;
        MultiAxpyArgs<T> a;
        for(int j = 0; j < kMaxMultiDot; ++j)
        {
            a.v[j] = (const T*)vs[j0 + (j < nv ? j : 0)]->d;
            a.c[j] = (T)coef[j0 + (j < nv ? j : 0)];
        }
        const int64_t np   = x->n / Pack<T>::N;
        const int64_t g    = (np + kBlock - 1) / kBlock;
        const int     grid = (int)(g < 1 ? 1 : g);
#define GO(NV)                                                                                              \
    case NV:                                                                                                \
        hipLaunchKernelGGL((k_multi_axpy<T, NV>), dim3(grid), dim3(kBlock), 0, b.cur, x->n, (T*)x->d, a); \
        break;
        switch(nv)
        {
            GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
        }
#undef GO
    }
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

extern "C" int ramd_fused_multi_axpy(ramd_vec_t x, const ramd_vec_t* vs, const double* coef, int count)
{
    if(!x || !vs || !coef || count < 1)
        RAMD_FAIL(RAMD_ERR_ARG, "fused_multi_axpy: bad arguments");
    for(int j = 0; j < count; ++j)
    {
        CHECK_SAMEV(vs[j], x);
        if(vs[j] == x)
            RAMD_FAIL(RAMD_ERR_ARG, "fused_multi_axpy: a vector aliases the target");
    }
    if(x->n == 0)
        return RAMD_OK;
    if(x->dtype == RAMD_F64)
        return multi_axpy_t<double>(x, vs, coef, count);
    if(x->dtype == RAMD_F32)
        return multi_axpy_t<float>(x, vs, coef, count);
    RAMD_FAIL(RAMD_ERR_ARG, "fused_multi_axpy needs real vectors");
}
