// device_utils.hpp -- wave64 / workgroup primitives shared by all kernels (gfx950).
#pragma once

#include "common.hpp"

namespace ramd
{

// 16-byte packets: the coalescing sweet spot for bandwidth-bound kernels (1 KiB / wave-instr)
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef float  v4f32 __attribute__((ext_vector_type(4)));
typedef int    v4i32 __attribute__((ext_vector_type(4)));
typedef int    v2i32 __attribute__((ext_vector_type(2)));

template <typename T>
struct Pack;
template <>
struct Pack<double>
{
    using type             = v2f64;
    static constexpr int N = 2;
};
template <>
struct Pack<float>
{
    using type             = v4f32;
    static constexpr int N = 4;
};
template <>
struct Pack<int>
{
    using type             = v4i32;
    static constexpr int N = 4;
};

template <typename T>
__device__ __forceinline__ T* pk_elems(typename Pack<T>::type& p)
{
    return reinterpret_cast<T*>(&p);
}

// streaming (read-once / write-once) accesses: keep val/col/y out of the way of the x vector
// that the SpMV wants resident in L2.
template <typename X>
__device__ __forceinline__ X nt_load(const X* p)
{
    return __builtin_nontemporal_load(p);
}
template <typename X>
__device__ __forceinline__ void nt_store(X v, X* p)
{
    __builtin_nontemporal_store(v, p);
}

// ---- wave64 / block reductions (fixed order => deterministic) ----
__device__ __forceinline__ double wave_reduce_sum(double v)
{
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

// sum over a 256-thread workgroup; result valid in thread 0. `lds` holds >= 4 doubles per value.
__device__ __forceinline__ double block_reduce_sum(double v, double* lds)
{
    v         = wave_reduce_sum(v);
    int lane  = threadIdx.x & 63;
    int wave  = threadIdx.x >> 6;
    int nwave = (blockDim.x + 63) >> 6;
    if(lane == 0)
        lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if(threadIdx.x == 0)
        for(int w = 0; w < nwave; ++w)
            r += lds[w];
    __syncthreads();
    return r;
}

// ---- single-launch grid reduction ---------------------------------------------------------
// Every workgroup deposits NS partial sums, takes a ticket, and the LAST arriver adds the
// partials of all workgroups in a fixed order and writes the NS results into the scalar
// record -- one launch, deterministic, no atomics on the data.  Hand-off follows the
// gfx950 rule (MI355X_MICROARCH.md "inter-workgroup visibility"): agent-scope release on the
// producer (+ explicit vmcnt drain, the compiler may drop it), ticket, agent-scope acquire on
// the consumer, L1-bypassing loads of the partials.
struct ReduceCtx
{
    double*       partials; // [slot][kReduceBlocks]
    unsigned int* ticket; // one counter per concurrently running reduction kernel
    double*       scalars; // device scalar record
};

enum ReduceOp
{
    RED_SUM  = 0,
    RED_SQRT = 1, // result = sqrt(sum)  (Norm)
    RED_ACC  = 2  // result = slot + sum (correction of a dot already in the slot)
};

template <int NS>
__device__ __forceinline__ void grid_reduce_finish(const ReduceCtx& ctx, const double (&val)[NS],
                                                   const int (&slot)[NS], const int (&op)[NS],
                                                   double* lds /* >= 4*NS + 1 doubles */)
{
    __shared__ int s_last;
    double         bsum[NS];
#pragma unroll
    for(int k = 0; k < NS; ++k)
        bsum[k] = block_reduce_sum(val[k], lds + 4 * k);
    if(threadIdx.x == 0)
    {
#pragma unroll
        for(int k = 0; k < NS; ++k)
            __hip_atomic_store(&ctx.partials[(size_t)slot[k] * kReduceBlocks + blockIdx.x], bsum[k],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The partials travel as 8-byte agent-scope atomics on both sides (write-through store here, L1/L2-bypassing
        // load in the last workgroup): the valid hand-off form that needs NO release fence.  An agent-scope release
        // fence is a write-back of the XCD L2's dirty lines -- in a streaming kernel that is megabytes, once per
        // workgroup (measured: the fused update kernels got slower when the grid grew from 2048 to 8192 workgroups).
        // Only the completion of the stores has to precede the ticket.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int t = __hip_atomic_fetch_add(ctx.ticket, 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
        s_last         = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if(s_last)
    {
        if(threadIdx.x == 0)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
#pragma unroll
        for(int k = 0; k < NS; ++k)
        {
            double a = 0.0;
            for(int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x)
                a += __hip_atomic_load(&ctx.partials[(size_t)slot[k] * kReduceBlocks + b],
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double tot = block_reduce_sum(a, lds + 4 * k);
            if(threadIdx.x == 0)
            {
                if(op[k] == RED_SQRT)
                    tot = sqrt(tot);
                else if(op[k] == RED_ACC)
                    tot = ctx.scalars[slot[k]] + tot;
                ctx.scalars[slot[k]] = tot;
            }
        }
        if(threadIdx.x == 0)
            __hip_atomic_store(ctx.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// every spin is bounded: a dependency that never arrives (a bug, by construction of the ticket
// order) becomes a loud kernel abort instead of a hung GPU
constexpr int kSpinLimit = 1 << 24;
__device__ __forceinline__ void spin_guard(int& spins)
{
    if(++spins > kSpinLimit)
        __builtin_trap();
}

// poll_turn (the sweeps in 64-row units, trisolve.hip / blocksched.hip): a wave none of whose lanes advanced in the last
// turn waits with ONE polling lane -- its first unfinished one -- instead of 64; everybody polls again once that lane gets
// through.  On a dependency graph with little parallelism (an FE shell: ~20 units per level) nearly all of the ~8000
// resident waves wait, and 64 polls per wave and turn saturate the L2 the few productive waves need: measured 0.37 s for
// the level sweep of the 1.5 M-row shell surrogate before, 0.19 s after (the rest was the unit structure, blocksched.hip).
// polling loops: sleep when no lane of the wave advanced, doubling up to 64 x 64 cycles; returns the next back-off
__device__ __forceinline__ int poll_backoff(bool wave_advanced, int backoff, int cap = 64)
{
    if(wave_advanced)
        return 1;
    for(int z = 0; z < backoff; ++z)
        __builtin_amdgcn_s_sleep(1);
    return backoff < cap ? backoff * 2 : cap;
}

// workgroup ticket: the k-th workgroup to START works on block k (deadlock freedom does not depend
// on the dispatch order).  `base` is the counter value before this launch.
__device__ __forceinline__ unsigned take_ticket(unsigned* counter, unsigned base)
{
    __shared__ unsigned s_t;
    if(threadIdx.x == 0)
        s_t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
    __syncthreads();
    return s_t;
}

// ... per wave: the k-th WAVE to start works on unit k (sweeps in units of 64 rows)
__device__ __forceinline__ unsigned take_wave_ticket(unsigned* counter, unsigned base)
{
    unsigned v = 0;
    if((threadIdx.x & 63) == 0)
        v = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

inline ReduceCtx reduce_ctx(int ticket_id = 0)
{
    Backend&  b = backend();
    ReduceCtx c;
    c.partials = b.d_partials;
    c.ticket   = b.d_ticket + ticket_id;
    c.scalars  = b.d_scalars;
    return c;
}

// grid for reduction kernels: never more than kReduceBlocks workgroups
inline int reduce_grid(int64_t n_items)
{
    int64_t g = (n_items + kBlock - 1) / kBlock;
    if(g > kReduceBlocks)
        g = kReduceBlocks;
    if(g < 1)
        g = 1;
    return (int)g;
}

// XCD-aware mapping: hardware places workgroup b on XCD b % 8 (observed, used for speed only).
// Every XCD gets one contiguous eighth of the row blocks and walks it in dispatch order, so rows that
// gather the same x planes are in flight on the same L2 at the same time.
//
// Band-aware traversal (P > 0): matrices with a far band at distance D rows (3-D stencils: D = one
// grid plane) gather x[r-D], x[r], x[r+D]; walking the rows plane by plane puts 2 planes (4 MiB of x at
// 512^3) between the first and the last use of an x entry -- more than an XCD's L2, so x is fetched from
// HBM three times (measured: +2 GB per SpMV).  Instead the XCD's range is cut into in-plane tiles of W
// row blocks and every tile is swept through all planes before the next tile starts: the reuse window
// shrinks to 3 x W x 2 KiB.  P = D / 256 row blocks per plane, Z = planes in the XCD's range.
struct BandMap
{
    int P, W, Z; // P == 0: linear order
};
__device__ __forceinline__ int xcd_block(int nblk, int per_xcd, BandMap bm)
{
    const int i = blockIdx.x >> 3; // position in this XCD's dispatch sequence
    int       l = i;
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


} // namespace ramd
