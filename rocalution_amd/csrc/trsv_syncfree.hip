// trsv_syncfree.hip -- sync-free grouped sparse triangular solve for gfx950 (k_trsv_sf), its plan fill and its launch.
//
// Replaces rocsparse_csrsv_solve of the reference HIP backend (src/base/hip/hip_matrix_csr.cpp:1756-1821 LUSolve, :2566-2870
// L / U Analyse + Solve) on triangles the tiles of trisolve.hip cannot take -- no chains of consecutively numbered dependent rows
// for their coordinates to grow along: a shell mesh (5 unknowns per node, ~35 entries per row) numbered by reverse Cuthill-McKee
// or by an advancing front has ~10 700 row levels of ~140 rows.  The level-scheduled kernel pays ~9 us per level there (one lane
// per row: the row's entries come in dependent chunks of eight, each behind a poll).  What such a graph needs per level is ONE
// hand-off and as little as possible behind it:
//   * the unit of work is a ROW GROUP (trisolve.hip k_ct_sn_breaks: the rows of one mesh node -- row t depends on row t-1 and
//     shares every other dependency with it): 2 140 group levels instead of 10 700; the in-group part of a step runs in registers,
//     in the order of the host loop (lower solve: the in-group entries are the LAST of a row, upper solve: the FIRST);
//   * LPR = 4 or 8 lanes share a row, kw <= 6 out-of-group entries each; a wave holds 64 / LPR rows = whole groups of ONE group
//     level (a "unit"); positions are sorted by (group level, group, row), units are contiguous pieces of them;
//   * persistent waves take the units by ticket: a wave is many levels ahead of the front when it starts on a unit, so the
//     unit's coefficients, right-hand side and diagonal are in registers long before its dependencies are;
//   * it waits with ONE load per turn on a position two levels back, then asks for every value once and after that only for
//     the values still missing (data-tagged granules): the waves ahead of the front cost the memory system one request per
//     turn each, and the turn that finds the last value is the only trip through memory on the critical path;
//   * WHILE it waits, everything that can be done with the values already there is done (round 6): the lower solve's chain of
//     out-of-group subtractions advances lane by lane as the lanes complete (its late values are the LAST entries of a row: the
//     highest columns belong to the level just before), the upper solve brings the products of completed lanes into the row's
//     first lane (its late values are the FIRST out-of-group entries, which sit in that lane already);
//   * the division by the diagonal (upper solve: five in a row per mesh node, each behind the chain of the row before) is
//     the division sequence of gfx950 with everything that depends on the divisor alone moved into the plan (sf_div): three
//     dependent operations instead of eleven, the same bits by construction;
//   * one publication per row.
// The operations per row are those of host_matrix_csr.cpp:1163-1221 (LUSolve), :1294-1341 (LLSolve), :1357-1466 (LSolve / USolve)
// in their order: bit-exact (forced over the parity suite, tests/test_gpu_syncfree.py).
#include "trsv_syncfree.hpp"

#include "device_utils.hpp"
#include "matrix_impl.hpp"
#include "trsv_handoff.hpp"

#include <string>
#include <vector>

namespace ramd
{

constexpr int kSfStreams = 8;

void sf_release(SfPlan** sp)
{
    SfPlan* q = *sp;
    if(!q)
        return;
    dev_free(&q->uinfo);
    dev_free(&q->ufar);
    dev_free(&q->punit);
    dev_free(&q->pinfo);
    dev_free(&q->ecol);
    dev_free(&q->tickets);
    if(q->eval)
        (void)cached_free(q->eval);
    if(q->gcoef)
        (void)cached_free(q->gcoef);
    if(q->rdiag)
        (void)cached_free(q->rdiag);
    delete q;
    *sp = nullptr;
}

// ---------------------------------------------------------------- the division
// hipcc expands an fp64 `a / d` (no fast math) into
//     ds = v_div_scale(d, d, a)   as = v_div_scale(a, d, a)          (vcc: a scaling happened)
//     y  = v_rcp(ds);  e = fma(-ds, y, 1);  y = fma(y, e, y);  e = fma(-ds, y, 1);  y = fma(y, e, y)
//     q  = as * y;  r = fma(-ds, q, as);  res = v_div_fmas(r, y, q);  res = v_div_fixup(res, d, a)
// v_div_scale leaves both operands alone and vcc clear (CDNA3/4 ISA, V_DIV_SCALE_F64) when neither is zero / denormal / not
// finite, exponent(a) - exponent(d) < 768, 1 / d and a / d are not denormal and exponent(a) > 53: certainly when both biased
// exponents lie in [640, 1407] (|x| in [2^-383, 2^385)).  Then ds = d, as = a, v_div_fmas is a plain fma and v_div_fixup returns
// |res| with the sign of a * d, which is res.  The second line depends on d alone: it is formed ONCE, with the same
// instructions, when the plan is filled (sf_recip_iterate), and the solve runs the third line only -- a multiplication and two
// fused multiply-adds behind the running sum instead of eleven dependent operations, the same bits by construction.  Operands
// outside the window (a zero right-hand side, an overflowed sum, a tiny pivot) take `/` -- found out per UNIT, by one look at
// its quotients (sf_quotient_in_window): a wave issues every instruction in order, and a test per division cost what the
// short sequence saved (round 6, measured: 2 532 -> 2 492 cycles per five-row unit with a test of every dividend).
constexpr unsigned kDivDLo = 923u, kDivDSpan = 201u, kDivQLo = 773u, kDivQSpan = 501u;
__device__ __forceinline__ double sf_recip_iterate(double d)
{
    // (kept only for divisors in the narrow window of sf_quotient_in_window below)
    const unsigned ex = ((unsigned)__double2hiint(d) >> 20) & 0x7ffu;
    if((ex - kDivDLo) >= kDivDSpan)
        return 0.0;
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y        = __builtin_fma(y, e, y);
    e        = __builtin_fma(-d, y, 1.0);
    y        = __builtin_fma(y, e, y);
    return y;
}
__device__ __forceinline__ float sf_recip_iterate(float)
{
    return 0.0f;
}
// the third line for an unscaled division: a / d where y = sf_recip_iterate(d) != 0 and a lies inside the window -- the caller's
// business (sf_unit checks the QUOTIENTS of a unit once, see sf_quotient_in_window)
__device__ __forceinline__ double sf_div_short(double a, double d, double y)
{
    const double q = a * y;
    const double r = __builtin_fma(-d, q, a);
    return __builtin_fma(r, y, q);
}
__device__ __forceinline__ float sf_div_short(float a, float d, float)
{
    return a / d;
}
// One test per RESULT instead of one per dividend.  The plan keeps the reciprocal iterate only for divisors with biased
// exponents in [923, 1123] (|d| in [2^-100, 2^101); 0 otherwise).  If the short sequence returns a quotient q with biased
// exponent in [773, 1273] (|q| in [2^-250, 2^251)), the dividend a = q d (1 + O(2^-52)) had its exponent in [672, 1374], inside
// the window of the short sequence: q is a / d.  Conversely a dividend outside that window cannot produce such a q with such a
// d: a = +-0 gives q = +-0, an infinity or a NaN gives a NaN, |a| < 2^-383 gives |q| < 2^-281, |a| >= 2^385 gives |q| >= 2^283
// (or an infinity), and y = 0 gives q = 0.
__device__ __forceinline__ bool sf_quotient_in_window(double q)
{
    const unsigned e = ((unsigned)__double2hiint(q) >> 20) & 0x7ffu;
    return (e - kDivQLo) < kDivQSpan;
}
__device__ __forceinline__ bool sf_quotient_in_window(float)
{
    return true;
}
// a / d the way a unit does it: short sequence, result checked, `/` otherwise (the probe kernel of the tests)
__device__ __forceinline__ double sf_div(double a, double d, double y)
{
    const double q = sf_div_short(a, d, y);
    return sf_quotient_in_window(q) ? q : a / d;
}

// a kernel of its own for the tests (ramd_selftest_sf_div): out[i] = sf_div(a[i], d[i]) next to a[i] / d[i]
__global__ __launch_bounds__(kBlock) void k_sf_div_probe(int64_t n, const double* __restrict__ a, const double* __restrict__ d,
                                                         double* __restrict__ fast, double* __restrict__ plain,
                                                         int* __restrict__ in_window)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
    {
        const double y = sf_recip_iterate(d[i]);
        const double q = sf_div_short(a[i], d[i], y);
        fast[i]        = sf_div(a[i], d[i], y);
        plain[i]       = a[i] / d[i];
        in_window[i]   = sf_quotient_in_window(q) ? 1 : 0; // (1: the short sequence's quotient is what fast[i] holds)
    }
}

// ---------------------------------------------------------------- plan fill
// one wave per unit: the out-of-group entries of its rows into the unit's planes (entry e of a row -> lane e / kw of the row,
// plane e % kw; only the nl lanes per row that hold entries are stored), the in-group coefficients, the diagonal and its
// reciprocal iterate per position, the unit's last dependency
template <typename T, bool LOWER, int LPR>
__global__ __launch_bounds__(64) void k_sf_fill(int n, int nunits, int* __restrict__ uinfo, const int* __restrict__ pinfo,
                                                const int* __restrict__ order, const int* __restrict__ pos,
                                                const int* __restrict__ rp, const int* __restrict__ ci, const T* __restrict__ val,
                                                int* __restrict__ ecol, T* __restrict__ eval, T* __restrict__ gcoef,
                                                T* __restrict__ diag, T* __restrict__ rdiag, int* __restrict__ nodiag,
                                                int* __restrict__ punit, int reverse)
{
    const int u = blockIdx.x;
    if(u >= nunits)
        return;
    const int     lane = threadIdx.x, slot = lane / LPR, l = lane % LPR;
    const int     w1 = uinfo[4 * u + 1];
    const int     p0 = uinfo[4 * u], cnt = w1 & 255, kw = (w1 >> 8) & 15, nl = (w1 >> 12) & 15;
    const int     ps = cnt * nl; // slots of a plane
    const int64_t e0 = (int64_t)uinfo[4 * u + 2] + slot * nl + l;
    const bool    have = slot < cnt, act = have && l < nl;
    int           maxdep = -1;
    int           k      = 0; // planes of this lane filled so far
    if(have)
    {
        const int p = p0 + slot, i = order[p];
        const int r = pinfo[p] & 15;
        if(l == 0)
            punit[p] = u;
        // (sweep index of the group's first row: rows of a group are consecutive positions AND consecutive sweep rows)
        const int t = LOWER ? i : n - 1 - i, tf = t - r;
        int       e = 0;
        bool      dg = false;
        const int rs = rp[i], re = rp[i + 1];
        for(int q = rs; q < re; ++q)
        {
            const int j   = reverse ? re - 1 - (q - rs) : q; // (reverse: the entries in descending storage order)
            const int col = ci[j];
            if(col == i)
            {
                if(l == 0)
                {
                    diag[p] = val[j];
                    if(rdiag)
                        rdiag[p] = sf_recip_iterate(val[j]);
                }
                dg = true;
                continue;
            }
            if(!(LOWER ? (col < i) : (col > i)))
                continue;
            const int tc = LOWER ? col : n - 1 - col;
            if(tc >= tf)
            {
                if(l == 0 && gcoef)
                    gcoef[(int64_t)p * 8 + (tc - tf)] = val[j];
                continue;
            }
            if(e / kw == l) // (l < nl: a row's entries fill its lanes from lane 0 on, nl = lanes of the unit's longest row)
            {
                const int pc = pos[col];
                ecol[e0 + (int64_t)k * ps] = pc;
                eval[e0 + (int64_t)k * ps] = val[j];
                maxdep                     = max(maxdep, pc);
                ++k;
            }
            ++e;
        }
        if(!dg && l == 0)
        {
            diag[p] = (T)1;
            if(rdiag)
                rdiag[p] = sf_recip_iterate((T)1);
            *nodiag = 1;
        }
    }
    if(act)
        for(; k < kw; ++k)
        {
            ecol[e0 + (int64_t)k * ps] = -1;
            eval[e0 + (int64_t)k * ps] = (T)0;
        }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
        maxdep = max(maxdep, __shfl_xor(maxdep, off, 64));
    if(lane == 0)
        uinfo[4 * u + 3] = maxdep;
}

// ufar[u]: the last dependency of the unit that holds the last dependency of ... (depth times) of unit u, -1: none that far back
__global__ __launch_bounds__(kBlock) void k_sf_far(int nunits, int depth, const int* __restrict__ uinfo, const int* __restrict__ punit,
                                                   int* __restrict__ ufar)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < nunits; u += gsz)
    {
        int q = uinfo[4 * u + 3];
        for(int d = 0; d < depth && q >= 0; ++d)
            q = uinfo[4 * punit[q] + 3];
        ufar[u] = q;
    }
}

template <typename T>
int sf_fill(SfPlan* S, int n, bool lower, bool reverse, const int* order, const int* pos, const int* rp, const int* ci, const T* val,
            T* diag, bool* nodiag_out)
{
    Backend& b      = backend();
    int*     nodiag = nullptr;
    RAMD_TRY(dev_alloc(&nodiag, 1));
    auto fail = [&](int s) {
        dev_free(&nodiag);
        return s;
    };
    if(hipMemsetAsync(nodiag, 0, sizeof(int), b.cur) != hipSuccess)
        return fail(RAMD_ERR_HIP);
    if(!S->tickets)
    {
        const int s = dev_alloc(&S->tickets, 32 * (1 + kSfStreams));
        if(s != RAMD_OK)
            return fail(s);
    }
#define SF_FILL(LO, LP)                                                                                                    \
    hipLaunchKernelGGL((k_sf_fill<T, LO, LP>), dim3(S->nunits), dim3(64), 0, b.cur, n, S->nunits, S->uinfo, S->pinfo, order, \
                       pos, rp, ci, val, S->ecol, (T*)S->eval, (T*)S->gcoef, diag, (T*)S->rdiag, nodiag, S->punit, reverse ? 1 : 0)
    if(lower && S->lpr == 4)
        SF_FILL(true, 4);
    else if(lower)
        SF_FILL(true, 8);
    else if(S->lpr == 4)
        SF_FILL(false, 4);
    else
        SF_FILL(false, 8);
#undef SF_FILL
    {
        static const int far_depth = getenv("RAMD_TRSV_SF_FAR") ? atoi(getenv("RAMD_TRSV_SF_FAR")) : 1;
        hipLaunchKernelGGL(k_sf_far, dim3(ew_grid(S->nunits)), dim3(kBlock), 0, b.cur, S->nunits, far_depth, S->uinfo, S->punit, S->ufar);
    }
    int nd = 0;
    if(hipMemcpyAsync(&nd, nodiag, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess
       || hipGetLastError() != hipSuccess)
        return fail(RAMD_ERR_HIP);
    *nodiag_out = nd != 0;
    return fail(RAMD_OK);
}
template int sf_fill<double>(SfPlan*, int, bool, bool, const int*, const int*, const int*, const int*, const double*, double*, bool*);
template int sf_fill<float>(SfPlan*, int, bool, bool, const int*, const int*, const int*, const int*, const float*, float*, bool*);

// ---------------------------------------------------------------- lane helpers
// value of lane `src` (uniform) in every lane
__device__ __forceinline__ double sf_from_lane(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float sf_from_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// value of lane + Q of the same 16-lane row (Q compile-time; Q = 0: the lane's own)
template <int Q>
__device__ __forceinline__ double sf_from_lane_after(double v)
{
    if constexpr(Q == 0)
        return v;
    else
    {
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x100 + Q, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x100 + Q, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
}
template <int Q>
__device__ __forceinline__ float sf_from_lane_after(float v)
{
    if constexpr(Q == 0)
        return v;
    else
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x100 + Q, 0xf, 0xf, true));
}

// lanes 0 of the rows of a wave (one bit per row)
template <int LPR>
struct SfRowMask;
template <>
struct SfRowMask<8>
{
    static constexpr unsigned long long value = 0x0101010101010101ull;
};
template <>
struct SfRowMask<4>
{
    static constexpr unsigned long long value = 0x1111111111111111ull;
};

// the products of lane + Q of every row into all[Q] of the row's first lane, for every Q whose lanes are complete in all rows
// (ready: one bit per lane) and that has not been brought over yet (done: one bit per Q, uniform)
template <typename T, int LPR, int NA, int Q>
__device__ __forceinline__ void sf_pull_ready(T (&all)[LPR][NA], const T (&prod)[NA], int nl, unsigned long long ready, unsigned& done)
{
    if constexpr(Q < LPR)
    {
        constexpr unsigned long long LM = SfRowMask<LPR>::value;
        if(Q < nl && !((done >> Q) & 1u) && ((ready >> Q) & LM) == LM)
        {
#pragma unroll
            for(int k = 0; k < NA; ++k)
                all[Q][k] = sf_from_lane_after<Q>(prod[k]);
            done |= 1u << Q;
        }
        sf_pull_ready<T, LPR, NA, Q + 1>(all, prod, nl, ready, done);
    }
}

// ---------------------------------------------------------------- one unit: wait for the dependencies, compute
// the arithmetic of a unit once its dependencies are there.  Lower solve (INFIRST false): `sum` has been through the row's
// chain of out-of-group subtractions (result in lane nl - 1 of the row); the in-group entries are the last of a row: row j of
// a group is final once the rows before it have been taken out of it.  Upper solve: the in-group entries come first -- the rows
// of a group one after the other (nearest row first), each with its in-group terms, its chain and its division: five chains
// per node instead of one; a round's chain is a straight line of subtractions in the row's first lane, whose operands (`all`)
// the waiting turns have brought there; the result of a row is in its lane 0.
// r: row number inside its group, nl (uniform): lanes per row in use, maxm (uniform): rows of the unit's longest group, ONE:
// the unit holds one group (the lane a finished row is taken from is uniform: v_readlane, else ds_bpermute).
// EXACT false: fp64 divisions by the short sequence, unchecked -- the caller checks the results (sf_unit).
template <typename T, int DMODE, bool INFIRST, int LPR, int NA, bool ONE, bool EXACT>
__device__ __forceinline__ T sf_rounds(T sum, const T (&all)[INFIRST ? LPR : 1][NA], T rhs, T dg, T rdg, const T (&gc)[7], int r, int l,
                                       int slot, int nl, int maxm)
{
    if constexpr(!INFIRST)
    {
        const int gl0 = (slot - r) * LPR + nl - 1; // where the group's first row ends its chain
#pragma unroll
        for(int j = 0; j < kGrpMax; ++j)
        {
            if(j >= maxm)
                break;
            if(DMODE != 0)
            {
                const T q = DMODE == 1 ? (EXACT ? sum / dg : sf_div_short(sum, dg, rdg)) : sum * dg;
                sum       = (r == j) ? q : sum;
            }
            if(j < 7 && j + 1 < maxm)
            {
                const T yj = ONE ? sf_from_lane(sum, j * LPR + nl - 1) : __shfl(sum, gl0 + j * LPR, 64);
                const T t  = sum - gc[j] * yj;
                sum        = (r > j) ? t : sum;
            }
        }
        return sum;
    }
    else
    {
        const int g0 = (slot - r) * LPR; // lane 0 of the group's first row
        // gyI = gc[I] * y_I, the in-group term of the rows after row I: formed once, when y_I is there (the same product every
        // later round would form).  Seven named values, not an array: an array that lives across the early exits of the rounds
        // travels through them as ONE register tuple, a dozen moves per round (seen in the ISA of round 6's first version).
        T gy0 = (T)0, gy1 = (T)0, gy2 = (T)0, gy3 = (T)0, gy4 = (T)0, gy5 = (T)0, gy6 = (T)0;
        T res = (T)0;
#define SF_ROUND(J, INGROUP, KEEP)                                                                         \
    if((J) < maxm)                                                                                         \
    {                                                                                                      \
        T s = rhs;                                                                                         \
        INGROUP;                                                                                           \
        _Pragma("unroll") for(int q = 0; q < LPR; q += 2)                                                  \
        {                                                                                                  \
            if(q >= nl)                                                                                    \
                break;                                                                                     \
            _Pragma("unroll") for(int k = 0; k < NA; ++k) s -= all[q][k];                                  \
            _Pragma("unroll") for(int k = 0; k < NA; ++k) s -= all[q + 1][k];                              \
        }                                                                                                  \
        if(DMODE == 1)                                                                                     \
            s = EXACT ? s / dg : sf_div_short(s, dg, rdg);                                                 \
        else if(DMODE == 2)                                                                                \
            s = s * dg;                                                                                    \
        res = (r == (J)) ? s : res;                                                                        \
        if((J) < 7 && (J) + 1 < maxm)                                                                      \
        {                                                                                                  \
            const T yj = ONE ? sf_from_lane(s, (J) * LPR) : __shfl(s, g0 + (J) * LPR, 64);                 \
            KEEP = gc[(J) < 7 ? (J) : 0] * yj;                                                             \
        }                                                                                                  \
    }
        // (two lanes of products per test: a lane beyond nl holds +0 products, and s - (+0) is s -- half the tests of a walk lane
        //  by lane; one wave runs a unit and issues every instruction in order, tests included.  The in-group terms nearest row
        //  first: the order of the host loop, ascending columns.)
        T unused = (T)0;
        SF_ROUND(0, (void)0, gy0)
        SF_ROUND(1, s -= gy0, gy1)
        SF_ROUND(2, s -= gy1; s -= gy0, gy2)
        SF_ROUND(3, s -= gy2; s -= gy1; s -= gy0, gy3)
        SF_ROUND(4, s -= gy3; s -= gy2; s -= gy1; s -= gy0, gy4)
        SF_ROUND(5, s -= gy4; s -= gy3; s -= gy2; s -= gy1; s -= gy0, gy5)
        SF_ROUND(6, s -= gy5; s -= gy4; s -= gy3; s -= gy2; s -= gy1; s -= gy0, gy6)
        SF_ROUND(7, s -= gy6; s -= gy5; s -= gy4; s -= gy3; s -= gy2; s -= gy1; s -= gy0, unused)
#undef SF_ROUND
        (void)unused;
        return res;
    }
}

// c / a: the lane's out-of-group entries (position, coefficient), NA of them (a slot without an entry: c < 0, coefficient +0: the
// product (+0)(+0) = +0, and s - (+0) is s bit for bit for every s, -0 included: no test per entry, NA subtractions per lane in
// a straight line -- every lane computes, the lane whose turn it is keeps: one wave runs a unit, and a taken branch costs it more
// than the arithmetic it would skip).  Waits for the unit's dependencies, does what can be done with the values that are there
// while it waits, finishes (sf_rounds), publishes the rows' results (w[p], and out[onat] where out is given).
template <typename T, int DMODE, bool INFIRST, int LPR, int NA, bool ONE>
__device__ __forceinline__ void sf_unit(T* w, T* __restrict__ out, int p, int onat, int pidle, const int (&c)[kSfKW], const T (&a)[kSfKW],
                                        T rhs, T dg, T rdg, const T (&gc)[7], bool have, int r, int l, int slot, int nl, int maxm,
                                        int flags, int stagger, unsigned long long* __restrict__ dbg, int64_t nunits, int u)
{
    using B = typename Sentinel<T>::bits;
    constexpr unsigned long long LM = SfRowMask<LPR>::value;
    const bool nowait = (flags & 1) != 0, pipe = (flags & 2) != 0;
    B x[NA];
#pragma unroll
    for(int k = 0; k < NA; ++k)
        x[k] = c[k] >= 0 ? Sentinel<T>::value : (B)0;
    // what the waiting turns leave behind: lower solve -- the running sum, advanced through `step` lanes of every row;
    // upper solve -- the products of the lanes in `done`, in the row's first lane
    T        sum  = rhs;
    int      step = 0;
    unsigned done = 0;
    T        all[INFIRST ? LPR : 1][NA];
#pragma unroll
    for(int q = 0; q < (INFIRST ? LPR : 1); ++q)
#pragma unroll
        for(int k = 0; k < NA; ++k)
            all[q][k] = (T)0;
    // A turn: ALL its requests are issued before the first answer is looked at -- a lane whose value k is there asks for the unit's
    // first position instead (one address for the whole wave: one request).  (Round 5 asked only the lanes still missing a
    // value, behind a test per k: the compiler then waits for every answer before the next request, and the turn that finds the
    // last values of a level -- on the critical path -- was up to kw trips through memory in a row: 1.04 us per hand-off.)
    auto issue = [&](B(&v)[NA]) {
#pragma unroll
        for(int k = 0; k < NA; ++k)
            v[k] = poll_load(w + (x[k] == Sentinel<T>::value ? c[k] : pidle));
        __builtin_amdgcn_sched_barrier(0); // (the scheduler must not sink a request behind the wait for another one's answer)
    };
    // the answers of a turn -> which lanes are complete; then everything that can be done with what is there
    auto take = [&](const B(&v)[NA]) -> unsigned long long {
        bool full = true;
#pragma unroll
        for(int k = 0; k < NA; ++k)
        {
            x[k] = (x[k] == Sentinel<T>::value) ? v[k] : x[k];
            if(nowait && x[k] == Sentinel<T>::value) // (diagnostic, RAMD_TRSV_SF_GATHER=2: no dependency waits -- wrong results)
                x[k] = (B)0;
            full = full && (x[k] != Sentinel<T>::value);
        }
        const unsigned long long ready = __ballot(full);
        T                        prod[NA];
#pragma unroll
        for(int k = 0; k < NA; ++k)
            prod[k] = a[k] * Sentinel<T>::from_bits(x[k]); // (a lane that is not complete holds NaNs here: nobody takes them)
        if constexpr(!INFIRST)
        {
            // the out-of-group entries of a row, one after the other through its lanes: lane 0 starts from rhs, lane k + 1
            // continues lane k's sum; a step runs as soon as its lane is complete in every row of the unit
            while(step < nl && ((ready >> step) & LM) == LM)
            {
                T t = step > 0 ? lane_before_in_row<T>(sum) : sum;
#pragma unroll
                for(int k = 0; k < NA; ++k)
                    t -= prod[k];
                sum = (l == step) ? t : sum;
                ++step;
            }
        }
        else
            sf_pull_ready<T, LPR, NA, 0>(all, prod, nl, ready, done);
        return ready;
    };
    // Two generations of requests in flight, half a trip apart (pipe): a value that lands in memory is seen by the next request
    // to pass there, and with one generation that is on average half a trip away, with two a quarter.  A generation is asked
    // again the moment its answers have been looked at; the stagger of the start persists.
    B   va[NA], vb[NA];
    int spins = 0;
    issue(va);
    if(pipe)
    {
        for(int z = 0; z < stagger; ++z)
            __builtin_amdgcn_s_sleep(1);
        while(true)
        {
            spin_guard(spins);
            issue(vb);
            if(take(va) == ~0ull)
                break;
            issue(va);
            if(take(vb) == ~0ull)
                break;
        }
    }
    else
    {
#pragma unroll
        for(int k = 0; k < NA; ++k)
            vb[k] = (B)0;
        while(take(va) != ~0ull)
        {
            spin_guard(spins);
            __builtin_amdgcn_s_sleep(1);
            issue(va);
        }
    }
    unsigned long long d_w0 = 0, d_c0 = 0, d_c2 = 0;
    if(dbg)
    {
        d_w0 = wall_clock64();
        d_c0 = clock64();
    }
    const int reslane = INFIRST ? 0 : nl - 1;
    T         res;
    if constexpr(DMODE == 1 && sizeof(T) == 8)
    {
        // the divisions by the short sequence, unchecked; then ONE look at the rows' results: a quotient inside its window, by a
        // divisor inside its own (rdg != 0), came from a dividend inside the window in which the short sequence IS the division
        // (sf_div_short).  Otherwise -- a zero right-hand side, an overflowed sum, a tiny pivot -- the unit is done again with `/`.
        res = sf_rounds<T, DMODE, INFIRST, LPR, NA, ONE, false>(sum, all, rhs, dg, rdg, gc, r, l, slot, nl, maxm);
        if(__ballot(have && l == reslane && !sf_quotient_in_window(res)) != 0ull)
        {
            res = sf_rounds<T, DMODE, INFIRST, LPR, NA, ONE, true>(sum, all, rhs, dg, rdg, gc, r, l, slot, nl, maxm);
            if(dbg && (threadIdx.x & 63) == 0)
                atomicAdd(dbg + 4 * nunits + 4, 1ull);
        }
    }
    else
        res = sf_rounds<T, DMODE, INFIRST, LPR, NA, ONE, true>(sum, all, rhs, dg, rdg, gc, r, l, slot, nl, maxm);
    if(dbg)
    {
        asm volatile("" : "+v"(res));
        d_c2 = clock64();
    }
    if(have && l == reslane)
    {
        publish(w + p, res);
        if(out)
            out[onat] = res;
    }
    if(dbg)
    {
        const unsigned long long t1 = wall_clock64(), c3 = clock64();
        if((threadIdx.x & 63) == 0)
        {
            dbg[2 * (int64_t)u]     = d_w0;
            dbg[2 * (int64_t)u + 1] = t1;
            // cycles: dependencies seen -> (unused) | -> result final | -> publication issued
            dbg[2 * nunits + 4 + 2 * (int64_t)u]     = (d_c2 - d_c0) << 32;
            dbg[2 * nunits + 4 + 2 * (int64_t)u + 1] = c3 - d_c0;
        }
    }
    // the generation still in flight when the loop ended lands in these registers: they stay reserved until here, behind the
    // publication, so that nothing on the way to it has to wait for answers nobody needs
#pragma unroll
    for(int k = 0; k < NA; ++k)
        asm volatile("" ::"v"(va[k]), "v"(vb[k]));
}

// (Tried and removed, round 5: the waves of ONE XCD only -- every wave registers with the XCD its XCC_ID names, the XCD with most
//  waves goes on -- with values published by stores that keep their line in that XCD's L2 (workgroup scope, sc0) and polled by
//  the same sc1 loads: bit-exact, and SLOWER -- 2.0 us from publication to "dependencies seen" against 1.5 us through memory with
//  sc1 stores from all eight XCDs, LUSolve on the RCM shell 12.1 against 10.1 ms.  Also measured then and removed: waiting with
//  one word for the unit's LAST dependency before a gather of everything (two trips through memory per level: 4.1 / 5.5 ms
//  against 3.8 / 5.2), gathers of everything every turn (the same), gathers from the start (5.4 / 7.6 ms).)
template <typename T, int DMODE, bool INFIRST, int LPR>
__global__ __launch_bounds__(64) void k_trsv_sf(int nunits, const v4i32* __restrict__ uinfo, const int* __restrict__ pinfo,
                                                const int* __restrict__ ecol, const T* __restrict__ eval,
                                                const T* __restrict__ gcoef, const T* __restrict__ diag, const T* __restrict__ rdiag,
                                                const T* __restrict__ rhs_src, const int* __restrict__ rhs_idx, T* w,
                                                T* __restrict__ out, const int* __restrict__ order, int poll_cap, int flags, int stagger,
                                                const int* __restrict__ ufar, unsigned long long* __restrict__ dbg, unsigned* tickets)
{
    // Units are taken by TICKET, so that a wave only ever waits for units held by waves that are running -- whatever share of the
    // device this launch gets (a static round-robin over the grid would wait for waves that may never become resident next to
    // another process's kernel).  One counter word serves ~88 tickets per microsecond and this solve wants ~75: kSfStreams words (own
    // 128-byte lines), unit u belongs to stream u % kSfStreams, a wave is bound to the stream its START ticket names (the first
    // kSfStreams waves to run cover every stream) and asks for its next unit while it works on this one.
    // (dbg: RAMD_TRSV_SF_DBG, two timestamps per unit -- dependencies there, result published)
    // flags: 1 = no dependency waits (diagnostic, wrong results), 2 = two generations of requests in flight
    const int lane = threadIdx.x, slot = lane / LPR, l = lane % LPR;
    if(dbg && blockIdx.x == 0 && lane == 0) // (shader clock against the 100 MHz counter: what a cycle is worth in this kernel)
    {
        dbg[2 * (int64_t)nunits]     = clock64();
        dbg[2 * (int64_t)nunits + 1] = wall_clock64();
    }
    unsigned start = 0;
    if(lane == 0)
        start = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int stream = (int)((unsigned)__builtin_amdgcn_readfirstlane((int)start) % (unsigned)kSfStreams);
    unsigned* const my_counter = tickets + 32 * (1 + stream);
    unsigned  tk = 0;
    if(lane == 0)
        tk = __hip_atomic_fetch_add(my_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for(;;)
    {
        const long long uu = (long long)stream + (long long)kSfStreams * (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
        if(uu >= nunits)
            break;
        const int u = (int)uu;
        if(lane == 0) // (the next unit: asked for now, looked at when this one is done)
            tk = __hip_atomic_fetch_add(my_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const v4i32 ui  = uinfo[u];
        const int   p0  = __builtin_amdgcn_readfirstlane(ui.x);
        const int   w1  = __builtin_amdgcn_readfirstlane(ui.y);
        const int   cnt = w1 & 255, kw = (w1 >> 8) & 15;
        // lanes per row in use in this unit (a row's entries fill its lanes from lane 0 on), one group only, rows of the longest group
        const int  nl   = (w1 >> 12) & 15;
        const bool one  = ((w1 >> 16) & 1) != 0;
        const int  maxm = (w1 >> 17) & 15;
        const bool  have = slot < cnt, act = have && l < nl;
        const int   p    = p0 + (have ? slot : 0);
        const int   ps   = cnt * nl;
        int         c[kSfKW];
        T           a[kSfKW];
        const int64_t e0 = (int64_t)__builtin_amdgcn_readfirstlane(ui.z) + slot * nl + l;
#pragma unroll
        for(int k = 0; k < kSfKW; ++k)
        {
            c[k] = -1;
            a[k] = (T)0;
            if(k < kw && act)
            {
                c[k] = nt_load(ecol + e0 + (int64_t)k * ps);
                a[k] = nt_load(eval + e0 + (int64_t)k * ps);
            }
        }
        const int info = pinfo[p];
        const int r    = have ? (info & 15) : 0;
        const int m    = have ? ((info >> 4) & 15) : 0;
        const T   rhs  = rhs_src[rhs_idx[p]];
        const T   dg   = (DMODE == 0) ? (T)1 : diag[p];
        T         rdg  = (T)0;
        if constexpr(DMODE == 1 && sizeof(T) == 8)
            rdg = rdiag[p];
        const int onat = out ? order[p] : 0;
        const bool grouped = __ballot(m > 1) != 0ull;
        T          gc[7];
#pragma unroll
        for(int j = 0; j < 7; ++j)
            gc[j] = (T)0;
        if(grouped)
        {
#pragma unroll
            for(int j = 0; j < 7; ++j)
                gc[j] = gcoef[(int64_t)p * 8 + j];
        }
        // one request per turn while the front is levels away
        const int far = (flags & 1) ? -1 : __builtin_amdgcn_readfirstlane(ufar[u]);
        if(far >= 0)
        {
            int spins = 0, backoff = 1;
            while(poll_load(w + far) == Sentinel<T>::value)
            {
                spin_guard(spins);
                backoff = poll_backoff(false, backoff, poll_cap);
            }
        }
        // (straight-line bodies for 2 / 3 / 4 / 6 subtractions per lane, one group or several: chosen per unit)
#define SF_BODY(NA_)                                                                                                               \
    do                                                                                                                             \
    {                                                                                                                              \
        if(one)                                                                                                                    \
            sf_unit<T, DMODE, INFIRST, LPR, NA_, true>(w, out, p, onat, p0, c, a, rhs, dg, rdg, gc, have, r, l, slot, nl, maxm, flags, \
                                                       stagger, dbg, (int64_t)nunits, u);                                         \
        else                                                                                                                       \
            sf_unit<T, DMODE, INFIRST, LPR, NA_, false>(w, out, p, onat, p0, c, a, rhs, dg, rdg, gc, have, r, l, slot, nl, maxm, flags, \
                                                        stagger, dbg, (int64_t)nunits, u);                                        \
    } while(0)
        if(kw <= 2)
            SF_BODY(2);
        else if(kw == 3)
            SF_BODY(3);
        else if(kw == 4)
            SF_BODY(4);
        else
            SF_BODY(kSfKW);
#undef SF_BODY
    }
    if(dbg && blockIdx.x == 0 && lane == 0)
    {
        dbg[2 * (int64_t)nunits + 2] = clock64();
        dbg[2 * (int64_t)nunits + 3] = wall_clock64();
    }
}

template <typename T>
int sf_run(const SfPlan* S, int n, int dm, const T* diag, T* w, const int* order, const T* rhs_src, const int* rhs_idx, T* out)
{
    Backend& b = backend();
    hipLaunchKernelGGL((k_fill_sentinel<T>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, (int64_t)n, w);
    // persistent waves, as many as the device holds (units by ticket: see k_trsv_sf)
    static const int waves_env = getenv("RAMD_TRSV_SF_WAVES") ? atoi(getenv("RAMD_TRSV_SF_WAVES")) : 0; // (per CU; experiments)
    static const int cap_env   = getenv("RAMD_TRSV_SF_POLLCAP") ? atoi(getenv("RAMD_TRSV_SF_POLLCAP")) : 8;
    static const int gat_env   = getenv("RAMD_TRSV_SF_GATHER") ? atoi(getenv("RAMD_TRSV_SF_GATHER")) : 4; // (2: no waits, diagnostic)
    // two generations of requests in flight per waiting unit, the second `stagger` x 64 cycles behind the first (0: one generation)
    // (measured on the RCM shell, lower / upper ms per triangle: 0: 2.95 / 4.38, 2 to 8: 2.91 / 4.55, 24: 3.02 / 4.54 -- what the
    //  second generation gains in phase it loses in the consumer CU's memory queue; off by default)
    static const int stag_env  = getenv("RAMD_TRSV_SF_STAGGER") ? atoi(getenv("RAMD_TRSV_SF_STAGGER")) : 0;
    const int        flags     = (gat_env == 2 ? 1 : 0) | (stag_env > 0 ? 2 : 0);
    // ticket words of the launch: [0] start tickets, [32 (1 + s)] units of stream s -- the plan's own, zeroed before every launch
    RAMD_HIP(hipMemsetAsync(S->tickets, 0, sizeof(unsigned) * 32 * (1 + kSfStreams), b.cur));
    unsigned nwg = 0;
    // RAMD_TRSV_SF_DBG=<file prefix> (tools/ diagnostics): two timestamps per unit, dumped after every solve with the unit table
    static const char*  dbg_path = getenv("RAMD_TRSV_SF_DBG");
    unsigned long long* dbg      = nullptr;
    if(dbg_path)
    {
        RAMD_HIP(hipMalloc(&dbg, sizeof(unsigned long long) * (4 * (size_t)S->nunits + 12)));
        RAMD_HIP(hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * (4 * (size_t)S->nunits + 12), b.cur));
    }
#define TRSV_SF(DM, INF, LP)                                                                                                  \
    do                                                                                                                        \
    {                                                                                                                         \
        static int occ_max = 0;                                                                                               \
        if(occ_max == 0)                                                                                                      \
        {                                                                                                                     \
            int nb_cu = 0;                                                                                                    \
            RAMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_cu, k_trsv_sf<T, DM, INF, LP>, 64, 0));                \
            occ_max = nb_cu < 1 ? 1 : (nb_cu - 2 > 8 ? 8 : (nb_cu > 2 ? nb_cu - 2 : 1));                                      \
        }                                                                                                                     \
        /* waves per CU: a narrow graph is bound by its chain of hand-offs and runs best with few waves around it (1 / 2 / 4 / */ \
        /* 6 / 8 per CU on the RCM shell: 5.3 / 4.3 / 4.4 / 4.4 / 4.5 ms per triangle), a wide one by its throughput (the     */ \
        /* random numbering, 16 700 units per level: 1.10 ms with 4 per CU, 0.67 with 8)                                      */ \
        int occ = ((int64_t)S->nunits >= (int64_t)1024 * S->nglev) ? occ_max : (occ_max < 4 ? occ_max : 4);                   \
        if(waves_env > 0 && waves_env < occ_max)                                                                              \
            occ = waves_env;                                                                                                  \
        const int64_t cap = (int64_t)occ * b.num_cu;                                                                          \
        nwg               = (unsigned)(S->nunits < cap ? S->nunits : cap);                                                    \
        hipLaunchKernelGGL((k_trsv_sf<T, DM, INF, LP>), dim3(nwg), dim3(64), 0, b.cur, S->nunits, (const v4i32*)S->uinfo,     \
                           S->pinfo, S->ecol, (const T*)S->eval, (const T*)S->gcoef, diag, (const T*)S->rdiag, rhs_src, rhs_idx, \
                           w, out, order, cap_env, flags, stag_env, S->ufar, dbg, S->tickets);                                \
    } while(0)
#define TRSV_SF_L(DM, INF)      \
    do                          \
    {                           \
        if(S->lpr == 4)         \
            TRSV_SF(DM, INF, 4); \
        else                    \
            TRSV_SF(DM, INF, 8); \
    } while(0)
#define TRSV_SF_I(DM)              \
    do                             \
    {                              \
        if(S->infirst)             \
            TRSV_SF_L(DM, true);   \
        else                       \
            TRSV_SF_L(DM, false);  \
    } while(0)
    prof_begin(RAMD_PROF_TRSV, b.cur);
    if(dm == 0)
        TRSV_SF_I(0);
    else if(dm == 1)
        TRSV_SF_I(1);
    else
        TRSV_SF_I(2);
    prof_end(RAMD_PROF_TRSV, b.cur);
#undef TRSV_SF_I
#undef TRSV_SF_L
#undef TRSV_SF
    RAMD_HIP(hipGetLastError());
    if(dbg)
    {
        std::vector<unsigned long long> ht(4 * (size_t)S->nunits + 12);
        std::vector<int>                hu(4 * (size_t)S->nunits), hp((size_t)n);
        RAMD_HIP(hipMemcpy(ht.data(), dbg, sizeof(unsigned long long) * ht.size(), hipMemcpyDeviceToHost));
        RAMD_HIP(hipMemcpy(hu.data(), S->uinfo, sizeof(int) * hu.size(), hipMemcpyDeviceToHost));
        RAMD_HIP(hipMemcpy(hp.data(), S->punit, sizeof(int) * hp.size(), hipMemcpyDeviceToHost));
        (void)hipFree(dbg);
        const std::string fn = std::string(dbg_path) + (S->infirst ? "_upper.bin" : "_lower.bin");
        if(FILE* f = fopen(fn.c_str(), "wb"))
        {
            const int hdr[4] = {S->nunits, n, (int)nwg, S->lpr};
            fwrite(hdr, sizeof(int), 4, f);
            fwrite(ht.data(), sizeof(unsigned long long), 2 * (size_t)S->nunits, f);
            fwrite(hu.data(), sizeof(int), hu.size(), f);
            fwrite(hp.data(), sizeof(int), hp.size(), f);
            fwrite(ht.data() + 2 * (size_t)S->nunits, sizeof(unsigned long long), 4 + 2 * (size_t)S->nunits, f);
            fclose(f);
            fprintf(stderr, "k_trsv_sf (%s): %llu wave-level divisions took `/` (outside the window of the short sequence), %d units\n",
                    S->infirst ? "upper" : "lower", ht[4 * (size_t)S->nunits + 4], S->nunits);
        }
    }
    return RAMD_OK;
}
template int sf_run<double>(const SfPlan*, int, int, const double*, double*, const int*, const double*, const int*, double*);
template int sf_run<float>(const SfPlan*, int, int, const float*, float*, const int*, const float*, const int*, float*);

} // namespace ramd

// a probe for the tests: fast[i] = sf_div(a[i], d[i]) (the plan-time reciprocal iterate formed on the way), plain[i] = a[i] / d[i],
// in_window[i] = 1 where the short sequence was what ran.  Device pointers.
extern "C" int ramd_selftest_sf_div(long long n, const double* a, const double* d, double* fast, double* plain, int* in_window)
{
    using namespace ramd;
    if(n < 0 || (n > 0 && (!a || !d || !fast || !plain || !in_window)))
        RAMD_FAIL(RAMD_ERR_ARG, "selftest_sf_div: bad arguments");
    if(n == 0)
        return RAMD_OK;
    Backend& b = backend();
    hipLaunchKernelGGL(k_sf_div_probe, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, (int64_t)n, a, d, fast, plain, in_window);
    RAMD_HIP(hipGetLastError());
    RAMD_HIP(hipStreamSynchronize(b.cur));
    return RAMD_OK;
}
