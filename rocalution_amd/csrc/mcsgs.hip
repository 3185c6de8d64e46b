// mcsgs.hip -- multi-coloured symmetric Gauss-Seidel apply as 2*nb-1 fused colour sweeps.
//
// Reference (decomposed form): src/solvers/preconditioners/preconditioner_multicolored.cpp:348-413 and
// preconditioner_multicolored_gs.cpp:127-199 -- per apply, for nb colours:
//   x = P rhs ; slice copies ; for i: [for j<i: x_i += -1*A_ij x_j] ; x_i *= Dinv_i      (SolveL_)
//   x_i *= D_i (SolveD_) ; for i desc: [for j>i DESC: x_i += -1*A_ij x_j] ; x_i *= Dinv_i (SolveR_)
//   gather slices ; x = P^T x_
// i.e. 1 + nb + (nb(nb-1)/2 + nb) + nb + (nb(nb-1)/2 + nb) + nb + 1 kernels (14 for 2 colours), each a
// separate pass over memory.  Here every row's whole L-part (resp. U-part) is one sweep: the permuted
// matrix is split ONCE into a strictly-lower and a strictly-upper wave-sliced ELL (64 rows per slice,
// column-major => coalesced), U entries stored in the reference's accumulation order (colour block
// descending, column ascending), the permutation gather/scatter and the three diagonal scalings are
// folded into the sweeps.  Arithmetic per row is the reference's, operation for operation
// (x + (-1*a)*y == x - a*y exactly), so results are bit-identical to the block form.
#include <atomic>
#include "device_utils.hpp"
#include "matrix_impl.hpp"

#include <algorithm>
#include <vector>

namespace ramd
{

struct McsgsPlan
{
    int   dtype = RAMD_F64;
    int   n = 0, nb = 0;
    std::vector<int> off; // [nb+1] colour block offsets (positions in the permuted order)
    std::vector<char> identity; // block (i,i) empty -> Jacobi::Solve is the identity
    int*  iperm   = nullptr; // [n] position -> original row
    int*  blk_of  = nullptr; // [n] colour of a position
    void* d       = nullptr; // [n] diag_block_ (0 where the row stores no diagonal)
    void* dinv    = nullptr; // [n] Jacobi inverse diagonal (1 for a zero diagonal, 0 where none is stored)
    void* xp      = nullptr; // [n] permuted work vector
    // sliced ELL of the strictly lower / upper colour parts (slice = 64 consecutive positions)
    int * l_off = nullptr, *l_col = nullptr, *u_off = nullptr, *u_col = nullptr;
    void *l_val = nullptr, *u_val = nullptr;
    // row patterns of the two parts (spmv.hip: sell_analyse_pattern): the sweeps of a structured operator rebuild their
    // columns from one byte per row instead of reading four per slot
    int            l_pat = 0, u_pat = 0, l_pat_n = 0, u_pat_n = 0; // state: 1 usable
    unsigned char *l_pat_id = nullptr, *u_pat_id = nullptr;
    int *          l_pat_dict = nullptr, *u_pat_dict = nullptr;
    // order in which the workgroups of a colour sweep take the 256-row blocks of the colour (device_utils.hpp: every XCD a
    // contiguous eighth, walked tile by tile through the planes of a far band): per colour, for the L and the U part
    std::vector<BandMap> l_bm, u_bm;
    // Output pairs.  out[iperm[t]] of a colour's rows is a store of every nb-th element: 16 bytes leave the L2 for 8 (counters:
    // WRITE_SIZE 24 bytes per row where xp and out are written).  The LAST sweep of an apply (colour 0) therefore also
    // stores the neighbour in the aligned 16-byte pair of its row where that neighbour belongs to another colour -- final
    // by then, in xp -- and that row's own sweep leaves its store out:
    //   pair_of [t], t < off[1]   : position of the row that shares the 16-byte pair of out with row t, or -1
    //   covered [t], t >= off[1]  : 1 = out of this row is stored by its partner's sweep
    int*           pair_of = nullptr;
    unsigned char* covered = nullptr;
    // Colour 0 folded away (SGS apply, structured operators).  The first sweep of an apply has no entries -- colour 0 has no
    // lower colours -- and only scales: x_0 = Dinv_0 rhs_0, 36 bytes per row of the colour and a launch to leave 8 behind.  Its
    // readers form that product themselves instead: the forward sweeps of the other colours at their gathers (rhs and the
    // inverse diagonal in NATURAL order, where a row's neighbours share lines with the row itself), colour 0's backward
    // sweep for its own row.  Same operands, same product: bit-identical.  Needs the natural-order offset of every slot of
    // the lower part's row patterns to be one number per pattern (nat_dict; verified row by row at Build()).
    bool  fold0    = false;
    void* dinv_nat = nullptr; // [n] inverse diagonal at the ORIGINAL row index (rows of colour 0)
    int*  nat_dict = nullptr; // [l_pat_n * kPatMaxW] original-index offset of a slot whose column has colour 0
    // Red-black lattice form (k_mc_rb, SGS applies): the operator is a 7- / 5-point lattice operator and the two colours are the
    // two parities of x + y + z.  ONE pass over both colours instead of a sweep per colour: see k_mc_rb.
    bool  rb = false;
    int   rb_nx = 0, rb_ny = 0, rb_nz = 0, rb_p0 = 0; // rb_p0: parity of the cells of colour 0
    void* rb_val[2] = {nullptr, nullptr}; // per colour: [6][lines * hx] off-diagonal values, slot = ascending column offset
    void* rb_d[2]   = {nullptr, nullptr}; // per colour: [lines * hx] diagonal / inverse diagonal
    void* rb_di[2]  = {nullptr, nullptr};
    void  release()
    {
        dev_free(&pair_of);
        dev_free(&covered);
        dev_free(&nat_dict);
        if(dinv_nat)
            (void)cached_free(dinv_nat);
        dinv_nat = nullptr;
        fold0    = false;
        for(int c = 0; c < 2; ++c)
        {
            void** ps2[] = {&rb_val[c], &rb_d[c], &rb_di[c]};
            for(void** q : ps2)
            {
                if(*q)
                    (void)cached_free(*q);
                *q = nullptr;
            }
        }
        rb = false;
        dev_free(&iperm);
        dev_free(&blk_of);
        dev_free(&l_off);
        dev_free(&l_col);
        dev_free(&u_off);
        dev_free(&u_col);
        dev_free(&l_pat_id);
        dev_free(&u_pat_id);
        dev_free(&l_pat_dict);
        dev_free(&u_pat_dict);
        l_pat = u_pat = 0;
        void** ps[] = {&d, &dinv, &xp, &l_val, &u_val};
        for(void** p : ps)
        {
            if(*p)
                (void)cached_free(*p);
            *p = nullptr;
        }
    }
};

__global__ __launch_bounds__(kBlock) void k_mc_prepare(int n, int nb, const int* __restrict__ off,
                                                       const int* __restrict__ perm,
                                                       int* __restrict__ iperm, int* __restrict__ blk_of)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
    {
        iperm[perm[i]] = (int)i;
        int b = 0;
        while(b + 1 < nb && (int)i >= off[b + 1])
            ++b;
        blk_of[i] = b;
    }
}

// per position: entries strictly below / above its colour block; per slice: 64 * max
template <bool LOWER>
__global__ __launch_bounds__(kBlock) void k_mc_width(int n, const int* __restrict__ rp,
                                                     const int* __restrict__ ci,
                                                     const int* __restrict__ off,
                                                     const int* __restrict__ blk_of,
                                                     int* __restrict__ slice_w)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int           c = 0;
    if(t < n)
    {
        const int b  = blk_of[t];
        const int lo = off[b], hi = off[b + 1];
        for(int j = rp[t]; j < rp[t + 1]; ++j)
            if(LOWER ? (ci[j] < lo) : (ci[j] >= hi))
                ++c;
    }
#pragma unroll
    for(int o = 32; o > 0; o >>= 1)
        c = max(c, __shfl_xor(c, o, 64));
    // (an even number of slots: the sweeps read the values of two slots of a row with one 16-byte access)
    if((threadIdx.x & 63) == 0 && (t >> 6) * 64 < n)
        slice_w[t >> 6] = ((c + 1) & ~1) * 64;
}

// where slot k of lane `lane` sits in a slice's piece of the value array: slots in pairs, {2j, 2j + 1} of a row next to each other
// (one 16-byte access per pair: 8-byte accesses stream at 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md), lanes
// consecutive inside a pair.  The column array keeps the slot-major layout (the row-pattern analysis reads it).
__device__ __forceinline__ int mc_val_at(int k, int lane)
{
    return (k >> 1) * 128 + lane * 2 + (k & 1);
}

// fill; also extracts diag / inverse diagonal (host_matrix_csr.cpp:772-845 semantics)
template <typename T, bool LOWER>
__global__ __launch_bounds__(kBlock) void k_mc_fill(int n, int nb, const int* __restrict__ rp,
                                                    const int* __restrict__ ci,
                                                    const T* __restrict__ val,
                                                    const int* __restrict__ off,
                                                    const int* __restrict__ blk_of,
                                                    const int* __restrict__ slice_off,
                                                    int* __restrict__ ecol, T* __restrict__ eval,
                                                    T* __restrict__ d, T* __restrict__ dinv)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n)
        return;
    const int b    = blk_of[t];
    const int lane = (int)(t & 63);
    const int base = slice_off[t >> 6];
    const int w    = (slice_off[(t >> 6) + 1] - base) >> 6;
    const int rs = rp[t], re = rp[t + 1];
    int       k = 0;
    if(LOWER)
    {
        for(int j = rs; j < re && ci[j] < off[b]; ++j, ++k) // ascending columns == ascending blocks
        {
            ecol[base + k * 64 + lane] = ci[j];
            eval[base + mc_val_at(k, lane)] = val[j];
        }
        // diagonal of the colour block (first match wins)
        T dv = (T)0, iv = (T)0;
        for(int j = rs; j < re; ++j)
            if(ci[j] == (int)t)
            {
                dv = val[j];
                iv = (dv != (T)0) ? (T)1 / dv : (T)1;
                break;
            }
        d[t]    = dv;
        dinv[t] = iv;
    }
    else
    {
        // reference order of SolveR_: colour blocks DESCENDING, columns ascending inside a block
        int hi_end = re;
        for(int bb = nb - 1; bb > b; --bb)
        {
            int lo = hi_end;
            while(lo > rs && ci[lo - 1] >= off[bb])
                --lo;
            for(int j = lo; j < hi_end; ++j, ++k)
            {
                ecol[base + k * 64 + lane] = ci[j];
                eval[base + mc_val_at(k, lane)] = val[j];
            }
            hi_end = lo;
        }
    }
    for(; k < w; ++k)
    {
        ecol[base + k * 64 + lane] = -1;
        eval[base + mc_val_at(k, lane)] = (T)0;
    }
}

// one colour sweep over positions [p0, p1)
//   FROM_RHS : s = rhs[iperm[t]]          (L sweep: first touch of the row; folds x = P rhs)
//   else     : s = xp[t]
//   MULT_D   : s = s * d[t]               (SolveD_)           -- before the U entries
//   entries  : s = s - a * xp[col]        (ApplyAdd with scalar -1, storage order)
//   !ident   : s = s * dinv[t]            (diag_solver_[i]->Solve in place)
//   BOTH     : last colour: L sweep, D and R sweep of the same row in one go (it has no U part)
//   TO_OUT   : out[iperm[t]] = s          (folds x = P^T x_)
//   PAT      : the columns come from the row-pattern dictionary (one byte per row; -1 slots are dictionary entries too)
//   keep_xp  : 0 where no later sweep reads this colour's xp (the last sweep of an apply): the store is left out
// Block order: a row of colour i gathers xp of OTHER colours at the positions of its grid neighbours -- under the colour
// permutation the rows of one colour keep their natural order, so those are the same three "planes" a stencil product
// gathers, at half the pitch.  With workgroup b on block b every line of xp was fetched by three XCDs (blocks b, b +- 1
// sit on different XCDs; counters: 24 instead of 8 bytes per row); the XCD- and band-aware order of the CSR product
// (xcd_block) keeps a line's readers on one L2.
//   FOLD     : colour 0 is not in xp (McsgsPlan::fold0): a gathered column of colour 0 is rhs * dinv at its original index,
//              the own value of a colour-0 row (its backward sweep) rhs[orow] * dinv[t]
template <typename T, bool FROM_RHS, bool MULT_D, bool BOTH, bool TO_OUT, bool PAT, bool FOLD = false>
__global__ __launch_bounds__(kBlock) void k_mc_sweep(int p0, int p1, const int* __restrict__ slice_off,
                                                     const int* __restrict__ ecol,
                                                     const T* __restrict__ eval,
                                                     const T* __restrict__ d, const T* __restrict__ dinv,
                                                     const int* __restrict__ iperm,
                                                     const T* __restrict__ rhs, T* xp,
                                                     T* __restrict__ out, int identity, CsrPattern pat, int nblk, int per_xcd,
                                                     BandMap bm, int keep_xp, const int* __restrict__ pair_of,
                                                     const unsigned char* __restrict__ covered, int n0 = 0,
                                                     const int* __restrict__ nat_dict = nullptr,
                                                     const T* __restrict__ dinv_nat = nullptr)
{
    static_assert(!FOLD || PAT, "the natural-order offsets belong to the row patterns");
    // (bm.W < 0: workgroup b takes block b -- the order of rounds 1 to 3, kept for A/B runs)
    const int blk = bm.W < 0 ? ((int)blockIdx.x < nblk ? (int)blockIdx.x : -1) : xcd_block(nblk, per_xcd, bm); // (uniform)
    if(blk < 0)
        return;
    __shared__ int sdict[PAT ? kPatMax * kPatMaxW : 1];
    __shared__ int snat[(FOLD && !MULT_D) ? kPatMax * kPatMaxW : 1];
    // Everything a row needs that hangs on nothing but its position is requested BEFORE the dictionary is staged (the
    // barrier would otherwise keep these loads behind the dictionary's round trip): the chain of a workgroup is then
    // {dictionary, row data} -> {values, gathers} -> store instead of dictionary -> row data -> values -> store.
    constexpr int NDW = PAT ? kPatMax * kPatMaxW / kBlock : 0;
    int           dreg[NDW > 0 ? NDW : 1], nreg[NDW > 0 ? NDW : 1];
#pragma unroll
    for(int q = 0; q < NDW; ++q)
    {
        const int i = q * kBlock + threadIdx.x;
        dreg[q]     = i < pat.n * kPatMaxW ? pat.dict[i] : 0;
        if(FOLD && !MULT_D)
            nreg[q] = i < pat.n * kPatMaxW ? nat_dict[i] : 0;
    }
    const int64_t t    = (int64_t)p0 + (int64_t)blk * kBlock + threadIdx.x;
    const bool    live = t < p1;
    const int64_t tl   = live ? t : (int64_t)p1 - 1; // (lanes beyond the colour read what its last row reads)
    const int     pid  = PAT ? (int)pat.id[tl] : 0;
    const int     lane = (int)(tl & 63);
    const int     base = slice_off[tl >> 6];
    const int     w    = (slice_off[(tl >> 6) + 1] - base) >> 6;
    const int     orow = (FROM_RHS || TO_OUT || FOLD) ? iperm[tl] : 0;
    const T       dv   = (MULT_D || BOTH) ? d[tl] : (T)0;
    const T       iv   = (!identity || (FOLD && MULT_D)) ? dinv[tl] : (T)0;
    const int     pq   = (TO_OUT && pair_of) ? pair_of[tl] : -1;
    const bool    skip_out = TO_OUT && !pair_of && covered && covered[tl];
    T             s    = (FROM_RHS || (FOLD && MULT_D)) ? rhs[orow] : xp[tl];
    // ... and the values of the first batch of slots (they hang on the slice offset only)
    using P2 = T __attribute__((ext_vector_type(2)));
    T a0[8];
#pragma unroll
    for(int e = 0; e < 8; e += 2)
        if(e < w)
        {
            const P2 av = nt_load(reinterpret_cast<const P2*>(eval + base + mc_val_at(e, lane)));
            a0[e]       = av.x;
            a0[e + 1]   = av.y;
        }
    if(PAT)
    {
#pragma unroll
        for(int q = 0; q < NDW; ++q)
        {
            const int i = q * kBlock + threadIdx.x;
            if(i < pat.n * kPatMaxW)
            {
                sdict[i] = dreg[q];
                if(FOLD && !MULT_D)
                    snat[i] = nreg[q];
            }
        }
        __syncthreads();
    }
    if(!live)
        return;
    const int dbase = pid * kPatMaxW;
    if(FOLD && MULT_D)
        s = s * iv; // the forward value of this colour-0 row, formed here instead of read
    if(MULT_D)
        s = s * dv;
    // masked batches of 8 slots: loads, then gathers, then the updates IN ORDER (padding col = -1 ends a row)
    bool done = false;
    for(int k0 = 0; k0 < w && !done; k0 += 8)
    {
        int c[8];
        T   a[8], xv[8];
#pragma unroll
        for(int e = 0; e < 8; e += 2) // (w is even: slots come in pairs, mc_val_at)
            if(k0 + e < w)
            {
                if(k0 == 0)
                {
                    a[e]     = a0[e];
                    a[e + 1] = a0[e + 1];
                }
                else
                {
                    const P2 av = nt_load(reinterpret_cast<const P2*>(eval + base + mc_val_at(k0 + e, lane)));
                    a[e]        = av.x;
                    a[e + 1]    = av.y;
                }
            }
#pragma unroll
        for(int e = 0; e < 8; ++e)
        {
            c[e] = -1;
            if(k0 + e < w)
            {
                if(PAT)
                {
                    const int o = sdict[dbase + k0 + e];
                    c[e]        = o == kPatEnd ? -1 : (int)t + o;
                }
                else
                    c[e] = nt_load(ecol + base + (k0 + e) * 64 + lane);
            }
        }
#pragma unroll
        for(int e = 0; e < 8; ++e)
        {
            if(c[e] < 0)
                done = true;
            if(!done)
            {
                if(FOLD && !MULT_D && c[e] < n0)
                {
                    const int o = orow + snat[dbase + k0 + e];
                    xv[e]       = rhs[o] * dinv_nat[o];
                }
                else
                    xv[e] = xp[c[e]];
            }
            else
                c[e] = -1;
        }
#pragma unroll
        for(int e = 0; e < 8; ++e)
            if(c[e] >= 0)
                s -= a[e] * xv[e];
    }
    if(!identity)
        s = s * iv;
    if(BOTH)
    {
        s = s * dv;
        if(!identity)
            s = s * iv;
    }
    if(keep_xp)
        xp[t] = s;
    if(TO_OUT)
    {
        if(pair_of) // (the last sweep: rows of colour 0)
        {
            if(pq >= 0)
            {
                using P2 = T __attribute__((ext_vector_type(2)));
                const T other = xp[pq]; // final: every other colour is done
                P2      v;
                v.x = (orow & 1) ? other : s;
                v.y = (orow & 1) ? s : other;
                *reinterpret_cast<P2*>(out + (orow & ~1)) = v;
            }
            else
                out[orow] = s;
        }
        else if(!skip_out)
            out[orow] = s;
    }
}

// Measured and removed (round 4): TWO ROWS PER THREAD for these sweeps (pattern numbers, permutation, diagonal, inverse, work
// values and the slot pairs of both rows read with accesses of twice the width, twice the gathers in flight per thread --
// the step that took the ELL product from 2.15 to 1.85 ms, k_ell2): MC-SGS apply 3.30-3.34 ms against 3.17 ms in
// alternating runs (gpurun_out/r04v).  The sweeps' values already come in 16-byte slot pairs; what the ELL product gained
// was exactly that.

// natural-order offsets of the slots of the lower part's patterns that point into colour 0 (McsgsPlan::fold0):
// pass 0: the first row of a pattern to arrive writes iperm[col] - iperm[row]; pass 1: every row checks its own against it
constexpr int kNatNone = -2147483647 - 1;
__global__ __launch_bounds__(kBlock) void k_mc_nat_dict(int p0, int n, int n0, int pass, const unsigned char* __restrict__ pid,
                                                        const int* __restrict__ dict, const int* __restrict__ iperm,
                                                        int* nat, int* __restrict__ bad)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
    {
        const int base = (int)pid[t] * kPatMaxW;
        const int me   = iperm[t];
        for(int k = 0; k < kPatMaxW; ++k)
        {
            const int o = dict[base + k];
            if(o == kPatEnd)
                break;
            const int c = (int)t + o;
            if(c < 0 || c >= n0)
                continue; // (a column of another colour: read from the work vector as before)
            const int dlt = iperm[c] - me;
            if(pass == 0)
            {
                // (one row per pattern slot writes; the others see the word filled and leave it alone: 67 M compare-and-swaps
                //  on a few dozen words took 1.2 s per launch at 512^3)
                if(__hip_atomic_load(nat + base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kNatNone)
                    atomicCAS(nat + base + k, kNatNone, dlt);
            }
            else if(__hip_atomic_load(nat + base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != dlt)
                *bad = 1;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mc_dinv_nat(int n0, const int* __restrict__ iperm, const T* __restrict__ dinv,
                                                        T* __restrict__ dinv_nat)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n0; t += gsz)
        dinv_nat[iperm[t]] = dinv[t];
}

// ======================================================================= red-black lattice form
// A two-colour SGS apply moves every natural-order stream (rhs, d, 1/d, out) twice at half density -- once per colour -- and
// sends the second colour's values through memory between the sweeps: 15 GB per apply at 512^3 for values 48 + rhs 8 + d 8 +
// 1/d 8 + out 8 = 80 bytes per row = 10.7 GB.  Where the operator is a 7- / 5-point lattice operator (every row has exactly the
// neighbours r -+ 1, r -+ nx, r -+ nx ny its position allows) and the colours are the parities of x + y + z, ONE pass does it:
// a workgroup owns a 128 x 16 (x, y) tile and a run of z planes and marches along z; per plane z it
//   A. reads rhs of plane z + 2 on the tile + 2 cells around it: colour-0 cells become x0 = rhs * dinv, colour-1 cells keep rhs (LDS)
//   B. forms the colour-1 values of plane z + 1 on the tile + 1 cell: s = rhs - sum a_k x0_k (ascending columns), s *= dinv, d, dinv
//   C. forms the colour-0 values of plane z: s = x0 * d - sum a_k y1_k, s *= dinv; stores the plane's rows of out, both colours
// with three planes of each kind in LDS.  The one-cell ring of colour-1 values around the tile is recomputed by every tile that
// needs it (they depend on x0 only, which is elementwise): 14 % more colour-1 rows, no exchange between workgroups.  The tile is
// WIDE because the arrays are read in 128-byte lines: a 32-wide tile reads one line per array and row and a second one for its
// single ring cell of colour 1 (measured: 32 x 32 3.04 ms, 64 x 16 2.57, 128 x 8 2.43, 128 x 16 2.24 ms per apply at 512^3).
// The operations per row are those of the two sweeps (preconditioner_multicolored_gs.cpp:127-215) in their order; the
// values are stored per colour in the order of the cells along x (cell x of a line at x / 2), so that both colours read
// whole lines.  A cell the lattice does not have contributes no term (masked, not multiplied by zero).
constexpr int kRbTX = 128, kRbTY = 16, kRbAX = kRbTX + 4, kRbAY = kRbTY + 4, kRbBX = kRbTX + 2, kRbBY = kRbTY + 2, kRbPX = kRbAX + 1,
              kRbPY = kRbBX + 1, kRbChunkMax = 64, kRbChunkMin = 16, kRbThreads = 512;
struct RbDims
{
    int nx, ny, nz, hx, p0, tiles_x, tiles_y, chunks, chunk;
};
// The global loads of a stage do not depend on what the stages before it left in LDS, so they are issued one plane AHEAD, into
// registers, right after the registers' previous contents were consumed: while a plane is computed, the loads of the next one
// are in flight (a workgroup has 60 KB of LDS, two fit a CU -- occupancy alone would not hide three dependent round trips per
// plane).  The barriers wait for LDS only (rb_barrier: __syncthreads() would drain the loads in flight as well).
__device__ __forceinline__ int rb_clamp(int v, int hi)
{
    return min(max(v, 0), hi);
}
__device__ __forceinline__ void rb_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <typename T>
__global__ __launch_bounds__(kRbThreads) void k_mc_rb(RbDims g, const T* __restrict__ val0, const T* __restrict__ val1,
                                               const T* __restrict__ d0, const T* __restrict__ di0, const T* __restrict__ d1,
                                               const T* __restrict__ di1, const T* __restrict__ rhs, T* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char rb_lds[];
    T* XR = reinterpret_cast<T*>(rb_lds); // [3][kRbAY][kRbPX]: x0 on colour-0 cells, rhs on colour-1 cells
    T* Y1 = XR + 3 * kRbAY * kRbPX; // [3][kRbBY][kRbPY]: finished colour-1 values
    constexpr int kAC = kRbAX * kRbAY, kBH = kRbBX / 2, kBC = kRbBY * kBH, kCH = kRbTX / 2, kCC = kRbTY * kCH;
    constexpr int NA = (kAC + kRbThreads - 1) / kRbThreads, NB = (kBC + kRbThreads - 1) / kRbThreads, NC = (kCC + kRbThreads - 1) / kRbThreads;
    const int tid = threadIdx.x;
    // workgroup w runs on XCD w % 8: every XCD gets a contiguous run of tiles (the rings of neighbouring tiles meet in one L2)
    int wg = blockIdx.x;
    if(gridDim.x % 8 == 0)
        wg = (wg % 8) * (int)(gridDim.x / 8) + wg / 8;
    const int tx = wg % g.tiles_x, ty = (wg / g.tiles_x) % g.tiles_y, cz = wg / (g.tiles_x * g.tiles_y);
    const int x0t = tx * kRbTX, y0t = ty * kRbTY, z0 = cz * g.chunk, z1 = min(z0 + g.chunk, g.nz);
    const size_t nc = (size_t)g.ny * g.nz * g.hx; // elements of a per-colour array (one slot)
    const size_t pl = (size_t)g.ny * g.nx, plc = (size_t)g.ny * g.hx;
    T ra[NA], rd[NA], vb[NB][8], vc[NC][8];
#define RB_SLOT(p) ((((p) % 3) + 3) % 3)
#define RB_A_CELL(i)                                                                              \
    const int  idx = tid + kRbThreads * (i), lx = idx % kRbAX, ly = idx / kRbAX, gx = x0t - 2 + lx, gy = y0t - 2 + ly; \
    const bool in  = idx < kAC && gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny
#define RB_B_CELL(i, p)                                                                           \
    const int  idx = tid + kRbThreads * (i), ly = idx / kBH, gy = y0t - 1 + ly,                          \
              lx  = 2 * (idx % kBH) + ((1 - g.p0 - gy - (p) - (x0t - 1)) & 1), gx = x0t - 1 + lx; \
    const bool in  = idx < kBC && (p) >= 0 && (p) < g.nz && gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny
#define RB_C_CELL(i, p)                                                                           \
    const int  idx = tid + kRbThreads * (i), ly = idx / kCH, gy = y0t + ly,                              \
              lx  = 2 * (idx % kCH) + ((g.p0 - gy - (p) - x0t) & 1), gx = x0t + lx;               \
    const bool in  = idx < kCC && gx < g.nx && gy < g.ny
    // ---- A: plane p, rhs / x0 on the tile + 2
#define RB_LOAD_A(p)                                                                \
    _Pragma("unroll") for(int i = 0; i < NA; ++i)                                   \
    {                                                                               \
        RB_A_CELL(i);                                                               \
        (void)in;                                                                   \
        const int cx = rb_clamp(gx, g.nx - 1), cy = rb_clamp(gy, g.ny - 1), cp = rb_clamp((p), g.nz - 1); \
        ra[i] = rhs[(size_t)cp * pl + (size_t)cy * g.nx + cx];                      \
        rd[i] = di0[(size_t)cp * plc + (size_t)cy * g.hx + (cx >> 1)];              \
    }
#define RB_PUT_A(p)                                                                 \
    _Pragma("unroll") for(int i = 0; i < NA; ++i)                                   \
    {                                                                               \
        RB_A_CELL(i);                                                               \
        T v = (((gx + gy + (p)) & 1) == g.p0) ? ra[i] * rd[i] : ra[i];              \
        if(!(in && (p) >= 0 && (p) < g.nz))                                         \
            v = (T)0;                                                               \
        if(idx < kAC)                                                               \
            XR[(RB_SLOT(p) * kRbAY + ly) * kRbPX + lx] = v;                         \
    }
    // ---- B: plane p, colour-1 values on the tile + 1
#define RB_LOAD8(dst, vp, dp, ip, ci)                                                                                   \
    dst[0] = vp[ci], dst[1] = vp[nc + ci], dst[2] = vp[2 * nc + ci], dst[3] = vp[3 * nc + ci], dst[4] = vp[4 * nc + ci], \
    dst[5] = vp[5 * nc + ci], dst[6] = dp[ci], dst[7] = ip[ci]
#define RB_LOAD_B(p)                                                                \
    _Pragma("unroll") for(int i = 0; i < NB; ++i)                                   \
    {                                                                               \
        RB_B_CELL(i, p);                                                            \
        (void)in;                                                                   \
        const size_t ci = (size_t)rb_clamp((p), g.nz - 1) * plc + (size_t)rb_clamp(gy, g.ny - 1) * g.hx + (rb_clamp(gx, g.nx - 1) >> 1); \
        RB_LOAD8(vb[i], val1, d1, di1, ci);                                         \
    }
#define RB_PUT_B(p)                                                                 \
    _Pragma("unroll") for(int i = 0; i < NB; ++i)                                   \
    {                                                                               \
        RB_B_CELL(i, p);                                                            \
        T s = (T)0;                                                                 \
        if(in)                                                                      \
        {                                                                           \
            const T* X0 = XR + (RB_SLOT((p)-1) * kRbAY + ly + 1) * kRbPX + lx + 1;  \
            const T* X1 = XR + (RB_SLOT(p) * kRbAY + ly + 1) * kRbPX + lx + 1;      \
            const T* X2 = XR + (RB_SLOT((p) + 1) * kRbAY + ly + 1) * kRbPX + lx + 1; \
            s = X1[0];                                                              \
            if((p) > 0)                                                             \
                s -= vb[i][0] * X0[0];                                              \
            if(gy > 0)                                                              \
                s -= vb[i][1] * X1[-kRbPX];                                         \
            if(gx > 0)                                                              \
                s -= vb[i][2] * X1[-1];                                             \
            if(gx < g.nx - 1)                                                       \
                s -= vb[i][3] * X1[1];                                              \
            if(gy < g.ny - 1)                                                       \
                s -= vb[i][4] * X1[kRbPX];                                          \
            if((p) < g.nz - 1)                                                      \
                s -= vb[i][5] * X2[0];                                              \
            s = s * vb[i][7];                                                       \
            s = s * vb[i][6];                                                       \
            s = s * vb[i][7];                                                       \
        }                                                                           \
        if(idx < kBC)                                                               \
            Y1[(RB_SLOT(p) * kRbBY + ly) * kRbPY + lx] = s;                         \
    }
    // ---- C: plane p, colour-0 values on the tile (into the cell's place in XR)
#define RB_LOAD_C(p)                                                                \
    _Pragma("unroll") for(int i = 0; i < NC; ++i)                                   \
    {                                                                               \
        RB_C_CELL(i, p);                                                            \
        (void)in;                                                                   \
        const size_t ci = (size_t)rb_clamp((p), g.nz - 1) * plc + (size_t)rb_clamp(gy, g.ny - 1) * g.hx + (rb_clamp(gx, g.nx - 1) >> 1); \
        RB_LOAD8(vc[i], val0, d0, di0, ci);                                         \
    }
#define RB_PUT_C(p)                                                                 \
    _Pragma("unroll") for(int i = 0; i < NC; ++i)                                   \
    {                                                                               \
        RB_C_CELL(i, p);                                                            \
        if(in)                                                                      \
        {                                                                           \
            T*       X  = XR + (RB_SLOT(p) * kRbAY + ly + 2) * kRbPX + lx + 2;      \
            const T* W0 = Y1 + (RB_SLOT((p)-1) * kRbBY + ly + 1) * kRbPY + lx + 1;  \
            const T* W1 = Y1 + (RB_SLOT(p) * kRbBY + ly + 1) * kRbPY + lx + 1;      \
            const T* W2 = Y1 + (RB_SLOT((p) + 1) * kRbBY + ly + 1) * kRbPY + lx + 1; \
            T        s  = X[0] * vc[i][6];                                          \
            if((p) > 0)                                                             \
                s -= vc[i][0] * W0[0];                                              \
            if(gy > 0)                                                              \
                s -= vc[i][1] * W1[-kRbPY];                                         \
            if(gx > 0)                                                              \
                s -= vc[i][2] * W1[-1];                                             \
            if(gx < g.nx - 1)                                                       \
                s -= vc[i][3] * W1[1];                                              \
            if(gy < g.ny - 1)                                                       \
                s -= vc[i][4] * W1[kRbPY];                                          \
            if((p) < g.nz - 1)                                                      \
                s -= vc[i][5] * W2[0];                                              \
            X[0] = s * vc[i][7];                                                    \
        }                                                                           \
    }
    // ---- the rows of out of plane p, both colours, whole lines
#define RB_STORE(p)                                                                 \
    for(int idx = tid; idx < kRbTX * kRbTY; idx += kRbThreads)                               \
    {                                                                               \
        const int lx = idx % kRbTX, ly = idx / kRbTX, gx = x0t + lx, gy = y0t + ly;   \
        if(gx < g.nx && gy < g.ny)                                                  \
        {                                                                           \
            const T o = ((gx + gy + (p)) & 1) == g.p0 ? XR[(RB_SLOT(p) * kRbAY + ly + 2) * kRbPX + lx + 2]  \
                                                      : Y1[(RB_SLOT(p) * kRbBY + ly + 1) * kRbPY + lx + 1]; \
            nt_store(o, out + (size_t)(p)*pl + (size_t)gy * g.nx + gx);             \
        }                                                                           \
    }
    // warm-up: planes z0 - 2 .. z0 + 1 of A, z0 - 1 and z0 of B
    RB_LOAD_A(z0 - 2);
    RB_PUT_A(z0 - 2);
    RB_LOAD_A(z0 - 1);
    RB_PUT_A(z0 - 1);
    RB_LOAD_A(z0);
    RB_PUT_A(z0);
    RB_LOAD_B(z0 - 1);
    RB_LOAD_A(z0 + 1);
    rb_barrier();
    RB_PUT_B(z0 - 1);
    RB_LOAD_B(z0);
    rb_barrier();
    RB_PUT_A(z0 + 1);
    RB_LOAD_A(z0 + 2);
    rb_barrier();
    RB_PUT_B(z0);
    RB_LOAD_B(z0 + 1);
    RB_LOAD_C(z0);
    rb_barrier();
    for(int z = z0; z < z1; ++z)
    {
        RB_PUT_A(z + 2);
        RB_LOAD_A(z + 3); // (every load is issued by every lane of every wave, at a clamped address where it has no cell:
        rb_barrier(); //      the number of loads in flight is then a constant the compiler can count its waits against)
        RB_PUT_B(z + 1);
        RB_LOAD_B(z + 2);
        rb_barrier();
        RB_PUT_C(z);
        RB_LOAD_C(z + 1);
        rb_barrier();
        RB_STORE(z);
        rb_barrier();
    }
#undef RB_SLOT
#undef RB_A_CELL
#undef RB_B_CELL
#undef RB_C_CELL
#undef RB_LOAD_A
#undef RB_PUT_A
#undef RB_LOAD8
#undef RB_LOAD_B
#undef RB_PUT_B
#undef RB_LOAD_C
#undef RB_PUT_C
#undef RB_STORE
}

// min over the off-diagonal entries of |original index of the column - original index of the row| beyond thr
__global__ __launch_bounds__(kBlock) void k_rb_min_offset(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                          const int* __restrict__ iperm, int thr, int* __restrict__ out_min)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    int           mn  = 0x7fffffff;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
    {
        const int r = iperm[t];
        for(int a = rp[t]; a < rp[t + 1]; ++a)
        {
            const int dd = abs(iperm[ci[a]] - r);
            if(dd > thr)
                mn = min(mn, dd);
        }
    }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
        mn = min(mn, __shfl_xor(mn, off, 64));
    if((threadIdx.x & 63) == 0 && mn != 0x7fffffff)
        atomicMin(out_min, mn);
}
// every row: exactly its lattice neighbours + the diagonal, its colour the parity of its cell; values and diagonals laid out per colour
template <typename T>
__global__ __launch_bounds__(kBlock) void k_rb_fill(int n, int nx, int ny, int nz, int hx, int p0, const int* __restrict__ rp,
                                                    const int* __restrict__ ci, const T* __restrict__ val,
                                                    const int* __restrict__ iperm, const int* __restrict__ blk_of,
                                                    const T* __restrict__ d, const T* __restrict__ dinv, T* __restrict__ v0,
                                                    T* __restrict__ v1, T* __restrict__ od0, T* __restrict__ odi0,
                                                    T* __restrict__ od1, T* __restrict__ odi1, int* __restrict__ bad)
{
    const int64_t gsz  = (int64_t)gridDim.x * blockDim.x;
    const int     nxny = nx * ny;
    const size_t  nc   = (size_t)ny * nz * hx;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
    {
        const int r = iperm[t], x = r % nx, y = (r / nx) % ny, z = r / nxny;
        const int c = blk_of[t];
        bool      ok = (((x + y + z) & 1) == p0) == (c == 0);
        T         a[6] = {(T)0, (T)0, (T)0, (T)0, (T)0, (T)0};
        int       seen = 0, last = -2147483647;
        bool      diag = false;
        for(int q = rp[t]; q < rp[t + 1]; ++q)
        {
            const int dd = iperm[ci[q]] - r;
            int       k  = -1;
            // the colour sweeps subtract in the storage order of the permuted row, k_mc_rb in ascending natural offsets: the
            // two orders must be the same one (a colouring permutation that does not keep the natural order within a colour
            // would still be a valid one -- the sweeps then remain the form that runs)
            if(dd != 0)
            {
                ok   = ok && dd > last;
                last = dd;
            }
            if(dd == 0)
                diag = true;
            else if(dd == -nxny && z > 0)
                k = 0;
            else if(dd == -nx && y > 0)
                k = 1;
            else if(dd == -1 && x > 0)
                k = 2;
            else if(dd == 1 && x < nx - 1)
                k = 3;
            else if(dd == nx && y < ny - 1)
                k = 4;
            else if(dd == nxny && z < nz - 1)
                k = 5;
            else
                ok = false;
            if(k >= 0)
            {
                ok   = ok && !(seen & (1 << k));
                seen |= 1 << k;
                a[k] = val[q];
            }
        }
        const int want = (z > 0 ? 1 : 0) | (y > 0 ? 2 : 0) | (x > 0 ? 4 : 0) | (x < nx - 1 ? 8 : 0) | (y < ny - 1 ? 16 : 0)
                         | (z < nz - 1 ? 32 : 0);
        ok = ok && diag && seen == want;
        if(!ok)
        {
            *bad = 1;
            continue;
        }
        const size_t cidx = ((size_t)z * ny + y) * hx + (x >> 1);
        T*           v    = c == 0 ? v0 : v1;
        for(int k = 0; k < 6; ++k)
            v[k * nc + cidx] = a[k];
        (c == 0 ? od0 : od1)[cidx]   = d[t];
        (c == 0 ? odi0 : odi1)[cidx] = dinv[t];
    }
}

// pairs of the output (see McsgsPlan): row t of colour 0 takes the row next to it in its aligned pair of out along where that
// row has another colour
__global__ __launch_bounds__(kBlock) void k_mc_pairs(int n, int n0, const int* __restrict__ iperm, const int* __restrict__ perm,
                                                     int* __restrict__ pair_of, unsigned char* __restrict__ covered)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n0; t += gsz)
    {
        const int o = iperm[t] ^ 1;
        int       q = -1;
        if(o < n)
        {
            const int pq = perm[o];
            if(pq >= n0)
            {
                q           = pq;
                covered[pq] = 1;
            }
        }
        pair_of[t] = q;
    }
}

// half the spread of the columns of sampled rows of [p0, p1): the distance between rows that gather the same far line
__global__ __launch_bounds__(kBlock) void k_mc_band_sample(int p0, int p1, int stride, const int* __restrict__ slice_off,
                                                           const int* __restrict__ ecol, int* __restrict__ out)
{
    const int     s    = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t    = (int64_t)p0 + ((int64_t)s * stride + stride / 2) % (p1 - p0);
    const int     base = slice_off[t >> 6];
    const int     w    = (slice_off[(t >> 6) + 1] - base) >> 6;
    int           lo = 0x7fffffff, hi = -1;
    for(int k = 0; k < w; ++k)
    {
        const int c = ecol[base + k * 64 + (int)(t & 63)];
        if(c < 0)
            break;
        lo = min(lo, c);
        hi = max(hi, c);
    }
    out[s] = hi >= lo ? (hi - lo) / 2 : 0;
}

static int mc_band_map(const McsgsPlan* P, int colour, bool lower, BandMap* bm)
{
    *bm = BandMap{0, 0, 0};
    // RAMD_MC_XCD: 1 (default) band-aware tiles inside every XCD's contiguous eighth; 2: the contiguous eighth in linear
    // order; 0: workgroup b takes block b (round robin over the XCDs)
    static const int xcd_env = getenv("RAMD_MC_XCD") ? atoi(getenv("RAMD_MC_XCD")) : 1;
    const int p0 = P->off[(size_t)colour], p1 = P->off[(size_t)colour + 1];
    if(xcd_env == 0)
        bm->W = -1;
    if(xcd_env == 0 || xcd_env == 2 || p1 - p0 < (1 << 20))
        return RAMD_OK;
    Backend&  b       = backend();
    const int samples = 1024;
    int*      d       = nullptr;
    RAMD_TRY(dev_alloc(&d, samples));
    hipLaunchKernelGGL(k_mc_band_sample, dim3(samples / kBlock), dim3(kBlock), 0, b.cur, p0, p1, std::max(1, (p1 - p0) / samples),
                       lower ? P->l_off : P->u_off, lower ? P->l_col : P->u_col, d);
    std::vector<int> h((size_t)samples);
    hipError_t       e = hipMemcpyAsync(h.data(), d, sizeof(int) * samples, hipMemcpyDeviceToHost, b.cur);
    if(e == hipSuccess)
        e = hipStreamSynchronize(b.cur);
    dev_free(&d);
    RAMD_HIP(e);
    std::sort(h.begin(), h.end());
    const int med = h[samples / 2];
    const int cnt = (int)(std::upper_bound(h.begin(), h.end(), med) - std::lower_bound(h.begin(), h.end(), med));
    if(cnt * 2 >= samples && med > 0 && med % kBlock == 0 && (int64_t)med * 8 >= (1 << 19) && med < (p1 - p0) / 16)
    {
        const int nblk = (p1 - p0 + kBlock - 1) / kBlock, per_xcd = (nblk + 7) / 8;
        bm->P = med / kBlock;
        bm->Z = per_xcd / bm->P;
        static const int w_env = getenv("RAMD_MC_BANDW") ? atoi(getenv("RAMD_MC_BANDW")) : 32; // (row blocks per plane of a tile)
        bm->W = w_env > 0 ? w_env : 32;
        while(bm->W > 1 && bm->P % bm->W != 0)
            bm->W >>= 1;
        if(bm->Z < 3)
            bm->P = 0;
    }
    return RAMD_OK;
}

template <typename T>
static int mc_pack(McsgsPlan* P, const ramd_mat_s* m, bool lower, const int* d_off)
{
    Backend&  b       = backend();
    const int n       = P->n;
    const int nslices = (n + 63) / 64;
    int**     soff    = lower ? &P->l_off : &P->u_off;
    int**     scol    = lower ? &P->l_col : &P->u_col;
    void**    sval    = lower ? &P->l_val : &P->u_val;
    RAMD_TRY(dev_alloc(soff, (int64_t)nslices + 1));
    RAMD_HIP(hipMemsetAsync(*soff, 0, sizeof(int) * ((size_t)nslices + 1), b.cur));
    const unsigned g64 = (unsigned)(((int64_t)nslices * 64 + kBlock - 1) / kBlock);
    if(lower)
        hipLaunchKernelGGL((k_mc_width<true>), dim3(g64), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, d_off,
                           P->blk_of, *soff);
    else
        hipLaunchKernelGGL((k_mc_width<false>), dim3(g64), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, d_off,
                           P->blk_of, *soff);
    RAMD_TRY(device_exclusive_scan(*soff, *soff, (int64_t)nslices + 1));
    int total = 0;
    RAMD_HIP(hipMemcpyAsync(&total, *soff + nslices, sizeof(int), hipMemcpyDeviceToHost, b.cur));
    RAMD_HIP(hipStreamSynchronize(b.cur));
    RAMD_TRY(dev_alloc(scol, total));
    RAMD_HIP(cached_malloc(sval, (size_t)total * sizeof(T) + kPad));
    const unsigned nbk = (unsigned)((n + kBlock - 1) / kBlock);
    if(lower)
        hipLaunchKernelGGL((k_mc_fill<T, true>), dim3(nbk), dim3(kBlock), 0, b.cur, n, P->nb, m->rp, m->ci,
                           (const T*)m->val, d_off, P->blk_of, *soff, *scol, (T*)*sval, (T*)P->d,
                           (T*)P->dinv);
    else
        hipLaunchKernelGGL((k_mc_fill<T, false>), dim3(nbk), dim3(kBlock), 0, b.cur, n, P->nb, m->rp, m->ci,
                           (const T*)m->val, d_off, P->blk_of, *soff, *scol, (T*)*sval, (T*)P->d,
                           (T*)P->dinv);
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

// does the diagonal colour block (i,i) store any entry?  cnt[i] = 0: no -> Jacobi inverse diagonal empty
__global__ __launch_bounds__(kBlock) void k_mc_block_nnz(int n, const int* __restrict__ rp,
                                                         const int* __restrict__ ci,
                                                         const int* __restrict__ off,
                                                         const int* __restrict__ blk_of,
                                                         int* cnt)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for(int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gsz)
    {
        const int b = blk_of[t];
        int       c = 0;
        for(int j = rp[t]; j < rp[t + 1]; ++j)
            if(ci[j] >= off[b] && ci[j] < off[b + 1])
                ++c;
        // only "none at all" matters to the caller, so cnt[b] is a flag: raised by a plain agent-scope store, and only
        // where a load says it is not raised yet (67 M same-address atomic adds took 1.5 s at 512^3, one per wave still
        // 24 ms: an RMW on one address costs ~10 ns whoever issues it)
        if(c && __hip_atomic_load(cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T>
static int mc_build(McsgsPlan* P, const ramd_mat_s* m, const int* perm)
{
    Backend&  b = backend();
    const int n = P->n, nb = P->nb;
    int*      d_off = nullptr;
    RAMD_TRY(dev_alloc(&d_off, (int64_t)nb + 1));
    RAMD_HIP(hipMemcpyAsync(d_off, P->off.data(), sizeof(int) * ((size_t)nb + 1), hipMemcpyHostToDevice, b.cur));
    RAMD_TRY(dev_alloc(&P->iperm, n));
    RAMD_TRY(dev_alloc(&P->blk_of, n));
    RAMD_HIP(cached_malloc(&P->d, (size_t)n * sizeof(T) + kPad));
    RAMD_HIP(cached_malloc(&P->dinv, (size_t)n * sizeof(T) + kPad));
    RAMD_HIP(cached_malloc(&P->xp, (size_t)n * sizeof(T) + kPad));
    hipLaunchKernelGGL(k_mc_prepare, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, nb, d_off, perm, P->iperm,
                       P->blk_of);
    int s = mc_pack<T>(P, m, true, d_off);
    if(s == RAMD_OK)
        s = mc_pack<T>(P, m, false, d_off);
    // structured operators: row patterns of both parts (same switch and threshold as the SpMV: RAMD_CSR_PAT, 2^20 entries)
    static const int pat_env = getenv("RAMD_CSR_PAT") ? atoi(getenv("RAMD_CSR_PAT")) : -1;
    if(s == RAMD_OK && pat_env != 0 && (pat_env > 0 || m->nnz >= (1 << 20)))
    {
        s = sell_analyse_pattern(n, P->l_off, P->l_col, &P->l_pat, &P->l_pat_n, &P->l_pat_id, &P->l_pat_dict);
        if(s == RAMD_OK)
            s = sell_analyse_pattern(n, P->u_off, P->u_col, &P->u_pat, &P->u_pat_n, &P->u_pat_id, &P->u_pat_dict);
    }
    P->l_bm.assign((size_t)nb, BandMap{0, 0, 0});
    P->u_bm.assign((size_t)nb, BandMap{0, 0, 0});
    for(int i = 0; i < nb && s == RAMD_OK; ++i)
    {
        s = mc_band_map(P, i, true, &P->l_bm[(size_t)i]);
        if(s == RAMD_OK)
            s = mc_band_map(P, i, false, &P->u_bm[(size_t)i]);
    }
    // (8-byte values only: a pair of fp32 values is no full 16-byte store; RAMD_MC_PAIR=0: off)
    const int pair_env = getenv("RAMD_MC_PAIR") ? atoi(getenv("RAMD_MC_PAIR")) : 1;
    if(s == RAMD_OK && pair_env != 0 && nb > 1 && sizeof(T) == 8 && (n >= (1 << 16) || pair_env == 2)) // (2: any size -- tests)
    {
        s = dev_alloc(&P->pair_of, P->off[1]);
        unsigned char* cov = nullptr;
        if(s == RAMD_OK)
            s = dev_alloc(&cov, n);
        if(s == RAMD_OK)
        {
            P->covered = cov;
            hipError_t e = hipMemsetAsync(cov, 0, (size_t)n, b.cur);
            hipLaunchKernelGGL(k_mc_pairs, dim3(ew_grid(P->off[1])), dim3(kBlock), 0, b.cur, n, P->off[1], P->iperm, perm,
                               P->pair_of, cov);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
        }
    }
    int* cnt = nullptr;
    if(s == RAMD_OK)
        s = dev_alloc(&cnt, nb);
    if(s == RAMD_OK)
    {
        hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)nb, b.cur);
        hipLaunchKernelGGL(k_mc_block_nnz, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, d_off,
                           P->blk_of, cnt);
        std::vector<int> h((size_t)nb);
        if(e == hipSuccess)
            e = hipMemcpyAsync(h.data(), cnt, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, b.cur);
        if(e == hipSuccess)
            e = hipStreamSynchronize(b.cur);
        if(e != hipSuccess)
            s = RAMD_ERR_HIP;
        P->identity.assign((size_t)nb, 0);
        for(int i = 0; i < nb; ++i)
            P->identity[(size_t)i] = (h[(size_t)i] == 0) ? 1 : 0;
    }
    dev_free(&cnt);
    // red-black lattice form (k_mc_rb): RAMD_MC_RB = 0 off, 1 (default) operators of 2^16 rows and more, 2 any size (tests)
    // (read at every Build(), not once per process: the tests switch it inside one process)
    const int rb_env = getenv("RAMD_MC_RB") ? atoi(getenv("RAMD_MC_RB")) : 1;
    if(s == RAMD_OK && rb_env != 0 && nb == 2 && !P->identity[0] && !P->identity[1] && (n >= (1 << 16) || rb_env == 2) && n >= 8)
    {
        int* dmin = nullptr;
        int  h2[2] = {0x7fffffff, 0};
        auto pass  = [&](int thr) -> int {
            h2[0] = 0x7fffffff;
            if(hipMemcpyAsync(dmin, h2, sizeof(int), hipMemcpyHostToDevice, b.cur) != hipSuccess)
                return RAMD_ERR_HIP;
            hipLaunchKernelGGL(k_rb_min_offset, dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, m->rp, m->ci, P->iperm, thr, dmin);
            if(hipMemcpyAsync(h2, dmin, sizeof(int), hipMemcpyDeviceToHost, b.cur) != hipSuccess || hipStreamSynchronize(b.cur) != hipSuccess)
                return RAMD_ERR_HIP;
            return RAMD_OK;
        };
        s = dev_alloc(&dmin, 2);
        int nx = 0, ny = 0, nz = 0;
        if(s == RAMD_OK)
            s = pass(1);
        if(s == RAMD_OK && h2[0] != 0x7fffffff && h2[0] >= 2 && n % h2[0] == 0)
        {
            nx = h2[0];
            s  = pass(nx);
            const int64_t nxny = (s == RAMD_OK && h2[0] != 0x7fffffff) ? h2[0] : n;
            if(s == RAMD_OK && nxny % nx == 0 && n % nxny == 0 && nxny / nx >= 2)
            {
                ny = (int)(nxny / nx);
                nz = (int)(n / nxny);
            }
        }
        if(s == RAMD_OK && nz > 0)
        {
            const int    hx = (nx + 1) / 2;
            const size_t nc = (size_t)ny * nz * hx;
            int          c0 = 0, t0 = 0; // colour of the original row 0 (cell (0, 0, 0): parity 0), its position
            hipError_t   e  = hipMemcpyAsync(&t0, perm, sizeof(int), hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e == hipSuccess)
                e = hipMemcpyAsync(&c0, P->blk_of + t0, sizeof(int), hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            const int p0 = c0 == 0 ? 0 : 1;
            for(int c = 0; c < 2 && e == hipSuccess; ++c)
            {
                e = cached_malloc(&P->rb_val[c], 6 * nc * sizeof(T) + kPad);
                if(e == hipSuccess)
                    e = cached_malloc(&P->rb_d[c], nc * sizeof(T) + kPad);
                if(e == hipSuccess)
                    e = cached_malloc(&P->rb_di[c], nc * sizeof(T) + kPad);
                if(e == hipSuccess)
                    e = hipMemsetAsync(P->rb_val[c], 0, 6 * nc * sizeof(T), b.cur);
                if(e == hipSuccess)
                    e = hipMemsetAsync(P->rb_d[c], 0, nc * sizeof(T), b.cur);
                if(e == hipSuccess)
                    e = hipMemsetAsync(P->rb_di[c], 0, nc * sizeof(T), b.cur);
            }
            int hb = 1;
            if(e == hipSuccess)
                e = hipMemsetAsync(dmin + 1, 0, sizeof(int), b.cur);
            if(e == hipSuccess)
            {
                hipLaunchKernelGGL((k_rb_fill<T>), dim3(ew_grid(n)), dim3(kBlock), 0, b.cur, n, nx, ny, nz, hx, p0, m->rp, m->ci,
                                   (const T*)m->val, P->iperm, P->blk_of, (const T*)P->d, (const T*)P->dinv, (T*)P->rb_val[0],
                                   (T*)P->rb_val[1], (T*)P->rb_d[0], (T*)P->rb_di[0], (T*)P->rb_d[1], (T*)P->rb_di[1], dmin + 1);
                e = hipMemcpyAsync(&hb, dmin + 1, sizeof(int), hipMemcpyDeviceToHost, b.cur);
            }
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
            P->rb = (s == RAMD_OK && hb == 0);
            P->rb_nx = nx, P->rb_ny = ny, P->rb_nz = nz, P->rb_p0 = p0;
            if(getenv("RAMD_MC_VERBOSE"))
                fprintf(stderr, "mc plan: red-black lattice form %s on %d x %d x %d (colour 0 = parity %d)\n", P->rb ? "taken" : "REFUSED by the row check",
                        nx, ny, nz, p0);
            if(!P->rb)
                for(int c = 0; c < 2; ++c)
                {
                    void** ps2[] = {&P->rb_val[c], &P->rb_d[c], &P->rb_di[c]};
                    for(void** q : ps2)
                    {
                        if(*q)
                            (void)cached_free(*q);
                        *q = nullptr;
                    }
                }
        }
        else if(getenv("RAMD_MC_VERBOSE"))
            fprintf(stderr, "mc plan: no lattice found for the red-black form (n = %d, smallest offsets %d / %d)\n", n, nx, h2[0]);
        dev_free(&dmin);
    }
    else if(getenv("RAMD_MC_VERBOSE"))
        fprintf(stderr, "mc plan: red-black form not tried (n = %d, %d colours, identity blocks %d %d, RAMD_MC_RB = %d)\n", n, nb,
                nb > 0 ? (int)P->identity[0] : -1, nb > 1 ? (int)P->identity[1] : -1, rb_env);
    // colour 0 folded into its readers (see McsgsPlan::fold0): RAMD_MC_FOLD=0 switches it off (A/B, and the tests run both forms).
    // Not prepared where the red-black form was taken: k_mc_rb does not use it (round 5 spent 4.7 s of Build() here for nothing).
    const int fold_env = getenv("RAMD_MC_FOLD") ? atoi(getenv("RAMD_MC_FOLD")) : 1;
    if(s == RAMD_OK && fold_env != 0 && !P->rb && nb > 1 && P->l_pat == 1 && P->u_pat == 1 && !P->identity[0] && P->off[1] > 0)
    {
        const int n0 = P->off[1];
        int*      bad = nullptr;
        s = dev_alloc(&P->nat_dict, (int64_t)P->l_pat_n * kPatMaxW);
        if(s == RAMD_OK)
            s = dev_alloc(&bad, 1);
        if(s == RAMD_OK && cached_malloc(&P->dinv_nat, (size_t)n * sizeof(T) + kPad) != hipSuccess)
            s = RAMD_ERR_HIP;
        if(s == RAMD_OK)
        {
            std::vector<int> init((size_t)P->l_pat_n * kPatMaxW, kNatNone);
            hipError_t       e = hipMemcpyAsync(P->nat_dict, init.data(), sizeof(int) * init.size(), hipMemcpyHostToDevice, b.cur);
            if(e == hipSuccess)
                e = hipMemsetAsync(bad, 0, sizeof(int), b.cur);
            if(e == hipSuccess)
                e = hipMemsetAsync(P->dinv_nat, 0, (size_t)n * sizeof(T), b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur); // (init goes out of scope)
            for(int pass = 0; pass < 2; ++pass)
                hipLaunchKernelGGL(k_mc_nat_dict, dim3(ew_grid(n - n0)), dim3(kBlock), 0, b.cur, n0, n, n0, pass, P->l_pat_id,
                                   P->l_pat_dict, P->iperm, P->nat_dict, bad);
            hipLaunchKernelGGL((k_mc_dinv_nat<T>), dim3(ew_grid(n0)), dim3(kBlock), 0, b.cur, n0, P->iperm, (const T*)P->dinv,
                               (T*)P->dinv_nat);
            int hb = 1;
            if(e == hipSuccess)
                e = hipMemcpyAsync(&hb, bad, sizeof(int), hipMemcpyDeviceToHost, b.cur);
            if(e == hipSuccess)
                e = hipStreamSynchronize(b.cur);
            if(e != hipSuccess)
                s = RAMD_ERR_HIP;
            P->fold0 = (s == RAMD_OK && hb == 0);
        }
        dev_free(&bad);
        if(!P->fold0)
        {
            dev_free(&P->nat_dict);
            if(P->dinv_nat)
                (void)cached_free(P->dinv_nat);
            P->dinv_nat = nullptr;
        }
    }
    dev_free(&d_off);
    return s;
}

template <typename T>
static int mc_apply(McsgsPlan* P, int kind, const T* rhs, T* out)
{
    Backend&  b  = backend();
    const int nb = P->nb;
    const CsrPattern lpat = {P->l_pat_id, P->l_pat_dict, P->l_pat_n, kPatMaxW};
    const CsrPattern upat = {P->u_pat_id, P->u_pat_dict, P->u_pat_n, kPatMaxW};
    struct ProfScope // (HIP events around the whole apply when the channel is on: bench.py)
    {
        hipStream_t s;
        explicit ProfScope(hipStream_t st) : s(st) { prof_begin(RAMD_PROF_PRECOND, s); }
        ~ProfScope() { prof_end(RAMD_PROF_PRECOND, s); }
    } prof_scope(b.cur);
    // (pairs: only in the applies whose LAST sweep with an output is colour 0's and in which every colour stores its output --
    //  all three kinds; the colour-0 sweep of an apply runs after all others)
    const bool pairs_on = P->pair_of != nullptr && nb > 1;
    // (KEEP: does a later sweep of this apply read the colour's xp?)
#define SWEEP_ID(FR, MD, BO, TO, i, OFFP, COLP, VALP, IDENT, KEEP)                                       \
    do                                                                                                   \
    {                                                                                                    \
        const int  p0 = P->off[(size_t)(i)], p1 = P->off[(size_t)(i) + 1];                               \
        const bool lower_part = (OFFP) == P->l_off;                                                      \
        const bool use_pat    = lower_part ? P->l_pat == 1 : P->u_pat == 1;                              \
        const int  nblk = (p1 - p0 + kBlock - 1) / kBlock, per_xcd = (nblk + 7) / 8;                     \
        const BandMap bm = lower_part ? P->l_bm[(size_t)(i)] : P->u_bm[(size_t)(i)];                     \
        if(p1 > p0 && use_pat)                                                                           \
            hipLaunchKernelGGL((k_mc_sweep<T, FR, MD, BO, TO, true>), dim3(per_xcd * 8),                 \
                               dim3(kBlock), 0, b.cur, p0, p1, OFFP, COLP, (const T*)VALP,               \
                               (const T*)P->d, (const T*)P->dinv, P->iperm, rhs, (T*)P->xp, out,         \
                               (int)(IDENT), lower_part ? lpat : upat, nblk, per_xcd, bm, (int)(KEEP),   \
                               (const int*)(((TO) && (i) == 0 && pairs_on) ? P->pair_of : nullptr),      \
                               (const unsigned char*)(((TO) && (i) > 0 && pairs_on) ? P->covered : nullptr)); \
        else if(p1 > p0)                                                                                 \
            hipLaunchKernelGGL((k_mc_sweep<T, FR, MD, BO, TO, false>), dim3(per_xcd * 8),                \
                               dim3(kBlock), 0, b.cur, p0, p1, OFFP, COLP, (const T*)VALP,               \
                               (const T*)P->d, (const T*)P->dinv, P->iperm, rhs, (T*)P->xp, out,         \
                               (int)(IDENT), lpat, nblk, per_xcd, bm, (int)(KEEP),                       \
                               (const int*)(((TO) && (i) == 0 && pairs_on) ? P->pair_of : nullptr),      \
                               (const unsigned char*)(((TO) && (i) > 0 && pairs_on) ? P->covered : nullptr)); \
    } while(0)
#define SWEEP(FR, MD, BO, TO, i, OFFP, COLP, VALP, KEEP)                                                 \
    SWEEP_ID(FR, MD, BO, TO, i, OFFP, COLP, VALP, P->identity[(size_t)(i)], KEEP)
    if(kind == RAMD_MC_GS)
    {
        // MultiColoredGS (preconditioner_multicolored_gs.cpp:250-288): x = P rhs, SolveR_ only
        for(int i = nb - 1; i >= 0; --i)
            SWEEP(true, false, false, true, i, P->u_off, P->u_col, P->u_val, i > 0);
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    if(kind == RAMD_MC_ILU)
    {
        // MultiColoredILU on the ILU(0) factors of P A P^T (preconditioner_multicolored_ilu.cpp:187-232):
        // SolveL_ has no diagonal solve (unit L), SolveD_ is empty, SolveR_ divides by the U diagonal
        for(int i = 0; i + 1 < nb; ++i)
            SWEEP_ID(true, false, false, false, i, P->l_off, P->l_col, P->l_val, 1, true);
        SWEEP(true, false, false, true, nb - 1, P->l_off, P->l_col, P->l_val, nb > 1); // last colour: L and R in one
        for(int i = nb - 2; i >= 0; --i)
            SWEEP(false, false, false, true, i, P->u_off, P->u_col, P->u_val, i > 0);
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    if(P->rb)
    {
        RbDims g;
        g.nx = P->rb_nx, g.ny = P->rb_ny, g.nz = P->rb_nz, g.hx = (P->rb_nx + 1) / 2, g.p0 = P->rb_p0;
        g.tiles_x = (g.nx + kRbTX - 1) / kRbTX, g.tiles_y = (g.ny + kRbTY - 1) / kRbTY, g.chunks = 1;
        // runs of planes: four workgroups per CU where the lattice has them (a run repeats three planes of its predecessor)
        const int want = (4 * b.num_cu + g.tiles_x * g.tiles_y - 1) / (g.tiles_x * g.tiles_y);
        g.chunk        = std::min(kRbChunkMax, std::max(kRbChunkMin, (g.nz + want - 1) / want));
        g.chunks       = (g.nz + g.chunk - 1) / g.chunk;
        const size_t lds = (size_t)(3 * kRbAY * kRbPX + 3 * kRbBY * kRbPY) * sizeof(T);
        hipLaunchKernelGGL((k_mc_rb<T>), dim3((unsigned)(g.tiles_x * g.tiles_y * g.chunks)), dim3(kRbThreads), lds, b.cur, g,
                           (const T*)P->rb_val[0], (const T*)P->rb_val[1], (const T*)P->rb_d[0], (const T*)P->rb_di[0],
                           (const T*)P->rb_d[1], (const T*)P->rb_di[1], rhs, out);
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    if(P->fold0)
    {
        // colour 0's forward sweep (a scaling, no entries) is not run: its readers form Dinv_0 rhs_0 themselves
#define SWEEPF(FR, MD, BO, TO, i, OFFP, COLP, VALP, KEEP, LOWERP)                                            \
    do                                                                                                       \
    {                                                                                                        \
        const int     p0 = P->off[(size_t)(i)], p1 = P->off[(size_t)(i) + 1];                                \
        const int     nblk = (p1 - p0 + kBlock - 1) / kBlock, per_xcd = (nblk + 7) / 8;                      \
        const BandMap bm = (LOWERP) ? P->l_bm[(size_t)(i)] : P->u_bm[(size_t)(i)];                           \
        if(p1 > p0)                                                                                          \
            hipLaunchKernelGGL((k_mc_sweep<T, FR, MD, BO, TO, true, true>), dim3(per_xcd * 8), dim3(kBlock), 0, b.cur, p0, p1, \
                               OFFP, COLP, (const T*)VALP, (const T*)P->d, (const T*)P->dinv, P->iperm, rhs, (T*)P->xp, out,   \
                               (int)P->identity[(size_t)(i)], (LOWERP) ? lpat : upat, nblk, per_xcd, bm, (int)(KEEP),         \
                               (const int*)(((TO) && (i) == 0 && pairs_on) ? P->pair_of : nullptr),                          \
                               (const unsigned char*)(((TO) && (i) > 0 && pairs_on) ? P->covered : nullptr), P->off[1],      \
                               P->nat_dict, (const T*)P->dinv_nat);                                                          \
    } while(0)
        for(int i = 1; i + 1 < nb; ++i)
            SWEEPF(true, false, false, false, i, P->l_off, P->l_col, P->l_val, true, true);
        SWEEPF(true, false, true, true, nb - 1, P->l_off, P->l_col, P->l_val, nb > 1, true);
        for(int i = nb - 2; i >= 1; --i)
            SWEEP(false, true, false, true, i, P->u_off, P->u_col, P->u_val, true);
        SWEEPF(false, true, false, true, 0, P->u_off, P->u_col, P->u_val, false, false);
#undef SWEEPF
        RAMD_HIP(hipGetLastError());
        return RAMD_OK;
    }
    // SolveL_ for colours 0 .. nb-2
    for(int i = 0; i + 1 < nb; ++i)
        SWEEP(true, false, false, false, i, P->l_off, P->l_col, P->l_val, true);
    // last colour: SolveL_ + SolveD_ + SolveR_ of its rows in one sweep (no U part)
    SWEEP(true, false, true, true, nb - 1, P->l_off, P->l_col, P->l_val, nb > 1);
    // SolveD_ + SolveR_ for colours nb-2 .. 0
    for(int i = nb - 2; i >= 0; --i)
        SWEEP(false, true, false, true, i, P->u_off, P->u_col, P->u_val, i > 0);
#undef SWEEP
#undef SWEEP_ID
    RAMD_HIP(hipGetLastError());
    return RAMD_OK;
}

} // namespace ramd

using namespace ramd;

struct ramd_mcsgs_s
{
    McsgsPlan plan;
};

namespace
{
// what the plan built last in this process looks like (ramd_mcsgs_info with a NULL handle: the Python drivers and bench.py do
// not hold the handle, the C++ preconditioner object does).  One word per field, written at the end of a Build().
std::atomic<long long> g_mc_last[8];
void mc_fill_info(const McsgsPlan& P, long long* o)
{
    o[0] = P.rb ? 2 : P.fold0 ? 1 : 0;
    o[1] = P.nb;
    o[2] = P.l_pat;
    o[3] = P.u_pat;
    o[4] = P.rb_nx, o[5] = P.rb_ny, o[6] = P.rb_nz;
    o[7] = P.n;
}
} // namespace

extern "C" {

int ramd_mcsgs_build(ramd_mat_t permuted, int num_blocks, const int* block_sizes, ramd_vec_t perm,
                     ramd_mcsgs_t* out)
{
    if(!permuted || !block_sizes || !perm || !out || num_blocks < 1)
        RAMD_FAIL(RAMD_ERR_ARG, "mcsgs_build: bad arguments");
    if(permuted->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(perm->dtype != RAMD_I32 || perm->n != permuted->nrow || permuted->nrow != permuted->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "mcsgs_build: permutation / matrix size mismatch");
    ramd_mcsgs_s* h = new ramd_mcsgs_s;
    McsgsPlan&    P = h->plan;
    P.dtype         = permuted->dtype;
    P.n             = permuted->nrow;
    P.nb            = num_blocks;
    P.off.assign((size_t)num_blocks + 1, 0);
    for(int i = 0; i < num_blocks; ++i)
        P.off[(size_t)i + 1] = P.off[(size_t)i] + block_sizes[i];
    if(P.off[(size_t)num_blocks] != P.n)
    {
        delete h;
        RAMD_FAIL(RAMD_ERR_ARG, "mcsgs_build: block sizes do not add up to the matrix size");
    }
    int s = (P.dtype == RAMD_F64) ? mc_build<double>(&P, permuted, (const int*)perm->d)
                                  : mc_build<float>(&P, permuted, (const int*)perm->d);
    if(s != RAMD_OK)
    {
        P.release();
        delete h;
        return s;
    }
    long long o[8];
    mc_fill_info(P, o);
    for(int i = 0; i < 8; ++i)
        g_mc_last[i].store(o[i], std::memory_order_relaxed);
    *out = h;
    return RAMD_OK;
}

int ramd_mcsgs_info(ramd_mcsgs_t h, int64_t* out8)
{
    if(!out8)
        RAMD_FAIL(RAMD_ERR_ARG, "mcsgs_info: bad arguments");
    long long o[8];
    if(h)
        mc_fill_info(h->plan, o);
    else
        for(int i = 0; i < 8; ++i)
            o[i] = g_mc_last[i].load(std::memory_order_relaxed);
    for(int i = 0; i < 8; ++i)
        out8[i] = (int64_t)o[i];
    return RAMD_OK;
}

int ramd_mcsgs_apply(ramd_mcsgs_t h, ramd_vec_t rhs, ramd_vec_t x)
{
    return ramd_mcsgs_apply_kind(h, RAMD_MC_SGS, rhs, x);
}

int ramd_mcsgs_apply_kind(ramd_mcsgs_t h, int kind, ramd_vec_t rhs, ramd_vec_t x)
{
    if(!h || !rhs || !x || rhs == x || kind < RAMD_MC_SGS || kind > RAMD_MC_ILU)
        RAMD_FAIL(RAMD_ERR_ARG, "mcsgs_apply: bad arguments");
    McsgsPlan& P = h->plan;
    if(rhs->dtype != P.dtype || x->dtype != P.dtype || rhs->n != P.n || x->n != P.n)
        RAMD_FAIL(RAMD_ERR_ARG, "mcsgs_apply: vector size / type mismatch");
    if(P.n == 0)
        return RAMD_OK;
    if(P.dtype == RAMD_F64)
        return mc_apply<double>(&P, kind, (const double*)rhs->d, (double*)x->d);
    return mc_apply<float>(&P, kind, (const float*)rhs->d, (float*)x->d);
}

int ramd_mcsgs_destroy(ramd_mcsgs_t h)
{
    if(h)
    {
        h->plan.release();
        delete h;
    }
    return RAMD_OK;
}

} // extern "C"
