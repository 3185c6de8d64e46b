// matrix_impl.hpp -- internal (non-ABI) declarations shared by the .hip translation units.
#pragma once

#include "common.hpp"

namespace ramd
{

size_t val_size(int dtype);
void   mat_free_csr(ramd_mat_s* m);
void   mat_free_ell(ramd_mat_s* m);
void   mat_free_coo(ramd_mat_s* m);
void   mat_free_analysis(ramd_mat_s* m);
int    mat_alloc_csr(ramd_mat_s* m, int nrow, int ncol, int64_t nnz);

// spmv.hip: detect a far band (3-D stencil plane distance) for the band-aware row-block traversal
int csr_analyse_band(ramd_mat_s* m);
int csr_analyse_groups(ramd_mat_s* m);
int csr_analyse_shift(ramd_mat_s* m); // rows that are their predecessor shifted by one column (stencils): ramd_mat_s::shift_rows
// row patterns (spmv.hip): rows whose column offsets col - row coincide share a dictionary entry of kPatMaxW slots
constexpr int kPatMaxW = 28; // longest row a pattern may have (round 6: the 27 entries of the reference's own 3-D operator; 16 before)
constexpr int kPatMax  = 64; // dictionary entries
constexpr int kPatEnd  = -2147483647 - 1; // dictionary entry of an ELL slot that holds no column (col < 0)
// x tiles of a structured CSR product (spmv.hip, k_csr_xl): the distinct column offsets of the dictionary fall into a few
// clusters; a 256-row block needs, per cluster, ONE contiguous piece of x -- loaded into LDS with 16-byte packets
constexpr int kXlSegs = 8; // clusters
struct XlSegs
{
    int nseg;
    int omin[kXlSegs]; // first element of the piece relative to the block's first row (a multiple of the packet size)
    int npk[kXlSegs]; // 16-byte packets of the piece
    int base[kXlSegs]; // where the piece starts in the LDS area (elements)
    int total; // elements of the LDS area in use (a multiple of the packet size)
};
// rows sharing the column list of the row before them (FE matrices: the unknowns of one mesh node): k_csr_tr<GRP>
struct CsrGroups
{
    const int*           lead; // [nrow] entry offset from a row's entries to its group leader's (<= 0)
    const unsigned char* need; // [nnz / 4] column packets that have to be read
};
struct CsrPattern
{
    const unsigned char* id; // [nrow]
    const int*           dict; // [n * kPatMaxW]
    int                  n, w;
};
// the same analysis for a wave-sliced ELL (slices of 64 rows, column-major inside a slice: the MC-SGS sweeps); on success
// *state = 1 and *id / *dict are device arrays the caller owns, else *state = -1
int sell_analyse_pattern(int nrow, const int* slice_off, const int* ecol, int* state, int* n, unsigned char** id, int** dict);

// backend.hip: optional HIP-event bracket around every SpMV launch (bench.py roofline leg)
void prof_spmv_begin();
void prof_spmv_end();
void prof_begin(int channel, hipStream_t s); // s == nullptr: the current stream
void prof_end(int channel, hipStream_t s);
void prof_count(int channel); // count an occurrence without timing it

// trisolve.hip
void tri_release(ramd_mat_s* m);
int  mat_transpose(const ramd_mat_s* m, ramd_mat_s* t); // t = m^T, rows sorted

// matrix_algebra.hip: *out = new matrix with the (row-sorted) pattern of a^q, SymbolicPower(q)
int mat_symbolic_power(const ramd_mat_s* a, int q, ramd_mat_s** out);

// blocksched.hip: hyperplane order of the row blocks for the natural-order sync-free sweeps (nullptr: natural)
int block_schedule(const ramd_mat_s* m, bool lower, int** order_out);
// ... and the wave units (<= 64 consecutive sweep rows that belong together) of the sweeps with in-wave register resolution
// (trisolve.hip), with their order by level of the unit graph, all computed on the device
struct UnitPlan
{
    int  nunits = 0;
    int* ustart = nullptr; // [nunits + 1] first sweep row of every unit
    int* order  = nullptr; // [nunits] or nullptr: natural order
    void release();
};
struct UnitView // (kernel argument)
{
    const int* ustart;
    const int* order;
    int        nunits;
};
inline UnitView unit_view(const UnitPlan& p)
{
    return UnitView{p.ustart, p.order, p.nunits};
}
int unit_schedule(const ramd_mat_s* m, bool lower, UnitPlan* out);

// coloring.hip: device greedy colouring; RAMD_ERR_UNSUPPORTED -> caller runs the host sweep
int multicoloring_device(const ramd_mat_s* m, int* num_colors, int* size_colors, ramd_vec_s* perm);

// spmv.hip
template <typename T>
int mat_apply_impl(const ramd_mat_s* m, const T* x, T* y, int mode, T scalar);
template <typename T>
int mat_apply_dot_impl(const ramd_mat_s* m, const T* x, T* y, int slot, const T* dotv = nullptr);
template <typename T>
int mat_jacobi_sweep_impl(const ramd_mat_s* m, const T* dinv, const T* rhs, const T* x, T* xnew, T omega);
template <typename T>
int mat_apply_add_dot_impl(const ramd_mat_s* m, const T* x, T* y, T scalar, const T* p, int slot);

// vector.hip: scalars[slot] = sum(a[0..n)) in one launch, fixed order
int reduce_sum_to_slot(const double* a, int64_t n, int slot);

// scan.hip: out[i] = sum_{k<i} in[k] for i < n (in and out may alias); int32 sums
int device_exclusive_scan(const int* in, int* out, int64_t n);
// order_out = indices 0..n-1 sorted by keys (0 <= key <= max_key), stable
int device_stable_sort_by_key(const int* keys, int64_t n, int max_key, int* order_out);
// max over an int array -> host
int device_max_int(const int* in, int64_t n, int* result);

} // namespace ramd
