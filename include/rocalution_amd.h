/* ==========================================================================
 * rocalution_amd.h -- C ABI of the MI355X-native Krylov backend (librocalution_amd.so)
 *
 * This is the drop-in boundary.  In the reference the host library talks to
 * its accelerator plugin through 30 C++ symbols (lifecycle + factories, declared in
 * src/base/hip/backend_hip.hpp:40-82) and through the virtual tables of
 * AcceleratorVector<T> (src/base/base_vector.hpp:43-232) and AcceleratorMatrix<T>
 * (src/base/base_matrix.hpp:78-866).  Each entry point below is the flat-C form of ONE of
 * those symbols / virtual methods; the comment on it cites the reference interface it
 * replaces.  Plain pointers and sizes only -- no torch / STL types.
 *
 * Conventions
 *   - every function returns an int status (RAMD_OK == 0); ramd_last_error() gives text.
 *     (The reference aborts via LOG_INFO+exit(1), src/utils/log.hpp:95-100; the C++ API
 *      layer in include/rocalution/ restores that behaviour on top of these codes.)
 *   - "optional" backend operations that the reference lets a backend decline by returning
 *     false (host fallback protocol, src/base/local_matrix.cpp:2299-2340) return
 *     RAMD_ERR_UNSUPPORTED here; nothing in this library falls back to host compute.
 *   - one host thread drives the library; work is queued on the CURRENT stream
 *     (ramd_compute_default/interior/ghost switch it, like the reference).
 *   - value types: fp64 and fp32; index vectors: int32.  Row offsets and column indices
 *     are int32 (the reference's default PtrType / int, src/utils/types.hpp.in:30-32).
 * ========================================================================== */
#ifndef ROCALUTION_AMD_H_
#define ROCALUTION_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---- */
enum
{
    RAMD_OK              = 0,
    RAMD_ERR_HIP         = 1, /* a HIP runtime call failed */
    RAMD_ERR_ARG         = 2, /* bad argument (the reference asserts) */
    RAMD_ERR_UNSUPPORTED = 3, /* optional op not provided by this backend ("return false") */
    RAMD_ERR_REFUSED     = 4, /* conversion refused by the reference's own rule (ELL width) */
    RAMD_ERR_NO_DEVICE   = 5,
    RAMD_ERR_STATE       = 6 /* object in the wrong state (not analysed, wrong format ...) */
};

/* value / index types of vectors and matrices */
enum { RAMD_F64 = 0, RAMD_F32 = 1, RAMD_I32 = 2 };

/* matrix formats: numbering of src/base/matrix_formats.hpp (CSR=1, COO=4, ELL=6, HYB=7) */
enum { RAMD_CSR = 1, RAMD_COO = 4, RAMD_ELL = 6, RAMD_HYB = 7 }; /* (5 = DIA: not provided, as MCSR / BCSR / DENSE) */

typedef struct ramd_vec_s* ramd_vec_t; /* an AcceleratorVector<T> instance */
typedef struct ramd_mat_s* ramd_mat_t; /* an AcceleratorMatrix<T> instance (any format) */

/* ======================================================================= lifecycle
 * rocalution_init_hip / rocalution_stop_hip / rocalution_info_hip / rocalution_get_arch_hip
 * (src/base/hip/backend_hip.hpp:40-60, backend_hip.cpp:50-190), called from
 * init_rocalution/stop_rocalution (src/base/backend_manager.cpp:110-290). */
int         ramd_init(int device); /* device < 0: keep the current device */
int         ramd_stop(void);
int         ramd_is_initialized(void);
int         ramd_info(char* buf, int buflen); /* human readable backend description */
const char* ramd_get_arch(void); /* "gfx950" ... */
const char* ramd_last_error(void);
void        ramd_set_last_error(const char* msg); /* used by the host layers above this ABI to leave their message here */
int         ramd_device_count(int* count);

/* rocalution_hip_sync{,_default,_interior,_ghost} (backend_hip.hpp:48-57) */
int ramd_sync(void);
int ramd_sync_default(void);
int ramd_sync_interior(void);
int ramd_sync_ghost(void);
/* rocalution_hip_compute_{default,interior,ghost} (backend_hip.hpp:63-69): select the
 * stream every following op is queued on (reference: HIP_stream_current). */
int   ramd_compute_default(void);
int   ramd_compute_interior(void);
int   ramd_compute_ghost(void);
void* ramd_current_stream(void); /* hipStream_t, for callers that interleave own work */

/* allocate_pinned<T>/free_pinned<T> (src/base/hip/hip_allocate_free.hpp) */
int ramd_alloc_pinned(void** ptr, int64_t bytes);
int ramd_free_pinned(void* ptr);

/* ======================================================================= vectors
 * factory: _rocalution_init_base_hip_vector<T> (backend_hip.hpp:73). */
int ramd_vec_create(int dtype, ramd_vec_t* out);
int ramd_vec_destroy(ramd_vec_t v);
int ramd_vec_allocate(ramd_vec_t v, int64_t n); /* BaseVector::Allocate (base_vector.hpp:63), zero-filled */
int ramd_vec_clear(ramd_vec_t v); /* ::Clear :71 */
int ramd_vec_size(ramd_vec_t v, int64_t* n);
int ramd_vec_dtype(ramd_vec_t v, int* dtype);
void* ramd_vec_data(ramd_vec_t v); /* raw device pointer (LeaveDataPtr-style view, :68) */
/* Allocate (base_vector.hpp:68, hip_vector.cpp:117-150) with a placement hint: the block comes from the placement class
 * opposite to the one `other` lives in -- two vectors a fused update WRITES in one pass stream 8-15 % faster from different
 * classes (csrc/backend.hip, "Placement classes").  other == NULL or unclassified (< 64 MiB): plain ramd_vec_allocate. */
int ramd_vec_allocate_apart(ramd_vec_t v, int64_t n, ramd_vec_t other);
/* ... for a vector that is already allocated: how fast ONE kernel writes this vector and `other` together is measured (a
 * write pass over both blocks, contents saved and restored), and the vector moves to the best of a few fresh candidate
 * blocks if one is clearly faster; *moved tells.  Same sizes and types only, blocks from 64 MiB on; otherwise a no-op. */
int ramd_vec_place_apart(ramd_vec_t v, ramd_vec_t other, int* moved);
/* ... and by trial: `run(ctx)` launches the kernels that use the vector on the current stream (return RAMD_OK); it is timed
 * with the vector in its own block and in up to `tries` fresh ones, and the vector moves to the fastest (contents kept; what
 * `run` does to other vectors is the caller's business).  `run` must not contain a collective: whether this rank has the
 * memory for a candidate block is its own affair, and ranks leaving the trials at different points would hang the others
 * (the solvers time kernels on the rank's own vectors only).  stop_ratio = 0: up to `tries` trials; stop_ratio in (0, 1): stop once the best time is below stop_ratio x the worst one seen.  apart_from != NULL: the
 * first candidates are drawn from the placement class that vector is not in.  No-op below 64 MiB.  Which pairs and triples
 * of big blocks stream well together is decided by their physical placement and is only partly predicted by the placement
 * class.  The fused CG and BiCGStab loops place the vectors their update kernels write this way at their first Solve, by
 * default, for vectors of 64 MiB and more (RAMD_ALLOC_CLASSES=0 switches every placement measurement off).
 * Placement is an optimisation and never the reason a solve runs out of memory: every spare block (saved contents,
 * candidates, the 1-GiB reference block of the class probe) is only taken while it leaves a sixteenth of the device (at
 * least 2 GiB) free -- ramd_placement_room answers that question for `blocks` blocks of `bytes` -- and a failed allocation
 * inside a placement call means "stay where you are" (RAMD_OK, *moved = 0). */
int ramd_placement_room(int64_t bytes, int blocks, int* ok);
/* wall time this process has spent measuring placements so far (ramd_vec_place_apart, ramd_vec_place_by_trial); reset != 0
 * sets it back to zero */
int ramd_placement_seconds(double* seconds, int reset);
typedef int (*ramd_trial_cb)(void* ctx);
int ramd_vec_place_by_trial(ramd_vec_t v, ramd_trial_cb run, void* ctx, int tries, double stop_ratio, ramd_vec_t apart_from,
                            int* moved);
/* 0 / 1: the placement class of the vector's block, -1: not classified (small, host, empty) */
int ramd_vec_placement_class(ramd_vec_t v, int* cls);
int ramd_vec_zeros(ramd_vec_t v); /* :73 */
int ramd_vec_ones(ramd_vec_t v); /* :75 */
int ramd_vec_set_values(ramd_vec_t v, double val); /* :77 */
/* AcceleratorVector::CopyFromHost / CopyToHost (base_vector.hpp:224-226), CopyFromHostData :114 */
int ramd_vec_copy_from_host(ramd_vec_t v, const void* host);
int ramd_vec_copy_to_host(ramd_vec_t v, void* host);
int ramd_vec_copy_from(ramd_vec_t v, ramd_vec_t src); /* CopyFrom :85 (resizes like the reference) */
int ramd_vec_copy_from_offset(ramd_vec_t v, ramd_vec_t src, int64_t src_offset, int64_t dst_offset,
                              int64_t size); /* CopyFrom(src,so,do,size) :98 */
int ramd_vec_copy_from_float(ramd_vec_t v_f64, ramd_vec_t src_f32); /* :89 */
int ramd_vec_copy_from_double(ramd_vec_t v_f32, ramd_vec_t src_f64); /* :91 */
int ramd_vec_copy_from_permute(ramd_vec_t v, ramd_vec_t src, ramd_vec_t perm_i32); /* :105 v[p[i]]=src[i] */
int ramd_vec_copy_from_permute_backward(ramd_vec_t v, ramd_vec_t src, ramd_vec_t perm_i32); /* :109 */
int ramd_vec_add_scale(ramd_vec_t v, ramd_vec_t x, double alpha); /* AddScale :126  v = v + alpha*x */
int ramd_vec_scale_add(ramd_vec_t v, double alpha, ramd_vec_t x); /* ScaleAdd :128  v = alpha*v + x */
int ramd_vec_scale_add_scale(ramd_vec_t v, double alpha, ramd_vec_t x, double beta); /* :130 */
/* the sub-range form (host_vector.cpp:693-720): v[dst+i] = alpha*v[dst+i] + beta*x[src+i], i < size */
int ramd_vec_scale_add_scale_offset(ramd_vec_t v, double alpha, ramd_vec_t x, double beta, int64_t src_offset,
                                    int64_t dst_offset, int64_t size);
int ramd_vec_scale_add2(ramd_vec_t v, double alpha, ramd_vec_t x, double beta, ramd_vec_t y,
                        double gamma); /* ScaleAdd2 :142 */
int ramd_vec_scale(ramd_vec_t v, double alpha); /* Scale :149 */
int ramd_vec_dot(ramd_vec_t v, ramd_vec_t x, double* result); /* Dot :151 / DotNonConj :153 (blocking) */
int ramd_vec_norm(ramd_vec_t v, double* result); /* Norm :155  sqrt(sum x^2) */
int ramd_vec_reduce(ramd_vec_t v, double* result); /* Reduce :157 */
int ramd_vec_asum(ramd_vec_t v, double* result); /* Asum :163 */
int ramd_vec_amax(ramd_vec_t v, double* value, int64_t* index); /* Amax :165 */
int ramd_vec_pointwise_mult(ramd_vec_t v, ramd_vec_t x); /* :167  v = v*x */
int ramd_vec_pointwise_mult2(ramd_vec_t v, ramd_vec_t x, ramd_vec_t y); /* :169  v = y*x */
int ramd_vec_get_index_values(ramd_vec_t v, ramd_vec_t index_i32, ramd_vec_t out); /* :175 halo pack */

/* ======================================================================= matrices
 * factory: _rocalution_init_base_hip_matrix<T>(desc, format, blockdim) (backend_hip.hpp:78);
 * here one object can hold any of the four formats and ramd_mat_convert switches it. */
int ramd_mat_create(int dtype, ramd_mat_t* out);
int ramd_mat_destroy(ramd_mat_t m);
int ramd_mat_clear(ramd_mat_t m); /* BaseMatrix::Clear (base_matrix.hpp:167) */
int ramd_mat_info(ramd_mat_t m, int* nrow, int* ncol, int64_t* nnz, int* format, int* dtype);
/* AcceleratorMatrix::CopyFromHost / CopyFromHostCSR (base_matrix.hpp:847, :262): host CSR in */
int ramd_mat_set_csr_from_host(ramd_mat_t m, int nrow, int ncol, int64_t nnz, const int32_t* row_offset,
                               const int32_t* col, const void* val);
/* CopyToHost / CopyToCSR (:853, :253); only valid in CSR format */
int ramd_mat_copy_csr_to_host(ramd_mat_t m, int32_t* row_offset, int32_t* col, void* val);
int ramd_mat_clone(ramd_mat_t src, ramd_mat_t* out); /* CopyFrom :238 (LocalMatrix::CloneFrom) */
int ramd_mat_cast(ramd_mat_t src_f64, ramd_mat_t* out_f32); /* value-cast CSR copy (mixed_precision.cpp:201-229) */
/* ConvertFrom :235 -- layout rules of src/base/host/host_conversion.cpp:621-687 (ELL, may return
 * RAMD_ERR_REFUSED and leave the matrix CSR) and :1117-1239 (HYB); COO :582 */
int ramd_mat_convert(ramd_mat_t m, int format);
/* ELL/HYB/COO raw views for tests (device -> host) */
int ramd_mat_ell_info(ramd_mat_t m, int* width, int64_t* coo_nnz);
int ramd_mat_copy_ell_to_host(ramd_mat_t m, int32_t* ell_col, void* ell_val);
int ramd_mat_copy_coo_to_host(ramd_mat_t m, int32_t* row, int32_t* col, void* val);
/* Apply :450 / ApplyAdd :452 -- y = A x ; y += scalar * A x, in the matrix' current format */
int ramd_mat_apply(ramd_mat_t m, ramd_vec_t x, ramd_vec_t y);
int ramd_mat_apply_add(ramd_mat_t m, ramd_vec_t x, double scalar, ramd_vec_t y);
/* what the CSR product learned about the matrix on its first call (no reference counterpart): state 0 = not analysed yet,
 * 1 = the rows fall into `entries` patterns of column offsets (col - row) of at most `width` entries, the kernel rebuilds the
 * columns from one byte per row; -1 = not structured (too many patterns or rows longer than 28): columns are read;
 * 2 = no dictionary, but most rows carry the column list of the row before them (the unknowns of one mesh node of an FE
 * matrix): only the first row of such a group has its columns read */
int ramd_mat_pattern_info(ramd_mat_t m, int* state, int* entries, int* width);
/* on = 0: the products of this matrix read its stored columns even where a dictionary exists (the general CSR / ELL path:
 * bench.py reports both figures); on != 0 (default): use the dictionary where the matrix is structured.  Results are
 * bit-identical either way (same values, same order of additions). */
int ramd_mat_pattern_use(ramd_mat_t m, int on);
int ramd_mat_extract_diag(ramd_mat_t m, ramd_vec_t d); /* :193 */
int ramd_mat_extract_inv_diag(ramd_mat_t m, ramd_vec_t d); /* :195 ; *d resized to min(nrow,ncol) */
int ramd_mat_extract_submatrix(ramd_mat_t m, int row_offset, int col_offset, int row_size, int col_size,
                               ramd_mat_t out); /* :186 */
int ramd_mat_permute(ramd_mat_t m, ramd_vec_t perm_i32); /* :206  P A P^T */
/* MultiColoring (host-serial greedy in BOTH reference backends: host_matrix_csr.cpp:2469-2599,
 * hip_matrix_csr.cpp:3915-4060).  size_colors must hold nrow ints. */
int ramd_mat_multicoloring(ramd_mat_t m, int* num_colors, int* size_colors, ramd_vec_t perm_i32);
int ramd_mat_ilu0_factorize(ramd_mat_t m); /* :321 */
/* LocalMatrix::ILUpFactorize(p, level) (local_matrix.cpp:3910-4040): p = 0 is ILU(0); level != 0: fill levels on the
 * pattern of A^(p+1) (host_matrix_csr.cpp:3149-3312), level == 0: ILU(0) on that whole pattern.  In place, CSR. */
int ramd_mat_ilup_factorize(ramd_mat_t m, int p, int level);
/* Incomplete Cholesky (IC preconditioner, preconditioner.cpp:862-925):
 *   ICFactorize (host_matrix_csr.cpp:2344-2466) in place on L = ExtractL(A, diag = true); returns the inverse
 *   diagonal; RAMD_ERR_STATE on the reference's "IC breakdown" conditions.
 *   LLAnalyse / LLSolve(in, inv_diag, out) (:1294-1341): L y = b, L^T x = y, both scaled by inv_diag. */
int ramd_mat_ic_factorize(ramd_mat_t m, ramd_vec_t inv_diag);
int ramd_mat_ll_analyse(ramd_mat_t m);
int ramd_mat_ll_analyse_clear(ramd_mat_t m);
int ramd_mat_ll_solve(ramd_mat_t m, ramd_vec_t in, ramd_vec_t inv_diag, ramd_vec_t out);
/* Iterative triangular solves, TriSolverAlg_Iterative (solver.hpp:33-64; host_matrix_csr.cpp:1469-2092 ItLU* / ItLL* /
 * ItL* / ItU*; sweeps host_sparse.cpp:195-530): Jacobi sweeps started from the content of `out`, at most max_iter per
 * triangle, stopped when the sweep's max-norm figure is <= tol (use_tol != 0); the sweep count a tolerance stop leaves
 * behind caps the second triangle (one max_iter variable in the reference).  ItLU / ItLL keep their intermediate
 * vector between solves (zero after the analysis). */
int ramd_mat_it_lu_analyse(ramd_mat_t m);
int ramd_mat_it_lu_analyse_clear(ramd_mat_t m);
int ramd_mat_it_lu_solve(ramd_mat_t m, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out);
int ramd_mat_it_ll_analyse(ramd_mat_t m);
int ramd_mat_it_ll_analyse_clear(ramd_mat_t m);
int ramd_mat_it_ll_solve(ramd_mat_t m, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out);
int ramd_mat_it_l_analyse(ramd_mat_t m, int diag_unit);
int ramd_mat_it_l_analyse_clear(ramd_mat_t m);
int ramd_mat_it_l_solve(ramd_mat_t m, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out);
int ramd_mat_it_u_analyse(ramd_mat_t m, int diag_unit);
int ramd_mat_it_u_analyse_clear(ramd_mat_t m);
int ramd_mat_it_u_solve(ramd_mat_t m, int max_iter, double tol, int use_tol, ramd_vec_t in, ramd_vec_t out);
int ramd_mat_lu_analyse(ramd_mat_t m); /* :344 */
int ramd_mat_lu_analyse_clear(ramd_mat_t m); /* :346 */
int ramd_mat_lu_solve(ramd_mat_t m, ramd_vec_t in, ramd_vec_t out); /* :349 */
/* statistics of the triangular-solve plans of the most recent LUAnalyse / LAnalyse / UAnalyse of this process (a measurement
 * hook, no reference counterpart; which = 0 lower, 1 upper).  Process-global and not thread-safe: analyses running concurrently
 * in several host threads overwrite each other's record.
 * out[0] form: 1 level-scheduled rows, 2 box tiles in record form, 3 box tiles with row groups, 4 lattice pencils, 6 row groups
 * handed from wave to wave (sync-free grouped form, trsv_syncfree.hip; 5 was a form removed in round 5).
 * Forms 1-4: [1] rows; [2] dependency levels; [3] tiles / pencils; [4] steps of all tiles; [5] values handed from tile to tile
 * per solve; [6] most rows of a tile; [7] longest triangular row; [8] lanes per row; [9..11] box edges in the three dependency
 * coordinates (lattice: nx, ny, nz); [12] bytes of the plan (lattice form) / for form 1 the reason the box-tile form was not
 * taken: 1 no chains of consecutively numbered dependent rows, 2 no dependencies, 3 rows longer than 32 entries without row
 * groups, 4 / 5 index ranges, 6 tiles do not fit the LDS, 7 too few rows, 8 switched off, 9 row groups with more than 24 entries
 * outside the group; [13] chains; [14] most steps of a tile; [15] most external values of a tile.
 * Form 6: [1] rows; [2] GROUP levels (one hand-off each); [3] units (whole row groups of one group level, one wave each);
 * [4] row groups; [5] 0; [6] rows a unit holds at most (64 / lanes per row); [7] most entries of a row outside its group;
 * [8] lanes per row; [9] rows of the longest group; [12] the reason the box-tile form was not taken (as above); [14] bytes of the
 * out-of-group entries as stored (positions + coefficients).
 * Form 7 (sheared pencils of the 27-point stencil, csrc/trsv_box27.hip): [2] dependency levels (planes x + 2 y + 4 z); [3] pencils;
 * [4] pencils x steps; [6] 64; [7] 13; [8] 1; [9]-[11] the lattice; [12] bytes of the packed coefficients */
int ramd_tri_plan_stats(int which, long long* out16);
/* test hook of the sync-free grouped form's division (trsv_syncfree.hip sf_div: the fp64 division sequence of gfx950 with the
 * part that depends on the divisor alone formed once per plan; no reference counterpart -- the reference divides,
 * host_matrix_csr.cpp:1216).  Device pointers, n elements each: fast[i] = sf_div(a[i], d[i]), plain[i] = a[i] / d[i],
 * in_window[i] = 1 where the short sequence is what produced fast[i] (both operands inside its exponent window) */
int ramd_selftest_sf_div(long long n, const double* a, const double* d, double* fast, double* plain, int* in_window);
int ramd_mat_l_analyse(ramd_mat_t m, int diag_unit); /* :365 */
int ramd_mat_l_analyse_clear(ramd_mat_t m);
int ramd_mat_l_solve(ramd_mat_t m, ramd_vec_t in, ramd_vec_t out); /* :370 */
int ramd_mat_u_analyse(ramd_mat_t m, int diag_unit); /* :375 */
int ramd_mat_u_analyse_clear(ramd_mat_t m);
int ramd_mat_u_solve(ramd_mat_t m, ramd_vec_t in, ramd_vec_t out); /* :380 */

/* device-side synthetic operator: 3-D 7-point Poisson N^3 in CSR */
int ramd_mat_gen_poisson7(ramd_mat_t m, int N);
/* the reference's own 3-D test operator, generated on the device: the 27-point Laplacian of gen_3d_laplacian
 * (clients/include/utility.hpp:110-177: 26 on the diagonal, -1 at every lattice neighbour of the 3 x 3 x 3 box, ascending
 * columns) on an nx x ny x nz lattice, x fastest (the reference generates cubes: nx = ny = nz = ndim) */
int ramd_mat_gen_laplace27(ramd_mat_t m, int nx, int ny, int nz);
/* the planes [z_begin, z_end) of that operator split into interior and ghost parts like ramd_mat_gen_poisson7_slab (ghost columns:
 * [plane z_begin - 1 | plane z_end], in-plane index y nx + x): a rank's piece in the reference's MPI generator
 * (clients/include/common.hpp:926-1249) */
int ramd_mat_gen_laplace27_slab(ramd_mat_t interior, ramd_mat_t ghost, int nx, int ny, int nz, int z_begin, int z_end);
/* rows [row_begin,row_end) of the same operator split into interior (local columns) and ghost
 * (remote columns, renumbered into the halo receive buffer) parts -- the per-rank pieces a
 * GlobalMatrix holds (src/base/global_matrix.cpp:913-921). */
int ramd_mat_gen_poisson7_slab(ramd_mat_t interior, ramd_mat_t ghost, int N, int64_t row_begin,
                               int64_t row_end);

/* ======================================================================= fused hot-path ops
 * New entry points (no counterpart in the reference's plugin interface): single-launch
 * fusions of the BLAS-1 sequences of the Krylov loops.  Scalars live in a device record of
 * RAMD_NSCALARS doubles; ramd_scalars_fetch copies a record to the host (blocking on the
 * stream).  Element-wise arithmetic is the reference's expression for each op, so results
 * equal the unfused sequence except for the summation order of the reductions. */
enum { RAMD_NSCALARS = 512 };
/* ---- device-resident scalar algebra (the recurrences of the Krylov drivers without host round trips).
 * The reference computes every recurrence coefficient on the host: each Dot / Norm is a blocking read-back
 * (hip_vector.cpp:785-931 + hipStreamSynchronize), the quotient is formed in C++ and travels back as a kernel argument
 * (src/solvers/krylov/{cr,fcg,bicgstabl,qmrcgstab,idr}.cpp).  Here a dot lands in a slot of the device record
 * (ramd_fused_multi_dot), a short program of scalar operations runs on the record in ONE single-thread launch, and the
 * vector updates take their coefficients from slots -- the host reads one residual per iteration.
 * A program is a list of {op, dst, a, b, imm}: dst = a (+,-,*,/) b on slots, SET dst = imm, NEG / SQRT / ABS / MOV of a,
 * ZFLAG: dst = 1 if slot a == 0 (else unchanged) -- the breakdown tests of the drivers (rho == 0 ...) without a read-back:
 * every later ramd_vec_combine_s guarded by that slot becomes a no-op, the host sees the flag with the next residual.
 * single != 0: every result is rounded to float (drivers instantiated for float compute their scalars in float). */
typedef struct
{
    int    op, dst, a, b;
    double imm;
} ramd_sop_t;
enum
{
    RAMD_SOP_SET = 0, RAMD_SOP_MOV, RAMD_SOP_ADD, RAMD_SOP_SUB, RAMD_SOP_MUL, RAMD_SOP_DIV, RAMD_SOP_NEG, RAMD_SOP_SQRT,
    RAMD_SOP_ABS, RAMD_SOP_ZFLAG, RAMD_SOP_BADFLAG, /* like ZFLAG, for a == 0, NaN or +-Inf (bicgstab.cpp:430-447) */
    RAMD_SOP_CMOVLT /* if slot a < slot b: dst = slot (int)imm   (idr.cpp: omega *= kappa / rho where rho < kappa) */
};
enum { RAMD_SOP_MAX = 96 }; /* operations per program */
int ramd_scalars_eval(const ramd_sop_t* ops, int count, int single);
/* x = sum_k c_k * v_k over nterms <= 3 terms, evaluated left to right exactly as the reference's AddScale / ScaleAdd /
 * ScaleAddScale / ScaleAdd2 / Scale expressions (host_vector.cpp:635-760); c_k = factor[k] * slot[slots[k]] (slots[k] >= 0)
 * or factor[k] alone; v_k may be x itself.  guard >= 0: nothing happens when that slot is non-zero. */
int ramd_vec_combine_s(ramd_vec_t x, int nterms, const ramd_vec_t* vs, const int* slots, const double* factors, int guard);
int ramd_scalars_set(int slot, double value);
int ramd_scalars_fetch(double* host, int first, int count);
int ramd_scalars_fetch_async_begin(int record, int first, int count); /* record in 0..7 */
int ramd_scalars_fetch_async_end(int record, double* host, int count);

/* y = A x  and  s[slot_dot] = <x, y>   (cg.cpp:415-418: q = A p ; p.q) */
int ramd_fused_apply_dot(ramd_mat_t m, ramd_vec_t x, ramd_vec_t y, int slot_dot);
/* y = A x  and  s[slot_dot] = <w, y>   (bicgstab.cpp:397-400: q = A z ; r0.q) */
int ramd_fused_apply_dotv(ramd_mat_t m, ramd_vec_t x, ramd_vec_t y, ramd_vec_t w, int slot_dot);
/* one sweep of FixedPoint(omega) + Jacobi (solver.cpp:686-720 with preconditioner.cpp:137-166): xnew = x + omega * dinv * (rhs - A x)
 * in one pass, the same operations as Apply, ScaleAdd(-1, rhs), PointWiseMult, AddScale; CSR only (else RAMD_ERR_UNSUPPORTED) */
int ramd_fused_jacobi_sweep(ramd_mat_t m, ramd_vec_t dinv, ramd_vec_t rhs, ramd_vec_t x, ramd_vec_t xnew, double omega);
/* y += scalar * A x  and  s[slot_dot] = <p, y>, GIVEN that s[slot_dot] already holds <p, y> of the
 * incoming y: only the rows A touches are corrected (the ghost part of GlobalMatrix::Apply,
 * global_matrix.cpp:1001-1007, followed by the interior part of GlobalVector::Dot,
 * global_vector.cpp:549-560, without a second pass over the vectors) */
int ramd_fused_apply_add_dot(ramd_mat_t m, ramd_vec_t x, double scalar, ramd_vec_t y, ramd_vec_t p,
                             int slot_dot);
/* alpha = s[slot_rho] / s[slot_pq];  r += (-alpha) q;  s[slot_rr] = <r,r>;
 * if dinv: z = dinv * r, s[slot_rz] = <r,z>   else s[slot_rz] = <r,r>            (cg.cpp:418-438) */
/* BiCGStab (bicgstab.cpp:365-489) with its scalars on the device: alpha = s[rho]/s[r0q],
 * omega = s[tr]/s[tr+1] (<t,r>, <t,t>), beta = (s[new]/s[rho]) * (alpha/omega).
 *   r_update : r += (-alpha) q
 *   xr_update: x = 1*x + alpha*dir + omega*sv ; r += (-omega) t ; s[rr] = <r,r> ; s[new] = <r0,r> ; s[flag] = 0
 *              (dir = sv = NULL: the unpreconditioned form, dir = p and sv = old r);
 *              omega 0/NaN/Inf: only x += alpha*p and s[flag] = 1 -- the caller runs the reference's
 *              breakdown branch (:430-447)
 *   direction: p = beta*p + (-beta*omega)*q + 1*r */
int ramd_fused_bicg_r_update(ramd_vec_t r, ramd_vec_t q, int slot_rho, int slot_r0q);
int ramd_fused_bicg_xr_update(ramd_vec_t x, ramd_vec_t dir, ramd_vec_t sv, ramd_vec_t r, ramd_vec_t t,
                              ramd_vec_t r0, ramd_vec_t p, int slot_rho, int slot_r0q, int slot_tr,
                              int slot_rr, int slot_new, int slot_flag);
int ramd_fused_bicg_direction(ramd_vec_t p, ramd_vec_t q, ramd_vec_t r, int slot_rho, int slot_r0q, int slot_tr,
                              int slot_new);
int ramd_fused_cg_update(ramd_vec_t r, ramd_vec_t q, ramd_vec_t dinv, ramd_vec_t z, int slot_rho,
                         int slot_pq, int slot_rr, int slot_rz);
/* alpha = s[slot_rho] / s[slot_pq];  beta = s[slot_new] / s[slot_rho];
 * x = x + alpha*p (cg.cpp:421, with the OLD p);  p = beta*p + z (cg.cpp:441-442).  Moving the x update
 * next to the direction update reads p once per iteration instead of twice. */
int ramd_fused_cg_direction(ramd_vec_t x, ramd_vec_t p, ramd_vec_t z, int slot_rho, int slot_pq,
                            int slot_new);
/* multi-coloured SGS apply as 2*nb-1 fused colour sweeps on the permuted matrix P A P^T (arithmetic of
 * MultiColored::Solve decomposed form, preconditioner_multicolored.cpp:348-413, _gs.cpp:127-199) */
typedef struct ramd_mcsgs_s* ramd_mcsgs_t;
int ramd_mcsgs_build(ramd_mat_t permuted, int num_blocks, const int* block_sizes, ramd_vec_t perm_i32,
                     ramd_mcsgs_t* out);
int ramd_mcsgs_apply(ramd_mcsgs_t h, ramd_vec_t rhs, ramd_vec_t x);
/* which form of the SGS apply Build() chose (the counterpart of ramd_tri_plan_stats for the colour sweeps; the reference has no
 * such query -- its MultiColored::Build, preconditioner_multicolored.cpp:303-340, has one form).  out8[0]: 0 = one sweep per
 * colour and direction (k_mc_sweep), 1 = the same with colour 0's forward sweep folded into its readers, 2 = both colours of a
 * red-black lattice operator in one pass (k_mc_rb); [1] colours; [2] / [3] 1 = the lower / upper part runs on row patterns;
 * [4..6] lattice extents of form 2; [7] rows.  h == NULL: the plan built last in this process (not thread-safe: a diagnostic). */
int ramd_mcsgs_info(ramd_mcsgs_t h, int64_t* out8);
/* the same sweep plan applied as MultiColoredGS (backward sweep only,
 * preconditioner_multicolored_gs.cpp:250-288) or, when `permuted` held the ILU(0) factors of P A P^T,
 * as MultiColoredILU(0,1) (preconditioner_multicolored_ilu.cpp:187-232) */
enum { RAMD_MC_SGS = 0, RAMD_MC_GS = 1, RAMD_MC_ILU = 2 };
int ramd_mcsgs_apply_kind(ramd_mcsgs_t h, int kind, ramd_vec_t rhs, ramd_vec_t x);
int ramd_mcsgs_destroy(ramd_mcsgs_t h);
/* several dot products against one vector in one pass: s[slot0+k] = <v_k, w>, k < count */
int ramd_fused_multi_dot(const ramd_vec_t* vs, int count, ramd_vec_t w, int slot0);
/* x = x + coef[0] vs[0]; x = x + coef[1] vs[1]; ... in this order per element: a sequence of AddScale calls
 * (src/base/base_vector.hpp AddScale; the GMRES solution update, gmres.cpp:522-532) with x read and written once */
int ramd_fused_multi_axpy(ramd_vec_t x, const ramd_vec_t* vs, const double* coef, int count);
/* w += (-h) v ; s[slot_dot] = <u, w>   (one MGS step fused with the next dot, gmres.cpp:480-486);
 * h is read from s[slot_h]; u may be NULL (then only the update and s[slot_dot]=<w,w>) */
int ramd_fused_mgs_step(ramd_vec_t w, ramd_vec_t v, int slot_h, ramd_vec_t u, int slot_dot);
/* the same MGS recurrence (gmres.cpp:480-486) in blocks of up to ramd_fused_mgs_block_max() (4 in the shipped build, at most 8) basis vectors, one pass
 * per block:
 *   h_m = <v_m, w - sum_{k<m} h_k v_k> = <v_m, w> - sum_{k<m} h_k <v_k, v_m>
 * The pass first solves the PREVIOUS block's h from the sums its pass left at s[slot_eprev ...] (e_0..e_{nprev-1}, then
 * the strict upper triangle of the block's Gram matrix, row-major), stores them at s[slot_h ...] and applies
 * w -= h_0 vprev_0; w -= h_1 vprev_1; ... (per element, in this order); then it leaves the sums of the CURRENT block at
 * s[slot_e ...] in the same layout (ncur + ncur(ncur-1)/2 slots).  nprev == 0: first block (w is only read);
 * ncur == 0: last pass, s[slot_e] = <w, w> of the updated w.  A block that is followed by another is full. */
int ramd_fused_mgs_block_max(void);
int ramd_fused_mgs_block(ramd_vec_t w, const ramd_vec_t* vprev, int nprev, int slot_h, int slot_eprev,
                         const ramd_vec_t* vcur, int ncur, int slot_e);
/* v = v * (1/s[slot]) with s[slot] = sqrt(s[slot_sq]) computed on device (gmres.cpp:493-496) */
int ramd_fused_normalize(ramd_vec_t v, int slot_sq, int slot_norm);

/* ======================================================================= measurement hooks
 * HIP-event timing on the stream the kernels are launched on (bench.py): a stopwatch around any
 * region, and an optional per-launch bracket of every SpMV (Apply / fused Apply+dot) so that the
 * kernel's average duration is measured live inside a solver run. */
/* ---- LocalMatrix utilities next to the solver path (host_matrix_csr.cpp, CSR only; other formats:
 * RAMD_ERR_UNSUPPORTED, the front end converts).
 *   Gershgorin (:3465-3506): per row  sum_{j!=i}|a_ij| left to right and the stored diagonal; bounds start at 0.
 *   ExtractL / ExtractLDiagonal / ExtractU / ExtractUDiagonal (:919-1160): upper != 0 -> U part, with_diag.
 *   Scale / ScaleDiagonal / ScaleOffDiagonal (:3509-3568), AddScalar* (:3570-3630): which = 0 all, 1 the first
 *     stored diagonal entry of every row, 2 off-diagonal entries.
 *   UpdateValuesCSR: new values (host array of nnz entries) into the existing pattern. */
#ifdef RAMD_WITH_OFFSCOPE /* out of scope: not in the default build of librocalution_amd.so */
int ramd_mat_gershgorin(ramd_mat_t m, double* lambda_min, double* lambda_max);
#endif
int ramd_mat_extract_tri(ramd_mat_t m, ramd_mat_t out, int upper, int with_diag);
/* CSR matrix algebra (host_matrix_csr.cpp): Sort :3812-3846 (stable, by column), Transpose(T) :3757-3806,
 * MatrixAdd :3324-3462 (this = alpha*this + beta*other; structure == 0: pattern of other is a subset; != 0: union
 * pattern; rows sorted), MatMatMult :2805-2938 (C = A*B, products summed in the host's order, rows sorted) */
/* Unsmoothed-aggregation AMG setup, CoarseningStrategy PMIS (local_matrix.cpp:6519-6640 AMGPMISAggregate: strong
 * connections, PMIS rounds, root nodes, aggregate ranks, two passes for the unassigned rows; :6852-6930
 * AMGUnsmoothedAggregation: P with one entry per aggregated row).  Int vectors (the reference: bool / int64_t). */
int ramd_mat_amg_pmis_aggregate(ramd_mat_t m, double eps, ramd_vec_t connections, ramd_vec_t aggregates,
                                ramd_vec_t aggregate_root_nodes);
/* Ruge-Stueben AMG (local_matrix.cpp RSPMISCoarsening / RSDirectInterpolation): C/F splitting by PMIS (cfmap: 1 coarse,
 * 2 fine; S: strong influences per entry) and direct interpolation */
#ifdef RAMD_WITH_OFFSCOPE /* out of scope: not in the default build of librocalution_amd.so */
int ramd_mat_rs_pmis_coarsening(ramd_mat_t m, float eps, ramd_vec_t cfmap, ramd_vec_t S);
int ramd_mat_rs_direct_interpolation(ramd_mat_t m, ramd_vec_t cfmap, ramd_vec_t S, ramd_mat_t prolong);
#endif
/* AMGGreedyAggregate (local_matrix.cpp:6409-6517; host sweep host_matrix_csr.cpp:4841-4938), the reference's default
 * CoarseningStrategy: same aggregates as the sequential sweep; RAMD_ERR_UNSUPPORTED for a non-symmetric strength graph */
int ramd_mat_amg_greedy_aggregate(ramd_mat_t m, double eps, ramd_vec_t connections, ramd_vec_t aggregates,
                                  ramd_vec_t aggregate_root_nodes);
int ramd_mat_amg_unsmoothed_prolong(ramd_mat_t m, ramd_vec_t aggregates, ramd_vec_t aggregate_root_nodes,
                                    ramd_mat_t prolong);
/* AMGSmoothedAggregation (local_matrix.cpp:6642-6760; host_matrix_csr.cpp:5936-6330): P = (I - relax D_f^-1 A_f) P_tent
 * on the strength-filtered matrix, lumping_strat 0 adds / 1 subtracts the weak couplings to the diagonal */
int ramd_mat_amg_smoothed_prolong(ramd_mat_t m, double relax, int lumping_strat, ramd_vec_t connections,
                                  ramd_vec_t aggregates, ramd_vec_t aggregate_root_nodes, ramd_mat_t prolong);
/* FSAI (host_matrix_csr.cpp:6514-6662): m becomes the factorised sparse approximate inverse factor on the lower pattern
 * of the operator's power (power >= 1: pattern of A^power, :6532-6538), or of a pattern matrix handed in */
#ifdef RAMD_WITH_OFFSCOPE /* out of scope: not in the default build of librocalution_amd.so */
int ramd_mat_fsai(ramd_mat_t m, int power);
int ramd_mat_fsai_pattern(ramd_mat_t m, ramd_mat_t pattern); /* FSAI(power, pattern != NULL), :6525-6531 */
#endif
/* SPAI (host_matrix_csr.cpp:6665-6780): m becomes the sparse approximate inverse on its own pattern (per row a dense
 * least-squares problem solved by Householder QR, host_matrix_dense.cpp:361-520) */
#ifdef RAMD_WITH_OFFSCOPE /* out of scope: not in the default build of librocalution_amd.so */
int ramd_mat_spai(ramd_mat_t m);
#endif
int ramd_mat_diag_mult(ramd_mat_t m, ramd_vec_t diag, int left); /* DiagonalMatrixMultL (1) / R (0), :3631-3676 */
int ramd_mat_sort(ramd_mat_t m);
int ramd_mat_transpose(ramd_mat_t m, ramd_mat_t out);
int ramd_mat_matrix_add(ramd_mat_t m, ramd_mat_t other, double alpha, double beta, int structure);
int ramd_mat_mat_mult(ramd_mat_t c, ramd_mat_t a, ramd_mat_t b);
int ramd_mat_scale_values(ramd_mat_t m, double alpha, int which);
int ramd_mat_add_scalar_values(ramd_mat_t m, double alpha, int which);
int ramd_mat_update_values(ramd_mat_t m, const void* host_val);
/* free / total device memory in bytes (hipMemGetInfo): leak checks, sizing */
int ramd_mem_info(uint64_t* free_bytes, uint64_t* total_bytes);
int ramd_timer_start(void); /* records an event on the current stream */
int ramd_timer_stop(double* elapsed_ms); /* records, synchronises, returns the elapsed time */
int ramd_prof_spmv_enable(int on); /* bracket SpMV launches with event pairs (ring of 8192) */
int ramd_prof_spmv_result(int* launches, double* avg_ms, double* min_ms, double* max_ms);
/* the same for the other launch kinds of a Krylov iteration (measurement only; the reference has no counterpart --
 * its benchmark driver brackets whole calls with host timers, clients/samples/benchmark.cpp:62-226):
 *   SPMV      every SpMV launch (Apply / ApplyAdd / fused Apply+dot) on the stream it runs on
 *   TRSV      every sparse triangular solve launch (one per triangle of LUSolve / LSolve / USolve / LLSolve)
 *   HALO      every halo exchange (grouped ncclSend/ncclRecv) on the ghost stream
 *   HALO_WAIT the part of a halo exchange the compute stream had to wait for (exposed, not overlapped)
 *   ALLREDUCE every scalar all-reduce (counted; timed on the compute stream)
 *   VEC       the fused vector-update launches of the Krylov loops (k_cg_update, k_cg_direction, k_mgs_step, ...)
 * ramd_prof_count: occurrences since the channel was enabled (not limited by the event ring). */
enum
{
    RAMD_PROF_SPMV      = 0,
    RAMD_PROF_TRSV      = 1,
    RAMD_PROF_HALO      = 2,
    RAMD_PROF_HALO_WAIT = 3,
    RAMD_PROF_ALLREDUCE = 4,
    RAMD_PROF_VEC       = 5,
    RAMD_PROF_PRECOND   = 6, /* one multi-colour preconditioner apply (all its colour sweeps) */
    RAMD_PROF_NCHAN     = 7
};
int ramd_prof_enable(int channel, int on);
int ramd_prof_result(int channel, int* launches, double* avg_ms, double* min_ms, double* max_ms);
int ramd_prof_count(int channel, int64_t* count);

/* ======================================================================= communicator
 * Replaces the reference's MPI layer for the hot path (src/utils/communicator.cpp:41-95 allreduce,
 * :606-748 Isend/Irecv/Waitall, used by GlobalMatrix::Apply src/base/global_matrix.cpp:924-1009 and
 * GlobalVector::Dot/Norm src/base/global_vector.cpp:547-588).  One process per GPU.
 *   rccl     : device buffers go straight over xGMI (ncclSend/ncclRecv group on the ghost stream,
 *              ncclAllReduce of a short fp64 scalar vector); bootstrap = 128-byte unique id that the
 *              launcher (torch.distributed, MPI, ...) broadcasts.
 *   callback : host-staged exchange through user callbacks (the reference's own pattern); used to
 *              run >1 rank on a single GPU in tests and as a fallback transport.
 */
typedef struct ramd_comm_s* ramd_comm_t;
typedef int (*ramd_exchange_cb)(void* user, int npeers, const int* peers, const void* send_host,
                                const int64_t* send_offset_bytes, void* recv_host,
                                const int64_t* recv_offset_bytes);
typedef int (*ramd_allreduce_cb)(void* user, double* values, int count);
int ramd_comm_unique_id(char id[128]);
int ramd_comm_init_rccl(int rank, int nranks, const char id[128], ramd_comm_t* out);
int ramd_comm_init_callback(int rank, int nranks, ramd_exchange_cb exchange, ramd_allreduce_cb allreduce,
                            void* user, ramd_comm_t* out);
int ramd_comm_destroy(ramd_comm_t c);
int ramd_comm_rank(ramd_comm_t c, int* rank);
int ramd_comm_size(ramd_comm_t c, int* size);
/* all-gather of `count` 64-bit integers per rank between HOST arrays (out[q * count ..] = rank q's): the setup exchanges of the
 * distributed AMG -- what Communicator::AllGather / MPI_Allgather of src/utils/communicator.cpp do for parallel_manager.cpp */
int ramd_comm_allgather_i64(ramd_comm_t c, const int64_t* mine, int count, int64_t* out);
int ramd_comm_rccl_count(ramd_comm_t c, int* nranks); /* ncclCommCount of the data-plane communicator (0: callback transport) */
/* in-place sum over all ranks of scalar slots [first, first+count) of the device record, queued on
 * the current stream (one call for ALL scalars of a fused reduction) */
int ramd_comm_allreduce_scalars(ramd_comm_t c, int first, int count);
/* halo exchange (CommunicateAsync_/CommunicateSync_, src/base/parallel_manager.cpp:726-787):
 * begin: after the work already queued on the current stream (the pack kernel), exchange
 *        send[send_offset[k] .. send_offset[k+1]) -> peer k and recv[recv_offset[k] ..) <- peer k on
 *        the ghost stream, so it overlaps whatever is queued on the current stream next;
 * end  : the current stream waits for the exchange. */
/* COLLECTIVE (every rank of the communicator, also one without neighbours): announces an exchange plan and agrees on
 * its form -- grouped ncclSend/ncclRecv pairs, or, when some rank has more than four peers (or RAMD_COMM_HALO=allgather),
 * ONE ncclAllGather of equally padded boundary buffers from which every rank picks what it needs (the reference posts
 * one MPI_Isend/Irecv per neighbour whatever their number, parallel_manager.cpp:726-782).
 * *allgather = 0: pairs (ramd_comm_halo_begin).  *allgather = k > 0: the all-gather form; k is the plan's number -- the
 * count of this collective call, the same on every rank, NOT a function of this rank's own peers and offsets (two
 * matrices may look alike from one rank and differ on the others) -- and every rank has to call
 * ramd_comm_halo_begin_plan(c, k, ...) / _end for every exchange of this plan, also with npeers = 0.
 * ramd_comm_halo_release gives a plan's device buffers back (the owner of the plan calls it when it goes away). */
int ramd_comm_halo_select(ramd_comm_t c, int npeers, const int* peers, const int64_t* send_offset,
                          const int64_t* recv_offset, int* allgather);
int ramd_comm_halo_release(ramd_comm_t c, int plan, long long generation);
/* a number no other communicator of this process has or will have: the owner of a halo plan keeps it next to the plan number,
 * so that a late release cannot hit a plan of a newer communicator that was allocated at the same address */
int ramd_comm_generation(ramd_comm_t c, long long* generation);
int ramd_comm_halo_begin(ramd_comm_t c, ramd_vec_t send, ramd_vec_t recv, int npeers, const int* peers,
                         const int64_t* send_offset, const int64_t* recv_offset);
int ramd_comm_halo_begin_plan(ramd_comm_t c, int plan, ramd_vec_t send, ramd_vec_t recv, int npeers, const int* peers,
                              const int64_t* send_offset, const int64_t* recv_offset);
int ramd_comm_halo_end(ramd_comm_t c);

/* Aggregation AMG across the row blocks of a distributed matrix (global_matrix.cpp:2647-3121 AMGPMISAggregate, :3123-3558
 * AMGSmoothedAggregation / AMGUnsmoothedAggregation with the host kernels host_matrix_csr.cpp:5098-5660, :5936-6512).
 * ramd_mat_merge_columns: out = [interior | ghost], one CSR operator of the block's rows whose ghost_ncol ghost columns
 * follow the interior ones (ghost may be NULL: a block without ghost entries) (per row: the interior entries, then the ghost entries -- the order the reference's loops visit them in).
 * ramd_mat_amg_pmis_aggregate_global: the PMIS aggregation on such a block; plan / npeers / peers / offsets / boundary
 * (device int vector) describe the halo exchange of the matrix (as for ramd_comm_halo_begin_plan), first_row is the
 * global number of the block's row 0.  Results over the block's nrow + nghost nodes: numbers (global node numbers),
 * aggregates (global aggregate number, -2 isolated), aggregate_root_nodes (global number of the root node);
 * connections over the block's entries.  agg_first / agg_mine / agg_total: this rank's range of aggregate numbers and
 * the global count.  The numbering is the one a single rank produces on the whole matrix.  A collective: every rank of
 * the communicator calls it.  ramd_mat_amg_prolong_global: this block's rows of the prolongation (smoothed != 0: the
 * smoothed one) with GLOBAL aggregate numbers as columns (global_ncol = agg_total). */
int ramd_mat_merge_columns(ramd_mat_t interior, ramd_mat_t ghost, int ghost_ncol, ramd_mat_t out);
int ramd_mat_amg_pmis_aggregate_global(ramd_mat_t block, double eps, ramd_comm_t comm, int plan, int npeers,
                                       const int* peers, const int64_t* send_offset, const int64_t* recv_offset,
                                       ramd_vec_t boundary, int64_t first_row, ramd_vec_t numbers,
                                       ramd_vec_t connections, ramd_vec_t aggregates, ramd_vec_t aggregate_root_nodes,
                                       int64_t* agg_first, int64_t* agg_mine, int64_t* agg_total);
int ramd_mat_amg_prolong_global(ramd_mat_t block, int smoothed, double relax, int lumping_strat, ramd_vec_t connections,
                                ramd_vec_t aggregates, ramd_vec_t aggregate_root_nodes, int64_t global_ncol,
                                ramd_mat_t prolong);

/* ======================================================================= solver layer
 * C handles onto the compiled C++ API layer (include/rocalution/: Solver<Operator,Vector>::Build()/
 * Solve(), src/solvers/solver.hpp:179-444 of the reference) for callers without a C++ compiler
 * (the Python tests and bench.py).  Semantics are those of the C++ classes of the same name. */
typedef struct ramd_solver_s* ramd_solver_t;
enum { RAMD_SOLVER_CG = 0, RAMD_SOLVER_GMRES = 1, RAMD_SOLVER_BICGSTAB = 2,
       /* src/solvers/krylov/{fcg,cr,fgmres,bicgstabl,qmrcgstab}.cpp; ramd_solver_set_basis sets the
        * restart length of (F)GMRES and the order l of BiCGStab(l) */
       RAMD_SOLVER_FCG = 3, RAMD_SOLVER_CR = 4, RAMD_SOLVER_FGMRES = 5, RAMD_SOLVER_BICGSTABL = 6,
       RAMD_SOLVER_QMRCGSTAB = 7,
       RAMD_SOLVER_IDR = 8, /* idr.cpp; set_basis = SetShadowSpace, ramd_solver_set_seed = SetRandomSeed */
       /* solver.cpp:517-775 FixedPoint; relaxation / smoother flag through ramd_solver_set_params */
       RAMD_SOLVER_FIXEDPOINT = 9 };
enum { RAMD_PC_NONE = 0, RAMD_PC_JACOBI = 1, RAMD_PC_ILU0 = 2, RAMD_PC_MCSGS = 3, RAMD_PC_MCGS = 4, RAMD_PC_MCILU = 5,
       RAMD_PC_GS = 6, RAMD_PC_SGS = 7, /* preconditioner.cpp:206-257 / :302-379 */
       RAMD_PC_IC = 8, /* :862-925 */
       /* unsmoothed_amg.cpp / smoothed_amg.cpp with CoarseningStrategy PMIS, default smoothers and coarse solver */
       RAMD_PC_UAAMG = 9, RAMD_PC_SAAMG = 10,
       /* ramd_gsolver_create only: UAAMG / SAAMG on the GlobalMatrix itself (coarse levels coupled across the ranks; the
        * aggregates stay inside a rank's row block) instead of BlockJacobi around a local AMG */
       RAMD_PC_GLOBAL_UAAMG = 11, RAMD_PC_GLOBAL_SAAMG = 12 };
int ramd_solver_create(int solver, int precond, int dtype, ramd_solver_t* out);
/* MixedPrecisionDC<fp64 outer, fp32 inner>: inner solver/preconditioner kinds */
int ramd_solver_create_mixed(int inner_solver, int inner_precond, ramd_solver_t* out);
int ramd_solver_destroy(ramd_solver_t s);
int ramd_solver_init(ramd_solver_t s, double abs_tol, double rel_tol, double div_tol, int min_iter,
                     int max_iter); /* IterativeLinearSolver::Init */
int ramd_solver_init_inner(ramd_solver_t s, double abs_tol, double rel_tol, double div_tol, int max_iter);
int ramd_solver_set_basis(ramd_solver_t s, int size_basis); /* GMRES::SetBasisSize */
int ramd_solver_set_seed(ramd_solver_t s, unsigned long long seed); /* IDR::SetRandomSeed (idr.cpp:277-285) */
/* FixedPoint: p0 = SetRelaxation(omega), p1 != 0 -> FlagSmoother() */
int ramd_solver_set_params(ramd_solver_t s, double p0, double p1);
/* Solver::SetSolverDescriptor on the preconditioner (solver.cpp:293-301, SolverDescr solver.hpp:82-148): iterative != 0
 * selects TriSolverAlg_Iterative with the given sweep limit / tolerance / tolerance switch; before build */
int ramd_solver_set_tri_solver(ramd_solver_t s, int iterative, int max_iter, double tol, int use_tol);
/* parameters of the preconditioner: ILU::Set(p0 = p, p1 != 0: level) */
int ramd_solver_set_precond_params(ramd_solver_t s, double p0, double p1, double p2);
int ramd_solver_set_fused(ramd_solver_t s, int on); /* fused device loops on/off (default on) */
int ramd_solver_set_verbose(ramd_solver_t s, int verb);
int ramd_solver_set_precond_format(ramd_solver_t s, int format); /* MultiColored::SetPrecondMatrixFormat */
int ramd_solver_build(ramd_solver_t s, ramd_mat_t op); /* SetOperator + [SetPreconditioner] + Build */
int ramd_solver_solve(ramd_solver_t s, ramd_vec_t rhs, ramd_vec_t x);
int ramd_solver_precond_apply(ramd_solver_t s, ramd_vec_t rhs, ramd_vec_t x); /* M^-1 rhs (test hook) */
int ramd_solver_result(ramd_solver_t s, int* iters, int* status, double* final_res);
/* measurement hook (bench.py: W warm-up iterations and exactly K timed ones inside ONE Solve): the solver drains the device
 * and notes the wall clock when it has checked iteration `iteration`; seconds_since: now minus that instant (< 0: the
 * iteration was never reached).  Not a reference entry point. */
int ramd_solver_set_time_mark(ramd_solver_t s, int iteration);
int ramd_solver_seconds_since_time_mark(ramd_solver_t s, double* seconds);
int ramd_solver_history(ramd_solver_t s, double* buf, int cap, int* len);
int ramd_solver_num_colors(ramd_solver_t s, int* ncolors);
/* Solver::ReBuildNumeric (solver.hpp:214-218): the operator got new values in the same pattern */
int ramd_solver_rebuild_numeric(ramd_solver_t s);
int ramd_solver_clear(ramd_solver_t s);

/* LocalMatrix::ReadFileMTX (src/base/local_matrix.cpp:1269-1326, src/base/host/host_io.cpp:51-320):
 * MatrixMarket coordinate file -> sorted CSR on the accelerator, reference semantics (1-based indices,
 * pattern -> 1, symmetric/hermitian mirrored without duplicating the diagonal). */
int ramd_mat_read_mtx(const char* filename, int dtype, ramd_mat_t* out);
/* LocalMatrix::ReadFileMTX / ReadFileCSR and WriteFileMTX / WriteFileCSR (local_matrix.cpp:1269-1460;
 * formats: host_io.cpp:51-395 MatrixMarket, :497-609 + :3236-3289 "#rocALUTION binary csr file").
 * kind: 0 = MatrixMarket, 1 = rocALUTION binary CSR */
enum { RAMD_FILE_MTX = 0, RAMD_FILE_CSR = 1, RAMD_FILE_ASCII = 0, RAMD_FILE_BINARY = 1 };
int ramd_mat_read_file(const char* filename, int kind, int dtype, ramd_mat_t* out);
int ramd_mat_write_file(ramd_mat_t m, const char* filename, int kind);
/* LocalVector::ReadFileASCII / ReadFileBinary / WriteFileASCII / WriteFileBinary
 * (host_vector.cpp:415-632).  kind: 0 = ASCII (one value per line), 1 = binary (values as double) */
int ramd_vec_read_file(ramd_vec_t v, const char* filename, int kind);
int ramd_vec_write_file(ramd_vec_t v, const char* filename, int kind);
/* MultiColored::SetDecomposition (preconditioner_multicolored.cpp:140-146): false = L/U sweeps on the
 * permuted matrix (LSolve/USolve) instead of the colour-block decomposition */
int ramd_solver_set_decomposition(ramd_solver_t s, int decomp);
int ramd_solver_set_fused_sweeps(ramd_solver_t s, int on); /* MultiColored::SetFusedSweeps (default on) */

/* distributed driver: GlobalMatrix/GlobalVector + Solver<GlobalMatrix,GlobalVector> on one rank of a
 * row-block decomposition (clients/samples/cg_mpi.cpp, bicgstab_mpi.cpp of the reference).  The local
 * preconditioner (any RAMD_PC_* kind but Jacobi, which is the global diagonal: ILU(0), MC-SGS, IC, SA-/UA-AMG on
 * the interior block, ...) is wrapped in BlockJacobi as the reference samples do
 * (preconditioner_blockjacobi.cpp:80-141). */
typedef struct ramd_gsolver_s* ramd_gsolver_t;
int ramd_gsolver_create(ramd_comm_t comm, int solver, int precond, ramd_gsolver_t* out);
/* MixedPrecisionDC<fp64 Global outer, fp32 Global inner> (config 5 of BASELINE.json); inner
 * preconditioner: RAMD_PC_NONE or RAMD_PC_JACOBI */
int ramd_gsolver_create_mixed(ramd_comm_t comm, int inner_solver, int inner_precond, ramd_gsolver_t* out);
int ramd_gsolver_init_inner(ramd_gsolver_t g, double abs_tol, double rel_tol, double div_tol, int max_iter);
int ramd_gsolver_destroy(ramd_gsolver_t g);
/* rank's slab of the 3-D 7-point Poisson operator N^3: planes [z_begin, z_end) */
int ramd_gsolver_setup_poisson(ramd_gsolver_t g, int N, int z_begin, int z_end);
/* the same decomposition of the 27-point Laplacian N^3 (ramd_mat_gen_laplace27_slab): whole planes to the lower and upper neighbour */
int ramd_gsolver_setup_laplace27(ramd_gsolver_t g, int N, int z_begin, int z_end);
/* general operator: interior CSR (local columns), ghost CSR (columns = positions in the receive
 * buffer), boundary index list and neighbour offsets (ParallelManager setters) */
int ramd_gsolver_setup_csr(ramd_gsolver_t g, int64_t global_nrow, int local_nrow, int64_t int_nnz,
                           const int32_t* int_rp, const int32_t* int_ci, const double* int_val,
                           int64_t gh_nnz, const int32_t* gh_rp, const int32_t* gh_ci, const double* gh_val,
                           int npeers, const int* peers, const int* send_offset, const int* recv_offset,
                           const int* boundary_index);
int ramd_gsolver_convert(ramd_gsolver_t g, int format); /* GlobalMatrix::ConvertTo */
int ramd_gsolver_init(ramd_gsolver_t g, double abs_tol, double rel_tol, double div_tol, int min_iter,
                      int max_iter);
int ramd_gsolver_set_basis(ramd_gsolver_t g, int size_basis);
int ramd_gsolver_set_verbose(ramd_gsolver_t g, int verb);
int ramd_gsolver_build(ramd_gsolver_t g);
/* RAMD_PC_GLOBAL_* after Build: depth of the hierarchy, global rows of the coarsest operator, and the largest relative
 * defect of the Galerkin identity A_c x = R A_f P x over the levels, x random per rank (test hook: exercises the ghost
 * parts and halo plans of the coarse operators) */
int ramd_gsolver_amg_info(ramd_gsolver_t g, int* levels, int64_t* coarsest_rows, double* worst_galerkin_defect);
/* ... and a fingerprint of the operator of one level that does not depend on how the rows are distributed: its global
 * row count, this rank's entries (interior + ghost; the caller adds the ranks up) and || A_l 1 ||_2 (a collective) */
int ramd_gsolver_amg_level(ramd_gsolver_t g, int level, int64_t* global_rows, int64_t* local_entries,
                           double* norm_of_row_sums);
int ramd_gsolver_apply(ramd_gsolver_t g, const double* x_local, double* y_local); /* y = A x (test hook) */
int ramd_gsolver_solve(ramd_gsolver_t g, const double* rhs_local, double* x_local); /* NULL rhs: A*1 ; x0 = x_local or 0 */
int ramd_gsolver_solve_ones(ramd_gsolver_t g); /* rhs = A*1, x0 = 0, everything stays on the device */
int ramd_gsolver_prepare_ones(ramd_gsolver_t g); /* rhs = A*1, x = 0 (on the device) */
int ramd_gsolver_solve_device(ramd_gsolver_t g); /* Solve(rhs, &x) on the prepared device vectors */
int ramd_gsolver_result(ramd_gsolver_t g, int* iters, int* status, double* final_res);
int ramd_gsolver_set_time_mark(ramd_gsolver_t g, int iteration); /* as ramd_solver_set_time_mark */
int ramd_gsolver_seconds_since_time_mark(ramd_gsolver_t g, double* seconds);
int ramd_gsolver_dot_check(ramd_gsolver_t g, double* x_dot_x); /* <x,x> over all ranks after a solve */

#ifdef __cplusplus
}
#endif
#endif /* ROCALUTION_AMD_H_ */
