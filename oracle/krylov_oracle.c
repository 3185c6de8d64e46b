/* ==========================================================================
 * krylov_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU oracle: a plain-C restatement of the preconditioned-Krylov hot path of
 * the reference's host/OpenMP backend (rocALUTION 3.2.0, /root/reference).
 * Each function cites the reference file:line it follows.
 *
 * Who may use this: tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg -- as the checker / the reported CPU baseline only.
 * The product (rocalution_amd/) never links, imports or calls it.
 *
 * Pinning: see oracle/README.md.  The reference has no stored golden vectors
 * for this path; the oracle is pinned (1) bit-for-bit against
 * the known answers captured from the 3.2.0 host backend (BASELINE.md §2,
 * tests/golden/known_answers.json) and (2) against fixtures generated in the
 * dev container from the genuine rocALUTION host backend shipped in the image
 * (/opt/rocm/lib/librocalution.so, v4.1.0) by oracle/ref_probe (tests/golden).
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp)
 * ========================================================================== */
#define _GNU_SOURCE 1 /* M_PI */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "krylov_oracle.h"

/* ---- OpenMP policy: src/base/backend_manager.cpp:78 (threshold 10000), :590-607 ---- */
static int g_threads   = 1;
static int g_threshold = 10000;
/* SolverDescr of the preconditioners built next (src/solvers/solver.hpp:82-148): defaults direct, 30, 1e-3, tol on */
static int    g_tri_iterative = 0;
static int    g_it_max_iter   = 30;
static double g_it_tol        = 1e-3;
static int    g_it_use_tol    = 1;
void          orc_set_solver_descr(int iterative, int max_iter, double tol, int use_tol)
{
    g_tri_iterative = iterative;
    g_it_max_iter   = max_iter;
    g_it_tol        = tol;
    g_it_use_tol    = use_tol;
}

void orc_set_threads(int n)
{
    g_threads = n > 0 ? n : 1;
}
int orc_get_threads(void)
{
    return g_threads;
}
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
static int orc_threads_for(int64_t size)
{
    return (size <= g_threshold) ? 1 : g_threads;
}

/* ---- iteration control: src/solvers/iter_ctrl.cpp ------------------------------- */

typedef struct
{
    double abs_tol, rel_tol, div_tol;
    int    min_iter, max_iter;
    int    iteration;
    int    reached;
    double initial_residual, current_res;
    double* history;
    int     history_cap, history_len;
} orc_iter_ctrl;

static void ic_record(orc_iter_ctrl* ic, double res)
{
    if(ic->history && ic->history_len < ic->history_cap)
        ic->history[ic->history_len] = res;
    if(ic->history)
        ic->history_len++;
}

static void orc_ic_setup(orc_iter_ctrl* ic, const orc_solve_cfg* cfg)
{
    memset(ic, 0, sizeof(*ic));
    ic->abs_tol     = cfg->abs_tol;
    ic->rel_tol     = cfg->rel_tol;
    ic->div_tol     = cfg->div_tol;
    ic->min_iter    = cfg->min_iter;
    ic->max_iter    = cfg->max_iter;
    ic->history     = cfg->history;
    ic->history_cap = cfg->history_cap;
}

static void orc_ic_finish(const orc_iter_ctrl* ic, orc_solve_cfg* cfg)
{
    cfg->iters       = ic->iteration;
    cfg->status      = ic->reached;
    cfg->init_res    = ic->initial_residual;
    cfg->final_res   = ic->current_res;
    cfg->history_len = ic->history_len;
}

/* iter_ctrl.cpp:89-121 InitResidual: returns 0 ("false") when the solver must not iterate */
static int orc_ic_init_residual(orc_iter_ctrl* ic, double res)
{
    ic->initial_residual = res; /* current_res_ is not touched by InitResidual */
    ic->reached          = 0;
    ic->iteration        = 0;
    ic_record(ic, res);
    if((fabs(res) == INFINITY) || (res != res))
        return 0;
    if(fabs(res) <= ic->abs_tol)
    {
        ic->reached = 1;
        return 0;
    }
    return 1;
}

/* iter_ctrl.cpp:195-248 CheckResidual: NaN/Inf -> (iter>=min) abs -> rel -> max-iter ; div */
static int orc_ic_check_residual(orc_iter_ctrl* ic, double res)
{
    ic->iteration++;
    ic->current_res = res;
    ic_record(ic, res);
    if((fabs(res) == INFINITY) || (res != res))
        return 1;
    if(ic->iteration >= ic->min_iter)
    {
        if(fabs(res) <= ic->abs_tol)
        {
            ic->reached = 1;
            return 1;
        }
        if(res / ic->initial_residual <= ic->rel_tol)
        {
            ic->reached = 2;
            return 1;
        }
        if(ic->iteration >= ic->max_iter)
        {
            ic->reached = 4;
            return 1;
        }
    }
    if(res / ic->initial_residual >= ic->div_tol)
    {
        ic->reached = 3;
        return 1;
    }
    return 0;
}

/* iter_ctrl.cpp:295-306 CheckMaximumIterNoCount (FixedPoint: skip the last residual) */
static int orc_ic_check_max_iter_nocount(orc_iter_ctrl* ic)
{
    if(ic->iteration + 1 >= ic->max_iter)
    {
        ic->reached = 4;
        return 1;
    }
    return 0;
}

/* iter_ctrl.cpp:256-289 CheckResidualNoCount (GMRES restart check) */
static int orc_ic_check_residual_nocount(orc_iter_ctrl* ic, double res)
{
    if((fabs(res) == INFINITY) || (res != res))
        return 1;
    if(fabs(res) <= ic->abs_tol)
    {
        ic->reached = 1;
        return 1;
    }
    if(res / ic->initial_residual <= ic->rel_tol)
    {
        ic->reached = 2;
        return 1;
    }
    if(res / ic->initial_residual >= ic->div_tol)
    {
        ic->reached = 3;
        return 1;
    }
    if(ic->iteration >= ic->max_iter)
    {
        ic->reached = 4;
        return 1;
    }
    return 0;
}

/* ---- layout rules (type independent) --------------------------------------------- */

/* src/base/host/host_conversion.cpp:633-651: ELL width = max row nnz, refused (-1) when
 * width > 5 * (nnz / nrow) with INTEGER division */
int orc_csr_ell_width(int nrow, int64_t nnz, const int* row_offset)
{
    int max_row = 0;
    for(int i = 0; i < nrow; ++i)
    {
        int r = row_offset[i + 1] - row_offset[i];
        if(r > max_row)
            max_row = r;
    }
    if(nrow > 0 && max_row > 5 * (nnz / nrow))
        return -1;
    return max_row;
}

/* src/base/host/host_conversion.cpp:1130-1135: HYB ELL width = (nnz-1)/nrow + 1 */
int orc_csr_hyb_width(int nrow, int64_t nnz)
{
    return (int)((nnz - 1) / nrow + 1);
}

/* src/base/host/host_conversion.cpp:1153-1172: COO overflow count */
int64_t orc_csr_hyb_coo_nnz(int nrow, const int* row_offset, int ell_max_row)
{
    int64_t c = 0;
    for(int i = 0; i < nrow; ++i)
    {
        int r = row_offset[i + 1] - row_offset[i] - ell_max_row;
        c += (r > 0) ? r : 0;
    }
    return c;
}

/* src/base/host/host_matrix_csr.cpp:2469-2599  MultiColoring: greedy first-fit in natural row
 * order over row AND column (CSC) neighbours; colours from 1; perm[i] = offset[colour_i]++ */
int orc_csr_multicoloring(int nrow, int64_t nnz, const int* row_offset, const int* col,
                          int* num_colors_out, int* size_colors, int* perm)
{
    int* csc_ptr = (int*)calloc((size_t)nrow + 1, sizeof(int));
    int* csc_ind = (int*)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    for(int64_t i = 0; i < nnz; ++i)
        csc_ptr[col[i] + 1] += 1;
    for(int i = 1; i < nrow + 1; ++i)
        csc_ptr[i] += csc_ptr[i - 1];
    for(int i = 0; i < nrow; ++i)
        for(int k = row_offset[i]; k < row_offset[i + 1]; ++k)
            csc_ind[csc_ptr[col[k]]++] = i;
    for(int i = nrow; i > 0; --i)
        csc_ptr[i] = csc_ptr[i - 1];
    csc_ptr[0] = 0;

    int*  color      = (int*)calloc((size_t)(nrow > 0 ? nrow : 1), sizeof(int));
    int   num_colors = 0;
    char* row_col    = (char*)malloc((size_t)nrow + 3);
    for(int ai = 0; ai < nrow; ++ai)
    {
        color[ai] = 1;
        memset(row_col, 0, (size_t)num_colors + 2);
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
            if(ai != col[aj])
                row_col[color[col[aj]]] = 1;
        for(int aj = csc_ptr[ai]; aj < csc_ptr[ai + 1]; ++aj)
            if(ai != csc_ind[aj])
                row_col[color[csc_ind[aj]]] = 1;
        int count = row_offset[ai + 1] - row_offset[ai] + csc_ptr[ai + 1] - csc_ptr[ai];
        for(int aj = 0; aj < count; ++aj)
        {
            if(row_col[color[ai]])
                ++color[ai];
            else
                break;
        }
        if(color[ai] > num_colors)
            num_colors = color[ai];
    }
    free(csc_ptr);
    free(csc_ind);
    free(row_col);

    int* offsets_color = (int*)calloc((size_t)(num_colors > 0 ? num_colors : 1), sizeof(int));
    for(int i = 0; i < num_colors; ++i)
        size_colors[i] = 0;
    for(int i = 0; i < nrow; ++i)
        ++size_colors[color[i] - 1];
    int total = 0;
    for(int i = 1; i < num_colors; ++i)
    {
        total += size_colors[i - 1];
        offsets_color[i] = total;
    }
    for(int i = 0; i < nrow; ++i)
    {
        perm[i] = offsets_color[color[i] - 1];
        ++offsets_color[color[i] - 1];
    }
    free(color);
    free(offsets_color);
    *num_colors_out = num_colors;
    return 1;
}

/* ---- the two instantiations -------------------------------------------------------- */

#define T double
#define SUF(x) x##_f64
#define ORC_SQRT sqrt
#define ORC_FABS fabs
#include "krylov_oracle_impl.h"
#undef T
#undef SUF
#undef ORC_SQRT
#undef ORC_FABS

#define T float
#define SUF(x) x##_f32
#define ORC_SQRT sqrtf
#define ORC_FABS fabsf
#define ORC_T_IS_FLOAT 1
#include "krylov_oracle_impl.h"
#undef ORC_T_IS_FLOAT
#undef T
#undef SUF
#undef ORC_SQRT
#undef ORC_FABS

/* ---- mixed precision defect correction ---------------------------------------------
 * src/solvers/mixed_precision.cpp:159-236 (Build: value-cast CSR copy, inner solver built on it)
 * and :372-437 (SolveNonPrecond_): outer fp64 loop
 *     r = b - A x ; while(!Check(res)) { r_l=(float)r ; d_l=0 ; inner.Solve(r_l,d_l) ;
 *                                        x += (double)d_l ; r = b - A x ; res = ||r|| }
 * The inner solver's iteration control is re-initialised by every inner Solve(). */
int orc_solve_mixed(int nrow, int64_t nnz, const int* row_offset, const int* col,
                    const double* val, const double* rhs, double* x, orc_solve_cfg* outer,
                    orc_solve_cfg* inner, int* inner_iters_total)
{
    float*  val_l = (float*)malloc(sizeof(float) * (size_t)nnz);
    float*  r_l   = (float*)malloc(sizeof(float) * (size_t)nrow);
    float*  d_l   = (float*)malloc(sizeof(float) * (size_t)nrow);
    double* r_h   = (double*)malloc(sizeof(double) * (size_t)nrow);
    double* d_h   = (double*)malloc(sizeof(double) * (size_t)nrow);
    for(int64_t i = 0; i < nnz; ++i)
        val_l[i] = (float)val[i];

    orc_op_f64 A;
    op_build_f64(&A, ORC_CSR, nrow, nnz, row_offset, col, val);
    orc_op_f32 Al;
    orc_pc_f32 Pl;
    int        have_pc = inner->precond != ORC_PC_NONE;
    if(have_pc)
        pc_build_f32(&Pl, inner->precond, nrow, nnz, row_offset, col, val_l);
    op_build_f32(&Al, inner->format, nrow, nnz, row_offset, col, val_l);

    orc_iter_ctrl ic;
    orc_ic_setup(&ic, outer);
    int total = 0;

    residual_f64(&A, rhs, x, r_h);
    double res = orc_norm_f64(nrow, r_h);
    if(orc_ic_init_residual(&ic, res))
    {
        while(!orc_ic_check_residual(&ic, res))
        {
            for(int i = 0; i < nrow; ++i) /* CopyFromDouble, host_vector.cpp:295-330 */
                r_l[i] = (float)r_h[i];
            for(int i = 0; i < nrow; ++i)
                d_l[i] = 0.0f;
            orc_iter_ctrl ici;
            orc_solve_cfg icfg = *inner;
            icfg.history       = NULL;
            orc_ic_setup(&ici, &icfg);
            if(inner->solver == ORC_CG)
                solve_cg_f32(&Al, have_pc ? &Pl : NULL, r_l, d_l, &ici);
            else if(inner->solver == ORC_GMRES)
                solve_gmres_f32(&Al, have_pc ? &Pl : NULL, r_l, d_l, &ici,
                                inner->basis > 0 ? inner->basis : 30);
            else
                solve_bicgstab_f32(&Al, have_pc ? &Pl : NULL, r_l, d_l, &ici);
            total += ici.iteration;
            for(int i = 0; i < nrow; ++i) /* CopyFromFloat, host_vector.cpp:258-293 */
                d_h[i] = (double)d_l[i];
            orc_add_scale_f64(nrow, x, d_h, 1.0);
            residual_f64(&A, rhs, x, r_h);
            res = orc_norm_f64(nrow, r_h);
        }
    }
    orc_ic_finish(&ic, outer);
    if(inner_iters_total)
        *inner_iters_total = total;
    op_free_f64(&A);
    op_free_f32(&Al);
    if(have_pc)
        pc_free_f32(&Pl);
    free(val_l);
    free(r_l);
    free(d_l);
    free(r_h);
    free(d_h);
    return 1;
}
