/* krylov_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see krylov_oracle.c). */
#ifndef KRYLOV_ORACLE_H_
#define KRYLOV_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_CSR = 1, ORC_DIA = 5, ORC_ELL = 6, ORC_HYB = 7 }; /* numbering of src/base/matrix_formats.hpp */
enum { ORC_CG = 0, ORC_GMRES = 1, ORC_BICGSTAB = 2, ORC_FCG = 3, ORC_CR = 4, ORC_FGMRES = 5, ORC_BICGSTABL = 6,
       ORC_QMRCGSTAB = 7, ORC_IDR = 8, ORC_FIXEDPOINT = 9, ORC_CHEBYSHEV = 10 };
enum { ORC_PC_NONE = 0, ORC_PC_JACOBI = 1, ORC_PC_ILU0 = 2, ORC_PC_MCSGS = 3, ORC_PC_MCGS = 4, ORC_PC_MCILU = 5,
       ORC_PC_GS = 6, ORC_PC_SGS = 7, ORC_PC_IC = 8 };
#define ORC_PC_IS_MC(k) ((k) == ORC_PC_MCSGS || (k) == ORC_PC_MCGS || (k) == ORC_PC_MCILU)

typedef struct
{
    /* in */
    int    solver; /* ORC_CG / ORC_GMRES / ORC_BICGSTAB */
    int    precond; /* ORC_PC_* */
    int    format; /* operator format used by Apply during Solve */
    int    basis; /* GMRES / FGMRES restart length (default 30, gmres.cpp:50); BiCGStab(l): l (default 2);
                     IDR(s): s (default 4) */
    double p0, p1; /* FixedPoint: p0 = omega (0 -> 1), p1 != 0 -> FlagSmoother(); Chebyshev: lambda_min, lambda_max */
    unsigned long long seed; /* IDR: SetRandomSeed (shadow space P_i = SetRandomNormal((i+1)*seed)) */
    double abs_tol, rel_tol, div_tol; /* defaults 1e-15 / 1e-6 / 1e8 (iter_ctrl.cpp:52-56) */
    int    min_iter, max_iter;
    double* history; /* optional: residual per InitResidual/CheckResidual call */
    int     history_cap;
    /* out */
    int    iters, status;
    double init_res, final_res;
    int    history_len;
    /* in (appended): P > 1 = the preconditioner is BlockJacobi over P contiguous row blocks (the reference's multi-rank
     * setup: preconditioner_blockjacobi.cpp:80-141 applies the local preconditioner to the interior block of every rank;
     * row split as in clients/include/common.hpp:92-113).  Jacobi is the global diagonal either way. */
    int    nblocks;
} orc_solve_cfg;

void orc_set_threads(int n);
int  orc_get_threads(void);
int  orc_max_threads(void);
/* SolverDescr applied to every preconditioner built afterwards: TriSolverAlg_Iterative on/off, max sweeps,
 * tolerance, tolerance used (solver.hpp:82-148; defaults 0, 30, 1e-3, 1) */
void orc_set_solver_descr(int iterative, int max_iter, double tol, int use_tol);

int     orc_csr_ell_width(int nrow, int64_t nnz, const int* row_offset);
int     orc_csr_hyb_width(int nrow, int64_t nnz);
int64_t orc_csr_hyb_coo_nnz(int nrow, const int* row_offset, int ell_max_row);
int     orc_csr_multicoloring(int nrow, int64_t nnz, const int* row_offset, const int* col,
                              int* num_colors, int* size_colors, int* perm);

#define ORC_DECL(T, S)                                                                             \
    void orc_csr_apply##S(int, const int*, const int*, const T*, const T*, T*);                    \
    void orc_csr_apply_add##S(int, int64_t, const int*, const int*, const T*, const T*, T, T*);    \
    void orc_ell_apply##S(int, int, const int*, const T*, const T*, T*);                           \
    void orc_ell_apply_add##S(int, int, const int*, const T*, const T*, T, T*);                    \
    void orc_coo_apply##S(int, int64_t, const int*, const int*, const T*, const T*, T*);           \
    void orc_coo_apply_add##S(int64_t, const int*, const int*, const T*, const T*, T, T*);         \
    void orc_hyb_apply##S(int, int, int, const int*, const T*, int64_t, const int*, const int*,    \
                          const T*, const T*, T*);                                                 \
    void orc_hyb_apply_add##S(int, int, int, const int*, const T*, int64_t, const int*,            \
                              const int*, const T*, const T*, T, T*);                              \
    void orc_csr_transpose##S(int, int, int64_t, const int*, const int*, const T*, int*, int*, T*); \
    int64_t orc_csr_matmult##S(int, int, const int*, const int*, const T*, const int*, const int*, \
                               const T*, int*, int*, T*);                                          \
    void orc_csr_matrix_add_subset##S(int, const int*, const int*, T*, const int*, const int*,     \
                                      const T*, T, T);                                             \
    int64_t orc_csr_matrix_add_union##S(int, const int*, const int*, const T*, const int*,         \
                                        const int*, const T*, T, T, int*, int*, T*);               \
    int  orc_csr_to_dia##S(int, int, int64_t, const int*, const int*, const T*, int*, T*);         \
    void orc_dia_apply##S(int, int, const int*, const T*, const T*, T*);                           \
    void orc_dia_apply_add##S(int, int, const int*, const T*, const T*, T, T*);                    \
    int64_t orc_dia_to_csr##S(int, int, int, const int*, const T*, int*, int*, T*);                \
    int  orc_csr_to_ell_fill##S(int, int64_t, const int*, const int*, const T*, int, int*, T*);    \
    int  orc_csr_to_hyb_fill##S(int, const int*, const int*, const T*, int, int*, T*, int*, int*,  \
                               T*);                                                                \
    void orc_csr_extract_diag##S(int, const int*, const int*, const T*, T*);                       \
    int  orc_csr_extract_inv_diag##S(int, const int*, const int*, const T*, T*);                   \
    void orc_add_scale##S(int64_t, T*, const T*, T);                                               \
    void orc_scale_add##S(int64_t, T*, T, const T*);                                               \
    void orc_scale_add_scale##S(int64_t, T*, T, const T*, T);                                      \
    void orc_scale_add2##S(int64_t, T*, T, const T*, T, const T*, T);                              \
    void orc_scale##S(int64_t, T*, T);                                                             \
    T    orc_dot##S(int64_t, const T*, const T*);                                                  \
    T    orc_norm##S(int64_t, const T*);                                                           \
    void orc_pointwise_mult##S(int64_t, T*, const T*);                                             \
    void orc_pointwise_mult2##S(int64_t, T*, const T*, const T*);                                  \
    void orc_copy_permute##S(int64_t, T*, const T*, const int*);                                   \
    void orc_copy_permute_backward##S(int64_t, T*, const T*, const int*);                          \
    int  orc_csr_ilu0##S(int, const int*, const int*, T*);                                         \
    int  orc_csr_ilup_numeric##S(int, int, const int*, const int*, const int*, const int*, const T*, T*, int*); \
    void orc_csr_lusolve##S(int, int64_t, const int*, const int*, const T*, const T*, T*);         \
    void orc_csr_lsolve##S(int, const int*, const int*, const T*, int, const T*, T*);              \
    void orc_csr_usolve##S(int, int64_t, const int*, const int*, const T*, int, const T*, T*);     \
    void orc_csr_itlusolve##S(int, double, int, int, int64_t, const int*, const int*, const T*,    \
                              const T*, T*, T*, T*);                                               \
    void orc_csr_itllsolve##S(int, double, int, int, int64_t, const int*, const int*, const T*,    \
                              const T*, T*, T*, T*);                                               \
    void orc_csr_itlsolve##S(int, double, int, int, int64_t, const int*, const int*, const T*,     \
                             int, const T*, T*, T*);                                               \
    void orc_csr_itusolve##S(int, double, int, int, int64_t, const int*, const int*, const T*,     \
                             int, const T*, T*, T*);                                               \
    void orc_csr_permute##S(int, int64_t, const int*, const int*, const T*, const int*, int*,      \
                            int*, T*);                                                             \
    int64_t orc_csr_extract_submatrix##S(const int*, const int*, const T*, int, int, int, int,     \
                                         int*, int*, T*);                                          \
    int orc_precond_apply##S(int, int, int64_t, const int*, const int*, const T*, const T*, T*);   \
    int orc_precond_apply_rep##S(int, int, int64_t, const int*, const int*, const T*, const T*,   \
                                 T*, int);                                                         \
    int orc_solve##S(int, int64_t, const int*, const int*, const T*, const T*, T*, orc_solve_cfg*);

ORC_DECL(double, _f64)
ORC_DECL(float, _f32)

int orc_solve_mixed(int nrow, int64_t nnz, const int* row_offset, const int* col,
                    const double* val, const double* rhs, double* x, orc_solve_cfg* outer,
                    orc_solve_cfg* inner, int* inner_iters_total);

#ifdef __cplusplus
}
#endif
#endif
