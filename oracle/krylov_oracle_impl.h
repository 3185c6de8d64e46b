/* ==========================================================================
 * krylov_oracle_impl.h  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Type-generic body of the CPU oracle.  Included twice by krylov_oracle.c,
 * once with T=double / SUF(x)=x##_f64 and once with T=float / SUF(x)=x##_f32.
 *
 * Every function is a plain-C restatement of the arithmetic of the reference
 * host/OpenMP backend (rocALUTION 3.2.0, /root/reference); the file:line each
 * function follows is cited above it (paths relative to /root/reference/).
 * The evaluation order of every floating point expression is the reference's
 * (left-to-right, no FMA contraction: build with -ffp-contract=off).
 * ========================================================================== */

#ifndef T
#error "define T and SUF before including"
#endif

/* ---- OpenMP thread policy ------------------------------------------------
 * src/base/backend_manager.cpp:590-607 (_set_omp_backend_threads): host kernels
 * run on ONE thread when size <= OpenMP_threshold (10000, :78), else on the
 * configured thread count. orc_threads_for() is defined in krylov_oracle.c. */

/* ---- SpMV ---------------------------------------------------------------- */

/* src/base/host/host_matrix_csr.cpp:702-735  HostMatrixCSR::Apply */
void SUF(orc_csr_apply)(int nrow, const int* row_offset, const int* col, const T* val,
                        const T* in, T* out)
{
    int nt = orc_threads_for(nrow);
#pragma omp parallel for num_threads(nt)
    for(int ai = 0; ai < nrow; ++ai)
    {
        T sum = (T)0;
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
        {
            sum += val[aj] * in[col[aj]];
        }
        out[ai] = sum;
    }
}

/* src/base/host/host_matrix_csr.cpp:737-769  HostMatrixCSR::ApplyAdd
 * (accumulates term by term straight into out[ai]; scalar*val first) */
void SUF(orc_csr_apply_add)(int nrow, int64_t nnz, const int* row_offset, const int* col,
                            const T* val, const T* in, T scalar, T* out)
{
    if(nnz <= 0)
        return;
    int nt = orc_threads_for(nrow);
#pragma omp parallel for num_threads(nt)
    for(int ai = 0; ai < nrow; ++ai)
    {
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
        {
            out[ai] += scalar * val[aj] * in[col[aj]];
        }
    }
}

/* src/base/host/host_matrix_ell.cpp:280-323  HostMatrixELL::Apply
 * ELL_IND(row,el,nrow,max_row) = el*nrow + row  (src/base/matrix_formats_ind.hpp:38-40)
 * stops at the first negative column; does nothing when nnz == 0 */
void SUF(orc_ell_apply)(int nrow, int max_row, const int* col, const T* val, const T* in, T* out)
{
    if((int64_t)nrow * max_row <= 0)
        return;
    int nt = orc_threads_for(nrow);
#pragma omp parallel for num_threads(nt)
    for(int ai = 0; ai < nrow; ++ai)
    {
        T sum = (T)0;
        for(int n = 0; n < max_row; ++n)
        {
            int aj     = n * nrow + ai;
            int col_aj = col[aj];
            if(col_aj >= 0)
                sum += val[aj] * in[col_aj];
            else
                break;
        }
        out[ai] = sum;
    }
}

/* src/base/host/host_matrix_ell.cpp:325-370  HostMatrixELL::ApplyAdd */
void SUF(orc_ell_apply_add)(int nrow, int max_row, const int* col, const T* val, const T* in,
                            T scalar, T* out)
{
    if((int64_t)nrow * max_row <= 0)
        return;
    int nt = orc_threads_for(nrow);
#pragma omp parallel for num_threads(nt)
    for(int ai = 0; ai < nrow; ++ai)
    {
        for(int n = 0; n < max_row; ++n)
        {
            int aj     = n * nrow + ai;
            int col_aj = col[aj];
            if(col_aj >= 0)
                out[ai] += scalar * val[aj] * in[col_aj];
            else
                break;
        }
    }
}

/* src/base/host/host_matrix_coo.cpp:354-378  HostMatrixCOO::Apply (serial) */
void SUF(orc_coo_apply)(int nrow, int64_t nnz, const int* row, const int* col, const T* val,
                        const T* in, T* out)
{
    for(int i = 0; i < nrow; ++i)
        out[i] = (T)0;
    for(int64_t i = 0; i < nnz; ++i)
        out[row[i]] += val[i] * in[col[i]];
}

/* src/base/host/host_matrix_coo.cpp:380-402  HostMatrixCOO::ApplyAdd (serial) */
void SUF(orc_coo_apply_add)(int64_t nnz, const int* row, const int* col, const T* val,
                            const T* in, T scalar, T* out)
{
    for(int64_t i = 0; i < nnz; ++i)
        out[row[i]] += scalar * val[i] * in[col[i]];
}

/* src/base/host/host_matrix_hyb.cpp:315-369  HostMatrixHYB::Apply
 * ELL part skips (does not stop at) invalid columns and accumulates into out[ai];
 * COO part is serial. Nothing happens when nnz == 0. */
void SUF(orc_hyb_apply)(int nrow, int ncol, int ell_max_row, const int* ell_col, const T* ell_val,
                        int64_t coo_nnz, const int* coo_row, const int* coo_col, const T* coo_val,
                        const T* in, T* out)
{
    int64_t ell_nnz = (int64_t)ell_max_row * nrow;
    if(ell_nnz + coo_nnz <= 0)
        return;
    if(ell_nnz > 0)
    {
        int nt = orc_threads_for(nrow);
#pragma omp parallel for num_threads(nt)
        for(int ai = 0; ai < nrow; ++ai)
        {
            out[ai] = (T)0;
            for(int n = 0; n < ell_max_row; ++n)
            {
                int aj = n * nrow + ai;
                if((ell_col[aj] >= 0) && (ell_col[aj] < ncol))
                    out[ai] += ell_val[aj] * in[ell_col[aj]];
            }
        }
    }
    for(int64_t i = 0; i < coo_nnz; ++i)
        out[coo_row[i]] += coo_val[i] * in[coo_col[i]];
}

/* src/base/host/host_matrix_hyb.cpp:371-420  HostMatrixHYB::ApplyAdd */
void SUF(orc_hyb_apply_add)(int nrow, int ncol, int ell_max_row, const int* ell_col,
                            const T* ell_val, int64_t coo_nnz, const int* coo_row,
                            const int* coo_col, const T* coo_val, const T* in, T scalar, T* out)
{
    int64_t ell_nnz = (int64_t)ell_max_row * nrow;
    if(ell_nnz + coo_nnz <= 0)
        return;
    if(ell_nnz > 0)
    {
        int nt = orc_threads_for(nrow);
#pragma omp parallel for num_threads(nt)
        for(int ai = 0; ai < nrow; ++ai)
        {
            for(int n = 0; n < ell_max_row; ++n)
            {
                int aj = n * nrow + ai;
                if((ell_col[aj] >= 0) && (ell_col[aj] < ncol))
                    out[ai] += scalar * ell_val[aj] * in[ell_col[aj]];
            }
        }
    }
    for(int64_t i = 0; i < coo_nnz; ++i)
        out[coo_row[i]] += scalar * coo_val[i] * in[coo_col[i]];
}

/* ---- format conversion (layout rules) ------------------------------------ */

/* src/base/host/host_conversion.cpp:621-687  csr_to_ell
 * width = max row nnz; REFUSES (returns 0) when width > 5*(nnz/nrow) (integer division);
 * padding col=-1, val=0; in-row order preserved. ell_col/ell_val must hold width*nrow. */
int SUF(orc_csr_to_ell_fill)(int nrow, int64_t nnz, const int* row_offset, const int* col,
                             const T* val, int max_row, int* ell_col, T* ell_val)
{
    (void)nnz;
    for(int i = 0; i < nrow; ++i)
    {
        int n = 0;
        for(int j = row_offset[i]; j < row_offset[i + 1]; ++j)
        {
            int64_t ind  = (int64_t)n * nrow + i;
            ell_val[ind] = val[j];
            ell_col[ind] = col[j];
            ++n;
        }
        for(int j = row_offset[i + 1] - row_offset[i]; j < max_row; ++j)
        {
            int64_t ind  = (int64_t)n * nrow + i;
            ell_val[ind] = (T)0;
            ell_col[ind] = -1;
            ++n;
        }
    }
    return 1;
}

/* src/base/host/host_conversion.cpp:1117-1239  csr_to_hyb
 * ELL width = (nnz-1)/nrow+1; entries beyond the width spill to COO in row order. */
int SUF(orc_csr_to_hyb_fill)(int nrow, const int* row_offset, const int* col, const T* val,
                             int ell_max_row, int* ell_col, T* ell_val, int* coo_row, int* coo_col,
                             T* coo_val)
{
    int64_t coo_idx = 0;
    for(int i = 0; i < nrow; ++i)
    {
        int p = 0;
        for(int j = row_offset[i]; j < row_offset[i + 1]; ++j)
        {
            if(p < ell_max_row)
            {
                int64_t idx  = (int64_t)(p++) * nrow + i;
                ell_col[idx] = col[j];
                ell_val[idx] = val[j];
            }
            else
            {
                coo_row[coo_idx] = i;
                coo_col[coo_idx] = col[j];
                coo_val[coo_idx] = val[j];
                ++coo_idx;
            }
        }
        for(int j = row_offset[i + 1] - row_offset[i]; j < ell_max_row; ++j)
        {
            int64_t idx  = (int64_t)(p++) * nrow + i;
            ell_col[idx] = -1;
            ell_val[idx] = (T)0;
        }
    }
    return 1;
}

/* ---- diagonal ------------------------------------------------------------- */

/* src/base/host/host_matrix_csr.cpp:772-797  ExtractDiagonal (rows w/o diagonal untouched) */
void SUF(orc_csr_extract_diag)(int nrow, const int* row_offset, const int* col, const T* val,
                               T* diag)
{
    for(int ai = 0; ai < nrow; ++ai)
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
            if(ai == col[aj])
            {
                diag[ai] = val[aj];
                break;
            }
}

/* src/base/host/host_matrix_csr.cpp:800-845  ExtractInverseDiagonal
 * d = 1/a_ii; a_ii == 0 -> 1; rows without a stored diagonal are left unwritten
 * (the caller's Allocate zero-filled them, src/base/local_matrix.cpp:2294-2299).
 * returns 1 if a zero diagonal was detected. */
int SUF(orc_csr_extract_inv_diag)(int nrow, const int* row_offset, const int* col, const T* val,
                                  T* inv_diag)
{
    int detect_zero_diag = 0;
    for(int ai = 0; ai < nrow; ++ai)
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
            if(ai == col[aj])
            {
                if(val[aj] != (T)0)
                    inv_diag[ai] = (T)1 / val[aj];
                else
                {
                    inv_diag[ai]     = (T)1;
                    detect_zero_diag = 1;
                }
                break;
            }
    return detect_zero_diag;
}

/* ---- BLAS-1 (src/base/host/host_vector.cpp) ------------------------------- */

/* :635-651 AddScale   this = this + alpha*x */
void SUF(orc_add_scale)(int64_t n, T* v, const T* x, T alpha)
{
    int nt = orc_threads_for(n);
#pragma omp parallel for num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        v[i] = v[i] + alpha * x[i];
}

/* :654-670 ScaleAdd   this = alpha*this + x */
void SUF(orc_scale_add)(int64_t n, T* v, T alpha, const T* x)
{
    int nt = orc_threads_for(n);
#pragma omp parallel for num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        v[i] = alpha * v[i] + x[i];
}

/* :672-690 ScaleAddScale   this = alpha*this + beta*x */
void SUF(orc_scale_add_scale)(int64_t n, T* v, T alpha, const T* x, T beta)
{
    int nt = orc_threads_for(n);
#pragma omp parallel for num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        v[i] = alpha * v[i] + beta * x[i];
}

/* :723-747 ScaleAdd2   this = alpha*this + beta*x + gamma*y */
void SUF(orc_scale_add2)(int64_t n, T* v, T alpha, const T* x, T beta, const T* y, T gamma)
{
    int nt = orc_threads_for(n);
#pragma omp parallel for num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        v[i] = alpha * v[i] + beta * x[i] + gamma * y[i];
}

/* :750-760 Scale */
void SUF(orc_scale)(int64_t n, T* v, T alpha)
{
    int nt = orc_threads_for(n);
#pragma omp parallel for num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        v[i] *= alpha;
}

/* :771-791 Dot, :852-872 DotNonConj (identical for real types): OpenMP reduction(+) */
T SUF(orc_dot)(int64_t n, const T* a, const T* b)
{
    T   dot = (T)0;
    int nt  = orc_threads_for(n);
#pragma omp parallel for reduction(+ : dot) num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        dot += a[i] * b[i];
    return dot;
}

/* :1019-1035 Norm = sqrt(sum x^2), OpenMP reduction(+) */
T SUF(orc_norm)(int64_t n, const T* a)
{
    T   norm2 = (T)0;
    int nt    = orc_threads_for(n);
#pragma omp parallel for reduction(+ : norm2) num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        norm2 += a[i] * a[i];
    return (T)ORC_SQRT(norm2);
}

/* :1231-1255 PointWiseMult(x): this = this*x ;  :1257-1279 PointWiseMult(x,y): this = y*x */
void SUF(orc_pointwise_mult)(int64_t n, T* v, const T* x)
{
    int nt = orc_threads_for(n);
#pragma omp parallel for num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        v[i] = v[i] * x[i];
}
void SUF(orc_pointwise_mult2)(int64_t n, T* v, const T* x, const T* y)
{
    int nt = orc_threads_for(n);
#pragma omp parallel for num_threads(nt)
    for(int64_t i = 0; i < n; ++i)
        v[i] = y[i] * x[i];
}

/* :1365-1388 CopyFromPermute: this[perm[i]] = src[i] */
void SUF(orc_copy_permute)(int64_t n, T* dst, const T* src, const int* perm)
{
    for(int64_t i = 0; i < n; ++i)
        dst[perm[i]] = src[i];
}
/* :1390-1412 CopyFromPermuteBackward: this[i] = src[perm[i]] */
void SUF(orc_copy_permute_backward)(int64_t n, T* dst, const T* src, const int* perm)
{
    for(int64_t i = 0; i < n; ++i)
        dst[i] = src[perm[i]];
}

/* ---- ILU(0) and triangular solves ----------------------------------------- */

/* src/base/host/host_matrix_csr.cpp:2096-2171  ILU0Factorize (serial IKJ, sorted CSR).
 * Keeps the reference's "0 == absent" scatter-map quirk (:2142). */
int SUF(orc_csr_ilu0)(int nrow, const int* row_offset, const int* col, T* val)
{
    int* diag_offset = (int*)calloc((size_t)nrow, sizeof(int));
    int* nnz_entries = (int*)calloc((size_t)nrow, sizeof(int));
    if(!diag_offset || !nnz_entries)
    {
        free(diag_offset);
        free(nnz_entries);
        return 0;
    }
    for(int ai = 0; ai < nrow; ++ai)
    {
        int row_start = row_offset[ai];
        int row_end   = row_offset[ai + 1];
        int j;
        for(j = row_start; j < row_end; ++j)
            nnz_entries[col[j]] = j;
        for(j = row_start; j < row_end; ++j)
        {
            if(col[j] < ai)
            {
                int col_j  = col[j];
                int diag_j = diag_offset[col_j];
                if(val[diag_j] != (T)0)
                {
                    val[j] = val[j] / val[diag_j];
                    for(int k = diag_j + 1; k < row_offset[col_j + 1]; ++k)
                    {
                        if(nnz_entries[col[k]] != 0)
                            val[nnz_entries[col[k]]] -= val[j] * val[k];
                    }
                }
            }
            else
                break;
        }
        diag_offset[ai] = j;
        for(j = row_start; j < row_end; ++j)
            nnz_entries[col[j]] = 0;
    }
    free(diag_offset);
    free(nnz_entries);
    return 1;
}

/* src/base/host/host_matrix_csr.cpp:3149-3312  ILUpFactorizeNumeric(p, structure): the level-controlled IKJ sweep on
 * the pattern S (= pattern of A^(p+1), local_matrix.cpp:3929-3935).  val/lev are S-sized work arrays: on return the
 * entries with lev <= p are the factor's entries (the caller compacts them, host :3276-3300). */
int SUF(orc_csr_ilup_numeric)(int nrow, int p, const int* s_row_offset, const int* s_col, const int* a_row_offset,
                              const int* a_col, const T* a_val, T* val, int* levels)
{
    const int inf_level = 99999; /* :3171 */
    int*      ind_diag  = (int*)calloc((size_t)nrow, sizeof(int));
    if(!ind_diag)
        return 0;
    for(int ai = 0; ai < nrow; ++ai) /* :3178-3190 */
        for(int aj = s_row_offset[ai]; aj < s_row_offset[ai + 1]; ++aj)
            if(ai == s_col[aj])
            {
                ind_diag[ai] = aj;
                break;
            }
    for(int ai = 0; ai < nrow; ++ai) /* :3199-3226: A's values at level 0, the rest 0 at "infinity" */
        for(int aj = s_row_offset[ai]; aj < s_row_offset[ai + 1]; ++aj)
        {
            levels[aj] = inf_level;
            val[aj]    = (T)0;
            for(int ajj = a_row_offset[ai]; ajj < a_row_offset[ai + 1]; ++ajj)
                if(s_col[aj] == a_col[ajj])
                {
                    val[aj]    = a_val[ajj];
                    levels[aj] = 0;
                    break;
                }
        }
    for(int ai = 1; ai < nrow; ++ai) /* :3229-3273 */
    {
        for(int ak = s_row_offset[ai]; ai > s_col[ak]; ++ak)
        {
            if(levels[ak] <= p)
            {
                val[ak] /= val[ind_diag[s_col[ak]]];
                for(int aj = ak + 1; aj < s_row_offset[ai + 1]; ++aj)
                {
                    T   val_kj   = (T)0;
                    int level_kj = inf_level;
                    for(int kj = s_row_offset[s_col[ak]]; kj < s_row_offset[s_col[ak] + 1]; ++kj)
                        if(s_col[aj] == s_col[kj])
                        {
                            level_kj = levels[kj];
                            val_kj   = val[kj];
                            break;
                        }
                    int lev = level_kj + levels[ak] + 1;
                    if(levels[aj] > lev)
                        levels[aj] = lev;
                    val[aj] -= val[ak] * val_kj;
                }
            }
        }
        for(int ak = s_row_offset[ai]; ak < s_row_offset[ai + 1]; ++ak)
            if(levels[ak] > p)
            {
                levels[ak] = inf_level;
                val[ak]    = (T)0;
            }
    }
    free(ind_diag);
    return 1;
}

/* src/base/host/host_matrix_csr.cpp:1163-1221  LUSolve (serial): unit-lower forward
 * sweep that stops at the first col >= row, then backward sweep dividing by the diagonal
 * found by equality scan (falls back to the previously found position). */
void SUF(orc_csr_lusolve)(int nrow, int64_t nnz, const int* row_offset, const int* col,
                          const T* val, const T* in, T* out)
{
    for(int ai = 0; ai < nrow; ++ai)
    {
        out[ai] = in[ai];
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
        {
            if(col[aj] < ai)
                out[ai] -= val[aj] * out[col[aj]];
            else
                break;
        }
    }
    int64_t diag_aj = nnz - 1;
    for(int ai = nrow - 1; ai >= 0; --ai)
    {
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
        {
            if(col[aj] > ai)
                out[ai] -= val[aj] * out[col[aj]];
            if(col[aj] == ai)
                diag_aj = aj;
        }
        out[ai] /= val[diag_aj];
    }
}

/* src/base/host/host_matrix_csr.cpp:1357-1404  LSolve with L_diag_unit_ flag */
void SUF(orc_csr_lsolve)(int nrow, const int* row_offset, const int* col, const T* val,
                         int diag_unit, const T* in, T* out)
{
    int diag_aj = 0;
    for(int ai = 0; ai < nrow; ++ai)
    {
        out[ai] = in[ai];
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
        {
            if(col[aj] < ai)
                out[ai] -= val[aj] * out[col[aj]];
            else
            {
                if(!diag_unit)
                    diag_aj = aj;
                break;
            }
        }
        if(!diag_unit)
            out[ai] /= val[diag_aj];
    }
}

/* src/base/host/host_matrix_csr.cpp:1420-1466  USolve with U_diag_unit_ flag */
void SUF(orc_csr_usolve)(int nrow, int64_t nnz, const int* row_offset, const int* col,
                         const T* val, int diag_unit, const T* in, T* out)
{
    int64_t diag_aj = nnz - 1;
    for(int ai = nrow - 1; ai >= 0; --ai)
    {
        out[ai] = in[ai];
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
        {
            if(col[aj] > ai)
                out[ai] -= val[aj] * out[col[aj]];
            if(!diag_unit)
                if(col[aj] == ai)
                    diag_aj = aj;
        }
        if(!diag_unit)
            out[ai] /= val[diag_aj];
    }
}

/* src/base/host/host_matrix_csr.cpp:2344-2466  ICFactorize on L = lower part incl. diagonal (sorted rows,
 * diagonal last in every row): in place, inverse diagonal returned.  Returns 0 on the reference's breakdowns
 * (structural / numerical zero diagonal: FATAL_ERROR there). */
int SUF(orc_csr_ic_factorize)(int nrow, const int* row_offset, const int* col, T* val, T* inv_diag)
{
    int* diag_offset = (int*)calloc((size_t)nrow, sizeof(int));
    int* nnz_entries = (int*)calloc((size_t)nrow, sizeof(int));
    int  ok          = 1;
    for(int i = 0; i < nrow && ok; ++i)
    {
        int row_begin = row_offset[i], row_end = row_offset[i + 1];
        for(int j = row_begin; j < row_end; ++j)
            nnz_entries[col[j]] = j;
        T   sum      = (T)0;
        int has_diag = 0;
        int j;
        for(j = row_begin; j < row_end; ++j)
        {
            int col_j = col[j];
            T   val_j = val[j];
            if(col_j == i)
            {
                has_diag = 1;
                break;
            }
            if(col_j > i)
                break;
            int row_begin_j = row_offset[col_j];
            int row_diag_j  = diag_offset[col_j];
            T   local_sum   = (T)0;
            T   inv_d       = val[row_diag_j];
            if(inv_d == (T)0)
            {
                ok = 0;
                break;
            }
            inv_d = (T)1 / inv_d;
            for(int k = row_begin_j; k < row_diag_j; ++k)
            {
                int col_k = col[k];
                if(nnz_entries[col_k] != 0)
                    local_sum += val[k] * val[nnz_entries[col_k]];
            }
            val_j = (val_j - local_sum) * inv_d;
            sum += val_j * val_j;
            val[j] = val_j;
        }
        if(!ok || !has_diag)
        {
            ok = 0;
            break;
        }
        T diag_entry = ORC_SQRT(ORC_FABS(val[j] - sum));
        val[j]       = diag_entry;
        if(diag_entry == (T)0)
        {
            ok = 0;
            break;
        }
        inv_diag[i]    = (T)1 / diag_entry;
        diag_offset[i] = j;
        for(j = row_begin; j < row_end; ++j)
            nnz_entries[col[j]] = 0;
    }
    free(diag_offset);
    free(nnz_entries);
    return ok;
}

/* host_matrix_csr.cpp:1294-1341  LLSolve(in, inv_diag, out): forward sweep by rows, backward sweep by COLUMNS
 * (scatter), both scaled with the stored inverse diagonal */
void SUF(orc_csr_llsolve)(int nrow, const int* row_offset, const int* col, const T* val, const T* in,
                          const T* inv_diag, T* out)
{
    for(int ai = 0; ai < nrow; ++ai)
    {
        T   value    = in[ai];
        int diag_idx = row_offset[ai + 1] - 1;
        for(int aj = row_offset[ai]; aj < diag_idx; ++aj)
            value -= val[aj] * out[col[aj]];
        out[ai] = value * inv_diag[ai];
    }
    for(int ai = nrow - 1; ai >= 0; --ai)
    {
        int diag_idx = row_offset[ai + 1] - 1;
        T   value    = out[ai] * inv_diag[ai];
        for(int aj = row_offset[ai]; aj < diag_idx; ++aj)
            out[col[aj]] -= value * val[aj];
        out[ai] = value;
    }
}

/* ---- CSR matrix algebra ---------------------------------------------------------------------
 * host_matrix_csr.cpp:3757-3806 Transpose: counting sort by column, entries of a transposed row in ascending
 * original-row order (out arrays sized by the caller: ncol+1, nnz, nnz) */
void SUF(orc_csr_transpose)(int nrow, int ncol, int64_t nnz, const int* rp, const int* col, const T* val, int* trp,
                            int* tcol, T* tval)
{
    for(int i = 0; i <= ncol; ++i)
        trp[i] = 0;
    for(int64_t k = 0; k < nnz; ++k)
        trp[col[k] + 1] += 1;
    for(int i = 0; i < ncol; ++i)
        trp[i + 1] += trp[i];
    int* cur = (int*)malloc(sizeof(int) * (size_t)(ncol > 0 ? ncol : 1));
    for(int i = 0; i < ncol; ++i)
        cur[i] = trp[i];
    for(int ai = 0; ai < nrow; ++ai)
        for(int aj = rp[ai]; aj < rp[ai + 1]; ++aj)
        {
            int p   = cur[col[aj]]++;
            tcol[p] = ai;
            tval[p] = val[aj];
        }
    free(cur);
}

/* :2805-2938 MatMatMult C = A*B: Gustavson with a column marker, products added in (ja, jb) order; then Sort
 * (:3812-3846, stable by column).  Two calls: ccol == NULL counts (crp filled), else fills. */
int64_t SUF(orc_csr_matmult)(int n, int m, const int* arp, const int* acol, const T* aval, const int* brp,
                             const int* bcol, const T* bval, int* crp, int* ccol, T* cval)
{
    int* marker = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    for(int i = 0; i < m; ++i)
        marker[i] = -1;
    if(!ccol)
    {
        crp[0] = 0;
        for(int ia = 0; ia < n; ++ia)
        {
            int c = 0;
            for(int ja = arp[ia]; ja < arp[ia + 1]; ++ja)
                for(int jb = brp[acol[ja]]; jb < brp[acol[ja] + 1]; ++jb)
                    if(marker[bcol[jb]] != ia)
                    {
                        marker[bcol[jb]] = ia;
                        ++c;
                    }
            crp[ia + 1] = crp[ia] + c;
        }
        free(marker);
        return crp[n];
    }
    for(int ia = 0; ia < n; ++ia)
    {
        int row_begin = crp[ia], row_end = row_begin;
        for(int ja = arp[ia]; ja < arp[ia + 1]; ++ja)
        {
            int ca = acol[ja];
            T   va = aval[ja];
            for(int jb = brp[ca]; jb < brp[ca + 1]; ++jb)
            {
                int cb = bcol[jb];
                T   vb = bval[jb];
                if(marker[cb] < row_begin)
                {
                    marker[cb]    = row_end;
                    ccol[row_end] = cb;
                    cval[row_end] = va * vb;
                    ++row_end;
                }
                else
                    cval[marker[cb]] += va * vb;
            }
        }
        /* Sort(): bubble sort of the row by column */
        for(int j = crp[ia]; j < crp[ia + 1]; ++j)
            for(int jj = crp[ia]; jj < crp[ia + 1] - 1; ++jj)
                if(ccol[jj] > ccol[jj + 1])
                {
                    int ti        = ccol[jj];
                    T   tv        = cval[jj];
                    ccol[jj]      = ccol[jj + 1];
                    cval[jj]      = cval[jj + 1];
                    ccol[jj + 1]  = ti;
                    cval[jj + 1]  = tv;
                }
    }
    free(marker);
    return crp[n];
}

/* :3324-3365 MatrixAdd, structure == false: this = alpha*this + beta*mat on the entries of mat (subset pattern) */
void SUF(orc_csr_matrix_add_subset)(int nrow, const int* rp, const int* col, T* val, const int* brp, const int* bcol,
                                    const T* bval, T alpha, T beta)
{
    for(int ai = 0; ai < nrow; ++ai)
    {
        int first_col = brp[ai];
        for(int ajj = rp[ai]; ajj < rp[ai + 1]; ++ajj)
            for(int aj = first_col; aj < brp[ai + 1]; ++aj)
                if(bcol[aj] == col[ajj])
                {
                    val[ajj] = alpha * val[ajj] + beta * bval[aj];
                    ++first_col;
                    break;
                }
    }
}
/* :3366-3459 structure == true: union pattern (sorted, unique), zero-filled, += alpha*A then += beta*B.
 * ccol == NULL: count only (crp filled). */
int64_t SUF(orc_csr_matrix_add_union)(int nrow, const int* arp, const int* acol, const T* aval, const int* brp,
                                      const int* bcol, const T* bval, T alpha, T beta, int* crp, int* ccol, T* cval)
{
    if(!ccol)
        crp[0] = 0;
    for(int i = 0; i < nrow; ++i)
    {
        int na = arp[i + 1] - arp[i], nb = brp[i + 1] - brp[i];
        int* u = (int*)malloc(sizeof(int) * (size_t)(na + nb + 1));
        int  k = 0;
        for(int j = arp[i]; j < arp[i + 1]; ++j)
            u[k++] = acol[j];
        for(int j = brp[i]; j < brp[i + 1]; ++j)
            u[k++] = bcol[j];
        for(int a = 1; a < k; ++a) /* std::sort */
        {
            int c = u[a], q = a - 1;
            for(; q >= 0 && u[q] > c; --q)
                u[q + 1] = u[q];
            u[q + 1] = c;
        }
        int w = 0;
        for(int a = 0; a < k; ++a) /* std::unique */
            if(a == 0 || u[a] != u[a - 1])
                u[w++] = u[a];
        if(!ccol)
            crp[i + 1] = crp[i] + w;
        else
        {
            int Aj = arp[i], Bj = brp[i];
            for(int j = 0; j < w; ++j)
            {
                int o   = crp[i] + j;
                ccol[o] = u[j];
                cval[o] = (T)0;
                for(int jj = Aj; jj < arp[i + 1]; ++jj)
                    if(ccol[o] == acol[jj])
                    {
                        cval[o] += alpha * aval[jj];
                        ++Aj;
                        break;
                    }
                for(int jj = Bj; jj < brp[i + 1]; ++jj)
                    if(ccol[o] == bcol[jj])
                    {
                        cval[o] += beta * bval[jj];
                        ++Bj;
                        break;
                    }
            }
        }
        free(u);
    }
    return crp[nrow];
}

/* ---- DIA ------------------------------------------------------------------------------------
 * host_conversion.cpp:958-1038 csr_to_dia: one slot per populated diagonal (col - row), offsets ascending, values
 * column-major d*nrow + row (matrix_formats_ind.hpp:43-45), absent entries zero; refused (-1) when the number of
 * diagonals exceeds 5 * (nnz / min(nrow, ncol)).  offset / val may be NULL (count only). */
int SUF(orc_csr_to_dia)(int nrow, int ncol, int64_t nnz, const int* row_offset, const int* col, const T* val,
                        int* offset, T* dval)
{
    int* slot = (int*)calloc((size_t)nrow + ncol + 1, sizeof(int));
    int  nd   = 0;
    for(int i = 0; i < nrow; ++i)
        for(int j = row_offset[i]; j < row_offset[i + 1]; ++j)
        {
            int o = col[j] - i + nrow;
            if(!slot[o])
            {
                slot[o] = 1;
                ++nd;
            }
        }
    int size = nrow < ncol ? nrow : ncol;
    if(size <= 0 || nd > 5 * (nnz / size))
    {
        free(slot);
        return -1;
    }
    if(offset && dval)
    {
        for(int64_t k = 0; k < (int64_t)size * nd; ++k)
            dval[k] = (T)0;
        for(int i = 0, d = 0; i < nrow + ncol; ++i)
            if(slot[i])
            {
                slot[i]     = d;
                offset[d++] = i - nrow;
            }
        for(int i = 0; i < nrow; ++i)
            for(int j = row_offset[i]; j < row_offset[i + 1]; ++j)
                dval[(int64_t)slot[col[j] - i + nrow] * nrow + i] = val[j];
    }
    free(slot);
    return nd;
}

/* host_matrix_dia.cpp:300-356 Apply / :359-412 ApplyAdd: diagonal j covers rows [max(0,-off), nrow - max(0,off));
 * a row past the end of a diagonal leaves the loop (offsets ascend, so nothing is lost) */
void SUF(orc_dia_apply)(int nrow, int ndiag, const int* offset, const T* dval, const T* in, T* out)
{
    for(int i = 0; i < nrow; ++i)
    {
        T sum = (T)0;
        for(int j = 0; j < ndiag; ++j)
        {
            int off = offset[j], start = 0, end = nrow;
            if(off < 0)
                start = -off;
            else
                end = nrow - off;
            if(i >= start && i < end)
                sum += dval[(int64_t)j * nrow + i] * in[i + off];
            else if(i >= end)
                break;
        }
        out[i] = sum;
    }
}
void SUF(orc_dia_apply_add)(int nrow, int ndiag, const int* offset, const T* dval, const T* in, T scalar, T* out)
{
    for(int i = 0; i < nrow; ++i)
        for(int j = 0; j < ndiag; ++j)
        {
            int off = offset[j], start = 0, end = nrow;
            if(off < 0)
                start = -off;
            else
                end = nrow - off;
            if(i >= start && i < end)
                out[i] += scalar * dval[(int64_t)j * nrow + i] * in[i + off];
            else if(i >= end)
                break;
        }
}
/* host_conversion.cpp:1040-1113 dia_to_csr: valid columns with a non-zero value, diagonal order; returns nnz
 * (col / val may be NULL: row offsets only) */
int64_t SUF(orc_dia_to_csr)(int nrow, int ncol, int ndiag, const int* offset, const T* dval, int* row_offset,
                            int* col, T* val)
{
    row_offset[0] = 0;
    for(int i = 0; i < nrow; ++i)
    {
        int idx = row_offset[i];
        for(int n = 0; n < ndiag; ++n)
        {
            int j = i + offset[n];
            if(j >= 0 && j < ncol && dval[(int64_t)n * nrow + i] != (T)0)
            {
                if(col && val)
                {
                    col[idx] = j;
                    val[idx] = dval[(int64_t)n * nrow + i];
                }
                ++idx;
            }
        }
        row_offset[i + 1] = idx;
    }
    return row_offset[nrow];
}

/* ---- iterative triangular solves (TriSolverAlg_Iterative) --------------------------------
 * src/base/host/host_sparse.cpp:195-530 host_csritsv_solve, matrix_type general, alpha = 1:
 * Jacobi sweeps  y <- y + D^-1 (x - (D + T) y)  on one triangle of a CSR matrix with sorted rows, started from
 * whatever y holds, stopped after *nmaxiter sweeps or when the sweep's max-norm figure drops to tol (tol == NULL:
 * never); *nmaxiter is overwritten with the sweeps done when the tolerance stops it.
 *   lower: entries [row begin, first col >= i)            (+ the diagonal, last, when non-unit)
 *   upper: entries (diagonal, row end)                    (+ the diagonal, FIRST, when non-unit)
 *   transposed (lower, non-unit only here): column sums of the lower triangle incl. diagonal, rows ascending
 * A missing or zero diagonal (non-unit) leaves y untouched (:388-391, :425-429).  work: 2*m values. */
static void SUF(orc_csritsv)(int* nmaxiter, const T* tol, int transposed, int m, int64_t nnz, int lower,
                             int unit, const T* val, const int* rp, const int* col, const T* x, T* y, T* work)
{
    if(m == 0 || nnz == 0)
    {
        if(nnz == 0 && unit)
        {
            for(int i = 0; i < m; ++i)
                y[i] = (T)1 * x[i];
            *nmaxiter = 1;
        }
        return;
    }
    T*   yp  = work;
    T*   idg = work + m;
    int* b   = (int*)malloc(sizeof(int) * (size_t)m);
    int* e   = (int*)malloc(sizeof(int) * (size_t)m);
    int  bad = 0;
    for(int i = 0; i < m && !bad; ++i)
    {
        int k = rp[i], end = rp[i + 1];
        if(unit)
        {
            /* first entry with col >= i (lower) resp. col > i (upper) */
            while(k < end && (lower ? col[k] < i : col[k] <= i))
                ++k;
            b[i] = lower ? rp[i] : k;
            e[i] = lower ? k : end;
        }
        else
        {
            while(k < end && col[k] != i)
                ++k;
            if(k == end)
                bad = 1; /* structural zero pivot */
            else if(val[k] == (T)0)
                bad = 1; /* numerical zero pivot */
            else
                idg[i] = (T)1 / val[k];
            b[i] = lower ? rp[i] : k;
            e[i] = lower ? k + 1 : end;
        }
    }
    if(!bad)
        for(int iter = 0; iter < *nmaxiter; ++iter)
        {
            for(int i = 0; i < m; ++i)
                yp[i] = y[i];
            T mx = (T)0, mxr = (T)0;
            if(!transposed && !unit)
                for(int i = 0; i < m; ++i)
                {
                    if(e[i] > b[i] + 1)
                    {
                        T sum = (T)0;
                        for(int k = b[i]; k < e[i]; ++k)
                            sum += val[k] * yp[col[k]];
                        T r = (T)1 * x[i] - sum;
                        T h = idg[i] * r;
                        if(mx < ORC_FABS(h))
                            mx = ORC_FABS(h);
                        if(mxr < ORC_FABS(r))
                            mxr = ORC_FABS(r);
                        y[i] = yp[i] + h;
                    }
                    else
                    {
                        y[i] = idg[i] * (T)1 * x[i];
                        T d  = ORC_FABS(y[i] - yp[i]);
                        if(mx < d)
                            mx = d;
                        T r = ORC_FABS((T)1 * x[i] - yp[i] / idg[i]);
                        if(mxr < r)
                            mxr = r;
                    }
                }
            else if(!transposed)
                for(int i = 0; i < m; ++i)
                {
                    T sum = (T)0;
                    for(int k = b[i]; k < e[i]; ++k)
                        sum += val[k] * yp[col[k]];
                    y[i] = (T)1 * x[i] - sum;
                    T h  = ORC_FABS(y[i] - yp[i]);
                    if(mx < h)
                        mx = h;
                    mxr = mx;
                }
            else
            {
                for(int i = 0; i < m; ++i)
                    y[i] = (T)0;
                for(int i = 0; i < m; ++i)
                    for(int k = b[i]; k < e[i]; ++k)
                        y[col[k]] += val[k] * yp[i];
                for(int i = 0; i < m; ++i)
                {
                    /* as written in the reference (:487): the residual figure is max(mx so far, this row) */
                    T r = ORC_FABS((T)1 * x[i] - y[i]);
                    mxr = (mx < r) ? r : mx;
                    T h = idg[i] * ((T)1 * x[i] - y[i]);
                    if(mx < ORC_FABS(h))
                        mx = ORC_FABS(h);
                    y[i] = h + yp[i];
                }
            }
            if(tol && mxr <= *tol)
            {
                *nmaxiter = iter + 1;
                break;
            }
        }
    free(b);
    free(e);
}

/* host_matrix_csr.cpp:1565-1650 ItLUSolve: L (unit) into the persistent tmp vector, then U (non-unit) into out;
 * ONE max_iter variable serves both calls, so a tolerance stop of the L stage caps the U stage */
void SUF(orc_csr_itlusolve)(int max_iter, double tolerance, int use_tol, int nrow, int64_t nnz, const int* rp,
                            const int* col, const T* val, const T* in, T* tmp, T* out, T* work)
{
    if(nnz <= 0)
        return;
    T        t   = (T)tolerance;
    const T* tol = use_tol ? &t : NULL;
    SUF(orc_csritsv)(&max_iter, tol, 0, nrow, nnz, 1, 1, val, rp, col, in, tmp, work);
    SUF(orc_csritsv)(&max_iter, tol, 0, nrow, nnz, 0, 0, val, rp, col, tmp, out, work);
}
/* :1748-1833 ItLLSolve: L (non-unit), then L^T */
void SUF(orc_csr_itllsolve)(int max_iter, double tolerance, int use_tol, int nrow, int64_t nnz, const int* rp,
                            const int* col, const T* val, const T* in, T* tmp, T* out, T* work)
{
    if(nnz <= 0)
        return;
    T        t   = (T)tolerance;
    const T* tol = use_tol ? &t : NULL;
    SUF(orc_csritsv)(&max_iter, tol, 0, nrow, nnz, 1, 0, val, rp, col, in, tmp, work);
    SUF(orc_csritsv)(&max_iter, tol, 1, nrow, nnz, 1, 0, val, rp, col, tmp, out, work);
}
/* :1909-1968 ItLSolve / :2033-2092 ItUSolve */
void SUF(orc_csr_itlsolve)(int max_iter, double tolerance, int use_tol, int nrow, int64_t nnz, const int* rp,
                           const int* col, const T* val, int diag_unit, const T* in, T* out, T* work)
{
    if(nnz <= 0)
        return;
    T        t   = (T)tolerance;
    const T* tol = use_tol ? &t : NULL;
    SUF(orc_csritsv)(&max_iter, tol, 0, nrow, nnz, 1, diag_unit, val, rp, col, in, out, work);
}
void SUF(orc_csr_itusolve)(int max_iter, double tolerance, int use_tol, int nrow, int64_t nnz, const int* rp,
                           const int* col, const T* val, int diag_unit, const T* in, T* out, T* work)
{
    if(nnz <= 0)
        return;
    T        t   = (T)tolerance;
    const T* tol = use_tol ? &t : NULL;
    SUF(orc_csritsv)(&max_iter, tol, 0, nrow, nnz, 0, diag_unit, val, rp, col, in, out, work);
}

/* ---- permutation / extraction --------------------------------------------- */

/* src/base/host/host_matrix_csr.cpp:3848-3958  Permute:  B = P A P^T, rows moved to
 * perm[i], columns relabelled perm[col] and insertion-sorted inside each row. */
void SUF(orc_csr_permute)(int nrow, int64_t nnz, const int* row_offset, const int* col,
                          const T* val, const int* perm, int* out_row_offset, int* out_col,
                          T* out_val)
{
    if(nnz <= 0)
    {
        for(int i = 0; i <= nrow; ++i)
            out_row_offset[i] = 0;
        return;
    }
    int* perm_row_nnz = (int*)malloc(sizeof(int) * (size_t)nrow);
    int* tcol         = (int*)malloc(sizeof(int) * (size_t)nnz);
    T*   tval         = (T*)malloc(sizeof(T) * (size_t)nnz);
    for(int i = 0; i < nrow; ++i)
        perm_row_nnz[perm[i]] = row_offset[i + 1] - row_offset[i];
    int sum = 0;
    for(int i = 0; i < nrow; ++i)
    {
        out_row_offset[i] = sum;
        sum += perm_row_nnz[i];
    }
    out_row_offset[nrow] = sum;
    for(int i = 0; i < nrow; ++i)
    {
        int permIndex = out_row_offset[perm[i]];
        int prevIndex = row_offset[i];
        int rn        = row_offset[i + 1] - row_offset[i];
        for(int j = 0; j < rn; ++j)
        {
            tcol[permIndex + j] = col[prevIndex + j];
            tval[permIndex + j] = val[prevIndex + j];
        }
    }
    for(int i = 0; i < nrow; ++i)
    {
        int row_index = out_row_offset[i];
        for(int j = 0; j < perm_row_nnz[i]; ++j)
        {
            int k    = j - 1;
            int comp = perm[tcol[row_index + j]];
            for(; k >= 0; --k)
            {
                if(out_col[row_index + k] > comp)
                {
                    out_val[row_index + k + 1] = out_val[row_index + k];
                    out_col[row_index + k + 1] = out_col[row_index + k];
                }
                else
                    break;
            }
            out_val[row_index + k + 1] = tval[row_index + j];
            out_col[row_index + k + 1] = comp;
        }
    }
    free(perm_row_nnz);
    free(tcol);
    free(tval);
}

/* src/base/host/host_matrix_csr.cpp:848-916  ExtractSubMatrix: count pass (out_col==NULL)
 * or fill pass; columns shifted by col_offset, in-row order preserved. returns nnz. */
int64_t SUF(orc_csr_extract_submatrix)(const int* row_offset, const int* col, const T* val,
                                       int r0, int c0, int rsize, int csize, int* out_row_offset,
                                       int* out_col, T* out_val)
{
    int64_t m = 0;
    if(out_row_offset)
        out_row_offset[0] = 0;
    for(int ai = r0; ai < r0 + rsize; ++ai)
    {
        for(int aj = row_offset[ai]; aj < row_offset[ai + 1]; ++aj)
        {
            if((col[aj] >= c0) && (col[aj] < c0 + csize))
            {
                if(out_col)
                {
                    out_col[m] = col[aj] - c0;
                    out_val[m] = val[aj];
                }
                ++m;
            }
        }
        if(out_row_offset)
            out_row_offset[ai - r0 + 1] = (int)m;
    }
    return m;
}

/* ---- operator + preconditioner objects used by the solver restatements ---- */

typedef struct
{
    int        format; /* ORC_CSR / ORC_ELL / ORC_HYB */
    int        nrow;
    int64_t    nnz;
    const int* row_offset;
    const int* col;
    const T*   val;
    /* ELL / HYB storage built by the oracle's own conversion */
    int  ell_max_row;
    int* ell_col;
    T*   ell_val;
    int64_t coo_nnz;
    int *   coo_row, *coo_col;
    T*      coo_val;
    int     dia_ndiag;
    int*    dia_offset;
    T*      dia_val;
} SUF(orc_op);

static void SUF(op_apply)(const SUF(orc_op) * A, const T* in, T* out)
{
    /* src/base/local_matrix.cpp:2154-2182: nnz == 0 -> zero fill */
    if(A->nnz <= 0)
    {
        for(int i = 0; i < A->nrow; ++i)
            out[i] = (T)0;
        return;
    }
    if(A->format == ORC_DIA)
        SUF(orc_dia_apply)(A->nrow, A->dia_ndiag, A->dia_offset, A->dia_val, in, out);
    else if(A->format == ORC_ELL)
        SUF(orc_ell_apply)(A->nrow, A->ell_max_row, A->ell_col, A->ell_val, in, out);
    else if(A->format == ORC_HYB)
        SUF(orc_hyb_apply)(A->nrow, A->nrow, A->ell_max_row, A->ell_col, A->ell_val, A->coo_nnz,
                           A->coo_row, A->coo_col, A->coo_val, in, out);
    else
        SUF(orc_csr_apply)(A->nrow, A->row_offset, A->col, A->val, in, out);
}

static int SUF(op_build)(SUF(orc_op) * A, int format, int nrow, int64_t nnz, const int* row_offset,
                         const int* col, const T* val)
{
    memset(A, 0, sizeof(*A));
    A->format     = ORC_CSR;
    A->nrow       = nrow;
    A->nnz        = nnz;
    A->row_offset = row_offset;
    A->col        = col;
    A->val        = val;
    if(format == ORC_ELL)
    {
        int w = orc_csr_ell_width(nrow, nnz, row_offset);
        if(w < 0) /* refused: stays CSR (src/base/local_matrix.cpp:2093-2114) */
            return 1;
        A->ell_max_row = w;
        A->ell_col     = (int*)malloc(sizeof(int) * (size_t)w * nrow + 16);
        A->ell_val     = (T*)malloc(sizeof(T) * (size_t)w * nrow + 16);
        SUF(orc_csr_to_ell_fill)(nrow, nnz, row_offset, col, val, w, A->ell_col, A->ell_val);
        A->format = ORC_ELL;
    }
    else if(format == ORC_DIA)
    {
        int nd = SUF(orc_csr_to_dia)(nrow, nrow, nnz, row_offset, col, val, NULL, NULL);
        if(nd < 0) /* refused: stays CSR */
            return 1;
        A->dia_ndiag  = nd;
        A->dia_offset = (int*)malloc(sizeof(int) * (size_t)nd + 16);
        A->dia_val    = (T*)malloc(sizeof(T) * (size_t)nd * nrow + 16);
        SUF(orc_csr_to_dia)(nrow, nrow, nnz, row_offset, col, val, A->dia_offset, A->dia_val);
        A->format = ORC_DIA;
    }
    else if(format == ORC_HYB)
    {
        int     w = orc_csr_hyb_width(nrow, nnz);
        int64_t c = orc_csr_hyb_coo_nnz(nrow, row_offset, w);
        A->ell_max_row = w;
        A->ell_col     = (int*)malloc(sizeof(int) * (size_t)w * nrow + 16);
        A->ell_val     = (T*)malloc(sizeof(T) * (size_t)w * nrow + 16);
        A->coo_nnz     = c;
        A->coo_row     = (int*)malloc(sizeof(int) * (size_t)c + 16);
        A->coo_col     = (int*)malloc(sizeof(int) * (size_t)c + 16);
        A->coo_val     = (T*)malloc(sizeof(T) * (size_t)c + 16);
        SUF(orc_csr_to_hyb_fill)(nrow, row_offset, col, val, w, A->ell_col, A->ell_val, A->coo_row,
                                 A->coo_col, A->coo_val);
        A->format = ORC_HYB;
    }
    return 1;
}

static void SUF(op_free)(SUF(orc_op) * A)
{
    free(A->ell_col);
    free(A->ell_val);
    free(A->coo_row);
    free(A->coo_col);
    free(A->coo_val);
    free(A->dia_offset);
    free(A->dia_val);
}

typedef struct
{
    int kind; /* ORC_PC_* */
    int n;
    /* Jacobi */
    T* inv_diag;
    /* ILU(0): factored copy (shares pattern with A) */
    const int* row_offset;
    const int* col;
    T*         lu_val;
    int64_t    nnz;
    /* IC: own lower-triangular pattern */
    int* ic_row_offset;
    int* ic_col;
    /* MC-SGS (decomposed, omega = 1) */
    int   num_blocks;
    int*  block_sizes;
    int*  block_offsets;
    int*  perm;
    int** blk_row_offset; /* [i*nb+j] */
    int** blk_col;
    T**   blk_val;
    int64_t* blk_nnz;
    T**   blk_inv_diag; /* diag_solver_[i] = Jacobi(block_ii) */
    T**   blk_diag; /* diag_block_[i] */
    T*    xperm; /* x_ */
    T*    xtmp;
    /* TriSolverAlg_Iterative (descriptor captured at build): persistent tmp_vec_ and the csritsv buffer */
    int    it_on, it_max_iter, it_use_tol;
    double it_tol;
    T*     it_tmp;
    T*     it_work;
} SUF(orc_pc);

static void SUF(pc_free)(SUF(orc_pc) * P)
{
    free(P->it_tmp);
    free(P->it_work);
    free(P->inv_diag);
    free(P->lu_val);
    if(P->kind == ORC_PC_SGS)
        free(P->xtmp);
    free(P->ic_row_offset);
    free(P->ic_col);
    if(ORC_PC_IS_MC(P->kind))
    {
        int nb = P->num_blocks;
        for(int i = 0; i < nb * nb; ++i)
        {
            free(P->blk_row_offset[i]);
            free(P->blk_col[i]);
            free(P->blk_val[i]);
        }
        for(int i = 0; i < nb; ++i)
        {
            free(P->blk_inv_diag[i]);
            free(P->blk_diag[i]);
        }
        free(P->blk_row_offset);
        free(P->blk_col);
        free(P->blk_val);
        free(P->blk_nnz);
        free(P->blk_inv_diag);
        free(P->blk_diag);
        free(P->block_sizes);
        free(P->block_offsets);
        free(P->perm);
        free(P->xperm);
        free(P->xtmp);
    }
    memset(P, 0, sizeof(*P));
}

/* Build: Jacobi  src/solvers/preconditioners/preconditioner.cpp:95-112
 *        ILU(0)  :449-470 (CloneFrom -> ILUpFactorize(0) -> LUAnalyse [host no-op])
 *        MC-SGS  src/solvers/preconditioners/preconditioner_multicolored.cpp:303-340
 *                (Analyse_ :162-177, Permute_ :180-187, Decompose_ :195-300) */
static int SUF(pc_build)(SUF(orc_pc) * P, int kind, int nrow, int64_t nnz, const int* row_offset,
                         const int* col, const T* val)
{
    memset(P, 0, sizeof(*P));
    P->kind = kind;
    P->n    = nrow;
    if(g_tri_iterative && (kind == ORC_PC_ILU0 || kind == ORC_PC_IC || kind == ORC_PC_GS || kind == ORC_PC_SGS))
    {
        /* ItLUAnalyse / ItLLAnalyse allocate tmp_vec_ zero-filled (host_matrix_csr.cpp:1469-1543, :1652-1726) */
        P->it_on       = 1;
        P->it_max_iter = g_it_max_iter;
        P->it_tol      = g_it_tol;
        P->it_use_tol  = g_it_use_tol;
        P->it_tmp      = (T*)calloc((size_t)(nrow > 0 ? nrow : 1), sizeof(T));
        P->it_work     = (T*)calloc((size_t)(nrow > 0 ? 2 * nrow : 1), sizeof(T));
    }
    if(kind == ORC_PC_JACOBI)
    {
        P->inv_diag = (T*)calloc((size_t)nrow, sizeof(T)); /* Allocate() zero-fills */
        SUF(orc_csr_extract_inv_diag)(nrow, row_offset, col, val, P->inv_diag);
    }
    else if(kind == ORC_PC_ILU0)
    {
        P->row_offset = row_offset;
        P->col        = col;
        P->nnz        = nnz;
        P->lu_val     = (T*)malloc(sizeof(T) * (size_t)nnz);
        memcpy(P->lu_val, val, sizeof(T) * (size_t)nnz);
        SUF(orc_csr_ilu0)(nrow, row_offset, col, P->lu_val);
    }
    else if(kind == ORC_PC_IC)
    {
        /* IC::Build (preconditioner.cpp:862-880): IC_ = ExtractL(op, diag = true); ICFactorize(&inv_diag) */
        int64_t nl = 0;
        for(int i = 0; i < nrow; ++i)
            for(int j = row_offset[i]; j < row_offset[i + 1]; ++j)
                if(col[j] <= i)
                    ++nl;
        P->ic_row_offset = (int*)calloc((size_t)nrow + 1, sizeof(int));
        P->ic_col        = (int*)malloc(sizeof(int) * (size_t)(nl > 0 ? nl : 1));
        P->lu_val        = (T*)malloc(sizeof(T) * (size_t)(nl > 0 ? nl : 1));
        int64_t k        = 0;
        for(int i = 0; i < nrow; ++i)
        {
            for(int j = row_offset[i]; j < row_offset[i + 1]; ++j)
                if(col[j] <= i)
                {
                    P->ic_col[k] = col[j];
                    P->lu_val[k] = val[j];
                    ++k;
                }
            P->ic_row_offset[i + 1] = (int)k;
        }
        P->nnz      = nl;
        P->inv_diag = (T*)calloc((size_t)nrow, sizeof(T));
        if(!SUF(orc_csr_ic_factorize)(nrow, P->ic_row_offset, P->ic_col, P->lu_val, P->inv_diag))
            return 0;
    }
    else if(kind == ORC_PC_GS || kind == ORC_PC_SGS)
    {
        /* GS / SGS (preconditioner.cpp:206-225 / :302-325): clone of A, LAnalyse(false) [+ UAnalyse(false)];
         * SGS::Build extracts the INVERSE diagonal into diag_entries_ (:318) */
        P->row_offset = row_offset;
        P->col        = col;
        P->nnz        = nnz;
        P->lu_val     = (T*)malloc(sizeof(T) * (size_t)nnz); /* the clone's values */
        memcpy(P->lu_val, val, sizeof(T) * (size_t)nnz);
        if(kind == ORC_PC_SGS)
        {
            P->inv_diag = (T*)calloc((size_t)nrow, sizeof(T));
            SUF(orc_csr_extract_inv_diag)(nrow, row_offset, col, val, P->inv_diag);
            P->xtmp = (T*)calloc((size_t)nrow, sizeof(T));
        }
    }
    else if(ORC_PC_IS_MC(kind))
    {
        /* MC-GS: same build as MC-SGS (class MultiColoredGS : MultiColoredSGS);
         * MC-ILU(0,1): preconditioner_multicolored_ilu.cpp:95-130 -- colouring of A itself (q = 1),
         * Permute_, Factorize_ = ILUpFactorize(0) = ILU0Factorize on the permuted matrix, Decompose_ */
        P->perm       = (int*)malloc(sizeof(int) * (size_t)nrow);
        int  nb       = 0;
        int* sizes    = (int*)malloc(sizeof(int) * (size_t)(nrow > 0 ? nrow : 1));
        orc_csr_multicoloring(nrow, nnz, row_offset, col, &nb, sizes, P->perm);
        P->num_blocks    = nb;
        P->block_sizes   = (int*)malloc(sizeof(int) * (size_t)nb);
        P->block_offsets = (int*)malloc(sizeof(int) * (size_t)(nb + 1));
        P->block_offsets[0] = 0;
        for(int i = 0; i < nb; ++i)
        {
            P->block_sizes[i]       = sizes[i];
            P->block_offsets[i + 1] = P->block_offsets[i] + sizes[i];
        }
        free(sizes);
        /* permuted copy */
        int* pro  = (int*)malloc(sizeof(int) * (size_t)(nrow + 1));
        int* pcol = (int*)malloc(sizeof(int) * (size_t)nnz);
        T*   pval = (T*)malloc(sizeof(T) * (size_t)nnz);
        SUF(orc_csr_permute)(nrow, nnz, row_offset, col, val, P->perm, pro, pcol, pval);
        if(kind == ORC_PC_MCILU)
            SUF(orc_csr_ilu0)(nrow, pro, pcol, pval);
        P->blk_row_offset = (int**)calloc((size_t)nb * nb, sizeof(int*));
        P->blk_col        = (int**)calloc((size_t)nb * nb, sizeof(int*));
        P->blk_val        = (T**)calloc((size_t)nb * nb, sizeof(T*));
        P->blk_nnz        = (int64_t*)calloc((size_t)nb * nb, sizeof(int64_t));
        P->blk_inv_diag   = (T**)calloc((size_t)nb, sizeof(T*));
        P->blk_diag       = (T**)calloc((size_t)nb, sizeof(T*));
        for(int i = 0; i < nb; ++i)
            for(int j = 0; j < nb; ++j)
            {
                int r0 = P->block_offsets[i], c0 = P->block_offsets[j];
                int rs = P->block_sizes[i], cs = P->block_sizes[j];
                int64_t m = SUF(orc_csr_extract_submatrix)(pro, pcol, pval, r0, c0, rs, cs, NULL,
                                                           NULL, NULL);
                int id                = i * nb + j;
                P->blk_nnz[id]        = m;
                P->blk_row_offset[id] = (int*)calloc((size_t)rs + 1, sizeof(int));
                P->blk_col[id]        = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
                P->blk_val[id]        = (T*)malloc(sizeof(T) * (size_t)(m > 0 ? m : 1));
                if(m > 0)
                    SUF(orc_csr_extract_submatrix)(pro, pcol, pval, r0, c0, rs, cs,
                                                   P->blk_row_offset[id], P->blk_col[id],
                                                   P->blk_val[id]);
            }
        for(int i = 0; i < nb; ++i)
        {
            int id = i * nb + i, rs = P->block_sizes[i];
            P->blk_diag[i]     = (T*)calloc((size_t)rs, sizeof(T));
            P->blk_inv_diag[i] = (T*)calloc((size_t)rs, sizeof(T));
            SUF(orc_csr_extract_diag)(rs, P->blk_row_offset[id], P->blk_col[id], P->blk_val[id],
                                      P->blk_diag[i]);
            /* Jacobi::Build on block ii; an empty block leaves an empty inverse diagonal,
             * which Jacobi::Solve treats as identity (preconditioner.cpp:137-152) */
            if(P->blk_nnz[id] > 0)
                SUF(orc_csr_extract_inv_diag)(rs, P->blk_row_offset[id], P->blk_col[id],
                                              P->blk_val[id], P->blk_inv_diag[i]);
            else
            {
                free(P->blk_inv_diag[i]);
                P->blk_inv_diag[i] = NULL;
            }
        }
        free(pro);
        free(pcol);
        free(pval);
        P->xperm = (T*)calloc((size_t)nrow, sizeof(T));
        P->xtmp  = (T*)calloc((size_t)nrow, sizeof(T));
    }
    return 1;
}

/* Solve(rhs, x):
 *   Jacobi  preconditioner.cpp:137-166 (x = inv_diag * rhs)
 *   ILU     preconditioner.cpp:501-511 -> LUSolve
 *   MC-SGS  preconditioner_multicolored.cpp:348-413 + preconditioner_multicolored_gs.cpp:127-199
 *   MC-GS   same frame, SolveL_/SolveD_ empty (preconditioner_multicolored_gs.cpp:250-288)
 *   MC-ILU  same frame, SolveL_ without the diagonal solve, SolveD_ empty, SolveR_ as MC-SGS
 *           (preconditioner_multicolored_ilu.cpp:187-232) */
static void SUF(pc_solve)(SUF(orc_pc) * P, const T* rhs, T* x)
{
    int n = P->n;
    if(P->kind == ORC_PC_JACOBI)
    {
        if(x != rhs)
            SUF(orc_pointwise_mult2)(n, x, P->inv_diag, rhs);
        else
            SUF(orc_pointwise_mult)(n, x, P->inv_diag);
    }
    else if(P->it_on && P->kind == ORC_PC_ILU0)
        SUF(orc_csr_itlusolve)(P->it_max_iter, P->it_tol, P->it_use_tol, n, P->nnz, P->row_offset, P->col,
                               P->lu_val, rhs, P->it_tmp, x, P->it_work);
    else if(P->it_on && P->kind == ORC_PC_IC)
        SUF(orc_csr_itllsolve)(P->it_max_iter, P->it_tol, P->it_use_tol, n, P->nnz, P->ic_row_offset, P->ic_col,
                               P->lu_val, rhs, P->it_tmp, x, P->it_work);
    else if(P->it_on && P->kind == ORC_PC_GS)
        SUF(orc_csr_itlsolve)(P->it_max_iter, P->it_tol, P->it_use_tol, n, P->nnz, P->row_offset, P->col, P->lu_val,
                              0, rhs, x, P->it_work);
    else if(P->it_on && P->kind == ORC_PC_SGS)
    {
        SUF(orc_csr_itlsolve)(P->it_max_iter, P->it_tol, P->it_use_tol, n, P->nnz, P->row_offset, P->col, P->lu_val,
                              0, rhs, P->xtmp, P->it_work);
        SUF(orc_pointwise_mult)(n, P->xtmp, P->inv_diag);
        SUF(orc_csr_itusolve)(P->it_max_iter, P->it_tol, P->it_use_tol, n, P->nnz, P->row_offset, P->col, P->lu_val,
                              0, P->xtmp, x, P->it_work);
    }
    else if(P->kind == ORC_PC_ILU0)
    {
        SUF(orc_csr_lusolve)(n, P->nnz, P->row_offset, P->col, P->lu_val, rhs, x);
    }
    else if(P->kind == ORC_PC_IC)
    {
        /* IC::Solve (preconditioner.cpp:916-925): LLSolve(rhs, inv_diag_entries_, x) */
        SUF(orc_csr_llsolve)(n, P->ic_row_offset, P->ic_col, P->lu_val, rhs, P->inv_diag, x);
    }
    else if(P->kind == ORC_PC_GS)
    {
        /* GS::Solve (preconditioner.cpp:250-257): LSolve with the stored (non-unit) diagonal */
        SUF(orc_csr_lsolve)(n, P->row_offset, P->col, P->lu_val, 0, rhs, x);
    }
    else if(P->kind == ORC_PC_SGS)
    {
        /* SGS::Solve (preconditioner.cpp:367-379): v = LSolve(rhs); v *= diag_entries_; x = USolve(v) */
        SUF(orc_csr_lsolve)(n, P->row_offset, P->col, P->lu_val, 0, rhs, P->xtmp);
        SUF(orc_pointwise_mult)(n, P->xtmp, P->inv_diag);
        SUF(orc_csr_usolve)(n, P->nnz, P->row_offset, P->col, P->lu_val, 0, P->xtmp, x);
    }
    else if(ORC_PC_IS_MC(P->kind))
    {
        const int do_l = (P->kind != ORC_PC_MCGS), l_diag = (P->kind == ORC_PC_MCSGS);
        const int do_d = (P->kind == ORC_PC_MCSGS);
        int nb = P->num_blocks;
        T*  xb = P->xtmp; /* x_block_[i] = xb + block_offsets[i] */
        /* ExtractRHSinX_: x = P rhs ; slices copied into x_block_ */
        SUF(orc_copy_permute)(n, x, rhs, P->perm);
        memcpy(xb, x, sizeof(T) * (size_t)n);
        /* SolveL_ */
        for(int i = 0; do_l && i < nb; ++i)
        {
            T* xi = xb + P->block_offsets[i];
            for(int j = 0; j < i; ++j)
            {
                int id = i * nb + j;
                if(P->blk_nnz[id] > 0)
                    SUF(orc_csr_apply_add)(P->block_sizes[i], P->blk_nnz[id],
                                           P->blk_row_offset[id], P->blk_col[id], P->blk_val[id],
                                           xb + P->block_offsets[j], (T)-1, xi);
            }
            if(l_diag && P->blk_inv_diag[i])
                SUF(orc_pointwise_mult)(P->block_sizes[i], xi, P->blk_inv_diag[i]);
        }
        /* SolveD_ */
        for(int i = 0; do_d && i < nb; ++i)
            SUF(orc_pointwise_mult)(P->block_sizes[i], xb + P->block_offsets[i], P->blk_diag[i]);
        /* SolveR_ (j descending) */
        for(int i = nb - 1; i >= 0; --i)
        {
            T* xi = xb + P->block_offsets[i];
            for(int j = nb - 1; j > i; --j)
            {
                int id = i * nb + j;
                if(P->blk_nnz[id] > 0)
                    SUF(orc_csr_apply_add)(P->block_sizes[i], P->blk_nnz[id],
                                           P->blk_row_offset[id], P->blk_col[id], P->blk_val[id],
                                           xb + P->block_offsets[j], (T)-1, xi);
            }
            if(P->blk_inv_diag[i])
                SUF(orc_pointwise_mult)(P->block_sizes[i], xi, P->blk_inv_diag[i]);
        }
        /* InsertSolution_: x_ = concat blocks ; x = P^T x_ */
        memcpy(P->xperm, xb, sizeof(T) * (size_t)n);
        SUF(orc_copy_permute_backward)(n, x, P->xperm, P->perm);
    }
}

/* stand-alone preconditioner apply for parity tests: builds, applies once, frees */
int SUF(orc_precond_apply)(int kind, int nrow, int64_t nnz, const int* row_offset, const int* col,
                           const T* val, const T* rhs, T* x)
{
    SUF(orc_pc) P;
    if(kind == ORC_PC_NONE)
    {
        memcpy(x, rhs, sizeof(T) * (size_t)nrow);
        return 1;
    }
    SUF(pc_build)(&P, kind, nrow, nnz, row_offset, col, val);
    SUF(pc_solve)(&P, rhs, x);
    SUF(pc_free)(&P);
    return 1;
}

/* the same preconditioner applied `reps` times to rhs, x carried over (x is in/out): the iterative triangular
 * solves start from x's content and keep their intermediate vector between applies */
int SUF(orc_precond_apply_rep)(int kind, int nrow, int64_t nnz, const int* row_offset, const int* col,
                               const T* val, const T* rhs, T* x, int reps)
{
    SUF(orc_pc) P;
    if(kind == ORC_PC_NONE)
    {
        memcpy(x, rhs, sizeof(T) * (size_t)nrow);
        return 1;
    }
    SUF(pc_build)(&P, kind, nrow, nnz, row_offset, col, val);
    for(int r = 0; r < reps; ++r)
        SUF(pc_solve)(&P, rhs, x);
    SUF(pc_free)(&P);
    return 1;
}

/* ---- Krylov drivers --------------------------------------------------------- */

static void SUF(residual)(const SUF(orc_op) * A, const T* rhs, const T* x, T* r)
{
    /* op->Apply(*x, r); r->ScaleAdd(-1, rhs);   (cg.cpp:388-389 and twins) */
    SUF(op_apply)(A, x, r);
    SUF(orc_scale_add)(A->nrow, r, (T)-1, rhs);
}

/* src/solvers/krylov/cg.cpp:291-362 (no preconditioner) and :366-446 (preconditioned) */
static void SUF(solve_cg)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x,
                          orc_iter_ctrl* ic)
{
    int n = A->nrow;
    T*  r = (T*)calloc((size_t)n, sizeof(T));
    T*  z = (T*)calloc((size_t)n, sizeof(T));
    T*  p = (T*)calloc((size_t)n, sizeof(T));
    T*  q = (T*)calloc((size_t)n, sizeof(T));
    T   alpha, beta, rho, rho_old;

    SUF(residual)(A, rhs, x, r);
    T res_norm = SUF(orc_norm)(n, r);
    if(orc_ic_init_residual(ic, fabs((double)res_norm)))
    {
        if(P)
        {
            SUF(pc_solve)(P, r, z);
            memcpy(p, z, sizeof(T) * (size_t)n);
            rho = SUF(orc_dot)(n, r, z);
        }
        else
        {
            memcpy(p, r, sizeof(T) * (size_t)n);
            rho = SUF(orc_dot)(n, r, r);
        }
        while(1)
        {
            SUF(op_apply)(A, p, q);
            alpha = rho / SUF(orc_dot)(n, p, q);
            SUF(orc_add_scale)(n, x, p, alpha);
            SUF(orc_add_scale)(n, r, q, -alpha);
            res_norm = SUF(orc_norm)(n, r);
            if(orc_ic_check_residual(ic, fabs((double)res_norm)))
                break;
            rho_old = rho;
            if(P)
            {
                SUF(pc_solve)(P, r, z);
                rho  = SUF(orc_dot)(n, r, z);
                beta = rho / rho_old;
                SUF(orc_scale_add)(n, p, beta, z);
            }
            else
            {
                rho  = SUF(orc_dot)(n, r, r);
                beta = rho / rho_old;
                SUF(orc_scale_add)(n, p, beta, r);
            }
        }
    }
    free(r);
    free(z);
    free(p);
    free(q);
}

/* src/solvers/krylov/gmres.cpp:565-607 Givens helpers */
static void SUF(gen_givens)(T dx, T dy, T* c, T* s)
{
    if(dy == (T)0)
    {
        *c = (T)1;
        *s = (T)0;
    }
    else if(dx == (T)0)
    {
        *c = (T)0;
        *s = (T)1;
    }
    else if(ORC_FABS(dy) > ORC_FABS(dx))
    {
        T tmp = dx / dy;
        *s    = (T)1 / (T)ORC_SQRT((T)1 + tmp * tmp);
        *c    = tmp * *s;
    }
    else
    {
        T tmp = dy / dx;
        *c    = (T)1 / (T)ORC_SQRT((T)1 + tmp * tmp);
        *s    = tmp * *c;
    }
}
static void SUF(app_givens)(T c, T s, T* dx, T* dy)
{
    T temp = *dx;
    *dx    = c * *dx + s * *dy;
    *dy    = -s * temp + c * *dy;
}

/* src/solvers/krylov/gmres.cpp:274-413 (no preconditioner) / :416-562 (left preconditioned).
 * H is (m+1) x m column-major: DENSE_IND(i,j,m+1,m) = i + j*(m+1) (matrix_formats_ind.hpp:30) */
static void SUF(solve_gmres)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x,
                             orc_iter_ctrl* ic, int size)
{
    int  n = A->nrow;
    T**  v = (T**)malloc(sizeof(T*) * (size_t)(size + 1));
    for(int i = 0; i <= size; ++i)
        v[i] = (T*)calloc((size_t)n, sizeof(T));
    T* z = (T*)calloc((size_t)n, sizeof(T));
    T* c = (T*)calloc((size_t)size, sizeof(T));
    T* s = (T*)calloc((size_t)size, sizeof(T));
    T* r = (T*)calloc((size_t)size + 1, sizeof(T));
    T* H = (T*)calloc((size_t)(size + 1) * size, sizeof(T));
    T  one = (T)1;
#define HIND(i, j) ((i) + (j) * (size + 1))

    if(P)
    {
        SUF(residual)(A, rhs, x, z);
        SUF(pc_solve)(P, z, v[0]);
    }
    else
        SUF(residual)(A, rhs, x, v[0]);
    for(int i = 0; i <= size; ++i)
        r[i] = (T)0;
    r[0] = SUF(orc_norm)(n, v[0]);

    if(orc_ic_init_residual(ic, fabs((double)r[0])))
    {
        while(1)
        {
            SUF(orc_scale)(n, v[0], one / r[0]);
            int i = 0;
            while(i < size)
            {
                if(P)
                {
                    SUF(op_apply)(A, v[i], z);
                    SUF(pc_solve)(P, z, v[i + 1]);
                }
                else
                    SUF(op_apply)(A, v[i], v[i + 1]);
                for(int k = 0; k <= i; ++k)
                {
                    int idx = HIND(k, i);
                    H[idx]  = SUF(orc_dot)(n, v[k], v[i + 1]);
                    SUF(orc_add_scale)(n, v[i + 1], v[k], -H[idx]);
                }
                int ii = HIND(i, i), ip1i = HIND(i + 1, i);
                H[ip1i] = SUF(orc_norm)(n, v[i + 1]);
                SUF(orc_scale)(n, v[i + 1], one / H[ip1i]);
                for(int k = 0; k < i; ++k)
                    SUF(app_givens)(c[k], s[k], &H[HIND(k, i)], &H[HIND(k + 1, i)]);
                SUF(gen_givens)(H[ii], H[ip1i], &c[i], &s[i]);
                SUF(app_givens)(c[i], s[i], &H[ii], &H[ip1i]);
                SUF(app_givens)(c[i], s[i], &r[i], &r[i + 1]);
                ++i;
                if(orc_ic_check_residual(ic, fabs((double)r[i])))
                    break;
            }
            for(int j = i - 1; j >= 0; --j)
            {
                r[j] /= H[HIND(j, j)];
                for(int k = 0; k < j; ++k)
                    r[k] -= H[HIND(k, j)] * r[j];
            }
            SUF(orc_add_scale)(n, x, v[0], r[0]);
            for(int j = 1; j < i; ++j)
                SUF(orc_add_scale)(n, x, v[j], r[j]);
            if(P)
            {
                SUF(residual)(A, rhs, x, z);
                SUF(pc_solve)(P, z, v[0]);
            }
            else
                SUF(residual)(A, rhs, x, v[0]);
            for(int k = 0; k <= size; ++k)
                r[k] = (T)0;
            r[0] = SUF(orc_norm)(n, v[0]);
            if(orc_ic_check_residual_nocount(ic, fabs((double)r[0])))
                break;
        }
    }
#undef HIND
    for(int i = 0; i <= size; ++i)
        free(v[i]);
    free(v);
    free(z);
    free(c);
    free(s);
    free(r);
    free(H);
}

/* src/solvers/krylov/fcg.cpp:232-318 (no preconditioner) / :321-420 (preconditioned).
 * The return value of InitResidual is NOT consulted there. */
static void SUF(solve_fcg)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x, orc_iter_ctrl* ic)
{
    int n = A->nrow;
    T*  r = (T*)calloc((size_t)n, sizeof(T));
    T*  w = (T*)calloc((size_t)n, sizeof(T));
    T*  z = (T*)calloc((size_t)n, sizeof(T));
    T*  p = (T*)calloc((size_t)n, sizeof(T));
    T*  q = (T*)calloc((size_t)n, sizeof(T));
    T   alpha, beta, rho, gamma, gamma_rho;
    SUF(residual)(A, rhs, x, r);
    T res = SUF(orc_norm)(n, r);
    (void)orc_ic_init_residual(ic, fabs((double)res));
    const T* zz = r; /* non-precond: z == r */
    if(P)
    {
        SUF(pc_solve)(P, r, z);
        zz = z;
    }
    SUF(op_apply)(A, zz, w);
    alpha = SUF(orc_dot)(n, zz, r);
    beta  = SUF(orc_dot)(n, zz, w);
    memcpy(p, zz, sizeof(T) * (size_t)n);
    memcpy(q, w, sizeof(T) * (size_t)n);
    rho = beta;
    SUF(orc_add_scale)(n, x, p, alpha / rho);
    SUF(orc_add_scale)(n, r, q, -alpha / rho);
    res = SUF(orc_norm)(n, r);
    while(!orc_ic_check_residual(ic, fabs((double)res)))
    {
        if(P)
            SUF(pc_solve)(P, r, z);
        SUF(op_apply)(A, zz, w);
        beta      = SUF(orc_dot)(n, zz, w);
        gamma     = SUF(orc_dot)(n, zz, q);
        gamma_rho = -gamma / rho;
        SUF(orc_scale_add)(n, p, gamma_rho, zz);
        SUF(orc_scale_add)(n, q, gamma_rho, w);
        rho   = beta + gamma * gamma_rho;
        alpha = SUF(orc_dot)(n, zz, r) / rho;
        SUF(orc_add_scale)(n, x, p, alpha);
        SUF(orc_add_scale)(n, r, q, -alpha);
        res = SUF(orc_norm)(n, r);
    }
    free(r);
    free(w);
    free(z);
    free(p);
    free(q);
}

/* src/solvers/krylov/cr.cpp:240-318 (no preconditioner) / :321-430 (preconditioned; the convergence
 * test runs on t, the unpreconditioned residual) */
static void SUF(solve_cr)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x, orc_iter_ctrl* ic)
{
    int n = A->nrow;
    T*  r = (T*)calloc((size_t)n, sizeof(T));
    T*  z = (T*)calloc((size_t)n, sizeof(T));
    T*  p = (T*)calloc((size_t)n, sizeof(T));
    T*  q = (T*)calloc((size_t)n, sizeof(T));
    T*  v = (T*)calloc((size_t)n, sizeof(T));
    T*  t = (T*)calloc((size_t)n, sizeof(T));
    T   alpha, beta, rho, rho_old, res_norm;
    if(!P)
    {
        SUF(residual)(A, rhs, x, r);
        memcpy(p, r, sizeof(T) * (size_t)n);
        res_norm = SUF(orc_norm)(n, r);
        if(orc_ic_init_residual(ic, fabs((double)res_norm)))
        {
            SUF(op_apply)(A, r, v);
            rho = SUF(orc_dot)(n, r, v);
            SUF(op_apply)(A, p, q);
            alpha = rho / SUF(orc_dot)(n, q, q);
            SUF(orc_add_scale)(n, x, p, alpha);
            SUF(orc_add_scale)(n, r, q, -alpha);
            res_norm = SUF(orc_norm)(n, r);
            while(!orc_ic_check_residual(ic, fabs((double)res_norm)))
            {
                rho_old = rho;
                SUF(op_apply)(A, r, v);
                rho  = SUF(orc_dot)(n, r, v);
                beta = rho / rho_old;
                SUF(orc_scale_add)(n, p, beta, r);
                SUF(orc_scale_add)(n, q, beta, v);
                alpha = rho / SUF(orc_dot)(n, q, q);
                SUF(orc_add_scale)(n, x, p, alpha);
                SUF(orc_add_scale)(n, r, q, -alpha);
                res_norm = SUF(orc_norm)(n, r);
            }
        }
    }
    else
    {
        SUF(residual)(A, rhs, x, z);
        SUF(pc_solve)(P, z, r);
        memcpy(p, r, sizeof(T) * (size_t)n);
        memcpy(t, z, sizeof(T) * (size_t)n);
        res_norm = SUF(orc_norm)(n, t);
        if(orc_ic_init_residual(ic, fabs((double)res_norm)))
        {
            SUF(op_apply)(A, r, v);
            rho = SUF(orc_dot)(n, r, v);
            SUF(op_apply)(A, p, q);
            SUF(pc_solve)(P, q, z);
            alpha = rho / SUF(orc_dot)(n, q, z);
            SUF(orc_add_scale)(n, x, p, alpha);
            SUF(orc_add_scale)(n, r, z, -alpha);
            SUF(orc_add_scale)(n, t, q, -alpha);
            res_norm = SUF(orc_norm)(n, t);
            while(!orc_ic_check_residual(ic, fabs((double)res_norm)))
            {
                rho_old = rho;
                SUF(op_apply)(A, r, v);
                rho  = SUF(orc_dot)(n, r, v);
                beta = rho / rho_old;
                SUF(orc_scale_add)(n, p, beta, r);
                SUF(orc_scale_add)(n, q, beta, v);
                SUF(pc_solve)(P, q, z);
                alpha = rho / SUF(orc_dot)(n, q, z);
                SUF(orc_add_scale)(n, x, p, alpha);
                SUF(orc_add_scale)(n, r, z, -alpha);
                SUF(orc_add_scale)(n, t, q, -alpha);
                res_norm = SUF(orc_norm)(n, t);
            }
        }
    }
    free(r);
    free(z);
    free(p);
    free(q);
    free(v);
    free(t);
}

/* src/solvers/krylov/fgmres.cpp:298-419 / :422-548: right-preconditioned GMRES that keeps every
 * z_i = M^-1 v_i (the residual itself is NOT preconditioned) */
static void SUF(solve_fgmres)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x,
                              orc_iter_ctrl* ic, int size)
{
    int  n = A->nrow;
    T**  v = (T**)malloc(sizeof(T*) * (size_t)(size + 1));
    T**  z = (T**)malloc(sizeof(T*) * (size_t)(size + 1));
    for(int i = 0; i <= size; ++i)
    {
        v[i] = (T*)calloc((size_t)n, sizeof(T));
        z[i] = P ? (T*)calloc((size_t)n, sizeof(T)) : NULL;
    }
    T* c = (T*)calloc((size_t)size, sizeof(T));
    T* s = (T*)calloc((size_t)size, sizeof(T));
    T* r = (T*)calloc((size_t)size + 1, sizeof(T));
    T* H = (T*)calloc((size_t)(size + 1) * size, sizeof(T));
    T  one = (T)1;
#define HIND(i, j) ((i) + (j) * (size + 1))
    SUF(residual)(A, rhs, x, v[0]);
    r[0] = SUF(orc_norm)(n, v[0]);
    if(orc_ic_init_residual(ic, fabs((double)r[0])))
    {
        while(1)
        {
            SUF(orc_scale)(n, v[0], one / r[0]);
            int i = 0;
            while(i < size)
            {
                if(P)
                {
                    SUF(pc_solve)(P, v[i], z[i]);
                    SUF(op_apply)(A, z[i], v[i + 1]);
                }
                else
                    SUF(op_apply)(A, v[i], v[i + 1]);
                for(int k = 0; k <= i; ++k)
                {
                    int idx = HIND(k, i);
                    H[idx]  = SUF(orc_dot)(n, v[k], v[i + 1]);
                    SUF(orc_add_scale)(n, v[i + 1], v[k], -H[idx]);
                }
                int ii = HIND(i, i), ip1i = HIND(i + 1, i);
                H[ip1i] = SUF(orc_norm)(n, v[i + 1]);
                SUF(orc_scale)(n, v[i + 1], one / H[ip1i]);
                for(int k = 0; k < i; ++k)
                    SUF(app_givens)(c[k], s[k], &H[HIND(k, i)], &H[HIND(k + 1, i)]);
                SUF(gen_givens)(H[ii], H[ip1i], &c[i], &s[i]);
                SUF(app_givens)(c[i], s[i], &H[ii], &H[ip1i]);
                SUF(app_givens)(c[i], s[i], &r[i], &r[i + 1]);
                ++i;
                if(orc_ic_check_residual(ic, fabs((double)r[i])))
                    break;
            }
            for(int j = i - 1; j >= 0; --j)
            {
                r[j] /= H[HIND(j, j)];
                for(int k = 0; k < j; ++k)
                    r[k] -= H[HIND(k, j)] * r[j];
            }
            T** upd = P ? z : v;
            SUF(orc_add_scale)(n, x, upd[0], r[0]);
            for(int j = 1; j < i; ++j)
                SUF(orc_add_scale)(n, x, upd[j], r[j]);
            SUF(residual)(A, rhs, x, v[0]);
            for(int k = 0; k <= size; ++k)
                r[k] = (T)0;
            r[0] = SUF(orc_norm)(n, v[0]);
            if(orc_ic_check_residual_nocount(ic, fabs((double)r[0])))
                break;
        }
    }
#undef HIND
    for(int i = 0; i <= size; ++i)
    {
        free(v[i]);
        free(z[i]);
    }
    free(v);
    free(z);
    free(c);
    free(s);
    free(r);
    free(H);
}

/* src/solvers/krylov/bicgstabl.cpp:292-496 / :499-695.  The preconditioned variant is LEFT
 * preconditioned: every A u / A r is followed by M^-1, the residual in the convergence test is the
 * preconditioned one.  The return value of InitResidual is not consulted. */
static void SUF(solve_bicgstabl)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x,
                                 orc_iter_ctrl* ic, int l)
{
    int  n  = A->nrow;
    T*   r0 = (T*)calloc((size_t)n, sizeof(T));
    T*   z  = (T*)calloc((size_t)n, sizeof(T));
    T**  r  = (T**)malloc(sizeof(T*) * (size_t)(l + 1));
    T**  u  = (T**)malloc(sizeof(T*) * (size_t)(l + 1));
    for(int i = 0; i <= l; ++i)
    {
        r[i] = (T*)calloc((size_t)n, sizeof(T));
        u[i] = (T*)calloc((size_t)n, sizeof(T));
    }
    T* gamma0 = (T*)calloc((size_t)l, sizeof(T));
    T* gamma1 = (T*)calloc((size_t)l, sizeof(T));
    T* gamma2 = (T*)calloc((size_t)l, sizeof(T));
    T* sigma  = (T*)calloc((size_t)l, sizeof(T));
    T* tau    = (T*)calloc((size_t)l * l, sizeof(T)); /* tau[i][j] -> tau[i*l+j] */
    int converged = 0;
    T   alpha = (T)0, beta = (T)0, omega = (T)1, rho_old = (T)-1, rho, res;
    if(P)
    {
        SUF(residual)(A, rhs, x, z);
        SUF(pc_solve)(P, z, r0);
    }
    else
        SUF(residual)(A, rhs, x, r0);
    res = SUF(orc_norm)(n, r0);
    (void)orc_ic_init_residual(ic, fabs((double)res));
    memcpy(r[0], r0, sizeof(T) * (size_t)n);
    memset(u[0], 0, sizeof(T) * (size_t)n);
    while(1)
    {
        rho_old *= -omega;
        for(int j = 0; j < l; ++j)
        {
            rho = SUF(orc_dot)(n, r0, r[j]);
            if(rho == (T)0)
            {
                converged = 1;
                break;
            }
            beta = alpha * rho / rho_old;
            for(int i = 0; i <= j; ++i)
                SUF(orc_scale_add)(n, u[i], -beta, r[i]);
            if(P)
            {
                SUF(op_apply)(A, u[j], z);
                SUF(pc_solve)(P, z, u[j + 1]);
            }
            else
                SUF(op_apply)(A, u[j], u[j + 1]);
            rho_old = SUF(orc_dot)(n, r0, u[j + 1]);
            if(rho_old == (T)0)
            {
                converged = 1;
                break;
            }
            alpha   = rho / rho_old;
            rho_old = rho;
            for(int i = 0; i <= j; ++i)
                SUF(orc_add_scale)(n, r[i], u[i + 1], -alpha);
            if(P)
            {
                SUF(op_apply)(A, r[j], z);
                SUF(pc_solve)(P, z, r[j + 1]);
            }
            else
                SUF(op_apply)(A, r[j], r[j + 1]);
            SUF(orc_add_scale)(n, x, u[0], alpha);
            res = SUF(orc_norm)(n, r[0]);
            if(orc_ic_check_residual_nocount(ic, fabs((double)res)))
            {
                converged = 1;
                break;
            }
        }
        if(converged)
            break;
        for(int j = 0; j < l; ++j)
        {
            for(int i = 0; i < j; ++i)
            {
                tau[i * l + j] = SUF(orc_dot)(n, r[j + 1], r[i + 1]) / sigma[i];
                SUF(orc_add_scale)(n, r[j + 1], r[i + 1], -tau[i * l + j]);
            }
            sigma[j]  = SUF(orc_dot)(n, r[j + 1], r[j + 1]);
            gamma1[j] = SUF(orc_dot)(n, r[0], r[j + 1]) / sigma[j];
        }
        gamma0[l - 1] = gamma1[l - 1];
        omega         = gamma1[l - 1];
        for(int j = l - 2; j >= 0; --j)
        {
            gamma0[j] = gamma1[j];
            for(int i = j + 1; i < l; ++i)
                gamma0[j] -= tau[j * l + i] * gamma0[i];
        }
        for(int j = 0; j < l - 1; ++j)
        {
            gamma2[j] = gamma0[j + 1];
            for(int i = j + 1; i < l - 1; ++i)
                gamma2[j] += tau[j * l + i] * gamma0[i + 1];
        }
        SUF(orc_add_scale)(n, x, r[0], gamma0[0]);
        SUF(orc_add_scale)(n, r[0], r[l], -gamma1[l - 1]);
        SUF(orc_add_scale)(n, u[0], u[l], -gamma0[l - 1]);
        for(int j = 1; j < l; ++j)
        {
            SUF(orc_add_scale)(n, u[0], u[j], -gamma0[j - 1]);
            SUF(orc_add_scale)(n, x, r[j], gamma2[j - 1]);
            SUF(orc_add_scale)(n, r[0], r[j], -gamma1[j - 1]);
        }
        res = SUF(orc_norm)(n, r[0]);
        if(orc_ic_check_residual(ic, fabs((double)res)))
            break;
    }
    for(int i = 0; i <= l; ++i)
    {
        free(r[i]);
        free(u[i]);
    }
    free(r);
    free(u);
    free(r0);
    free(z);
    free(gamma0);
    free(gamma1);
    free(gamma2);
    free(sigma);
    free(tau);
}

/* src/solvers/krylov/qmrcgstab.cpp:262-460 / :463-690 (right preconditioned through z).
 * The residual handed to the iteration control is the bound sqrt(#iter+1)*|tau|; after the loop the
 * true residual is computed and checked once more (which counts one more iteration). */
static void SUF(solve_qmrcgstab)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x,
                                 orc_iter_ctrl* ic)
{
    int n  = A->nrow;
    T*  r0 = (T*)calloc((size_t)n, sizeof(T));
    T*  r  = (T*)calloc((size_t)n, sizeof(T));
    T*  p  = (T*)calloc((size_t)n, sizeof(T)); /* Build() zero-fills; p += r below */
    T*  t  = (T*)calloc((size_t)n, sizeof(T));
    T*  v  = (T*)calloc((size_t)n, sizeof(T));
    T*  d  = (T*)calloc((size_t)n, sizeof(T));
    T*  z  = (T*)calloc((size_t)n, sizeof(T));
    T   alpha, beta, omega, theta1, theta1sq, theta2, theta2sq, eta1, eta2, tau1, tau2, rho, rho_old, c;
    SUF(residual)(A, rhs, x, r0);
    memcpy(r, r0, sizeof(T) * (size_t)n);
    tau2            = SUF(orc_norm)(n, r0);
    double res_norm = fabs((double)tau2);
    (void)orc_ic_init_residual(ic, res_norm);
    rho  = SUF(orc_dot)(n, r0, r);
    beta = rho;
    (void)beta;
    SUF(orc_add_scale)(n, p, r, (T)1);
    const T* pz = p; /* direction handed to A: p, or z = M^-1 p */
    if(P)
    {
        SUF(pc_solve)(P, p, z);
        pz = z;
    }
    SUF(op_apply)(A, pz, v);
    rho_old = SUF(orc_dot)(n, r0, v);
    alpha   = rho / rho_old;
    SUF(orc_add_scale)(n, r, v, -alpha);
    theta1   = SUF(orc_norm)(n, r) / tau2;
    theta1sq = theta1 * theta1;
    c        = (T)1 / (T)sqrt((double)((T)1 + theta1sq));
    tau1     = tau2 * theta1 * c;
    eta1     = c * c * alpha;
    memcpy(d, pz, sizeof(T) * (size_t)n);
    SUF(orc_add_scale)(n, x, d, eta1);
    const T* rz = r;
    if(P)
    {
        SUF(pc_solve)(P, r, z);
        rz = z;
    }
    SUF(op_apply)(A, rz, t);
    omega = SUF(orc_dot)(n, t, r) / SUF(orc_dot)(n, t, t);
    SUF(orc_scale_add)(n, d, theta1sq * eta1 / omega, rz);
    SUF(orc_add_scale)(n, r, t, -omega);
    theta2   = SUF(orc_norm)(n, r) / tau1;
    theta2sq = theta2 * theta2;
    c        = (T)1 / (T)sqrt((double)((T)1 + theta2sq));
    tau2     = tau1 * theta2 * c;
    eta2     = c * c * omega;
    SUF(orc_add_scale)(n, x, d, eta2);
    res_norm = sqrt((double)(ic->iteration + 1)) * fabs((double)tau2);
    while(!orc_ic_check_residual(ic, res_norm))
    {
        rho_old = rho;
        rho     = SUF(orc_dot)(n, r0, r);
        beta    = (rho * alpha) / (rho_old * omega);
        SUF(orc_add_scale)(n, p, v, -omega);
        SUF(orc_scale)(n, p, beta);
        SUF(orc_add_scale)(n, p, r, (T)1);
        if(P)
            SUF(pc_solve)(P, p, z);
        SUF(op_apply)(A, pz, v);
        rho_old = SUF(orc_dot)(n, r0, v);
        if(rho_old == (T)0)
            break;
        alpha = rho / rho_old;
        SUF(orc_add_scale)(n, r, v, -alpha);
        theta1   = SUF(orc_norm)(n, r) / tau2;
        theta1sq = theta1 * theta1;
        c        = (T)1 / (T)sqrt((double)((T)1 + theta1sq));
        tau1     = tau2 * theta1 * c;
        eta1     = c * c * alpha;
        SUF(orc_scale_add)(n, d, theta2sq * eta2 / alpha, pz);
        SUF(orc_add_scale)(n, x, d, eta1);
        if(P)
            SUF(pc_solve)(P, r, z);
        SUF(op_apply)(A, rz, t);
        omega = SUF(orc_dot)(n, t, t);
        if(omega == (T)0)
            break;
        omega = SUF(orc_dot)(n, t, r) / omega;
        SUF(orc_scale_add)(n, d, theta1sq * eta1 / omega, rz);
        SUF(orc_add_scale)(n, r, t, -omega);
        theta2   = SUF(orc_norm)(n, r) / tau1;
        theta2sq = theta2 * theta2;
        c        = (T)1 / (T)sqrt((double)((T)1 + theta2sq));
        tau2     = tau1 * theta2 * c;
        eta2     = c * c * omega;
        SUF(orc_add_scale)(n, x, d, eta2);
        res_norm = sqrt((double)(ic->iteration + 1)) * fabs((double)tau2);
    }
    SUF(residual)(A, rhs, x, r0);
    (void)orc_ic_check_residual(ic, fabs((double)SUF(orc_norm)(n, r0)));
    free(r0);
    free(r);
    free(p);
    free(t);
    free(v);
    free(d);
    free(z);
}

/* HostVector::SetRandomNormal (src/base/host/host_vector.cpp:388-405): srand/rand Box-Muller */
void SUF(orc_set_random_normal)(int64_t n, T* v, unsigned long long seed, T mean, T var)
{
    srand((unsigned)seed);
    for(int64_t i = 0; i < n; ++i)
    {
        T u1 = (T)rand() / (T)RAND_MAX;
        T u2 = (T)rand() / (T)RAND_MAX;
#ifdef ORC_T_IS_FLOAT
        v[i] = sqrtf((T)-2 * logf(u1)) * cosf((T)(2 * M_PI) * u2);
#else
        v[i] = sqrt((T)-2 * log(u1)) * cos((T)(2 * M_PI) * u2);
#endif
        v[i] = mean + var * v[i];
    }
}

/* src/solvers/krylov/idr.cpp: Build :127-185 (shadow space P: random normal vectors made orthonormal
 * by modified Gram-Schmidt), SolveNonPrecond_ :335-520, SolvePrecond_ :523-730 */
static void SUF(solve_idr)(const SUF(orc_op) * A, SUF(orc_pc) * P_, const T* rhs, T* x, orc_iter_ctrl* ic,
                           int s, unsigned long long seed)
{
    int  n = A->nrow;
    T*   r = (T*)calloc((size_t)n, sizeof(T));
    T*   v = (T*)calloc((size_t)n, sizeof(T));
    T*   t = (T*)calloc((size_t)n, sizeof(T));
    T**  G = (T**)malloc(sizeof(T*) * (size_t)s);
    T**  U = (T**)malloc(sizeof(T*) * (size_t)s);
    T**  P = (T**)malloc(sizeof(T*) * (size_t)s);
    T*   c = (T*)calloc((size_t)s, sizeof(T));
    T*   f = (T*)calloc((size_t)s, sizeof(T));
    T*   M = (T*)calloc((size_t)s * s, sizeof(T));
#define MIND(i, j) ((i) + (j) * s)
    const T zero = (T)0, one = (T)1, kappa = (T)0.7f;
    T       alpha, beta, rho, omega = one;
    for(int i = 0; i < s; ++i)
    {
        G[i] = (T*)calloc((size_t)n, sizeof(T));
        U[i] = (T*)calloc((size_t)n, sizeof(T));
        P[i] = (T*)calloc((size_t)n, sizeof(T));
        SUF(orc_set_random_normal)(n, P[i], (unsigned long long)(i + 1) * seed, (T)0.0, (T)1.0);
    }
    for(int k = 0; k < s; ++k)
    {
        SUF(orc_scale)(n, P[k], one / SUF(orc_norm)(n, P[k]));
        T invdotk = one / SUF(orc_dot)(n, P[k], P[k]);
        for(int j = k + 1; j < s; ++j)
            SUF(orc_add_scale)(n, P[j], P[k], -SUF(orc_dot)(n, P[j], P[k]) * invdotk);
    }
    SUF(residual)(A, rhs, x, r);
    T res_norm = SUF(orc_norm)(n, r);
    if(orc_ic_init_residual(ic, fabs((double)res_norm)))
    {
        for(int i = 0; i < s; ++i)
            for(int j = 0; j < s; ++j)
                M[MIND(i, j)] = (i == j) ? one : zero;
        while(1)
        {
            for(int i = 0; i < s; ++i)
                f[i] = SUF(orc_dot)(n, P[i], r);
            int stop = 0;
            for(int k = 0; k < s; ++k)
            {
                memcpy(v, r, sizeof(T) * (size_t)n);
                for(int i = k; i < s; ++i)
                {
                    c[i] = f[i];
                    for(int j = k; j < i; ++j)
                        c[i] -= M[MIND(i, j)] * c[j];
                    c[i] /= M[MIND(i, i)];
                    SUF(orc_add_scale)(n, v, G[i], -c[i]);
                }
                if(P_)
                {
                    SUF(pc_solve)(P_, v, t);
                    SUF(orc_scale_add_scale)(n, U[k], c[k], t, omega);
                }
                else
                    SUF(orc_scale_add_scale)(n, U[k], c[k], v, omega);
                for(int i = k + 1; i < s; ++i)
                    SUF(orc_add_scale)(n, U[k], U[i], c[i]);
                SUF(op_apply)(A, U[k], G[k]);
                for(int i = 0; i < k; ++i)
                {
                    alpha = SUF(orc_dot)(n, P[i], G[k]) / M[MIND(i, i)];
                    SUF(orc_add_scale)(n, G[k], G[i], -alpha);
                    SUF(orc_add_scale)(n, U[k], U[i], -alpha);
                }
                for(int i = k; i < s; ++i)
                    M[MIND(i, k)] = SUF(orc_dot)(n, P[i], G[k]);
                if(M[MIND(k, k)] == zero || M[MIND(k, k)] != M[MIND(k, k)] || M[MIND(k, k)] == (T)INFINITY)
                {
                    stop = 2; /* the reference aborts here (FATAL_ERROR) */
                    break;
                }
                beta = f[k] / M[MIND(k, k)];
                SUF(orc_add_scale)(n, r, G[k], -beta);
                SUF(orc_add_scale)(n, x, U[k], beta);
                res_norm = SUF(orc_norm)(n, r);
                if(orc_ic_check_residual_nocount(ic, fabs((double)res_norm)))
                    break;
                for(int i = k + 1; i < s; ++i)
                    f[i] -= beta * M[MIND(i, k)];
            }
            if(stop == 2)
                break;
            if(orc_ic_check_residual(ic, fabs((double)res_norm)))
                break;
            T rt, nt;
            if(P_)
            {
                SUF(pc_solve)(P_, r, v);
                SUF(op_apply)(A, v, t);
                rt = SUF(orc_dot)(n, t, r);
                nt = SUF(orc_norm)(n, t);
            }
            else
            {
                SUF(op_apply)(A, r, v);
                rt = SUF(orc_dot)(n, v, r);
                nt = SUF(orc_norm)(n, v);
            }
            rt /= nt;
            rho   = (T)fabs((double)(rt / res_norm));
            omega = rt / nt;
            if(rho < kappa)
                omega *= kappa / rho;
            if(omega == zero || omega != omega || omega == (T)INFINITY)
                break; /* FATAL_ERROR in the reference */
            if(P_)
            {
                SUF(orc_add_scale)(n, r, t, -omega);
                SUF(orc_add_scale)(n, x, v, omega);
            }
            else
            {
                SUF(orc_add_scale)(n, x, r, omega);
                SUF(orc_add_scale)(n, r, v, -omega);
            }
            res_norm = SUF(orc_norm)(n, r);
        }
    }
#undef MIND
    for(int i = 0; i < s; ++i)
    {
        free(G[i]);
        free(U[i]);
        free(P[i]);
    }
    free(G);
    free(U);
    free(P);
    free(c);
    free(f);
    free(M);
    free(r);
    free(v);
    free(t);
}

/* src/solvers/solver.cpp:679-775 FixedPoint::SolvePrecond_ (modified Richardson, x += omega M^-1 (b - A x));
 * the smoother form (FlagSmoother) runs max_iter steps without any norm */
static void SUF(solve_fixedpoint)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x, orc_iter_ctrl* ic,
                                  T omega, int smoother)
{
    int n     = A->nrow;
    T*  x_res = (T*)calloc((size_t)n, sizeof(T));
    T*  x_old = (T*)calloc((size_t)n, sizeof(T));
    if(P && smoother)
    {
        int steps = ic->max_iter;
        if(steps >= 1)
        {
            (void)orc_ic_init_residual(ic, 1.0);
            for(int iter = 0; iter < steps; ++iter)
            {
                SUF(residual)(A, rhs, x, x_res);
                SUF(pc_solve)(P, x_res, x_old);
                SUF(orc_add_scale)(n, x, x_old, omega);
            }
        }
    }
    else if(P && ic->max_iter >= 1)
    {
        SUF(residual)(A, rhs, x, x_res);
        T res = SUF(orc_norm)(n, x_res);
        if(orc_ic_init_residual(ic, fabs((double)res)))
        {
            while(1)
            {
                SUF(pc_solve)(P, x_res, x_old);
                SUF(orc_add_scale)(n, x, x_old, omega);
                if(orc_ic_check_max_iter_nocount(ic))
                    break;
                SUF(residual)(A, rhs, x, x_res);
                res = SUF(orc_norm)(n, x_res);
                if(orc_ic_check_residual(ic, fabs((double)res)))
                    break;
            }
        }
    }
    free(x_res);
    free(x_old);
}

/* src/solvers/chebyshev.cpp:230-287 / :290-360 (lambda_min / lambda_max set by the caller) */
static void SUF(solve_chebyshev)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x, orc_iter_ctrl* ic,
                                 T lambda_min, T lambda_max)
{
    int n = A->nrow;
    T*  r = (T*)calloc((size_t)n, sizeof(T));
    T*  z = (T*)calloc((size_t)n, sizeof(T));
    T*  p = (T*)calloc((size_t)n, sizeof(T));
    T   two = (T)2, alpha, beta;
    T   d = (lambda_max + lambda_min) / two;
    T   c = (lambda_max - lambda_min) / two;
    SUF(residual)(A, rhs, x, r);
    T res = SUF(orc_norm)(n, r);
    if(orc_ic_init_residual(ic, fabs((double)res)))
    {
        const T* zz = r;
        if(P)
        {
            SUF(pc_solve)(P, r, z);
            zz = z;
        }
        memcpy(p, zz, sizeof(T) * (size_t)n);
        alpha = two / d;
        SUF(orc_add_scale)(n, x, p, alpha);
        SUF(residual)(A, rhs, x, r);
        res = SUF(orc_norm)(n, r);
        while(!orc_ic_check_residual(ic, fabs((double)res)))
        {
            if(P)
                SUF(pc_solve)(P, r, z);
            beta  = (c * alpha / two) * (c * alpha / two);
            alpha = (T)1 / (d - beta);
            SUF(orc_scale_add)(n, p, beta, zz);
            SUF(orc_add_scale)(n, x, p, alpha);
            SUF(residual)(A, rhs, x, r);
            res = SUF(orc_norm)(n, r);
        }
    }
    free(r);
    free(z);
    free(p);
}

/* src/solvers/krylov/bicgstab.cpp:245-361 (no preconditioner) / :365-489 (right preconditioned) */
static void SUF(solve_bicgstab)(const SUF(orc_op) * A, SUF(orc_pc) * P, const T* rhs, T* x,
                                orc_iter_ctrl* ic)
{
    int n  = A->nrow;
    T*  r  = (T*)calloc((size_t)n, sizeof(T));
    T*  r0 = (T*)calloc((size_t)n, sizeof(T));
    T*  p  = (T*)calloc((size_t)n, sizeof(T));
    T*  q  = (T*)calloc((size_t)n, sizeof(T));
    T*  t  = (T*)calloc((size_t)n, sizeof(T));
    T*  v  = (T*)calloc((size_t)n, sizeof(T));
    T*  z  = (T*)calloc((size_t)n, sizeof(T));
    T   alpha, beta, omega, rho, rho_old;

    SUF(residual)(A, rhs, x, r0);
    T res_norm = SUF(orc_norm)(n, r0);
    if(orc_ic_init_residual(ic, fabs((double)res_norm)))
    {
        memcpy(r, r0, sizeof(T) * (size_t)n);
        memcpy(p, r, sizeof(T) * (size_t)n);
        rho = SUF(orc_dot)(n, r, r);
        if(P)
            SUF(pc_solve)(P, r, z);
        while(1)
        {
            const T* dir = P ? z : p; /* q = A z  (precond)  /  q = A p */
            SUF(op_apply)(A, dir, q);
            alpha = rho / SUF(orc_dot)(n, r0, q);
            SUF(orc_add_scale)(n, r, q, -alpha);
            const T* sv = r;
            if(P)
            {
                SUF(pc_solve)(P, r, v);
                sv = v;
            }
            SUF(op_apply)(A, sv, t);
            omega = SUF(orc_dot)(n, t, r) / SUF(orc_dot)(n, t, t);
            if((ORC_FABS(omega) == (T)INFINITY) || (omega != omega) || (omega == (T)0))
            {
                SUF(orc_add_scale)(n, x, p, alpha);
                SUF(op_apply)(A, x, p);
                SUF(orc_scale_add)(n, p, (T)-1, rhs);
                res_norm = SUF(orc_norm)(n, p);
                orc_ic_check_residual(ic, fabs((double)res_norm));
                break;
            }
            /* x = x + alpha*dir + omega*sv */
            SUF(orc_scale_add2)(n, x, (T)1, dir, alpha, sv, omega);
            SUF(orc_add_scale)(n, r, t, -omega);
            res_norm = SUF(orc_norm)(n, r);
            if(orc_ic_check_residual(ic, fabs((double)res_norm)))
                break;
            rho_old = rho;
            rho     = SUF(orc_dot)(n, r0, r);
            if(rho == (T)0)
                break;
            beta = (rho / rho_old) * (alpha / omega);
            SUF(orc_scale_add2)(n, p, beta, q, -beta * omega, r, (T)1);
            if(P)
                SUF(pc_solve)(P, p, z);
        }
    }
    free(r);
    free(r0);
    free(p);
    free(q);
    free(t);
    free(v);
    free(z);
}

/* Build()+Solve() of one solver/preconditioner/format combination.
 * The preconditioner is built from the CSR state and the operator is converted afterwards,
 * as in the reference tests (clients/include/testing_cg.hpp:151-155). */
int SUF(orc_solve)(int nrow, int64_t nnz, const int* row_offset, const int* col, const T* val,
                   const T* rhs, T* x, orc_solve_cfg* cfg)
{
    SUF(orc_op) A;
    SUF(orc_pc) P;
    orc_iter_ctrl ic;
    int           have_pc = cfg->precond != ORC_PC_NONE;
    int*          brp     = NULL; /* block-diagonal copy (the preconditioner keeps pointers into its pattern) */
    int*          bci     = NULL;
    T*            bva     = NULL;
    if(have_pc && cfg->nblocks > 1 && cfg->precond != ORC_PC_JACOBI)
    {
        /* BlockJacobi over cfg->nblocks ranks: every rank builds the preconditioner from its interior block only
         * (preconditioner_blockjacobi.cpp:80-141, GlobalMatrix::GetInterior) and applies it to its slice of the vectors.
         * Blocks do not couple, so this equals the same preconditioner built from the block-diagonal part of the
         * operator: factorisations stay inside the blocks, a greedy colouring sees only in-block neighbours (the same
         * colours as colouring every block by itself), and permuted sweeps keep the relative order inside a block. */
        const int P_     = cfg->nblocks;
        brp              = (int*)malloc(sizeof(int) * ((size_t)nrow + 1));
        bci              = (int*)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
        bva              = (T*)malloc(sizeof(T) * (size_t)(nnz > 0 ? nnz : 1));
        int64_t   k      = 0;
        int       lo     = 0;
        brp[0]           = 0;
        for(int b = 0; b < P_; ++b)
        {
            const int size = nrow / P_ + (b < nrow % P_ ? 1 : 0); /* common.hpp:92-113 */
            const int hi   = lo + size;
            for(int i = lo; i < hi; ++i)
            {
                for(int j = row_offset[i]; j < row_offset[i + 1]; ++j)
                    if(col[j] >= lo && col[j] < hi)
                    {
                        bci[k] = col[j];
                        bva[k] = val[j];
                        ++k;
                    }
                brp[i + 1] = (int)k;
            }
            lo = hi;
        }
        SUF(pc_build)(&P, cfg->precond, nrow, k, brp, bci, bva);
    }
    else if(have_pc)
        SUF(pc_build)(&P, cfg->precond, nrow, nnz, row_offset, col, val);
    SUF(op_build)(&A, cfg->format, nrow, nnz, row_offset, col, val);
    orc_ic_setup(&ic, cfg);
    if(cfg->solver == ORC_CG)
        SUF(solve_cg)(&A, have_pc ? &P : NULL, rhs, x, &ic);
    else if(cfg->solver == ORC_GMRES)
        SUF(solve_gmres)(&A, have_pc ? &P : NULL, rhs, x, &ic, cfg->basis > 0 ? cfg->basis : 30);
    else if(cfg->solver == ORC_BICGSTAB)
        SUF(solve_bicgstab)(&A, have_pc ? &P : NULL, rhs, x, &ic);
    else if(cfg->solver == ORC_FCG)
        SUF(solve_fcg)(&A, have_pc ? &P : NULL, rhs, x, &ic);
    else if(cfg->solver == ORC_CR)
        SUF(solve_cr)(&A, have_pc ? &P : NULL, rhs, x, &ic);
    else if(cfg->solver == ORC_FGMRES)
        SUF(solve_fgmres)(&A, have_pc ? &P : NULL, rhs, x, &ic, cfg->basis > 0 ? cfg->basis : 30);
    else if(cfg->solver == ORC_BICGSTABL)
        SUF(solve_bicgstabl)(&A, have_pc ? &P : NULL, rhs, x, &ic, cfg->basis > 0 ? cfg->basis : 2);
    else if(cfg->solver == ORC_QMRCGSTAB)
        SUF(solve_qmrcgstab)(&A, have_pc ? &P : NULL, rhs, x, &ic);
    else if(cfg->solver == ORC_FIXEDPOINT)
        SUF(solve_fixedpoint)(&A, have_pc ? &P : NULL, rhs, x, &ic, cfg->p0 != 0.0 ? (T)cfg->p0 : (T)1,
                              cfg->p1 != 0.0);
    else if(cfg->solver == ORC_CHEBYSHEV)
        SUF(solve_chebyshev)(&A, have_pc ? &P : NULL, rhs, x, &ic, (T)cfg->p0, (T)cfg->p1);
    else if(cfg->solver == ORC_IDR)
        SUF(solve_idr)(&A, have_pc ? &P : NULL, rhs, x, &ic, cfg->basis > 0 ? cfg->basis : 4,
                       cfg->seed ? cfg->seed : 1ULL);
    else
        return 0;
    orc_ic_finish(&ic, cfg);
    SUF(op_free)(&A);
    if(have_pc)
        SUF(pc_free)(&P);
    free(brp);
    free(bci);
    free(bva);
    return 1;
}
