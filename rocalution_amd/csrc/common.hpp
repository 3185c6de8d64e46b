// common.hpp -- shared definitions of the MI355X (gfx950) backend.
//
// Replaces the role of src/base/hip/{hip_utils.hpp,hip_allocate_free.*,backend_hip.*} of the
// reference (error macros, allocation, stream bookkeeping) -- written from scratch, no
// rocSPARSE/rocBLAS/rocPRIM anywhere.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rocalution_amd.h"

namespace ramd
{

// ---------------------------------------------------------------- error handling
void        set_error(const char* file, int line, const std::string& msg);
const char* last_error();

#define RAMD_FAIL(code, msg)                          \
    do                                                \
    {                                                 \
        ::ramd::set_error(__FILE__, __LINE__, (msg)); \
        return (code);                                \
    } while(0)

#define RAMD_HIP(expr)                                                                      \
    do                                                                                      \
    {                                                                                       \
        hipError_t e_ = (expr);                                                             \
        if(e_ != hipSuccess)                                                                \
        {                                                                                   \
            ::ramd::set_error(__FILE__, __LINE__,                                           \
                              std::string(#expr) + " -> " + hipGetErrorString(e_));         \
            return RAMD_ERR_HIP;                                                            \
        }                                                                                   \
    } while(0)

#define RAMD_TRY(expr)           \
    do                           \
    {                            \
        int s_ = (expr);         \
        if(s_ != RAMD_OK)        \
            return s_;           \
    } while(0)

// ---------------------------------------------------------------- backend state
// One host thread drives the backend (as in the reference, SURVEY.md §8b "Threading").
// Three streams mirror the reference's default / interior / ghost choreography
// (src/base/hip/backend_hip.cpp:358-410); `cur` is what every op launches on.
#ifndef RAMD_REDUCE_BLOCKS
#define RAMD_REDUCE_BLOCKS 8192
#endif
constexpr int kReduceBlocks  = RAMD_REDUCE_BLOCKS; // partial sums per reduction launch (32 workgroups per CU: tools/membench.hip)
constexpr int kScalarSlots   = 512; // doubles per scalar record (RAMD_NSCALARS)
constexpr int kScalarRecords = 8; // ring of records in host-mapped memory

struct Backend
{
    bool        initialized = false;
    int         device      = 0;
    int         num_cu      = 256;
    int         num_xcd     = 8;
    hipStream_t stream_default  = nullptr;
    hipStream_t stream_interior = nullptr;
    hipStream_t stream_ghost    = nullptr;
    hipStream_t cur             = nullptr;
    // reduction workspace
    double*       d_partials = nullptr; // [kScalarSlots][kReduceBlocks]
    unsigned int* d_ticket   = nullptr; // arrival counters (one per concurrent reduction)
    double*       d_scalars  = nullptr; // device-resident scalar records
    double*       h_scalars  = nullptr; // pinned host mirror (async copies land here)
    hipEvent_t    ev_scalar[kScalarRecords] = {};
    char          arch[64] = {0};
};

Backend& backend();
int      ensure_init();

// ---------------------------------------------------------------- allocation
// All device arrays are over-allocated by kPad bytes so that 16-byte vector loads that start
// inside an array may run past its logical end (never dereferenced for results).
constexpr size_t kPad = 256;

// caching device allocator (backend.hip): every device block of the library goes through these two
hipError_t cached_malloc_bytes(void** p, size_t bytes);
hipError_t cached_free(void* p);
hipError_t cached_malloc_apart(void** p, size_t bytes, const void* other); // in the placement class opposite to `other`'s
int        cached_block_class(const void* p); // 0 / 1, or -1 (small block, not classified)
void       build_mark(const char* what); // RAMD_BUILD_VERBOSE: device-synchronised wall time since the previous mark (stderr)
float      probe_write_pair_ms(void* a, void* b, size_t bytes); // one pass writing both blocks at once (zeros), best of 4
// may a placement measurement take `blocks` more blocks of `bytes` each?  Only while that leaves a sixteenth of the device
// (at least 2 GiB) free: placement is an optimisation and must never be what runs a memory-tight solve out of memory
bool       placement_room(size_t bytes, int blocks);
void       cached_release_all(void);
template <typename X>
inline hipError_t cached_malloc(X** p, size_t bytes)
{
    void*      q = nullptr;
    hipError_t e = cached_malloc_bytes(&q, bytes);
    *p           = static_cast<X*>(q);
    return e;
}

template <typename X>
int dev_alloc(X** p, int64_t n)
{
    *p = nullptr;
    void*  q     = nullptr;
    size_t bytes = (size_t)(n > 0 ? n : 0) * sizeof(X) + kPad;
    RAMD_HIP(cached_malloc_bytes(&q, bytes));
    *p = static_cast<X*>(q);
    return RAMD_OK;
}
template <typename X>
void dev_free(X** p)
{
    if(*p)
        (void)cached_free(*p);
    *p = nullptr;
}

// ---------------------------------------------------------------- launch geometry
constexpr int kBlock = 256;

// grid for bandwidth-bound elementwise kernels: enough workgroups to fill 256 CUs several
// times over, capped so that each thread still streams a few 16-byte packets (grid-stride).
inline int ew_grid(int64_t n_vec_items)
{
    int64_t g   = (n_vec_items + kBlock - 1) / kBlock;
    static int mult = -1; // workgroups per CU (RAMD_EW_GRID_MULT: experiments only)
    if(mult < 0)
    {
        const char* e = getenv("RAMD_EW_GRID_MULT");
        mult          = e ? atoi(e) : 16;
        if(mult < 1)
            mult = 16;
    }
    int64_t cap = (int64_t)backend().num_cu * mult;
    if(g > cap)
        g = cap;
    if(g < 1)
        g = 1;
    return (int)g;
}

} // namespace ramd

// ---------------------------------------------------------------- object layouts (C handles)
struct ramd_vec_s
{
    int     dtype = RAMD_F64;
    int64_t n     = 0;
    void*   d     = nullptr;
};

struct ramd_mat_s
{
    int     dtype  = RAMD_F64;
    int     format = RAMD_CSR;
    int     nrow = 0, ncol = 0;
    int64_t nnz = 0;
    // CSR (also the storage of the LU factors after ilu0_factorize)
    int*  rp  = nullptr;
    int*  ci  = nullptr;
    void* val = nullptr;
    // ELL part (ELL and HYB): column-major, ELL_IND(row,el) = el*nrow + row
    int   ell_width = 0;
    int*  ell_col   = nullptr;
    void* ell_val   = nullptr;
    // COO part (COO and HYB tail)
    int64_t coo_nnz = 0;
    int*    coo_row = nullptr;
    int*    coo_col = nullptr;
    void*   coo_val = nullptr;
    // COO row grouping (built once): touched rows and their entry ranges in stable row order
    int   coo_ngroups = 0;
    int*  coo_grow    = nullptr; // [ngroups] row index
    int*  coo_gptr    = nullptr; // [ngroups+1] entry ranges (COO data is row-sorted)
    // triangular-solve analysis (LUAnalyse / LAnalyse / UAnalyse)
    bool  lu_analysed = false;
    bool  l_analysed = false, u_analysed = false;
    bool  l_diag_unit = true, u_diag_unit = false;
    int*  diag_pos = nullptr; // [nrow] position of the first entry with col >= row (ILU0)
    void* tri      = nullptr; // ramd::TriState* (level-ordered solve plans), trisolve.hip
    // workspace of the fused CSR SpMV + <x,y> (spmv.hip)
    int     band_dist = -1; // far-band distance in rows for the band-aware traversal (-1 unknown, 0 none)
    int     shift_rows = -1; // 1: most rows carry the columns of the row before them shifted by one (a stencil: gathers of
                             // consecutive rows fall on consecutive elements of x); 0: not; -1 unknown (csr_analyse_shift)
    // row patterns of the CSR SpMV (spmv.hip, csr_analyse_pattern): rows whose column offsets col - row coincide share a
    // dictionary entry, and the kernel rebuilds the columns from one byte per row instead of reading 4 bytes per entry
    int            pat_state = 0; // 0 unknown, 1 usable, -1 not usable (too many patterns / rows too long)
    bool           pat_off   = false; // ramd_mat_pattern_use(m, 0): the products of this matrix read its columns
    int            pat_len[64] = {0}; // row length of every dictionary entry (host copy)
    // rows that share their column list with the row before them (spmv.hip, csr_analyse_groups): 0 unknown, 1 usable, -1 not
    int            grp_state = 0;
    int*           grp_lead  = nullptr; // [nrow] rp[leader of the row's group] - rp[row]  (0 for a leader)
    unsigned char* grp_need  = nullptr; // [ceil(nnz / 4)] 1: this 16-byte packet of column indices holds a leader's entries
    // x tiles in LDS (spmv.hip, csr_analyse_xl): 0 unknown, 1 usable, -1 not usable
    int            xl_state = 0;
    int*           xl_dict  = nullptr; // [pat_n * pat_w] LDS element of x for thread 0, per dictionary entry and slot
    int            xl_segs[2 + 3 * 8] = {0}; // XlSegs (matrix_impl.hpp), kept as plain ints here
    int            pat_n = 0, pat_w = 0; // dictionary entries, padded row length
    unsigned char* pat_id   = nullptr; // [nrow]
    int*           pat_dict = nullptr; // [pat_n * pat_w] column offsets in storage order
    int*           blk_rp   = nullptr; // [ceil(nrow / 256) + 1] row offsets of the 256-row blocks (k_csr_pat2: a compact, cache-resident copy)
    int*           wav_rp   = nullptr; // [ceil(nrow / 64) + 1] row offsets of the 64-row groups (k_csr_wr: the same for a wave's rows)
    int            blk_span = 0; // most entries a 256-row block stages (from its 4-aligned start); 0: not measured yet
    double* dot_part1 = nullptr; // [dot_nblk] per-workgroup partials
    int     dot_nblk  = 0;
};
