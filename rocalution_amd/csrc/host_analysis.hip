// host_analysis.hip -- setup steps that the reference itself runs SERIALLY ON THE HOST in both of
// its backends.  MultiColoring: the reference HIP backend copies row_offset/col to the host and runs
// the same greedy loop as the host backend (src/base/hip/hip_matrix_csr.cpp:3915-4060 ==
// src/base/host/host_matrix_csr.cpp:2469-2599).  The colour ORDER decides the permutation and with it
// every MC-SGS result, so this stays a sequential first-fit sweep with identical tie-breaking:
//   natural row order; neighbours = row entries AND column (CSC) entries; colours numbered from 1;
//   perm[i] = offset[colour(i)]++  (stable inside a colour).
#include "common.hpp"
#include "matrix_impl.hpp"

#include <vector>

using namespace ramd;

extern "C" int ramd_mat_multicoloring(ramd_mat_t m, int* num_colors, int* size_colors, ramd_vec_t perm)
{
    if(!m || !num_colors || !size_colors || !perm)
        RAMD_FAIL(RAMD_ERR_ARG, "MultiColoring: null argument");
    if(m->format != RAMD_CSR)
        return RAMD_ERR_UNSUPPORTED;
    if(perm->dtype != RAMD_I32)
        RAMD_FAIL(RAMD_ERR_ARG, "MultiColoring: permutation must be an int32 vector");
    if(m->nrow != m->ncol)
        RAMD_FAIL(RAMD_ERR_ARG, "MultiColoring: square matrix expected");
    // device sweep when it applies (structurally symmetric pattern, <= 64 colours): same colours;
    // RAMD_COLORING=host forces the serial sweep below
    static const bool force_host = [] {
        const char* e = getenv("RAMD_COLORING");
        return e && std::string(e) == "host";
    }();
    if(!force_host)
    {
        int s = multicoloring_device(m, num_colors, size_colors, perm);
        if(s != RAMD_ERR_UNSUPPORTED)
            return s;
    }
    Backend&  b   = backend();
    const int n   = m->nrow;
    const int64_t nnz = m->nnz;
    std::vector<int> rp((size_t)n + 1), ci((size_t)nnz);
    RAMD_HIP(hipMemcpyAsync(rp.data(), m->rp, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost, b.cur));
    if(nnz > 0)
        RAMD_HIP(hipMemcpyAsync(ci.data(), m->ci, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost, b.cur));
    RAMD_HIP(hipStreamSynchronize(b.cur));

    // column view (who has an entry in my column)
    std::vector<int> csc_ptr((size_t)n + 1, 0), csc_ind((size_t)nnz);
    for(int64_t i = 0; i < nnz; ++i)
        csc_ptr[ci[i] + 1] += 1;
    for(int i = 1; i <= n; ++i)
        csc_ptr[i] += csc_ptr[i - 1];
    {
        std::vector<int> cur(csc_ptr.begin(), csc_ptr.end() - 1);
        for(int i = 0; i < n; ++i)
            for(int k = rp[i]; k < rp[i + 1]; ++k)
                csc_ind[cur[ci[k]]++] = i;
    }

    std::vector<int>  color((size_t)n, 0);
    std::vector<char> used;
    int               ncol = 0;
    for(int ai = 0; ai < n; ++ai)
    {
        color[ai] = 1;
        used.assign((size_t)ncol + 2, 0);
        for(int aj = rp[ai]; aj < rp[ai + 1]; ++aj)
            if(ai != ci[aj])
                used[color[ci[aj]]] = 1;
        for(int aj = csc_ptr[ai]; aj < csc_ptr[ai + 1]; ++aj)
            if(ai != csc_ind[aj])
                used[color[csc_ind[aj]]] = 1;
        const int count = rp[ai + 1] - rp[ai] + csc_ptr[ai + 1] - csc_ptr[ai];
        for(int aj = 0; aj < count; ++aj)
        {
            if(used[color[ai]])
                ++color[ai];
            else
                break;
        }
        if(color[ai] > ncol)
            ncol = color[ai];
    }
    std::vector<int> offsets((size_t)std::max(ncol, 1), 0);
    for(int i = 0; i < ncol; ++i)
        size_colors[i] = 0;
    for(int i = 0; i < n; ++i)
        ++size_colors[color[i] - 1];
    int total = 0;
    for(int i = 1; i < ncol; ++i)
    {
        total += size_colors[i - 1];
        offsets[i] = total;
    }
    std::vector<int> hperm((size_t)n);
    for(int i = 0; i < n; ++i)
        hperm[i] = offsets[color[i] - 1]++;
    *num_colors = ncol;
    RAMD_TRY(ramd_vec_allocate(perm, n));
    if(n > 0)
        RAMD_TRY(ramd_vec_copy_from_host(perm, hperm.data()));
    return RAMD_OK;
}
