// rocalution/io.hpp -- LocalMatrix::ReadFileMTX with the reference's MatrixMarket semantics
// (src/base/host/host_io.cpp:51-320 reader, src/base/local_matrix.cpp:1269-1326 front end):
//   * banner "%%MatrixMarket matrix coordinate {real|integer|pattern} {general|symmetric|hermitian}"
//     (case-insensitive); anything else is rejected;
//   * 1-based indices -> 0-based; pattern entries get the value 1;
//   * symmetric / hermitian storage is mirrored without duplicating the diagonal (:218-272);
//   * the result is CSR with every row sorted by column (ReadFileMTX always calls Sort()).
// Host-side setup code; the matrix lands in host storage and is moved by MoveToAccelerator().
#pragma once

#include <cctype>
#include <numeric>

#include "base.hpp"

namespace rocalution
{

template <typename ValueType>
bool LocalMatrix<ValueType>::ReadFileMTX(const std::string& filename)
{
    LOG_INFO("ReadFileMTX: filename=" << filename << "; reading...");
    std::ifstream f(filename.c_str(), std::ios::binary | std::ios::ate);
    if(!f)
    {
        LOG_INFO("ReadFileMTX: cannot open file " << filename);
        FATAL_ERROR(__FILE__, __LINE__);
    }
    std::string buf((size_t)f.tellg(), '\0');
    f.seekg(0);
    f.read(&buf[0], (std::streamsize)buf.size());
    const char* p   = buf.c_str();
    const char* end = p + buf.size();

    auto next_line = [&](std::string& line) -> bool {
        if(p >= end)
            return false;
        const char* e = (const char*)memchr(p, '\n', (size_t)(end - p));
        if(!e)
            e = end;
        line.assign(p, e);
        p = (e < end) ? e + 1 : end;
        return true;
    };
    std::string line;
    if(!next_line(line))
    {
        LOG_INFO("ReadFileMTX: invalid matrix market banner");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    std::istringstream bs(line);
    std::string        banner, mtx, array_type, matrix_type, storage_type;
    bs >> banner >> mtx >> array_type >> matrix_type >> storage_type;
    auto lower = [](std::string& s) {
        for(char& c : s)
            c = (char)tolower((unsigned char)c);
    };
    lower(mtx);
    lower(array_type);
    lower(matrix_type);
    lower(storage_type);
    const bool is_pattern = matrix_type.compare(0, 7, "pattern") == 0;
    const bool is_complex = matrix_type.compare(0, 7, "complex") == 0;
    const bool val_ok     = matrix_type.compare(0, 4, "real") == 0
                        || matrix_type.compare(0, 7, "integer") == 0 || is_pattern;
    const bool general = storage_type.compare(0, 7, "general") == 0;
    const bool sym_ok  = general || storage_type.compare(0, 9, "symmetric") == 0
                        || storage_type.compare(0, 9, "hermitian") == 0;
    if(banner.compare(0, 14, "%%MatrixMarket") != 0 || mtx.compare(0, 6, "matrix") != 0
       || array_type.compare(0, 10, "coordinate") != 0 || !sym_ok || (!val_ok && !is_complex))
    {
        LOG_INFO("ReadFileMTX: invalid matrix market banner");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    if(is_complex)
    {
        LOG_INFO("ReadFileMTX: complex matrices are not provided by this backend");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    // skip comments, read "m n nnz"
    long long nrow = 0, ncol = 0, nnz = 0;
    while(true)
    {
        if(!next_line(line))
        {
            LOG_INFO("ReadFileMTX: invalid matrix data");
            FATAL_ERROR(__FILE__, __LINE__);
        }
        if(!line.empty() && line[0] == '%')
            continue;
        if(sscanf(line.c_str(), "%lld %lld %lld", &nrow, &ncol, &nnz) == 3)
            break;
    }
    std::vector<int>       row((size_t)nnz), col((size_t)nnz);
    std::vector<ValueType> val((size_t)nnz);
    char*                  q = const_cast<char*>(p);
    for(long long i = 0; i < nnz; ++i)
    {
        char* e = NULL;
        long  r = strtol(q, &e, 10);
        if(e == q)
        {
            LOG_INFO("ReadFileMTX: invalid matrix data");
            FATAL_ERROR(__FILE__, __LINE__);
        }
        q      = e;
        long c = strtol(q, &e, 10);
        q      = e;
        double v = 1.0;
        if(!is_pattern)
        {
            v = strtod(q, &e);
            q = e;
        }
        row[(size_t)i] = (int)r - 1;
        col[(size_t)i] = (int)c - 1;
        val[(size_t)i] = (ValueType)v;
    }
    if(!general) // mirror the stored triangle, diagonal entries once
    {
        std::vector<int>       r2, c2;
        std::vector<ValueType> v2;
        r2.reserve((size_t)nnz * 2);
        c2.reserve((size_t)nnz * 2);
        v2.reserve((size_t)nnz * 2);
        for(long long i = 0; i < nnz; ++i)
        {
            r2.push_back(row[(size_t)i]);
            c2.push_back(col[(size_t)i]);
            v2.push_back(val[(size_t)i]);
            if(row[(size_t)i] != col[(size_t)i])
            {
                r2.push_back(col[(size_t)i]);
                c2.push_back(row[(size_t)i]);
                v2.push_back(val[(size_t)i]);
            }
        }
        row.swap(r2);
        col.swap(c2);
        val.swap(v2);
        nnz = (long long)row.size();
    }
    // COO -> CSR, rows sorted by column (stable: duplicates keep their file order)
    std::vector<PtrType> rp((size_t)nrow + 1, 0);
    for(long long i = 0; i < nnz; ++i)
        ++rp[(size_t)row[(size_t)i] + 1];
    for(long long i = 0; i < nrow; ++i)
        rp[(size_t)i + 1] += rp[(size_t)i];
    std::vector<PtrType>   cur(rp.begin(), rp.end() - 1);
    std::vector<int>       ci((size_t)nnz);
    std::vector<ValueType> va((size_t)nnz);
    for(long long i = 0; i < nnz; ++i)
    {
        PtrType d = cur[(size_t)row[(size_t)i]]++;
        ci[(size_t)d] = col[(size_t)i];
        va[(size_t)d] = val[(size_t)i];
    }
    std::vector<int> idx;
    for(long long r = 0; r < nrow; ++r)
    {
        const PtrType b = rp[(size_t)r], e = rp[(size_t)r + 1];
        if(std::is_sorted(ci.begin() + b, ci.begin() + e))
            continue;
        idx.resize((size_t)(e - b));
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(),
                         [&](int a, int c) { return ci[(size_t)b + a] < ci[(size_t)b + c]; });
        std::vector<int>       tc(idx.size());
        std::vector<ValueType> tv(idx.size());
        for(size_t k = 0; k < idx.size(); ++k)
        {
            tc[k] = ci[(size_t)b + idx[k]];
            tv[k] = va[(size_t)b + idx[k]];
        }
        std::copy(tc.begin(), tc.end(), ci.begin() + b);
        std::copy(tv.begin(), tv.end(), va.begin() + b);
    }
    const bool was_accel = this->on_accel_;
    this->Clear();
    if(was_accel)
        RAMD_CHECK(ramd_mat_clear(this->dev_));
    this->on_accel_ = false;
    this->name_     = filename;
    this->h_rp_.swap(rp);
    this->h_ci_.swap(ci);
    this->h_val_.swap(va);
    this->h_nrow_ = nrow;
    this->h_ncol_ = ncol;
    if(was_accel)
        this->MoveToAccelerator();
    LOG_INFO("ReadFileMTX: filename=" << filename << "; done");
    return true;
}

} // namespace rocalution
