// rocalution/io.hpp -- LocalMatrix::ReadFileMTX with the reference's MatrixMarket semantics
// (src/base/host/host_io.cpp:51-320 reader, src/base/local_matrix.cpp:1269-1326 front end):
//   * banner "%%MatrixMarket matrix coordinate {real|integer|pattern} {general|symmetric|hermitian}"
//     (case-insensitive); anything else is rejected;
//   * 1-based indices -> 0-based; pattern entries get the value 1;
//   * symmetric / hermitian storage is mirrored without duplicating the diagonal (:218-272);
//   * the result is CSR with every row sorted by column (ReadFileMTX always calls Sort()).
// Also: WriteFileMTX (host_io.cpp:367-395: "general" coordinate file, CSR order, %0.12g values),
// ReadFileCSR / WriteFileCSR (binary "#rocALUTION binary csr file", host_io.cpp:497-609, :3236-3289:
// int version, then 64-bit sizes (32-bit before version 30000), 32-bit row offsets while
// nnz < INT_MAX, int32 columns, values ALWAYS stored as double), and the vector files of
// host_vector.cpp:415-632 (ASCII: one value per line, scientific; binary: header, version, size, doubles).
// Host-side setup code; the matrix lands in host storage and is moved by MoveToAccelerator().
#pragma once

#include <cctype>
#include <fstream>
#include <limits>
#include <numeric>

#include "base.hpp"

namespace rocalution
{

template <typename ValueType>
bool LocalMatrix<ValueType>::ReadFileMTX(const std::string& filename)
{
    say("ReadFileMTX: filename=", filename, "; reading...");
    std::ifstream f(filename.c_str(), std::ios::binary | std::ios::ate);
    if(!f)
    {
        say("ReadFileMTX: cannot open file ", filename);
        RAMD_DIE();
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
        say("ReadFileMTX: invalid matrix market banner");
        RAMD_DIE();
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
        say("ReadFileMTX: invalid matrix market banner");
        RAMD_DIE();
    }
    if(is_complex)
    {
        say("ReadFileMTX: complex matrices are not provided by this backend");
        RAMD_DIE();
    }
    // skip comments, read "m n nnz"
    long long nrow = 0, ncol = 0, nnz = 0;
    while(true)
    {
        if(!next_line(line))
        {
            say("ReadFileMTX: invalid matrix data");
            RAMD_DIE();
        }
        if(!line.empty() && line[0] == '%')
            continue;
        if(sscanf(line.c_str(), "%lld %lld %lld", &nrow, &ncol, &nnz) == 3)
            break;
    }
    // sizes must be usable as 32-bit row / column indices (and a negative count must not become a huge allocation)
    if(nrow < 0 || ncol < 0 || nnz < 0 || nrow > 2147483646LL || ncol > 2147483646LL)
    {
        say("ReadFileMTX: invalid matrix data (sizes ", nrow, " x ", ncol, ", ", nnz, " entries)");
        RAMD_DIE();
    }
    // every entry takes at least "1 1\n" in the file: a header count the rest of the file cannot hold is a damaged file, not
    // an allocation request
    if(nnz > (long long)(end - p) / 4 + 1)
    {
        say("ReadFileMTX: invalid matrix data (", nnz, " entries announced, ", (long long)(end - p), " bytes left in the file)");
        RAMD_DIE();
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
            say("ReadFileMTX: invalid matrix data");
            RAMD_DIE();
        }
        q      = e;
        long c = strtol(q, &e, 10);
        bool bad = (e == q); // the reference reads every entry with fscanf(...) != 3 -> error (host_io.cpp:200-216)
        q        = e;
        double v = 1.0;
        if(!is_pattern)
        {
            v   = strtod(q, &e);
            bad = bad || (e == q);
            q   = e;
        }
        if(bad || r < 1 || r > nrow || c < 1 || c > ncol)
        {
            say("ReadFileMTX: invalid matrix data (entry ", i + 1, ")");
            RAMD_DIE();
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
    say("ReadFileMTX: filename=", filename, "; done");
    return true;
}

// version stamp written into binary files: that of the reference this API mirrors (3.2.0)
#ifndef __ROCALUTION_VER
#define __ROCALUTION_VER 30200
#endif

namespace detail
{
// CSR arrays of a matrix wherever it lives (a converted matrix is cloned and converted back first)
template <typename ValueType>
inline void csr_on_host(const LocalMatrix<ValueType>& A, std::vector<PtrType>& rp, std::vector<int>& ci,
                        std::vector<ValueType>& va)
{
    const LocalMatrix<ValueType>* src = &A;
    LocalMatrix<ValueType>        tmp;
    if(A.GetFormat() != CSR)
    {
        tmp.CloneFrom(A);
        tmp.ConvertTo(CSR);
        src = &tmp;
    }
    rp.assign((size_t)src->GetM() + 1, 0);
    ci.assign((size_t)src->GetNnz(), 0);
    va.assign((size_t)src->GetNnz(), ValueType(0));
    if(src->GetM() > 0)
        src->CopyToCSR(rp.data(), ci.data(), va.data());
}
} // namespace detail

template <typename ValueType>
bool LocalMatrix<ValueType>::WriteFileMTX(const std::string& filename) const
{
    say("WriteFileMTX: filename=", filename, "; writing...");
    std::vector<PtrType>   rp;
    std::vector<int>       ci;
    std::vector<ValueType> va;
    detail::csr_on_host(*this, rp, ci, va);
    FILE* file = fopen(filename.c_str(), "w");
    if(!file)
    {
        say("WriteFileMTX: cannot open file ", filename);
        RAMD_DIE();
    }
    fprintf(file, "%%%%MatrixMarket matrix coordinate real general\n");
    fprintf(file, "%d %d %lld\n", (int)this->GetM(), (int)this->GetN(), (long long)ci.size());
    for(size_t i = 0; i + 1 < rp.size(); ++i)
        for(PtrType j = rp[i]; j < rp[i + 1]; ++j)
            fprintf(file, "%d %d %0.12g\n", (int)i + 1, ci[j] + 1, (double)va[j]);
    fclose(file);
    say("WriteFileMTX: filename=", filename, "; done");
    return true;
}

template <typename ValueType>
bool LocalMatrix<ValueType>::WriteFileCSR(const std::string& filename) const
{
    say("WriteFileCSR: filename=", filename, "; writing...");
    std::vector<PtrType>   rp;
    std::vector<int>       ci;
    std::vector<ValueType> va;
    detail::csr_on_host(*this, rp, ci, va);
    std::ofstream out(filename.c_str(), std::ios::out | std::ios::binary);
    if(!out.is_open())
    {
        say("WriteFileCSR: cannot open file ", filename);
        RAMD_DIE();
    }
    out << "#rocALUTION binary csr file" << std::endl;
    const int     version = __ROCALUTION_VER;
    const int64_t nrow = this->GetM(), ncol = this->GetN(), nnz = (int64_t)ci.size();
    out.write((const char*)&version, sizeof(int));
    out.write((const char*)&nrow, sizeof(int64_t));
    out.write((const char*)&ncol, sizeof(int64_t));
    out.write((const char*)&nnz, sizeof(int64_t));
    static_assert(sizeof(PtrType) == 4, "row offsets are 32-bit here (nnz < INT_MAX)");
    out.write((const char*)rp.data(), sizeof(int) * rp.size());
    out.write((const char*)ci.data(), sizeof(int) * ci.size());
    std::vector<double> dv(va.begin(), va.end()); // values are always stored in double precision
    out.write((const char*)dv.data(), sizeof(double) * dv.size());
    if(!out)
    {
        say("WriteFileCSR: filename=", filename, "; could not write to file");
        RAMD_DIE();
    }
    out.close();
    say("WriteFileCSR: filename=", filename, "; done");
    return true;
}

template <typename ValueType>
bool LocalMatrix<ValueType>::ReadFileCSR(const std::string& filename)
{
    say("ReadFileCSR: filename=", filename, "; reading...");
    std::ifstream in(filename.c_str(), std::ios::in | std::ios::binary);
    if(!in.is_open())
    {
        say("ReadFileCSR: cannot open file ", filename);
        RAMD_DIE();
    }
    std::string header;
    std::getline(in, header);
    if(header != "#rocALUTION binary csr file")
    {
        say("ReadFileCSR: invalid rocALUTION matrix header");
        RAMD_DIE();
    }
    int     version = 0;
    int64_t nrow = 0, ncol = 0, nnz = 0;
    in.read((char*)&version, sizeof(int));
    if(version < 30000) // 32-bit sizes before 3.0.0
    {
        int s32[3] = {0, 0, 0};
        in.read((char*)s32, sizeof(s32));
        nrow = s32[0];
        ncol = s32[1];
        nnz  = s32[2];
    }
    else
    {
        in.read((char*)&nrow, sizeof(int64_t));
        in.read((char*)&ncol, sizeof(int64_t));
        in.read((char*)&nnz, sizeof(int64_t));
    }
    if(!in || nrow < 0 || ncol < 0 || nnz < 0)
    {
        say("ReadFileCSR: invalid matrix data");
        RAMD_DIE();
    }
    if(version >= 30000 && nnz >= std::numeric_limits<int>::max())
    {
        say("ReadFileCSR: cannot read 64 bit sparsity pattern into 32 bit structure");
        RAMD_DIE();
    }
    std::vector<PtrType>   rp((size_t)nrow + 1);
    std::vector<int>       ci((size_t)nnz);
    std::vector<ValueType> va((size_t)nnz);
    in.read((char*)rp.data(), sizeof(int) * rp.size());
    in.read((char*)ci.data(), sizeof(int) * ci.size());
    {
        std::vector<double> dv((size_t)nnz);
        in.read((char*)dv.data(), sizeof(double) * dv.size());
        for(size_t i = 0; i < dv.size(); ++i)
            va[i] = static_cast<ValueType>(dv[i]);
    }
    if(!in)
    {
        say("ReadFileCSR: invalid matrix data");
        RAMD_DIE();
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
    say("ReadFileCSR: filename=", filename, "; done");
    return true;
}

// ---- vectors
template <typename ValueType>
void LocalVector<ValueType>::ReadFileASCII(const std::string& filename)
{
    say("ReadFileASCII: filename=", filename, "; reading...");
    std::ifstream file(filename.c_str(), std::ifstream::in);
    if(!file.is_open())
    {
        say("Can not open vector file [read]:", filename);
        RAMD_DIE();
    }
    int64_t     n = 0; // the size is the number of LINES (host_vector.cpp:433-437)
    std::string line;
    while(std::getline(file, line))
        ++n;
    std::vector<ValueType> data((size_t)n, ValueType(0));
    file.clear();
    file.seekg(0, std::ios_base::beg);
    for(int64_t i = 0; i < n; ++i)
        file >> data[(size_t)i];
    file.close();
    this->Allocate(filename, n);
    this->CopyFromHostData(data.data());
    say("ReadFileASCII: filename=", filename, "; done");
}

template <typename ValueType>
void LocalVector<ValueType>::WriteFileASCII(const std::string& filename) const
{
    say("WriteFileASCII: filename=", filename, "; writing...");
    std::vector<ValueType> data((size_t)this->GetSize());
    this->CopyToHostData(data.data());
    std::ofstream file(filename.c_str(), std::ifstream::out);
    if(!file.is_open())
    {
        say("Can not open vector file [write]:", filename);
        RAMD_DIE();
    }
    file.setf(std::ios::scientific);
    for(size_t i = 0; i < data.size(); ++i)
        file << data[i] << std::endl;
    file.close();
    say("WriteFileASCII: filename=", filename, "; done");
}

template <typename ValueType>
void LocalVector<ValueType>::ReadFileBinary(const std::string& filename)
{
    say("ReadFileBinary: filename=", filename, "; reading...");
    std::ifstream in(filename.c_str(), std::ios::in | std::ios::binary);
    if(!in.is_open())
    {
        say("ReadFileBinary: filename=", filename, "; cannot open file");
        RAMD_DIE();
    }
    std::string header;
    std::getline(in, header);
    if(header != "#rocALUTION binary vector file")
    {
        say("ReadFileBinary: filename=", filename, " is not a rocALUTION vector");
        RAMD_DIE();
    }
    int     version = 0;
    int64_t n       = 0;
    in.read((char*)&version, sizeof(int));
    if(version < 30000)
    {
        int size32 = 0;
        in.read((char*)&size32, sizeof(int));
        n = size32;
    }
    else
        in.read((char*)&n, sizeof(int64_t));
    if(!in || n < 0)
    {
        say("ReadFileBinary: filename=", filename, "; could not read from file");
        RAMD_DIE();
    }
    std::vector<ValueType> data((size_t)n);
    if(std::is_floating_point<ValueType>::value) // real data is always stored in double precision
    {
        std::vector<double> tmp((size_t)n);
        in.read((char*)tmp.data(), sizeof(double) * tmp.size());
        for(size_t i = 0; i < tmp.size(); ++i)
            data[i] = static_cast<ValueType>(tmp[i]);
    }
    else
        in.read((char*)data.data(), sizeof(ValueType) * data.size());
    if(!in)
    {
        say("ReadFileBinary: filename=", filename, "; could not read from file");
        RAMD_DIE();
    }
    this->Allocate(filename, n);
    this->CopyFromHostData(data.data());
    say("ReadFileBinary: filename=", filename, "; done");
}

template <typename ValueType>
void LocalVector<ValueType>::WriteFileBinary(const std::string& filename) const
{
    say("WriteFileBinary: filename=", filename, "; writing...");
    const int64_t          n = this->GetSize();
    std::vector<ValueType> data((size_t)n);
    this->CopyToHostData(data.data());
    std::ofstream out(filename.c_str(), std::ios::out | std::ios::binary);
    if(!out.is_open())
    {
        say("WriteFileBinary: filename=", filename, "; cannot open file");
        RAMD_DIE();
    }
    out << "#rocALUTION binary vector file" << std::endl;
    const int version = __ROCALUTION_VER;
    out.write((const char*)&version, sizeof(int));
    out.write((const char*)&n, sizeof(int64_t));
    if(std::is_floating_point<ValueType>::value)
    {
        std::vector<double> tmp(data.begin(), data.end());
        out.write((const char*)tmp.data(), sizeof(double) * tmp.size());
    }
    else
        out.write((const char*)data.data(), sizeof(ValueType) * data.size());
    if(!out)
    {
        say("WriteFileBinary: filename=", filename, "; could not write to file");
        RAMD_DIE();
    }
    out.close();
    say("WriteFileBinary: filename=", filename, "; done");
}

} // namespace rocalution
