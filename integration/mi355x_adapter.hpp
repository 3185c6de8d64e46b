// mi355x_adapter.hpp -- the binding a rocALUTION maintainer adds to plug librocalution_amd.so in as an accelerator backend
// (INTEGRATION.md section B).  Two adapter classes over the reference's own interfaces -- AcceleratorVector<ValueType>
// (src/base/base_vector.hpp:216-232) and AcceleratorMatrix<ValueType> (src/base/base_matrix.hpp:839-857) -- whose virtual
// methods forward to the C ABI of include/rocalution_amd.h; nothing else in the reference changes but the two factory branches
// shown at the end.  This file is COMPILED against the reference's headers by tests/test_cpu_host.py
// (test_integration_adapter_compiles_against_the_reference_interfaces: g++ -fsyntax-only, every pure virtual overridden, both
// value types instantiated) wherever /root/reference is present; nothing of the reference is copied here or travels.
//
// Conventions:
//   * an operation the ABI does not provide on the accelerator is answered the way the reference's HIP backend answers an
//     operation it lacks: optional matrix operations return false (LocalMatrix then falls back to the host,
//     src/base/local_matrix.cpp:2299-2340), mandatory vector operations the Krylov path never calls stop with a message
//     (RAMD_ADAPTER_UNSUPPORTED), as FATAL_ERROR does in src/base/hip/hip_vector.cpp;
//   * host objects are read through their PUBLIC interface (CopyToHostData / CopyToCSR): a maintainer would add the two
//     adapter classes to the friend lists of HostVector / HostMatrixCSR (host_vector.hpp:164-200) and pass vec_ / mat_ directly.
#pragma once

#include <rocalution_amd.h>

#include "base/base_matrix.hpp"
#include "base/base_vector.hpp"
#include "base/host/host_matrix_csr.hpp"
#include "base/host/host_vector.hpp"
#include "base/matrix_formats.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define RAMD_ADAPTER_CHECK(call)                                                                              \
    do                                                                                                        \
    {                                                                                                         \
        const int ramd_status_ = (call);                                                                      \
        if(ramd_status_ != RAMD_OK)                                                                           \
        {                                                                                                     \
            std::fprintf(stderr, "MI355X backend: %s -> %d (%s)\n", #call, ramd_status_, ramd_last_error());  \
            std::abort();                                                                                     \
        }                                                                                                     \
    } while(0)
#define RAMD_ADAPTER_UNSUPPORTED(what)                                                             \
    do                                                                                             \
    {                                                                                              \
        std::fprintf(stderr, "MI355X backend: %s is not provided on the accelerator\n", what);     \
        std::abort();                                                                              \
    } while(0)

namespace rocalution
{

template <typename ValueType>
struct MI355XType;
template <>
struct MI355XType<double>
{
    static constexpr int id = RAMD_F64;
};
template <>
struct MI355XType<float>
{
    static constexpr int id = RAMD_F32;
};
template <>
struct MI355XType<int>
{
    static constexpr int id = RAMD_I32;
};

template <typename ValueType>
class MI355XAcceleratorMatrix;

template <typename ValueType>
class MI355XAcceleratorVector : public AcceleratorVector<ValueType>
{
public:
    explicit MI355XAcceleratorVector(const Rocalution_Backend_Descriptor& local_backend)
    {
        this->set_backend(local_backend);
        RAMD_ADAPTER_CHECK(ramd_vec_create(MI355XType<ValueType>::id, &this->h_));
    }
    virtual ~MI355XAcceleratorVector()
    {
        ramd_vec_destroy(this->h_);
    }
    ramd_vec_t handle(void) const
    {
        return this->h_;
    }
    // the handle behind any vector the reference hands to a method of this backend (it only ever mixes objects of one backend)
    template <typename T>
    static ramd_vec_t of(const BaseVector<T>& v)
    {
        const MI355XAcceleratorVector<T>* c = dynamic_cast<const MI355XAcceleratorVector<T>*>(&v);
        if(c == NULL)
            RAMD_ADAPTER_UNSUPPORTED("an operand that does not live on the MI355X backend");
        return c->handle();
    }

    virtual void Info(void) const
    {
        char buf[256];
        ramd_info(buf, (int)sizeof(buf));
        std::printf("MI355XAcceleratorVector<ValueType>, %s\n", buf);
    }
    virtual void Allocate(int64_t n) // base_vector.hpp:63
    {
        RAMD_ADAPTER_CHECK(ramd_vec_allocate(this->h_, n));
        this->size_ = n;
    }
    virtual void SetDataPtr(ValueType**, int64_t)
    {
        RAMD_ADAPTER_UNSUPPORTED("SetDataPtr (the backend owns its blocks: placement classes)");
    }
    virtual void LeaveDataPtr(ValueType**)
    {
        RAMD_ADAPTER_UNSUPPORTED("LeaveDataPtr");
    }
    virtual void Clear(void)
    {
        RAMD_ADAPTER_CHECK(ramd_vec_clear(this->h_));
        this->size_ = 0;
    }
    virtual void Zeros(void)
    {
        RAMD_ADAPTER_CHECK(ramd_vec_zeros(this->h_));
    }
    virtual void Ones(void)
    {
        RAMD_ADAPTER_CHECK(ramd_vec_ones(this->h_));
    }
    virtual void SetValues(ValueType val)
    {
        RAMD_ADAPTER_CHECK(ramd_vec_set_values(this->h_, (double)val));
    }
    virtual void Permute(const BaseVector<int>& permutation) // :80 -- in place: through a copy
    {
        MI355XAcceleratorVector<ValueType> tmp(this->local_backend_);
        tmp.CopyFrom(*this);
        RAMD_ADAPTER_CHECK(ramd_vec_copy_from_permute(this->h_, tmp.handle(), of(permutation)));
    }
    virtual void PermuteBackward(const BaseVector<int>& permutation)
    {
        MI355XAcceleratorVector<ValueType> tmp(this->local_backend_);
        tmp.CopyFrom(*this);
        RAMD_ADAPTER_CHECK(ramd_vec_copy_from_permute_backward(this->h_, tmp.handle(), of(permutation)));
    }
    virtual void CopyFrom(const BaseVector<ValueType>& vec) // :85
    {
        const HostVector<ValueType>* host = dynamic_cast<const HostVector<ValueType>*>(&vec);
        if(host != NULL)
        {
            this->CopyFromHost(*host);
            return;
        }
        RAMD_ADAPTER_CHECK(ramd_vec_copy_from(this->h_, of(vec)));
        this->size_ = vec.GetSize();
    }
    virtual void CopyTo(BaseVector<ValueType>* vec) const // :93
    {
        HostVector<ValueType>* host = dynamic_cast<HostVector<ValueType>*>(vec);
        if(host != NULL)
        {
            this->CopyToHost(host);
            return;
        }
        vec->CopyFrom(*this);
    }
    virtual void CopyFrom(const BaseVector<ValueType>& src, int64_t src_offset, int64_t dst_offset, int64_t size) // :98
    {
        RAMD_ADAPTER_CHECK(ramd_vec_copy_from_offset(this->h_, of(src), src_offset, dst_offset, size));
    }
    virtual void CopyFromPermute(const BaseVector<ValueType>& src, const BaseVector<int>& permutation) // :105
    {
        RAMD_ADAPTER_CHECK(ramd_vec_copy_from_permute(this->h_, of(src), of(permutation)));
    }
    virtual void CopyFromPermuteBackward(const BaseVector<ValueType>& src, const BaseVector<int>& permutation) // :109
    {
        RAMD_ADAPTER_CHECK(ramd_vec_copy_from_permute_backward(this->h_, of(src), of(permutation)));
    }
    virtual void CopyFromHostData(const ValueType* data) // :114
    {
        RAMD_ADAPTER_CHECK(ramd_vec_copy_from_host(this->h_, data));
    }
    virtual void CopyToHostData(ValueType* data) const // :116
    {
        RAMD_ADAPTER_CHECK(ramd_vec_copy_to_host(this->h_, data));
    }
    virtual void CopyFromHost(const HostVector<ValueType>& src) // :224
    {
        if(this->size_ != src.GetSize())
            this->Allocate(src.GetSize());
        if(src.GetSize() > 0)
        {
            std::vector<ValueType> stage((size_t)src.GetSize()); // (a friend of HostVector passes src.vec_ instead)
            src.CopyToHostData(stage.data());
            RAMD_ADAPTER_CHECK(ramd_vec_copy_from_host(this->h_, stage.data()));
        }
    }
    virtual void CopyToHost(HostVector<ValueType>* dst) const // :226
    {
        if(dst->GetSize() != this->size_)
            dst->Allocate(this->size_);
        if(this->size_ > 0)
        {
            std::vector<ValueType> stage((size_t)this->size_);
            RAMD_ADAPTER_CHECK(ramd_vec_copy_to_host(this->h_, stage.data()));
            dst->CopyFromHostData(stage.data());
        }
    }
    virtual void AddScale(const BaseVector<ValueType>& x, ValueType alpha) // :126
    {
        RAMD_ADAPTER_CHECK(ramd_vec_add_scale(this->h_, of(x), (double)alpha));
    }
    virtual void ScaleAdd(ValueType alpha, const BaseVector<ValueType>& x) // :128
    {
        RAMD_ADAPTER_CHECK(ramd_vec_scale_add(this->h_, (double)alpha, of(x)));
    }
    virtual void ScaleAddScale(ValueType alpha, const BaseVector<ValueType>& x, ValueType beta) // :130
    {
        RAMD_ADAPTER_CHECK(ramd_vec_scale_add_scale(this->h_, (double)alpha, of(x), (double)beta));
    }
    virtual void ScaleAddScale(ValueType alpha, const BaseVector<ValueType>& x, ValueType beta, int64_t src_offset,
                               int64_t dst_offset, int64_t size) // :133
    {
        RAMD_ADAPTER_CHECK(ramd_vec_scale_add_scale_offset(this->h_, (double)alpha, of(x), (double)beta, src_offset, dst_offset, size));
    }
    virtual void ScaleAdd2(ValueType alpha, const BaseVector<ValueType>& x, ValueType beta, const BaseVector<ValueType>& y,
                           ValueType gamma) // :142
    {
        RAMD_ADAPTER_CHECK(ramd_vec_scale_add2(this->h_, (double)alpha, of(x), (double)beta, of(y), (double)gamma));
    }
    virtual void Scale(ValueType alpha) // :149
    {
        RAMD_ADAPTER_CHECK(ramd_vec_scale(this->h_, (double)alpha));
    }
    virtual ValueType Dot(const BaseVector<ValueType>& x) const // :151
    {
        double r = 0.0;
        RAMD_ADAPTER_CHECK(ramd_vec_dot(this->h_, of(x), &r));
        return (ValueType)r;
    }
    virtual ValueType DotNonConj(const BaseVector<ValueType>& x) const // :153 (real types: the same)
    {
        return this->Dot(x);
    }
    virtual ValueType Norm(void) const // :155
    {
        double r = 0.0;
        RAMD_ADAPTER_CHECK(ramd_vec_norm(this->h_, &r));
        return (ValueType)r;
    }
    virtual ValueType Reduce(void) const // :157
    {
        double r = 0.0;
        RAMD_ADAPTER_CHECK(ramd_vec_reduce(this->h_, &r));
        return (ValueType)r;
    }
    virtual ValueType InclusiveSum(const BaseVector<ValueType>&)
    {
        RAMD_ADAPTER_UNSUPPORTED("InclusiveSum");
    }
    virtual ValueType ExclusiveSum(const BaseVector<ValueType>&)
    {
        RAMD_ADAPTER_UNSUPPORTED("ExclusiveSum");
    }
    virtual ValueType Asum(void) const // :163
    {
        double r = 0.0;
        RAMD_ADAPTER_CHECK(ramd_vec_asum(this->h_, &r));
        return (ValueType)r;
    }
    virtual int64_t Amax(ValueType& value) const // :165
    {
        double  v   = 0.0;
        int64_t idx = 0;
        RAMD_ADAPTER_CHECK(ramd_vec_amax(this->h_, &v, &idx));
        value = (ValueType)v;
        return idx;
    }
    virtual void PointWiseMult(const BaseVector<ValueType>& x) // :167
    {
        RAMD_ADAPTER_CHECK(ramd_vec_pointwise_mult(this->h_, of(x)));
    }
    virtual void PointWiseMult(const BaseVector<ValueType>& x, const BaseVector<ValueType>& y) // :169
    {
        RAMD_ADAPTER_CHECK(ramd_vec_pointwise_mult2(this->h_, of(x), of(y)));
    }
    virtual void Power(double)
    {
        RAMD_ADAPTER_UNSUPPORTED("Power");
    }
    virtual void GetIndexValues(const BaseVector<int>& index, BaseVector<ValueType>* values) const // :175 (halo pack)
    {
        RAMD_ADAPTER_CHECK(ramd_vec_get_index_values(this->h_, of(index), of(*values)));
    }
    virtual void SetIndexValues(const BaseVector<int>&, const BaseVector<ValueType>&)
    {
        RAMD_ADAPTER_UNSUPPORTED("SetIndexValues");
    }
    virtual void AddIndexValues(const BaseVector<int>&, const BaseVector<ValueType>&)
    {
        RAMD_ADAPTER_UNSUPPORTED("AddIndexValues");
    }
    virtual void GetContinuousValues(int64_t start, int64_t end, ValueType* values) const // :186 (through a host copy)
    {
        std::vector<ValueType> all((size_t)this->size_);
        if(this->size_ > 0)
            RAMD_ADAPTER_CHECK(ramd_vec_copy_to_host(this->h_, all.data()));
        for(int64_t i = start; i < end; ++i)
            values[i - start] = all[(size_t)i];
    }
    virtual void SetContinuousValues(int64_t start, int64_t end, const ValueType* values) // :188
    {
        std::vector<ValueType> all((size_t)this->size_);
        if(this->size_ > 0)
            RAMD_ADAPTER_CHECK(ramd_vec_copy_to_host(this->h_, all.data()));
        for(int64_t i = start; i < end; ++i)
            all[(size_t)i] = values[i - start];
        if(this->size_ > 0)
            RAMD_ADAPTER_CHECK(ramd_vec_copy_from_host(this->h_, all.data()));
    }
    virtual void RSPMISUpdateCFmap(const BaseVector<int>&, BaseVector<ValueType>*)
    {
        RAMD_ADAPTER_UNSUPPORTED("RSPMISUpdateCFmap (Ruge-Stueben AMG: out of scope)");
    }
    virtual void ExtractCoarseMapping(int64_t, int64_t, const int*, int, int*, int*) const
    {
        RAMD_ADAPTER_UNSUPPORTED("ExtractCoarseMapping");
    }
    virtual void ExtractCoarseBoundary(int64_t, int64_t, const int*, int, int*, int*) const
    {
        RAMD_ADAPTER_UNSUPPORTED("ExtractCoarseBoundary");
    }
    virtual void SetRandomUniform(unsigned long long, ValueType, ValueType)
    {
        RAMD_ADAPTER_UNSUPPORTED("SetRandomUniform");
    }
    virtual void SetRandomNormal(unsigned long long, ValueType, ValueType)
    {
        RAMD_ADAPTER_UNSUPPORTED("SetRandomNormal");
    }
    virtual void Sort(BaseVector<ValueType>*, BaseVector<int>*) const
    {
        RAMD_ADAPTER_UNSUPPORTED("Sort");
    }

private:
    ramd_vec_t h_ = NULL;
    friend class MI355XAcceleratorMatrix<ValueType>;
};

// One class for CSR / ELL / HYB / COO: the format lives in the handle (ramd_mat_convert), GetMatFormat reports it.
template <typename ValueType>
class MI355XAcceleratorMatrix : public AcceleratorMatrix<ValueType>
{
    typedef MI355XAcceleratorVector<ValueType> Vec;

public:
    explicit MI355XAcceleratorMatrix(const Rocalution_Backend_Descriptor& local_backend)
    {
        this->set_backend(local_backend);
        RAMD_ADAPTER_CHECK(ramd_mat_create(MI355XType<ValueType>::id, &this->h_));
    }
    virtual ~MI355XAcceleratorMatrix()
    {
        ramd_mat_destroy(this->h_);
    }
    virtual void Info(void) const
    {
        std::printf("MI355XAcceleratorMatrix<ValueType>, %d x %d, %lld entries\n", this->nrow_, this->ncol_, (long long)this->nnz_);
    }
    virtual unsigned int GetMatFormat(void) const // base_matrix.hpp:94
    {
        int fmt = RAMD_CSR;
        RAMD_ADAPTER_CHECK(ramd_mat_info(this->h_, NULL, NULL, NULL, &fmt, NULL));
        return fmt == RAMD_CSR ? CSR : fmt == RAMD_ELL ? ELL : fmt == RAMD_HYB ? HYB : COO;
    }
    virtual void Clear(void) // :167
    {
        RAMD_ADAPTER_CHECK(ramd_mat_clear(this->h_));
        this->nrow_ = this->ncol_ = 0;
        this->nnz_  = 0;
    }
    virtual bool ConvertFrom(const BaseMatrix<ValueType>& mat) // :235 -- false: the reference converts on the host
    {
        const MI355XAcceleratorMatrix<ValueType>* src = dynamic_cast<const MI355XAcceleratorMatrix<ValueType>*>(&mat);
        if(src == NULL)
            return false;
        const unsigned int want = this->want_format_;
        this->CopyFrom(mat);
        const int fmt = want == CSR ? RAMD_CSR : want == ELL ? RAMD_ELL : want == HYB ? RAMD_HYB : want == COO ? RAMD_COO : -1;
        return fmt >= 0 && ramd_mat_convert(this->h_, fmt) == RAMD_OK;
    }
    void SetWantedFormat(unsigned int format) // (the factory of backend_manager.cpp creates the object for a format)
    {
        this->want_format_ = format;
    }
    virtual void CopyFrom(const BaseMatrix<ValueType>& mat) // :238
    {
        const MI355XAcceleratorMatrix<ValueType>* src = dynamic_cast<const MI355XAcceleratorMatrix<ValueType>*>(&mat);
        if(src == NULL)
        {
            const HostMatrix<ValueType>* host = dynamic_cast<const HostMatrix<ValueType>*>(&mat);
            if(host == NULL)
                RAMD_ADAPTER_UNSUPPORTED("CopyFrom a matrix of another accelerator backend");
            this->CopyFromHost(*host);
            return;
        }
        ramd_mat_t copy = NULL;
        RAMD_ADAPTER_CHECK(ramd_mat_clone(src->h_, &copy));
        ramd_mat_destroy(this->h_);
        this->h_ = copy;
        this->refresh_();
    }
    virtual void CopyTo(BaseMatrix<ValueType>* mat) const // :241
    {
        mat->CopyFrom(*this);
    }
    virtual void CopyFromHost(const HostMatrix<ValueType>& src) // :847
    {
        const HostMatrixCSR<ValueType>* csr = dynamic_cast<const HostMatrixCSR<ValueType>*>(&src);
        if(csr == NULL)
            RAMD_ADAPTER_UNSUPPORTED("CopyFromHost of a host matrix that is not CSR (LocalMatrix converts first)");
        const int     n   = csr->GetM();
        const int64_t nnz = csr->GetNnz();
        std::vector<PtrType>   rp((size_t)n + 1);
        std::vector<int>       ci((size_t)nnz);
        std::vector<ValueType> va((size_t)nnz);
        csr->CopyToCSR(rp.data(), ci.data(), va.data()); // (a friend of HostMatrixCSR passes mat_.row_offset / col / val instead)
        std::vector<int32_t> rp32(rp.begin(), rp.end());
        RAMD_ADAPTER_CHECK(ramd_mat_set_csr_from_host(this->h_, n, csr->GetN(), nnz, rp32.data(), ci.data(), va.data()));
        this->refresh_();
    }
    virtual void CopyToHost(HostMatrix<ValueType>* dst) const // :853
    {
        HostMatrixCSR<ValueType>* csr = dynamic_cast<HostMatrixCSR<ValueType>*>(dst);
        if(csr == NULL || this->GetMatFormat() != CSR)
            RAMD_ADAPTER_UNSUPPORTED("CopyToHost in a format other than CSR (LocalMatrix converts first)");
        std::vector<int32_t>   rp32((size_t)this->nrow_ + 1);
        std::vector<int>       ci((size_t)this->nnz_);
        std::vector<ValueType> va((size_t)this->nnz_);
        RAMD_ADAPTER_CHECK(ramd_mat_copy_csr_to_host(this->h_, rp32.data(), ci.data(), va.data()));
        std::vector<PtrType> rp(rp32.begin(), rp32.end());
        csr->AllocateCSR(this->nnz_, this->nrow_, this->ncol_);
        csr->CopyFromCSR(rp.data(), ci.data(), va.data());
    }
    virtual void Apply(const BaseVector<ValueType>& in, BaseVector<ValueType>* out) const // :450
    {
        RAMD_ADAPTER_CHECK(ramd_mat_apply(this->h_, Vec::of(in), Vec::of(*out)));
    }
    virtual void ApplyAdd(const BaseVector<ValueType>& in, ValueType scalar, BaseVector<ValueType>* out) const // :452
    {
        RAMD_ADAPTER_CHECK(ramd_mat_apply_add(this->h_, Vec::of(in), (double)scalar, Vec::of(*out)));
    }
    // ---- the optional operations of the preconditioned-Krylov path; RAMD_ERR_UNSUPPORTED -> false -> host fallback
    virtual bool ExtractDiagonal(BaseVector<ValueType>* vec_diag) const // :193
    {
        return ramd_mat_extract_diag(this->h_, Vec::of(*vec_diag)) == RAMD_OK;
    }
    virtual bool ExtractInverseDiagonal(BaseVector<ValueType>* vec_inv_diag) const // :195
    {
        return ramd_mat_extract_inv_diag(this->h_, Vec::of(*vec_inv_diag)) == RAMD_OK;
    }
    virtual bool Permute(const BaseVector<int>& permutation) // :206
    {
        return ramd_mat_permute(this->h_, Vec::template of<int>(permutation)) == RAMD_OK;
    }
    virtual bool ILU0Factorize(void) // :321
    {
        return ramd_mat_ilu0_factorize(this->h_) == RAMD_OK;
    }
    virtual void LUAnalyse(void) // :344
    {
        RAMD_ADAPTER_CHECK(ramd_mat_lu_analyse(this->h_));
    }
    virtual void LUAnalyseClear(void) // :346
    {
        RAMD_ADAPTER_CHECK(ramd_mat_lu_analyse_clear(this->h_));
    }
    virtual bool LUSolve(const BaseVector<ValueType>& in, BaseVector<ValueType>* out) const // :349
    {
        return ramd_mat_lu_solve(this->h_, Vec::of(in), Vec::of(*out)) == RAMD_OK;
    }
    virtual void LAnalyse(bool diag_unit = false)
    {
        RAMD_ADAPTER_CHECK(ramd_mat_l_analyse(this->h_, diag_unit ? 1 : 0));
    }
    virtual void LAnalyseClear(void)
    {
        RAMD_ADAPTER_CHECK(ramd_mat_l_analyse_clear(this->h_));
    }
    virtual bool LSolve(const BaseVector<ValueType>& in, BaseVector<ValueType>* out) const
    {
        return ramd_mat_l_solve(this->h_, Vec::of(in), Vec::of(*out)) == RAMD_OK;
    }
    virtual void UAnalyse(bool diag_unit = false)
    {
        RAMD_ADAPTER_CHECK(ramd_mat_u_analyse(this->h_, diag_unit ? 1 : 0));
    }
    virtual void UAnalyseClear(void)
    {
        RAMD_ADAPTER_CHECK(ramd_mat_u_analyse_clear(this->h_));
    }
    virtual bool USolve(const BaseVector<ValueType>& in, BaseVector<ValueType>* out) const
    {
        return ramd_mat_u_solve(this->h_, Vec::of(in), Vec::of(*out)) == RAMD_OK;
    }
    virtual bool MultiColoring(int& num_colors, int** size_colors, BaseVector<int>* permutation) const
    {
        std::vector<int> sizes((size_t)this->nrow_);
        if(ramd_mat_multicoloring(this->h_, &num_colors, sizes.data(), Vec::template of<int>(*permutation)) != RAMD_OK)
            return false;
        *size_colors = new int[(size_t)num_colors]; // (the reference allocates with allocate_host and frees with free_host)
        for(int i = 0; i < num_colors; ++i)
            (*size_colors)[i] = sizes[(size_t)i];
        return true;
    }

private:
    void refresh_(void)
    {
        int     nrow = 0, ncol = 0;
        int64_t nnz  = 0;
        RAMD_ADAPTER_CHECK(ramd_mat_info(this->h_, &nrow, &ncol, &nnz, NULL, NULL));
        this->nrow_ = nrow;
        this->ncol_ = ncol;
        this->nnz_  = nnz;
    }
    ramd_mat_t   h_           = NULL;
    unsigned int want_format_ = CSR;
};

// The two factory branches (src/base/backend_manager.cpp:427-470) and the lifecycle calls a maintainer adds:
//   template <typename ValueType> AcceleratorVector<ValueType>* _rocalution_init_base_backend_vector(const Rocalution_Backend_Descriptor& d)
//   { if(d.backend == MI355X) return new MI355XAcceleratorVector<ValueType>(d);  ... the existing HIP branch ... }
//   template <typename ValueType> AcceleratorMatrix<ValueType>* _rocalution_init_base_backend_matrix(const Rocalution_Backend_Descriptor& d,
//                                                                                                    unsigned int matrix_format, int blockdim)
//   { if(d.backend == MI355X) { MI355XAcceleratorMatrix<ValueType>* m = new MI355XAcceleratorMatrix<ValueType>(d);
//                               m->SetWantedFormat(matrix_format); return m; }  ... }
//   rocalution_init_hip() -> ramd_init(dev); rocalution_stop_hip() -> ramd_stop(); rocalution_hip_sync*() -> ramd_sync*();
//   rocalution_hip_compute_{interior,ghost,default}() -> ramd_compute_*(); allocate_pinned / free_pinned -> ramd_alloc_pinned / ramd_free_pinned.

} // namespace rocalution
