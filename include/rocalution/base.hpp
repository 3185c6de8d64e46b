// rocalution/base.hpp -- source-compatible front end (subset) of rocALUTION's LocalVector /
// LocalMatrix and backend free functions, implemented on the C ABI of librocalution_amd.so.
//
// Mirrors, for the preconditioned-Krylov hot path only:
//   src/base/backend_manager.hpp:169-377   init_rocalution, stop_rocalution, info_rocalution, ...
//   src/base/local_vector.hpp:126-641      LocalVector<ValueType>
//   src/base/local_matrix.hpp:77-1027      LocalMatrix<ValueType>
// Same names, argument meaning and error behaviour (LOG + exit(1) on fatal errors,
// src/utils/log.hpp:95-100).  Drivers written against <rocalution/rocalution.hpp> that stay inside
// this subset recompile unchanged with a plain host compiler (no HIP headers needed here).
//
// There is NO host compute backend in this implementation: objects hold plain host storage until
// MoveToAccelerator() and every numerical operation requires the accelerator.  (The reference's
// host/OpenMP backend is what the oracle restates; it is deliberately not shipped here.)
#pragma once

#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <functional>
#include <vector>

#include "../rocalution_amd.h"

namespace rocalution
{

typedef int32_t PtrType; // src/utils/types.hpp.in:30-32 (default build)

enum _matrix_format // src/base/matrix_formats.hpp
{
    DENSE = 0,
    CSR   = 1,
    MCSR  = 2,
    BCSR  = 3,
    COO   = 4,
    DIA   = 5,
    ELL   = 6,
    HYB   = 7
};

#define LOG_INFO(stream)                   \
    do                                     \
    {                                      \
        std::cout << stream << std::endl;  \
    } while(0)

// Error convention of the reference (src/utils/log.hpp:95-100): log, then exit(1).  A translation unit that embeds this
// layer behind a status-code interface (csrc/capi_solvers.cpp: the C ABI Python binds) defines RAMD_FATAL_THROWS before
// including it; a fatal error then unwinds as rocalution::fatal_error and becomes an error status there.
struct fatal_error : public std::runtime_error
{
    fatal_error(const char* file, int line)
        : std::runtime_error(std::string("fatal error in ") + file + ":" + std::to_string(line))
    {
    }
};
// (two functions with two names -- one inline function with two bodies would be an ODR violation as soon as a throwing and
//  an exiting translation unit met in one program)
[[noreturn]] inline void _fatal_throw(const char* file, int line)
{
    throw fatal_error(file, line);
}
[[noreturn]] inline void _fatal_exit(const char* file, int line)
{
    std::cout << "Fatal error - the program will be terminated" << std::endl;
    std::cout << "File: " << file << "; line: " << line << std::endl;
    exit(1);
}
#ifdef RAMD_FATAL_THROWS
#define FATAL_ERROR(file, line) ::rocalution::_fatal_throw(file, line)
#else
#define FATAL_ERROR(file, line) ::rocalution::_fatal_exit(file, line)
#endif
#define RAMD_DIE() FATAL_ERROR(__FILE__, __LINE__)
// precondition checks stay active in release builds, as the reference's asserts do (src/utils/def.hpp:50-58), also
// under -DNDEBUG, and fail through the same channel as every other fatal error (a status code behind the C ABI, not abort())
#define RAMD_EXPECT(cond)                  \
    do                                     \
    {                                      \
        if(!(cond))                        \
        {                                  \
            std::cout << "Assertion failed: " << #cond << std::endl; \
            RAMD_DIE();                    \
        }                                  \
    } while(0)

// one log line from any number of streamable pieces
inline void say_more(void) {}
template <typename First, typename... Rest>
inline void say_more(const First& first, const Rest&... rest)
{
    std::cout << first;
    say_more(rest...);
}
template <typename... Args>
inline void say(const Args&... args)
{
    say_more(args...);
    std::cout << std::endl;
}
// a literal in the solver's value type
template <typename T>
inline T num(double v)
{
    return static_cast<T>(v);
}

// status of a C-ABI call -> reference error convention
inline void _check(int status, const char* what, const char* file, int line)
{
    if(status != RAMD_OK)
    {
        LOG_INFO("rocalution_amd: " << what << " failed (status " << status
                                    << "): " << ramd_last_error());
        FATAL_ERROR(file, line);
    }
}
#define RAMD_CHECK(call) ::rocalution::_check((call), #call, __FILE__, __LINE__)

template <typename T>
struct _dtype;
template <>
struct _dtype<double>
{
    static constexpr int value = RAMD_F64;
};
template <>
struct _dtype<float>
{
    static constexpr int value = RAMD_F32;
};
template <>
struct _dtype<int>
{
    static constexpr int value = RAMD_I32;
};

// ---------------------------------------------------------------------------- backend
struct _backend_state
{
    bool init          = false;
    bool accel_disable = false;
    int  device        = -1;
};
inline _backend_state& _state()
{
    static _backend_state s;
    return s;
}

inline int init_rocalution(int rank = -1, int dev_per_node = 1)
{
    _backend_state& s = _state();
    if(s.init)
        return 0;
    if(s.accel_disable)
    {
        LOG_INFO("rocalution_amd: the accelerator is disabled, but this library has no host compute "
                 "backend (use the reference library for host runs)");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    int dev = s.device;
    if(rank >= 0 && dev_per_node > 0) // backend_manager.cpp:180-185: device = rank % dev_per_node
        dev = rank % dev_per_node;
    RAMD_CHECK(ramd_init(dev));
    s.init = true;
    return 0;
}
inline int stop_rocalution(void)
{
    if(_state().init)
        RAMD_CHECK(ramd_stop());
    _state().init = false;
    return 0;
}
inline void set_device_rocalution(int dev)
{
    _state().device = dev;
}
inline void disable_accelerator_rocalution(bool onoff = true)
{
    _state().accel_disable = onoff;
}
inline void set_omp_threads_rocalution(int) {}
inline void set_omp_affinity_rocalution(bool) {}
inline void set_omp_threshold_rocalution(int) {}
inline void info_rocalution(void)
{
    char buf[512];
    RAMD_CHECK(ramd_info(buf, (int)sizeof(buf)));
    LOG_INFO(buf);
}
inline void _rocalution_sync(void)
{
    if(_state().init)
        RAMD_CHECK(ramd_sync());
}
// microseconds, device synchronised first (src/utils/time_functions.cpp:46-69)
inline double rocalution_time(void)
{
    _rocalution_sync();
    auto now = std::chrono::steady_clock::now().time_since_epoch();
    return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(now).count() / 1e3;
}

template <typename DataType>
void allocate_host(int64_t n, DataType** ptr)
{
    *ptr = (n > 0) ? new DataType[n] : NULL;
}
template <typename DataType>
void free_host(DataType** ptr)
{
    delete[] * ptr;
    *ptr = NULL;
}
template <typename DataType>
void set_to_zero_host(int64_t n, DataType* ptr)
{
    if(n > 0)
        memset(ptr, 0, sizeof(DataType) * (size_t)n);
}

// ---------------------------------------------------------------------------- LocalVector
template <typename ValueType>
class LocalMatrix;

template <typename ValueType>
class LocalVector
{
public:
    LocalVector() {}
    ~LocalVector()
    {
        if(this->dev_ && this->own_) // adopted (non-owned) handles are left untouched
            ramd_vec_destroy(this->dev_);
    }
    LocalVector(const LocalVector&)            = delete;
    LocalVector& operator=(const LocalVector&) = delete;

    void MoveToAccelerator(void)
    {
        if(this->on_accel_)
            return;
        this->ensure_dev_();
        if(!this->host_.empty())
        {
            RAMD_CHECK(ramd_vec_allocate(this->dev_, (int64_t)this->host_.size()));
            RAMD_CHECK(ramd_vec_copy_from_host(this->dev_, this->host_.data()));
            std::vector<ValueType>().swap(this->host_);
        }
        this->on_accel_ = true;
    }
    void MoveToHost(void)
    {
        if(!this->on_accel_)
            return;
        int64_t n = this->GetSize();
        this->host_.assign((size_t)n, ValueType(0));
        if(n > 0)
            RAMD_CHECK(ramd_vec_copy_to_host(this->dev_, this->host_.data()));
        RAMD_CHECK(ramd_vec_clear(this->dev_));
        this->on_accel_ = false;
    }
    bool is_accel_(void) const
    {
        return this->on_accel_;
    }
    template <class Obj>
    void CloneBackend(const Obj& src)
    {
        if(src.is_accel_())
            this->MoveToAccelerator();
        else
            this->MoveToHost();
    }

    void Allocate(std::string name, int64_t size)
    {
        assert(size >= 0);
        this->name_ = name;
        if(this->on_accel_)
        {
            this->ensure_dev_();
            RAMD_CHECK(ramd_vec_allocate(this->dev_, size));
        }
        else
            this->host_.assign((size_t)size, ValueType(0));
    }
    // extension: Allocate with a placement hint -- `other` is a vector a fused kernel WRITES in the same pass as this one;
    // the block comes from the other placement class (ramd_vec_allocate_apart).  Same contents as Allocate (zeros).
    void AllocateApart(std::string name, int64_t size, const LocalVector<ValueType>& other)
    {
        assert(size >= 0);
        this->name_ = name;
        if(this->on_accel_ && other.on_accel_ && other.dev_ != nullptr)
        {
            this->ensure_dev_();
            RAMD_CHECK(ramd_vec_allocate_apart(this->dev_, size, other.dev_));
        }
        else
            this->Allocate(name, size);
    }
    // ... and for a vector that is already allocated: the pair is measured, the vector moves (contents kept) where a
    // candidate block pairs clearly better with `other` (ramd_vec_place_apart)
    void PlaceApartFrom(const LocalVector<ValueType>& other)
    {
        if(!this->on_accel_ || !other.on_accel_ || this->dev_ == nullptr || other.dev_ == nullptr || &other == this)
            return;
        int moved = 0;
        RAMD_CHECK(ramd_vec_place_apart(this->dev_, other.dev_, &moved));
    }
    // ... and by trial: `run` launches the kernels that use this vector; the vector moves to the fastest of its own block
    // and `tries` fresh ones (ramd_vec_place_by_trial).  `run` is executed 3 (tries + 1) times.
    void PlaceByTrial(const std::function<void()>& run, int tries, double stop_ratio = 0.0,
                      const LocalVector<ValueType>* apart_from = nullptr)
    {
        if(!this->on_accel_ || this->dev_ == nullptr)
            return;
        struct Hop
        {
            static int call(void* ctx)
            {
                (*static_cast<const std::function<void()>*>(ctx))();
                return RAMD_OK;
            }
        };
        int moved = 0;
        RAMD_CHECK(ramd_vec_place_by_trial(this->dev_, &Hop::call, const_cast<std::function<void()>*>(&run), tries, stop_ratio,
                                           (apart_from && apart_from->on_accel_) ? apart_from->dev_ : nullptr, &moved));
    }
    void Clear(void)
    {
        if(this->dev_)
            ramd_vec_clear(this->dev_);
        std::vector<ValueType>().swap(this->host_);
    }
    int64_t GetSize(void) const
    {
        if(this->on_accel_)
        {
            int64_t n = 0;
            if(this->dev_)
                RAMD_CHECK(ramd_vec_size(this->dev_, &n));
            return n;
        }
        return (int64_t)this->host_.size();
    }
    int64_t GetLocalSize(void) const
    {
        return this->GetSize();
    }
    void Info(void) const
    {
        LOG_INFO("LocalVector name=" << this->name_ << "; size=" << this->GetSize()
                                     << "; prec=" << 8 * sizeof(ValueType) << "bit; "
                                     << (this->on_accel_ ? "accelerator backend: MI355X-native HIP"
                                                         : "host storage (no host compute backend)"));
    }

    // storage-level operations work on both sides; numerical ones need the accelerator
    void Zeros(void)
    {
        if(this->on_accel_)
            RAMD_CHECK(ramd_vec_zeros(this->dev_));
        else
            std::fill(this->host_.begin(), this->host_.end(), ValueType(0));
    }
    void Ones(void)
    {
        this->SetValues(ValueType(1));
    }
    void SetValues(ValueType val)
    {
        if(this->on_accel_)
            RAMD_CHECK(ramd_vec_set_values(this->dev_, (double)val));
        else
            std::fill(this->host_.begin(), this->host_.end(), val);
    }
    // HostVector::SetRandomUniform / SetRandomNormal (src/base/host/host_vector.cpp:374-405): the C
    // library's srand/rand sequence (Box-Muller for the normal variant).  Generated on the host exactly
    // as the reference's host backend does and uploaded when the vector lives on the accelerator.
    void SetRandomUniform(unsigned long long seed, ValueType a = static_cast<ValueType>(-1),
                          ValueType b = static_cast<ValueType>(1))
    {
        assert(a <= b);
        std::vector<ValueType> h((size_t)this->GetSize());
        srand((unsigned)seed);
        for(size_t i = 0; i < h.size(); ++i)
            h[i] = a + static_cast<ValueType>(rand()) / static_cast<ValueType>(RAND_MAX) * (b - a);
        this->CopyFromHostData(h.data());
    }
    void SetRandomNormal(unsigned long long seed, ValueType mean = static_cast<ValueType>(0),
                         ValueType var = static_cast<ValueType>(1))
    {
        std::vector<ValueType> h((size_t)this->GetSize());
        srand((unsigned)seed);
        for(size_t i = 0; i < h.size(); ++i)
        {
            ValueType u1 = static_cast<ValueType>(rand()) / static_cast<ValueType>(RAND_MAX);
            ValueType u2 = static_cast<ValueType>(rand()) / static_cast<ValueType>(RAND_MAX);
            h[i] = std::sqrt(static_cast<ValueType>(-2) * std::log(u1)) * std::cos(static_cast<ValueType>(2 * 3.14159265358979323846) * u2);
            h[i] = mean + var * h[i];
        }
        this->CopyFromHostData(h.data());
    }
    ValueType& operator[](int64_t i)
    {
        assert(!this->on_accel_ && i >= 0 && i < (int64_t)this->host_.size());
        return this->host_[(size_t)i];
    }
    const ValueType& operator[](int64_t i) const
    {
        assert(!this->on_accel_ && i >= 0 && i < (int64_t)this->host_.size());
        return this->host_[(size_t)i];
    }
    void SetDataPtr(ValueType** ptr, std::string name, int64_t size)
    {
        assert(ptr != NULL && *ptr != NULL && size >= 0);
        this->Clear();
        this->name_ = name;
        if(this->on_accel_)
        {
            RAMD_CHECK(ramd_vec_allocate(this->dev_, size));
            RAMD_CHECK(ramd_vec_copy_from_host(this->dev_, *ptr));
        }
        else
            this->host_.assign(*ptr, *ptr + size);
        delete[] * ptr; // ownership moves into the object (local_vector.cpp SetDataPtr)
        *ptr = NULL;
    }
    void LeaveDataPtr(ValueType** ptr)
    {
        assert(*ptr == NULL);
        int64_t n = this->GetSize();
        allocate_host(n, ptr);
        this->CopyToHostData(*ptr);
        this->Clear();
    }
    // vector files (host_vector.cpp:415-632): include/rocalution/io.hpp
    void ReadFileASCII(const std::string& filename);
    void WriteFileASCII(const std::string& filename) const;
    void ReadFileBinary(const std::string& filename);
    void WriteFileBinary(const std::string& filename) const;
    void CopyFromData(const ValueType* data)
    {
        this->CopyFromHostData(data);
    }
    void CopyFromHostData(const ValueType* data)
    {
        if(this->GetSize() == 0)
            return;
        if(this->on_accel_)
            RAMD_CHECK(ramd_vec_copy_from_host(this->dev_, data));
        else
            std::copy(data, data + this->host_.size(), this->host_.begin());
    }
    void CopyToData(ValueType* data) const
    {
        this->CopyToHostData(data);
    }
    void CopyToHostData(ValueType* data) const
    {
        if(this->GetSize() == 0)
            return;
        if(this->on_accel_)
            RAMD_CHECK(ramd_vec_copy_to_host(this->dev_, data));
        else
            std::copy(this->host_.begin(), this->host_.end(), data);
    }
    void CopyFrom(const LocalVector<ValueType>& src)
    {
        assert(this != &src);
        if(this->on_accel_ && src.on_accel_)
            RAMD_CHECK(ramd_vec_copy_from(this->dev_, src.dev_));
        else if(!this->on_accel_ && !src.on_accel_)
            this->host_ = src.host_;
        else if(this->on_accel_)
        {
            if(this->GetSize() != src.GetSize())
                RAMD_CHECK(ramd_vec_allocate(this->dev_, src.GetSize()));
            if(src.GetSize() > 0)
                RAMD_CHECK(ramd_vec_copy_from_host(this->dev_, src.host_.data()));
        }
        else
        {
            this->host_.assign((size_t)src.GetSize(), ValueType(0));
            if(src.GetSize() > 0)
                RAMD_CHECK(ramd_vec_copy_to_host(src.dev_, this->host_.data()));
        }
    }
    void CopyFrom(const LocalVector<ValueType>& src, int64_t src_offset, int64_t dst_offset,
                  int64_t size)
    {
        if(this->on_accel_ && src.on_accel_)
            RAMD_CHECK(ramd_vec_copy_from_offset(this->dev_, src.dev_, src_offset, dst_offset, size));
        else if(!this->on_accel_ && !src.on_accel_)
            std::copy(src.host_.begin() + src_offset, src.host_.begin() + src_offset + size,
                      this->host_.begin() + dst_offset);
        else
            this->no_host_("CopyFrom(offset) across backends");
    }
    void CloneFrom(const LocalVector<ValueType>& src)
    {
        this->CloneBackend(src);
        this->CopyFrom(src);
    }
    void CopyFromFloat(const LocalVector<float>& src)
    {
        this->need_accel_("CopyFromFloat");
        RAMD_CHECK(ramd_vec_copy_from_float(this->dev_, src.handle()));
    }
    void CopyFromDouble(const LocalVector<double>& src)
    {
        this->need_accel_("CopyFromDouble");
        RAMD_CHECK(ramd_vec_copy_from_double(this->dev_, src.handle()));
    }
    void CopyFromPermute(const LocalVector<ValueType>& src, const LocalVector<int>& permutation)
    {
        this->need_accel_("CopyFromPermute");
        RAMD_CHECK(ramd_vec_copy_from_permute(this->dev_, src.dev_, permutation.handle()));
    }
    void CopyFromPermuteBackward(const LocalVector<ValueType>& src,
                                 const LocalVector<int>&       permutation)
    {
        this->need_accel_("CopyFromPermuteBackward");
        RAMD_CHECK(ramd_vec_copy_from_permute_backward(this->dev_, src.dev_, permutation.handle()));
    }

    // ---- BLAS-1 (accelerator only)
    void AddScale(const LocalVector<ValueType>& x, ValueType alpha)
    {
        this->need_accel_("AddScale");
        RAMD_CHECK(ramd_vec_add_scale(this->dev_, x.dev_, (double)alpha));
    }
    void ScaleAdd(ValueType alpha, const LocalVector<ValueType>& x)
    {
        this->need_accel_("ScaleAdd");
        RAMD_CHECK(ramd_vec_scale_add(this->dev_, (double)alpha, x.dev_));
    }
    void ScaleAddScale(ValueType alpha, const LocalVector<ValueType>& x, ValueType beta, int64_t src_offset,
                       int64_t dst_offset, int64_t size)
    {
        this->need_accel_("ScaleAddScale");
        RAMD_CHECK(ramd_vec_scale_add_scale_offset(this->dev_, (double)alpha, x.handle(), (double)beta, src_offset,
                                                   dst_offset, size));
    }
    void ScaleAddScale(ValueType alpha, const LocalVector<ValueType>& x, ValueType beta)
    {
        this->need_accel_("ScaleAddScale");
        RAMD_CHECK(ramd_vec_scale_add_scale(this->dev_, (double)alpha, x.dev_, (double)beta));
    }
    void ScaleAdd2(ValueType alpha, const LocalVector<ValueType>& x, ValueType beta,
                   const LocalVector<ValueType>& y, ValueType gamma)
    {
        this->need_accel_("ScaleAdd2");
        RAMD_CHECK(ramd_vec_scale_add2(this->dev_, (double)alpha, x.dev_, (double)beta, y.dev_,
                                       (double)gamma));
    }
    void Scale(ValueType alpha)
    {
        this->need_accel_("Scale");
        RAMD_CHECK(ramd_vec_scale(this->dev_, (double)alpha));
    }
    ValueType Dot(const LocalVector<ValueType>& x) const
    {
        this->need_accel_("Dot");
        double r = 0;
        RAMD_CHECK(ramd_vec_dot(this->dev_, x.dev_, &r));
        return (ValueType)r;
    }
    ValueType DotNonConj(const LocalVector<ValueType>& x) const
    {
        return this->Dot(x);
    }
    ValueType Norm(void) const
    {
        this->need_accel_("Norm");
        double r = 0;
        RAMD_CHECK(ramd_vec_norm(this->dev_, &r));
        return (ValueType)r;
    }
    ValueType Reduce(void) const
    {
        this->need_accel_("Reduce");
        double r = 0;
        RAMD_CHECK(ramd_vec_reduce(this->dev_, &r));
        return (ValueType)r;
    }
    ValueType Asum(void) const
    {
        this->need_accel_("Asum");
        double r = 0;
        RAMD_CHECK(ramd_vec_asum(this->dev_, &r));
        return (ValueType)r;
    }
    int64_t Amax(ValueType& value) const
    {
        this->need_accel_("Amax");
        double  r = 0;
        int64_t i = 0;
        RAMD_CHECK(ramd_vec_amax(this->dev_, &r, &i));
        value = (ValueType)r;
        return i;
    }
    void PointWiseMult(const LocalVector<ValueType>& x)
    {
        this->need_accel_("PointWiseMult");
        RAMD_CHECK(ramd_vec_pointwise_mult(this->dev_, x.dev_));
    }
    void PointWiseMult(const LocalVector<ValueType>& x, const LocalVector<ValueType>& y)
    {
        this->need_accel_("PointWiseMult");
        RAMD_CHECK(ramd_vec_pointwise_mult2(this->dev_, x.dev_, y.dev_));
    }
    void GetIndexValues(const LocalVector<int>& index, LocalVector<ValueType>* values) const
    {
        this->need_accel_("GetIndexValues");
        RAMD_CHECK(ramd_vec_get_index_values(this->dev_, index.handle(), values->dev_));
    }

    // ---- extensions used by this library's own layers
    ramd_vec_t handle(void) const
    {
        return this->dev_;
    }
    // non-owning view of an existing accelerator vector (Python / C bindings)
    void AdoptDeviceHandle(ramd_vec_t h)
    {
        std::vector<ValueType>().swap(this->host_);
        if(this->dev_ && this->own_)
            ramd_vec_destroy(this->dev_);
        this->dev_      = h;
        this->own_      = false;
        this->on_accel_ = true;
    }

private:
    void ensure_dev_(void)
    {
        if(!this->dev_)
        {
            RAMD_CHECK(ramd_vec_create(_dtype<ValueType>::value, &this->dev_));
            this->own_ = true;
        }
    }
    void need_accel_(const char* op) const
    {
        if(!this->on_accel_)
            this->no_host_(op);
    }
    void no_host_(const char* op) const
    {
        LOG_INFO("LocalVector::" << op << "() on a host object: this library has no host compute "
                                 << "backend - call MoveToAccelerator() first");
        FATAL_ERROR(__FILE__, __LINE__);
    }

    std::string            name_;
    std::vector<ValueType> host_;
    ramd_vec_t             dev_      = NULL;
    bool                   own_      = true;
    bool                   on_accel_ = false;
};

// ---------------------------------------------------------------------------- LocalMatrix
template <typename ValueType>
class LocalMatrix
{
public:
    LocalMatrix() {}
    ~LocalMatrix()
    {
        if(this->dev_ && this->own_) // adopted (non-owned) handles are left untouched
            ramd_mat_destroy(this->dev_);
    }
    LocalMatrix(const LocalMatrix&)            = delete;
    LocalMatrix& operator=(const LocalMatrix&) = delete;

    bool is_accel_(void) const
    {
        return this->on_accel_;
    }
    ramd_mat_t handle(void) const
    {
        return this->dev_;
    }
    void AdoptDeviceHandle(ramd_mat_t h)
    {
        this->Clear();
        if(this->dev_ && this->own_)
            ramd_mat_destroy(this->dev_);
        this->dev_      = h;
        this->own_      = false;
        this->on_accel_ = true;
    }

    void Clear(void)
    {
        if(this->dev_ && this->own_)
            ramd_mat_clear(this->dev_);
        std::vector<PtrType>().swap(this->h_rp_);
        std::vector<int>().swap(this->h_ci_);
        std::vector<ValueType>().swap(this->h_val_);
        this->h_nrow_ = this->h_ncol_ = 0;
    }
    void Info(void) const
    {
        static const char* fmt[] = {"DENSE", "CSR", "MCSR", "BCSR", "COO", "DIA", "ELL", "HYB"};
        LOG_INFO("LocalMatrix name=" << this->name_ << "; rows=" << this->GetM() << "; cols="
                                     << this->GetN() << "; nnz=" << this->GetNnz() << "; prec="
                                     << 8 * sizeof(ValueType) << "bit; format=" << fmt[this->GetFormat()]
                                     << "; "
                                     << (this->on_accel_ ? "accelerator backend: MI355X-native HIP"
                                                         : "host storage (no host compute backend)"));
    }
    int64_t GetM(void) const
    {
        if(!this->on_accel_)
            return this->h_nrow_;
        int nr = 0;
        this->dev_info_(&nr, NULL, NULL, NULL);
        return nr;
    }
    int64_t GetN(void) const
    {
        if(!this->on_accel_)
            return this->h_ncol_;
        int nc = 0;
        this->dev_info_(NULL, &nc, NULL, NULL);
        return nc;
    }
    int64_t GetNnz(void) const
    {
        if(!this->on_accel_)
            return (int64_t)this->h_ci_.size();
        int64_t nnz = 0;
        this->dev_info_(NULL, NULL, &nnz, NULL);
        return nnz;
    }
    int64_t GetLocalM(void) const
    {
        return this->GetM();
    }
    int64_t GetLocalN(void) const
    {
        return this->GetN();
    }
    int64_t GetLocalNnz(void) const
    {
        return this->GetNnz();
    }
    unsigned int GetFormat(void) const
    {
        if(!this->on_accel_)
            return CSR;
        int f = CSR;
        this->dev_info_(NULL, NULL, NULL, &f);
        return (unsigned int)f;
    }

    // ---- host-side construction (CSR only)
    void AllocateCSR(const std::string& name, int64_t nnz, int64_t nrow, int64_t ncol)
    {
        this->Clear();
        this->name_ = name;
        this->h_rp_.assign((size_t)nrow + 1, 0);
        this->h_ci_.assign((size_t)nnz, 0);
        this->h_val_.assign((size_t)nnz, ValueType(0));
        this->h_nrow_ = nrow;
        this->h_ncol_ = ncol;
        if(this->on_accel_)
            this->upload_();
    }
    void SetDataPtrCSR(PtrType** row_offset, int** col, ValueType** val, std::string name,
                       int64_t nnz, int64_t nrow, int64_t ncol)
    {
        assert(row_offset != NULL && *row_offset != NULL);
        this->Clear();
        this->name_ = name;
        this->h_rp_.assign(*row_offset, *row_offset + nrow + 1);
        if(nnz > 0)
        {
            this->h_ci_.assign(*col, *col + nnz);
            this->h_val_.assign(*val, *val + nnz);
        }
        this->h_nrow_ = nrow;
        this->h_ncol_ = ncol;
        // the object takes ownership of the arrays and nulls the caller's pointers
        // (src/base/local_matrix.cpp:714-780)
        delete[] * row_offset;
        delete[] * col;
        delete[] * val;
        *row_offset = NULL;
        *col        = NULL;
        *val        = NULL;
        if(this->on_accel_)
            this->upload_();
    }
    void LeaveDataPtrCSR(PtrType** row_offset, int** col, ValueType** val)
    {
        int64_t nr = this->GetM(), nnz = this->GetNnz();
        allocate_host(nr + 1, row_offset);
        allocate_host(nnz, col);
        allocate_host(nnz, val);
        this->CopyToCSR(*row_offset, *col, *val);
        this->Clear();
    }
    void CopyFromCSR(const PtrType* row_offsets, const int* col, const ValueType* val)
    {
        int64_t nr = this->GetM(), nc = this->GetN(), nnz = this->GetNnz();
        if(this->on_accel_)
            RAMD_CHECK(ramd_mat_set_csr_from_host(this->dev_, (int)nr, (int)nc, nnz, row_offsets, col,
                                                  val));
        else
        {
            std::copy(row_offsets, row_offsets + nr + 1, this->h_rp_.begin());
            std::copy(col, col + nnz, this->h_ci_.begin());
            std::copy(val, val + nnz, this->h_val_.begin());
        }
    }
    void CopyToCSR(PtrType* row_offsets, int* col, ValueType* val) const
    {
        if(this->on_accel_)
            RAMD_CHECK(ramd_mat_copy_csr_to_host(this->dev_, row_offsets, col, val));
        else
        {
            std::copy(this->h_rp_.begin(), this->h_rp_.end(), row_offsets);
            std::copy(this->h_ci_.begin(), this->h_ci_.end(), col);
            std::copy(this->h_val_.begin(), this->h_val_.end(), val);
        }
    }
    // file IO with the reference's formats and semantics: include/rocalution/io.hpp
    bool ReadFileMTX(const std::string& filename);
    bool WriteFileMTX(const std::string& filename) const;
    bool ReadFileCSR(const std::string& filename);
    bool WriteFileCSR(const std::string& filename) const;

    void MoveToAccelerator(void)
    {
        if(this->on_accel_)
            return;
        this->ensure_dev_();
        this->on_accel_ = true;
        if(this->h_nrow_ > 0 || !this->h_rp_.empty())
            this->upload_();
    }
    void MoveToHost(void)
    {
        if(!this->on_accel_)
            return;
        if(this->GetFormat() != CSR)
        {
            LOG_INFO("LocalMatrix::MoveToHost(): only CSR matrices can leave the accelerator");
            FATAL_ERROR(__FILE__, __LINE__);
        }
        int64_t nr = this->GetM(), nc = this->GetN(), nnz = this->GetNnz();
        this->h_rp_.assign((size_t)nr + 1, 0);
        this->h_ci_.assign((size_t)nnz, 0);
        this->h_val_.assign((size_t)nnz, ValueType(0));
        if(nr > 0)
            RAMD_CHECK(ramd_mat_copy_csr_to_host(this->dev_, this->h_rp_.data(), this->h_ci_.data(),
                                                 this->h_val_.data()));
        this->h_nrow_ = nr;
        this->h_ncol_ = nc;
        RAMD_CHECK(ramd_mat_clear(this->dev_));
        this->on_accel_ = false;
    }
    template <class Obj>
    void CloneBackend(const Obj& src)
    {
        if(src.is_accel_())
            this->MoveToAccelerator();
        else
            this->MoveToHost();
    }
    void CloneFrom(const LocalMatrix<ValueType>& src)
    {
        this->Clear();
        this->name_ = src.name_;
        if(src.on_accel_)
        {
            if(this->dev_ && this->own_)
                ramd_mat_destroy(this->dev_);
            this->dev_ = NULL;
            RAMD_CHECK(ramd_mat_clone(src.dev_, &this->dev_));
            this->own_      = true;
            this->on_accel_ = true;
        }
        else
        {
            if(this->on_accel_)
                RAMD_CHECK(ramd_mat_clear(this->dev_));
            this->on_accel_ = false;
            this->h_rp_     = src.h_rp_;
            this->h_ci_     = src.h_ci_;
            this->h_val_    = src.h_val_;
            this->h_nrow_   = src.h_nrow_;
            this->h_ncol_   = src.h_ncol_;
        }
    }

    // ---- format conversion: LocalMatrix::ConvertTo (src/base/local_matrix.cpp:2064-2151).
    // A refused ELL conversion leaves the matrix in CSR with a warning, as in the reference.
    void ConvertTo(unsigned int matrix_format, int blockdim = 1)
    {
        (void)blockdim;
        this->need_accel_("ConvertTo");
        // local_matrix.cpp:2085-2093: anything -> CSR first, then CSR -> target
        if(this->GetFormat() != CSR && matrix_format != CSR && matrix_format != this->GetFormat())
            RAMD_CHECK(ramd_mat_convert(this->dev_, (int)CSR));
        int s = ramd_mat_convert(this->dev_, (int)matrix_format);
        if(s == RAMD_ERR_REFUSED)
        {
            LOG_INFO("*** warning: LocalMatrix::ConvertTo() the conversion was refused ("
                     << ramd_last_error() << "); the matrix stays in CSR format");
            return;
        }
        if(s == RAMD_ERR_UNSUPPORTED)
        {
            LOG_INFO("LocalMatrix::ConvertTo(): format " << matrix_format
                                                         << " is not provided by this backend");
            FATAL_ERROR(__FILE__, __LINE__);
        }
        RAMD_CHECK(s);
    }
    void ConvertToCSR(void)
    {
        this->ConvertTo(CSR);
    }
    void ConvertToELL(void)
    {
        this->ConvertTo(ELL);
    }
    void ConvertToHYB(void)
    {
        this->ConvertTo(HYB);
    }
    void ConvertToDIA(void)
    {
        this->ConvertTo(DIA);
    }
    void ConvertToCOO(void)
    {
        this->ConvertTo(COO);
    }

    // ---- numerical operations
    void Apply(const LocalVector<ValueType>& in, LocalVector<ValueType>* out) const
    {
        this->need_accel_("Apply");
        RAMD_CHECK(ramd_mat_apply(this->dev_, in.handle(), out->handle()));
    }
    void ApplyAdd(const LocalVector<ValueType>& in, ValueType scalar, LocalVector<ValueType>* out) const
    {
        this->need_accel_("ApplyAdd");
        RAMD_CHECK(ramd_mat_apply_add(this->dev_, in.handle(), (double)scalar, out->handle()));
    }
    // ---- matrix utilities next to the solver path (local_matrix.cpp / host_matrix_csr.cpp:919-1160, :3465-3630)
#ifdef RAMD_WITH_OFFSCOPE // (Gershgorin: out of scope)
    void Gershgorin(ValueType& lambda_min, ValueType& lambda_max) const
    {
        this->need_accel_("Gershgorin");
        double lo = 0.0, hi = 0.0;
        RAMD_CHECK(ramd_mat_gershgorin(this->dev_, &lo, &hi));
        lambda_min = static_cast<ValueType>(lo);
        lambda_max = static_cast<ValueType>(hi);
    }
#endif
    void ExtractL(LocalMatrix<ValueType>* L, bool diag) const
    {
        this->need_accel_("ExtractL");
        assert(L != NULL && L != this);
        L->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_extract_tri(this->dev_, L->handle(), 0, diag ? 1 : 0));
    }
    void ExtractU(LocalMatrix<ValueType>* U, bool diag) const
    {
        this->need_accel_("ExtractU");
        assert(U != NULL && U != this);
        U->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_extract_tri(this->dev_, U->handle(), 1, diag ? 1 : 0));
    }
    void Scale(ValueType alpha)
    {
        this->need_accel_("Scale");
        RAMD_CHECK(ramd_mat_scale_values(this->dev_, (double)alpha, 0));
    }
    void ScaleDiagonal(ValueType alpha)
    {
        this->need_accel_("ScaleDiagonal");
        RAMD_CHECK(ramd_mat_scale_values(this->dev_, (double)alpha, 1));
    }
    void ScaleOffDiagonal(ValueType alpha)
    {
        this->need_accel_("ScaleOffDiagonal");
        RAMD_CHECK(ramd_mat_scale_values(this->dev_, (double)alpha, 2));
    }
    void AddScalar(ValueType alpha)
    {
        this->need_accel_("AddScalar");
        RAMD_CHECK(ramd_mat_add_scalar_values(this->dev_, (double)alpha, 0));
    }
    void AddScalarDiagonal(ValueType alpha)
    {
        this->need_accel_("AddScalarDiagonal");
        RAMD_CHECK(ramd_mat_add_scalar_values(this->dev_, (double)alpha, 1));
    }
    void AddScalarOffDiagonal(ValueType alpha)
    {
        this->need_accel_("AddScalarOffDiagonal");
        RAMD_CHECK(ramd_mat_add_scalar_values(this->dev_, (double)alpha, 2));
    }
    // new values into the existing CSR pattern (LocalMatrix::UpdateValuesCSR)
    void UpdateValuesCSR(ValueType* val)
    {
        assert(val != NULL || this->GetNnz() == 0);
        if(this->on_accel_)
            RAMD_CHECK(ramd_mat_update_values(this->dev_, val));
        else
            std::copy(val, val + this->h_val_.size(), this->h_val_.begin());
    }
    // LocalMatrix::Check (host_matrix_csr.cpp:137-240): structural sanity of the CSR data, run on the host like
    // the reference does for both of its backends; false (with a message) instead of garbage later
    bool Check(void) const
    {
        std::vector<PtrType>   rp((size_t)this->GetM() + 1, 0);
        std::vector<int>       ci((size_t)this->GetNnz());
        std::vector<ValueType> va((size_t)this->GetNnz());
        const int64_t          nrow = this->GetM(), ncol = this->GetN(), nnz = this->GetNnz();
        if(nnz == 0)
            return true;
        if(this->GetFormat() != CSR)
        {
            LocalMatrix<ValueType> tmp;
            tmp.CloneFrom(*this);
            tmp.ConvertTo(CSR);
            tmp.CopyToCSR(rp.data(), ci.data(), va.data());
        }
        else
            this->CopyToCSR(rp.data(), ci.data(), va.data());
        auto fail = [](const char* what) {
            LOG_INFO("*** error: Matrix CSR:Check - " << what);
            return false;
        };
        for(int64_t i = 0; i <= nrow; ++i)
            if(rp[(size_t)i] < 0 || rp[(size_t)i] > nnz)
                return fail("problems with matrix row offset pointers");
        bool sorted = true;
        for(int64_t i = 0; i < nrow; ++i)
            for(PtrType j = rp[(size_t)i]; j < rp[(size_t)i + 1]; ++j)
            {
                const int col  = ci[(size_t)j];
                const int prev = (j > rp[(size_t)i]) ? ci[(size_t)j - 1] : -1;
                if(col < 0 || col > ncol)
                    return fail("problems with matrix col values");
                if(col == prev)
                    return fail("problems with matrix col values - the matrix has duplicated column entries");
                const ValueType v = va[(size_t)j];
                if(v == std::numeric_limits<ValueType>::infinity() || v != v)
                    return fail("problems with matrix values");
                if(j > rp[(size_t)i] && prev >= col)
                    sorted = false;
            }
        if(!sorted)
            LOG_INFO("*** warning: Matrix CSR:Check - the matrix has not sorted columns");
        return true;
    }
    // COO input (LocalMatrix::SetDataPtrCOO, local_matrix.cpp:782-850): kept as row-sorted data -- entries are
    // moved to their rows by a STABLE counting sort, so the products of a row are still added in storage
    // order (the order of the reference's serial COO loop restricted to that row).  GetFormat() == COO once
    // the object is on the accelerator.
    void SetDataPtrCOO(int** row, int** col, ValueType** val, std::string name, int64_t nnz, int64_t nrow,
                       int64_t ncol)
    {
        assert(row != NULL && col != NULL && val != NULL);
        this->CopyFromCOO_(*row, *col, *val, name, nnz, nrow, ncol);
        delete[] * row;
        delete[] * col;
        delete[] * val;
        *row = NULL;
        *col = NULL;
        *val = NULL;
    }
    void LeaveDataPtrCOO(int** row, int** col, ValueType** val)
    {
        const int64_t nnz = this->GetNnz();
        allocate_host(nnz, row);
        allocate_host(nnz, col);
        allocate_host(nnz, val);
        this->CopyToCOO(*row, *col, *val);
        this->Clear();
    }
    void CopyToCOO(int* row, int* col, ValueType* val) const
    {
        const int64_t          nrow = this->GetM(), nnz = this->GetNnz();
        std::vector<PtrType>   rp((size_t)nrow + 1, 0);
        const LocalMatrix<ValueType>* src = this;
        LocalMatrix<ValueType>        tmp;
        if(this->GetFormat() != CSR)
        {
            tmp.CloneFrom(*this);
            tmp.ConvertTo(CSR);
            src = &tmp;
        }
        if(nrow > 0)
            src->CopyToCSR(rp.data(), col, val);
        for(int64_t i = 0; i < nrow; ++i)
            for(PtrType j = rp[(size_t)i]; j < rp[(size_t)i + 1]; ++j)
                row[j] = (int)i;
        (void)nnz;
    }

private:
    void CopyFromCOO_(const int* row, const int* col, const ValueType* val, const std::string& name, int64_t nnz,
                      int64_t nrow, int64_t ncol)
    {
        const bool was_accel = this->on_accel_;
        this->Clear();
        if(was_accel)
        {
            RAMD_CHECK(ramd_mat_clear(this->dev_));
            this->on_accel_ = false;
        }
        this->name_ = name;
        this->h_rp_.assign((size_t)nrow + 1, 0);
        this->h_ci_.assign((size_t)nnz, 0);
        this->h_val_.assign((size_t)nnz, ValueType(0));
        for(int64_t k = 0; k < nnz; ++k)
        {
            assert(row[k] >= 0 && row[k] < nrow);
            ++this->h_rp_[(size_t)row[k] + 1];
        }
        for(int64_t i = 0; i < nrow; ++i)
            this->h_rp_[(size_t)i + 1] += this->h_rp_[(size_t)i];
        std::vector<PtrType> cur(this->h_rp_.begin(), this->h_rp_.end() - 1);
        for(int64_t k = 0; k < nnz; ++k) // stable: storage order inside every row is kept
        {
            const PtrType p       = cur[(size_t)row[k]]++;
            this->h_ci_[(size_t)p]  = col[k];
            this->h_val_[(size_t)p] = val[k];
        }
        this->h_nrow_ = nrow;
        this->h_ncol_ = ncol;
        this->coo_input_ = true;
        if(was_accel)
            this->MoveToAccelerator();
    }

public:
    void ExtractDiagonal(LocalVector<ValueType>* vec_diag) const
    {
        this->need_accel_("ExtractDiagonal");
        if(vec_diag->GetSize() == 0)
            vec_diag->Allocate("Diagonal elements", std::min(this->GetM(), this->GetN()));
        RAMD_CHECK(ramd_mat_extract_diag(this->dev_, vec_diag->handle()));
    }
    void ExtractInverseDiagonal(LocalVector<ValueType>* vec_inv_diag) const
    {
        this->need_accel_("ExtractInverseDiagonal");
        RAMD_CHECK(ramd_mat_extract_inv_diag(this->dev_, vec_inv_diag->handle()));
    }
    void ExtractSubMatrix(int64_t row_offset, int64_t col_offset, int64_t row_size, int64_t col_size,
                          LocalMatrix<ValueType>* mat) const
    {
        this->need_accel_("ExtractSubMatrix");
        mat->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_extract_submatrix(this->dev_, (int)row_offset, (int)col_offset,
                                              (int)row_size, (int)col_size, mat->dev_));
    }
    // src/base/local_matrix.cpp:2492: mat[i][j] = rows [row_offset[i],row_offset[i+1]) x cols [...]
    void ExtractSubMatrices(int row_num_blocks, int col_num_blocks, const int* row_offset,
                            const int* col_offset, LocalMatrix<ValueType>*** mat) const
    {
        for(int i = 0; i < row_num_blocks; ++i)
            for(int j = 0; j < col_num_blocks; ++j)
                this->ExtractSubMatrix(row_offset[i], col_offset[j], row_offset[i + 1] - row_offset[i],
                                       col_offset[j + 1] - col_offset[j], mat[i][j]);
    }
    void Permute(const LocalVector<int>& permutation)
    {
        this->need_accel_("Permute");
        RAMD_CHECK(ramd_mat_permute(this->dev_, permutation.handle()));
    }
    void MultiColoring(int& num_colors, int** size_colors, LocalVector<int>* permutation) const
    {
        this->need_accel_("MultiColoring");
        assert(*size_colors == NULL);
        std::vector<int> sizes((size_t)std::max<int64_t>(this->GetM(), 1));
        permutation->MoveToAccelerator();
        if(permutation->handle() == NULL)
            permutation->Allocate("permutation", 0);
        RAMD_CHECK(ramd_mat_multicoloring(this->dev_, &num_colors, sizes.data(), permutation->handle()));
        allocate_host(num_colors, size_colors);
        std::copy(sizes.begin(), sizes.begin() + num_colors, *size_colors);
    }
    void ILU0Factorize(void)
    {
        this->need_accel_("ILU0Factorize");
        RAMD_CHECK(ramd_mat_ilu0_factorize(this->dev_));
    }
    // local_matrix.cpp:3910-4040: p = 0 -> ILU(0); level: fill levels on pattern(A^(p+1)), else ILU(0) on that pattern
    void ILUpFactorize(int p, bool level = true)
    {
        this->need_accel_("ILUpFactorize");
        assert(p >= 0);
        RAMD_CHECK(ramd_mat_ilup_factorize(this->dev_, p, level ? 1 : 0));
    }
    void LUAnalyse(void)
    {
        this->need_accel_("LUAnalyse");
        RAMD_CHECK(ramd_mat_lu_analyse(this->dev_));
    }
    void LUAnalyseClear(void)
    {
        if(this->on_accel_ && this->dev_)
            RAMD_CHECK(ramd_mat_lu_analyse_clear(this->dev_));
    }
    void LUSolve(const LocalVector<ValueType>& in, LocalVector<ValueType>* out) const
    {
        this->need_accel_("LUSolve");
        RAMD_CHECK(ramd_mat_lu_solve(this->dev_, in.handle(), out->handle()));
    }
    // ---- CSR matrix algebra (local_matrix.cpp Transpose / Sort / MatrixAdd / MatrixMult)
    void Sort(void)
    {
        this->need_accel_("Sort");
        RAMD_CHECK(ramd_mat_sort(this->dev_));
    }
    void Transpose(LocalMatrix<ValueType>* T) const
    {
        this->need_accel_("Transpose");
        assert(T != NULL && T != this);
        T->MoveToAccelerator();
        if(this->GetNnz() > 0)
            RAMD_CHECK(ramd_mat_transpose(this->dev_, T->dev_));
    }
    void Transpose(void)
    {
        if(this->GetNnz() > 0)
        {
            LocalMatrix<ValueType> tmp;
            tmp.CloneFrom(*this);
            tmp.Transpose(this);
        }
    }
    // this = alpha*this + beta*mat; structure == false: pattern(mat) is a subset of pattern(this)
    void MatrixAdd(const LocalMatrix<ValueType>& mat, ValueType alpha = static_cast<ValueType>(1),
                   ValueType beta = static_cast<ValueType>(1), bool structure = false)
    {
        this->need_accel_("MatrixAdd");
        assert(&mat != this);
        RAMD_CHECK(ramd_mat_matrix_add(this->dev_, mat.dev_, (double)alpha, (double)beta, structure ? 1 : 0));
    }
    // this = A * B
    void MatrixMult(const LocalMatrix<ValueType>& A, const LocalMatrix<ValueType>& B)
    {
        assert(&A != this && &B != this);
        A.need_accel_("MatrixMult");
        this->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_mat_mult(this->dev_, A.dev_, B.dev_));
    }
    // factorised sparse approximate inverse on the lower pattern of this matrix^power (this becomes the factor)
#ifdef RAMD_WITH_OFFSCOPE // (FSAI: out of scope)
    void FSAI(int power, const LocalMatrix<ValueType>* pattern)
    {
        this->need_accel_("FSAI");
        if(pattern != NULL)
        {
            RAMD_CHECK(ramd_mat_fsai_pattern(this->dev_, pattern->dev_));
            return;
        }
        RAMD_CHECK(ramd_mat_fsai(this->dev_, power));
    }
#endif
    // sparse approximate inverse on the pattern of this matrix (this becomes M ~ A^-1)
#ifdef RAMD_WITH_OFFSCOPE // (SPAI: out of scope)
    void SPAI(void)
    {
        this->need_accel_("SPAI");
        RAMD_CHECK(ramd_mat_spai(this->dev_));
    }
#endif
    void DiagonalMatrixMultR(const LocalVector<ValueType>& diag)
    {
        this->need_accel_("DiagonalMatrixMultR");
        RAMD_CHECK(ramd_mat_diag_mult(this->dev_, diag.handle(), 0));
    }
    void DiagonalMatrixMultL(const LocalVector<ValueType>& diag)
    {
        this->need_accel_("DiagonalMatrixMultL");
        RAMD_CHECK(ramd_mat_diag_mult(this->dev_, diag.handle(), 1));
    }
    void DiagonalMatrixMult(const LocalVector<ValueType>& diag) // deprecated alias of ...R in the reference
    {
        this->DiagonalMatrixMultR(diag);
    }
    // this = R * A * P as (R * A) * P, two MatrixMult (local_matrix.cpp:5515-5594)
    void TripleMatrixProduct(const LocalMatrix<ValueType>& R, const LocalMatrix<ValueType>& A,
                             const LocalMatrix<ValueType>& P)
    {
        assert(&R != this && &A != this && &P != this);
        LocalMatrix<ValueType> tmp;
        tmp.CloneBackend(*this);
        tmp.MatrixMult(R, A);
        this->MatrixMult(tmp, P);
    }
    // ---- unsmoothed-aggregation AMG setup, CoarseningStrategy PMIS (local_matrix.cpp:6519-6640, :6852-6930).
    // Element type of the index vectors: int (the reference: LocalVector<bool> / LocalVector<int64_t>).
    void AMGPMISAggregate(ValueType eps, LocalVector<int>* connections, LocalVector<int>* aggregates,
                          LocalVector<int>* aggregate_root_nodes) const
    {
        this->need_accel_("AMGPMISAggregate");
        assert(connections != NULL && aggregates != NULL && aggregate_root_nodes != NULL);
        connections->MoveToAccelerator();
        aggregates->MoveToAccelerator();
        aggregate_root_nodes->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_amg_pmis_aggregate(this->dev_, (double)eps, connections->handle(), aggregates->handle(),
                                               aggregate_root_nodes->handle()));
    }
    // ---- Ruge-Stueben AMG setup: PMIS C/F splitting and direct interpolation (int vectors instead of bool)
#ifdef RAMD_WITH_OFFSCOPE // (RSPMISCoarsening: out of scope)
    void RSPMISCoarsening(float eps, LocalVector<int>* CFmap, LocalVector<int>* S) const
    {
        this->need_accel_("RSPMISCoarsening");
        assert(CFmap != NULL && S != NULL);
        CFmap->MoveToAccelerator();
        S->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_rs_pmis_coarsening(this->dev_, eps, CFmap->handle(), S->handle()));
    }
#endif
#ifdef RAMD_WITH_OFFSCOPE // (RSDirectInterpolation: out of scope)
    void RSDirectInterpolation(const LocalVector<int>& CFmap, const LocalVector<int>& S, LocalMatrix<ValueType>* prolong) const
    {
        this->need_accel_("RSDirectInterpolation");
        assert(prolong != NULL && prolong != this);
        prolong->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_rs_direct_interpolation(this->dev_, CFmap.handle(), S.handle(), prolong->dev_));
    }
#endif
    // the reference's default strategy: its sequential sweep restated as a sync-free device sweep with the same result;
    // needs a symmetric strong-connection graph
    void AMGGreedyAggregate(ValueType eps, LocalVector<int>* connections, LocalVector<int>* aggregates,
                            LocalVector<int>* aggregate_root_nodes) const
    {
        this->need_accel_("AMGGreedyAggregate");
        assert(connections != NULL && aggregates != NULL && aggregate_root_nodes != NULL);
        connections->MoveToAccelerator();
        aggregates->MoveToAccelerator();
        aggregate_root_nodes->MoveToAccelerator();
        int s = ramd_mat_amg_greedy_aggregate(this->dev_, (double)eps, connections->handle(), aggregates->handle(),
                                              aggregate_root_nodes->handle());
        if(s == RAMD_ERR_UNSUPPORTED)
        {
            LOG_INFO("LocalMatrix::AMGGreedyAggregate(): " << ramd_last_error());
            FATAL_ERROR(__FILE__, __LINE__);
        }
        RAMD_CHECK(s);
    }
    void AMGSmoothedAggregation(ValueType relax, const LocalVector<int>& connections, const LocalVector<int>& aggregates,
                                const LocalVector<int>& aggregate_root_nodes, LocalMatrix<ValueType>* prolong,
                                int lumping_strat = 0) const
    {
        this->need_accel_("AMGSmoothedAggregation");
        assert(prolong != NULL && prolong != this && relax > static_cast<ValueType>(0));
        prolong->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_amg_smoothed_prolong(this->dev_, (double)relax, lumping_strat, connections.handle(),
                                                 aggregates.handle(), aggregate_root_nodes.handle(), prolong->dev_));
    }
    void AMGUnsmoothedAggregation(const LocalVector<int>& aggregates, const LocalVector<int>& aggregate_root_nodes,
                                  LocalMatrix<ValueType>* prolong) const
    {
        this->need_accel_("AMGUnsmoothedAggregation");
        assert(prolong != NULL && prolong != this);
        prolong->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_amg_unsmoothed_prolong(this->dev_, aggregates.handle(), aggregate_root_nodes.handle(),
                                                   prolong->dev_));
    }
    // TriSolverAlg_Iterative: Jacobi-sweep triangular solves (local_matrix.cpp ItLU* / ItLL* / ItL* / ItU*)
    void ItLUAnalyse(void)
    {
        this->need_accel_("ItLUAnalyse");
        RAMD_CHECK(ramd_mat_it_lu_analyse(this->dev_));
    }
    void ItLUAnalyseClear(void)
    {
        if(this->on_accel_ && this->dev_)
            RAMD_CHECK(ramd_mat_it_lu_analyse_clear(this->dev_));
    }
    void ItLUSolve(int max_iter, double tolerance, bool use_tol, const LocalVector<ValueType>& in,
                   LocalVector<ValueType>* out) const
    {
        this->need_accel_("ItLUSolve");
        RAMD_CHECK(ramd_mat_it_lu_solve(this->dev_, max_iter, tolerance, use_tol ? 1 : 0, in.handle(), out->handle()));
    }
    void ItLLAnalyse(void)
    {
        this->need_accel_("ItLLAnalyse");
        RAMD_CHECK(ramd_mat_it_ll_analyse(this->dev_));
    }
    void ItLLAnalyseClear(void)
    {
        if(this->on_accel_ && this->dev_)
            RAMD_CHECK(ramd_mat_it_ll_analyse_clear(this->dev_));
    }
    void ItLLSolve(int max_iter, double tolerance, bool use_tol, const LocalVector<ValueType>& in,
                   LocalVector<ValueType>* out) const
    {
        this->need_accel_("ItLLSolve");
        RAMD_CHECK(ramd_mat_it_ll_solve(this->dev_, max_iter, tolerance, use_tol ? 1 : 0, in.handle(), out->handle()));
    }
    void ItLLSolve(int max_iter, double tolerance, bool use_tol, const LocalVector<ValueType>& in,
                   const LocalVector<ValueType>& inv_diag, LocalVector<ValueType>* out) const
    {
        (void)inv_diag; // host_matrix_csr.cpp:1835-1843: forwarded, the sweeps invert the stored diagonal themselves
        this->ItLLSolve(max_iter, tolerance, use_tol, in, out);
    }
    void ItLAnalyse(bool diag_unit = false)
    {
        this->need_accel_("ItLAnalyse");
        RAMD_CHECK(ramd_mat_it_l_analyse(this->dev_, diag_unit ? 1 : 0));
    }
    void ItLAnalyseClear(void)
    {
        if(this->on_accel_ && this->dev_)
            RAMD_CHECK(ramd_mat_it_l_analyse_clear(this->dev_));
    }
    void ItLSolve(int max_iter, double tolerance, bool use_tol, const LocalVector<ValueType>& in,
                  LocalVector<ValueType>* out) const
    {
        this->need_accel_("ItLSolve");
        RAMD_CHECK(ramd_mat_it_l_solve(this->dev_, max_iter, tolerance, use_tol ? 1 : 0, in.handle(), out->handle()));
    }
    void ItUAnalyse(bool diag_unit = false)
    {
        this->need_accel_("ItUAnalyse");
        RAMD_CHECK(ramd_mat_it_u_analyse(this->dev_, diag_unit ? 1 : 0));
    }
    void ItUAnalyseClear(void)
    {
        if(this->on_accel_ && this->dev_)
            RAMD_CHECK(ramd_mat_it_u_analyse_clear(this->dev_));
    }
    void ItUSolve(int max_iter, double tolerance, bool use_tol, const LocalVector<ValueType>& in,
                  LocalVector<ValueType>* out) const
    {
        this->need_accel_("ItUSolve");
        RAMD_CHECK(ramd_mat_it_u_solve(this->dev_, max_iter, tolerance, use_tol ? 1 : 0, in.handle(), out->handle()));
    }
    // incomplete Cholesky on the lower part incl. diagonal (local_matrix.cpp ICFactorize / LLAnalyse / LLSolve)
    void ICFactorize(LocalVector<ValueType>* inv_diag)
    {
        this->need_accel_("ICFactorize");
        assert(inv_diag != NULL);
        RAMD_CHECK(ramd_mat_ic_factorize(this->dev_, inv_diag->handle()));
    }
    void LLAnalyse(void)
    {
        this->need_accel_("LLAnalyse");
        RAMD_CHECK(ramd_mat_ll_analyse(this->dev_));
    }
    void LLAnalyseClear(void)
    {
        if(this->on_accel_ && this->dev_)
            RAMD_CHECK(ramd_mat_ll_analyse_clear(this->dev_));
    }
    void LLSolve(const LocalVector<ValueType>& in, LocalVector<ValueType>* out) const
    {
        // the reference's two-argument LLSolve is not implemented by its own backends either
        // (host_matrix_csr.cpp:1288-1292 returns false); IC uses the inverse-diagonal form below
        (void)in;
        (void)out;
        LOG_INFO("LocalMatrix::LLSolve(in, out): use LLSolve(in, inv_diag, out)");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    void LLSolve(const LocalVector<ValueType>& in, const LocalVector<ValueType>& inv_diag,
                 LocalVector<ValueType>* out) const
    {
        this->need_accel_("LLSolve");
        RAMD_CHECK(ramd_mat_ll_solve(this->dev_, in.handle(), inv_diag.handle(), out->handle()));
    }
    void LAnalyse(bool diag_unit = false)
    {
        this->need_accel_("LAnalyse");
        RAMD_CHECK(ramd_mat_l_analyse(this->dev_, diag_unit ? 1 : 0));
    }
    void LAnalyseClear(void)
    {
        if(this->on_accel_ && this->dev_)
            RAMD_CHECK(ramd_mat_l_analyse_clear(this->dev_));
    }
    void LSolve(const LocalVector<ValueType>& in, LocalVector<ValueType>* out) const
    {
        this->need_accel_("LSolve");
        RAMD_CHECK(ramd_mat_l_solve(this->dev_, in.handle(), out->handle()));
    }
    void UAnalyse(bool diag_unit = false)
    {
        this->need_accel_("UAnalyse");
        RAMD_CHECK(ramd_mat_u_analyse(this->dev_, diag_unit ? 1 : 0));
    }
    void UAnalyseClear(void)
    {
        if(this->on_accel_ && this->dev_)
            RAMD_CHECK(ramd_mat_u_analyse_clear(this->dev_));
    }
    void USolve(const LocalVector<ValueType>& in, LocalVector<ValueType>* out) const
    {
        this->need_accel_("USolve");
        RAMD_CHECK(ramd_mat_u_solve(this->dev_, in.handle(), out->handle()));
    }

    // extension: device-side synthetic operator (3-D 7-point Poisson N^3)
    void GeneratePoisson7(int N)
    {
        this->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_gen_poisson7(this->dev_, N));
    }
    // extension: the reference's own 3-D test operator generated on the device (the 27-point Laplacian of gen_3d_laplacian,
    // clients/include/utility.hpp:110-177, there a cube: nx = ny = nz = ndim)
    void GenerateLaplace27(int nx, int ny, int nz)
    {
        this->MoveToAccelerator();
        RAMD_CHECK(ramd_mat_gen_laplace27(this->dev_, nx, ny, nz));
    }
    // extension: value-cast copy used by MixedPrecisionDC (mixed_precision.cpp:201-229)
    template <typename OtherType>
    void CastFrom(const LocalMatrix<OtherType>& src)
    {
        this->Clear();
        if(this->dev_ && this->own_)
            ramd_mat_destroy(this->dev_);
        this->dev_ = NULL;
        RAMD_CHECK(ramd_mat_cast(src.handle(), &this->dev_));
        this->own_      = true;
        this->on_accel_ = true;
    }

private:
    template <typename>
    friend class LocalMatrix;

    void ensure_dev_(void)
    {
        if(!this->dev_)
        {
            RAMD_CHECK(ramd_mat_create(_dtype<ValueType>::value, &this->dev_));
            this->own_ = true;
        }
    }
    void upload_(void)
    {
        this->ensure_dev_();
        RAMD_CHECK(ramd_mat_set_csr_from_host(this->dev_, (int)this->h_nrow_, (int)this->h_ncol_,
                                              (int64_t)this->h_ci_.size(), this->h_rp_.data(),
                                              this->h_ci_.data(), this->h_val_.data()));
        std::vector<PtrType>().swap(this->h_rp_);
        std::vector<int>().swap(this->h_ci_);
        std::vector<ValueType>().swap(this->h_val_);
        this->h_nrow_ = this->h_ncol_ = 0;
        if(this->coo_input_)
        {
            this->coo_input_ = false;
            RAMD_CHECK(ramd_mat_convert(this->dev_, (int)COO));
        }
    }
    void dev_info_(int* nr, int* nc, int64_t* nnz, int* fmt) const
    {
        if(!this->dev_)
        {
            if(nr)
                *nr = 0;
            if(nc)
                *nc = 0;
            if(nnz)
                *nnz = 0;
            if(fmt)
                *fmt = CSR;
            return;
        }
        RAMD_CHECK(ramd_mat_info(this->dev_, nr, nc, nnz, fmt, NULL));
    }
    void need_accel_(const char* op) const
    {
        if(!this->on_accel_)
        {
            LOG_INFO("LocalMatrix::" << op << "() on a host object: this library has no host compute "
                                     << "backend - call MoveToAccelerator() first");
            FATAL_ERROR(__FILE__, __LINE__);
        }
    }

    std::string            name_;
    std::vector<PtrType>   h_rp_;
    std::vector<int>       h_ci_;
    std::vector<ValueType> h_val_;
    int64_t                h_nrow_ = 0, h_ncol_ = 0;
    ramd_mat_t             dev_      = NULL;
    bool                   own_      = true;
    bool                   on_accel_ = false;
    bool                   coo_input_ = false; // host data came in as COO: becomes a COO object on the device
};

} // namespace rocalution
