// trsv_handoff.hpp -- the data-tagged hand-off of the sync-free triangular solves (trisolve.hip, trsv_syncfree.hip): a solution
// array pre-filled with a NaN sentinel, every finished value published by ONE agent-scope store and consumed by polling the
// array itself (MI355X_MICROARCH.md "handoff-1to1": an 8-byte granule, no flag, no fence).
#pragma once

#include "device_utils.hpp"

namespace ramd
{

template <typename T>
struct Sentinel;
template <>
struct Sentinel<double>
{
    using bits = unsigned long long;
    static constexpr bits value = 0x7FF8DEADBEEF0001ull; // quiet NaN with a private payload
    __device__ static __forceinline__ bits as_bits(double v)
    {
        return (bits)__double_as_longlong(v);
    }
    __device__ static __forceinline__ double from_bits(bits b)
    {
        return __longlong_as_double((long long)b);
    }
};
template <>
struct Sentinel<float>
{
    using bits = unsigned int;
    static constexpr bits value = 0x7FDEAD01u;
    __device__ static __forceinline__ bits as_bits(float v)
    {
        return __float_as_uint(v);
    }
    __device__ static __forceinline__ float from_bits(bits b)
    {
        return __uint_as_float(b);
    }
};

template <typename T>
__device__ __forceinline__ typename Sentinel<T>::bits poll_load(const T* p)
{
    using B = typename Sentinel<T>::bits;
    return __hip_atomic_load(reinterpret_cast<const B*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void publish(T* p, T v)
{
    using B = typename Sentinel<T>::bits;
    __hip_atomic_store(reinterpret_cast<B*>(p), Sentinel<T>::as_bits(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_fill_sentinel(int64_t n, T* __restrict__ w)
{
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const T       s   = Sentinel<T>::from_bits(Sentinel<T>::value);
    for(int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gsz)
        w[i] = s;
}

// value of lane - 1 of the same 16-lane row (DPP row_shr:1): the hand-over of a running sum from lane to lane of a row
template <typename T>
__device__ __forceinline__ T lane_before_in_row(T v);
template <>
__device__ __forceinline__ double lane_before_in_row<double>(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ float lane_before_in_row<float>(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x111, 0xf, 0xf, true));
}

} // namespace ramd
