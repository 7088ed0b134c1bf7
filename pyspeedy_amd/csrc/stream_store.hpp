// Stores of values that a LATER kernel consumes (never this one): written with the non-temporal hint so that the output
// stream does not displace the tables and shared inputs other workgroups keep hitting in the L2.  Measured on MI355X: +15 % on
// the spectral -> grid kernel, -2.4 % per step for the column / spectral / coupler kernels together.
#pragma once
#include <hip/hip_runtime.h>

namespace spd {
template <typename T, typename V>
__device__ __forceinline__ void stream_store(T *p, V v) {
    __builtin_nontemporal_store(static_cast<T>(v), p);
}
// ... and loads of values this launch reads exactly once and nobody else in it reads (grid fields coming from the previous
// kernel): the same hint on the way in (-0.6 % ... -0.9 % per step; NOT for inputs several workgroups share).
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) {
    return __builtin_nontemporal_load(p);
}
}  // namespace spd
