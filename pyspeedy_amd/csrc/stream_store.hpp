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
}  // namespace spd
