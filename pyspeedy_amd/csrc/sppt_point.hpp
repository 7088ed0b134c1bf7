// One (member, level, coefficient) of the SPPT pattern update (see sppt.hip for the scheme and its parity status): shared by
// sppt_update_kernel and by the launch of geopotential_kernel that carries the update in its tail blocks (dynamics.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_tables.hpp"

namespace spd {

struct SpptArgs {
    double *spec;       // [M][8][992] complex AR(1) state
    const double *el2;  // n (n + 1) / a^2 per coefficient
    int M, first;       // members in the array; first != 0: draw the stationary initial state instead of an AR(1) step
    unsigned long long seed;
    long long member_base, step;
    double phi, f0, quarter_len2;
};

__host__ __device__ inline unsigned long long sppt_mix64(unsigned long long z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// gid = (member * 8 + level) * 992 + coefficient
__device__ __forceinline__ void sppt_update_point(const SpptArgs &a, long gid) {
    using d2 = double __attribute__((ext_vector_type(2)));
    if (gid >= static_cast<long>(a.M) * KX * NSPEC) return;
    d2 *spec = reinterpret_cast<d2 *>(a.spec);
    const int idx = static_cast<int>(gid % NSPEC);
    const long mk = gid / NSPEC;
    const int k = static_cast<int>(mk % KX);
    const unsigned long long member = static_cast<unsigned long long>(a.member_base + mk / KX);
    const unsigned long long counter =
        (member << 40) | (static_cast<unsigned long long>(a.step) << 14) | static_cast<unsigned long long>(k * NSPEC + idx);
    const unsigned long long h1 = sppt_mix64(a.seed ^ sppt_mix64(counter));
    const unsigned long long h2 = sppt_mix64(h1 + 0x9E3779B97F4A7C15ull);
    const double u1 = (static_cast<double>(h1 >> 11) + 1.0) * 0x1.0p-53;  // (0, 1]
    const double u2 = static_cast<double>(h2 >> 11) * 0x1.0p-53;          // [0, 1)
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586 * u2;
    double er = rad * cos(ang), ei = rad * sin(ang);
    er = fmin(10.0, fabs(er)) * (er < 0.0 ? -1.0 : 1.0);  // sppt.f90:70-74
    ei = fmin(10.0, fabs(ei)) * (ei < 0.0 ? -1.0 : 1.0);
    const double sigma = a.f0 * exp(-a.quarter_len2 * a.el2[idx]);
    d2 r;
    if (a.first) {
        const double s0 = sigma / sqrt(1.0 - a.phi * a.phi);
        r = d2{s0 * er, s0 * ei};
    } else {
        const d2 old = spec[gid];
        r = d2{a.phi * old.x + sigma * er, a.phi * old.y + sigma * ei};
    }
    spec[gid] = r;
}

}  // namespace spd
