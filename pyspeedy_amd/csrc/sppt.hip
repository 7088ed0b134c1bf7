// Stochastically perturbed parametrisation tendencies (SPPT; Palmer et al. 2009, ECMWF Tech. Memo 598) -- the scheme
// sppt.f90 of the reference sets out to implement: an AR(1) process per spectral coefficient and level,
//     r(t+1) = phi r(t) + sigma(n) eta,   eta complex standard normal clipped at +-10,   r(0) = sigma eta / sqrt(1 - phi^2),
//     phi = exp(-(24 / nsteps) / 6 h)                                   (sppt.f90:27-30)
//     sigma(m, n) = f0 exp(-L^2 el2(m, n) / 4),  L = 500 km             (sppt.f90:33, 84-89)
//     f0 = sqrt(stddev^2 (1 - phi^2) / (2 sum_{n=1}^{trunc} (2n + 1) exp(-(L / a)^2 n (n + 1) / 2))),  stddev = 0.33 (:36)
// transformed to the grid (spec2grid, kcos = 1), clipped to +-1 and applied to the physical part of the u, v, T, q tendencies
// as (1 + r mu(k)) (tend - tend_dyn) + tend_dyn with mu = 1 (physics.f90:234-248).
//
// PARITY UNPINNED: the reference never runs this code (sppt_on is a compile-time .false., params.f90:44) and as written it
// cannot work -- the spectral pattern is a local allocatable freed on every call, so the AR(1) memory is lost, the result
// is deallocated before it is returned, and the generator is seeded from the wall clock.  What is built here is the scheme
// as documented, with a DETERMINISTIC counter-based generator: the normal pair of (seed, global member id, step, level,
// coefficient) is a pure function of those integers (splitmix64 finaliser -> Box-Muller), so a run is reproducible
// whatever the sharding of the members over GPUs.  oracle/sppt_oracle.py restates it in numpy; tests check that
// restatement, the AR(1) statistics and the tendency formula.
#include <hip/hip_runtime.h>

#include <cmath>

#include "device_tables.hpp"
#include "sppt_point.hpp"
#include "launch_events.hpp"

namespace spd {

namespace {
constexpr int kT = 256;

// one lane per (member, level, coefficient): AR(1) update of the spectral pattern (sppt_point.hpp)
__global__ __launch_bounds__(kT) void sppt_update_kernel(SpptArgs a) {
    sppt_update_point(a, static_cast<long>(blockIdx.x) * kT + threadIdx.x);
}
}  // namespace

// constants of the scheme from the reference's literals (default-real ones in fp32, as flang evaluates them)
void sppt_constants(double *phi, double *f0, double *quarter_len2) {
    const double time_decorr = 6.0f, len_decorr = 500000.0f, stddev = 0.33f, rearth = 6.371e+6f;
    const double ph = std::exp(-(24.0 / 36.0) / time_decorr);
    double sum = 0.0;
    for (int n = 1; n <= TRUNC; ++n)
        sum += (2 * n + 1) * std::exp(-0.5 * (len_decorr / rearth) * (len_decorr / rearth) * n * (n + 1));
    *phi = ph;
    *f0 = std::sqrt((stddev * stddev * (1.0 - ph * ph)) / (2.0 * sum));
    *quarter_len2 = 0.25 * len_decorr * len_decorr;
}

SpptArgs sppt_args(double *spec, const DeviceTables &T, int M, unsigned long long seed, long long member_base, long long step,
                   int first) {
    SpptArgs a{};
    a.spec = spec, a.el2 = T.el2, a.M = M, a.first = first, a.seed = seed, a.member_base = member_base, a.step = step;
    sppt_constants(&a.phi, &a.f0, &a.quarter_len2);
    return a;
}

hipError_t run_sppt_update(const SpptArgs &a, hipStream_t s) {
    const long n = static_cast<long>(a.M) * KX * NSPEC;
    launch(sppt_update_kernel, dim3(static_cast<unsigned>((n + kT - 1) / kT)), dim3(kT), 0, s, a);
    return hipGetLastError();
}

}  // namespace spd
