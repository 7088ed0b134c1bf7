// The range check of diagnostics.f90:16-76 for ONE member, by one workgroup of 512 threads (one wavefront per level): shared by
// the stand-alone diagnostics_kernel (dynamics.hip) and by the tail blocks of the spectral -> grid launch of the NEXT step
// (transforms.hip: spec2grid_table_check_kernel), which carry the check of a host that collects it one step late
// (spd_model_check_defer) -- the check then costs no launch of its own and no time on the step's stream.
// Writes err[member] = 4 * ticket + (1 if out of range) -- always, so that the caller does not have to clear it first.  `err` is
// pinned host memory: the ticket of the launch travels with every code, so the host can tell a fresh code from what an earlier
// launch left there by looking at the memory alone, the moment the store lands (model.hip: wait_codes), instead of waiting for a
// completion event behind the kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "device_tables.hpp"

namespace spd {

struct CheckArgs {
    const double *vor, *div, *t;  // [M][2][8][992] complex
    int tl;                       // time level (0-based)
    int *err;                     // [M], pinned host memory
    double *diag;                 // [M][3][8] or nullptr
    int ticket;
    int first = 0;                // block i of a launch checks member first + i (err and diag are indexed by the member itself)
};

// kBatch: how many of a lane's 16 rounds of loads are requested before anything is summed (the order of the sum is the same for
// every value).  The stand-alone kernel is one dependent chain per wavefront and asks for 8 at a time; the blocks that ride in
// the transform launch take 2: that launch lives on 62 registers per lane and 8 wavefronts per SIMD, and 8 rounds in flight
// would cost it half of them.
template <int kBatch = 8>
__device__ __forceinline__ void diagnostics_block(const CheckArgs &c, const DeviceTables &T, int mem) {
    using d2 = double __attribute__((ext_vector_type(2)));
    __shared__ int bad[KX];
    const int l = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t so = ((static_cast<size_t>(mem) * 2 + c.tl) * 8 + l) * NSPEC;
    const d2 *vor = reinterpret_cast<const d2 *>(c.vor) + so, *div = reinterpret_cast<const d2 *>(c.div) + so;
    double d1 = 0.0, d2s = 0.0;
    // (the 16 rounds of a lane are requested in two batches of 8 before anything is summed: the kernel is one dependent chain per
    // wavefront, launched once per model step by hosts with the reference's loop; the order of the sum is unchanged)
    constexpr int kRounds = (NSPEC + 63) / 64;
    static_assert(kRounds % kBatch == 0, "full batches");
#pragma unroll
    for (int r0 = 0; r0 < kRounds; r0 += kBatch) {
        d2 a[kBatch], b[kBatch];
        double e[kBatch];
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {
            const int k = lane + 64 * (r0 + r), kc = k < NSPEC ? k : NSPEC - 1;
            e[r] = T.elm2[kc];
            a[r] = vor[kc];
            b[r] = div[kc];
        }
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {
            const int k = lane + 64 * (r0 + r);
            if (k >= NSPEC || k % MX == 0) continue;  // m = 1 (zonal mean) is excluded: only the eddies count
            // temp = -x * elm2 ; diag -= real(temp * conjg(x))
            d1 = d1 - ((-a[r].x * e[r]) * a[r].x + (-a[r].y * e[r]) * a[r].y);
            d2s = d2s - ((-b[r].x * e[r]) * b[r].x + (-b[r].y * e[r]) * b[r].y);
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        d1 += __shfl_down(d1, s, 64);
        d2s += __shfl_down(d2s, s, 64);
    }
    if (lane == 0) {
        const double tmean = 0.707106769084930420 /* sqrt(0.5) in fp32 */ * c.t[2 * so];
        if (c.diag) {
            double *dg = c.diag + static_cast<size_t>(mem) * KX * 3;
            dg[l] = d1;
            dg[l + KX] = d2s;
            dg[l + 2 * KX] = tmean;
        }
        bad[l] = (d1 > 500.0f || d2s > 500.0f || tmean < 180.0f || tmean > 320.0f) ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int any = 0;
#pragma unroll
        for (int k = 0; k < KX; ++k) any |= bad[k];
        __hip_atomic_store(c.err + mem, 4 * c.ticket + (any ? 1 : 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace spd
