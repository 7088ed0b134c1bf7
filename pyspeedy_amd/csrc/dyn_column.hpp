// Grid-point dynamics of one column (tendencies.f90:125-224) and the products the forward transforms need (:242-266),
// shared by the stand-alone kernel of dynamics.hip and the fused dynamics + physics kernel of physics.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "model.hpp"
#include "stream_store.hpp"
#include "vertical_consts.hpp"

namespace spd {

// Column p (0 .. ix*il-1, latitude row j) of member `mem`.  Stores utend / vtend above the lowest level, the kinetic energy
// and flux products and the surface-pressure tendency; RETURNS the temperature / tracer tendencies of all levels and the
// wind tendencies of the lowest level in registers (the physics adds to exactly those) -- or stores them too when STORE_ALL.
// `before_products` is called once the tendencies are formed, in front of the 40 product stores: the fused kernel issues the
// loads of its next phase there, so that they travel while the stores are issued (nothing, for the stand-alone kernel).
// (dhs, dhsr, fsgr, tref, tref3: the compile-time tables of vertical_consts.hpp, not D's copies -- see physics.hip: ColTables)
struct NoPrefetch {
    __device__ void operator()() const {}
};
template <bool STORE_ALL, typename F = NoPrefetch>
__device__ __forceinline__ void dyn_column(const ModelPtrs &P, const DynDeviceTables &D, int mem, int p, int j,
                                           double (&ttend)[KX], double (&trtend)[KX], double &utend_kx, double &vtend_kx,
                                           F before_products = F{}) {
    constexpr int NG = IX * IL;
    constexpr double AKAPd = 2.0f / 7.0f, RGASd = AKAPd * static_cast<double>(1004.0f);
    const size_t o3 = static_cast<size_t>(mem) * KX * NG + p, o2 = static_cast<size_t>(mem) * NG + p;
    double ug[KX], vg[KX], tg[KX], trg[KX], vorg[KX], divg[KX];
    const double cor = D.coriol[j];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        ug[k] = stream_load(&P.ug2[o3 + NG * k]);
        vg[k] = stream_load(&P.vg2[o3 + NG * k]);
        tg[k] = stream_load(&P.tg2[o3 + NG * k]);
        trg[k] = stream_load(&P.trg2[o3 + NG * k]);
        vorg[k] = stream_load(&P.vorg[o3 + NG * k]);
        divg[k] = stream_load(&P.divg[o3 + NG * k]);
    }
    const double px = stream_load(&P.px[o2]), py = stream_load(&P.py[o2]);
    // all 50 loads are issued before the first value is used (left to itself the scheduler keeps ~14 in flight and the column
    // waits for memory four times instead of once)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < KX; ++k) vorg[k] = vorg[k] + cor;  // absolute vorticity
    double umean = 0.0, vmean = 0.0, dmean = 0.0;
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        umean = umean + ug[k] * vc::dhs[k];
        vmean = vmean + vg[k] * vc::dhs[k];
        dmean = dmean + divg[k] * vc::dhs[k];
    }
    stream_store(&P.psdtg[o2], -umean * px - vmean * py);
    double puv[KX], sigdt[KX + 1], sigm[KX + 1], tgg[KX], temp[KX + 1];
    sigdt[0] = 0.0;
    sigm[0] = 0.0;
#pragma unroll
    for (int k = 0; k < KX; ++k) puv[k] = (ug[k] - umean) * px + (vg[k] - vmean) * py;
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        sigdt[k + 1] = sigdt[k] - vc::dhs[k] * (puv[k] + divg[k] - dmean);
        sigm[k + 1] = sigm[k] - vc::dhs[k] * puv[k];
    }
    // (tendencies.f90:153-156 zeroes level kx+1 BEFORE this loop; the loop's last iteration stores it again, so the
    //  value used below is the accumulated one, ~1e-17, exactly as in the reference)
#pragma unroll
    for (int k = 0; k < KX; ++k) tgg[k] = tg[k] - vc::tref[k];
    temp[0] = 0.0;
    temp[KX] = 0.0;
    // zonal wind
#pragma unroll
    for (int k = 1; k < KX; ++k) temp[k] = sigdt[k] * (ug[k] - ug[k - 1]);
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const double v = vg[k] * vorg[k] - tgg[k] * RGASd * px - (temp[k + 1] + temp[k]) * vc::dhsr[k];
        if (STORE_ALL || k < KX - 1) stream_store(&P.utend[o3 + NG * k], v);
        if (k == KX - 1) utend_kx = v;
    }
    // meridional wind
#pragma unroll
    for (int k = 1; k < KX; ++k) temp[k] = sigdt[k] * (vg[k] - vg[k - 1]);
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const double v = -ug[k] * vorg[k] - tgg[k] * RGASd * py - (temp[k + 1] + temp[k]) * vc::dhsr[k];
        if (STORE_ALL || k < KX - 1) stream_store(&P.vtend[o3 + NG * k], v);
        if (k == KX - 1) vtend_kx = v;
    }
    // temperature
#pragma unroll
    for (int k = 1; k < KX; ++k) temp[k] = sigdt[k] * (tgg[k] - tgg[k - 1]) + sigm[k] * (vc::tref[k] - vc::tref[k - 1]);
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        ttend[k] = tgg[k] * divg[k] - (temp[k + 1] + temp[k]) * vc::dhsr[k] + vc::fsgr[k] * tgg[k] * (sigdt[k + 1] + sigdt[k]) +
                   vc::tref3[k] * (sigm[k + 1] + sigm[k]) + AKAPd * (tg[k] * puv[k] - tgg[k] * dmean);
        if (STORE_ALL) stream_store(&P.ttend[o3 + NG * k], ttend[k]);
    }
    // tracer
#pragma unroll
    for (int k = 1; k < KX; ++k) temp[k] = sigdt[k] * (trg[k] - trg[k - 1]);
    temp[1] = 0.0;
    temp[2] = 0.0;
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        trtend[k] = trg[k] * divg[k] - (temp[k + 1] + temp[k]) * vc::dhsr[k];
        if (STORE_ALL) stream_store(&P.trtend[o3 + NG * k], trtend[k]);
    }
    before_products();
    // inputs of the forward transforms (tendencies.f90:247-266)
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        stream_store(&P.keg[o3 + NG * k], 0.5f * (ug[k] * ug[k] + vg[k] * vg[k]));
        stream_store(&P.utg[o3 + NG * k], -ug[k] * tgg[k]);
        stream_store(&P.vtg[o3 + NG * k], -vg[k] * tgg[k]);
        stream_store(&P.uqg[o3 + NG * k], -ug[k] * trg[k]);
        stream_store(&P.vqg[o3 + NG * k], -vg[k] * trg[k]);
    }
}

}  // namespace spd
