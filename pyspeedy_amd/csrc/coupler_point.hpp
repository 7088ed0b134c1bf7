// One grid point of the per-step land / sea-ice coupling (couple_land_atm + run_land_model, land_model.f90:151-215;
// couple_sea_atm + run_sea_model, sea_model.f90:193-383), shared by the stand-alone coupler_kernel (surface.hip) and the
// tail blocks of spectral_step_kernel (dynamics.hip), which carry the coupling of small ensembles inside the last launch
// of the step.
#pragma once
#include <hip/hip_runtime.h>

#include "stream_store.hpp"
#include "surface.hpp"

namespace spd {

// what the coupling that follows a step needs besides the arrays: interpolation weights of the date after the step, run control
struct CouplerArgs {
    SurfacePtrs S;
    TimeInterp w;
    int first, count, day, land_coupling, sst_anomaly, anom_planes, fresh;
};

namespace cpl {
constexpr int NGc = IX * IL;
__device__ constexpr double SBCd = 5.67e-8f, ALHCd = 2501.0f, EMISFCd = 0.98f, ALBSEAd = 0.07f, ALBICEd = 0.60f;

__device__ inline double forin5(const double *f12, size_t p, const TimeInterp &w) {
    return w.w5[0] * f12[p + static_cast<size_t>(NGc) * w.m5[0]] + w.w5[1] * f12[p + static_cast<size_t>(NGc) * w.m5[1]] +
           w.w5[2] * f12[p + static_cast<size_t>(NGc) * w.m5[2]] + w.w5[3] * f12[p + static_cast<size_t>(NGc) * w.m5[3]] +
           w.w5[4] * f12[p + static_cast<size_t>(NGc) * w.m5[4]];
}
__device__ inline double forint(const double *f12, size_t p, const TimeInterp &w) {
    const double a = f12[p + static_cast<size_t>(NGc) * w.l0];
    return a + w.wlin * (f12[p + static_cast<size_t>(NGc) * w.l1] - a);
}
}  // namespace cpl


// `fresh` = the monthly climatologies are interpolated to the model date.  The interpolation weights depend on the month and
// the day only (model_control.f90:162-185), so between two midnights the reference recomputes, step after step, values it
// already holds in stlcl_obs, snowdcl_obs, soilwcl_obs, sstcl_ob, sicecl_ob, ticecl_ob, sstan_ob (and copies of them in
// snow_depth, soil_avail_water, sice_om, sice_am, sstan_am).  With fresh == 0 the kernel reads those stored values instead of
// the 16 climatology / anomaly planes and does not store them again: bitwise the same state, 29 instead of 55 doubles moved
// per column.  The host passes fresh != 0 on the first coupling of a day and after anything wrote to the state (model.hip).
__device__ __forceinline__ void coupler_point(const SurfacePtrs &S, int mem, int p, const TimeInterp &w, int day, int land_coupling,
                                              int sst_anomaly, int anom_planes, int fresh) {
    using namespace cpl;
    constexpr int NG = NGc;
    const size_t o = static_cast<size_t>(mem) * NG + p, o12 = static_cast<size_t>(mem) * 12 * NG + p;
    // ---- land (land_model.f90:151-215)
    double stlcl;
    if (fresh) {
        stlcl = forin5(S.stl12, o12, w);
        const double snowdcl = forint(S.snowd12, o12, w);
        const double soilwcl = forint(S.soilw12, o12, w);
        stream_store(&S.stlcl_obs[o], stlcl);
        stream_store(&S.snowdcl_obs[o], snowdcl);
        stream_store(&S.soilwcl_obs[o], soilwcl);
        stream_store(&S.snow_depth[o], snowdcl);
        stream_store(&S.soil_avail_water[o], soilwcl);
    } else {
        stlcl = S.stlcl_obs[o];
    }
    if (day == 0) {
        stream_store(&S.stl_lm[o], stlcl);
        stream_store(&S.land_temp[o], stlcl);
    } else if (land_coupling) {
        double tanom = S.stl_lm[o] - stlcl;
        tanom = S.cdland[o] * (tanom + S.rhcapl[o] * S.hfluxn[static_cast<size_t>(mem) * 3 * NG + p]);
        const double stl = tanom + stlcl;
        stream_store(&S.stl_lm[o], stl);
        stream_store(&S.land_temp[o], stl);
    } else if (fresh) {
        stream_store(&S.land_temp[o], stlcl);
    }

    // ---- sea (sea_model.f90:193-316)
    const double sstfr = 273.2f - 1.8f;  // single-precision subtraction, sea_model.f90:229
    double sstcl, sicecl, ticecl, sstan_ob = S.sstan_ob[o];
    if (fresh) {
        sstcl = forin5(S.sst12, o12, w);
        sicecl = forint(S.sea_ice_frac12, o12, w);
        if (sst_anomaly) {
            const size_t oa = static_cast<size_t>(mem) * anom_planes * NG + p;
            const double a = S.sst_anom[oa + static_cast<size_t>(NG) * w.a0];
            sstan_ob = a + w.wan * (S.sst_anom[oa + static_cast<size_t>(NG) * w.a1] - a);
            stream_store(&S.sstan_ob[o], sstan_ob);
        }
        if (sstcl > sstfr) {
            sicecl = fmin(0.5, sicecl);
            ticecl = sstfr;
            if (sicecl > 0.0) sstcl = sstfr + (sstcl - sstfr) / (1.0f - sicecl);
        } else {
            sicecl = fmax(0.5, sicecl);
            ticecl = sstfr + (sstcl - sstfr) / sicecl;
            sstcl = sstfr;
        }
        stream_store(&S.sstcl_ob[o], sstcl);
        stream_store(&S.sicecl_ob[o], sicecl);
        stream_store(&S.ticecl_ob[o], ticecl);
    } else {
        sstcl = S.sstcl_ob[o];
        sicecl = S.sicecl_ob[o];
        ticecl = S.ticecl_ob[o];
    }
    double sst_om, tice_om, sice_om;
    if (day == 0) {
        sst_om = 0.0;  // sea_coupling_flag <= 0 (sea_model.f90:261)
        tice_om = ticecl;
        sice_om = sicecl;
    } else {  // run_sea_model, sea_model.f90:318-383 (ice_coupling_flag = 1)
        const double tice_am = S.tice_am[o], sice_am = S.sice_am[o];
        const double hfl2 = S.hfluxn[static_cast<size_t>(mem) * 3 * NG + NG + p];
        tice_om = S.tice_om[o];
        sst_om = S.sst_om[o];
        const double s4 = (sstfr * sstfr) * (sstfr * sstfr), t4 = (tice_am * tice_am) * (tice_am * tice_am);
        const double difice = (ALBSEAd - ALBICEd) * S.ssrd[o] + EMISFCd * SBCd * (s4 - t4) +
                              S.shf[static_cast<size_t>(mem) * 3 * NG + NG + p] + S.evap[static_cast<size_t>(mem) * 3 * NG + NG + p] * ALHCd;
        const double hflux_i = hfl2 + difice * (1.0f - sice_am);
        double hflux = hfl2 - S.hfseacl[o] - sicecl * (hflux_i + 1.0 * (sstfr - tice_om));
        double tanom = sst_om - sstcl;
        tanom = S.cdsea[o] * (tanom + S.rhcaps[o] * hflux);
        sst_om = tanom + sstcl;
        hflux = hflux_i + 1.0 * (sstfr - tice_om);
        tanom = tice_om - ticecl;
        const double anom0 = 20.f;
        const double cdis = S.cdice[o] * (anom0 / (anom0 + fabs(tanom)));
        tanom = cdis * (tanom + S.rhcapi[o] * hflux);
        tice_om = tanom + ticecl;
        sice_om = sicecl;
    }
    stream_store(&S.sst_om[o], sst_om);
    stream_store(&S.tice_om[o], tice_om);
    const double sstan_am = sst_anomaly ? sstan_ob : 0.0;
    if (fresh) {  // (unchanged until the next fresh coupling: sice_om = sice_am = sicecl_ob, sstan_am = sstan_ob or 0)
        stream_store(&S.sice_om[o], sice_om);
        stream_store(&S.sstan_am[o], sstan_am);
        stream_store(&S.sice_am[o], sice_om);
    }
    double sst_am = sstcl + sstan_am;
    stream_store(&S.tice_am[o], tice_om);
    sst_am = sst_am + sice_om * (tice_om - sst_am);
    stream_store(&S.sst_am[o], sst_am);
    stream_store(&S.ssti_om[o], sst_om + sice_om * (tice_om - sst_om));
}

}  // namespace spd
