// Per-step land / sea coupling and daily forcing on the device, one lane per (member, grid point).
//   coupler_kernel   couple_land_atm + run_land_model (land_model.f90:151-215), couple_sea_atm + run_sea_model
//                    (sea_model.f90:193-383).  In this version of the reference the coupler runs EVERY step
//                    (speedy.f90:72), with monthly climatologies interpolated to the current date.
//   forcing_kernel   set_forcing (forcing.f90:15-102): snow cover, albedos, zonal radiation fields and the two grid
//                    fields whose spectral transforms are the orographic diffusion corrections tcorh / qcorh.
//   rest-state helpers for initialize_from_rest_state (prognostics.f90:29-120).
#include <hip/hip_runtime.h>

#include "surface.hpp"
#include "stream_store.hpp"

namespace spd {
namespace {
constexpr int NG = IX * IL;
constexpr int kT = 256;
__device__ constexpr double ALBSNd = 0.60f;

__device__ inline double qsat_p(double ta, double pr) {  // humidity.f90:44-78 with P = pr
    const double e0 = 6.108e-3, c1 = 17.269f, c2 = 21.875f, t0 = 273.16f, t1 = 35.86f, t2 = 7.66f;
    const double e = (ta >= t0) ? e0 * exp(c1 * (ta - t0) / (ta - t1)) : e0 * exp(c2 * (ta - t0) / (ta - t2));
    return 622.0f * e / (pr - 0.378f * e);
}
}  // namespace

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


__global__ __launch_bounds__(kT) void coupler_kernel(SurfacePtrs S, int first, int count, TimeInterp w, int day, int land_coupling,
                                                     int sst_anomaly, int anom_planes, int fresh) {
    const int gid = blockIdx.x * kT + threadIdx.x;
    if (gid >= count * NG) return;
    const int lm = gid / NG, mem = first + lm, p = gid - lm * NG;
    coupler_point(S, mem, p, w, day, land_coupling, sst_anomaly, anom_planes, fresh);
}

__global__ __launch_bounds__(kT) void forcing_kernel(SurfacePtrs S, int first, int count, ZonalDevice Z, double gamlat,
                                                     double *corh_t, double *corh_q) {
    const int gid = blockIdx.x * kT + threadIdx.x;
    if (gid >= count * NG) return;
    const int lm = gid / NG, mem = first + lm, p = gid - lm * NG, j = p / IX;
    const size_t o = static_cast<size_t>(mem) * NG + p;
    // zonally averaged radiation fields (shortwave_radiation.f90:256-274)
    S.flux_solar_in[o] = Z.v[0][j];
    S.flux_ozone_upper[o] = Z.v[1][j];
    S.flux_ozone_lower[o] = Z.v[2][j];
    S.zenit_correction[o] = Z.v[3][j];
    S.stratospheric_correction[o] = Z.v[4][j];
    // surface albedo (forcing.f90:53-63)
    const double snowc = fmin(1.0, S.snow_depth[o] / 60.0f);
    const double alb0 = S.alb0[o];
    const double alb_land = alb0 + snowc * (ALBSNd - alb0);
    const double alb_sea = cpl::ALBSEAd + S.sice_am[o] * (cpl::ALBICEd - cpl::ALBSEAd);
    const double fl = S.fmask_land[o];
    S.snowc[o] = snowc;
    S.alb_land[o] = alb_land;
    S.alb_sea[o] = alb_sea;
    S.alb_surface[o] = alb_sea + fl * (alb_land - alb_sea);
    // orographic corrections for horizontal diffusion (forcing.f90:75-101)
    const double corh = gamlat * S.phis0[o];
    corh_t[o] = corh;
    const double rgas = static_cast<double>(2.0f / 7.0f) * static_cast<double>(1004.0f);
    const double pexp = 1.f / (rgas * gamlat);
    const double tsfc = fl * S.land_temp[o] + S.fmask_sea[o] * S.sst_am[o];
    const double tref = tsfc + corh;
    const double psfc = pow(tsfc / tref, pexp);
    const double qref = qsat_p(tref, 1.0);         // get_qsat(tref, psfc/psfc, -1): P = (psfc/psfc)(1,1) = 1
    const double qsfc = qsat_p(tsfc, 1.0 * psfc);  // get_qsat(tsfc, psfc, 1)
    corh_q[o] = static_cast<double>(0.7f) * (qref - qsfc);  // refrh1
}

// spectral state of the resting reference atmosphere given phis and the transforms of the two surface fields
// (prognostics.f90:52-112); one lane per (member, coefficient)
__global__ __launch_bounds__(kT) void rest_state_kernel(RestPtrs R, int M, RestConsts c) {
    using d2 = double __attribute__((ext_vector_type(2)));
    const int gid = blockIdx.x * kT + threadIdx.x;
    if (gid >= M * NSPEC) return;
    const int mem = gid / NSPEC, k = gid - mem * NSPEC;
    const size_t lv = static_cast<size_t>(NSPEC);
    d2 *vor = reinterpret_cast<d2 *>(R.vor) + static_cast<size_t>(mem) * 16 * lv + k;
    d2 *div = reinterpret_cast<d2 *>(R.div) + static_cast<size_t>(mem) * 16 * lv + k;
    d2 *t = reinterpret_cast<d2 *>(R.t) + static_cast<size_t>(mem) * 16 * lv + k;
    d2 *tr = reinterpret_cast<d2 *>(R.tr) + static_cast<size_t>(mem) * 16 * lv + k;
    d2 *ps = reinterpret_cast<d2 *>(R.ps) + static_cast<size_t>(mem) * 2 * lv + k;
    const d2 phis = reinterpret_cast<const d2 *>(R.phis)[static_cast<size_t>(mem) * lv + k];
    const double trf = R.trfilt[k];
    d2 surfs = d2{-c.gam1 * phis.x, -c.gam1 * phis.y};
    if (k == 0) surfs = d2{c.sqrt2 * c.tref - c.gam1 * phis.x, 0.0 * c.tref - c.gam1 * phis.y};
    const d2 spq0 = reinterpret_cast<const d2 *>(R.spec_q)[static_cast<size_t>(mem) * lv + k];
    const d2 spq = d2{spq0.x * trf, spq0.y * trf};
    const d2 sps = reinterpret_cast<const d2 *>(R.spec_ps)[static_cast<size_t>(mem) * lv + k];
#pragma unroll
    for (int l = 0; l < KX; ++l) {
        vor[l * lv] = d2{0.0, 0.0};
        div[l * lv] = d2{0.0, 0.0};
        d2 tv, qv;
        if (l < 2) {
            tv = (k == 0) ? d2{c.sqrt2 * c.ttop, 0.0 * c.ttop} : d2{0.0, 0.0};
            qv = d2{0.0, 0.0};
        } else {
            tv = d2{surfs.x * c.fsg_rgam[l], surfs.y * c.fsg_rgam[l]};
            qv = d2{spq.x * c.fsg_qexp[l], spq.y * c.fsg_qexp[l]};
        }
        t[l * lv] = tv;
        tr[l * lv] = qv;
    }
    ps[0] = d2{sps.x * trf, sps.y * trf};
}

// phi0 = grav * orog (boundaries.f90:27); after the spectral filter: forog (surface_fluxes.f90:324-334) and the two
// surface fields of the resting atmosphere, ln ps and surface q (prognostics.f90:84-105)
__global__ __launch_bounds__(kT) void scale_orog_kernel(const double *orog, double *phi0, long n) {
    const long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x;
    if (i < n) phi0[i] = static_cast<double>(9.81f) * orog[i];
}

__global__ __launch_bounds__(kT) void rest_surface_kernel(const double *phis0, double *forog, double *surf_ps, double *surf_q,
                                                          RestConsts c, long n) {
    const long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x;
    if (i >= n) return;
    const double ph = phis0[i];
    const double grav = 9.81f, hdrag = 2000.0f;
    const double rhdrag = 1.0f / (grav * hdrag);
    forog[i] = 1.0f + rhdrag * (1.0f - exp(-fmax(ph, 0.0) * rhdrag));
    const double gam2 = c.gam1 / c.tref;
    const double rgas = static_cast<double>(2.0f / 7.0f) * static_cast<double>(1004.0f);
    const double rgamr = 1.0f / (rgas * c.gam1);
    const double rlog0 = 0.012916237115859985;  // log(1.013) evaluated in fp32, widened (prognostics.f90:85)
    const double sg = rlog0 + rgamr * log(1.0f - gam2 * ph);
    surf_ps[i] = sg;
    const double qref = static_cast<double>(0.7f) * 0.622f * 17.0f;  // refrh1 * 0.622 * esref
    const double qexp = static_cast<double>(7.5f) / static_cast<double>(2.5f);
    surf_q[i] = qref * exp(qexp * sg);
}

hipError_t run_scale_orog(const double *orog, double *phi0, long n, hipStream_t s) {
    hipLaunchKernelGGL(scale_orog_kernel, dim3((n + kT - 1) / kT), dim3(kT), 0, s, orog, phi0, n);
    return hipGetLastError();
}
hipError_t run_rest_surface(const double *phis0, double *forog, double *surf_ps, double *surf_q, const RestConsts &c, long n,
                            hipStream_t s) {
    hipLaunchKernelGGL(rest_surface_kernel, dim3((n + kT - 1) / kT), dim3(kT), 0, s, phis0, forog, surf_ps, surf_q, c, n);
    return hipGetLastError();
}

// members [first, first + count)
hipError_t run_coupler(const SurfacePtrs &S, int first, int count, const TimeInterp &w, int day, int land_coupling,
                       int sst_anomaly, int anom_planes, int fresh, hipStream_t s) {
    hipLaunchKernelGGL(coupler_kernel, dim3((count * NG + kT - 1) / kT), dim3(kT), 0, s, S, first, count, w, day, land_coupling,
                       sst_anomaly, anom_planes, fresh);
    return hipGetLastError();
}
hipError_t run_forcing(const SurfacePtrs &S, int first, int count, const ZonalDevice &Z, double gamlat, double *corh_t,
                       double *corh_q, hipStream_t s) {
    hipLaunchKernelGGL(forcing_kernel, dim3((count * NG + kT - 1) / kT), dim3(kT), 0, s, S, first, count, Z, gamlat, corh_t,
                       corh_q);
    return hipGetLastError();
}
hipError_t run_rest_state(const RestPtrs &R, int M, const RestConsts &c, hipStream_t s) {
    hipLaunchKernelGGL(rest_state_kernel, dim3((M * NSPEC + kT - 1) / kT), dim3(kT), 0, s, R, M, c);
    return hipGetLastError();
}

// Output units of the grid-space prognostic variables (prognostics.f90:143-150): q kg/kg, phi m, ps Pa; and back (:166-171).
__global__ __launch_bounds__(kT) void export_units_kernel(double *q, double *phi, double *ps, long n2d) {
    const long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x;
    if (i < 8 * n2d) {
        q[i] = q[i] * static_cast<double>(1.0e-3f);
        phi[i] = phi[i] / static_cast<double>(9.81f);
    }
    if (i < n2d) ps[i] = static_cast<double>(1.e+5f) * exp(ps[i]);
}

__global__ __launch_bounds__(kT) void log_ps_kernel(const double *ps_grid, double *out, long n2d) {
    const long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x;
    if (i < n2d) out[i] = log(ps_grid[i] / static_cast<double>(1.e+5f));
}

__global__ __launch_bounds__(kT) void export_spec_units_kernel(double *tr, double *phi, long ndoubles) {
    const long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x;
    if (i < ndoubles) {
        tr[i] = tr[i] / static_cast<double>(1.0e-3f);
        phi[i] = phi[i] * static_cast<double>(9.81f);
    }
}

hipError_t run_export_units(double *q, double *phi, double *ps, long n2d, hipStream_t s) {
    hipLaunchKernelGGL(export_units_kernel, dim3((8 * n2d + kT - 1) / kT), dim3(kT), 0, s, q, phi, ps, n2d);
    return hipGetLastError();
}
hipError_t run_log_ps(const double *ps_grid, double *out, long n2d, hipStream_t s) {
    hipLaunchKernelGGL(log_ps_kernel, dim3((n2d + kT - 1) / kT), dim3(kT), 0, s, ps_grid, out, n2d);
    return hipGetLastError();
}
hipError_t run_export_spec_units(double *tr, double *phi, long ncomplex, hipStream_t s) {
    const long n = 2 * ncomplex;
    hipLaunchKernelGGL(export_spec_units_kernel, dim3((n + kT - 1) / kT), dim3(kT), 0, s, tr, phi, n);
    return hipGetLastError();
}

}  // namespace spd
