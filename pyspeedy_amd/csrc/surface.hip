// Per-step land / sea coupling and daily forcing on the device, one lane per (member, grid point).
//   coupler_kernel   couple_land_atm + run_land_model (land_model.f90:151-215), couple_sea_atm + run_sea_model
//                    (sea_model.f90:193-383).  In this version of the reference the coupler runs EVERY step
//                    (speedy.f90:72), with monthly climatologies interpolated to the current date.
//   forcing_kernel   set_forcing (forcing.f90:15-102): snow cover, albedos, zonal radiation fields and the two grid
//                    fields whose spectral transforms are the orographic diffusion corrections tcorh / qcorh.
//   rest-state helpers for initialize_from_rest_state (prognostics.f90:29-120).
//   land_sea_rows_kernel, land_sea_init_kernel   land_model_init + sea_model_init (boundary-field preprocessing of init).
#include <hip/hip_runtime.h>

#include "coupler_point.hpp"
#include "surface.hpp"
#include "launch_events.hpp"
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

// land_model_init + sea_model_init (land_model.f90:23-148, sea_model.f90:33-192) with fill_missing_values and
// check_surface_fields (boundaries.f90:41-113), two launches over (plane, member) workgroups.
//  * Point by point: the fractional / binary masks, heat capacities and dissipation times (selections among the constants of
//    LandSeaConsts), soil water availability, snow depth, sea-ice fraction and SST anomalies blanked outside their mask.
//  * fill_missing_values on the 12 + 12 planes of stl12 and sst12: a negative value is replaced by the mean of its two zonal
//    neighbours, missing neighbours counting as the mean of the valid points of their row.  The row mean is a sequential sum
//    (the reference's bits); a row without a valid point takes the mean of the row visited before it -- rows 24 .. 1, then
//    25 .. 48, the previous month's last row at row 24, and the land sequence's last mean at the first sea row (a SAVEd
//    variable of the reference, boundaries.f90:77, which starts at 0 in a fresh process; every call starts there here).
//    land_sea_rows_kernel leaves (valid points, mean) of every row of every plane; land_sea_init_kernel walks its plane's 48
//    means in visiting order -- and, only when its first row has no valid point, the rows of the planes before it backwards
//    until it meets one that has -- and patches the plane from a copy in LDS.  The planes of a member are independent but for
//    that look-back, so they run side by side (one workgroup that did the 24 planes in turn took 0.68 ms per member).
// No contraction: every product and sum below is rounded on its own, as on the reference's host.
__device__ inline int visit_row(int k) { return k < IL / 2 ? IL / 2 - 1 - k : k; }  // the k-th row fill_missing_values visits

__global__ __launch_bounds__(kT) void land_sea_rows_kernel(LandSeaPtrs P, int first, double *rows /*[count][24][48][2]*/) {
#pragma clang fp contract(off)
    __shared__ double plane[NG];
    const int f = blockIdx.x, mem = first + blockIdx.y, t = threadIdx.x;
    const double *field = (f < 12 ? P.stl12 : P.sst12) + (static_cast<size_t>(mem) * 12 + f % 12) * NG;
    for (int p = t; p < NG; p += kT) plane[p] = field[p];
    __syncthreads();
    if (t < IL) {
        int nmis = 0;
        double s = 0.0;
        for (int i = 0; i < IX; ++i) {
            const double v = plane[t * IX + i];
            const bool missing = v < 0.0;
            nmis += missing ? 1 : 0;
            s = s + (missing ? 0.0 : v);
        }
        double *out = rows + ((static_cast<size_t>(blockIdx.y) * 24 + f) * IL + t) * 2;
        out[0] = IX - nmis;
        out[1] = nmis < IX ? s / static_cast<double>(static_cast<float>(IX - nmis)) : 0.0;
    }
}

// blockIdx.x < 24: plane x of the fill (stl12 months 0 .. 11, sst12 months 0 .. 11) and that month's share of the point-by-point
// work; blockIdx.x == 24: the fields without a month and the SST anomalies
__global__ __launch_bounds__(kT) void land_sea_init_kernel(LandSeaPtrs P, LandSeaConsts K, int first, const double *rows) {
#pragma clang fp contract(off)
    __shared__ double plane[NG];
    __shared__ double row_fill[IL];
    const int f = blockIdx.x, mem = first + blockIdx.y, t = threadIdx.x;
    const size_t o2 = static_cast<size_t>(mem) * NG, o12 = o2 * 12;
    if (f == 24) {
        for (int p = t; p < NG; p += kT) {
            const double fo = P.fmask_orig[o2 + p];
            double fl = fo, bl, fs = K.one - fo, bs;
            if (fl >= K.thrsh) {
                bl = 1.0;
                if (fo > K.one_minus_thrsh) fl = 1.0;
            } else {
                bl = 0.0;
                fl = 0.0;
            }
            if (fs >= K.thrsh) {
                bs = 1.0;
                if (fs > K.one_minus_thrsh) fs = 1.0;
            } else {
                bs = 0.0;
                fs = 0.0;
            }
            P.fmask_land[o2 + p] = fl;
            P.bmask_land[o2 + p] = bl;
            P.fmask_sea[o2 + p] = fs;
            P.bmask_sea[o2 + p] = bs;
            const int j = p / IX;
            P.rhcapl[o2 + p] = P.alb0[o2 + p] < K.alb_thr ? K.rhcapl[0] : K.rhcapl[1];
            P.cdland[o2 + p] = K.cdland[fl < K.flandmin ? 0 : 1];
            P.rhcaps[o2 + p] = K.rhcaps_row[j];
            P.rhcapi[o2 + p] = K.rhcapi_row[j];
            P.cdsea[o2 + p] = K.cdsea[fs < K.fseamin ? 0 : 1];
            P.cdice[o2 + p] = K.cdice[fs < K.fseamin ? 0 : 1];
            if (P.anom_planes >= 3 && !(bs > 0.0))
                for (int k = 0; k < 3; ++k) P.sst_anom[(static_cast<size_t>(mem) * P.anom_planes + k) * NG + p] = 0.0;
        }
        return;
    }
    const int month = f % 12;
    double *field = (f < 12 ? P.stl12 : P.sst12) + o12 + static_cast<size_t>(month) * NG;
    for (int p = t; p < NG; p += kT) plane[p] = field[p];
    if (t == 0) {
        const double *mine = rows + (static_cast<size_t>(blockIdx.y) * 24 + f) * IL * 2;
        double c = 0.0;
        if (mine[visit_row(0) * 2] == 0.0) {  // the first row has no valid point: the mean the rows before this plane leave behind
            bool found = false;
            for (int g = f - 1; g >= 0 && !found; --g)
                for (int k = IL - 1; k >= 0 && !found; --k) {
                    const double *row = rows + ((static_cast<size_t>(blockIdx.y) * 24 + g) * IL + visit_row(k)) * 2;
                    if (row[0] > 0.0) {
                        c = row[1];
                        found = true;
                    }
                }
        }
        for (int k = 0; k < IL; ++k) {
            const int r = visit_row(k);
            if (mine[r * 2] > 0.0) c = mine[r * 2 + 1];
            row_fill[r] = c;  // what a missing point of this row counts as
        }
    }
    __syncthreads();
    for (int p = t; p < NG; p += kT) {
        const int j = p / IX, i = p - j * IX;
        const double fo = P.fmask_orig[o2 + p];
        const bool land = fo >= K.thrsh, sea = (K.one - fo) >= K.thrsh;  // bmask_land, bmask_sea
        double v = plane[p];
        if (v < 0.0) {
            const double fm = row_fill[j];
            double a = plane[j * IX + (i == 0 ? IX - 1 : i - 1)], b = plane[j * IX + (i == IX - 1 ? 0 : i + 1)];
            a = a < 0.0 ? fm : a;
            b = b < 0.0 ? fm : b;
            const double both = a + b;
            v = 0.5 * both;
        }
        field[p] = (f < 12 ? land : sea) ? v : 273.0;
        // (MAX / MIN as the reference's compiler evaluates them: one ordered comparison and a select)
        const size_t q = o12 + static_cast<size_t>(month) * NG + p;
        if (f < 12) {
            const double weighted = K.veg_low_weight * P.veg_low[o2 + p];
            const double vsum = P.veg_high[o2 + p] + weighted;
            const double veg = 0.0 > vsum ? 0.0 : vsum;
            const double swroot = K.idep2 * P.soil_wc_l2[q];
            const double excess = swroot - K.swwil2;
            const double rooted = veg * (0.0 > excess ? 0.0 : excess);
            const double water = P.soil_wc_l1[q] + rooted;
            const double avail = K.rsw * water;
            P.soilw12[q] = land ? (1.0 < avail ? 1.0 : avail) : 0.0;
            if (!land) P.snowd12[q] = 0.0;
        } else {
            const double ice = P.sea_ice_frac12[q];
            P.sea_ice_frac12[q] = sea ? (ice > 0.0 ? ice : 0.0) : 0.0;
        }
    }
}
// members [first, first + count); `rows`: 2 * 24 * 48 doubles of scratch per member of the range
hipError_t run_land_sea_init(const LandSeaPtrs &P, const LandSeaConsts &K, int first, int count, double *rows, hipStream_t s) {
    hipLaunchKernelGGL(land_sea_rows_kernel, dim3(24, count), dim3(kT), 0, s, P, first, rows);
    hipLaunchKernelGGL(land_sea_init_kernel, dim3(25, count), dim3(kT), 0, s, P, K, first, rows);
    return hipGetLastError();
}

// A member's arrays from one model into another (spd_model_copy_member: the driver's gather / split of batched models): the
// ~150 registry arrays of a member lie in 150 places, and one hipMemcpyAsync each made a gather of 64 members cost 60 ms of host
// time.  One launch: block (x, y) copies every gridDim.y-th 4 KiB piece of array x.
__global__ __launch_bounds__(kT) void multi_copy_kernel(CopyList L) {
    const int a = blockIdx.x;
    const uint4 *src = reinterpret_cast<const uint4 *>(L.src[a]);
    uint4 *dst = reinterpret_cast<uint4 *>(L.dst[a]);
    const unsigned n = L.bytes[a] / 16;
    for (unsigned i = blockIdx.y * kT + threadIdx.x; i < n; i += gridDim.y * kT) dst[i] = src[i];
}
hipError_t run_multi_copy(const CopyList &L, hipStream_t s) {
    if (L.n == 0) return hipSuccess;
    hipLaunchKernelGGL(multi_copy_kernel, dim3(L.n, 16), dim3(kT), 0, s, L);
    return hipGetLastError();
}

// spd_model_set for all members at once: member 0's copy of an array handed to the members whose flag is 0
__global__ void copy_from_first_kernel(double *v, long n, const int *flags) {
    const int mem = blockIdx.y + 1;
    if (flags[mem]) return;
    double *mine = v + static_cast<size_t>(mem) * n;
    for (long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x; i < n; i += static_cast<long>(gridDim.x) * kT) mine[i] = v[i];
}
hipError_t run_copy_from_first(double *v, long n, int M, const int *flags, hipStream_t s) {
    if (M < 2) return hipSuccess;
    const long blocks = (n + kT - 1) / kT;
    hipLaunchKernelGGL(copy_from_first_kernel, dim3(blocks < 64 ? blocks : 64, M - 1), dim3(kT), 0, s, v, n, flags);
    return hipGetLastError();
}

// spd_model_set_physics_precision: an array changes between fp64 storage and fp32 storage in the first half of the same
// allocation.  Through a scratch array: in place the narrow writes of one lane would land on the wide values other lanes read.
__global__ void narrow_kernel(const double *in, float *out, long n) {
    const long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x;
    if (i < n) out[i] = static_cast<float>(in[i]);
}
__global__ void widen_kernel(const float *in, double *out, long n) {
    const long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x;
    if (i < n) out[i] = static_cast<double>(in[i]);
}
hipError_t run_change_storage(double *array, long n, bool to_float, void *scratch, hipStream_t s) {
    const dim3 grid(static_cast<unsigned>((n + kT - 1) / kT));
    if (to_float) hipLaunchKernelGGL(narrow_kernel, grid, dim3(kT), 0, s, array, static_cast<float *>(scratch), n);
    else hipLaunchKernelGGL(widen_kernel, grid, dim3(kT), 0, s, reinterpret_cast<const float *>(array), static_cast<double *>(scratch), n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(array, scratch, static_cast<size_t>(n) * (to_float ? sizeof(float) : sizeof(double)), hipMemcpyDeviceToDevice, s);
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
    launch(coupler_kernel, dim3((count * NG + kT - 1) / kT), dim3(kT), 0, s, S, first, count, w, day, land_coupling,
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

// One grid-space variable of `count` members as it goes into a NetCDF-3 file (pyspeedy/speedy.py:415-477: float32, levels counted
// upwards from the surface; the classic format is big-endian): dst[member][levels - 1 - k][point] <- bswap32(float(src[member][k][point])).
// Coalesced 8- (or 4-) byte reads, coalesced 4-byte writes; the output is consumed by a copy to the host: streamed past the L2.
template <typename SRC>
__global__ __launch_bounds__(kT) void export_pack_kernel(const SRC *__restrict__ src, unsigned *__restrict__ dst, int levels, long n) {
    const long i = static_cast<long>(blockIdx.x) * kT + threadIdx.x;
    if (i >= n) return;
    const long plane = i / NG, p = i - plane * NG;
    const long mem = plane / levels, k = plane - mem * levels;
    const float v = static_cast<float>(src[(mem * levels + (levels - 1 - k)) * NG + p]);
    __builtin_nontemporal_store(__builtin_bswap32(__float_as_uint(v)), &dst[i]);
}

hipError_t run_export_pack(const void *src, bool src_is_float, void *dst, int levels, int count, hipStream_t s) {
    const long n = static_cast<long>(count) * levels * NG;
    if (n == 0) return hipSuccess;
    const dim3 grid(static_cast<unsigned>((n + kT - 1) / kT));
    if (src_is_float)
        hipLaunchKernelGGL(export_pack_kernel<float>, grid, dim3(kT), 0, s, static_cast<const float *>(src), static_cast<unsigned *>(dst), levels, n);
    else
        hipLaunchKernelGGL(export_pack_kernel<double>, grid, dim3(kT), 0, s, static_cast<const double *>(src), static_cast<unsigned *>(dst), levels, n);
    return hipGetLastError();
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
