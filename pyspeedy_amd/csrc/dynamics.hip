// The callers of the hot path, on the device (SURVEY.md section 8f "next #1"): everything do_single_step does between
// the spectral transforms and the column physics, so that a model step never leaves HBM.
//
//   (vort2vel at two time levels and gradient(ln ps) run inside the spectral->grid kernel: FieldDesc::mode)
//   geopotential_kernel     hydrostatic integration in spectral space           geopotential.f90:49-77
//   dyn_grid_kernel         grid-point dynamics tendencies of one column        tendencies.f90:125-224
//   spectral_step_kernel    vel2vort + spectral tendencies + semi-implicit correction + horizontal diffusion +
//                           leapfrog / Robert-Asselin-Williams filter           tendencies.f90:238-352, implicit.f90:234-289,
//                                                                               time_stepping.f90:71-188
//   diagnostics_kernel      global-mean T / eddy KE range check                 diagnostics.f90:16-76
//
// Layouts (member-major, the reference's Fortran order inside a member):
//   vor, div, t, tr  complex [M][2 time levels][8][32][31] ; ps [M][2][32][31] ; phi [M][8][32][31] ; phis [M][32][31]
//   grid fields      [M][8][48][96] or [M][48][96]
// One lane per spectral coefficient or grid column (spectral_step_kernel: per coefficient and level); unit-stride accesses.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "coupler_point.hpp"
#include "device_tables.hpp"
#include "diagnostics_block.hpp"
#include "dyn_column.hpp"
#include "launch_events.hpp"
#include "model.hpp"
#include "sppt_point.hpp"

namespace spd {

namespace {
using d2 = double __attribute__((ext_vector_type(2)));
constexpr int NG = IX * IL;
constexpr int kT = 256;

__device__ constexpr double CPd = 1004.0f;
__device__ constexpr double AKAPd = 2.0f / 7.0f;
__device__ constexpr double RGASd = AKAPd * CPd;
__device__ constexpr double WILd = 0.53f, TDRSd = 24.0f * 30.0f;

__device__ inline d2 times_i(d2 z) { return d2{-z.y, z.x}; }
__device__ inline d2 operator_scale(double c, d2 z) { return d2{c * z.x, c * z.y}; }

// vort2vel (MODE 0, tables uvdx/uvdym/uvdyp) or vel2vort (MODE 1, gradx/vddym/vddyp) at coefficient k = m + 31 n of the
// fields a, b (pointers to the field start), in two halves: all six loads first (neighbour indices clamped into the field,
// so that the loads need no branch and a lane can have the loads of several stencils in flight at once), the arithmetic
// afterwards (spectral.f90:190-214 / 275-296: first, last and inner total wavenumbers have different formulas).
struct Stencil {
    d2 ac, bc, ap, bp, an, bn;
};
__device__ inline Stencil load_stencil(const d2 *a, const d2 *b, int k, int n) {
    const int kp = n > 0 ? k - MX : k, kn = n < NX - 1 ? k + MX : k;
    return Stencil{a[k], b[k], a[kp], b[kp], a[kn], b[kn]};
}
template <int MODE>
__device__ inline void apply_stencil(const Stencil &s, int k, int m, int n, const DeviceTables &T, d2 &ra, d2 &rb) {
    const double *tym = MODE == 0 ? T.uvdym : T.vddym, *typ = MODE == 0 ? T.uvdyp : T.vddyp;
    const double dx = MODE == 0 ? T.uvdx[k] : T.gradx[m];
    const double cm = tym[k], cp = typ[k];
    const d2 zp = times_i(d2{dx * s.ac.x, dx * s.ac.y}), zc = times_i(d2{dx * s.bc.x, dx * s.bc.y});
    if (n == 0) {
        ra = d2{zc.x - cp * s.an.x, zc.y - cp * s.an.y};
        rb = d2{zp.x + cp * s.bn.x, zp.y + cp * s.bn.y};
    } else if (n == NX - 1) {
        ra = d2{cm * s.ap.x, cm * s.ap.y};
        rb = d2{-cm * s.bp.x, -cm * s.bp.y};
    } else {
        ra = d2{cm * s.ap.x - cp * s.an.x + zc.x, cm * s.ap.y - cp * s.an.y + zc.y};
        rb = d2{-cm * s.bp.x + cp * s.bn.x + zp.x, -cm * s.bp.y + cp * s.bn.y + zp.y};
    }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// geopotential from temperature (geopotential.f90:49-77).  The arithmetic lives in two inline functions with explicit
// fused multiply-adds, because it exists twice -- in the stand-alone kernel below and at the end of spectral_step_kernel,
// which computes the geopotential of the NEXT step from the temperature it has just advanced -- and both must give the
// same bits (a run resumed from a checkpoint starts with the stand-alone kernel).
// ---------------------------------------------------------------------------------------------------------
namespace {
// phi(l) from phi(l + 1): geopotential.f90:62-65
__device__ __forceinline__ d2 geo_up(d2 ph_below, d2 t_below, d2 t_here, double xg2_below, double xg1_here) {
    return d2{fma(xg1_here, t_here.x, fma(xg2_below, t_below.x, ph_below.x)),
              fma(xg1_here, t_here.y, fma(xg2_below, t_below.y, ph_below.y))};
}
// lapse-rate correction of the zonal-mean coefficients in the free troposphere: geopotential.f90:68-74
__device__ __forceinline__ d2 geo_corr(d2 ph, d2 t_below, d2 t_above, double corf) {
    return d2{fma(corf, t_below.x - t_above.x, ph.x), fma(corf, t_below.y - t_above.y, ph.y)};
}
}  // namespace

// geopotential from temperature at time level `tl`.  SPPT: the blocks behind the geopotential ones advance the AR(1) pattern
// of the stochastic physics (sppt_point.hpp) -- both are the small launches that open a step of an ensemble with SPPT, neither
// depends on the other, so they share one (tendencies.f90:229 and physics.f90:234-236 in one launch).
template <bool SPPT>
__global__ __launch_bounds__(kT) void geopotential_kernel(ModelPtrs P, DynDeviceTables D, int first, int count, int tl, SpptArgs sp) {
    // (writes P.phi: the geopotential the current step uses)
    const int ngeo = (count * NSPEC + kT - 1) / kT;
    if (SPPT && static_cast<int>(blockIdx.x) >= ngeo) {
        sppt_update_point(sp, static_cast<long>(blockIdx.x - ngeo) * kT + threadIdx.x);
        return;
    }
    const int gid = blockIdx.x * kT + threadIdx.x;
    if (gid >= count * NSPEC) return;
    const int lm = gid / NSPEC, mem = first + lm, k = gid - lm * NSPEC, m = k % MX;
    const d2 *t = reinterpret_cast<const d2 *>(P.t) + (static_cast<size_t>(mem) * 2 + tl) * 8 * NSPEC + k;
    d2 *phi = reinterpret_cast<d2 *>(P.phi) + static_cast<size_t>(mem) * 8 * NSPEC + k;
    d2 tt[KX], ph[KX];
#pragma unroll
    for (int l = 0; l < KX; ++l) tt[l] = t[static_cast<size_t>(l) * NSPEC];
    const d2 phis = reinterpret_cast<const d2 *>(P.phis)[static_cast<size_t>(mem) * NSPEC + k];
    ph[KX - 1] = d2{fma(D.xgeop1[KX - 1], tt[KX - 1].x, phis.x), fma(D.xgeop1[KX - 1], tt[KX - 1].y, phis.y)};
#pragma unroll
    for (int l = KX - 2; l >= 0; --l) ph[l] = geo_up(ph[l + 1], tt[l + 1], tt[l], D.xgeop2[l + 1], D.xgeop1[l]);
    if (m == 0) {
#pragma unroll
        for (int l = 1; l < KX - 1; ++l) ph[l] = geo_corr(ph[l], tt[l + 1], tt[l - 1], D.geo_corf[l]);
    }
#pragma unroll
    for (int l = 0; l < KX; ++l) phi[static_cast<size_t>(l) * NSPEC] = ph[l];
}

// ---------------------------------------------------------------------------------------------------------
// grid-point dynamics of one column (dyn_column.hpp), stand-alone form.  The model step uses the fused dynamics +
// physics kernel of physics.hip instead; this kernel serves spd_model_step_dynamics' split mode (parity tests).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void dyn_grid_kernel(ModelPtrs P, DynDeviceTables D, int M) {
    const int gid = blockIdx.x * kT + threadIdx.x;
    if (gid >= M * NG) return;
    const int mem = gid / NG, p = gid - mem * NG, j = p / IX;
    double ttend[KX], trtend[KX], ut, vt;
    dyn_column<true>(P, D, mem, p, j, ttend, trtend, ut, vt);
}

// ---------------------------------------------------------------------------------------------------------
// everything in spectral space after the forward transforms.
// One lane per (coefficient, level): a wavefront owns 8 consecutive coefficients x 8 levels of one member
// (lane = 8 * level + coefficient-in-block, so each level's 8 coefficients are one 128-byte line).  What couples the
// levels -- the mass-weighted divergence sum, the sigma-dot prefix sum and the three 8x8 matrix products of the
// semi-implicit scheme -- gathers the 8 level values of a coefficient from the other lanes (ds_bpermute) and then runs
// the reference's loops in the reference's order on them, so the arithmetic is the same as with all levels in one lane
// while the kernel has 8x the lanes to hide memory latency with (the one-lane-per-coefficient form needed 255 VGPRs and
// took 45 us for 8 members / 72 us for 64: profiles/).
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int kCoefBlocks = NSPEC / 8;  // 124 blocks of 8 coefficients
static_assert(NSPEC % 8 == 0, "coefficient blocks");

__device__ inline void gather_levels(d2 v, int kk, d2 (&out)[KX]) {
#pragma unroll
    for (int l = 0; l < KX; ++l) out[l] = d2{__shfl(v.x, 8 * l + kk), __shfl(v.y, 8 * l + kk)};
}
template <int N>
__device__ inline double pick(const double (&a)[N], int l) {
    double r = a[0];
#pragma unroll
    for (int i = 1; i < N; ++i) r = (l == i) ? a[i] : r;
    return r;
}
template <int N>
__device__ inline d2 pick(const d2 (&a)[N], int l) {
    d2 r = a[0];
#pragma unroll
    for (int i = 1; i < N; ++i) r = (l == i) ? a[i] : r;
    return r;
}
}  // namespace

// FOLD: the kernel ends by computing the geopotential of the next step into P.phi_next (a template parameter and not a run-time
// switch: the extra code costs the 64-member launch 7 % even when it is skipped).
//
// CA = CouplerArgs: horizontal fusion with the land / sea-ice coupling of the step (speedy.f90:72).  The coupling depends on the
// column kernel only, as this kernel's spectral work depends on the transforms only: the blocks behind the spectral ones
// do the coupling for 256 grid points each, in the same launch -- one kernel boundary less and, for ensembles that do not
// fill the GPU, the coupling runs beside the spectral step instead of after it.  CA = NoCoupler (empty): spectral work only.
struct NoCoupler {};
template <bool FOLD, bool EARLY, typename CA>
__global__ __launch_bounds__(kT) void spectral_step_kernel(ModelPtrs P, DeviceTables T, DynDeviceTables D, int M, int first,
                                                           int count, int j1, double dt, double eps, CA cpl) {
    int nblocks = gridDim.x;  // blocks of spectral work
    if constexpr (std::is_same<CA, CouplerArgs>::value) {
        nblocks = (count * NSPEC * KX + kT - 1) / kT;
        if (static_cast<int>(blockIdx.x) >= nblocks) {
            const int gid = (blockIdx.x - nblocks) * kT + threadIdx.x;
            if (gid < cpl.count * NG) {
                const int lm = gid / NG;
                coupler_point(cpl.S, cpl.first + lm, gid - lm * NG, cpl.w, cpl.day, cpl.land_coupling, cpl.sst_anomaly,
                              cpl.anom_planes, cpl.fresh);
            }
            return;
        }
    }
    // XCD-aware block order: workgroups are dealt to the 8 XCDs round-robin, and the n-1 / n+1 neighbours of the vel2vort
    // stencils live one workgroup away (31 coefficients), so each XCD is given a CONTIGUOUS range of the work -- the halo
    // lines are then found in that XCD's L2 instead of being fetched again by every neighbour's XCD.
    const int xcd = blockIdx.x & 7, q = nblocks >> 3, r = nblocks & 7;  // XCD c owns q + (c < r) consecutive blocks
    const int block = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
    const int gid = block * kT + threadIdx.x;
    const int w = gid >> 6, lane = gid & 63;
    if (w >= count * kCoefBlocks) return;  // whole wavefronts only: the gathers below need all 64 lanes
    const int l = lane >> 3, kk = lane & 7;
    const int lm = w / kCoefBlocks, mem = first + lm, k = (w - lm * kCoefBlocks) * 8 + kk, n = k / MX, m = k - n * MX;
    const size_t f8 = static_cast<size_t>(mem) * 8 * NSPEC;        // [M][8] work arrays
    const size_t pair = static_cast<size_t>(M) * 8 * NSPEC;        // stride between the three (u,v)-pair outputs
    const size_t fo = static_cast<size_t>(l) * NSPEC;
    const d2 *su = reinterpret_cast<const d2 *>(P.specu) + f8 + fo, *sv = reinterpret_cast<const d2 *>(P.specv) + f8 + fo;
    const double el2 = T.el2[k];
    d2 vordt, divdt, tdt, trdt, dump;
    // ---- state at both time levels, this lane's level ----
    const size_t s0 = static_cast<size_t>(mem) * 2 * 8 * NSPEC + fo + k;
    const size_t lvl = static_cast<size_t>(8) * NSPEC;               // distance between the two time levels
    d2 *vorS = reinterpret_cast<d2 *>(P.vor) + s0, *divS = reinterpret_cast<d2 *>(P.div) + s0;
    d2 *tS = reinterpret_cast<d2 *>(P.t) + s0, *trS = reinterpret_cast<d2 *>(P.tr) + s0;
    d2 *psS = reinterpret_cast<d2 *>(P.ps) + static_cast<size_t>(mem) * 2 * NSPEC + k;
    const int l_tot = m + n;  // total wavenumber; xj(:, :, l_tot) with 1-based third index
    // EARLY: every global load of the kernel is issued in one batch, here, and not where its value is first needed.  The
    // kernel is a chain of short dependent phases; when the ensemble does not fill the GPU its duration is the number of
    // memory round trips on that chain (about 15 in the other form, whose loads sit inside the matrix loops and in front of
    // each use), and registers are free.  The other form keeps 4 wavefronts per SIMD (122 VGPRs).
    d2 ke, stt, str, psdt, ph, div1, ps1;                                  // first batch
    d2 vor1, t1, tr1, tcorh, qcorh, phis_k{0.0, 0.0}, vor2{}, div2{}, t2{}, tr2{}, ps2{};  // second batch
    double xc_l[KX], xj_l[KX], xd_l[KX], elz, dmp, dmp1, dmpd, dmp1d, dmps, dmp1s, trf;
    auto load_diffusion = [&]() {
        dmp = D.dmp[k], dmp1 = D.dmp1[k], dmpd = D.dmpd[k], dmp1d = D.dmp1d[k], dmps = D.dmps[k], dmp1s = D.dmp1s[k];
        trf = T.trfilt[k];
        tcorh = reinterpret_cast<const d2 *>(P.tcorh)[static_cast<size_t>(mem) * NSPEC + k];
        qcorh = reinterpret_cast<const d2 *>(P.qcorh)[static_cast<size_t>(mem) * NSPEC + k];
    };
    if constexpr (EARLY) {
        const Stencil s_uv = load_stencil(su, sv, k, n);
        const Stencil s_ut = load_stencil(su + pair, sv + pair, k, n);
        const Stencil s_uq = load_stencil(su + 2 * pair, sv + 2 * pair, k, n);
        ke = reinterpret_cast<const d2 *>(P.spec_ke)[f8 + fo + k];
        stt = reinterpret_cast<const d2 *>(P.spec_tt)[f8 + fo + k];
        str = reinterpret_cast<const d2 *>(P.spec_tr)[f8 + fo + k];
        psdt = reinterpret_cast<const d2 *>(P.spec_ps)[static_cast<size_t>(mem) * NSPEC + k];
        ph = reinterpret_cast<const d2 *>(P.phi)[f8 + fo + k];
        div1 = divS[0];
        ps1 = psS[0];
        const double *xj = D.xj + static_cast<size_t>(KX) * KX * (l_tot > 0 ? l_tot - 1 : 0);
#pragma unroll
        for (int k1 = 0; k1 < KX; ++k1) xc_l[k1] = D.xc[l + KX * k1], xj_l[k1] = xj[l + KX * k1], xd_l[k1] = D.xd[l + KX * k1];
        elz = D.elz[k];
        load_diffusion();
        vor1 = vorS[0], t1 = tS[0], tr1 = trS[0];
        vor2 = vorS[lvl], div2 = divS[lvl], t2 = tS[lvl], tr2 = trS[lvl], ps2 = psS[NSPEC];
        if (FOLD) phis_k = reinterpret_cast<const d2 *>(P.phis)[static_cast<size_t>(mem) * NSPEC + k];
        apply_stencil<1>(s_uv, k, m, n, T, vordt, divdt);                                // grid_vel2vort(utend, vtend)
        const d2 lap = d2{-ke.x * el2, -ke.y * el2};                                     // laplacian(grid2spec(KE))
        divdt = d2{divdt.x - lap.x, divdt.y - lap.y};
        apply_stencil<1>(s_ut, k, m, n, T, dump, tdt);                                   // div of (-uT', -vT')
        tdt = d2{tdt.x + stt.x, tdt.y + stt.y};
        apply_stencil<1>(s_uq, k, m, n, T, dump, trdt);                                  // div of (-uq, -vq)
        trdt = d2{trdt.x + str.x, trdt.y + str.y};
    } else {  // the stencils' 18 loads in one batch (their branches would each wait for their own), the rest where it is used
        const Stencil s_uv = load_stencil(su, sv, k, n);
        const Stencil s_ut = load_stencil(su + pair, sv + pair, k, n);
        const Stencil s_uq = load_stencil(su + 2 * pair, sv + 2 * pair, k, n);
        ke = reinterpret_cast<const d2 *>(P.spec_ke)[f8 + fo + k];
        stt = reinterpret_cast<const d2 *>(P.spec_tt)[f8 + fo + k];
        str = reinterpret_cast<const d2 *>(P.spec_tr)[f8 + fo + k];
        apply_stencil<1>(s_uv, k, m, n, T, vordt, divdt);
        const d2 lap = d2{-ke.x * el2, -ke.y * el2};
        divdt = d2{divdt.x - lap.x, divdt.y - lap.y};
        apply_stencil<1>(s_ut, k, m, n, T, dump, tdt);
        tdt = d2{tdt.x + stt.x, tdt.y + stt.y};
        apply_stencil<1>(s_uq, k, m, n, T, dump, trdt);
        trdt = d2{trdt.x + str.x, trdt.y + str.y};
        psdt = reinterpret_cast<const d2 *>(P.spec_ps)[static_cast<size_t>(mem) * NSPEC + k];
        div1 = divS[0];
        ps1 = psS[0];
    }
    if (k == 0) psdt = d2{0.0, 0.0};

    const double tref_l = pick(D.tref, l);

    // ---- spectral tendencies (tendencies.f90:283-352, called with time level 1 because alph = 0.5) ----
    d2 dmeanc{0.0, 0.0};
    {
        d2 dall[KX];
        gather_levels(div1, kk, dall);
#pragma unroll
        for (int j = 0; j < KX; ++j) dmeanc = d2{dmeanc.x + dall[j].x * D.dhs[j], dmeanc.y + dall[j].y * D.dhs[j]};
        psdt = d2{psdt.x - dmeanc.x, psdt.y - dmeanc.y};
        if (k == 0) psdt = d2{0.0, 0.0};
        // sigma-dot at the half levels below (l) and above (l + 1) this level: the reference's running sum
        // sig(j+1) = sig(j) - dhs(j) (div(j) - dmeanc), sig(1) = sig(kx+1) = 0, accumulated in the same order
        d2 sig_l{0.0, 0.0}, sig_l1{0.0, 0.0};
#pragma unroll
        for (int j = 0; j < KX - 1; ++j) {
            const d2 term = d2{D.dhs[j] * (dall[j].x - dmeanc.x), D.dhs[j] * (dall[j].y - dmeanc.y)};
            if (j < l) sig_l = d2{sig_l.x - term.x, sig_l.y - term.y};
            if (j <= l) sig_l1 = d2{sig_l1.x - term.x, sig_l1.y - term.y};
        }
        if (l == KX - 1) sig_l1 = d2{0.0, 0.0};
        // tref(l) - tref(l-1); zero at both ends (dumk(1) = dumk(kx+1) = 0)
        double dtr_l = 0.0, dtr_l1 = 0.0;
#pragma unroll
        for (int j = 1; j < KX; ++j) {
            const double d = D.tref[j] - D.tref[j - 1];
            dtr_l = (l == j) ? d : dtr_l;
            dtr_l1 = (l + 1 == j) ? d : dtr_l1;
        }
        const d2 dumk_l = d2{sig_l.x * dtr_l, sig_l.y * dtr_l}, dumk_l1 = d2{sig_l1.x * dtr_l1, sig_l1.y * dtr_l1};
        const double dhsr_l = pick(D.dhsr, l), tref3_l = pick(D.tref3, l), tref2_l = pick(D.tref2, l);
        tdt.x = tdt.x - (dumk_l1.x + dumk_l.x) * dhsr_l + tref3_l * (sig_l1.x + sig_l.x) - tref2_l * dmeanc.x;
        tdt.y = tdt.y - (dumk_l1.y + dumk_l.y) * dhsr_l + tref3_l * (sig_l1.y + sig_l.y) - tref2_l * dmeanc.y;
    }
    {   // geopotential (valid for time level 1: formed before the physics) and its Laplacian
        if constexpr (!EARLY) ph = reinterpret_cast<const d2 *>(P.phi)[f8 + fo + k];
        const double c = RGASd * tref_l;
        const d2 x = d2{ph.x + c * ps1.x, ph.y + c * ps1.y};
        const d2 lap = d2{-x.x * el2, -x.y * el2};
        divdt = d2{divdt.x - lap.x, divdt.y - lap.y};
    }

    // ---- semi-implicit correction (implicit.f90:234-289) ----
    {
        d2 all[KX];
        gather_levels(tdt, kk, all);
        d2 ye{0.0, 0.0};
#pragma unroll
        for (int k1 = 0; k1 < KX; ++k1) {
            const double x = EARLY ? xd_l[k1] : D.xd[l + KX * k1];
            ye = d2{ye.x + x * all[k1].x, ye.y + x * all[k1].y};
        }
        if constexpr (!EARLY) elz = D.elz[k];
        const double c = RGASd * tref_l;
        ye = d2{ye.x + c * psdt.x, ye.y + c * psdt.y};
        const d2 yf = d2{divdt.x + elz * ye.x, divdt.y + elz * ye.y};
        divdt = d2{0.0, 0.0};
        gather_levels(yf, kk, all);
        if (l_tot != 0) {
            const double *xj = D.xj + static_cast<size_t>(KX) * KX * (l_tot - 1);
#pragma unroll
            for (int k1 = 0; k1 < KX; ++k1) {
                const double x = EARLY ? xj_l[k1] : xj[l + KX * k1];
                divdt = d2{divdt.x + x * all[k1].x, divdt.y + x * all[k1].y};
            }
        }
        gather_levels(divdt, kk, all);
#pragma unroll
        for (int j = 0; j < KX; ++j) psdt = d2{psdt.x - all[j].x * D.dhsx[j], psdt.y - all[j].y * D.dhsx[j]};
#pragma unroll
        for (int k1 = 0; k1 < KX; ++k1) {
            const double x = EARLY ? xc_l[k1] : D.xc[l + KX * k1];
            tdt = d2{tdt.x + x * all[k1].x, tdt.y + x * all[k1].y};
        }
    }

    // ---- horizontal diffusion (time_stepping.f90:78-122) and time integration (:130-188) ----
    if constexpr (!EARLY) load_diffusion();
    const double sdrag = 1.0f / (TDRSd * 3600.0f);
    auto diffuse = [](d2 field, d2 fdt, double a, double b) { return d2{(fdt.x - a * field.x) * b, (fdt.y - a * field.y) * b}; };
    // step_field_2d, time_stepping.f90:164-188, on the two time levels o1, o2 of a field (loaded above); returns level 1
    auto advance = [&](d2 *base, size_t stride, d2 o1, d2 o2, d2 fdt) -> d2 {
        if (!EARLY) o1 = base[0], o2 = base[stride];
        fdt = d2{fdt.x * trf, fdt.y * trf};
        const d2 oj = (j1 == 0) ? o1 : o2;
        const d2 fnew = d2{o1.x + dt * fdt.x, o1.y + dt * fdt.y};
        const double we = WILd * eps;
        const d2 n1 = d2{oj.x + we * (o1.x - 2 * oj.x + fnew.x), oj.y + we * (o1.y - 2 * oj.y + fnew.y)};
        const d2 oja = (j1 == 0) ? n1 : oj;
        const double we2 = (1.0f - WILd) * eps;
        const d2 n2 = d2{fnew.x - we2 * (n1.x - 2.0f * oja.x + fnew.x), fnew.y - we2 * (n1.y - 2.0f * oja.y + fnew.y)};
        stream_store(&base[0], n1);
        stream_store(&base[stride], n2);
        return n1;
    };
    {
        if constexpr (!EARLY) vor1 = vorS[0], t1 = tS[0], tr1 = trS[0];
        d2 vd = diffuse(vor1, vordt, dmp, dmp1);
        d2 dd = diffuse(div1, divdt, dmpd, dmp1d);
        const double tcorv_l = pick(D.tcorv, l), qcorv_l = pick(D.qcorv, l);
        const d2 ct = d2{t1.x + tcorh.x * tcorv_l, t1.y + tcorh.y * tcorv_l};
        d2 td = diffuse(ct, tdt, dmp, dmp1);
        if (l == 0 && m == 0) {  // stratospheric zonal-wind drag on the zonal-mean flow of the top level
            vd = d2{vd.x - sdrag * vor1.x, vd.y - sdrag * vor1.y};
            dd = d2{dd.x - sdrag * div1.x, dd.y - sdrag * div1.y};
        }
        vd = diffuse(vor1, vd, dmps, dmp1s);
        dd = diffuse(div1, dd, dmps, dmp1s);
        td = diffuse(ct, td, dmps, dmp1s);
        const d2 cq = d2{tr1.x + qcorh.x * qcorv_l, tr1.y + qcorh.y * qcorv_l};
        const d2 qd = diffuse(cq, trdt, dmpd, dmp1d);
        advance(vorS, lvl, vor1, vor2, vd);
        advance(divS, lvl, div1, div2, dd);
        const d2 t_new = advance(tS, lvl, t1, t2, td);
        advance(trS, lvl, tr1, tr2, qd);
        // geopotential of the NEXT step (geopotential.f90:49-77 on the temperature at time level 1 as it is now): each lane
        // integrates from the lowest level up to its own, in the order and with the arithmetic of geopotential_kernel
        if (FOLD) {
            d2 tall[KX];
            gather_levels(t_new, kk, tall);
            if constexpr (!EARLY) phis_k = reinterpret_cast<const d2 *>(P.phis)[static_cast<size_t>(mem) * NSPEC + k];
            d2 pn = d2{fma(D.xgeop1[KX - 1], tall[KX - 1].x, phis_k.x), fma(D.xgeop1[KX - 1], tall[KX - 1].y, phis_k.y)};
#pragma unroll
            for (int j = KX - 2; j >= 0; --j) {
                const d2 up = geo_up(pn, tall[j + 1], tall[j], D.xgeop2[j + 1], D.xgeop1[j]);
                pn = (j >= l) ? up : pn;
            }
            if (m == 0 && l >= 1 && l < KX - 1) {
                d2 tb = tall[2], ta = tall[0];  // l == 1; the select chain below picks the neighbours of the other levels
                double corf = D.geo_corf[1];
#pragma unroll
                for (int j = 2; j < KX - 1; ++j) {
                    tb = (l == j) ? tall[j + 1] : tb;
                    ta = (l == j) ? tall[j - 1] : ta;
                    corf = (l == j) ? D.geo_corf[j] : corf;
                }
                pn = geo_corr(pn, tb, ta, corf);
            }
            stream_store(&reinterpret_cast<d2 *>(P.phi_next)[f8 + fo + k], pn);
        }
    }
    if (l == 0) advance(psS, NSPEC, ps1, ps2, psdt);  // ln ps has no vertical index: the two time levels are NSPEC apart
}

// ---------------------------------------------------------------------------------------------------------
// diagnostics (diagnostics.f90:16-76): one workgroup per member, one wavefront per level; writes err[member] = 4 * ticket + (1 if
// out of range) -- always, so that the caller does not have to clear it first.  `err` is pinned host memory: the ticket of the
// launch travels with every code, so the host can tell a fresh code from what an earlier launch left there by looking at the
// memory alone, the moment the store lands (model.hip: wait_codes), instead of waiting for a completion event behind the kernel.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * KX) void diagnostics_kernel(CheckArgs c, DeviceTables T) {
    diagnostics_block(c, T, c.first + static_cast<int>(blockIdx.x));
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
// (first, count): the members the launch works on; M: members in the arrays (strides between the blocks of specu / specv)
// sppt != nullptr: the launch also advances the SPPT pattern (all members of the model) in its tail blocks
hipError_t run_geopotential(const ModelPtrs &P, const DynDeviceTables &D, int first, int count, int tl, const SpptArgs *sppt,
                            hipStream_t s) {
    const int ngeo = (count * NSPEC + kT - 1) / kT;
    if (sppt) {
        const long n = static_cast<long>(sppt->M) * KX * NSPEC;
        launch(geopotential_kernel<true>, dim3(ngeo + static_cast<unsigned>((n + kT - 1) / kT)), dim3(kT), 0, s, P, D,
                           first, count, tl, *sppt);
    } else {
        launch(geopotential_kernel<false>, dim3(ngeo), dim3(kT), 0, s, P, D, first, count, tl, SpptArgs{});
    }
    return hipGetLastError();
}
hipError_t run_dyn_grid(const ModelPtrs &P, const DynDeviceTables &D, int M, hipStream_t s) {
    launch(dyn_grid_kernel, dim3((M * NG + kT - 1) / kT), dim3(kT), 0, s, P, D, M);
    return hipGetLastError();
}
// cpl != nullptr: the coupling of the step rides in the same launch (tail blocks).  Small launches use the form of the kernel
// with all loads up front (EARLY): same arithmetic, 2 instead of 4 wavefronts per SIMD, a fraction of the memory round trips.
namespace {
template <bool FOLD, bool EARLY, typename CA>
void launch_spectral_step(dim3 grid, hipStream_t s, const ModelPtrs &P, const DeviceTables &T, const DynDeviceTables &D, int M,
                          int first, int count, int j1, double dt, double eps, const CA &cpl) {
    launch(spectral_step_kernel<FOLD, EARLY, CA>, grid, dim3(kT), 0, s, P, T, D, M, first, count, j1, dt, eps, cpl);
}
template <typename CA>
void dispatch_spectral_step(bool fold, bool early, dim3 grid, hipStream_t s, const ModelPtrs &P, const DeviceTables &T,
                            const DynDeviceTables &D, int M, int first, int count, int j1, double dt, double eps, const CA &cpl) {
    if (fold && early) launch_spectral_step<true, true, CA>(grid, s, P, T, D, M, first, count, j1, dt, eps, cpl);
    else if (fold) launch_spectral_step<true, false, CA>(grid, s, P, T, D, M, first, count, j1, dt, eps, cpl);
    else if (early) launch_spectral_step<false, true, CA>(grid, s, P, T, D, M, first, count, j1, dt, eps, cpl);
    else launch_spectral_step<false, false, CA>(grid, s, P, T, D, M, first, count, j1, dt, eps, cpl);
}
}  // namespace
hipError_t run_spectral_step(const ModelPtrs &P, const DeviceTables &T, const DynDeviceTables &D, int M, int first, int count,
                             int j1, double dt, double eps, const CouplerArgs *cpl, bool early, hipStream_t s) {
    const int nspec = (count * NSPEC * KX + kT - 1) / kT;
    const bool fold = P.phi_next != nullptr;
    if (cpl)
        dispatch_spectral_step(fold, early, dim3(nspec + (cpl->count * NG + kT - 1) / kT), s, P, T, D, M, first, count, j1, dt, eps,
                               *cpl);
    else
        dispatch_spectral_step(fold, early, dim3(nspec), s, P, T, D, M, first, count, j1, dt, eps, NoCoupler{});
    return hipGetLastError();
}
hipError_t run_diagnostics(const ModelPtrs &P, const DeviceTables &T, int M, int tl, int *err, double *diag, int ticket,
                           hipStream_t s) {
    hipLaunchKernelGGL(diagnostics_kernel, dim3(M), dim3(64 * KX), 0, s, CheckArgs{P.vor, P.div, P.t, tl, err, diag, ticket}, T);
    return hipGetLastError();
}
// ... of the members [first, first + count) only (err, diag: the whole model's arrays)
hipError_t run_diagnostics_range(const ModelPtrs &P, const DeviceTables &T, int first, int count, int tl, int *err, double *diag,
                                 int ticket, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(diagnostics_kernel, dim3(count), dim3(64 * KX), 0, s, CheckArgs{P.vor, P.div, P.t, tl, err, diag, ticket, first}, T);
    return hipGetLastError();
}

}  // namespace spd
