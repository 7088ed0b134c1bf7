// Fused column physics for gfx950: everything get_physical_tendencies (speedy.f90/physics.f90:14-256) does after
// its spectral transforms, for all columns of all ensemble members in ONE kernel launch.
//
//   thermodynamics (physics.f90:107-116, humidity.f90:17-78) -> deep convection (convection.f90:27-253) ->
//   large-scale condensation (large_scale_condensation.f90:33-96) -> [every third step: clouds + shortwave
//   (shortwave_radiation.f90:50-214, 325-404)] -> longwave down (longwave_radiation.f90:16-121) -> surface fluxes
//   and skin temperature (surface_fluxes.f90:40-320) -> longwave up (:124-205) -> vertical diffusion / shallow
//   convection (vertical_diffusion.f90:30-146) -> flux-to-tendency conversion (physics.f90:127-130, 166-168,
//   207-209, 223-231).
//
// Mapping: one lane per column, a wavefront = 64 consecutive longitudes of one latitude row; the eight sigma
// levels live in registers, every level loop is fully unrolled.  All global accesses are unit-stride across lanes
// (fields are (ix, il, kx) with ix fastest), so each wave-level load/store is one 512-byte burst.  The reference
// heap-allocates ~45 full 3-D temporaries per call; here none of them ever reaches memory.
//
// Constants: the reference's default-real (fp32) literals are written as float literals and widened, exactly
// like the Fortran expression rules do.  Order of operations follows the reference; the
// device may contract a*b+c into an FMA, and exp() is the device library's (<= 1 ulp), so results agree with the
// reference to ~1e-15 relative, not bit for bit.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/pyspeedy_amd.h"
#include "device_tables.hpp"
#include "dyn_column.hpp"
#include "stream_store.hpp"

namespace spd {

namespace {

constexpr int NG = IX * IL;  // columns per member
constexpr int kPhysThreads = 64;

// physical_constants.f90 / mod_radcon.f90
__device__ constexpr double P0 = 1.e+5f, CP = 1004.0f, GRAV = 9.81f, ALHC = 2501.0f, SBC = 5.67e-8f;
__device__ constexpr double AKAP = 2.0f / 7.0f;
__device__ constexpr double RGAS = AKAP * CP;
__device__ constexpr double EPSLW = 0.05f, EMISFC = 0.98f;

__device__ inline double dmin(double a, double b) { return a < b ? a : b; }
__device__ inline double dmax(double a, double b) { return a > b ? a : b; }
__device__ inline double pow3(double x) { return (x * x) * x; }
__device__ inline double pow4(double x) { const double x2 = x * x; return x2 * x2; }

// humidity.f90:44-78 for one point, P = sig * ps
__device__ inline double qsat_point(double ta, double p) {
    const double e0 = 6.108e-3, c1 = 17.269f, c2 = 21.875f, t0 = 273.16f, t1 = 35.86f, t2 = 7.66f;
    const double e = (ta >= t0) ? e0 * exp(c1 * (ta - t0) / (ta - t1)) : e0 * exp(c2 * (ta - t0) / (ta - t2));
    return 622.0f * e / (p - 0.378f * e);
}

// fband(nint(T), band): the reference does not clamp (model_state_def.py:425-430 sizes the table 100:400);
// the clamp below only matters where the reference would read out of bounds.
__device__ inline double fband_at(const double *fband, double temp, int band /*0-based*/) {
    int it = static_cast<int>(round(temp));
    it = it < 100 ? 100 : (it > 400 ? 400 : it);
    return fband[(it - 100) + 301 * band];
}

struct Col {  // per-column pointers resolved once
    size_t p2;  // offset of this column in a (ix,il) plane array of this member: m*NG + p
};

}  // namespace

// W = minimum waves per SIMD the register allocator must leave room for (launch-bounds hint): 2 by default (256 VGPRs, a
// handful of spilled values), 1 with PYSPEEDY_AMD_PHYS_WAVES=1 for comparison; 3 and 4 spill heavily and were 1.5-2x slower.
//
// FUSED: the kernel first runs the grid-point dynamics of the column (dyn_column.hpp, tendencies.f90:125-224) and keeps the
// temperature / humidity tendencies it produces in registers -- the physics adds to exactly those -- so they are written
// once instead of written by one kernel and read and re-written by the next (the model step; 36 field moves per member
// less).  Not FUSED: the tendencies are read from a.ttend / a.qtend / a.utend / a.vtend (the C ABI's spd_physics).
template <int W, bool FUSED>
__global__ __launch_bounds__(kPhysThreads, W) void physics_kernel(spd_physics_args a, DeviceTables T, int first, int nmembers,
                                                                  ModelPtrs MP, DynDeviceTables MD) {
    // rad_tau2 of this lane's column while the two longwave sweeps run: [band * 8 + level][lane], 16 KB per wavefront
    __shared__ double tau_s[4 * KX][kPhysThreads];
    const int gid = blockIdx.x * kPhysThreads + threadIdx.x;
    if (gid >= nmembers * NG) return;
    const int lmem = gid / NG, mem = first + lmem, p = gid - lmem * NG, j = p / IX;  // members [first, first + nmembers)
    const int lane = threadIdx.x;
    const size_t o2 = static_cast<size_t>(mem) * NG + p;              // (ix,il)
    const size_t o3 = static_cast<size_t>(mem) * NG * KX + p;         // (ix,il,kx), + NG*k
    const size_t oa = static_cast<size_t>(mem) * NG * 3 + p;          // (ix,il,3),  + NG*c
    const size_t of4 = static_cast<size_t>(mem) * NG * 4 + p;         // (ix,il,4)
    const size_t ot = static_cast<size_t>(mem) * NG * KX * 4 + p;     // (ix,il,kx,4), + NG*(k + KX*b)
    const size_t os = static_cast<size_t>(mem) * NG * KX * 2 + p;     // (ix,il,kx,2)
    const size_t oc = static_cast<size_t>(mem) * NG * 2 + p;          // (ix,il,2)
    constexpr int nl1 = KX - 1;  // 1-based index of the level above the lowest; 0-based index nl1-1

    // ------------------------------------------------------------------ tendencies so far (dynamics)
    // FUSED: computed here and parked in LDS (the rows tau_s uses later) until the condensation scheme picks them up,
    // so that they do not occupy 32 VGPRs through the convection scheme, the register peak of the kernel.
    double utend_dyn = 0.0, vtend_dyn = 0.0;
    if (FUSED) {
        double tt[KX], qt[KX];
        dyn_column<false>(MP, MD, mem, p, j, tt, qt, utend_dyn, vtend_dyn);
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            tau_s[k][lane] = tt[k];
            tau_s[KX + k][lane] = qt[k];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the physics' loads below: the two phases must not add their registers
    }

    // ------------------------------------------------------------------ thermodynamics, physics.f90:107-116
    double ta[KX], qa[KX], phi[KX], se[KX], qsat[KX], rh[KX];
    const double psa = exp(a.pslg[o2]);
    const double rps = 1.0f / psa;
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        ta[k] = stream_load(&a.tg[o3 + NG * k]);
        qa[k] = dmax(stream_load(&a.qg[o3 + NG * k]), 0.0f);
        phi[k] = stream_load(&a.phig[o3 + NG * k]);
        se[k] = CP * ta[k] + phi[k];
        qsat[k] = qsat_point(ta[k], T.fsg[k] * psa);
        rh[k] = qa[k] / qsat[k];
    }
    // ------------------------------------------------------------------ deep convection, convection.f90
    const double psmin = 0.8f, trcnv = 6.0f, rhbl = 0.9f, rhil = 0.7f, entmax = 0.5f, smf = 0.8f;
    int itop = KX + 1;
    double qdif = 0.0, cbmf = 0.0, precnv = 0.0;
    double dfse[KX], dfqa[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) dfse[k] = dfqa[k] = 0.0;
    if (psa > psmin) {  // diagnose_convection, convection.f90:170-253
        const double mse0 = se[KX - 1] + ALHC * qa[KX - 1];
        double mse1 = se[nl1 - 1] + ALHC * qa[nl1 - 1];
        mse1 = dmin(mse0, mse1);
        const double mss_kx = se[KX - 1] + ALHC * qsat[KX - 1];
        const double mss0 = dmax(mse0, mss_kx);
        int ktop1 = KX, ktop2 = KX;
        double msthr = 0.0;
#pragma unroll
        for (int k = KX - 3; k >= 3; --k) {  // 1-based k
            const double mss_k = se[k - 1] + ALHC * qsat[k - 1], mss_k1 = se[k] + ALHC * qsat[k];
            const double mss2 = mss_k + T.wvi[8 + k - 1] * (mss_k1 - mss_k);
            if (mss0 > mss2) ktop1 = k;
            if (mse1 > mss2) {
                ktop2 = k;
                msthr = mss2;
            }
        }
        if (ktop1 < KX) {
            const double qthr0 = rhbl * qsat[KX - 1], qthr1 = rhbl * qsat[nl1 - 1];
            const bool lqthr = (qa[KX - 1] > qthr0) && (qa[nl1 - 1] > qthr1);
            if (ktop2 < KX) {
                itop = ktop1;
                qdif = dmax(qa[KX - 1] - qthr0, (mse0 - msthr) * (1.0 / ALHC));
            } else if (lqthr) {
                itop = ktop1;
                qdif = qa[KX - 1] - qthr0;
            }
        }
    }
    if (itop != KX + 1) {  // convection.f90:78-156
        const double fqmax = 5.0f;
        const double fm0 = P0 * T.dhs[KX - 1] / (GRAV * trcnv * 3600.0f);
        const double rdps = 2.0f / (1.0f - psmin);
        double entr[KX];  // entr(2:kx-1), 0-based index k-1
        double sentr = 0.0;
#pragma unroll
        for (int k = 2; k <= nl1; ++k) {
            const double d = dmax(0.0f, T.fsg[k - 1] - 0.5f);
            entr[k - 1] = d * d;
            sentr = sentr + entr[k - 1];
        }
        sentr = entmax / sentr;
#pragma unroll
        for (int k = 2; k <= nl1; ++k) entr[k - 1] = entr[k - 1] * sentr;

        const double wv_nl1 = T.wvi[8 + nl1 - 1];
        const double qmax = dmax(1.01f * qa[KX - 1], qsat[KX - 1]);
        double sb = se[nl1 - 1] + wv_nl1 * (se[KX - 1] - se[nl1 - 1]);
        double qb = qa[nl1 - 1] + wv_nl1 * (qa[KX - 1] - qa[nl1 - 1]);
        qb = dmin(qb, qa[KX - 1]);
        const double fpsa = psa * dmin(1.0f, (psa - psmin) * rdps);
        double fmass = fm0 * fpsa * dmin(fqmax, qdif / (qmax - qb));
        cbmf = fmass;
        double fus = fmass * se[KX - 1], fuq = fmass * qmax;
        double fds = fmass * sb, fdq = fmass * qb;
        dfse[KX - 1] = fds - fus;
        dfqa[KX - 1] = fdq - fuq;
#pragma unroll
        for (int k = KX - 1; k >= 2; --k) {  // 1-based k, active while k >= itop+1
            if (k >= itop + 1) {
                dfse[k - 1] = fus - fds;
                dfqa[k - 1] = fuq - fdq;
                const double enmass = entr[k - 1] * psa * cbmf;
                fmass = fmass + enmass;
                fus = fus + enmass * se[k - 1];
                fuq = fuq + enmass * qa[k - 1];
                const double wv = T.wvi[8 + k - 2];
                sb = se[k - 2] + wv * (se[k - 1] - se[k - 2]);
                qb = qa[k - 2] + wv * (qa[k - 1] - qa[k - 2]);
                fds = fmass * sb;
                fdq = fmass * qb;
                dfse[k - 1] = dfse[k - 1] + fds - fus;
                dfqa[k - 1] = dfqa[k - 1] + fdq - fuq;
                const double delq = rhil * qsat[k - 1] - qa[k - 1];
                if (delq > 0.0) {
                    const double fsq = smf * cbmf * delq;
                    dfqa[k - 1] = dfqa[k - 1] + fsq;
                    dfqa[KX - 1] = dfqa[KX - 1] - fsq;
                }
            }
        }
        // top layer k = itop (3 <= itop <= kx-3): static indexing through a select chain keeps arrays in registers
        double qs_t = 0.0, qs_t1 = 0.0, wv_t = 0.0;
#pragma unroll
        for (int k = 3; k <= KX - 3; ++k)
            if (k == itop) {
                qs_t = qsat[k - 1];
                qs_t1 = qsat[k];
                wv_t = T.wvi[8 + k - 1];
            }
        const double qsatb = qs_t + wv_t * (qs_t1 - qs_t);
        precnv = dmax(fuq - fmass * qsatb, 0.0);
#pragma unroll
        for (int k = 3; k <= KX - 3; ++k)
            if (k == itop) {
                dfse[k - 1] = fus - fds + ALHC * precnv;
                dfqa[k - 1] = fuq - fdq - precnv;
            }
    }
    stream_store(&a.cbmf[o2], cbmf);
    stream_store(&a.precnv[o2], precnv);
    const int icnv = KX - itop;  // physics.f90:132
    int iptop = itop;

    // ------------------------------------------------------------------ large-scale condensation
    double ttend[KX], qtend[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        ttend[k] = FUSED ? tau_s[k][lane] : a.ttend[o3 + NG * k];
        qtend[k] = FUSED ? tau_s[KX + k][lane] : a.qtend[o3 + NG * k];
    }
    double precls = 0.0;
    {
        const double trlsc = 4.0f, rhlsc = 0.9f, drhlsc = 0.1f, rhblsc = 0.95f, qsmax = 10.0f;
        const double rtlsc = 1.0f / (trlsc * 3600.0f), tfact = ALHC / CP, prg = P0 / GRAV;
        const double psa2 = psa * psa;
#pragma unroll
        for (int k = 2; k <= KX; ++k) {
            const double sig2 = T.fsg[k - 1] * T.fsg[k - 1];
            double rhref = rhlsc + drhlsc * (sig2 - 1.0f);
            if (k == KX) rhref = dmax(rhref, rhblsc);
            const double dqmax = qsmax * sig2 * rtlsc;
            const double dqa = rhref * qsat[k - 1] - qa[k - 1];
            double dq = 0.0, dt = 0.0;
            if (dqa < 0.0) {
                iptop = k < iptop ? k : iptop;
                dq = dqa * rtlsc;
                dt = tfact * dmin(-dq, dqmax * psa2);
            }
            // physics.f90:127-130, 138-139: ttend = ttend + tt_cnv + tt_lsc
            ttend[k - 1] = ttend[k - 1] + dfse[k - 1] * rps * T.grdscp[k - 1] + dt;
            qtend[k - 1] = qtend[k - 1] + dfqa[k - 1] * rps * T.grdsig[k - 1] + dq;
            precls = precls - (T.dhs[k - 1] * prg) * dq;
        }
        // level 1: tt_cnv(1) = dfse(1) (unscaled, zero), tt_lsc(1) = 0
        ttend[0] = ttend[0] + dfse[0] + 0.0;
        qtend[0] = qtend[0] + dfqa[0] + 0.0;
        precls = precls * psa;
    }
    stream_store(&a.precls[o2], precls);

    const double gse = (se[nl1 - 1] - se[KX - 1]) / (phi[nl1 - 1] - phi[KX - 1]);  // physics.f90:152 (used on shortwave steps)
    const double phi_kx = phi[KX - 1];
    // ------------------------------------------------------------------ vertical diffusion (vertical_diffusion.f90)
    // Evaluated here, right after the moist schemes, although the reference calls it after the radiation: it only
    // depends on se, rh, qa, qsat, phi and icnv, and computing it now lets those 40 per-column values die before the
    // register-hungry radiation sweeps.  Its tendencies are ADDED in the reference's order at the end.
    const double trshc = 6.0f, trvdi = 24.0f, trvds = 6.0f, redshc = 0.5f, rhgrad = 0.5f, segrad = 0.1f;
    const double cshc = T.dhs[KX - 1] / 3600.0f;
    const double cvdi = (T.sigh[nl1] - T.sigh[1]) / (static_cast<float>(nl1 - 1) * 3600.0f);
    const double fshcq = cshc / trshc, fshcse = cshc / (trshc * CP);
    const double fvdiq = cvdi / trvdi, fvdise = cvdi / (trvds * CP);
    double ttv[KX], qtv[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) ttv[k] = qtv[k] = 0.0;
    {
        const double rs_nl1 = 1.0f / T.dhs[nl1 - 1], rs_kx = 1.0f / T.dhs[KX - 1];
        const double drh0 = rhgrad * (T.fsg[KX - 1] - T.fsg[nl1 - 1]);
        const double fvdiq2 = fvdiq * T.sigh[nl1];
        const double dmse = se[KX - 1] - se[nl1 - 1] + ALHC * (qa[KX - 1] - qsat[nl1 - 1]);
        const double drh = rh[KX - 1] - rh[nl1 - 1];
        if (dmse >= 0.0) {
            const double fcnv = (icnv > 0) ? redshc : static_cast<double>(1.0f);
            const double fluxse = fcnv * fshcse * dmse;
            ttv[nl1 - 1] = fluxse * rs_nl1;
            ttv[KX - 1] = -fluxse * rs_kx;
            if (drh >= 0.0) {
                const double fluxq = fcnv * fshcq * qsat[KX - 1] * drh;
                qtv[nl1 - 1] = fluxq * rs_nl1;
                qtv[KX - 1] = -fluxq * rs_kx;
            }
        } else if (drh > drh0) {
            const double fluxq = fvdiq2 * qsat[nl1 - 1] * drh;
            qtv[nl1 - 1] = fluxq * rs_nl1;
            qtv[KX - 1] = -fluxq * rs_kx;
        }
    }
#pragma unroll
    for (int k = 3; k <= KX - 2; ++k) {
        if (T.sigh[k] > 0.5f) {
            const double drh0 = rhgrad * (T.fsg[k] - T.fsg[k - 1]);
            const double fvdiq2 = fvdiq * T.sigh[k];
            const double drh = rh[k] - rh[k - 1];
            if (drh >= drh0) {
                const double fluxq = fvdiq2 * qsat[k - 1] * drh;
                qtv[k - 1] = qtv[k - 1] + fluxq * (1.0f / T.dhs[k - 1]);
                qtv[k] = qtv[k] - fluxq * (1.0f / T.dhs[k]);
            }
        }
    }
#pragma unroll
    for (int k = 1; k <= nl1; ++k) {
        const double se0 = se[k] + segrad * (phi[k - 1] - phi[k]);
        if (se[k - 1] < se0) {
            const double fluxse = fvdise * (se0 - se[k - 1]);
            ttv[k - 1] = ttv[k - 1] + fluxse * (1.0f / T.dhs[k - 1]);
            const double r1 = 1.0f / (1.0f - T.sigh[k]);
#pragma unroll
            for (int k1 = k + 1; k1 <= KX; ++k1) ttv[k1 - 1] = ttv[k1 - 1] - fluxse * r1;
        }
    }

    // The diffusion tendencies join ttend / qtend now (the reference adds them last, physics.f90:229-231; the sums differ
    // from its order by rounding only) so that ttv / qtv do not stay live through the radiation.  qtend is final above the
    // lowest level: store it.
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        ttend[k] = ttend[k] + ttv[k];
        qtend[k] = qtend[k] + qtv[k];
    }
    // SPPT (physics.f90:234-248, csrc/sppt.hip): tend = (1 + r mu(k)) (tend - tend_dyn) + tend_dyn with mu = 1; the
    // dynamics-only tendency is still in memory because this kernel writes each output once, here or at its end
    const bool sppt = !FUSED && a.sppt_pattern != nullptr;
    auto perturb = [&](double tend, double tend_dyn, int k) {
        const double r = a.sppt_pattern[o3 + NG * k];
        const double rc = dmin(1.0, fabs(r)) * (r < 0.0 ? -1.0 : 1.0);  // sppt.f90:112
        return (1.0 + rc) * (tend - tend_dyn) + tend_dyn;
    };
#pragma unroll
    for (int k = 0; k < KX - 1; ++k) a.qtend[o3 + NG * k] = sppt ? perturb(qtend[k], a.qtend[o3 + NG * k], k) : qtend[k];
    const double qtend_kx = qtend[KX - 1];

    // ------------------------------------------------------------------ clouds + shortwave (every nstrad-th step)
    // rad_tau2(k, band) lives in HBM between shortwave steps (physics.f90 keeps it in the model state); while the two
    // longwave sweeps run it sits in LDS (tau_s), read one band at a time instead of holding all 32 values in registers.
    double ssrd, strat1, strat2;
    int icltop = 0;
    double cloudc = 0.0, clstr = 0.0;
    if (a.compute_shortwave) {
        const double rhcl1 = 0.30f, rhcl2 = 1.00f, qacl = 0.20f, wpcl = 0.2f, pmaxcl = 10.0f, clsmax = 0.60f,
                     clsminl = 0.15f, gse_s0 = 0.25f, gse_s1 = 0.40f, albcl = 0.43f, albcls = 0.50f, absdry = 0.033f,
                     absaer = 0.033f, abswv1 = 0.022f, abswv2 = 15.000f, abscl1 = 0.015f, abscl2 = 0.15f,
                     ablwin = 0.3f, ablwv1 = 0.7f, ablwv2 = 50.0f, ablcl1 = 12.0f, ablcl2 = 0.6f;
        const double fmask = a.fmask_land[o2];
        // clouds, shortwave_radiation.f90:325-404
        const double rrcl = 1.f / (rhcl2 - rhcl1);
        if (rh[nl1 - 1] > rhcl1) {
            cloudc = rh[nl1 - 1] - rhcl1;
            icltop = nl1;
        } else {
            cloudc = 0.0;
            icltop = KX + 1;
        }
#pragma unroll
        for (int k = 3; k <= KX - 2; ++k) {
            const double drh = rh[k - 1] - rhcl1;
            if (drh > cloudc && qa[k - 1] > qacl) {
                cloudc = drh;
                icltop = k;
            }
        }
        const double pr1 = dmin(pmaxcl, 86.4f * (precnv + precls));
        const double cq = dmin(1.0f, cloudc * rrcl);
        cloudc = dmin(1.0f, wpcl * sqrt(pr1) + cq * cq);
        icltop = iptop < icltop ? iptop : icltop;
        const double qcloud = qa[nl1 - 1];
        stream_store(&a.qcloud_equiv[o2], qcloud);
        const double clfact = 1.2f, rgse = 1.0f / (gse_s1 - gse_s0);
        const double fstab = dmax(0.0f, dmin(1.0f, rgse * (gse - gse_s0)));
        clstr = fstab * dmax(clsmax - clfact * cloudc, 0.0f);
        const double clstrl = dmax(clstr, clsminl) * rh[KX - 1];
        clstr = clstr + fmask * (clstrl - clstr);

        // shortwave, shortwave_radiation.f90:50-214
        const double fband2 = 0.05f, fband1 = 1.0f - fband2;
        double refl[KX];  // rad_tau2(:, 3) during the shortwave sweep: cloud reflectivities, then reflected fluxes
#pragma unroll
        for (int k = 0; k < KX; ++k) refl[k] = 0.0;
#pragma unroll
        for (int k = 1; k <= KX; ++k)
            if (k == icltop) refl[k - 1] = albcl * cloudc;
        refl[KX - 1] = albcls * clstr;
        const double psaz = psa * a.zenit_correction[o2];
        double acloud = cloudc * dmin(abscl1 * qcloud, abscl2);
        double tsw1[KX], tsw2[KX];  // shortwave transmissivities, bands 1 and 2
        tsw1[0] = exp(-psaz * T.dhs[0] * absdry);
        tsw2[0] = 0.0;
#pragma unroll
        for (int k = 2; k <= nl1; ++k) {
            const double abs1 = absdry + absaer * (T.fsg[k - 1] * T.fsg[k - 1]);
            tsw1[k - 1] = (k >= icltop) ? exp(-psaz * T.dhs[k - 1] * (abs1 + abswv1 * qa[k - 1] + acloud))
                                        : exp(-psaz * T.dhs[k - 1] * (abs1 + abswv1 * qa[k - 1]));
        }
        {
            const double abs1 = absdry + absaer * (T.fsg[KX - 1] * T.fsg[KX - 1]);
            tsw1[KX - 1] = exp(-psaz * T.dhs[KX - 1] * (abs1 + abswv1 * qa[KX - 1]));
        }
#pragma unroll
        for (int k = 2; k <= KX; ++k) tsw2[k - 1] = exp(-psaz * T.dhs[k - 1] * abswv2 * qa[k - 1]);

        const double solar = a.flux_solar_in[o2];
        double tsr = solar;
        double tt_rsw[KX];
        double f1 = solar * fband1, f2 = solar * fband2;
        tt_rsw[0] = f1;
        f1 = tsw1[0] * (f1 - a.flux_ozone_upper[o2] * psa);
        tt_rsw[0] = tt_rsw[0] - f1;
        tt_rsw[1] = f1;
        f1 = tsw1[1] * (f1 - a.flux_ozone_lower[o2] * psa);
        tt_rsw[1] = tt_rsw[1] - f1;
#pragma unroll
        for (int k = 3; k <= KX; ++k) {
            refl[k - 1] = f1 * refl[k - 1];
            f1 = f1 - refl[k - 1];
            tt_rsw[k - 1] = f1;
            f1 = tsw1[k - 1] * f1;
            tt_rsw[k - 1] = tt_rsw[k - 1] - f1;
        }
#pragma unroll
        for (int k = 2; k <= KX; ++k) {
            tt_rsw[k - 1] = tt_rsw[k - 1] + f2;
            f2 = tsw2[k - 1] * f2;
            tt_rsw[k - 1] = tt_rsw[k - 1] - f2;
        }
        ssrd = f1 + f2;
        f1 = f1 * a.alb_surface[o2];
        stream_store(&a.ssrd[o2], ssrd);
        stream_store(&a.ssr[o2], ssrd - f1);
#pragma unroll
        for (int k = KX; k >= 1; --k) {
            tt_rsw[k - 1] = tt_rsw[k - 1] + f1;
            f1 = tsw1[k - 1] * f1;
            tt_rsw[k - 1] = tt_rsw[k - 1] - f1;
            f1 = f1 + refl[k - 1];
        }
        stream_store(&a.tsr[o2], tsr - f1);
        // physics.f90:166-168
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            tt_rsw[k] = tt_rsw[k] * rps * T.grdscp[k];
            stream_store(&a.tt_rsw[o3 + NG * k], tt_rsw[k]);
            ttend[k] = ttend[k] + tt_rsw[k];
        }

        // longwave transmissivities, shortwave_radiation.f90:170-208
        const double co2 = a.air_absortivity_co2;
        const size_t NGs = static_cast<size_t>(NG);
        tau_s[0][lane] = exp(-psa * T.dhs[0] * ablwin);
        tau_s[KX][lane] = exp(-psa * T.dhs[0] * co2);
        stream_store(&a.rad_tau2[ot + NGs * (0 + KX * 0)], tau_s[0][lane]);
        stream_store(&a.rad_tau2[ot + NGs * (0 + KX * 1)], tau_s[KX][lane]);
        stream_store(&a.rad_tau2[ot + NGs * (0 + KX * 2)], 1.0);
        stream_store(&a.rad_tau2[ot + NGs * (0 + KX * 3)], 1.0);
        acloud = cloudc * ablcl2;
#pragma unroll
        for (int k = 2; k <= KX; ++k) {
            double t0, t1, t2, t3;
            if (k == 2 || k == KX) {
                t0 = exp(-psa * T.dhs[k - 1] * ablwin);
                t1 = exp(-psa * T.dhs[k - 1] * co2);
                t2 = exp(-psa * T.dhs[k - 1] * ablwv1 * qa[k - 1]);
                t3 = exp(-psa * T.dhs[k - 1] * ablwv2 * qa[k - 1]);
            } else {
                const double deltap = psa * T.dhs[k - 1];
                const double acloud1 = (k < icltop) ? acloud : ablcl1 * cloudc;
                t0 = exp(-deltap * (ablwin + acloud1));
                t1 = exp(-deltap * co2);
                t2 = exp(-deltap * dmax(ablwv1 * qa[k - 1], acloud));
                t3 = exp(-deltap * dmax(ablwv2 * qa[k - 1], acloud));
            }
            stream_store(&a.rad_tau2[ot + NGs * (k - 1 + KX * 0)], t0);
            stream_store(&a.rad_tau2[ot + NGs * (k - 1 + KX * 1)], t1);
            stream_store(&a.rad_tau2[ot + NGs * (k - 1 + KX * 2)], t2);
            stream_store(&a.rad_tau2[ot + NGs * (k - 1 + KX * 3)], t3);
            tau_s[k - 1 + KX * 0][lane] = t0;
            tau_s[k - 1 + KX * 1][lane] = t1;
            tau_s[k - 1 + KX * 2][lane] = t2;
            tau_s[k - 1 + KX * 3][lane] = t3;
        }
        const double eps1 = EPSLW / (T.dhs[0] + T.dhs[1]);
        strat1 = a.stratospheric_correction[o2] * psa;
        strat2 = eps1 * psa;
        stream_store(&a.rad_strat_corr[oc], strat1);
        stream_store(&a.rad_strat_corr[oc + NG], strat2);
    } else {
#pragma unroll
        for (int k = 0; k < KX; ++k) ttend[k] = ttend[k] + a.tt_rsw[o3 + NG * k];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = (b < 2 ? 0 : 1); k < KX; ++k)  // bands 3-4 do not use the top level
                tau_s[k + KX * b][lane] = a.rad_tau2[ot + static_cast<size_t>(NG) * (k + KX * b)];
        ssrd = a.ssrd[o2];
        strat1 = a.rad_strat_corr[oc];
        strat2 = a.rad_strat_corr[oc + NG];
    }

    // ------------------------------------------------------------------ longwave, downward sweep
    double st4a[KX][2], dfabs[KX], flux[4];
    {
        const double anis = 1.0f;
#pragma unroll
        for (int k = 1; k <= nl1; ++k) st4a[k - 1][0] = ta[k - 1] + T.wvi[8 + k - 1] * (ta[k] - ta[k - 1]);
        st4a[0][1] = 0.75f * ta[0] + 0.25f * st4a[0][0];
        st4a[1][1] = 0.50f * ta[1] + 0.25f * (st4a[0][0] + st4a[1][0]);
#pragma unroll
        for (int k = 3; k <= nl1; ++k) st4a[k - 1][1] = 0.5f * anis * dmax(st4a[k - 1][0] - st4a[k - 2][0], 0.0f);
        st4a[KX - 1][1] = anis * dmax(ta[KX - 1] - st4a[nl1 - 1][0], 0.0f);
#pragma unroll
        for (int k = 1; k <= 2; ++k) {
            st4a[k - 1][0] = SBC * pow4(st4a[k - 1][1]);
            st4a[k - 1][1] = 0.0;
        }
#pragma unroll
        for (int k = 3; k <= KX; ++k) {
            const double st3a = SBC * pow3(ta[k - 1]);
            st4a[k - 1][0] = st3a * ta[k - 1];
            st4a[k - 1][1] = 4.0f * st3a * st4a[k - 1][1];
        }
#pragma unroll
        for (int k = 0; k < KX; ++k) dfabs[k] = 0.0;
        int itab[KX];  // nint(ta(k)) - 100, clamped: row of fband for level k (both sweeps)
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            int it = static_cast<int>(round(ta[k]));
            itab[k] = (it < 100 ? 100 : (it > 400 ? 400 : it)) - 100;
        }
        // Stratosphere (bands 1-2, :73-79) first for both bands, as the reference does: dfabs(1) sums in that order.
        double tau0[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            tau0[b] = tau_s[KX * b][lane];
            const double emis = 1.0f - tau0[b];
            const double brad = T.fband[itab[0] + 301 * b] * (st4a[0][0] + emis * st4a[0][1]);
            flux[b] = emis * brad;
            dfabs[0] = dfabs[0] - flux[b];
        }
        flux[2] = flux[3] = 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double tb[KX];
#pragma unroll
            for (int k = 2; k <= KX; ++k) tb[k - 1] = tau_s[k - 1 + KX * b][lane];
#pragma unroll
            for (int k = 2; k <= KX; ++k) {
                const double emis = 1.0f - tb[k - 1];
                const double brad = T.fband[itab[k - 1] + 301 * b] * (st4a[k - 1][0] + emis * st4a[k - 1][1]);
                dfabs[k - 1] = dfabs[k - 1] + flux[b];
                flux[b] = tb[k - 1] * flux[b] + emis * brad;
                dfabs[k - 1] = dfabs[k - 1] - flux[b];
            }
        }
        double slrd = 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) slrd = slrd + EMISFC * flux[b];
        const double corlw = EPSLW * EMISFC * st4a[KX - 1][0];
        dfabs[KX - 1] = dfabs[KX - 1] - corlw;
        slrd = slrd + corlw;
        stream_store(&a.slrd[o2], slrd);
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            stream_store(&a.rad_st4a[os + static_cast<size_t>(NG) * k], st4a[k][0]);
            stream_store(&a.rad_st4a[os + static_cast<size_t>(NG) * (k + KX)], st4a[k][1]);
        }

        // -------------------------------------------------------------- surface fluxes, surface_fluxes.f90:40-320
        const double fwind0 = 0.95f, ftemp0 = 1.0f, cdl = 2.4e-3f, cds = 1.0e-3f, chl = 1.2e-3f, chs = 0.9e-3f,
                     vgust = 5.0f, ctday = 1.0e-2f, dtheta = 3.0f, fstab = 0.67f, clambda = 7.0f, clambsn = 7.0f;
        const double esbc = EMISFC * SBC;
        const double ua = a.ug[o3 + NG * (KX - 1)], va = a.vg[o3 + NG * (KX - 1)];
        const double fmask = a.fmask_land[o2], phi0 = a.phis0[o2], tsea = a.sst_am[o2], land_temp = a.land_temp[o2];
        const double alb_land = a.alb_land[o2], swav = a.soil_avail_water[o2];
        const double u0 = fwind0 * ua, v0 = fwind0 * va;
        const double gtemp0 = 1.0f - ftemp0, rcp = 1.0f / CP;
        const double dt1 = T.wvi[8 + KX - 1] * (ta[KX - 1] - ta[nl1 - 1]);
        double t1l = ta[KX - 1] + dt1;
        double t1s = t1l - phi0 * dt1 / (RGAS * 288.0f * T.sigl[KX - 1]);
        const double t2s = ta[KX - 1] + rcp * phi_kx;
        const double t2l = t2s - rcp * phi0;
        if (ta[KX - 1] > ta[nl1 - 1]) {
            t1l = ftemp0 * t1l + gtemp0 * t2l;
            t1s = ftemp0 * t1s + gtemp0 * t2s;
        } else {
            t1l = ta[KX - 1];
            t1s = ta[KX - 1];
        }
        const double t0 = t1s + fmask * (t1l - t1s);
        const double den0 = (P0 * psa / (RGAS * t0)) * sqrt(u0 * u0 + v0 * v0 + vgust * vgust);
        double tskin = land_temp + ctday * sqrt(T.coa[j]) * ssrd * (1.0f - alb_land) * psa;
        const double rdth = fstab / dtheta, astab = 0.5f;
        const double dthl = (tskin > t2l) ? dmin(dtheta, tskin - t2l) : dmax(-dtheta, astab * (tskin - t2l));
        const double den1 = den0 * (1.0f + dthl * rdth);
        const double cdldv = cdl * den0 * a.forog[o2];
        const double ustr1 = -cdldv * ua, vstr1 = -cdldv * va;
        const double chlcp = chl * CP;
        double shf1 = chlcp * den1 * (tskin - t1l);
        const double q1 = qa[KX - 1];
        const double qs0l = qsat_point(tskin, 1.0 * psa);
        double evap1 = chl * den1 * dmax(0.0f, swav * qs0l - q1);
        const double tsk3 = pow3(tskin);
        const double dslr = 4.0f * esbc * tsk3;
        double slru1 = esbc * tsk3 * tskin;
        double hfl1 = ssrd * (1.0f - alb_land) + slrd - (slru1 + shf1 + ALHC * evap1);
        const double clamb = clambda + a.snowc[o2] * (clambsn - clambda);
        hfl1 = hfl1 - clamb * (tskin - land_temp);
        double dqs = qsat_point(tskin + 1.0f, 1.0 * psa);
        dqs = (evap1 > 0.0) ? swav * (dqs - qs0l) : 0.0;
        const double dtskin = hfl1 / (clamb + dslr + chl * den1 * (CP + ALHC * dqs));
        tskin = tskin + dtskin;
        shf1 = shf1 + chlcp * den1 * dtskin;
        evap1 = evap1 + chl * den1 * dqs * dtskin;
        slru1 = slru1 + dslr * dtskin;
        hfl1 = clamb * (tskin - land_temp);
        const double dths = (tsea > t2s) ? dmin(dtheta, tsea - t2s) : dmax(-dtheta, astab * (tsea - t2s));
        const double den2 = den0 * (1.0f + dths * rdth);
        const double cdsdv = cds * den2;
        const double ustr2 = -cdsdv * ua, vstr2 = -cdsdv * va;
        const double shf2 = chs * CP * den2 * (tsea - t1s);
        const double qs0s = qsat_point(tsea, 1.0 * psa);
        const double evap2 = chs * den2 * (qs0s - q1);
        const double slru2 = esbc * pow4(tsea);
        const double hfl2 = ssrd * (1.0f - a.alb_sea[o2]) + slrd - slru2 + shf2 + ALHC * evap2;
        const double ustr3 = ustr2 + fmask * (ustr1 - ustr2), vstr3 = vstr2 + fmask * (vstr1 - vstr2);
        const double shf3 = shf2 + fmask * (shf1 - shf2), evap3 = evap2 + fmask * (evap1 - evap2);
        const double slru3 = slru2 + fmask * (slru1 - slru2);
        const double tsfc = tsea + fmask * (land_temp - tsea);
        const double tskin_avg = tsea + fmask * (tskin - tsea);
        stream_store(&a.ustr[oa], ustr1); a.ustr[oa + NG] = ustr2; a.ustr[oa + 2 * NG] = ustr3;
        stream_store(&a.vstr[oa], vstr1); a.vstr[oa + NG] = vstr2; a.vstr[oa + 2 * NG] = vstr3;
        stream_store(&a.shf[oa], shf1);   a.shf[oa + NG] = shf2;   a.shf[oa + 2 * NG] = shf3;
        stream_store(&a.evap[oa], evap1); a.evap[oa + NG] = evap2; a.evap[oa + 2 * NG] = evap3;
        stream_store(&a.slru[oa], slru1); a.slru[oa + NG] = slru2; a.slru[oa + 2 * NG] = slru3;
        stream_store(&a.hfluxn[oa], hfl1); a.hfluxn[oa + NG] = hfl2;
        if (a.ts) stream_store(&a.ts[o2], tsfc);
        if (a.tskin) stream_store(&a.tskin[o2], tskin_avg);
        if (a.u0) stream_store(&a.u0[o2], u0);
        if (a.v0) stream_store(&a.v0[o2], v0);
        if (a.t0) stream_store(&a.t0[o2], t0);

        // -------------------------------------------------------------- longwave, upward sweep (:124-205)
        const double refsfc = 1.0f - EMISFC;
        stream_store(&a.slr[o2], slru3 - slrd);
#pragma unroll
        for (int b = 0; b < 4; ++b) flux[b] = fband_at(T.fband, tsfc, b) * slru3 + refsfc * flux[b];
        dfabs[KX - 1] = dfabs[KX - 1] + EPSLW * slru3;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double tb[KX];
#pragma unroll
            for (int k = 2; k <= KX; ++k) tb[k - 1] = tau_s[k - 1 + KX * b][lane];
#pragma unroll
            for (int k = KX; k >= 2; --k) {
                const double emis = 1.0f - tb[k - 1];
                const double brad = T.fband[itab[k - 1] + 301 * b] * (st4a[k - 1][0] - emis * st4a[k - 1][1]);
                dfabs[k - 1] = dfabs[k - 1] + flux[b];
                flux[b] = tb[k - 1] * flux[b] + emis * brad;
                dfabs[k - 1] = dfabs[k - 1] - flux[b];
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const double emis = 1.0f - tau0[b];
            const double brad = T.fband[itab[0] + 301 * b] * (st4a[0][0] - emis * st4a[0][1]);
            dfabs[0] = dfabs[0] + flux[b];
            flux[b] = tau0[b] * flux[b] + emis * brad;
            dfabs[0] = dfabs[0] - flux[b];
        }
        const double corlw1 = T.dhs[0] * strat2 * st4a[0][0] + strat1;
        const double corlw2 = T.dhs[1] * strat2 * st4a[1][0];
        dfabs[0] = dfabs[0] - corlw1;
        dfabs[1] = dfabs[1] - corlw2;
        double olr = corlw1 + corlw2;
#pragma unroll
        for (int b = 0; b < 4; ++b) olr = olr + flux[b];
        stream_store(&a.olr[o2], olr);
#pragma unroll
        for (int b = 0; b < 4; ++b) a.rad_flux[of4 + static_cast<size_t>(NG) * b] = flux[b];
        // physics.f90:207-211: ttend = ttend + tt_rsw + tt_rlw
#pragma unroll
        for (int k = 0; k < KX; ++k) ttend[k] = ttend[k] + dfabs[k] * rps * T.grdscp[k];

        // physics.f90:223-231: surface-flux tendencies at the lowest level, then accumulate
        const double ut_kx = 0.0 + ustr3 * rps * T.grdsig[KX - 1];
        const double vt_kx = 0.0 + vstr3 * rps * T.grdsig[KX - 1];
        ttend[KX - 1] = ttend[KX - 1] + shf3 * rps * T.grdscp[KX - 1];
        const size_t okx = o3 + static_cast<size_t>(NG) * (KX - 1);
        const double ud = FUSED ? utend_dyn : a.utend[okx], vd = FUSED ? vtend_dyn : a.vtend[okx];
        const double qkx = qtend_kx + evap3 * rps * T.grdsig[KX - 1];
        if (sppt) {  // (above the lowest level the physics leaves the wind tendencies alone: nothing to perturb there)
            stream_store(&a.utend[okx], perturb(ud + ut_kx, ud, KX - 1));
            stream_store(&a.vtend[okx], perturb(vd + vt_kx, vd, KX - 1));
#pragma unroll
            for (int k = 0; k < KX; ++k) a.ttend[o3 + NG * k] = perturb(ttend[k], a.ttend[o3 + NG * k], k);
            stream_store(&a.qtend[okx], perturb(qkx, a.qtend[okx], KX - 1));
        } else {
            stream_store(&a.utend[okx], ud + ut_kx);  // ut_pbl, vt_pbl are zero above the lowest level
            stream_store(&a.vtend[okx], vd + vt_kx);
#pragma unroll
            for (int k = 0; k < KX; ++k) a.ttend[o3 + NG * k] = ttend[k];
            stream_store(&a.qtend[okx], qkx);
        }
    }
    if (a.iptop) a.iptop[o2] = iptop;
    if (a.icltop) a.icltop[o2] = icltop;
    if (a.cloudc) stream_store(&a.cloudc[o2], cloudc);
    if (a.clstr) stream_store(&a.clstr[o2], clstr);
}

static int physics_waves() {
    static const int waves = [] {
        const char *e = getenv("PYSPEEDY_AMD_PHYS_WAVES");
        return e ? atoi(e) : 2;
    }();
    return waves;
}

hipError_t run_physics(const DeviceTables &T, const spd_physics_args &a, int nmembers, hipStream_t s) {
    const long total = static_cast<long>(nmembers) * NG;
    const unsigned blocks = static_cast<unsigned>((total + kPhysThreads - 1) / kPhysThreads);
    const ModelPtrs mp{};
    const DynDeviceTables md{};
    if (physics_waves() == 1)
        hipLaunchKernelGGL((physics_kernel<1, false>), dim3(blocks), dim3(kPhysThreads), 0, s, a, T, 0, nmembers, mp, md);
    else
        hipLaunchKernelGGL((physics_kernel<2, false>), dim3(blocks), dim3(kPhysThreads), 0, s, a, T, 0, nmembers, mp, md);
    return hipGetLastError();
}

// grid-point dynamics + physics of every column in one launch (the model step); a.ttend / a.qtend / a.utend / a.vtend must be
// the dynamics' tendency arrays (P.ttend, P.trtend, P.utend, P.vtend)
hipError_t run_dyn_physics(const ModelPtrs &P, const DynDeviceTables &D, const DeviceTables &T, const spd_physics_args &a,
                           int first, int nmembers, hipStream_t s) {
    const long total = static_cast<long>(nmembers) * NG;
    const unsigned blocks = static_cast<unsigned>((total + kPhysThreads - 1) / kPhysThreads);
    if (physics_waves() == 1)
        hipLaunchKernelGGL((physics_kernel<1, true>), dim3(blocks), dim3(kPhysThreads), 0, s, a, T, first, nmembers, P, D);
    else
        hipLaunchKernelGGL((physics_kernel<2, true>), dim3(blocks), dim3(kPhysThreads), 0, s, a, T, first, nmembers, P, D);
    return hipGetLastError();
}

}  // namespace spd
