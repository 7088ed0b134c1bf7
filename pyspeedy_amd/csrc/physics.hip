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
#include <type_traits>

#include "../../include/pyspeedy_amd.h"
#include "device_tables.hpp"
#include "dyn_column.hpp"
#include "launch_events.hpp"
#include "stream_store.hpp"
#include "vertical_consts.hpp"

namespace spd {

namespace {

constexpr int NG = IX * IL;  // columns per member
#ifndef SPD_PHYS_THREADS
#define SPD_PHYS_THREADS 64  // one wavefront per workgroup (128 / 256 measured in round 4: profiles/r04_column_workgroup_size.txt)
#endif
constexpr int kPhysThreads = SPD_PHYS_THREADS;

// Arithmetic type of the column physics: R = double reproduces the reference (fp64 everywhere); R = float is BASELINE cfg 5's
// mixed precision -- state, dynamics tendencies and every array in memory stay fp64, the column arithmetic runs in fp32 and
// the physics increment is added to the fp64 dynamics tendency at the end.
//
// physical_constants.f90 / mod_radcon.f90: default-real literals, widened for R = double as the Fortran does
template <typename R>
struct PhysConst {
    static constexpr R P0 = 1.e+5f, CP = 1004.0f, GRAV = 9.81f, ALHC = 2501.0f, SBC = 5.67e-8f;
    static constexpr R AKAP = static_cast<R>(2.0f / 7.0f);
    static constexpr R RGAS = AKAP * CP;
    static constexpr R EPSLW = 0.05f, EMISFC = 0.98f;
};

template <typename R> __device__ __forceinline__ R rmin(R a, R b) { return a < b ? a : b; }
template <typename R> __device__ __forceinline__ R rmax(R a, R b) { return a > b ? a : b; }
template <typename R> __device__ __forceinline__ R pow3(R x) { return (x * x) * x; }
template <typename R> __device__ __forceinline__ R pow4(R x) { const R x2 = x * x; return x2 * x2; }
__device__ __forceinline__ double rexp(double x) { return exp(x); }
__device__ __forceinline__ float rexp(float x) { return expf(x); }
__device__ __forceinline__ double rsqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float rsqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ int rnint(double x) { return static_cast<int>(round(x)); }
__device__ __forceinline__ int rnint(float x) { return static_cast<int>(roundf(x)); }

// humidity.f90:44-78 for one point, P = sig * ps
template <typename R>
__device__ __forceinline__ R qsat_point(R ta, R p) {
    const R e0 = static_cast<R>(6.108e-3), c1 = 17.269f, c2 = 21.875f, t0 = 273.16f, t1 = 35.86f, t2 = 7.66f;
    const R e = (ta >= t0) ? e0 * rexp(c1 * (ta - t0) / (ta - t1)) : e0 * rexp(c2 * (ta - t0) / (ta - t2));
    return R(622.0f) * e / (p - R(0.378f) * e);
}

// fband(nint(T), band): the reference does not clamp (model_state_def.py:425-430 sizes the table 100:400);
// the clamp below only matters where the reference would read out of bounds.
template <typename R>
__device__ __forceinline__ R fband_at(const R *fband, R temp, int band /*0-based*/) {
    int it = rnint(temp);
    it = it < 100 ? 100 : (it > 400 ? 400 : it);
    return fband[(it - 100) + 301 * band];
}

// The tables the column physics indexes at run time, in the arithmetic type of the kernel.  The vertical-structure tables (fsg,
// dhs, sigl, sigh, grdsig, grdscp, wvi of geometry.f90:89-140) are compile-time constants (vertical_consts.hpp, generated from the
// host tables and checked against them when a context is made): every level loop is unrolled, so they reach the instructions as
// literals, and what depends on them alone -- the entrainment profile, 1 / dhs, the diffusion coefficients -- is folded by the
// compiler.  As kernel arguments they were 104 SGPRs (of 102 a wavefront has) that the register allocator kept alive by spilling
// them into VGPR lanes: 590 v_readlane / 130 v_writelane / 390 s_nop of the 7950 instructions of the fp64 kernel.
template <typename R>
struct ColTables {
    const R *fband;  // (301,4)
    const R *coa;    // 48, cos(latitude)
};

template <typename R>
__device__ __forceinline__ constexpr R vert(const double *table, int i) {
    return static_cast<R>(table[i]);
}

template <typename R> struct TablePtrs;
template <> struct TablePtrs<double> {
    static const double *fband(const DeviceTables &T) { return T.fband; }
    static const double *coa(const DeviceTables &T) { return T.coa; }
};
template <> struct TablePtrs<float> {
    static const float *fband(const DeviceTables &T) { return T.fband32; }
    static const float *coa(const DeviceTables &T) { return T.coa32; }
};

template <typename R>
ColTables<R> col_tables(const DeviceTables &T) {
    ColTables<R> c;
    c.fband = TablePtrs<R>::fband(T);
    c.coa = TablePtrs<R>::coa(T);
    return c;
}

}  // namespace

// W = minimum waves per SIMD the register allocator must leave room for (launch-bounds hint): 2 by default (256 VGPRs, no
// scratch), 1 with PYSPEEDY_AMD_PHYS_WAVES=1 for comparison; 3 and 4 spill heavily and were 1.5-2x slower.
//
// FUSED: the kernel first runs the grid-point dynamics of the column (dyn_column.hpp, tendencies.f90:125-224) and keeps the
// temperature / humidity tendencies it produces in registers -- the physics adds to exactly those -- so they are written
// once instead of written by one kernel and read and re-written by the next (the model step; 36 field moves per member
// less).  Not FUSED: the tendencies are read from a.ttend / a.qtend / a.utend / a.vtend (the C ABI's spd_physics).
// KEEP (FUSED only): the dynamics' temperature tendencies stay in LDS until the end of the kernel -- needed when the result is
// not simply "dynamics + physics summed in R": mixed precision (R = float) and SPPT both combine the fp64 dynamics
// tendency with the physics increment at the end.
template <int W, bool FUSED, bool KEEP, typename R, bool S32 = false>
__global__ __launch_bounds__(kPhysThreads, W) void physics_kernel(spd_physics_args a, ColTables<R> CT, int first, int nmembers,
                                                                  ModelPtrs MP, DynDeviceTables MD, int diag) {
    using C = PhysConst<R>;
    constexpr bool MIXED = !std::is_same<R, double>::value;
    // S32 (the model's cfg 5 step: fp32 arithmetic, fused; model option physics_storage32, default on): what only this kernel ever reads back -- its grid-point inputs at the
    // physics' time level (written by the spectral -> grid launch), the radiation state a shortwave step leaves for the next
    // two steps, the diagnostics-only outputs -- lives in memory as fp32, in the first half of the same arrays.  The kernel
    // narrows every one of those values to fp32 before it uses it and computes every one it stores in fp32, so nothing is lost
    // against fp64 storage: only the bytes are (13 % of this kernel's, 11 % of the spectral -> grid launch's).  State, dynamics
    // and the tendencies handed to the forward transforms stay fp64.
    static_assert(!S32 || (MIXED && FUSED), "fp32 storage belongs to the fused mixed-precision kernel");
    using IN = typename std::conditional<S32, float, double>::type;
    auto ld = [](const double *base, size_t i) -> IN {  // (streamed in: read once, by this lane only)
        if constexpr (S32) return stream_load(reinterpret_cast<const float *>(base) + i);
        else return stream_load(base + i);
    };
    auto ld_plain = [](const double *base, size_t i) -> IN {
        if constexpr (S32) return reinterpret_cast<const float *>(base)[i];
        else return base[i];
    };
    auto st_stream = [](double *base, size_t i, R v) {
        if constexpr (S32) stream_store(reinterpret_cast<float *>(base) + i, v);
        else stream_store(base + i, v);
    };
    auto st_plain = [](double *base, size_t i, R v) {
        if constexpr (S32) reinterpret_cast<float *>(base)[i] = v;
        else base[i] = v;
    };
    static_assert(!KEEP || FUSED, "KEEP is a variant of the fused kernel");
    static_assert(!(MIXED && FUSED) || KEEP, "the fused mixed-precision kernel keeps the fp64 dynamics tendencies");
    // LDS of the wavefront.  tau: rad_tau2 of this lane's column while the two longwave sweeps run, [band * 8 + level][lane].
    // park_t / park_q: the dynamics' T / q tendencies (fp64) from the end of the dynamics phase until the condensation
    // scheme picks them up; they alias tau (dead by then) except for park_t under KEEP, which lives to the end.
    constexpr int kRowD = kPhysThreads * sizeof(double);
    constexpr int kTauBytes = 4 * KX * kPhysThreads * sizeof(R), kKeepBytes = KEEP ? KX * kRowD : 0;
    constexpr int kAliasBytes = (KEEP ? 1 : 2) * KX * kRowD;  // parked tendencies that share the tau area
    __shared__ __attribute__((aligned(16))) unsigned char smem[kKeepBytes + (kTauBytes > kAliasBytes ? kTauBytes : kAliasBytes)];
    R(*tau_s)[kPhysThreads] = reinterpret_cast<R(*)[kPhysThreads]>(smem + kKeepBytes);
    double(*park_t)[kPhysThreads] = reinterpret_cast<double(*)[kPhysThreads]>(smem);
    double(*park_q)[kPhysThreads] = reinterpret_cast<double(*)[kPhysThreads]>(smem + KX * kRowD);
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
    // FUSED: computed here and parked in LDS until the condensation scheme picks them up, so that they do not occupy 32
    // VGPRs through the convection scheme, the register peak of the kernel.
    // (the dynamics' wind tendencies of the lowest level wait in LDS for the surface stress at the very end of the kernel:
    // kept in registers they are spilled to scratch memory by the allocator, 16 bytes out and back per lane)
    __shared__ double park_uv[2][kPhysThreads];
    // (FUSED, fp64: the physics' first inputs are requested by the dynamics phase, in front of its product stores)
    IN in_t[KX], in_q[KX], in_phi[KX], in_ps;
    auto load_column = [&]() {
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            in_t[k] = ld(a.tg, o3 + NG * k);
            in_q[k] = ld(a.qg, o3 + NG * k);
            in_phi[k] = ld(a.phig, o3 + NG * k);
        }
        in_ps = ld_plain(a.pslg, o2);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (FUSED) {
        double tt[KX], qt[KX], utend_dyn = 0.0, vtend_dyn = 0.0;
        if constexpr (MIXED)  // (the 3-wave fp32 kernel has no registers to spare for that: +4 % when tried)
            dyn_column<false>(MP, MD, mem, p, j, tt, qt, utend_dyn, vtend_dyn);
        else
            dyn_column<false>(MP, MD, mem, p, j, tt, qt, utend_dyn, vtend_dyn, load_column);
        park_uv[0][lane] = utend_dyn;
        park_uv[1][lane] = vtend_dyn;
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            park_t[k][lane] = tt[k];
            park_q[k][lane] = qt[k];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the physics' loads below: the two phases must not add their registers
    }
    // SPPT (physics.f90:234-248, csrc/sppt.hip): tend = (1 + r mu(k)) (tend - tend_dyn) + tend_dyn with mu = 1, r clipped to
    // [-1, 1] (sppt.f90:112).  `finish` turns what the physics accumulated for one (variable, level) into the value that is
    // stored: acc is the full tendency (dynamics + physics) for R = double and the physics increment alone when MIXED.
    const bool sppt = a.sppt_pattern != nullptr;
    const bool need_dyn = MIXED || sppt;  // the dynamics-only tendency is needed again when the result is stored
    auto finish = [&](R acc, double dyn, int k) -> double {
        if (sppt) {
            const double r = a.sppt_pattern[o3 + NG * k];
            const double rc = (r < -1.0) ? -1.0 : (r > 1.0 ? 1.0 : r);
            return MIXED ? dyn + (1.0 + rc) * static_cast<double>(acc) : (1.0 + rc) * (static_cast<double>(acc) - dyn) + dyn;
        }
        return MIXED ? dyn + static_cast<double>(acc) : static_cast<double>(acc);
    };

    // ------------------------------------------------------------------ thermodynamics, physics.f90:107-116
    R ta[KX], qa[KX], phi[KX], se[KX], qsat[KX], rh[KX];
    if (!FUSED || MIXED) load_column();  // all 25 loads first: one memory round trip, not one per group of levels
    const R psa = rexp(R(in_ps));
    const R rps = 1.0f / psa;
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        ta[k] = R(in_t[k]);
        phi[k] = R(in_phi[k]);
        qa[k] = rmax<R>(R(in_q[k]), 0.0f);
        se[k] = C::CP * ta[k] + phi[k];
        qsat[k] = qsat_point<R>(ta[k], vert<R>(vc::fsg, k) * psa);
        rh[k] = qa[k] / qsat[k];
    }
    // ------------------------------------------------------------------ deep convection, convection.f90
    const R psmin = 0.8f, trcnv = 6.0f, rhbl = 0.9f, rhil = 0.7f, entmax = 0.5f, smf = 0.8f;
    int itop = KX + 1;
    R qdif = R(0.0f), cbmf = R(0.0f), precnv = R(0.0f);
    R dfse[KX], dfqa[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) dfse[k] = dfqa[k] = R(0.0f);
    if (psa > psmin) {  // diagnose_convection, convection.f90:170-253
        const R mse0 = se[KX - 1] + C::ALHC * qa[KX - 1];
        R mse1 = se[nl1 - 1] + C::ALHC * qa[nl1 - 1];
        mse1 = rmin<R>(mse0, mse1);
        const R mss_kx = se[KX - 1] + C::ALHC * qsat[KX - 1];
        const R mss0 = rmax<R>(mse0, mss_kx);
        int ktop1 = KX, ktop2 = KX;
        R msthr = R(0.0f);
#pragma unroll
        for (int k = KX - 3; k >= 3; --k) {  // 1-based k
            const R mss_k = se[k - 1] + C::ALHC * qsat[k - 1], mss_k1 = se[k] + C::ALHC * qsat[k];
            const R mss2 = mss_k + vert<R>(vc::wvi, 8 + k - 1) * (mss_k1 - mss_k);
            if (mss0 > mss2) ktop1 = k;
            if (mse1 > mss2) {
                ktop2 = k;
                msthr = mss2;
            }
        }
        if (ktop1 < KX) {
            const R qthr0 = rhbl * qsat[KX - 1], qthr1 = rhbl * qsat[nl1 - 1];
            const bool lqthr = (qa[KX - 1] > qthr0) && (qa[nl1 - 1] > qthr1);
            if (ktop2 < KX) {
                itop = ktop1;
                qdif = rmax<R>(qa[KX - 1] - qthr0, (mse0 - msthr) * (R(1.0f) / C::ALHC));
            } else if (lqthr) {
                itop = ktop1;
                qdif = qa[KX - 1] - qthr0;
            }
        }
    }
    if (itop != KX + 1) {  // convection.f90:78-156
        const R fqmax = 5.0f;
        const R fm0 = C::P0 * vert<R>(vc::dhs, KX - 1) / (C::GRAV * trcnv * 3600.0f);
        const R rdps = 2.0f / (1.0f - psmin);
        R entr[KX];  // entr(2:kx-1), 0-based index k-1
        R sentr = R(0.0f);
#pragma unroll
        for (int k = 2; k <= nl1; ++k) {
            const R d = rmax<R>(0.0f, vert<R>(vc::fsg, k - 1) - 0.5f);
            entr[k - 1] = d * d;
            sentr = sentr + entr[k - 1];
        }
        sentr = entmax / sentr;
#pragma unroll
        for (int k = 2; k <= nl1; ++k) entr[k - 1] = entr[k - 1] * sentr;

        const R wv_nl1 = vert<R>(vc::wvi, 8 + nl1 - 1);
        const R qmax = rmax<R>(1.01f * qa[KX - 1], qsat[KX - 1]);
        R sb = se[nl1 - 1] + wv_nl1 * (se[KX - 1] - se[nl1 - 1]);
        R qb = qa[nl1 - 1] + wv_nl1 * (qa[KX - 1] - qa[nl1 - 1]);
        qb = rmin<R>(qb, qa[KX - 1]);
        const R fpsa = psa * rmin<R>(1.0f, (psa - psmin) * rdps);
        R fmass = fm0 * fpsa * rmin<R>(fqmax, qdif / (qmax - qb));
        cbmf = fmass;
        R fus = fmass * se[KX - 1], fuq = fmass * qmax;
        R fds = fmass * sb, fdq = fmass * qb;
        dfse[KX - 1] = fds - fus;
        dfqa[KX - 1] = fdq - fuq;
#pragma unroll
        for (int k = KX - 1; k >= 2; --k) {  // 1-based k, active while k >= itop+1
            if (k >= itop + 1) {
                dfse[k - 1] = fus - fds;
                dfqa[k - 1] = fuq - fdq;
                const R enmass = entr[k - 1] * psa * cbmf;
                fmass = fmass + enmass;
                fus = fus + enmass * se[k - 1];
                fuq = fuq + enmass * qa[k - 1];
                const R wv = vert<R>(vc::wvi, 8 + k - 2);
                sb = se[k - 2] + wv * (se[k - 1] - se[k - 2]);
                qb = qa[k - 2] + wv * (qa[k - 1] - qa[k - 2]);
                fds = fmass * sb;
                fdq = fmass * qb;
                dfse[k - 1] = dfse[k - 1] + fds - fus;
                dfqa[k - 1] = dfqa[k - 1] + fdq - fuq;
                const R delq = rhil * qsat[k - 1] - qa[k - 1];
                if (delq > R(0.0f)) {
                    const R fsq = smf * cbmf * delq;
                    dfqa[k - 1] = dfqa[k - 1] + fsq;
                    dfqa[KX - 1] = dfqa[KX - 1] - fsq;
                }
            }
        }
        // top layer k = itop (3 <= itop <= kx-3): static indexing through a select chain keeps arrays in registers
        R qs_t = R(0.0f), qs_t1 = R(0.0f), wv_t = R(0.0f);
#pragma unroll
        for (int k = 3; k <= KX - 3; ++k)
            if (k == itop) {
                qs_t = qsat[k - 1];
                qs_t1 = qsat[k];
                wv_t = vert<R>(vc::wvi, 8 + k - 1);
            }
        const R qsatb = qs_t + wv_t * (qs_t1 - qs_t);
        precnv = rmax<R>(fuq - fmass * qsatb, R(0.0f));
#pragma unroll
        for (int k = 3; k <= KX - 3; ++k)
            if (k == itop) {
                dfse[k - 1] = fus - fds + C::ALHC * precnv;
                dfqa[k - 1] = fuq - fdq - precnv;
            }
    }
    if (diag) {
        st_stream(a.cbmf, o2, cbmf);
        st_stream(a.precnv, o2, precnv);
    }
    const int icnv = KX - itop;  // physics.f90:132
    int iptop = itop;

    // ------------------------------------------------------------------ large-scale condensation
    R ttend[KX], qtend[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) {  // R = double: the physics adds to the dynamics' tendencies; MIXED: it starts from zero
        ttend[k] = MIXED ? R(0.0f) : static_cast<R>(FUSED ? park_t[k][lane] : R(a.ttend[o3 + NG * k]));
        qtend[k] = MIXED ? R(0.0f) : static_cast<R>(FUSED ? park_q[k][lane] : R(a.qtend[o3 + NG * k]));
    }
    R precls = R(0.0f);
    {
        const R trlsc = 4.0f, rhlsc = 0.9f, drhlsc = 0.1f, rhblsc = 0.95f, qsmax = 10.0f;
        const R rtlsc = 1.0f / (trlsc * 3600.0f), tfact = C::ALHC / C::CP, prg = C::P0 / C::GRAV;
        const R psa2 = psa * psa;
#pragma unroll
        for (int k = 2; k <= KX; ++k) {
            const R sig2 = vert<R>(vc::fsg, k - 1) * vert<R>(vc::fsg, k - 1);
            R rhref = rhlsc + drhlsc * (sig2 - 1.0f);
            if (k == KX) rhref = rmax<R>(rhref, rhblsc);
            const R dqmax = qsmax * sig2 * rtlsc;
            const R dqa = rhref * qsat[k - 1] - qa[k - 1];
            R dq = R(0.0f), dt = R(0.0f);
            if (dqa < R(0.0f)) {
                iptop = k < iptop ? k : iptop;
                dq = dqa * rtlsc;
                dt = tfact * rmin<R>(-dq, dqmax * psa2);
            }
            // physics.f90:127-130, 138-139: ttend = ttend + tt_cnv + tt_lsc
            ttend[k - 1] = ttend[k - 1] + dfse[k - 1] * rps * vert<R>(vc::grdscp, k - 1) + dt;
            qtend[k - 1] = qtend[k - 1] + dfqa[k - 1] * rps * vert<R>(vc::grdsig, k - 1) + dq;
            precls = precls - (vert<R>(vc::dhs, k - 1) * prg) * dq;
        }
        // level 1: tt_cnv(1) = dfse(1) (unscaled, zero), tt_lsc(1) = 0
        ttend[0] = ttend[0] + dfse[0] + R(0.0f);
        qtend[0] = qtend[0] + dfqa[0] + R(0.0f);
        precls = precls * psa;
    }
    if (diag) st_stream(a.precls, o2, precls);

    const R gse = (se[nl1 - 1] - se[KX - 1]) / (phi[nl1 - 1] - phi[KX - 1]);  // physics.f90:152 (used on shortwave steps)
    const R phi_kx = phi[KX - 1];
    // ------------------------------------------------------------------ vertical diffusion (vertical_diffusion.f90)
    // Evaluated here, right after the moist schemes, although the reference calls it after the radiation: it only
    // depends on se, rh, qa, qsat, phi and icnv, and computing it now lets those 40 per-column values die before the
    // register-hungry radiation sweeps.  Its tendencies are ADDED in the reference's order at the end.
    const R trshc = 6.0f, trvdi = 24.0f, trvds = 6.0f, redshc = 0.5f, rhgrad = 0.5f, segrad = 0.1f;
    const R cshc = vert<R>(vc::dhs, KX - 1) / 3600.0f;
    const R cvdi = (vert<R>(vc::sigh, nl1) - vert<R>(vc::sigh, 1)) / (static_cast<float>(nl1 - 1) * 3600.0f);
    const R fshcq = cshc / trshc, fshcse = cshc / (trshc * C::CP);
    const R fvdiq = cvdi / trvdi, fvdise = cvdi / (trvds * C::CP);
    R ttv[KX], qtv[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) ttv[k] = qtv[k] = R(0.0f);
    {
        const R rs_nl1 = 1.0f / vert<R>(vc::dhs, nl1 - 1), rs_kx = 1.0f / vert<R>(vc::dhs, KX - 1);
        const R drh0 = rhgrad * (vert<R>(vc::fsg, KX - 1) - vert<R>(vc::fsg, nl1 - 1));
        const R fvdiq2 = fvdiq * vert<R>(vc::sigh, nl1);
        const R dmse = se[KX - 1] - se[nl1 - 1] + C::ALHC * (qa[KX - 1] - qsat[nl1 - 1]);
        const R drh = rh[KX - 1] - rh[nl1 - 1];
        if (dmse >= R(0.0f)) {
            const R fcnv = (icnv > 0) ? redshc : static_cast<R>(1.0f);
            const R fluxse = fcnv * fshcse * dmse;
            ttv[nl1 - 1] = fluxse * rs_nl1;
            ttv[KX - 1] = -fluxse * rs_kx;
            if (drh >= R(0.0f)) {
                const R fluxq = fcnv * fshcq * qsat[KX - 1] * drh;
                qtv[nl1 - 1] = fluxq * rs_nl1;
                qtv[KX - 1] = -fluxq * rs_kx;
            }
        } else if (drh > drh0) {
            const R fluxq = fvdiq2 * qsat[nl1 - 1] * drh;
            qtv[nl1 - 1] = fluxq * rs_nl1;
            qtv[KX - 1] = -fluxq * rs_kx;
        }
    }
#pragma unroll
    for (int k = 3; k <= KX - 2; ++k) {
        if (vert<R>(vc::sigh, k) > 0.5f) {
            const R drh0 = rhgrad * (vert<R>(vc::fsg, k) - vert<R>(vc::fsg, k - 1));
            const R fvdiq2 = fvdiq * vert<R>(vc::sigh, k);
            const R drh = rh[k] - rh[k - 1];
            if (drh >= drh0) {
                const R fluxq = fvdiq2 * qsat[k - 1] * drh;
                qtv[k - 1] = qtv[k - 1] + fluxq * (1.0f / vert<R>(vc::dhs, k - 1));
                qtv[k] = qtv[k] - fluxq * (1.0f / vert<R>(vc::dhs, k));
            }
        }
    }
#pragma unroll
    for (int k = 1; k <= nl1; ++k) {
        const R se0 = se[k] + segrad * (phi[k - 1] - phi[k]);
        if (se[k - 1] < se0) {
            const R fluxse = fvdise * (se0 - se[k - 1]);
            ttv[k - 1] = ttv[k - 1] + fluxse * (1.0f / vert<R>(vc::dhs, k - 1));
            const R r1 = 1.0f / (1.0f - vert<R>(vc::sigh, k));
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
    // (when the dynamics-only tendency is needed again -- SPPT, mixed precision -- it is still in LDS (FUSED) or in memory:
    // this kernel writes each output exactly once, here or at its end)
    double qdyn_kx = 0.0;
    if (need_dyn) qdyn_kx = FUSED ? park_q[KX - 1][lane] : a.qtend[o3 + static_cast<size_t>(NG) * (KX - 1)];
#pragma unroll
    for (int k = 0; k < KX - 1; ++k) {
        double qdyn = 0.0;
        if (need_dyn) qdyn = FUSED ? park_q[k][lane] : a.qtend[o3 + NG * k];
        a.qtend[o3 + NG * k] = finish(qtend[k], qdyn, k);
    }
    const R qtend_kx = qtend[KX - 1];

    // ------------------------------------------------------------------ clouds + shortwave (every nstrad-th step)
    // rad_tau2(k, band) lives in HBM between shortwave steps (physics.f90 keeps it in the model state); while the two
    // longwave sweeps run it sits in LDS (tau_s), read one band at a time instead of holding all 32 values in registers.
    R ssrd, strat1, strat2;
    int icltop = 0;
    R cloudc = R(0.0f), clstr = R(0.0f);
    if (a.compute_shortwave) {
        const R rhcl1 = 0.30f, rhcl2 = 1.00f, qacl = 0.20f, wpcl = 0.2f, pmaxcl = 10.0f, clsmax = 0.60f,
                     clsminl = 0.15f, gse_s0 = 0.25f, gse_s1 = 0.40f, albcl = 0.43f, albcls = 0.50f, absdry = 0.033f,
                     absaer = 0.033f, abswv1 = 0.022f, abswv2 = 15.000f, abscl1 = 0.015f, abscl2 = 0.15f,
                     ablwin = 0.3f, ablwv1 = 0.7f, ablwv2 = 50.0f, ablcl1 = 12.0f, ablcl2 = 0.6f;
        const R fmask = R(a.fmask_land[o2]);
        // (the other inputs of the block, requested in the same batch)
        const R zenit = R(a.zenit_correction[o2]), solar = R(a.flux_solar_in[o2]), ozupp = R(a.flux_ozone_upper[o2]);
        const R ozlow = R(a.flux_ozone_lower[o2]), alb_sfc = R(a.alb_surface[o2]);
        // clouds, shortwave_radiation.f90:325-404
        const R rrcl = 1.f / (rhcl2 - rhcl1);
        if (rh[nl1 - 1] > rhcl1) {
            cloudc = rh[nl1 - 1] - rhcl1;
            icltop = nl1;
        } else {
            cloudc = R(0.0f);
            icltop = KX + 1;
        }
#pragma unroll
        for (int k = 3; k <= KX - 2; ++k) {
            const R drh = rh[k - 1] - rhcl1;
            if (drh > cloudc && qa[k - 1] > qacl) {
                cloudc = drh;
                icltop = k;
            }
        }
        const R pr1 = rmin<R>(pmaxcl, 86.4f * (precnv + precls));
        const R cq = rmin<R>(1.0f, cloudc * rrcl);
        cloudc = rmin<R>(1.0f, wpcl * rsqrt_(pr1) + cq * cq);
        icltop = iptop < icltop ? iptop : icltop;
        const R qcloud = qa[nl1 - 1];
        stream_store(&a.qcloud_equiv[o2], qcloud);  // (shortwave-step outputs persist over the next two steps: always stored)
        const R clfact = 1.2f, rgse = 1.0f / (gse_s1 - gse_s0);
        const R fstab = rmax<R>(0.0f, rmin<R>(1.0f, rgse * (gse - gse_s0)));
        clstr = fstab * rmax<R>(clsmax - clfact * cloudc, 0.0f);
        const R clstrl = rmax<R>(clstr, clsminl) * rh[KX - 1];
        clstr = clstr + fmask * (clstrl - clstr);

        // shortwave, shortwave_radiation.f90:50-214
        const R fband2 = 0.05f, fband1 = 1.0f - fband2;
        R refl[KX];  // rad_tau2(:, 3) during the shortwave sweep: cloud reflectivities, then reflected fluxes
#pragma unroll
        for (int k = 0; k < KX; ++k) refl[k] = R(0.0f);
#pragma unroll
        for (int k = 1; k <= KX; ++k)
            if (k == icltop) refl[k - 1] = albcl * cloudc;
        refl[KX - 1] = albcls * clstr;
        const R psaz = psa * zenit;
        R acloud = cloudc * rmin<R>(abscl1 * qcloud, abscl2);
        R tsw1[KX], tsw2[KX];  // shortwave transmissivities, bands 1 and 2
        tsw1[0] = rexp(-psaz * vert<R>(vc::dhs, 0) * absdry);
        tsw2[0] = R(0.0f);
#pragma unroll
        for (int k = 2; k <= nl1; ++k) {
            const R abs1 = absdry + absaer * (vert<R>(vc::fsg, k - 1) * vert<R>(vc::fsg, k - 1));
            tsw1[k - 1] = (k >= icltop) ? rexp(-psaz * vert<R>(vc::dhs, k - 1) * (abs1 + abswv1 * qa[k - 1] + acloud))
                                        : rexp(-psaz * vert<R>(vc::dhs, k - 1) * (abs1 + abswv1 * qa[k - 1]));
        }
        {
            const R abs1 = absdry + absaer * (vert<R>(vc::fsg, KX - 1) * vert<R>(vc::fsg, KX - 1));
            tsw1[KX - 1] = rexp(-psaz * vert<R>(vc::dhs, KX - 1) * (abs1 + abswv1 * qa[KX - 1]));
        }
#pragma unroll
        for (int k = 2; k <= KX; ++k) tsw2[k - 1] = rexp(-psaz * vert<R>(vc::dhs, k - 1) * abswv2 * qa[k - 1]);

        R tsr = solar;
        R tt_rsw[KX];
        R f1 = solar * fband1, f2 = solar * fband2;
        tt_rsw[0] = f1;
        f1 = tsw1[0] * (f1 - ozupp * psa);
        tt_rsw[0] = tt_rsw[0] - f1;
        tt_rsw[1] = f1;
        f1 = tsw1[1] * (f1 - ozlow * psa);
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
        f1 = f1 * alb_sfc;
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
            tt_rsw[k] = tt_rsw[k] * rps * vert<R>(vc::grdscp, k);
            st_stream(a.tt_rsw, o3 + NG * k, tt_rsw[k]);
            ttend[k] = ttend[k] + tt_rsw[k];
        }

        // longwave transmissivities, shortwave_radiation.f90:170-208
        const R co2 = a.air_absortivity_co2;
        const size_t NGs = static_cast<size_t>(NG);
        tau_s[0][lane] = rexp(-psa * vert<R>(vc::dhs, 0) * ablwin);
        tau_s[KX][lane] = rexp(-psa * vert<R>(vc::dhs, 0) * co2);
        st_stream(a.rad_tau2, ot + NGs * (0 + KX * 0), tau_s[0][lane]);
        st_stream(a.rad_tau2, ot + NGs * (0 + KX * 1), tau_s[KX][lane]);
        st_stream(a.rad_tau2, ot + NGs * (0 + KX * 2), R(1.0f));
        st_stream(a.rad_tau2, ot + NGs * (0 + KX * 3), R(1.0f));
        acloud = cloudc * ablcl2;
#pragma unroll
        for (int k = 2; k <= KX; ++k) {
            R t0, t1, t2, t3;
            if (k == 2 || k == KX) {
                t0 = rexp(-psa * vert<R>(vc::dhs, k - 1) * ablwin);
                t1 = rexp(-psa * vert<R>(vc::dhs, k - 1) * co2);
                t2 = rexp(-psa * vert<R>(vc::dhs, k - 1) * ablwv1 * qa[k - 1]);
                t3 = rexp(-psa * vert<R>(vc::dhs, k - 1) * ablwv2 * qa[k - 1]);
            } else {
                const R deltap = psa * vert<R>(vc::dhs, k - 1);
                const R acloud1 = (k < icltop) ? acloud : ablcl1 * cloudc;
                t0 = rexp(-deltap * (ablwin + acloud1));
                t1 = rexp(-deltap * co2);
                t2 = rexp(-deltap * rmax<R>(ablwv1 * qa[k - 1], acloud));
                t3 = rexp(-deltap * rmax<R>(ablwv2 * qa[k - 1], acloud));
            }
            st_stream(a.rad_tau2, ot + NGs * (k - 1 + KX * 0), t0);
            st_stream(a.rad_tau2, ot + NGs * (k - 1 + KX * 1), t1);
            st_stream(a.rad_tau2, ot + NGs * (k - 1 + KX * 2), t2);
            st_stream(a.rad_tau2, ot + NGs * (k - 1 + KX * 3), t3);
            tau_s[k - 1 + KX * 0][lane] = t0;
            tau_s[k - 1 + KX * 1][lane] = t1;
            tau_s[k - 1 + KX * 2][lane] = t2;
            tau_s[k - 1 + KX * 3][lane] = t3;
        }
        const R eps1 = C::EPSLW / (vert<R>(vc::dhs, 0) + vert<R>(vc::dhs, 1));
        strat1 = R(a.stratospheric_correction[o2]) * psa;  // (requested here: one more live value above costs the kernel scratch)
        strat2 = eps1 * psa;
        st_stream(a.rad_strat_corr, oc, strat1);
        st_stream(a.rad_strat_corr, oc + NG, strat2);
    } else {
#pragma unroll
        for (int k = 0; k < KX; ++k) ttend[k] = ttend[k] + R(ld_plain(a.tt_rsw, o3 + NG * k));
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = (b < 2 ? 0 : 1); k < KX; ++k)  // bands 3-4 do not use the top level
                tau_s[k + KX * b][lane] = R(ld_plain(a.rad_tau2, ot + static_cast<size_t>(NG) * (k + KX * b)));
        ssrd = R(a.ssrd[o2]);
        strat1 = R(ld_plain(a.rad_strat_corr, oc));
        strat2 = R(ld_plain(a.rad_strat_corr, oc + NG));
    }

    // ------------------------------------------------------------------ longwave, downward sweep
    R st4a[KX][2], dfabs[KX], flux[4];
    {
        const R anis = 1.0f;
#pragma unroll
        for (int k = 1; k <= nl1; ++k) st4a[k - 1][0] = ta[k - 1] + vert<R>(vc::wvi, 8 + k - 1) * (ta[k] - ta[k - 1]);
        st4a[0][1] = 0.75f * ta[0] + 0.25f * st4a[0][0];
        st4a[1][1] = 0.50f * ta[1] + 0.25f * (st4a[0][0] + st4a[1][0]);
#pragma unroll
        for (int k = 3; k <= nl1; ++k) st4a[k - 1][1] = 0.5f * anis * rmax<R>(st4a[k - 1][0] - st4a[k - 2][0], 0.0f);
        st4a[KX - 1][1] = anis * rmax<R>(ta[KX - 1] - st4a[nl1 - 1][0], 0.0f);
#pragma unroll
        for (int k = 1; k <= 2; ++k) {
            st4a[k - 1][0] = C::SBC * pow4<R>(st4a[k - 1][1]);
            st4a[k - 1][1] = R(0.0f);
        }
#pragma unroll
        for (int k = 3; k <= KX; ++k) {
            const R st3a = C::SBC * pow3<R>(ta[k - 1]);
            st4a[k - 1][0] = st3a * ta[k - 1];
            st4a[k - 1][1] = 4.0f * st3a * st4a[k - 1][1];
        }
#pragma unroll
        for (int k = 0; k < KX; ++k) dfabs[k] = R(0.0f);
        int itab[KX];  // nint(ta(k)) - 100, clamped: row of fband for level k (both sweeps)
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            int it = rnint(ta[k]);
            itab[k] = (it < 100 ? 100 : (it > 400 ? 400 : it)) - 100;
        }
        // Stratosphere (bands 1-2, :73-79) first for both bands, as the reference does: dfabs(1) sums in that order.
        R tau0[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            tau0[b] = tau_s[KX * b][lane];
            const R emis = 1.0f - tau0[b];
            const R brad = CT.fband[itab[0] + 301 * b] * (st4a[0][0] + emis * st4a[0][1]);
            flux[b] = emis * brad;
            dfabs[0] = dfabs[0] - flux[b];
        }
        flux[2] = flux[3] = R(0.0f);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            R tb[KX];
#pragma unroll
            for (int k = 2; k <= KX; ++k) tb[k - 1] = tau_s[k - 1 + KX * b][lane];
#pragma unroll
            for (int k = 2; k <= KX; ++k) {
                const R emis = 1.0f - tb[k - 1];
                const R brad = CT.fband[itab[k - 1] + 301 * b] * (st4a[k - 1][0] + emis * st4a[k - 1][1]);
                dfabs[k - 1] = dfabs[k - 1] + flux[b];
                flux[b] = tb[k - 1] * flux[b] + emis * brad;
                dfabs[k - 1] = dfabs[k - 1] - flux[b];
            }
        }
        R slrd = R(0.0f);
#pragma unroll
        for (int b = 0; b < 4; ++b) slrd = slrd + C::EMISFC * flux[b];
        const R corlw = C::EPSLW * C::EMISFC * st4a[KX - 1][0];
        dfabs[KX - 1] = dfabs[KX - 1] - corlw;
        slrd = slrd + corlw;
        if (diag) {
            st_stream(a.slrd, o2, slrd);
#pragma unroll
            for (int k = 0; k < KX; ++k) {
                st_stream(a.rad_st4a, os + static_cast<size_t>(NG) * k, st4a[k][0]);
                st_stream(a.rad_st4a, os + static_cast<size_t>(NG) * (k + KX), st4a[k][1]);
            }
        }

        // -------------------------------------------------------------- surface fluxes, surface_fluxes.f90:40-320
        const R fwind0 = 0.95f, ftemp0 = 1.0f, cdl = 2.4e-3f, cds = 1.0e-3f, chl = 1.2e-3f, chs = 0.9e-3f,
                     vgust = 5.0f, ctday = 1.0e-2f, dtheta = 3.0f, fstab = 0.67f, clambda = 7.0f, clambsn = 7.0f;
        const R esbc = C::EMISFC * C::SBC;
        const R ua = R(ld_plain(a.ug, o3 + NG * (KX - 1))), va = R(ld_plain(a.vg, o3 + NG * (KX - 1)));
        const R fmask = R(a.fmask_land[o2]), phi0 = R(a.phis0[o2]), tsea = R(a.sst_am[o2]), land_temp = R(a.land_temp[o2]);
        const R alb_land = R(a.alb_land[o2]), swav = R(a.soil_avail_water[o2]);
        // (every input of the block is requested here, in one batch: a load issued where its value is first needed costs the
        // wavefront one more memory round trip on its dependent chain)
        const R forog = R(a.forog[o2]), snowc = R(a.snowc[o2]), alb_sea = R(a.alb_sea[o2]);
        const R u0 = fwind0 * ua, v0 = fwind0 * va;
        const R gtemp0 = 1.0f - ftemp0, rcp = 1.0f / C::CP;
        const R dt1 = vert<R>(vc::wvi, 8 + KX - 1) * (ta[KX - 1] - ta[nl1 - 1]);
        R t1l = ta[KX - 1] + dt1;
        R t1s = t1l - phi0 * dt1 / (C::RGAS * 288.0f * vert<R>(vc::sigl, KX - 1));
        const R t2s = ta[KX - 1] + rcp * phi_kx;
        const R t2l = t2s - rcp * phi0;
        if (ta[KX - 1] > ta[nl1 - 1]) {
            t1l = ftemp0 * t1l + gtemp0 * t2l;
            t1s = ftemp0 * t1s + gtemp0 * t2s;
        } else {
            t1l = ta[KX - 1];
            t1s = ta[KX - 1];
        }
        const R t0 = t1s + fmask * (t1l - t1s);
        const R den0 = (C::P0 * psa / (C::RGAS * t0)) * rsqrt_(u0 * u0 + v0 * v0 + vgust * vgust);
        R tskin = land_temp + ctday * rsqrt_(CT.coa[j]) * ssrd * (1.0f - alb_land) * psa;
        const R rdth = fstab / dtheta, astab = 0.5f;
        const R dthl = (tskin > t2l) ? rmin<R>(dtheta, tskin - t2l) : rmax<R>(-dtheta, astab * (tskin - t2l));
        const R den1 = den0 * (1.0f + dthl * rdth);
        const R cdldv = cdl * den0 * forog;
        const R ustr1 = -cdldv * ua, vstr1 = -cdldv * va;
        const R chlcp = chl * C::CP;
        R shf1 = chlcp * den1 * (tskin - t1l);
        const R q1 = qa[KX - 1];
        const R qs0l = qsat_point<R>(tskin, R(1.0f) * psa);
        R evap1 = chl * den1 * rmax<R>(0.0f, swav * qs0l - q1);
        const R tsk3 = pow3<R>(tskin);
        const R dslr = 4.0f * esbc * tsk3;
        R slru1 = esbc * tsk3 * tskin;
        R hfl1 = ssrd * (1.0f - alb_land) + slrd - (slru1 + shf1 + C::ALHC * evap1);
        const R clamb = clambda + snowc * (clambsn - clambda);
        hfl1 = hfl1 - clamb * (tskin - land_temp);
        R dqs = qsat_point<R>(tskin + 1.0f, R(1.0f) * psa);
        dqs = (evap1 > R(0.0f)) ? swav * (dqs - qs0l) : R(0.0f);
        const R dtskin = hfl1 / (clamb + dslr + chl * den1 * (C::CP + C::ALHC * dqs));
        tskin = tskin + dtskin;
        shf1 = shf1 + chlcp * den1 * dtskin;
        evap1 = evap1 + chl * den1 * dqs * dtskin;
        slru1 = slru1 + dslr * dtskin;
        hfl1 = clamb * (tskin - land_temp);
        const R dths = (tsea > t2s) ? rmin<R>(dtheta, tsea - t2s) : rmax<R>(-dtheta, astab * (tsea - t2s));
        const R den2 = den0 * (1.0f + dths * rdth);
        const R cdsdv = cds * den2;
        const R ustr2 = -cdsdv * ua, vstr2 = -cdsdv * va;
        const R shf2 = chs * C::CP * den2 * (tsea - t1s);
        const R qs0s = qsat_point<R>(tsea, R(1.0f) * psa);
        const R evap2 = chs * den2 * (qs0s - q1);
        const R slru2 = esbc * pow4<R>(tsea);
        const R hfl2 = ssrd * (1.0f - alb_sea) + slrd - slru2 + shf2 + C::ALHC * evap2;
        const R ustr3 = ustr2 + fmask * (ustr1 - ustr2), vstr3 = vstr2 + fmask * (vstr1 - vstr2);
        const R shf3 = shf2 + fmask * (shf1 - shf2), evap3 = evap2 + fmask * (evap1 - evap2);
        const R slru3 = slru2 + fmask * (slru1 - slru2);
        const R tsfc = tsea + fmask * (land_temp - tsea);
        const R tskin_avg = tsea + fmask * (tskin - tsea);
        // consumed by the slab models of the coupler (land_model.f90:195-215, sea_model.f90:313-383): always stored
        a.shf[oa + NG] = shf2;
        a.evap[oa + NG] = evap2;
        stream_store(&a.hfluxn[oa], hfl1); a.hfluxn[oa + NG] = hfl2;
        if (diag) {
            st_stream(a.ustr, oa, ustr1); st_plain(a.ustr, oa + NG, ustr2); st_plain(a.ustr, oa + 2 * NG, ustr3);
            st_stream(a.vstr, oa, vstr1); st_plain(a.vstr, oa + NG, vstr2); st_plain(a.vstr, oa + 2 * NG, vstr3);
            stream_store(&a.shf[oa], shf1);   a.shf[oa + 2 * NG] = shf3;
            stream_store(&a.evap[oa], evap1); a.evap[oa + 2 * NG] = evap3;
            st_stream(a.slru, oa, slru1); st_plain(a.slru, oa + NG, slru2); st_plain(a.slru, oa + 2 * NG, slru3);
        }
        if (a.ts) stream_store(&a.ts[o2], tsfc);
        if (a.tskin) stream_store(&a.tskin[o2], tskin_avg);
        if (a.u0) stream_store(&a.u0[o2], u0);
        if (a.v0) stream_store(&a.v0[o2], v0);
        if (a.t0) stream_store(&a.t0[o2], t0);

        // -------------------------------------------------------------- longwave, upward sweep (:124-205)
        const R refsfc = 1.0f - C::EMISFC;
        if (diag) st_stream(a.slr, o2, slru3 - slrd);
        R fsfc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) fsfc[b] = fband_at(CT.fband, tsfc, b);
#pragma unroll
        for (int b = 0; b < 4; ++b) flux[b] = fsfc[b] * slru3 + refsfc * flux[b];
        dfabs[KX - 1] = dfabs[KX - 1] + C::EPSLW * slru3;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            R tb[KX];
#pragma unroll
            for (int k = 2; k <= KX; ++k) tb[k - 1] = tau_s[k - 1 + KX * b][lane];
            R fb[KX];  // the band's table values of all levels in one batch of loads, not one round trip per level
#pragma unroll
            for (int k = 2; k <= KX; ++k) fb[k - 1] = CT.fband[itab[k - 1] + 301 * b];
#pragma unroll
            for (int k = KX; k >= 2; --k) {
                const R emis = 1.0f - tb[k - 1];
                const R brad = fb[k - 1] * (st4a[k - 1][0] - emis * st4a[k - 1][1]);
                dfabs[k - 1] = dfabs[k - 1] + flux[b];
                flux[b] = tb[k - 1] * flux[b] + emis * brad;
                dfabs[k - 1] = dfabs[k - 1] - flux[b];
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const R emis = 1.0f - tau0[b];
            const R brad = CT.fband[itab[0] + 301 * b] * (st4a[0][0] - emis * st4a[0][1]);
            dfabs[0] = dfabs[0] + flux[b];
            flux[b] = tau0[b] * flux[b] + emis * brad;
            dfabs[0] = dfabs[0] - flux[b];
        }
        const R corlw1 = vert<R>(vc::dhs, 0) * strat2 * st4a[0][0] + strat1;
        const R corlw2 = vert<R>(vc::dhs, 1) * strat2 * st4a[1][0];
        dfabs[0] = dfabs[0] - corlw1;
        dfabs[1] = dfabs[1] - corlw2;
        R olr = corlw1 + corlw2;
#pragma unroll
        for (int b = 0; b < 4; ++b) olr = olr + flux[b];
        if (diag) {
            st_stream(a.olr, o2, olr);
#pragma unroll
            for (int b = 0; b < 4; ++b) st_plain(a.rad_flux, of4 + static_cast<size_t>(NG) * b, flux[b]);
        }
        // physics.f90:207-211: ttend = ttend + tt_rsw + tt_rlw
#pragma unroll
        for (int k = 0; k < KX; ++k) ttend[k] = ttend[k] + dfabs[k] * rps * vert<R>(vc::grdscp, k);

        // physics.f90:223-231: surface-flux tendencies at the lowest level, then accumulate
        const R ut_kx = R(0.0f) + ustr3 * rps * vert<R>(vc::grdsig, KX - 1);
        const R vt_kx = R(0.0f) + vstr3 * rps * vert<R>(vc::grdsig, KX - 1);
        ttend[KX - 1] = ttend[KX - 1] + shf3 * rps * vert<R>(vc::grdscp, KX - 1);
        const size_t okx = o3 + static_cast<size_t>(NG) * (KX - 1);
        const double ud = FUSED ? park_uv[0][lane] : a.utend[okx], vd = FUSED ? park_uv[1][lane] : a.vtend[okx];
        const R qkx = qtend_kx + evap3 * rps * vert<R>(vc::grdsig, KX - 1);
        // (above the lowest level the physics leaves the wind tendencies alone: ut_pbl, vt_pbl are zero there)
        stream_store(&a.utend[okx], finish(MIXED ? ut_kx : static_cast<R>(ud) + ut_kx, ud, KX - 1));
        stream_store(&a.vtend[okx], finish(MIXED ? vt_kx : static_cast<R>(vd) + vt_kx, vd, KX - 1));
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            double tdyn = 0.0;
            if (need_dyn) tdyn = FUSED ? park_t[k][lane] : a.ttend[o3 + NG * k];
            a.ttend[o3 + NG * k] = finish(ttend[k], tdyn, k);
        }
        stream_store(&a.qtend[okx], finish(qkx, qdyn_kx, KX - 1));
    }
    if (a.iptop) a.iptop[o2] = iptop;
    if (a.icltop) a.icltop[o2] = icltop;
    if (a.cloudc) stream_store(&a.cloudc[o2], cloudc);
    if (a.clstr) stream_store(&a.clstr[o2], clstr);
}

// Launch-bounds variant of the fp64 kernels: 2 waves per SIMD (256 VGPRs, no scratch in the fused kernel).  The 1-wave build
// (PYSPEEDY_AMD_PHYS_WAVES=1) exists for comparison: it was 2.8 % faster per step at 8 members before the kernel requested its
// loads in batches and is equal since (14 % slower at 16 members and above); with -ffp-contract=on (Makefile) the two builds
// give the same bits, which they did not while the back end was free to fuse across statements.
static int physics_waves(int) {
    static const int waves = [] {
        const char *e = getenv("PYSPEEDY_AMD_PHYS_WAVES");
        return e ? atoi(e) : 2;
    }();
    return waves;
}
// launch-bounds variant of the fp32 kernels (waves per SIMD the register allocator leaves room for: 2 or 3; a 4-wave build
// spilled 368 bytes per lane and was 40 % slower): profiles/r02_cfg5_fp32_vs_fp64.txt.  By the size of the launch since round 4
// (profiles/r04_cfg5_fp32_storage.txt): the 3-wave build (168 VGPRs, 100 bytes of scratch) wins when the launch brings more than
// two wavefronts to a SIMD -- 32 members in one launch: 48 against 59 us --, the 2-wave build (205 VGPRs, no scratch) when it
// does not -- the 10 / 11-member launches of the default grouped plan at 32 members: 0.144 against 0.150 ms per step.  Same
// bits either way (-ffp-contract=on; tests/test_variants_spawn.py).  PYSPEEDY_AMD_PHYS_WAVES32=2|3 forces one.
static int physics_waves32(int nmembers) {
    static const int forced = [] {
        const char *e = getenv("PYSPEEDY_AMD_PHYS_WAVES32");
        return e ? atoi(e) : 0;
    }();
    if (forced) return forced;
    return nmembers * (NG / 64) <= 2 * 1024 ? 2 : 3;  // wavefronts of the launch against two per SIMD of the GPU
}

template <int W, bool FUSED, bool KEEP, typename R, bool S32 = false>
static hipError_t launch_physics(const DeviceTables &T, const spd_physics_args &a, int first, int nmembers, const ModelPtrs &P,
                                 const DynDeviceTables &D, int diag, hipStream_t s) {
    const long total = static_cast<long>(nmembers) * NG;
    const unsigned blocks = static_cast<unsigned>((total + kPhysThreads - 1) / kPhysThreads);
    launch(physics_kernel<W, FUSED, KEEP, R, S32>, dim3(blocks), dim3(kPhysThreads), 0, s, a, col_tables<R>(T), first,
                       nmembers, P, D, diag);
    return hipGetLastError();
}

// the C ABI's spd_physics: tendencies of the dynamics are read from / written to a.utend ... a.qtend
hipError_t run_physics(const DeviceTables &T, const spd_physics_args &a, int nmembers, int fp32, hipStream_t s) {
    const ModelPtrs mp{};
    const DynDeviceTables md{};
    if (fp32) {
        switch (physics_waves32(nmembers)) {
            case 2: return launch_physics<2, false, false, float>(T, a, 0, nmembers, mp, md, 1, s);
            default: return launch_physics<3, false, false, float>(T, a, 0, nmembers, mp, md, 1, s);
        }
    }
    if (physics_waves(nmembers) == 1) return launch_physics<1, false, false, double>(T, a, 0, nmembers, mp, md, 1, s);
    return launch_physics<2, false, false, double>(T, a, 0, nmembers, mp, md, 1, s);
}

// grid-point dynamics + physics of every column in one launch (the model step); a.ttend / a.qtend / a.utend / a.vtend must be
// the dynamics' tendency arrays (P.ttend, P.trtend, P.utend, P.vtend)
//
// diag == 0: the outputs that no later kernel reads -- precipitation, cloud-base mass flux, the radiative fluxes and their
// band / level decompositions, wind stress, the land / average parts of the surface fluxes: 39 doubles per column -- are
// computed as always but NOT stored (what a shortwave step leaves for the following steps is always stored).  The model passes 0 for every step of a multi-step call except
// the last one: such a value would be overwritten by the next step before anything could read it (model.hip).
hipError_t run_dyn_physics(const ModelPtrs &P, const DynDeviceTables &D, const DeviceTables &T, const spd_physics_args &a,
                           int first, int nmembers, int fp32, int store32, int diag, hipStream_t s) {
    if (fp32 && store32) {  // cfg 5: fp32 arithmetic, physics-only arrays stored as fp32
        switch (physics_waves32(nmembers)) {
            case 2: return launch_physics<2, true, true, float, true>(T, a, first, nmembers, P, D, diag, s);
            default: return launch_physics<3, true, true, float, true>(T, a, first, nmembers, P, D, diag, s);
        }
    }
    if (fp32) {  // (model option physics_storage32 = 0: the same arithmetic over fp64 storage, for comparison)
        switch (physics_waves32(nmembers)) {
            case 2: return launch_physics<2, true, true, float>(T, a, first, nmembers, P, D, diag, s);
            default: return launch_physics<3, true, true, float>(T, a, first, nmembers, P, D, diag, s);
        }
    }
    if (a.sppt_pattern) return launch_physics<2, true, true, double>(T, a, first, nmembers, P, D, diag, s);
    if (physics_waves(nmembers) == 1) return launch_physics<1, true, false, double>(T, a, first, nmembers, P, D, diag, s);
    return launch_physics<2, true, false, double>(T, a, first, nmembers, P, D, diag, s);
}

}  // namespace spd
