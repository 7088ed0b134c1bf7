// Batched spectral transforms for gfx950: spec2grid (inverse Legendre + inverse zonal FFT) and grid2spec
// (forward FFT + direct Legendre), each as ONE fused kernel, so a field crosses HBM exactly once in each direction
// (15 872 B spectral + 36 864 B grid = 52 736 B algorithmic bytes per field).
//
// Reference behaviour reproduced: ModSpectral_spec2grid / grid2spec (speedy.f90/spectral.f90:251-273) =
// ModLegendre_legendre_inv / legendre (legendre.f90:130-221) composed with ModFourier_fourier_inv / fourier
// (fourier.f90:63-123).  The stage-only entry points run the same kernels with one stage disabled.
//
// Work decomposition: one workgroup = one field, 512 threads = 8 wavefronts, 40 064 B of LDS -> 4 workgroups =
//   32 waves per CU (the hardware maximum).  LDS plan: a compact Fourier-coefficient buffer C[48][63] (24 192 B,
//   only wavenumbers 0..30 are ever non-zero) followed by a spectral-field staging area S (15 872 B); the full-length
//   FFT row buffer R[48][97] (37 248 B) ALIASES both, which is legal because the FFT "group" stage moves every value
//   through registers across a workgroup barrier (C -> registers -> barrier -> R, or R -> registers -> barrier -> C).
//   The kernels are latency-bound (profiles/), so occupancy is what the layout is optimised for:
//   * FFT stages run IN PLACE on the row buffer: every (row, group|block) task loads its <= 16 inputs into
//     registers, the workgroup synchronises, then the outputs are written back (fft96.hpp).  Wave w handles
//     group/block w for all 48 rows, so the pass index and every twiddle are wave-uniform (scalar loads) and no
//     wave diverges; lanes sit on different rows and the odd row stride makes every ds_read/write_b64
//     bank-conflict free.
//   * Legendre: a lane owns one zonal wavenumber m (re and im together, 16-byte LDS reads) and two latitude pairs
//     (inverse) or two total wavenumbers of one parity (direct).  Inverse lanes are ordered m-major so that a
//     wavefront only loops over the total wavenumbers its smallest m needs (triangular truncation: 32 - m of them);
//     direct lanes come from a work list that holds only the coefficients inside the truncation (279 lanes).  The
//     associated-Legendre values stream from an L2-resident table laid out so that a wavefront reads 1 KiB contiguous
//     per step.
//   * Staging: every lane issues all its global loads before the first dependent LDS store (and, on the way out, all
//     scale-factor loads before the first global store).  Results leave with non-temporal stores: they are consumed by a
//     later kernel, and keeping them out of the L2 leaves it to the Legendre table and the shared inputs (measured: +15 %
//     on the spectral -> grid kernel).  In the model step the spectral -> grid kernel can apply the
//     spectral operator in front of the transform while it stages the coefficients (FieldDesc::mode: vort2vel, gradient),
//     so u, v and grad ln ps are never materialised as spectral fields.
#include <hip/hip_runtime.h>

#include "device_tables.hpp"
#include "diagnostics_block.hpp"
#include "launch_events.hpp"
#include "fft96.hpp"

namespace spd {

constexpr int kThreads = 512;
constexpr int kRowStride = 97;
constexpr int kRows = IL;                       // 48 rows per field
constexpr int kCStride = 63;                    // compact row: positions 0..60 + the parked Im(m=0) at 61
constexpr int kCBufDoubles = kRows * kCStride;  // 3024 doubles = 24 192 B
constexpr int kSpecPerLane = (NSPEC + kThreads - 1) / kThreads;      // 2 x 16 B per lane stage a spectral field
constexpr int kGridPerLane = (NGRID / 2 + kThreads - 1) / kThreads;  // 5 x 16 B per lane stage a grid field
constexpr int kPlanePerLane = (kRows * MX + kThreads - 1) / kThreads;  // 3 x 16 B per lane stage a Fourier plane (Legendre stage alone)
constexpr int kHalfPlanePerLane = (IY * MX + kThreads - 1) / kThreads; // 2 x (16 + 16) B per lane: a wavenumber of a latitude and its mirror
constexpr int kInvLanes = MX * 12;              // inverse Legendre tasks: (m, pair-of-latitude-pairs)
constexpr size_t kLdsBytes = (kCBufDoubles + 2 * NSPEC) * sizeof(double);  // 40 064
static_assert(kRows * kRowStride <= kCBufDoubles + 2 * NSPEC, "R must fit inside C + S");

enum class Stage { Fused, LegendreOnly, FourierOnly };

using d2 = double __attribute__((ext_vector_type(2)));
// field pointers handed to the kernels always point to global memory; saying so in the type gives global_load / global_store
// instead of flat accesses where the pointer comes out of a descriptor table
using gd2_in = const __attribute__((address_space(1))) d2 *;
using gd2_out = __attribute__((address_space(1))) d2 *;
typedef float f2 __attribute__((ext_vector_type(2)));
using gf2_out = __attribute__((address_space(1))) f2 *;

// position of coefficient (m, re|im) in an unpacked FFT row (fourier.f90:74-81): re(m) -> 2m-1, im(m) -> 2m,
// re(0) -> 0.  im(0) has no slot in the transform; it is parked at position 61 where a stage needs it.
__device__ inline int pos_re(int m) { return m == 0 ? 0 : 2 * m - 1; }
__device__ inline int pos_im(int m) { return m == 0 ? 61 : 2 * m; }

__device__ inline int wave_id() { return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6); }

// Phase timing for kernel tuning (compiled only with -DSPD_TRACE, see tools/trace_transforms.py): thread 0 of every
// workgroup adds the shader-clock time spent up to each phase boundary to a device-global table.
#ifdef SPD_TRACE
__device__ unsigned long long spd_trace_acc[2][8];
__device__ unsigned long long spd_trace_cnt[2];
#define TRACE_BEGIN() unsigned long long trace_t0 = (threadIdx.x == 0) ? wall_clock64() : 0ull
#define TRACE_MARK(dir, i) \
    do { if (threadIdx.x == 0) atomicAdd(&spd_trace_acc[dir][i], wall_clock64() - trace_t0); } while (0)
#define TRACE_END(dir) \
    do { if (threadIdx.x == 0) atomicAdd(&spd_trace_cnt[dir], 1ull); } while (0)
#else
#define TRACE_BEGIN()
#define TRACE_MARK(dir, i)
#define TRACE_END(dir)
#endif

// Coefficient k = m + 31 n of the spectral field a descriptor-table entry asks for (device_tables.hpp: FieldDesc::mode),
// computed from the model's prognostic fields with the arithmetic of specops.hip / spectral.f90:190-214, 275-296.
__device__ inline d2 times_i(d2 z) { return d2{-z.y, z.x}; }

// Branch-free: the reference treats the first and the last total wavenumber separately (no n-1 / n+1 neighbour, and no
// x-derivative term at the last one); here the neighbour index is clamped and the missing terms vanish because the
// reference's own coefficient tables are zero there (uvdym(:,1) = gradym(:,1) = 0, uvdyp(:,nx) = gradyp(:,nx) = 0) or are
// selected away -- every lane issues the same loads at once instead of running three divergent paths one after the other.
__device__ __forceinline__ d2 staged_coefficient(int mode, gd2_in a, gd2_in b, int k, const DeviceTables &T) {
    const int n = k / MX, m = k - n * MX;
    const int km = n == 0 ? k : k - MX, kp = n == NX - 1 ? k : k + MX;
    const bool last = n == NX - 1;
    if (mode <= 2) {  // vort2vel: a = vorticity, b = divergence
        const double dx = last ? 0.0 : T.uvdx[k], cm = T.uvdym[k], cp = T.uvdyp[k];
        if (mode == 1) {  // ucos = uvdym vor(n-1) - uvdyp vor(n+1) + i uvdx div(n)
            const d2 zb = b[k], ap = a[km], an = a[kp];
            const d2 zc = times_i(d2{dx * zb.x, dx * zb.y});
            return d2{cm * ap.x - cp * an.x + zc.x, cm * ap.y - cp * an.y + zc.y};
        }
        const d2 za = a[k], bp = b[km], bn = b[kp];  // vcos = -uvdym div(n-1) + uvdyp div(n+1) + i uvdx vor(n)
        const d2 zp = times_i(d2{dx * za.x, dx * za.y});
        return d2{-cm * bp.x + cp * bn.x + zp.x, -cm * bp.y + cp * bn.y + zp.y};
    }
    if (mode == 3) {  // d/dx
        const d2 z = a[k];
        const double g = T.gradx[m];
        return times_i(d2{g * z.x, g * z.y});
    }
    const d2 zp = a[km], zn = a[kp];  // d/dy
    const double gm = T.gradym[k], gp = T.gradyp[k];
    return d2{-gm * zp.x + gp * zn.x, -gm * zp.y + gp * zn.y};
}

// ------------------------------------------------------------------------------------------------
// spec -> grid
// ------------------------------------------------------------------------------------------------
// src / dst point at ONE field (spectral field or Fourier plane in, Fourier plane or grid field out)
template <Stage ST>
__device__ __forceinline__ void spec2grid_body(const double *__restrict__ src, double *__restrict__ dst,
                                               const DeviceTables &T, int kcos, int mode = 0,
                                               const double *__restrict__ src2 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *cbuf = lds;                                        // C[48][63]
    double *rows = lds;                                        // R[48][97], aliases C and S (see header)
    d2 *s = reinterpret_cast<d2 *>(lds + kCBufDoubles);        // S: 992 complex
    const int tid = threadIdx.x;
    const int wave = wave_id(), lane = tid & 63;

    TRACE_BEGIN();
    if (ST != Stage::FourierOnly) {
        // ---- stage spectral coefficients: 16 B per lane, fully coalesced ----
        // (all loads of a lane are issued before the first LDS store: a rolled loop would wait for each one in turn)
        gd2_in g = (gd2_in)src;
        d2 sv[kSpecPerLane];
        if (mode == 0) {
#pragma unroll
            for (int it = 0; it < kSpecPerLane; ++it)
                if (tid + it * kThreads < NSPEC) sv[it] = g[tid + it * kThreads];
        } else {  // the spectral operator in front of the transform, applied on the fly (FieldDesc::mode)
#pragma unroll
            for (int it = 0; it < kSpecPerLane; ++it)
                if (tid + it * kThreads < NSPEC) sv[it] = staged_coefficient(mode, g, (gd2_in)src2, tid + it * kThreads, T);
        }
#pragma unroll
        for (int it = 0; it < kSpecPerLane; ++it)
            if (tid + it * kThreads < NSPEC) s[tid + it * kThreads] = sv[it];
        __syncthreads();
        TRACE_MARK(0, 0);

        // ---- inverse Legendre (legendre.f90:130-169): lane = (m, jq), latitude pairs 2jq, 2jq+1 ----
        if (tid < kInvLanes) {
            const int m = tid / 12, jq = tid - 12 * m;
            // total wavenumbers needed by this wavefront: its smallest m needs 32 - m of them (nsh2, legendre.f90:73)
            const int m_first = __builtin_amdgcn_readfirstlane((tid & ~63) / 12);
            const int ncount = ((32 - m_first) + 1) & ~1;
            double ev[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, od[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
            const d2 *pol = reinterpret_cast<const d2 *>(T.pinv) + tid;  // [n][372] x {pair 2jq, pair 2jq+1}
#pragma unroll 4
            for (int n = 0; n < ncount; n += 2) {
                const d2 pe = pol[n * kInvLanes], po = pol[(n + 1) * kInvLanes];
                const d2 xe = s[n * MX + m], xo = s[(n + 1) * MX + m];
                // n even (0-based) <-> reference n odd: symmetric ("even") sum; n+1: antisymmetric ("odd") sum
                ev[0][0] += xe.x * pe.x;  ev[0][1] += xe.y * pe.x;
                ev[1][0] += xe.x * pe.y;  ev[1][1] += xe.y * pe.y;
                od[0][0] += xo.x * po.x;  od[0][1] += xo.y * po.x;
                od[1][0] += xo.x * po.y;  od[1][1] += xo.y * po.y;
            }
            const int pr = pos_re(m), pi = pos_im(m);
            const bool keep_im = (m != 0) || (ST == Stage::LegendreOnly);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int js = 2 * jq + q, jn = kRows - 1 - js;  // reference j and il+1-j
                double *rs = cbuf + js * kCStride, *rn = cbuf + jn * kCStride;
                rn[pr] = ev[q][0] + od[q][0];
                rs[pr] = ev[q][0] - od[q][0];
                if (keep_im) {
                    rn[pi] = ev[q][1] + od[q][1];
                    rs[pi] = ev[q][1] - od[q][1];
                }
            }
        }
        __syncthreads();
        TRACE_MARK(0, 1);
    } else {
        // ---- Fourier plane from memory: (62, 48) -> unpacked rows ----
        const double *g = src;
        for (int idx = tid; idx < NFOUR; idx += kThreads) {
            const int row = idx / 62, r = idx - row * 62;
            if (r != 1) cbuf[row * kCStride + (r == 0 ? 0 : r - 1)] = g[idx];
        }
        __syncthreads();
    }

    if (ST == Stage::LegendreOnly) {
        // the Fourier plane (62, 48) leaves 16 B per lane = one zonal wavenumber (re, im) of one latitude, coalesced, all of a
        // lane's LDS reads before its first store, past the L2 like every output a later kernel consumes
        gd2_out g = (gd2_out)dst;
        d2 v[kPlanePerLane];
#pragma unroll
        for (int it = 0; it < kPlanePerLane; ++it) {
            const int item = tid + it * kThreads;
            if (item < kRows * MX) {
                const int row = item / MX, m = item - row * MX;
                const double *c = cbuf + row * kCStride;
                v[it] = d2{c[pos_re(m)], c[pos_im(m)]};
            }
        }
#pragma unroll
        for (int it = 0; it < kPlanePerLane; ++it)
            if (tid + it * kThreads < kRows * MX) __builtin_nontemporal_store(v[it], &g[tid + it * kThreads]);
        return;
    }

    double *row = rows + lane * kRowStride;
    const double *crow = cbuf + lane * kCStride;
    const bool active = lane < kRows;
    // ---- inverse FFT, group stage (radb2 + radb4 ido=12): C -> registers -> barrier -> R; wave = group ----
    // (positions beyond 60 of a compact row are never used: the ZeroPad variants substitute the structural zeros)
    {
        double o[2][8];
        if (active && wave < fft::kNumGroups) {
            if (wave < 5)
                fft::bwd_group_general_r<true>(crow, T.work, 3 + 2 * wave, o);
            else if (wave == 5)
                fft::bwd_group_first_r<true>(crow, T.work, o);
            else
                fft::bwd_group_last_r<true>(crow, T.work, o);
        }
        __syncthreads();
        if (active && wave < fft::kNumGroups) {
            if (wave < 5)
                fft::group_general_store(row, 3 + 2 * wave, o);
            else
                fft::group_edge_store(row, wave == 5 ? 0 : 11, o);
        }
    }
    __syncthreads();
    TRACE_MARK(0, 2);
    // ---- block stage (radb4 ido=3 + radb3), in place; wave = block ----
    {
        double o[4][3];
        if (active) fft::bwd_block_r(row, T.work, wave, o);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int sidx = 0; sidx < 3; ++sidx) row[wave + 8 * j + 32 * sidx] = o[j][sidx];
        }
    }
    __syncthreads();
    TRACE_MARK(0, 3);

    // ---- store grid rows: 16 B per lane, coalesced; optional 1/cos(lat) scaling (fourier.f90:87-91) ----
    // (the scale factors are fetched first: on this ISA a wait for a load also waits for the stores issued before it)
    // kGridAsFloat (descriptor-table launches of the cfg 5 model step): the field is wanted by the fp32 column physics only,
    // which narrows every value it reads -- it is narrowed here instead and takes half the bytes (8 B per lane)
    const bool as_float = (kcos & kGridAsFloat) != 0;
    kcos &= ~kGridAsFloat;
    gd2_out g = (gd2_out)dst;
    double cs[kGridPerLane];
#pragma unroll
    for (int it = 0; it < kGridPerLane; ++it) {
        const int idx = tid + it * kThreads;
        cs[it] = (kcos != 1 && idx < NGRID / 2) ? T.cosgr[idx / (IX / 2)] : 1.0;
    }
#pragma unroll
    for (int it = 0; it < kGridPerLane; ++it) {
        const int idx = tid + it * kThreads;
        if (idx < NGRID / 2) {
            const int r = idx / (IX / 2), ip = idx - r * (IX / 2);
            const double *a = rows + r * kRowStride + 2 * ip;
            d2 v{a[0], a[1]};
            if (kcos != 1) {
                v.x *= cs[it];
                v.y *= cs[it];
            }
            if (as_float) {
                const f2 w{static_cast<float>(v.x), static_cast<float>(v.y)};
                __builtin_nontemporal_store(w, &((gf2_out)dst)[idx]);
                continue;
            }
            __builtin_nontemporal_store(v, &g[idx]);  // streamed out once: do not let it displace L2 lines
        }
    }
    TRACE_MARK(0, 4);
    TRACE_END(0);
}

// ------------------------------------------------------------------------------------------------
// grid -> spec.  prescale: 0 none, 1 multiply rows by cosgr, 2 by cosgr2 (spectral.f90:229-243)
// ------------------------------------------------------------------------------------------------
template <Stage ST>
__device__ __forceinline__ void grid2spec_body(const double *__restrict__ src, double *__restrict__ dst,
                                               const DeviceTables &T, int prescale) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *rows = lds;                                    // R[48][97] while the FFT runs
    double *cbuf = lds;                                    // C[48][63] afterwards (aliases R)
    d2 *s = reinterpret_cast<d2 *>(lds + kCBufDoubles);    // output staging, 992 complex (aliases the tail of R)
    const int tid = threadIdx.x;
    const int wave = wave_id(), lane = tid & 63;

    TRACE_BEGIN();
    if (ST == Stage::Fused) {
        // ---- load grid rows (16 B per lane, coalesced) and form the symmetric / antisymmetric parts on the way in ----
        // The direct Legendre transform works on even(j) = (row(il+1-j) + row(j)) wt(j) and odd(j) = (row(il+1-j) - row(j)) wt(j)
        // of the FOURIER coefficients (legendre.f90:186-197).  The zonal FFT is linear, so the two combinations are formed here
        // from the grid rows themselves -- a lane loads the same 16 bytes of the northern and of the southern row of a latitude
        // pair -- and the FFT transforms them: the pass over the Fourier buffer that formed them after the FFT (13 % of a
        // workgroup's lifetime, and a barrier) is gone.  Rounding-level reordering: (a + b) w transformed instead of the
        // transforms added, inside the 1e-13 tolerance of tests/test_transforms_gpu.py.  The northern row of the pair takes the
        // symmetric part, the southern row the antisymmetric one, as the Legendre loop below expects.
        // (all loads of a lane are issued before the first LDS store: a rolled loop would wait for each one in turn)
        gd2_in g = (gd2_in)src;
        constexpr int kPairTasks = IY * (IX / 2);                              // 24 latitude pairs x 48 16-byte pieces
        constexpr int kPairPerLane = (kPairTasks + kThreads - 1) / kThreads;  // 3
        d2 gn[kPairPerLane], gs[kPairPerLane];
        double wj[kPairPerLane], cn[kPairPerLane], cs[kPairPerLane];
        const double *ctab = (prescale == 1) ? T.cosgr : T.cosgr2;
#pragma unroll
        for (int it = 0; it < kPairPerLane; ++it) {
            const int idx = tid + it * kThreads;
            if (idx < kPairTasks) {
                const int j = idx / (IX / 2), ip = idx - j * (IX / 2);
                gn[it] = __builtin_nontemporal_load(&g[(kRows - 1 - j) * (IX / 2) + ip]);  // read once, by this workgroup only
                gs[it] = __builtin_nontemporal_load(&g[j * (IX / 2) + ip]);
                wj[it] = T.wt[j];
                cn[it] = (prescale != 0) ? ctab[kRows - 1 - j] : 1.0;
                cs[it] = (prescale != 0) ? ctab[j] : 1.0;
            }
        }
#pragma unroll
        for (int it = 0; it < kPairPerLane; ++it) {
            const int idx = tid + it * kThreads;
            if (idx < kPairTasks) {
                const int j = idx / (IX / 2), ip = idx - j * (IX / 2);
                d2 n = gn[it], so = gs[it];
                if (prescale != 0) {  // rows times cosgr / cosgr2 first (spectral.f90:229-243), then the pair's Gaussian weight
                    n.x *= cn[it];
                    n.y *= cn[it];
                    so.x *= cs[it];
                    so.y *= cs[it];
                }
                const double w = wj[it];
                double *an = rows + (kRows - 1 - j) * kRowStride + 2 * ip, *as = rows + j * kRowStride + 2 * ip;
                an[0] = (n.x + so.x) * w;
                an[1] = (n.y + so.y) * w;
                as[0] = (n.x - so.x) * w;
                as[1] = (n.y - so.y) * w;
            }
        }
    } else if (ST == Stage::FourierOnly) {
        // ---- load grid rows (16 B per lane, coalesced) ----
        // (all loads of a lane are issued before the first LDS store: a rolled loop would wait for each one in turn)
        gd2_in g = (gd2_in)src;
        d2 gv[kGridPerLane];
        double cs[kGridPerLane];
        const double *ctab = (prescale == 1) ? T.cosgr : T.cosgr2;
#pragma unroll
        for (int it = 0; it < kGridPerLane; ++it) {
            const int idx = tid + it * kThreads;
            if (idx < NGRID / 2) gv[it] = __builtin_nontemporal_load(&g[idx]);  // read once, by this workgroup only
            cs[it] = (prescale != 0 && idx < NGRID / 2) ? ctab[idx / (IX / 2)] : 1.0;
        }
#pragma unroll
        for (int it = 0; it < kGridPerLane; ++it) {
            const int idx = tid + it * kThreads;
            if (idx < NGRID / 2) {
                const int r = idx / (IX / 2), ip = idx - r * (IX / 2);
                d2 v = gv[it];
                if (prescale != 0) {
                    v.x *= cs[it];
                    v.y *= cs[it];
                }
                double *a = rows + r * kRowStride + 2 * ip;
                a[0] = v.x;
                a[1] = v.y;
            }
        }
    }
    if (ST != Stage::LegendreOnly) {
        __syncthreads();
        TRACE_MARK(1, 0);
        double *row = rows + lane * kRowStride;
        const bool active = lane < kRows;
        // ---- forward FFT, block stage (radf3 + radf4 ido=3), in place; wave = block ----
        {
            double o[12];
            if (active) fft::fwd_block_r(row, T.work, wave, o);
            __syncthreads();
            if (active) {
#pragma unroll
                for (int e = 0; e < 12; ++e) row[12 * wave + e] = o[e];
            }
        }
        __syncthreads();
        TRACE_MARK(1, 1);
        // ---- group stage (radf4 ido=12 + radf2), scaled by fp32(1/96), retained wavenumbers only; wave = group ----
        {
            fft::Quad q[4];
            double o5[5];
            if (active && wave < fft::kNumGroups) {
                if (wave < 5)
                    fft::fwd_group_general_r(row, T.work, 3 + 2 * wave, T.fft_scale, q);
                else if (wave == 5)
                    fft::fwd_group_first_r(row, T.work, T.fft_scale, o5);
                else
                    fft::fwd_group_last_r(row, T.work, T.fft_scale, q);
            }
            __syncthreads();
            // (the FFT rows are dead from here on: the output staging area, which aliases their tail, can be cleared -- the
            // Legendre loop only writes the coefficients the reference fills)
            if (ST == Stage::Fused)
                for (int idx = tid; idx < NSPEC; idx += kThreads) s[idx] = d2{0.0, 0.0};
            double *crow = cbuf + lane * kCStride;  // R -> registers -> barrier -> C
            if (active && wave < fft::kNumGroups) {
                if (wave < 5) {
                    fft::fwd_group_general_store(crow, 3 + 2 * wave, q);
                } else if (wave == 5) {
                    crow[0] = o5[0]; crow[47] = o5[1]; crow[48] = o5[2]; crow[23] = o5[3]; crow[24] = o5[4];
                    crow[61] = 0.0;  // Im of the zonal mean: fourier.f90:117 sets output(2, j) = 0
                } else {
                    crow[11] = q[0].lo_r; crow[12] = q[0].lo_i;
                    crow[35] = q[1].lo_r; crow[36] = q[1].lo_i;
                    crow[59] = q[1].hi_r; crow[60] = q[1].hi_i;
                }
            }
        }
        __syncthreads();
        TRACE_MARK(1, 2);
    } else {
        // The Legendre stage on its own: the Fourier plane (62, 48) comes in 16 B per lane = one zonal wavenumber (re, im) of one
        // latitude; the lane fetches the same wavenumber of the mirrored latitude with it and stores step 1 of the direct
        // transform (legendre.f90:190-199): north row <- symmetric part, south row <- antisymmetric part, both times the
        // Gaussian weight -- as the fused kernel does while it loads the grid.  All loads of a lane before its first LDS store.
        gd2_in g = (gd2_in)src;
        d2 nv[kHalfPlanePerLane], sv[kHalfPlanePerLane];
        double w[kHalfPlanePerLane];
#pragma unroll
        for (int it = 0; it < kHalfPlanePerLane; ++it) {
            const int item = tid + it * kThreads;
            if (item < IY * MX) {
                const int j = item / MX, m = item - j * MX;
                // (plain loads: with the streaming hint the same kernel is 20 % slower while its input still sits in the 256 MB
                // Infinity Cache -- batches up to 4096 fields -- and no faster beyond, profiles/r05_legendre_only.txt)
                nv[it] = g[(kRows - 1 - j) * MX + m];
                sv[it] = g[j * MX + m];
                w[it] = T.wt[j];
            }
        }
        for (int idx = tid; idx < NSPEC; idx += kThreads) s[idx] = d2{0.0, 0.0};  // (step 2 only writes what the reference fills)
#pragma unroll
        for (int it = 0; it < kHalfPlanePerLane; ++it) {
            const int item = tid + it * kThreads;
            if (item < IY * MX) {
                const int j = item / MX, m = item - j * MX;
                const int pr = pos_re(m), pi = pos_im(m);
                double *rn = cbuf + (kRows - 1 - j) * kCStride, *rs = cbuf + j * kCStride;
                rn[pr] = (nv[it].x + sv[it].x) * w[it];
                rs[pr] = (nv[it].x - sv[it].x) * w[it];
                rn[pi] = (nv[it].y + sv[it].y) * w[it];
                rs[pi] = (nv[it].y - sv[it].y) * w[it];
            }
        }
        __syncthreads();
    }

    if (ST == Stage::FourierOnly) {
        double *g = dst;
        for (int idx = tid; idx < NFOUR; idx += kThreads) {
            const int r0 = idx / 62, r = idx - r0 * 62;
            g[idx] = (r == 1) ? 0.0 : cbuf[r0 * kCStride + (r == 0 ? 0 : r - 1)];  // fourier.f90:117
        }
        return;
    }

    // ---- direct Legendre (legendre.f90:175-221) ----
    // (step 1 -- north row <- symmetric part, south row <- antisymmetric part, both times the Gaussian weight -- happened while
    // the input was staged, in both forms of the kernel)
    TRACE_MARK(1, 3);
    // step 2: lane = (m, parity, two valid n of that parity) from the work list of the context (capi.hip: dir_lanes);
    // sum over the 24 latitude pairs in reference order
    if (tid < T.ndir) {
        const int4 meta = reinterpret_cast<const int4 *>(T.dirmeta)[tid];
        const int pr = meta.x & 0xff, pi = meta.x >> 8, par = meta.y;
        double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        const d2 *pol = reinterpret_cast<const d2 *>(T.pdir) + tid;  // [j][dir_stride] x {n_a, n_b}
        const int pstride = T.dir_stride;
        const double *base = cbuf + (par ? 0 : (kRows - 1) * kCStride);
        const int rstep = par ? kCStride : -kCStride;
#pragma unroll 4
        for (int j = 0; j < IY; ++j) {
            const d2 p = pol[j * pstride];
            const double *r = base + j * rstep;
            const double xr = r[pr], xi = r[pi];
            acc[0][0] += p.x * xr;  acc[0][1] += p.x * xi;
            acc[1][0] += p.y * xr;  acc[1][1] += p.y * xi;
        }
        s[meta.z] = d2{acc[0][0], acc[0][1]};
        if (meta.w >= 0) s[meta.w] = d2{acc[1][0], acc[1][1]};
    }
    __syncthreads();
    TRACE_MARK(1, 4);
    gd2_out g = (gd2_out)dst;
    for (int idx = tid; idx < NSPEC; idx += kThreads) __builtin_nontemporal_store(s[idx], &g[idx]);
    TRACE_MARK(1, 5);
    TRACE_END(1);
}

// ------------------------------------------------------------------------------------------------
// kernels: contiguous batch (C ABI) and descriptor table (model step: one launch for all fields of all members)
// ------------------------------------------------------------------------------------------------
template <Stage ST>
__global__ __launch_bounds__(kThreads) void spec2grid_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                             DeviceTables T, int kcos) {
    const size_t f = blockIdx.x;
    constexpr size_t in = (ST == Stage::FourierOnly) ? NFOUR : 2 * NSPEC, out = (ST == Stage::LegendreOnly) ? NFOUR : NGRID;
    spec2grid_body<ST>(src + f * in, dst + f * out, T, kcos);
}

template <Stage ST>
__global__ __launch_bounds__(kThreads) void grid2spec_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                             DeviceTables T, int prescale) {
    const size_t f = blockIdx.x;
    constexpr size_t in = (ST == Stage::LegendreOnly) ? NFOUR : NGRID, out = (ST == Stage::FourierOnly) ? NFOUR : 2 * NSPEC;
    grid2spec_body<ST>(src + f * in, dst + f * out, T, prescale);
}

__global__ __launch_bounds__(kThreads) void spec2grid_table_kernel(const FieldDesc *__restrict__ table, DeviceTables T) {
    const FieldDesc e = table[blockIdx.x];
    spec2grid_body<Stage::Fused>(e.src, e.dst, T, e.flag, e.mode, e.src2);
}

// The same launch with the range check of every member in front of the fields: workgroup i < members checks member i on the
// spectral state the step is about to read -- the check of the PREVIOUS step, for hosts that collect it one step late
// (spd_model_check_defer, diagnostics_block.hpp).  In front, so that the 8 us chain of a check runs beside the bulk of the launch
// instead of behind it; padded to a multiple of 8 workgroups, so that the field blocks keep their XCDs (blockIdx mod 8).
// Held to the transform's 64 registers and 8 wavefronts per SIMD (the check takes its loads two rounds at a time here).
static_assert(kThreads == 64 * KX, "a check block is a workgroup of the transform launch");
__global__ __launch_bounds__(kThreads, 8) void spec2grid_table_check_kernel(const FieldDesc *__restrict__ table, DeviceTables T,
                                                                            int members, CheckArgs check) {
    const int front = (members + 7) & ~7;
    if (static_cast<int>(blockIdx.x) < front) {
        if (static_cast<int>(blockIdx.x) < members) diagnostics_block<2>(check, T, check.first + static_cast<int>(blockIdx.x));
        return;
    }
    const FieldDesc e = table[blockIdx.x - front];
    spec2grid_body<Stage::Fused>(e.src, e.dst, T, e.flag, e.mode, e.src2);
}

__global__ __launch_bounds__(kThreads) void grid2spec_table_kernel(const FieldDesc *__restrict__ table, DeviceTables T) {
    const FieldDesc e = table[blockIdx.x];
    grid2spec_body<Stage::Fused>(e.src, e.dst, T, e.flag);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// (the attribute is per device: a process that drives several GPUs sets it once on each; `done` belongs to the call site)
static hipError_t configure_once(const void *kernel, bool (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsBytes));
    done[dev] = e == hipSuccess;
    return e;
}
#define SPD_CONFIGURE(kernel)                                            \
    ([] {                                                                \
        static bool done[64] = {};                                       \
        return configure_once(reinterpret_cast<const void *>(kernel), done); \
    }())

template <Stage ST>
static hipError_t launch_s2g(const double *src, double *dst, const DeviceTables &T, int nfields, int kcos, hipStream_t st) {
    const hipError_t cfg = SPD_CONFIGURE(&spec2grid_kernel<ST>);
    if (cfg != hipSuccess) return cfg;
    hipLaunchKernelGGL((spec2grid_kernel<ST>), dim3(nfields), dim3(kThreads), kLdsBytes, st, src, dst, T, kcos);
    return hipGetLastError();
}

template <Stage ST>
static hipError_t launch_g2s(const double *src, double *dst, const DeviceTables &T, int nfields, int prescale,
                             hipStream_t st) {
    const hipError_t cfg = SPD_CONFIGURE(&grid2spec_kernel<ST>);
    if (cfg != hipSuccess) return cfg;
    hipLaunchKernelGGL((grid2spec_kernel<ST>), dim3(nfields), dim3(kThreads), kLdsBytes, st, src, dst, T, prescale);
    return hipGetLastError();
}

hipError_t run_spec2grid_table(const DeviceTables &T, const FieldDesc *table, int nfields, hipStream_t st) {
    if (nfields == 0) return hipSuccess;
    const hipError_t cfg = SPD_CONFIGURE(&spec2grid_table_kernel);
    if (cfg != hipSuccess) return cfg;
    launch(spec2grid_table_kernel, dim3(nfields), dim3(kThreads), kLdsBytes, st, table, T);
    return hipGetLastError();
}

hipError_t run_spec2grid_table_check(const DeviceTables &T, const FieldDesc *table, int nfields, const CheckArgs &check, int members,
                                     hipStream_t st) {
    const hipError_t cfg = SPD_CONFIGURE(&spec2grid_table_check_kernel);
    if (cfg != hipSuccess) return cfg;
    launch(spec2grid_table_check_kernel, dim3(((members + 7) & ~7) + nfields), dim3(kThreads), kLdsBytes, st, table, T, members, check);
    return hipGetLastError();
}

hipError_t run_grid2spec_table(const DeviceTables &T, const FieldDesc *table, int nfields, hipStream_t st) {
    if (nfields == 0) return hipSuccess;
    const hipError_t cfg = SPD_CONFIGURE(&grid2spec_table_kernel);
    if (cfg != hipSuccess) return cfg;
    launch(grid2spec_table_kernel, dim3(nfields), dim3(kThreads), kLdsBytes, st, table, T);
    return hipGetLastError();
}

hipError_t run_spec2grid(const DeviceTables &T, int stage, const double *src, double *dst, int kcos, int nfields,
                         hipStream_t stream) {
    if (nfields == 0) return hipSuccess;
    switch (stage) {
        case 0: return launch_s2g<Stage::Fused>(src, dst, T, nfields, kcos, stream);
        case 1: return launch_s2g<Stage::LegendreOnly>(src, dst, T, nfields, kcos, stream);
        default: return launch_s2g<Stage::FourierOnly>(src, dst, T, nfields, kcos, stream);
    }
}

hipError_t run_grid2spec(const DeviceTables &T, int stage, const double *src, double *dst, int prescale, int nfields,
                         hipStream_t stream) {
    if (nfields == 0) return hipSuccess;
    switch (stage) {
        case 0: return launch_g2s<Stage::Fused>(src, dst, T, nfields, prescale, stream);
        case 1: return launch_g2s<Stage::LegendreOnly>(src, dst, T, nfields, prescale, stream);
        default: return launch_g2s<Stage::FourierOnly>(src, dst, T, nfields, prescale, stream);
    }
}

}  // namespace spd

#ifdef SPD_TRACE
// mean time (in 100 MHz wall-clock ticks = 10 ns) from kernel entry to each phase boundary; resets the counters
extern "C" int spd_trace_read(double *out /* [2][8] */, double *count /* [2] */) {
    unsigned long long acc[2][8], cnt[2], zero[2][8] = {};
    if (hipMemcpyFromSymbol(acc, HIP_SYMBOL(spd::spd_trace_acc), sizeof(acc)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(cnt, HIP_SYMBOL(spd::spd_trace_cnt), sizeof(cnt)) != hipSuccess) return -1;
    for (int d = 0; d < 2; ++d) {
        count[d] = static_cast<double>(cnt[d]);
        for (int i = 0; i < 8; ++i) out[d * 8 + i] = cnt[d] ? static_cast<double>(acc[d][i]) / cnt[d] : 0.0;
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(spd::spd_trace_acc), zero, sizeof(acc));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(spd::spd_trace_cnt), zero, sizeof(cnt));
    return 0;
}
#endif
