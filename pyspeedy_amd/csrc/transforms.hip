// Batched spectral transforms for gfx950: spec2grid (inverse Legendre + inverse zonal FFT) and grid2spec
// (forward FFT + direct Legendre), each as ONE fused kernel per batch tile, so a field crosses HBM exactly
// once in each direction (15 872 B spectral + 36 864 B grid = 52 736 B algorithmic bytes per field).
//
// Reference behaviour reproduced: ModSpectral_spec2grid / grid2spec (speedy.f90/spectral.f90:251-273) =
// ModLegendre_legendre_inv / legendre (legendre.f90:130-221) composed with ModFourier_fourier_inv / fourier
// (fourier.f90:63-123).  The stage-only entry points run the same kernel with one stage disabled.
//
// Work decomposition (one workgroup = FPW fields, 256 threads = 4 wavefronts):
//   Legendre   lane <-> zonal wavenumber m (31 of them), a thread owns 4 latitude pairs (inverse) or 4 total
//              wavenumbers n (direct) of FPW fields; re and im of one coefficient stay in one lane (16-byte LDS
//              reads), the associated-Legendre values are streamed from an L2-resident table laid out so that a
//              wavefront reads 32 contiguous bytes per lane (table is zero outside the triangle -> no masks).
//   FFT        (row, block) / (row, group) task lists over all FPW*48 rows, lanes on different rows, see fft96.hpp.
//   LDS        two row buffers [FPW*48][97] doubles (odd row stride: conflict-free for lanes-on-rows access);
//              the staged spectral input aliases the second buffer.  74 496 B per field.
#include <hip/hip_runtime.h>

#include "device_tables.hpp"
#include "fft96.hpp"

namespace spd {

constexpr int kThreads = 256;
constexpr int kRowStride = 97;
constexpr int kRows = IL;  // 48 rows per field

enum class Stage { Fused, LegendreOnly, FourierOnly };

using d2 = double __attribute__((ext_vector_type(2)));

// position of coefficient (m, re|im) in an unpacked FFT row (fourier.f90:74-81): re(m) -> 2m-1, im(m) -> 2m,
// re(0) -> 0.  im(0) has no slot in the transform; the stage-only kernels park it at position 61.
__device__ inline int pos_re(int m) { return m == 0 ? 0 : 2 * m - 1; }
__device__ inline int pos_im(int m) { return m == 0 ? 61 : 2 * m; }

// ------------------------------------------------------------------------------------------------
// spec -> grid
// ------------------------------------------------------------------------------------------------
template <Stage ST, int FPW>
__global__ __launch_bounds__(kThreads) void spec2grid_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                             DeviceTables T, int nfields, int kcos) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *bufA = lds;                                  // [FPW*48][97]
    double *bufB = lds + FPW * kRows * kRowStride;       // [FPW*48][97], spectral staging aliases it
    const int tid = threadIdx.x;
    const int f0 = blockIdx.x * FPW;
    const int nf = min(FPW, nfields - f0);  // fields present in this tile (>= 1)

    if (ST != Stage::FourierOnly) {
        // ---- stage spectral coefficients: 992 complex per field, 16 B per lane, fully coalesced ----
        const d2 *g = reinterpret_cast<const d2 *>(src) + static_cast<size_t>(f0) * NSPEC;
        d2 *s = reinterpret_cast<d2 *>(bufB);
        for (int idx = tid; idx < FPW * NSPEC; idx += kThreads) s[idx] = (idx < nf * NSPEC) ? g[idx] : d2{0.0, 0.0};
        __syncthreads();

        // ---- inverse Legendre (legendre.f90:130-169) ----
        if (tid < MX * 6) {
            const int m = tid % MX, jq = tid / MX;
            double ev[FPW][4][2], od[FPW][4][2];
#pragma unroll
            for (int f = 0; f < FPW; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) ev[f][q][0] = ev[f][q][1] = od[f][q][0] = od[f][q][1] = 0.0;
            const d2 *pol = reinterpret_cast<const d2 *>(T.pinv) + 2 * tid;  // [n][jq*31+m][4]
#pragma unroll 2
            for (int n = 0; n < NX; n += 2) {
                // n even (0-based) <-> reference n odd: "even" sum; n+1 -> "odd" sum
                const d2 pe0 = pol[(n * 186) * 2], pe1 = pol[(n * 186) * 2 + 1];
                const d2 po0 = pol[((n + 1) * 186) * 2], po1 = pol[((n + 1) * 186) * 2 + 1];
                const double pe[4] = {pe0.x, pe0.y, pe1.x, pe1.y}, po[4] = {po0.x, po0.y, po1.x, po1.y};
#pragma unroll
                for (int f = 0; f < FPW; ++f) {
                    const d2 xe = s[f * NSPEC + n * MX + m], xo = s[f * NSPEC + (n + 1) * MX + m];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ev[f][q][0] += xe.x * pe[q];
                        ev[f][q][1] += xe.y * pe[q];
                        od[f][q][0] += xo.x * po[q];
                        od[f][q][1] += xo.y * po[q];
                    }
                }
            }
            const int pr = pos_re(m), pi = pos_im(m);
            const bool keep_im = (m != 0) || (ST == Stage::LegendreOnly);
#pragma unroll
            for (int f = 0; f < FPW; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int js = 4 * jq + q, jn = kRows - 1 - js;  // reference j and il+1-j
                    double *rs = bufA + (f * kRows + js) * kRowStride, *rn = bufA + (f * kRows + jn) * kRowStride;
                    rn[pr] = ev[f][q][0] + od[f][q][0];
                    rs[pr] = ev[f][q][0] - od[f][q][0];
                    if (keep_im) {
                        rn[pi] = ev[f][q][1] + od[f][q][1];
                        rs[pi] = ev[f][q][1] - od[f][q][1];
                    }
                }
        }
        __syncthreads();
    } else {
        // ---- Fourier plane from memory: (62, 48) per field -> unpacked rows ----
        const double *g = src + static_cast<size_t>(f0) * NFOUR;
        for (int idx = tid; idx < nf * NFOUR; idx += kThreads) {
            const int f = idx / NFOUR, rem = idx - f * NFOUR, row = rem / 62, r = rem - row * 62;
            if (r != 1) bufA[(f * kRows + row) * kRowStride + (r == 0 ? 0 : r - 1)] = g[idx];
        }
        __syncthreads();
    }

    if (ST == Stage::LegendreOnly) {
        double *g = dst + static_cast<size_t>(f0) * NFOUR;
        for (int idx = tid; idx < nf * NFOUR; idx += kThreads) {
            const int f = idx / NFOUR, rem = idx - f * NFOUR, row = rem / 62, r = rem - row * 62;
            g[idx] = bufA[(f * kRows + row) * kRowStride + (r == 0 ? 0 : (r == 1 ? 61 : r - 1))];
        }
        return;
    }

    // ---- inverse FFT, group stage (radb2 + radb4 ido=12): A -> B ----
    const int nrows = nf * kRows;
    for (int task = tid; task < nrows * fft::kNumGroups; task += kThreads) {
        const int g = task / nrows, row = task - g * nrows;
        fft::bwd_group<true>(bufA + row * kRowStride, bufB + row * kRowStride, T.work, g);
    }
    __syncthreads();
    // ---- block stage (radb4 ido=3 + radb3): B -> A ----
    for (int task = tid; task < nrows * fft::kNumBlocks; task += kThreads) {
        const int kk = task / nrows, row = task - kk * nrows;
        fft::bwd_block(bufB + row * kRowStride, bufA + row * kRowStride, T.work, kk);
    }
    __syncthreads();

    // ---- store grid rows: 16 B per lane, coalesced; optional 1/cos(lat) scaling (fourier.f90:87-91) ----
    d2 *g = reinterpret_cast<d2 *>(dst) + static_cast<size_t>(f0) * (NGRID / 2);
    for (int idx = tid; idx < nf * (NGRID / 2); idx += kThreads) {
        const int row = idx / (IX / 2), ip = idx - row * (IX / 2);  // row counts over all fields of the tile
        const double *a = bufA + row * kRowStride + 2 * ip;
        d2 v{a[0], a[1]};
        if (kcos != 1) {
            const double c = T.cosgr[row % kRows];
            v.x *= c;
            v.y *= c;
        }
        g[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// grid -> spec.  prescale: 0 none, 1 multiply rows by cosgr, 2 by cosgr2 (spectral.f90:229-243)
// ------------------------------------------------------------------------------------------------
template <Stage ST, int FPW>
__global__ __launch_bounds__(kThreads) void grid2spec_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                             DeviceTables T, int nfields, int prescale) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *bufA = lds;
    double *bufB = lds + FPW * kRows * kRowStride;
    const int tid = threadIdx.x;
    const int f0 = blockIdx.x * FPW;
    const int nf = min(FPW, nfields - f0);
    const int nrows = nf * kRows;

    if (ST != Stage::LegendreOnly) {
        // ---- load grid rows (16 B per lane, coalesced) into A ----
        const d2 *g = reinterpret_cast<const d2 *>(src) + static_cast<size_t>(f0) * (NGRID / 2);
        for (int idx = tid; idx < nf * (NGRID / 2); idx += kThreads) {
            const int row = idx / (IX / 2), ip = idx - row * (IX / 2);
            d2 v = g[idx];
            if (prescale != 0) {
                const double c = (prescale == 1) ? T.cosgr[row % kRows] : T.cosgr2[row % kRows];
                v.x *= c;
                v.y *= c;
            }
            double *a = bufA + row * kRowStride + 2 * ip;
            a[0] = v.x;
            a[1] = v.y;
        }
        __syncthreads();
        // ---- forward FFT, block stage (radf3 + radf4 ido=3): A -> B ----
        for (int task = tid; task < nrows * fft::kNumBlocks; task += kThreads) {
            const int kk = task / nrows, row = task - kk * nrows;
            fft::fwd_block(bufA + row * kRowStride, bufB + row * kRowStride, T.work, kk);
        }
        __syncthreads();
        // ---- group stage (radf4 ido=12 + radf2), scaled by fp32(1/96), retained wavenumbers only: B -> A ----
        for (int task = tid; task < nrows * fft::kNumGroups; task += kThreads) {
            const int g2 = task / nrows, row = task - g2 * nrows;
            fft::fwd_group(bufB + row * kRowStride, bufA + row * kRowStride, T.work, g2, T.fft_scale);
        }
        __syncthreads();
    } else {
        const double *g = src + static_cast<size_t>(f0) * NFOUR;
        for (int idx = tid; idx < nf * NFOUR; idx += kThreads) {
            const int f = idx / NFOUR, rem = idx - f * NFOUR, row = rem / 62, r = rem - row * 62;
            bufA[(f * kRows + row) * kRowStride + (r == 0 ? 0 : (r == 1 ? 61 : r - 1))] = g[idx];
        }
        __syncthreads();
    }

    if (ST == Stage::FourierOnly) {
        double *g = dst + static_cast<size_t>(f0) * NFOUR;
        for (int idx = tid; idx < nf * NFOUR; idx += kThreads) {
            const int f = idx / NFOUR, rem = idx - f * NFOUR, row = rem / 62, r = rem - row * 62;
            g[idx] = (r == 1) ? 0.0 : bufA[(f * kRows + row) * kRowStride + (r == 0 ? 0 : r - 1)];  // fourier.f90:117
        }
        return;
    }

    // ---- direct Legendre (legendre.f90:175-221): thread = (m, 4 consecutive n), loop over latitude pairs ----
    if (tid < MX * 8) {
        const int m = tid % MX, nq = tid / MX;
        const int pr = pos_re(m), pi = pos_im(m);
        const bool has_im = (m != 0) || (ST == Stage::LegendreOnly);
        double acc[FPW][4][2];
#pragma unroll
        for (int f = 0; f < FPW; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[f][q][0] = acc[f][q][1] = 0.0;
        const d2 *pol = reinterpret_cast<const d2 *>(T.pdir) + 2 * tid;  // [j][nq*31+m][4]
#pragma unroll 2
        for (int j = 0; j < IY; ++j) {
            const d2 p0 = pol[(j * 248) * 2], p1 = pol[(j * 248) * 2 + 1];
            const double p[4] = {p0.x, p0.y, p1.x, p1.y};
            const double w = T.wt[j];
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                const double *rs = bufA + (f * kRows + j) * kRowStride, *rn = bufA + (f * kRows + kRows - 1 - j) * kRowStride;
                const double nr = rn[pr], sr = rs[pr];
                const double ni = has_im ? rn[pi] : 0.0, si = has_im ? rs[pi] : 0.0;
                const double er = (nr + sr) * w, orr = (nr - sr) * w;
                const double ei = (ni + si) * w, oi = (ni - si) * w;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // n = 4*nq + q ; q even <-> reference n odd <-> symmetric ("even") part
                    acc[f][q][0] += p[q] * ((q & 1) ? orr : er);
                    acc[f][q][1] += p[q] * ((q & 1) ? oi : ei);
                }
            }
        }
        d2 *g = reinterpret_cast<d2 *>(dst) + static_cast<size_t>(f0) * NSPEC;
#pragma unroll
        for (int f = 0; f < FPW; ++f)
            if (f < nf)
#pragma unroll
                for (int q = 0; q < 4; ++q) g[f * NSPEC + (4 * nq + q) * MX + m] = d2{acc[f][q][0], acc[f][q][1]};
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <int FPW>
static constexpr size_t lds_bytes() { return static_cast<size_t>(2) * FPW * kRows * kRowStride * sizeof(double); }

template <Stage ST, int FPW>
static hipError_t launch_s2g(const double *src, double *dst, const DeviceTables &T, int nfields, int kcos,
                             hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&spec2grid_kernel<ST, FPW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes<FPW>()));
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int grid = (nfields + FPW - 1) / FPW;
    hipLaunchKernelGGL((spec2grid_kernel<ST, FPW>), dim3(grid), dim3(kThreads), lds_bytes<FPW>(), stream, src, dst, T,
                       nfields, kcos);
    return hipGetLastError();
}

template <Stage ST, int FPW>
static hipError_t launch_g2s(const double *src, double *dst, const DeviceTables &T, int nfields, int prescale,
                             hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&grid2spec_kernel<ST, FPW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes<FPW>()));
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int grid = (nfields + FPW - 1) / FPW;
    hipLaunchKernelGGL((grid2spec_kernel<ST, FPW>), dim3(grid), dim3(kThreads), lds_bytes<FPW>(), stream, src, dst, T,
                       nfields, prescale);
    return hipGetLastError();
}

// fields-per-workgroup policy.  Measured on MI355X (tools/perf_transforms.py): one field per workgroup (74.5 KB of
// LDS -> two workgroups per CU) beats two fields per workgroup (149 KB -> one per CU) at every batch size, because
// the kernel is latency-bound and needs the occupancy; the 2-field variant stays available for experiments.
static inline int pick_fpw(int nfields, int forced) {
    (void)nfields;
    return forced == 2 ? 2 : 1;
}

hipError_t run_spec2grid(const DeviceTables &T, int stage, const double *src, double *dst, int kcos, int nfields,
                         hipStream_t stream, int fpw) {
    if (nfields == 0) return hipSuccess;
    const bool two = pick_fpw(nfields, fpw) == 2;
    switch (stage) {
        case 0: return two ? launch_s2g<Stage::Fused, 2>(src, dst, T, nfields, kcos, stream)
                           : launch_s2g<Stage::Fused, 1>(src, dst, T, nfields, kcos, stream);
        case 1: return two ? launch_s2g<Stage::LegendreOnly, 2>(src, dst, T, nfields, kcos, stream)
                           : launch_s2g<Stage::LegendreOnly, 1>(src, dst, T, nfields, kcos, stream);
        default: return two ? launch_s2g<Stage::FourierOnly, 2>(src, dst, T, nfields, kcos, stream)
                            : launch_s2g<Stage::FourierOnly, 1>(src, dst, T, nfields, kcos, stream);
    }
}

hipError_t run_grid2spec(const DeviceTables &T, int stage, const double *src, double *dst, int prescale, int nfields,
                         hipStream_t stream, int fpw) {
    if (nfields == 0) return hipSuccess;
    const bool two = pick_fpw(nfields, fpw) == 2;
    switch (stage) {
        case 0: return two ? launch_g2s<Stage::Fused, 2>(src, dst, T, nfields, prescale, stream)
                           : launch_g2s<Stage::Fused, 1>(src, dst, T, nfields, prescale, stream);
        case 1: return two ? launch_g2s<Stage::LegendreOnly, 2>(src, dst, T, nfields, prescale, stream)
                           : launch_g2s<Stage::LegendreOnly, 1>(src, dst, T, nfields, prescale, stream);
        default: return two ? launch_g2s<Stage::FourierOnly, 2>(src, dst, T, nfields, prescale, stream)
                            : launch_g2s<Stage::FourierOnly, 1>(src, dst, T, nfields, prescale, stream);
    }
}

}  // namespace spd
