// Spectral-space operators of ModSpectral_t (speedy.f90/spectral.f90:134-296) on batches of fields.
// Pure streaming stencils in the total-wavenumber index n: one thread per complex coefficient, 16-byte accesses,
// neighbours n-1 / n+1 are 496 B away in the same field (L1/L2 hits).  Multiplication by the imaginary unit
// is written out: (a + ib) * i = (-b) + ia.
#include <hip/hip_runtime.h>

#include "device_tables.hpp"

namespace spd {

using d2 = double __attribute__((ext_vector_type(2)));
constexpr int kOpThreads = 256;

__device__ inline d2 times_i(d2 z) { return d2{-z.y, z.x}; }

// mode 0: vort2vel (tables uvdx/uvdym/uvdyp, x-derivative coefficient depends on (m,n))
// mode 1: vel2vort (tables gradx/vddym/vddyp, x-derivative coefficient depends on m only)
template <int MODE>
__global__ __launch_bounds__(kOpThreads) void uv_vordiv_kernel(const d2 *__restrict__ a, const d2 *__restrict__ b,
                                                               d2 *__restrict__ oa, d2 *__restrict__ ob, DeviceTables T,
                                                               long total) {
    const long gid = static_cast<long>(blockIdx.x) * kOpThreads + threadIdx.x;
    if (gid >= total) return;
    const int k = static_cast<int>(gid % NSPEC), n = k / MX, m = k - n * MX;
    const double *tym = MODE == 0 ? T.uvdym : T.vddym, *typ = MODE == 0 ? T.uvdyp : T.vddyp;
    const double dx = MODE == 0 ? T.uvdx[k] : T.gradx[m];
    const double cm = tym[k], cp = typ[k];
    const d2 za = a[gid], zb = b[gid];
    // zp = dx * a * i ; zc = dx * b * i   (spectral.f90:168-171, 198-199)
    const d2 zp = times_i(d2{dx * za.x, dx * za.y}), zc = times_i(d2{dx * zb.x, dx * zb.y});
    d2 ra, rb;
    if (n == 0) {
        const d2 an = a[gid + MX], bn = b[gid + MX];
        ra = d2{zc.x - cp * an.x, zc.y - cp * an.y};
        rb = d2{zp.x + cp * bn.x, zp.y + cp * bn.y};
    } else if (n == NX - 1) {
        const d2 ap = a[gid - MX], bp = b[gid - MX];
        ra = d2{cm * ap.x, cm * ap.y};
        rb = d2{-cm * bp.x, -cm * bp.y};
    } else {
        const d2 ap = a[gid - MX], bp = b[gid - MX], an = a[gid + MX], bn = b[gid + MX];
        ra = d2{cm * ap.x - cp * an.x + zc.x, cm * ap.y - cp * an.y + zc.y};
        rb = d2{-cm * bp.x + cp * bn.x + zp.x, -cm * bp.y + cp * bn.y + zp.y};
    }
    oa[gid] = ra;
    ob[gid] = rb;
}

__global__ __launch_bounds__(kOpThreads) void gradient_kernel(const d2 *__restrict__ psi, d2 *__restrict__ dx,
                                                              d2 *__restrict__ dy, DeviceTables T, long total) {
    const long gid = static_cast<long>(blockIdx.x) * kOpThreads + threadIdx.x;
    if (gid >= total) return;
    const int k = static_cast<int>(gid % NSPEC), n = k / MX, m = k - n * MX;
    const d2 z = psi[gid];
    const double g = T.gradx[m];
    dx[gid] = times_i(d2{g * z.x, g * z.y});
    d2 r;
    if (n == 0) {
        const d2 zn = psi[gid + MX];
        const double cp = T.gradyp[k];
        r = d2{cp * zn.x, cp * zn.y};
    } else if (n == NX - 1) {
        const d2 zpv = psi[gid - MX];
        const double cm = T.gradym[k];
        r = d2{-cm * zpv.x, -cm * zpv.y};
    } else {
        const d2 zpv = psi[gid - MX], zn = psi[gid + MX];
        const double cm = T.gradym[k], cp = T.gradyp[k];
        r = d2{-cm * zpv.x + cp * zn.x, -cm * zpv.y + cp * zn.y};
    }
    dy[gid] = r;
}

// out = sign * in * table   (laplacian: -el2, laplacian_inv: -elm2, truncate: +trfilt)
__global__ __launch_bounds__(kOpThreads) void scale_kernel(const d2 *__restrict__ in, d2 *__restrict__ out,
                                                           const double *__restrict__ table, double sign, long total) {
    const long gid = static_cast<long>(blockIdx.x) * kOpThreads + threadIdx.x;
    if (gid >= total) return;
    const double c = table[gid % NSPEC];
    const d2 z = in[gid];
    out[gid] = d2{sign * z.x * c, sign * z.y * c};
}

static inline unsigned blocks_for(long total) { return static_cast<unsigned>((total + kOpThreads - 1) / kOpThreads); }

hipError_t run_vort2vel(const DeviceTables &T, const double *vor, const double *div, double *ucos, double *vcos,
                        int nfields, hipStream_t s) {
    const long total = static_cast<long>(nfields) * NSPEC;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(uv_vordiv_kernel<0>, dim3(blocks_for(total)), dim3(kOpThreads), 0, s,
                       reinterpret_cast<const d2 *>(vor), reinterpret_cast<const d2 *>(div),
                       reinterpret_cast<d2 *>(ucos), reinterpret_cast<d2 *>(vcos), T, total);
    return hipGetLastError();
}

hipError_t run_vel2vort(const DeviceTables &T, const double *ucos, const double *vcos, double *vor, double *div,
                        int nfields, hipStream_t s) {
    const long total = static_cast<long>(nfields) * NSPEC;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(uv_vordiv_kernel<1>, dim3(blocks_for(total)), dim3(kOpThreads), 0, s,
                       reinterpret_cast<const d2 *>(ucos), reinterpret_cast<const d2 *>(vcos),
                       reinterpret_cast<d2 *>(vor), reinterpret_cast<d2 *>(div), T, total);
    return hipGetLastError();
}

hipError_t run_gradient(const DeviceTables &T, const double *psi, double *psdx, double *psdy, int nfields,
                        hipStream_t s) {
    const long total = static_cast<long>(nfields) * NSPEC;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(gradient_kernel, dim3(blocks_for(total)), dim3(kOpThreads), 0, s,
                       reinterpret_cast<const d2 *>(psi), reinterpret_cast<d2 *>(psdx), reinterpret_cast<d2 *>(psdy), T,
                       total);
    return hipGetLastError();
}

hipError_t run_scale(const double *in, double *out, const double *table, double sign, int nfields, hipStream_t s) {
    const long total = static_cast<long>(nfields) * NSPEC;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(scale_kernel, dim3(blocks_for(total)), dim3(kOpThreads), 0, s, reinterpret_cast<const d2 *>(in),
                       reinterpret_cast<d2 *>(out), table, sign, total);
    return hipGetLastError();
}

}  // namespace spd
