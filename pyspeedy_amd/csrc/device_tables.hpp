// Device-resident constant tables, passed by value to every kernel (all pointers are device memory owned by
// the spd_context).  Layouts are chosen for the kernels, not for the reference: see the comments.
#pragma once
#include "tables.hpp"

namespace spd {

struct DeviceTables {
    // inverse-Legendre polynomials  [n=32][m*12+jq][2 lat pairs]      (zero where m + n > 31)
    const double *pinv;
    // direct-Legendre polynomials   [j=24][lane (dir_stride)][2 n]    one lane per two valid coefficients of one
    // (m, parity); dirmeta[lane] = {pos_re | pos_im << 8, parity, output index a, output index b or -1}
    const double *pdir;
    const int *dirmeta;
    int ndir, dir_stride;
    const double *work;    // FFTPACK twiddles, 96
    const double *cosgr;   // 48
    const double *cosgr2;  // 48
    const double *wt;      // 24 Gaussian weights
    double fft_scale;      // fp32(1/96) widened (fourier.f90:113)
    // spectral-operator coefficients, (31,32) each
    const double *el2, *elm2, *trfilt, *gradx, *gradym, *gradyp, *uvdx, *uvdym, *uvdyp, *vddym, *vddyp;
    // physics
    const double *fband;   // (301,4)
    const double *coa;     // 48, cos(latitude)
    const float *fband32, *coa32;  // the same two tables rounded to fp32, for the mixed-precision physics (cfg 5)
    double fsg[8], dhs[8], sigl[8], sigh[9], grdsig[8], grdscp[8], wvi[16];
};

// One transform of a descriptor-table launch: a spectral->grid entry carries kcos in `flag`, a grid->spectral entry the
// pre-scale mode (0 none, 1 cosgr, 2 cosgr2).  Spectral->grid entries can ask for the spectral-space operator that feeds the
// transform to be applied while the coefficients are staged (`mode`), so that u, v and grad ln ps never exist as spectral
// arrays in memory: 0 = transform src as it is; 1 / 2 = ucos / vcos of vort2vel(vor = src, div = src2)
// (spectral.f90:190-214); 3 / 4 = x / y component of gradient(src) (spectral.f90:275-296).
// flag of a spectral -> grid entry: kcos (1: none, 2: rows times 1 / cos(lat)) [| kGridAsFloat: dst receives the field as
// fp32, 96 x 48 floats from dst on -- fields that only the fp32 column physics of cfg 5 reads].  grid -> spectral: the prescale.
constexpr int kGridAsFloat = 0x100;
struct FieldDesc {
    const double *src;
    double *dst;
    int flag;
    int mode;
    const double *src2;
};

}  // namespace spd
