// Real FFT of length 96 for gfx950, organised for many rows at once.
//
// The reference uses FFTPACK (speedy.f90/fftpack.f90) with the factor order 2,4,4,3 and a twiddle table built
// from an fp32 value of 2*pi (rffti1, fftpack.f90:39).  Because those twiddles are not exact roots of unity,
// matching the reference to 1e-13 requires the same pass structure with each table entry in the same role:
//   backward (rfftb1, :69-134): radb2(ido=48,l1=1) -> radb4(ido=12,l1=2) -> radb4(ido=3,l1=8) -> radb3(ido=1,l1=32)
//   forward  (rfftf1, :136-202): radf3(ido=1,l1=32) -> radf4(ido=3,l1=8) -> radf4(ido=12,l1=2) -> radf2(ido=48,l1=1)
//
// MI355X mapping.  The four passes are regrouped into two register-resident stages with one LDS transposition
// between them, so that one lane never holds more than 16 values:
//   "block" stage  = the ido=3 radix-4 pass + the ido=1 radix-3 pass: 8 independent blocks of 12 values per row
//   "group" stage  = the ido=12 radix-4 pass + the ido=48 radix-2 pass: 5 general groups of 16 values (pass index
//                    i = 3,5,..,11) + the i=1 group + the i=12 group (8 values each)
// Lanes of a wavefront work on different rows (row stride in LDS is odd, so ds_read/ds_write_b64 are
// bank-conflict free), and a (row, block|group) task list is spread over the whole workgroup.
//
// Indices below are 0-based positions in a 96-element row; W[] is the reference's `work` table (0-based).
#pragma once
#include <hip/hip_runtime.h>

namespace spd {
namespace fft {

// radix constants exactly as the reference forms them in fp32 (fftpack.f90:268-269, 341, 786-787, 857)
__device__ constexpr double kTaur = -0.5;
__device__ constexpr double kTaui = 0.866025388240814209;   // .5*sqrt(3.) in fp32
__device__ constexpr double kSqrt2 = 1.414213538169860840;  // sqrt(2.) in fp32
__device__ constexpr double kHsqt2 = 0.707106769084930420;  // .5*sqrt(2.) in fp32
constexpr int kNumGroups = 7;   // group ids 0..4 -> i = 3+2g ; 5 -> i = 1 ; 6 -> i = 12
constexpr int kNumBlocks = 8;
constexpr int kInvLast = 60;    // highest non-zero input position of the inverse transform (wavenumber 30)

struct Pair { double r, i; };

// ============================== backward ==============================
// radb2 general butterfly i1 (odd, 3..47): returns the pair at positions (i1-2, i1-1) for both output halves.
// `in` is the unpacked Fourier row; positions > kInvLast are structurally zero (fourier.f90:79-81).
template <bool ZeroPad>
__device__ inline void radb2_pair(const double *in, const double *W, int i1, Pair &h1, Pair &h2) {
    const double ar = in[i1 - 2], ai = in[i1 - 1];
    const int q = 96 - i1;  // mirror pair (q, q+1) in the second input half
    double br = in[q], bi = in[q + 1];
    if (ZeroPad) {
        br = (q > kInvLast) ? 0.0 : br;
        bi = (q + 1 > kInvLast) ? 0.0 : bi;
    }
    h1.r = ar + br;
    const double tr2 = ar - br;
    h1.i = ai - bi;
    const double ti2 = ai + bi;
    const double wr = W[i1 - 3], wi = W[i1 - 2];
    h2.r = wr * tr2 - wi * ti2;
    h2.i = wr * ti2 + wi * tr2;
}

// group stage, general group with pass index i (odd, 3..11): 16 outputs.
template <bool ZeroPad>
__device__ inline void bwd_group_general(const double *in, double *out, const double *W, int i) {
    Pair a1, a2, b1, b2, c1, c2, d1, d2;
    radb2_pair<ZeroPad>(in, W, i, a1, a2);        // feeds cc(i-1:i , 1, k)
    radb2_pair<ZeroPad>(in, W, 26 - i, b1, b2);   // feeds cc(ic-1:ic, 2, k)
    radb2_pair<ZeroPad>(in, W, i + 24, c1, c2);   // feeds cc(i-1:i , 3, k)
    radb2_pair<ZeroPad>(in, W, 50 - i, d1, d2);   // feeds cc(ic-1:ic, 4, k)
    const double w1r = W[48 + i - 3], w1i = W[48 + i - 2];
    const double w2r = W[60 + i - 3], w2i = W[60 + i - 2];
    const double w3r = W[72 + i - 3], w3i = W[72 + i - 2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // radb4 general butterfly (i, k+1), fftpack.f90:361-386
        const Pair &A = k ? a2 : a1, &B = k ? b2 : b1, &C = k ? c2 : c1, &D = k ? d2 : d1;
        const double ti1 = A.i + D.i, ti2 = A.i - D.i, ti3 = C.i - B.i, tr4 = C.i + B.i;
        const double tr1 = A.r - D.r, tr2 = A.r + D.r, ti4 = C.r - B.r, tr3 = C.r + B.r;
        const double cr3 = tr2 - tr3, ci3 = ti2 - ti3;
        const double cr2 = tr1 - tr4, cr4 = tr1 + tr4, ci2 = ti1 + ti4, ci4 = ti1 - ti4;
        double *o = out + (i - 2) + 12 * k;
        o[0] = tr2 + tr3;
        o[1] = ti2 + ti3;
        o[24] = w1r * cr2 - w1i * ci2;
        o[25] = w1r * ci2 + w1i * cr2;
        o[48] = w2r * cr3 - w2i * ci3;
        o[49] = w2r * ci3 + w2i * cr3;
        o[72] = w3r * cr4 - w3i * ci4;
        o[73] = w3r * ci4 + w3i * cr4;
    }
}

// group i = 1: radb2 first + last butterflies and the pair i1 = 25, then radb4 "first" for k = 1, 2.
template <bool ZeroPad>
__device__ inline void bwd_group_first(const double *in, double *out, const double *W) {
    const double c95 = ZeroPad ? 0.0 : in[95], c48 = in[48];
    const double p0[2] = {in[0] + c95, in[0] - c95};               // ch(1,1,1:2)      fftpack.f90:217-218
    const double p47[2] = {in[47] + in[47], -(c48 + c48)};         // ch(48,1,1:2)     :250-251
    Pair m1, m2;
    radb2_pair<ZeroPad>(in, W, 25, m1, m2);                         // positions 23, 24
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // radb4 i = 1, fftpack.f90:343-352
        const Pair &M = k ? m2 : m1;
        const double tr1 = p0[k] - p47[k], tr2 = p0[k] + p47[k];
        const double tr3 = M.r + M.r, tr4 = M.i + M.i;             // cc(12,2,k)=pos 23 ; cc(1,3,k)=pos 24
        double *o = out + 12 * k;
        o[0] = tr2 + tr3;
        o[24] = tr1 - tr4;
        o[48] = tr2 - tr3;
        o[72] = tr1 + tr4;
    }
}

// group i = 12 (ido even): pairs i1 = 13 and 37, then radb4 "last" for k = 1, 2.
template <bool ZeroPad>
__device__ inline void bwd_group_last(const double *in, double *out, const double *W) {
    Pair a1, a2, b1, b2;
    radb2_pair<ZeroPad>(in, W, 13, a1, a2);  // positions 11, 12
    radb2_pair<ZeroPad>(in, W, 37, b1, b2);  // positions 35, 36
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // fftpack.f90:413-422
        const Pair &A = k ? a2 : a1, &B = k ? b2 : b1;
        const double ti1 = A.i + B.i, ti2 = B.i - A.i;  // cc(1,2,k)=pos 12 ; cc(1,4,k)=pos 36
        const double tr1 = A.r - B.r, tr2 = A.r + B.r;  // cc(12,1,k)=pos 11 ; cc(12,3,k)=pos 35
        double *o = out + 11 + 12 * k;
        o[0] = tr2 + tr2;
        o[24] = kSqrt2 * (tr1 - ti1);
        o[48] = ti2 + ti2;
        o[72] = -kSqrt2 * (tr1 + ti1);
    }
}

template <bool ZeroPad>
__device__ inline void bwd_group(const double *in, double *out, const double *W, int g) {
    if (g < 5)
        bwd_group_general<ZeroPad>(in, out, W, 3 + 2 * g);
    else if (g == 5)
        bwd_group_first<ZeroPad>(in, out, W);
    else
        bwd_group_last<ZeroPad>(in, out, W);
}

// block stage: radb4(ido=3, l1=8) butterfly k = kk+1, then four radb3(ido=1) butterflies k' = kk + 8*j.
__device__ inline void bwd_block(const double *in, double *out, const double *W, int kk) {
    double x[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) x[j][i] = in[12 * kk + 3 * j + i];
    double y[4][3];
    {  // i = 1
        const double tr1 = x[0][0] - x[3][2], tr2 = x[0][0] + x[3][2];
        const double tr3 = x[1][2] + x[1][2], tr4 = x[2][0] + x[2][0];
        y[0][0] = tr2 + tr3;
        y[1][0] = tr1 - tr4;
        y[2][0] = tr2 - tr3;
        y[3][0] = tr1 + tr4;
    }
    {  // i = 3, ic = 2
        const double ti1 = x[0][2] + x[3][1], ti2 = x[0][2] - x[3][1], ti3 = x[2][2] - x[1][1], tr4 = x[2][2] + x[1][1];
        const double tr1 = x[0][1] - x[3][0], tr2 = x[0][1] + x[3][0], ti4 = x[2][1] - x[1][0], tr3 = x[2][1] + x[1][0];
        const double cr3 = tr2 - tr3, ci3 = ti2 - ti3;
        const double cr2 = tr1 - tr4, cr4 = tr1 + tr4, ci2 = ti1 + ti4, ci4 = ti1 - ti4;
        y[0][1] = tr2 + tr3;
        y[0][2] = ti2 + ti3;
        y[1][1] = W[84] * cr2 - W[85] * ci2;
        y[1][2] = W[84] * ci2 + W[85] * cr2;
        y[2][1] = W[87] * cr3 - W[88] * ci3;
        y[2][2] = W[87] * ci3 + W[88] * cr3;
        y[3][1] = W[90] * cr4 - W[91] * ci4;
        y[3][2] = W[90] * ci4 + W[91] * cr4;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // radb3, ido = 1, fftpack.f90:270-277
        const double tr2 = y[j][1] + y[j][1];
        const double cr2 = y[j][0] + kTaur * tr2;
        const double ci3 = kTaui * (y[j][2] + y[j][2]);
        double *o = out + kk + 8 * j;
        o[0] = y[j][0] + tr2;
        o[32] = cr2 - ci3;
        o[64] = cr2 + ci3;
    }
}

// ============================== forward ==============================
// block stage: four radf3(ido=1) butterflies k' = kk + 8*j, then radf4(ido=3, l1=8) butterfly k = kk+1.
__device__ inline void fwd_block(const double *in, double *out, const double *W, int kk) {
    double y[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // fftpack.f90:788-793
        const double *c = in + kk + 8 * j;
        const double c1 = c[0], c2 = c[32], c3 = c[64];
        const double cr2 = c2 + c3;
        y[j][0] = c1 + cr2;
        y[j][2] = kTaui * (c3 - c2);
        y[j][1] = c1 + kTaur * cr2;
    }
    double *o = out + 12 * kk;  // ch(i, j, k) -> o[(i-1) + 3*(j-1)]
    {  // i = 1, fftpack.f90:858-865
        const double tr1 = y[1][0] + y[3][0], tr2 = y[0][0] + y[2][0];
        o[0] = tr1 + tr2;
        o[11] = tr2 - tr1;           // ch(3,4,k)
        o[5] = y[0][0] - y[2][0];    // ch(3,2,k)
        o[6] = y[3][0] - y[1][0];    // ch(1,3,k)
    }
    {  // i = 3, ic = 2, fftpack.f90:871-895
        const double cr2 = W[84] * y[1][1] + W[85] * y[1][2], ci2 = W[84] * y[1][2] - W[85] * y[1][1];
        const double cr3 = W[87] * y[2][1] + W[88] * y[2][2], ci3 = W[87] * y[2][2] - W[88] * y[2][1];
        const double cr4 = W[90] * y[3][1] + W[91] * y[3][2], ci4 = W[90] * y[3][2] - W[91] * y[3][1];
        const double tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
        const double ti2 = y[0][2] + ci3, ti3 = y[0][2] - ci3, tr2 = y[0][1] + cr3, tr3 = y[0][1] - cr3;
        o[1] = tr1 + tr2;   // ch(2,1,k)
        o[9] = tr2 - tr1;   // ch(1,4,k)
        o[2] = ti1 + ti2;   // ch(3,1,k)
        o[10] = ti1 - ti2;  // ch(2,4,k)
        o[7] = ti4 + tr3;   // ch(2,3,k)
        o[3] = tr3 - ti4;   // ch(1,2,k)
        o[8] = tr4 + ti3;   // ch(3,3,k)
        o[4] = tr4 - ti3;   // ch(2,2,k)
    }
}

// radf2 general butterfly i1: inputs are the pairs at positions (i1-2, i1-1) of both halves; writes the final
// coefficients at (i1-2, i1-1) and, when they are among the retained wavenumbers, at (96-i1, 97-i1).
// Everything is multiplied by `scale` on the way out (fourier.f90:113-121).
__device__ inline void radf2_pair(const Pair &h1, const Pair &h2, const double *W, int i1, double *out, double scale) {
    const double wr = W[i1 - 3], wi = W[i1 - 2];
    const double tr2 = wr * h2.r + wi * h2.i;
    const double ti2 = wr * h2.i - wi * h2.r;
    out[i1 - 1] = (h1.i + ti2) * scale;
    out[i1 - 2] = (h1.r + tr2) * scale;
    if (97 - i1 <= kInvLast) {
        out[97 - i1] = (ti2 - h1.i) * scale;
        out[96 - i1] = (h1.r - tr2) * scale;
    }
}

// group stage, general group i (odd, 3..11): radf4(ido=12,l1=2) butterflies (i,1),(i,2) then four radf2 pairs.
__device__ inline void fwd_group_general(const double *in, double *out, const double *W, int i, double scale) {
    const double w1r = W[48 + i - 3], w1i = W[48 + i - 2];
    const double w2r = W[60 + i - 3], w2i = W[60 + i - 2];
    const double w3r = W[72 + i - 3], w3i = W[72 + i - 2];
    Pair a[2], b[2], c[2], d[2];  // pairs at i1 = i, 26-i, i+24, 50-i for halves k = 1, 2
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // fftpack.f90:871-895
        const double *x = in + (i - 2) + 12 * k;
        const double x1r = x[0], x1i = x[1], x2r = x[24], x2i = x[25], x3r = x[48], x3i = x[49], x4r = x[72], x4i = x[73];
        const double cr2 = w1r * x2r + w1i * x2i, ci2 = w1r * x2i - w1i * x2r;
        const double cr3 = w2r * x3r + w2i * x3i, ci3 = w2r * x3i - w2i * x3r;
        const double cr4 = w3r * x4r + w3i * x4i, ci4 = w3r * x4i - w3i * x4r;
        const double tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
        const double ti2 = x1i + ci3, ti3 = x1i - ci3, tr2 = x1r + cr3, tr3 = x1r - cr3;
        a[k].r = tr1 + tr2;  a[k].i = ti1 + ti2;   // ch(i-1:i , 1, k)
        d[k].r = tr2 - tr1;  d[k].i = ti1 - ti2;   // ch(ic-1:ic, 4, k)
        c[k].r = ti4 + tr3;  c[k].i = tr4 + ti3;   // ch(i-1:i , 3, k)
        b[k].r = tr3 - ti4;  b[k].i = tr4 - ti3;   // ch(ic-1:ic, 2, k)
    }
    radf2_pair(a[0], a[1], W, i, out, scale);
    radf2_pair(b[0], b[1], W, 26 - i, out, scale);
    radf2_pair(c[0], c[1], W, i + 24, out, scale);
    radf2_pair(d[0], d[1], W, 50 - i, out, scale);
}

// group i = 1
__device__ inline void fwd_group_first(const double *in, double *out, const double *W, double scale) {
    double p0[2], p47[2];
    Pair m[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // radf4 i = 1, fftpack.f90:858-865 ; cc(1,k,j) = in[12k + 24(j-1)]
        const double *x = in + 12 * k;
        const double tr1 = x[24] + x[72], tr2 = x[0] + x[48];
        p0[k] = tr1 + tr2;       // pos 0
        p47[k] = tr2 - tr1;      // pos 47
        m[k].r = x[0] - x[48];   // pos 23
        m[k].i = x[72] - x[24];  // pos 24
    }
    out[0] = (p0[0] + p0[1]) * scale;  // radf2 first, :733-734 (the Nyquist term at pos 95 is not retained)
    out[48] = (-p47[1]) * scale;       // radf2 last, :766-767
    out[47] = p47[0] * scale;
    radf2_pair(m[0], m[1], W, 25, out, scale);
}

// group i = 12
__device__ inline void fwd_group_last(const double *in, double *out, const double *W, double scale) {
    Pair a[2], b[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // radf4 last, fftpack.f90:934-941 ; cc(12,k,j) = in[11 + 12k + 24(j-1)]
        const double *x = in + 11 + 12 * k;
        const double ti1 = -kHsqt2 * (x[24] + x[72]);
        const double tr1 = kHsqt2 * (x[24] - x[72]);
        a[k].r = tr1 + x[0];    // pos 11
        b[k].r = x[0] - tr1;    // pos 35
        a[k].i = ti1 - x[48];   // pos 12
        b[k].i = ti1 + x[48];   // pos 36
    }
    radf2_pair(a[0], a[1], W, 13, out, scale);
    radf2_pair(b[0], b[1], W, 37, out, scale);
}

__device__ inline void fwd_group(const double *in, double *out, const double *W, int g, double scale) {
    if (g < 5)
        fwd_group_general(in, out, W, 3 + 2 * g, scale);
    else if (g == 5)
        fwd_group_first(in, out, W, scale);
    else
        fwd_group_last(in, out, W, scale);
}


// ======================================================================================================
// Register-output variants: a task loads all of its inputs, computes, and hands its outputs back in registers,
// so that a whole stage can run IN PLACE on one row buffer (load -> workgroup barrier -> store).
// ======================================================================================================

// ---- backward, group stage.  o[k][0..7] belongs at base + {0,1,24,25,48,49,72,73}, base = (i-2) + 12k ----
template <bool ZeroPad>
__device__ inline void bwd_group_general_r(const double *in, const double *W, int i, double (&o)[2][8]) {
    Pair a1, a2, b1, b2, c1, c2, d1, d2;
    radb2_pair<ZeroPad>(in, W, i, a1, a2);
    radb2_pair<ZeroPad>(in, W, 26 - i, b1, b2);
    radb2_pair<ZeroPad>(in, W, i + 24, c1, c2);
    radb2_pair<ZeroPad>(in, W, 50 - i, d1, d2);
    const double w1r = W[48 + i - 3], w1i = W[48 + i - 2];
    const double w2r = W[60 + i - 3], w2i = W[60 + i - 2];
    const double w3r = W[72 + i - 3], w3i = W[72 + i - 2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const Pair &A = k ? a2 : a1, &B = k ? b2 : b1, &C = k ? c2 : c1, &D = k ? d2 : d1;
        const double ti1 = A.i + D.i, ti2 = A.i - D.i, ti3 = C.i - B.i, tr4 = C.i + B.i;
        const double tr1 = A.r - D.r, tr2 = A.r + D.r, ti4 = C.r - B.r, tr3 = C.r + B.r;
        const double cr3 = tr2 - tr3, ci3 = ti2 - ti3;
        const double cr2 = tr1 - tr4, cr4 = tr1 + tr4, ci2 = ti1 + ti4, ci4 = ti1 - ti4;
        o[k][0] = tr2 + tr3;
        o[k][1] = ti2 + ti3;
        o[k][2] = w1r * cr2 - w1i * ci2;
        o[k][3] = w1r * ci2 + w1i * cr2;
        o[k][4] = w2r * cr3 - w2i * ci3;
        o[k][5] = w2r * ci3 + w2i * cr3;
        o[k][6] = w3r * cr4 - w3i * ci4;
        o[k][7] = w3r * ci4 + w3i * cr4;
    }
}

__device__ inline void group_general_store(double *row, int i, const double (&o)[2][8]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double *b = row + (i - 2) + 12 * k;
        b[0] = o[k][0]; b[1] = o[k][1]; b[24] = o[k][2]; b[25] = o[k][3];
        b[48] = o[k][4]; b[49] = o[k][5]; b[72] = o[k][6]; b[73] = o[k][7];
    }
}

// first / last groups: o[k][0..3] belongs at base + {0,24,48,72}, base = 12k (first) or 11 + 12k (last)
template <bool ZeroPad>
__device__ inline void bwd_group_first_r(const double *in, const double *W, double (&o)[2][8]) {
    const double c95 = ZeroPad ? 0.0 : in[95], c48 = in[48];
    const double p0[2] = {in[0] + c95, in[0] - c95};
    const double p47[2] = {in[47] + in[47], -(c48 + c48)};
    Pair m1, m2;
    radb2_pair<ZeroPad>(in, W, 25, m1, m2);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const Pair &M = k ? m2 : m1;
        const double tr1 = p0[k] - p47[k], tr2 = p0[k] + p47[k];
        const double tr3 = M.r + M.r, tr4 = M.i + M.i;
        o[k][0] = tr2 + tr3;
        o[k][1] = tr1 - tr4;
        o[k][2] = tr2 - tr3;
        o[k][3] = tr1 + tr4;
    }
}

template <bool ZeroPad>
__device__ inline void bwd_group_last_r(const double *in, const double *W, double (&o)[2][8]) {
    Pair a1, a2, b1, b2;
    radb2_pair<ZeroPad>(in, W, 13, a1, a2);
    radb2_pair<ZeroPad>(in, W, 37, b1, b2);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const Pair &A = k ? a2 : a1, &B = k ? b2 : b1;
        const double ti1 = A.i + B.i, ti2 = B.i - A.i;
        const double tr1 = A.r - B.r, tr2 = A.r + B.r;
        o[k][0] = tr2 + tr2;
        o[k][1] = kSqrt2 * (tr1 - ti1);
        o[k][2] = ti2 + ti2;
        o[k][3] = -kSqrt2 * (tr1 + ti1);
    }
}

__device__ inline void group_edge_store(double *row, int base, const double (&o)[2][8]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double *b = row + base + 12 * k;
        b[0] = o[k][0]; b[24] = o[k][1]; b[48] = o[k][2]; b[72] = o[k][3];
    }
}

// ---- backward, block stage: o[j][s] belongs at kk + 8j + 32s ----
__device__ inline void bwd_block_r(const double *in, const double *W, int kk, double (&o)[4][3]) {
    double x[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) x[j][i] = in[12 * kk + 3 * j + i];
    double y[4][3];
    {
        const double tr1 = x[0][0] - x[3][2], tr2 = x[0][0] + x[3][2];
        const double tr3 = x[1][2] + x[1][2], tr4 = x[2][0] + x[2][0];
        y[0][0] = tr2 + tr3; y[1][0] = tr1 - tr4; y[2][0] = tr2 - tr3; y[3][0] = tr1 + tr4;
    }
    {
        const double ti1 = x[0][2] + x[3][1], ti2 = x[0][2] - x[3][1], ti3 = x[2][2] - x[1][1], tr4 = x[2][2] + x[1][1];
        const double tr1 = x[0][1] - x[3][0], tr2 = x[0][1] + x[3][0], ti4 = x[2][1] - x[1][0], tr3 = x[2][1] + x[1][0];
        const double cr3 = tr2 - tr3, ci3 = ti2 - ti3;
        const double cr2 = tr1 - tr4, cr4 = tr1 + tr4, ci2 = ti1 + ti4, ci4 = ti1 - ti4;
        y[0][1] = tr2 + tr3;
        y[0][2] = ti2 + ti3;
        y[1][1] = W[84] * cr2 - W[85] * ci2;
        y[1][2] = W[84] * ci2 + W[85] * cr2;
        y[2][1] = W[87] * cr3 - W[88] * ci3;
        y[2][2] = W[87] * ci3 + W[88] * cr3;
        y[3][1] = W[90] * cr4 - W[91] * ci4;
        y[3][2] = W[90] * ci4 + W[91] * cr4;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double tr2 = y[j][1] + y[j][1];
        const double cr2 = y[j][0] + kTaur * tr2;
        const double ci3 = kTaui * (y[j][2] + y[j][2]);
        o[j][0] = y[j][0] + tr2;
        o[j][1] = cr2 - ci3;
        o[j][2] = cr2 + ci3;
    }
}

// ---- forward, block stage: reads kk + 8j + 32s, o[0..11] belongs at 12kk + 0..11 ----
__device__ inline void fwd_block_r(const double *in, const double *W, int kk, double (&o)[12]) {
    double y[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double *c = in + kk + 8 * j;
        const double c1 = c[0], c2 = c[32], c3 = c[64];
        const double cr2 = c2 + c3;
        y[j][0] = c1 + cr2;
        y[j][2] = kTaui * (c3 - c2);
        y[j][1] = c1 + kTaur * cr2;
    }
    {
        const double tr1 = y[1][0] + y[3][0], tr2 = y[0][0] + y[2][0];
        o[0] = tr1 + tr2;
        o[11] = tr2 - tr1;
        o[5] = y[0][0] - y[2][0];
        o[6] = y[3][0] - y[1][0];
    }
    {
        const double cr2 = W[84] * y[1][1] + W[85] * y[1][2], ci2 = W[84] * y[1][2] - W[85] * y[1][1];
        const double cr3 = W[87] * y[2][1] + W[88] * y[2][2], ci3 = W[87] * y[2][2] - W[88] * y[2][1];
        const double cr4 = W[90] * y[3][1] + W[91] * y[3][2], ci4 = W[90] * y[3][2] - W[91] * y[3][1];
        const double tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
        const double ti2 = y[0][2] + ci3, ti3 = y[0][2] - ci3, tr2 = y[0][1] + cr3, tr3 = y[0][1] - cr3;
        o[1] = tr1 + tr2; o[9] = tr2 - tr1; o[2] = ti1 + ti2; o[10] = ti1 - ti2;
        o[7] = ti4 + tr3; o[3] = tr3 - ti4; o[8] = tr4 + ti3; o[4] = tr4 - ti3;
    }
}

// ---- forward, group stage.  radf2 pair -> 4 values: positions (i1-2, i1-1) and mirrored (96-i1, 97-i1) ----
struct Quad { double lo_r, lo_i, hi_r, hi_i; };

__device__ inline Quad radf2_pair_r(const Pair &h1, const Pair &h2, const double *W, int i1, double scale) {
    const double wr = W[i1 - 3], wi = W[i1 - 2];
    const double tr2 = wr * h2.r + wi * h2.i;
    const double ti2 = wr * h2.i - wi * h2.r;
    Quad q;
    q.lo_i = (h1.i + ti2) * scale;   // pos i1-1
    q.lo_r = (h1.r + tr2) * scale;   // pos i1-2
    q.hi_i = (ti2 - h1.i) * scale;   // pos 97-i1
    q.hi_r = (h1.r - tr2) * scale;   // pos 96-i1
    return q;
}

// general group i: returns the quads of i1 = i, 26-i, i+24, 50-i (only the last one's mirrored pair is retained)
__device__ inline void fwd_group_general_r(const double *in, const double *W, int i, double scale, Quad (&q)[4]) {
    const double w1r = W[48 + i - 3], w1i = W[48 + i - 2];
    const double w2r = W[60 + i - 3], w2i = W[60 + i - 2];
    const double w3r = W[72 + i - 3], w3i = W[72 + i - 2];
    Pair a[2], b[2], c[2], d[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double *x = in + (i - 2) + 12 * k;
        const double x1r = x[0], x1i = x[1], x2r = x[24], x2i = x[25], x3r = x[48], x3i = x[49], x4r = x[72], x4i = x[73];
        const double cr2 = w1r * x2r + w1i * x2i, ci2 = w1r * x2i - w1i * x2r;
        const double cr3 = w2r * x3r + w2i * x3i, ci3 = w2r * x3i - w2i * x3r;
        const double cr4 = w3r * x4r + w3i * x4i, ci4 = w3r * x4i - w3i * x4r;
        const double tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
        const double ti2 = x1i + ci3, ti3 = x1i - ci3, tr2 = x1r + cr3, tr3 = x1r - cr3;
        a[k].r = tr1 + tr2;  a[k].i = ti1 + ti2;
        d[k].r = tr2 - tr1;  d[k].i = ti1 - ti2;
        c[k].r = ti4 + tr3;  c[k].i = tr4 + ti3;
        b[k].r = tr3 - ti4;  b[k].i = tr4 - ti3;
    }
    q[0] = radf2_pair_r(a[0], a[1], W, i, scale);
    q[1] = radf2_pair_r(b[0], b[1], W, 26 - i, scale);
    q[2] = radf2_pair_r(c[0], c[1], W, i + 24, scale);
    q[3] = radf2_pair_r(d[0], d[1], W, 50 - i, scale);
}

__device__ inline void fwd_group_general_store(double *row, int i, const Quad (&q)[4]) {
    row[i - 2] = q[0].lo_r;       row[i - 1] = q[0].lo_i;
    row[24 - i] = q[1].lo_r;      row[25 - i] = q[1].lo_i;
    row[i + 22] = q[2].lo_r;      row[i + 23] = q[2].lo_i;
    row[48 - i] = q[3].lo_r;      row[49 - i] = q[3].lo_i;
    row[46 + i] = q[3].hi_r;      row[47 + i] = q[3].hi_i;   // mirrored pair of i1 = 50-i: 96-i1 = 46+i <= 57
}

// first group: o = {pos0, pos47, pos48, pos23, pos24}
__device__ inline void fwd_group_first_r(const double *in, const double *W, double scale, double (&o)[5]) {
    double p0[2], p47[2];
    Pair m[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double *x = in + 12 * k;
        const double tr1 = x[24] + x[72], tr2 = x[0] + x[48];
        p0[k] = tr1 + tr2;
        p47[k] = tr2 - tr1;
        m[k].r = x[0] - x[48];
        m[k].i = x[72] - x[24];
    }
    o[0] = (p0[0] + p0[1]) * scale;
    o[1] = p47[0] * scale;
    o[2] = (-p47[1]) * scale;
    const Quad q = radf2_pair_r(m[0], m[1], W, 25, scale);
    o[3] = q.lo_r;
    o[4] = q.lo_i;
}

// last group: quads of i1 = 13 (mirrored pair discarded) and i1 = 37 (mirrored pair = positions 59, 60)
__device__ inline void fwd_group_last_r(const double *in, const double *W, double scale, Quad (&q)[4]) {
    Pair a[2], b[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double *x = in + 11 + 12 * k;
        const double ti1 = -kHsqt2 * (x[24] + x[72]);
        const double tr1 = kHsqt2 * (x[24] - x[72]);
        a[k].r = tr1 + x[0];
        b[k].r = x[0] - tr1;
        a[k].i = ti1 - x[48];
        b[k].i = ti1 + x[48];
    }
    q[0] = radf2_pair_r(a[0], a[1], W, 13, scale);
    q[1] = radf2_pair_r(b[0], b[1], W, 37, scale);
}

}  // namespace fft
}  // namespace spd
