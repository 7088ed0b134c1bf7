/* TEST INFRASTRUCTURE -- CPU oracle (see speedy_oracle.h).  NOT PART OF THE PRODUCT.
 *
 * The callers of the hot path ("next #1" of SURVEY.md section 8f), restated in plain C: grid-point and spectral
 * tendencies (tendencies.f90), the semi-implicit correction and its tables (implicit.f90, matrix_inversion.f90),
 * horizontal diffusion (horizontal_diffusion.f90), geopotential (geopotential.f90), leapfrog step with the
 * Robert-Asselin-Williams filter (time_stepping.f90) and the range check (diagnostics.f90).
 * Arrays are Fortran column-major; complex values are interleaved (re, im) doubles.
 */
#include "speedy_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IX ORC_IX
#define IL ORC_IL
#define KX ORC_KX
#define MX ORC_MX
#define NX ORC_NX
#define TRUNC ORC_TRUNC
#define NG (IX * IL)
#define NS (MX * NX) /* complex coefficients per field */

static const double REARTH = 6.371e+6f, GRAV = 9.81f, CP = 1004.0f;
#define AKAP ((double)(2.0f / 7.0f))
#define RGAS (AKAP * CP)
static const double GAMMA = 6.0f, HSCALE = 7.5f, HSHUM = 2.5f, THD = 2.4f, THDD = 2.4f, THDS = 12.0f,
                    TDRS = 24.0f * 30.0f;
static const double ROB = 0.05f, WIL = 0.53f, ALPH = 0.5f;

/* ------------------------------------------------------------------ horizontal_diffusion.f90:50-110 */
static void init_hdiff(const orc_tables *t, orc_dyn_tables *d) {
    const double hdiff = 1.f / (THD * 3600.f), hdifd = 1.f / (THDD * 3600.f), hdifs = 1.f / (THDS * 3600.f);
    const double rlap = (double)(1.f / (float)(TRUNC * (TRUNC + 1)));
    for (int j = 1; j <= NX; ++j)
        for (int k = 1; k <= MX; ++k) {
            double twn = (double)(float)(k + j - 2);
            double elap = (twn * (twn + 1.f) * rlap);
            double elapn = ((elap * elap) * elap) * elap; /* elap**4: flang expands the integer power sequentially */
            int idx = (k - 1) + MX * (j - 1);
            d->dmp[idx] = hdiff * elapn;
            d->dmpd[idx] = hdifd * elapn;
            d->dmps[idx] = hdifs * elap;
        }
    const double rgam = RGAS * GAMMA / (1000.f * GRAV);
    const double qexp = HSCALE / HSHUM;
    d->tcorv[0] = 0.;
    d->qcorv[0] = 0.;
    d->qcorv[1] = 0.;
    for (int k = 2; k <= KX; ++k) {
        d->tcorv[k - 1] = pow(t->fsg[k - 1], rgam);
        if (k > 2) d->qcorv[k - 1] = pow(t->fsg[k - 1], qexp);
    }
}

/* ------------------------------------------------------------------ implicit.f90:44-80 and geopotential.f90:16-31 */
void orc_dyn_tables_init(const orc_tables *t, orc_dyn_tables *d) {
    memset(d, 0, sizeof *d);
    init_hdiff(t, d);
    const double rgam = RGAS * GAMMA / (1000.f * GRAV);
    for (int k = 0; k < KX; ++k) {
        double f = t->fsg[k] > 0.2f ? t->fsg[k] : (double)0.2f;
        d->tref[k] = 288.f * pow(f, rgam);
        d->tref2[k] = AKAP * d->tref[k];
        d->tref3[k] = t->fsgr[k] * d->tref[k];
    }
    for (int k = 1; k <= KX; ++k) {
        d->xgeop1[k - 1] = RGAS * log(t->hsg[k] / t->fsg[k - 1]);
        if (k != KX) d->xgeop2[k] = RGAS * log(t->fsg[k] / t->hsg[k]);
    }
}

/* ------------------------------------------------------------------ matrix_inversion.f90 (Numerical-Recipes LU) */
static void ludcmp(double *a, int n, int *indx) { /* a(n,n) column-major */
#define A(i, j) a[((i)-1) + n * ((j)-1)]
    const double tiny = 1.0e-20f;
    double vv[100];
    int imax = 0;
    for (int i = 1; i <= n; ++i) {
        double aamax = 0.;
        for (int j = 1; j <= n; ++j)
            if (fabs(A(i, j)) > aamax) aamax = fabs(A(i, j));
        vv[i - 1] = 1. / aamax;
    }
    for (int j = 1; j <= n; ++j) {
        for (int i = 1; i <= j - 1; ++i) {
            double sum = A(i, j);
            if (i > 1) {
                for (int k = 1; k <= i - 1; ++k) sum = sum - A(i, k) * A(k, j);
                A(i, j) = sum;
            }
        }
        double aamax = 0.;
        for (int i = j; i <= n; ++i) {
            double sum = A(i, j);
            if (j > 1) {
                for (int k = 1; k <= j - 1; ++k) sum = sum - A(i, k) * A(k, j);
                A(i, j) = sum;
            }
            double dum = vv[i - 1] * fabs(sum);
            if (dum >= aamax) {
                imax = i;
                aamax = dum;
            }
        }
        if (j != imax) {
            for (int k = 1; k <= n; ++k) {
                double dum = A(imax, k);
                A(imax, k) = A(j, k);
                A(j, k) = dum;
            }
            vv[imax - 1] = vv[j - 1];
        }
        indx[j - 1] = imax;
        if (j != n) {
            if (A(j, j) == 0) A(j, j) = tiny;
            double dum = 1. / A(j, j);
            for (int i = j + 1; i <= n; ++i) A(i, j) = A(i, j) * dum;
        }
    }
    if (A(n, n) == 0.) A(n, n) = tiny;
}

static void lubksb(const double *a, int n, const int *indx, double *b) {
    int ii = 0;
    for (int i = 1; i <= n; ++i) {
        int ll = indx[i - 1];
        double sum = b[ll - 1];
        b[ll - 1] = b[i - 1];
        if (ii != 0) {
            for (int j = ii; j <= i - 1; ++j) sum = sum - A(i, j) * b[j - 1];
        } else if (sum != 0) {
            ii = i;
        }
        b[i - 1] = sum;
    }
    for (int i = n; i >= 1; --i) {
        double sum = b[i - 1];
        if (i < n)
            for (int j = i + 1; j <= n; ++j) sum = sum - A(i, j) * b[j - 1];
        b[i - 1] = sum / A(i, i);
    }
#undef A
}

/* ------------------------------------------------------------------ implicit.f90:83-218 */
void orc_dyn_set_time_step(const orc_tables *t, orc_dyn_tables *d, double dt) {
    double xa[KX * KX], xb[KX * KX], xe[KX * KX], ya[KX * KX], dsum[KX], xf[KX * KX];
    int indx[KX];
#define M2(a, k, k1) a[((k)-1) + KX * ((k1)-1)]
    for (int i = 0; i < NS; ++i) {
        d->dmp1[i] = 1.f / (1.f + d->dmp[i] * dt);
        d->dmp1d[i] = 1.f / (1.f + d->dmpd[i] * dt);
        d->dmp1s[i] = 1.f / (1.f + d->dmps[i] * dt);
    }
    const double xi = dt * ALPH;
    const double xxi = xi / (REARTH * REARTH);
    for (int k = 0; k < KX; ++k) d->dhsx[k] = xi * t->dhs[k];
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m)
            d->elz[(m - 1) + MX * (n - 1)] = (double)((float)(m + n - 2) * (float)(m + n - 1)) * xxi;
    memset(xa, 0, sizeof xa); /* xa(:kx, :kx-1) = 0 ; column kx is never read */
    memset(xb, 0, sizeof xb);
    for (int k = 1; k <= KX; ++k)
        for (int k1 = 1; k1 <= KX; ++k1) M2(ya, k, k1) = -AKAP * d->tref[k - 1] * t->dhs[k1 - 1];
    for (int k = 2; k <= KX; ++k)
        M2(xa, k, k - 1) = 0.5f * (AKAP * d->tref[k - 1] / t->fsg[k - 1] - (d->tref[k - 1] - d->tref[k - 2]) / t->dhs[k - 1]);
    for (int k = 1; k <= KX - 1; ++k)
        M2(xa, k, k) = 0.5f * (AKAP * d->tref[k - 1] / t->fsg[k - 1] - (d->tref[k] - d->tref[k - 1]) / t->dhs[k - 1]);
    dsum[0] = t->dhs[0];
    for (int k = 2; k <= KX; ++k) dsum[k - 1] = dsum[k - 2] + t->dhs[k - 1];
    for (int k = 1; k <= KX - 1; ++k)
        for (int k1 = 1; k1 <= KX; ++k1) {
            M2(xb, k, k1) = t->dhs[k1 - 1] * dsum[k - 1];
            if (k1 <= k) M2(xb, k, k1) = M2(xb, k, k1) - t->dhs[k1 - 1];
        }
    for (int k = 1; k <= KX; ++k)
        for (int k1 = 1; k1 <= KX; ++k1) {
            M2(d->xc, k, k1) = M2(ya, k, k1);
            for (int k2 = 1; k2 <= KX - 1; ++k2) M2(d->xc, k, k1) = M2(d->xc, k, k1) + M2(xa, k, k2) * M2(xb, k2, k1);
        }
    memset(d->xd, 0, sizeof d->xd);
    for (int k = 1; k <= KX; ++k)
        for (int k1 = k + 1; k1 <= KX; ++k1) M2(d->xd, k, k1) = RGAS * log(t->hsg[k1] / t->hsg[k1 - 1]);
    for (int k = 1; k <= KX; ++k) M2(d->xd, k, k) = RGAS * log(t->hsg[k] / t->fsg[k - 1]);
    for (int k = 1; k <= KX; ++k)
        for (int k1 = 1; k1 <= KX; ++k1) {
            M2(xe, k, k1) = 0.;
            for (int k2 = 1; k2 <= KX; ++k2) M2(xe, k, k1) = M2(xe, k, k1) + M2(d->xd, k, k2) * M2(d->xc, k2, k1);
        }
    for (int l = 1; l <= MX + NX + 1; ++l) {
        double xxx = (double)((float)l * (float)(l + 1)) / (REARTH * REARTH);
        for (int k = 1; k <= KX; ++k)
            for (int k1 = 1; k1 <= KX; ++k1)
                M2(xf, k, k1) = xi * xi * xxx * (RGAS * d->tref[k - 1] * t->dhs[k1 - 1] - M2(xe, k, k1));
        for (int k = 1; k <= KX; ++k) M2(xf, k, k) = M2(xf, k, k) + 1.f;
        double *y = d->xj + KX * KX * (l - 1);
        memset(y, 0, sizeof(double) * KX * KX);
        for (int i = 0; i < KX; ++i) y[i + KX * i] = 1.;
        ludcmp(xf, KX, indx);
        for (int i = 0; i < KX; ++i) lubksb(xf, KX, indx, y + KX * i);
    }
    for (int i = 0; i < KX * KX; ++i) d->xc[i] = d->xc[i] * xi;
#undef M2
}

/* ------------------------------------------------------------------ geopotential.f90:49-77 */
#define C3(a, m, n, k) ((a) + 2 * (((m)-1) + MX * (((n)-1) + NX * ((k)-1))))
void orc_geopotential(const orc_tables *t, const orc_dyn_tables *d, const double *tt /*(mx,nx,kx)*/, const double *phis,
                      double *phi) {
    for (int i = 0; i < 2 * NS; ++i) phi[i + 2 * NS * (KX - 1)] = phis[i] + d->xgeop1[KX - 1] * tt[i + 2 * NS * (KX - 1)];
    for (int k = KX - 1; k >= 1; --k)
        for (int i = 0; i < 2 * NS; ++i)
            phi[i + 2 * NS * (k - 1)] = phi[i + 2 * NS * k] + d->xgeop2[k] * tt[i + 2 * NS * k] + d->xgeop1[k - 1] * tt[i + 2 * NS * (k - 1)];
    for (int k = 2; k <= KX - 1; ++k) {
        double corf = d->xgeop1[k - 1] * 0.5f * log(t->hsg[k] / t->fsg[k - 1]) / log(t->fsg[k] / t->fsg[k - 2]);
        for (int n = 1; n <= NX; ++n)
            for (int c = 0; c < 2; ++c)
                C3(phi, 1, n, k)[c] = C3(phi, 1, n, k)[c] + corf * (C3(tt, 1, n, k + 1)[c] - C3(tt, 1, n, k - 1)[c]);
    }
}

/* ------------------------------------------------------------------ tendencies.f90:51-276 */
static void grid_point_tendencies(const orc_tables *t, const orc_dyn_tables *d, orc_state *s, double *vordt, double *divdt,
                                  double *tdt, double *psdt, double *trdt, int j1, int j2) {
    const size_t n3 = (size_t)NG * KX;
    double *w = (double *)malloc(sizeof(double) * (n3 * 12 + (size_t)NG * (KX + 1) * 3 + NG * 6));
    double *utend = w, *vtend = utend + n3, *ttend = vtend + n3, *trtend = ttend + n3, *ug = trtend + n3, *vg = ug + n3,
           *tg = vg + n3, *vorg = tg + n3, *divg = vorg + n3, *tgg = divg + n3, *puv = tgg + n3, *trg = puv + n3;
    double *sigdt = trg + n3, *temp = sigdt + (size_t)NG * (KX + 1), *sigm = temp + (size_t)NG * (KX + 1);
    double *px = sigm + (size_t)NG * (KX + 1), *py = px + NG, *umean = py + NG, *vmean = umean + NG, *dmean = vmean + NG,
           *tmpg = dmean + NG;
    double dumc[2][2 * NS], spec_tmp[2 * NS];
    const size_t lev = (size_t)2 * NS * KX; /* doubles per time level of a 3-D spectral variable */
#define G3(a, p, k) (a)[(p) + (size_t)NG * ((k)-1)]
#define S3(a, k, l) ((a) + (size_t)2 * NS * ((k)-1) + lev * ((l)-1))
    for (int k = 1; k <= KX; ++k) {
        orc_spec2grid(t, S3(s->vor, k, j2), vorg + (size_t)NG * (k - 1), 1);
        orc_spec2grid(t, S3(s->div, k, j2), divg + (size_t)NG * (k - 1), 1);
        orc_spec2grid(t, S3(s->t, k, j2), tg + (size_t)NG * (k - 1), 1);
        orc_spec2grid(t, S3(s->tr, k, j2), trg + (size_t)NG * (k - 1), 1);
        orc_vort2vel(t, S3(s->vor, k, j2), S3(s->div, k, j2), dumc[0], dumc[1]);
        orc_spec2grid(t, dumc[1], vg + (size_t)NG * (k - 1), 2);
        orc_spec2grid(t, dumc[0], ug + (size_t)NG * (k - 1), 2);
        for (int j = 0; j < IL; ++j)
            for (int i = 0; i < IX; ++i) G3(vorg, i + IX * j, k) = G3(vorg, i + IX * j, k) + t->coriol[j];
    }
    for (int p = 0; p < NG; ++p) umean[p] = vmean[p] = dmean[p] = 0.0;
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) {
            umean[p] = umean[p] + G3(ug, p, k) * t->dhs[k - 1];
            vmean[p] = vmean[p] + G3(vg, p, k) * t->dhs[k - 1];
            dmean[p] = dmean[p] + G3(divg, p, k) * t->dhs[k - 1];
        }
    {
        double psi[2 * NS];
        memcpy(psi, s->ps + (size_t)2 * NS * (j2 - 1), sizeof psi);
        orc_gradient(t, psi, dumc[0], dumc[1]);
    }
    orc_spec2grid(t, dumc[0], px, 2);
    orc_spec2grid(t, dumc[1], py, 2);
    for (int p = 0; p < NG; ++p) tmpg[p] = -umean[p] * px[p] - vmean[p] * py[p];
    orc_grid2spec(t, tmpg, psdt);
    psdt[0] = psdt[1] = 0.0;
    for (int p = 0; p < NG; ++p) {
        sigdt[p] = 0.0;
        sigdt[p + (size_t)NG * KX] = 0.0;
        sigm[p] = 0.0;
        sigm[p + (size_t)NG * KX] = 0.0;
    }
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) G3(puv, p, k) = (G3(ug, p, k) - umean[p]) * px[p] + (G3(vg, p, k) - vmean[p]) * py[p];
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) {
            G3(sigdt, p, k + 1) = G3(sigdt, p, k) - t->dhs[k - 1] * (G3(puv, p, k) + G3(divg, p, k) - dmean[p]);
            G3(sigm, p, k + 1) = G3(sigm, p, k) - t->dhs[k - 1] * G3(puv, p, k);
        }
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) G3(tgg, p, k) = G3(tg, p, k) - d->tref[k - 1];
    for (int p = 0; p < NG; ++p) {
        temp[p] = 0.0;
        temp[p + (size_t)NG * KX] = 0.0;
    }
    for (int k = 2; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) G3(temp, p, k) = G3(sigdt, p, k) * (G3(ug, p, k) - G3(ug, p, k - 1));
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p)
            G3(utend, p, k) = G3(vg, p, k) * G3(vorg, p, k) - G3(tgg, p, k) * RGAS * px[p] -
                              (G3(temp, p, k + 1) + G3(temp, p, k)) * t->dhsr[k - 1];
    for (int k = 2; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) G3(temp, p, k) = G3(sigdt, p, k) * (G3(vg, p, k) - G3(vg, p, k - 1));
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p)
            G3(vtend, p, k) = -G3(ug, p, k) * G3(vorg, p, k) - G3(tgg, p, k) * RGAS * py[p] -
                              (G3(temp, p, k + 1) + G3(temp, p, k)) * t->dhsr[k - 1];
    for (int k = 2; k <= KX; ++k)
        for (int p = 0; p < NG; ++p)
            G3(temp, p, k) = G3(sigdt, p, k) * (G3(tgg, p, k) - G3(tgg, p, k - 1)) + G3(sigm, p, k) * (d->tref[k - 1] - d->tref[k - 2]);
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p)
            G3(ttend, p, k) = G3(tgg, p, k) * G3(divg, p, k) - (G3(temp, p, k + 1) + G3(temp, p, k)) * t->dhsr[k - 1] +
                              t->fsgr[k - 1] * G3(tgg, p, k) * (G3(sigdt, p, k + 1) + G3(sigdt, p, k)) +
                              d->tref3[k - 1] * (G3(sigm, p, k + 1) + G3(sigm, p, k)) +
                              AKAP * (G3(tg, p, k) * G3(puv, p, k) - G3(tgg, p, k) * dmean[p]);
    for (int k = 2; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) G3(temp, p, k) = G3(sigdt, p, k) * (G3(trg, p, k) - G3(trg, p, k - 1));
    for (int p = 0; p < NG; ++p) G3(temp, p, 2) = G3(temp, p, 3) = 0.0;
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p)
            G3(trtend, p, k) = G3(trg, p, k) * G3(divg, p, k) - (G3(temp, p, k + 1) + G3(temp, p, k)) * t->dhsr[k - 1];

    /* physics (tendencies.f90:229-232) */
    orc_geopotential(t, d, s->t + lev * (j1 - 1), s->phis, s->phi);
    orc_physics_from_spectral(t, s, j1, utend, vtend, ttend, trtend);

    /* back to spectral space (tendencies.f90:238-268) */
    double *a = tmpg, *b = (double *)malloc(sizeof(double) * NG);
    for (int k = 1; k <= KX; ++k) {
        orc_grid_vel2vort(t, utend + (size_t)NG * (k - 1), vtend + (size_t)NG * (k - 1), S3(vordt, k, 1), S3(divdt, k, 1), 2);
        for (int p = 0; p < NG; ++p) a[p] = 0.5f * (G3(ug, p, k) * G3(ug, p, k) + G3(vg, p, k) * G3(vg, p, k));
        orc_grid2spec(t, a, spec_tmp);
        orc_laplacian(t, spec_tmp, dumc[0], 0);
        for (int i = 0; i < 2 * NS; ++i) S3(divdt, k, 1)[i] = S3(divdt, k, 1)[i] - dumc[0][i];
        for (int p = 0; p < NG; ++p) {
            a[p] = -G3(ug, p, k) * G3(tgg, p, k);
            b[p] = -G3(vg, p, k) * G3(tgg, p, k);
        }
        orc_grid_vel2vort(t, a, b, dumc[0], S3(tdt, k, 1), 2);
        orc_grid2spec(t, ttend + (size_t)NG * (k - 1), spec_tmp);
        for (int i = 0; i < 2 * NS; ++i) S3(tdt, k, 1)[i] = S3(tdt, k, 1)[i] + spec_tmp[i];
        for (int p = 0; p < NG; ++p) {
            a[p] = -G3(ug, p, k) * G3(trg, p, k);
            b[p] = -G3(vg, p, k) * G3(trg, p, k);
        }
        orc_grid_vel2vort(t, a, b, dumc[0], S3(trdt, k, 1), 2);
        orc_grid2spec(t, trtend + (size_t)NG * (k - 1), spec_tmp);
        for (int i = 0; i < 2 * NS; ++i) S3(trdt, k, 1)[i] = S3(trdt, k, 1)[i] + spec_tmp[i];
    }
    free(b);
    free(w);
}

/* ------------------------------------------------------------------ tendencies.f90:283-352 */
static void spectral_tendencies(const orc_tables *t, const orc_dyn_tables *d, orc_state *s, double *divdt, double *tdt,
                                double *psdt, int j2) {
    const size_t lev = (size_t)2 * NS * KX;
    double dmeanc[2 * NS], *sigdtc = (double *)calloc((size_t)2 * NS * (KX + 1) * 2, sizeof(double));
    double *dumk = sigdtc + (size_t)2 * NS * (KX + 1);
    memset(dmeanc, 0, sizeof dmeanc);
    for (int k = 1; k <= KX; ++k)
        for (int i = 0; i < 2 * NS; ++i) dmeanc[i] = dmeanc[i] + S3(s->div, k, j2)[i] * t->dhs[k - 1];
    for (int i = 0; i < 2 * NS; ++i) psdt[i] = psdt[i] - dmeanc[i];
    psdt[0] = psdt[1] = 0.0;
#define K2(a, k) ((a) + (size_t)2 * NS * ((k)-1))
    for (int k = 1; k <= KX - 1; ++k)
        for (int i = 0; i < 2 * NS; ++i) K2(sigdtc, k + 1)[i] = K2(sigdtc, k)[i] - t->dhs[k - 1] * (S3(s->div, k, j2)[i] - dmeanc[i]);
    for (int k = 2; k <= KX; ++k)
        for (int i = 0; i < 2 * NS; ++i) K2(dumk, k)[i] = K2(sigdtc, k)[i] * (d->tref[k - 1] - d->tref[k - 2]);
    for (int k = 1; k <= KX; ++k)
        for (int i = 0; i < 2 * NS; ++i)
            K2(tdt, k)[i] = K2(tdt, k)[i] - (K2(dumk, k + 1)[i] + K2(dumk, k)[i]) * t->dhsr[k - 1] +
                            d->tref3[k - 1] * (K2(sigdtc, k + 1)[i] + K2(sigdtc, k)[i]) - d->tref2[k - 1] * dmeanc[i];
    orc_geopotential(t, d, s->t + lev * (j2 - 1), s->phis, s->phi);
    double tmp[2 * NS], lap[2 * NS];
    for (int k = 1; k <= KX; ++k) {
        for (int i = 0; i < 2 * NS; ++i) tmp[i] = K2(s->phi, k)[i] + RGAS * d->tref[k - 1] * (s->ps + (size_t)2 * NS * (j2 - 1))[i];
        orc_laplacian(t, tmp, lap, 0);
        for (int i = 0; i < 2 * NS; ++i) K2(divdt, k)[i] = K2(divdt, k)[i] - lap[i];
    }
    free(sigdtc);
}

/* ------------------------------------------------------------------ implicit.f90:234-289 */
static void implicit_terms(const orc_dyn_tables *d, double *divdt, double *tdt, double *psdt) {
    double *ye = (double *)calloc((size_t)2 * NS * KX * 2, sizeof(double)), *yf = ye + (size_t)2 * NS * KX;
#define XD(k, k1) d->xd[((k)-1) + KX * ((k1)-1)]
#define XC(k, k1) d->xc[((k)-1) + KX * ((k1)-1)]
#define XJ(k, k1, l) d->xj[((k)-1) + KX * (((k1)-1) + KX * ((l)-1))]
    for (int k1 = 1; k1 <= KX; ++k1)
        for (int k = 1; k <= KX; ++k)
            for (int i = 0; i < 2 * NS; ++i) K2(ye, k)[i] = K2(ye, k)[i] + XD(k, k1) * K2(tdt, k1)[i];
    for (int k = 1; k <= KX; ++k)
        for (int i = 0; i < 2 * NS; ++i) K2(ye, k)[i] = K2(ye, k)[i] + RGAS * d->tref[k - 1] * psdt[i];
    for (int k = 1; k <= KX; ++k)
        for (int q = 0; q < NS; ++q)
            for (int c = 0; c < 2; ++c) K2(yf, k)[2 * q + c] = K2(divdt, k)[2 * q + c] + d->elz[q] * K2(ye, k)[2 * q + c];
    memset(divdt, 0, sizeof(double) * 2 * NS * KX);
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m) {
            if ((m + n - 2) == 0) continue;
            int q = (m - 1) + MX * (n - 1);
            for (int k1 = 1; k1 <= KX; ++k1)
                for (int k = 1; k <= KX; ++k)
                    for (int c = 0; c < 2; ++c)
                        K2(divdt, k)[2 * q + c] = K2(divdt, k)[2 * q + c] + XJ(k, k1, m + n - 2) * K2(yf, k1)[2 * q + c];
        }
    for (int k = 1; k <= KX; ++k)
        for (int i = 0; i < 2 * NS; ++i) psdt[i] = psdt[i] - K2(divdt, k)[i] * d->dhsx[k - 1];
    for (int k = 1; k <= KX; ++k)
        for (int k1 = 1; k1 <= KX; ++k1)
            for (int i = 0; i < 2 * NS; ++i) K2(tdt, k)[i] = K2(tdt, k)[i] + XC(k, k1) * K2(divdt, k1)[i];
    free(ye);
}

void orc_get_tendencies(const orc_tables *t, const orc_dyn_tables *d, orc_state *s, double *vordt, double *divdt,
                        double *tdt, double *psdt, double *trdt, int j2) { /* tendencies.f90:11-39, alph = 0.5 */
    grid_point_tendencies(t, d, s, vordt, divdt, tdt, psdt, trdt, 1, j2);
    spectral_tendencies(t, d, s, divdt, tdt, psdt, 1);
    implicit_terms(d, divdt, tdt, psdt);
}

/* ------------------------------------------------------------------ time_stepping.f90:38-188 */
static void hdiff(const double *field, double *fdt, const double *dmp, const double *dmp1) { /* 3-D, in place on fdt */
    for (int k = 0; k < KX; ++k)
        for (int q = 0; q < NS; ++q)
            for (int c = 0; c < 2; ++c) {
                size_t i = (size_t)2 * NS * k + 2 * q + c;
                fdt[i] = (fdt[i] - dmp[q] * field[i]) * dmp1[q];
            }
}

static void step_field_2d(const orc_tables *t, int j1, double dt, double eps, double *f1, double *f2, double *fdt) {
    const double eps2 = 1.0f - 2.0f * eps;
    (void)eps2;
    orc_truncate(t, fdt);
    double *fj1 = (j1 == 1) ? f1 : f2;
    for (int i = 0; i < 2 * NS; ++i) {
        double o1 = f1[i], oj = fj1[i];
        double fnew = o1 + dt * fdt[i];
        double n1 = oj + WIL * eps * (o1 - 2 * oj + fnew);
        /* output(:,:,1) is overwritten first; the Williams term then uses the NEW level 1 and, when j1 = 1, the NEW
         * output(:,:,j1) as well (time_stepping.f90:184-187) */
        double oj_after = (j1 == 1) ? n1 : oj;
        double n2 = fnew - (1.0f - WIL) * eps * (n1 - 2.0f * oj_after + fnew);
        f1[i] = n1;
        f2[i] = n2;
    }
}

void orc_step(const orc_tables *t, const orc_dyn_tables *d, orc_state *s, int j1, int j2, double dt) {
    const size_t lev = (size_t)2 * NS * KX;
    double *vordt = (double *)calloc(lev * 5 + 2 * NS, sizeof(double)), *divdt = vordt + lev, *tdt = divdt + lev,
           *trdt = tdt + lev, *ctmp = trdt + lev, *psdt = ctmp + lev;
    orc_get_tendencies(t, d, s, vordt, divdt, tdt, psdt, trdt, j2);
    hdiff(s->vor, vordt, d->dmp, d->dmp1);
    hdiff(s->div, divdt, d->dmpd, d->dmp1d);
    for (int k = 0; k < KX; ++k)
        for (int q = 0; q < NS; ++q)
            for (int c = 0; c < 2; ++c)
                ctmp[(size_t)2 * NS * k + 2 * q + c] = s->t[(size_t)2 * NS * k + 2 * q + c] + s->tcorh[2 * q + c] * d->tcorv[k];
    hdiff(ctmp, tdt, d->dmp, d->dmp1);
    const double sdrag = 1.0f / (TDRS * 3600.0f);
    for (int n = 1; n <= NX; ++n)
        for (int c = 0; c < 2; ++c) {
            size_t i = (size_t)2 * (0 + MX * (n - 1)) + c; /* m = 1, level 1 */
            vordt[i] = vordt[i] - sdrag * s->vor[i];
            divdt[i] = divdt[i] - sdrag * s->div[i];
        }
    hdiff(s->vor, vordt, d->dmps, d->dmp1s);
    hdiff(s->div, divdt, d->dmps, d->dmp1s);
    hdiff(ctmp, tdt, d->dmps, d->dmp1s);
    for (int k = 0; k < KX; ++k)
        for (int q = 0; q < NS; ++q)
            for (int c = 0; c < 2; ++c)
                ctmp[(size_t)2 * NS * k + 2 * q + c] = s->tr[(size_t)2 * NS * k + 2 * q + c] + s->qcorh[2 * q + c] * d->qcorv[k];
    hdiff(ctmp, trdt, d->dmpd, d->dmp1d);
    const double eps = (j1 == 1) ? 0.0 : ROB;
    step_field_2d(t, j1, dt, eps, s->ps, s->ps + 2 * NS, psdt);
    for (int k = 0; k < KX; ++k) {
        size_t o = (size_t)2 * NS * k;
        step_field_2d(t, j1, dt, eps, s->vor + o, s->vor + lev + o, vordt + o);
        step_field_2d(t, j1, dt, eps, s->div + o, s->div + lev + o, divdt + o);
        step_field_2d(t, j1, dt, eps, s->t + o, s->t + lev + o, tdt + o);
        step_field_2d(t, j1, dt, eps, s->tr + o, s->tr + lev + o, trdt + o);
    }
    free(vordt);
}

/* ------------------------------------------------------------------ diagnostics.f90:16-76 ; returns 0 or -2 */
int orc_check_diagnostics(const orc_tables *t, const orc_state *s, int time_lev, double *diag /* (kx,3) or NULL */) {
    const size_t lev = (size_t)2 * NS * KX;
    double local[KX * 3];
    double *dg = diag ? diag : local;
    double tmp[2 * NS];
    int err = 0;
    for (int k = 1; k <= KX; ++k) {
        const double *vor = s->vor + lev * (time_lev - 1) + (size_t)2 * NS * (k - 1);
        const double *div = s->div + lev * (time_lev - 1) + (size_t)2 * NS * (k - 1);
        double d1 = 0.0, d2 = 0.0;
        orc_laplacian(t, vor, tmp, 1);
        for (int m = 2; m <= MX; ++m)
            for (int n = 1; n <= NX; ++n) {
                int q = (m - 1) + MX * (n - 1);
                d1 = d1 - (tmp[2 * q] * vor[2 * q] + tmp[2 * q + 1] * vor[2 * q + 1]); /* real(temp * conjg(vor)) */
            }
        orc_laplacian(t, div, tmp, 1);
        for (int m = 2; m <= MX; ++m)
            for (int n = 1; n <= NX; ++n) {
                int q = (m - 1) + MX * (n - 1);
                d2 = d2 - (tmp[2 * q] * div[2 * q] + tmp[2 * q + 1] * div[2 * q + 1]);
            }
        dg[k - 1] = d1;
        dg[k - 1 + KX] = d2;
        dg[k - 1 + 2 * KX] = (double)sqrtf(0.5f) * (s->t + lev * (time_lev - 1) + (size_t)2 * NS * (k - 1))[0];
    }
    for (int k = 0; k < KX; ++k)
        if (dg[k] > 500.0f || dg[k + KX] > 500.0f || dg[k + 2 * KX] < 180.0f || dg[k + 2 * KX] > 320.0f) err = -2;
    return err;
}
