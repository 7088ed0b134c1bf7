/* TEST INFRASTRUCTURE -- CPU oracle (see speedy_oracle.h).  NOT PART OF THE PRODUCT.
 *
 * Tables and spectral transforms of the reference, restated in plain C with the same operation
 * order and the same single-precision-seeded constants.  Build with
 * -O2 -ffp-contract=off and no -march flag so that arithmetic is plain IEEE double / float,
 * as in the flang-built reference on generic x86-64.
 */
#include "speedy_oracle.h"

#include <math.h>
#include <string.h>

#define IX ORC_IX
#define IL ORC_IL
#define IY ORC_IY
#define KX ORC_KX
#define MX ORC_MX
#define NX ORC_NX
#define TRUNC ORC_TRUNC

/* physical_constants.f90:16-30: default-real literals widened to double */
static const double REARTH = 6.371e+6f;
static const double OMEGA = 7.292e-05f;
static const double GRAV = 9.81f;
static const double P0 = 1.e+5f;
static const double CP = 1004.0f;
#define AKAP ((double)(2.0f / 7.0f))

/* ------------------------------------------------------------------------------------------
 * geometry.f90:67-170
 * ---------------------------------------------------------------------------------------- */
static void init_geometry(orc_tables *t) {
    static const float hsg8[9] = {0.000f, 0.050f, 0.140f, 0.260f, 0.420f, 0.600f, 0.770f, 0.900f, 1.000f}; /* :89 */
    for (int k = 0; k < 9; ++k) t->hsg[k] = hsg8[k];
    for (int k = 0; k < KX; ++k) { /* :93-102 */
        t->dhs[k] = t->hsg[k + 1] - t->hsg[k];
        t->fsg[k] = 0.5 * (t->hsg[k + 1] + t->hsg[k]);
    }
    for (int k = 0; k < KX; ++k) {
        t->dhsr[k] = 0.5 / t->dhs[k];
        t->fsgr[k] = AKAP / (2. * t->fsg[k]);
    }
    for (int j = 1; j <= IY; ++j) { /* :108-119, the sine is evaluated entirely in single precision */
        int jj = IL + 1 - j;
        float arg = 3.141592654f * ((float)j - 0.25f) / ((float)IL + 0.5f);
        double sh = (double)cosf(arg);
        double ch = sqrt(1.0 - sh * sh);
        t->sia_half[j - 1] = sh;
        t->coa_half[j - 1] = ch;
        t->sia[j - 1] = -sh;
        t->sia[jj - 1] = sh;
        t->coa[j - 1] = ch;
        t->coa[jj - 1] = ch;
        t->radang[j - 1] = -asin(sh);
        t->radang[jj - 1] = asin(sh);
        t->cosgr[j - 1] = t->cosgr[jj - 1] = 1. / ch; /* :122-130 */
        t->cosgr2[j - 1] = t->cosgr2[jj - 1] = 1. / (ch * ch);
    }
    for (int j = 0; j < IL; ++j) t->coriol[j] = 2.0 * OMEGA * t->sia[j]; /* :132 */
    t->sigh[0] = t->hsg[0];
    for (int k = 0; k < KX; ++k) { /* :137-142 */
        t->sigl[k] = log(t->fsg[k]);
        t->sigh[k + 1] = t->hsg[k + 1];
        t->grdsig[k] = GRAV / (t->dhs[k] * P0);
        t->grdscp[k] = t->grdsig[k] / CP;
    }
    /* wvi(kx,2), :148-154 ; column-major wvi(k,c) -> wvi[k + 8*c] */
    for (int k = 0; k < KX - 1; ++k) {
        t->wvi[k] = 1. / (t->sigl[k + 1] - t->sigl[k]);
        t->wvi[k + 8] = (log(t->sigh[k + 1]) - t->sigl[k]) * t->wvi[k];
    }
    t->wvi[KX - 1] = 0.;
    t->wvi[KX - 1 + 8] = ((double)logf(0.99f) - t->sigl[KX - 1]) * t->wvi[KX - 2];
}

/* ------------------------------------------------------------------------------------------
 * legendre.f90:38-112, 224-307
 * ---------------------------------------------------------------------------------------- */
#define EPSI(m, n) t->epsi[((m)-1) + (MX + 1) * ((n)-1)]
#define REPSI(m, n) t->repsi[((m)-1) + (MX + 1) * ((n)-1)]
#define CPOL(m, n, j) t->cpol[((m)-1) + 2 * MX * (((n)-1) + NX * ((j)-1))]

static void gauss_weights(double *w) { /* legendre.f90:224-257 */
    const int n = 2 * IY;
    double z1 = 2.0, pp = 0.0;
    for (int i = 1; i <= IY; ++i) {
        double z = cos(3.141592654 * ((double)i - 0.25) / ((double)n + 0.5));
        while (fabs(z - z1) > 2.220446049250313e-16) {
            double p1 = 1.0, p2 = 0.0, p3;
            for (int j = 1; j <= n; ++j) {
                p3 = p2;
                p2 = p1;
                p1 = ((2.0 * (double)j - 1.0) * z * p2 - ((double)j - 1.0) * p3) / j;
            }
            pp = (double)n * (z * p1 - p2) / (z * z - 1.0);
            z1 = z;
            z = z1 - p1 / pp;
        }
        w[i - 1] = 2.0 / ((1.0 - z * z) * (pp * pp));
    }
}

static void legendre_poly(const orc_tables *t, int j, double *poly /* mx x nx */) { /* legendre.f90:260-307 */
    static const double small = 1.e-30f;
    double alp[(MX + 1) * NX];
    double consq[MX];
#define ALP(m, n) alp[((m)-1) + (MX + 1) * ((n)-1)]
    double y = t->coa_half[j - 1], x = t->sia_half[j - 1];
    for (int m = 1; m <= MX; ++m) consq[m - 1] = (double)sqrtf(0.5f * (2.0f * (float)m + 1.0f) / (float)m);
    ALP(1, 1) = (double)sqrtf(0.5f);
    for (int m = 2; m <= MX + 1; ++m) ALP(m, 1) = consq[m - 2] * y * ALP(m - 1, 1);
    for (int m = 1; m <= MX + 1; ++m) ALP(m, 2) = (x * ALP(m, 1)) * REPSI(m, 2);
    for (int n = 3; n <= NX; ++n)
        for (int m = 1; m <= MX + 1; ++m)
            ALP(m, n) = (x * ALP(m, n - 1) - EPSI(m, n - 1) * ALP(m, n - 2)) * REPSI(m, n);
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX + 1; ++m)
            if (fabs(ALP(m, n)) <= small) ALP(m, n) = 0.0;
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m) poly[(m - 1) + MX * (n - 1)] = ALP(m, n);
#undef ALP
}

static void init_legendre(orc_tables *t) {
    double poly[MX * NX];
    gauss_weights(t->wt);
    for (int n = 1; n <= NX; ++n) { /* :68-77 */
        t->nsh2[n - 1] = 0;
        for (int m = 1; m <= MX; ++m)
            if ((m - 1) + (n - 1) <= TRUNC + 1) t->nsh2[n - 1] += 2;
    }
    for (int m = 1; m <= MX + 1; ++m) /* :79-96, squares taken in single precision */
        for (int n = 1; n <= NX + 1; ++n) {
            float fm = (float)(m - 1), fl = (float)(n + m - 2);
            double emm2 = (double)(fm * fm), ell2 = (double)(fl * fl);
            if (n == NX + 1 || (n == 1 && m == 1))
                EPSI(m, n) = 0.0;
            else
                EPSI(m, n) = sqrt((ell2 - emm2) / (4.0 * ell2 - 1.0));
            REPSI(m, n) = 0.0;
            if (EPSI(m, n) > 0.) REPSI(m, n) = 1.0 / EPSI(m, n);
        }
    for (int j = 1; j <= IY; ++j) { /* :98-108 */
        legendre_poly(t, j, poly);
        for (int n = 1; n <= NX; ++n)
            for (int m = 1; m <= MX; ++m) {
                CPOL(2 * m - 1, n, j) = poly[(m - 1) + MX * (n - 1)];
                CPOL(2 * m, n, j) = poly[(m - 1) + MX * (n - 1)];
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * fftpack.f90:1-67 (rffti1) for general n built from factors 4,2,3,5
 * ---------------------------------------------------------------------------------------- */
static void fft_init(int n, double *wa, int *ifac) {
    static const int ntryh[4] = {4, 2, 3, 5};
    int nl = n, nf = 0, j = 0, ntry = 0;
    memset(wa, 0, sizeof(double) * n);
    for (;;) {
        ++j;
        ntry = (j <= 4) ? ntryh[j - 1] : ntry + 2;
        while (nl % ntry == 0) {
            ++nf;
            ifac[nf + 1] = ntry;
            nl /= ntry;
            if (ntry == 2 && nf != 1) { /* keep the factor 2 in front, :28-34 */
                for (int i = 2; i <= nf; ++i) {
                    int ib = nf - i + 2;
                    ifac[ib + 1] = ifac[ib];
                }
                ifac[2] = 2;
            }
            if (nl == 1) break;
        }
        if (nl == 1) break;
    }
    ifac[0] = n;
    ifac[1] = nf;
    double tpi = (double)(8.f * atanf(1.f)); /* :39, single precision */
    double argh = tpi / n;
    int is = 0, l1 = 1;
    for (int k1 = 1; k1 <= nf - 1; ++k1) {
        int ip = ifac[k1 + 1], ld = 0, l2 = l1 * ip, ido = n / l2;
        for (int jj = 1; jj <= ip - 1; ++jj) {
            ld += l1;
            int i = is;
            double argld = ld * argh, fi = 0.;
            for (int ii = 3; ii <= ido; ii += 2) {
                i += 2;
                fi += 1.;
                double arg = fi * argld;
                wa[i - 2] = cos(arg);
                wa[i - 1] = sin(arg);
            }
            is += ido;
        }
        l1 = l2;
    }
}

/* radix passes.  Index macros are 1-based like the reference's dummy arrays. */
#define CCF(i, k, j) cc[((i)-1) + ido * (((k)-1) + l1 * ((j)-1))] /* forward input  cc(ido,l1,ip) */
#define CHF(i, j, k) ch[((i)-1) + ido * (((j)-1) + ip * ((k)-1))] /* forward output ch(ido,ip,l1) */
#define CCB(i, j, k) cc[((i)-1) + ido * (((j)-1) + ip * ((k)-1))] /* backward input  cc(ido,ip,l1) */
#define CHB(i, k, j) ch[((i)-1) + ido * (((k)-1) + l1 * ((j)-1))] /* backward output ch(ido,l1,ip) */

static void radf2(int ido, int l1, const double *cc, double *ch, const double *wa1) { /* fftpack.f90:722-772 */
    const int ip = 2;
    for (int k = 1; k <= l1; ++k) {
        CHF(1, 1, k) = CCF(1, k, 1) + CCF(1, k, 2);
        CHF(ido, 2, k) = CCF(1, k, 1) - CCF(1, k, 2);
    }
    if (ido < 2) return;
    if (ido > 2) {
        for (int k = 1; k <= l1; ++k)
            for (int i = 3; i <= ido; i += 2) {
                int ic = ido + 2 - i;
                double tr2 = wa1[i - 3] * CCF(i - 1, k, 2) + wa1[i - 2] * CCF(i, k, 2);
                double ti2 = wa1[i - 3] * CCF(i, k, 2) - wa1[i - 2] * CCF(i - 1, k, 2);
                CHF(i, 1, k) = CCF(i, k, 1) + ti2;
                CHF(ic, 2, k) = ti2 - CCF(i, k, 1);
                CHF(i - 1, 1, k) = CCF(i - 1, k, 1) + tr2;
                CHF(ic - 1, 2, k) = CCF(i - 1, k, 1) - tr2;
            }
        if (ido % 2 == 1) return;
    }
    for (int k = 1; k <= l1; ++k) {
        CHF(1, 2, k) = -CCF(ido, k, 2);
        CHF(ido, 1, k) = CCF(ido, k, 1);
    }
}

static void radf3(int ido, int l1, const double *cc, double *ch, const double *wa1, const double *wa2) { /* :774-842 */
    const int ip = 3;
    const double taur = -.5f, taui = (double)(.5f * sqrtf(3.f));
    for (int k = 1; k <= l1; ++k) {
        double cr2 = CCF(1, k, 2) + CCF(1, k, 3);
        CHF(1, 1, k) = CCF(1, k, 1) + cr2;
        CHF(1, 3, k) = taui * (CCF(1, k, 3) - CCF(1, k, 2));
        CHF(ido, 2, k) = CCF(1, k, 1) + taur * cr2;
    }
    if (ido == 1) return;
    for (int k = 1; k <= l1; ++k)
        for (int i = 3; i <= ido; i += 2) {
            int ic = ido + 2 - i;
            double dr2 = wa1[i - 3] * CCF(i - 1, k, 2) + wa1[i - 2] * CCF(i, k, 2);
            double di2 = wa1[i - 3] * CCF(i, k, 2) - wa1[i - 2] * CCF(i - 1, k, 2);
            double dr3 = wa2[i - 3] * CCF(i - 1, k, 3) + wa2[i - 2] * CCF(i, k, 3);
            double di3 = wa2[i - 3] * CCF(i, k, 3) - wa2[i - 2] * CCF(i - 1, k, 3);
            double cr2 = dr2 + dr3, ci2 = di2 + di3;
            CHF(i - 1, 1, k) = CCF(i - 1, k, 1) + cr2;
            CHF(i, 1, k) = CCF(i, k, 1) + ci2;
            double tr2 = CCF(i - 1, k, 1) + taur * cr2;
            double ti2 = CCF(i, k, 1) + taur * ci2;
            double tr3 = taui * (di2 - di3);
            double ti3 = taui * (dr3 - dr2);
            CHF(i - 1, 3, k) = tr2 + tr3;
            CHF(ic - 1, 2, k) = tr2 - tr3;
            CHF(i, 3, k) = ti2 + ti3;
            CHF(ic, 2, k) = ti3 - ti2;
        }
}

static void radf4(int ido, int l1, const double *cc, double *ch, const double *wa1, const double *wa2,
                  const double *wa3) { /* :844-943 */
    const int ip = 4;
    const double hsqt2 = (double)(.5f * sqrtf(2.f));
    for (int k = 1; k <= l1; ++k) {
        double tr1 = CCF(1, k, 2) + CCF(1, k, 4);
        double tr2 = CCF(1, k, 1) + CCF(1, k, 3);
        CHF(1, 1, k) = tr1 + tr2;
        CHF(ido, 4, k) = tr2 - tr1;
        CHF(ido, 2, k) = CCF(1, k, 1) - CCF(1, k, 3);
        CHF(1, 3, k) = CCF(1, k, 4) - CCF(1, k, 2);
    }
    if (ido < 2) return;
    if (ido > 2) {
        for (int k = 1; k <= l1; ++k)
            for (int i = 3; i <= ido; i += 2) {
                int ic = ido + 2 - i;
                double cr2 = wa1[i - 3] * CCF(i - 1, k, 2) + wa1[i - 2] * CCF(i, k, 2);
                double ci2 = wa1[i - 3] * CCF(i, k, 2) - wa1[i - 2] * CCF(i - 1, k, 2);
                double cr3 = wa2[i - 3] * CCF(i - 1, k, 3) + wa2[i - 2] * CCF(i, k, 3);
                double ci3 = wa2[i - 3] * CCF(i, k, 3) - wa2[i - 2] * CCF(i - 1, k, 3);
                double cr4 = wa3[i - 3] * CCF(i - 1, k, 4) + wa3[i - 2] * CCF(i, k, 4);
                double ci4 = wa3[i - 3] * CCF(i, k, 4) - wa3[i - 2] * CCF(i - 1, k, 4);
                double tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
                double ti2 = CCF(i, k, 1) + ci3, ti3 = CCF(i, k, 1) - ci3;
                double tr2 = CCF(i - 1, k, 1) + cr3, tr3 = CCF(i - 1, k, 1) - cr3;
                CHF(i - 1, 1, k) = tr1 + tr2;
                CHF(ic - 1, 4, k) = tr2 - tr1;
                CHF(i, 1, k) = ti1 + ti2;
                CHF(ic, 4, k) = ti1 - ti2;
                CHF(i - 1, 3, k) = ti4 + tr3;
                CHF(ic - 1, 2, k) = tr3 - ti4;
                CHF(i, 3, k) = tr4 + ti3;
                CHF(ic, 2, k) = tr4 - ti3;
            }
        if (ido % 2 == 1) return;
    }
    for (int k = 1; k <= l1; ++k) {
        double ti1 = -hsqt2 * (CCF(ido, k, 2) + CCF(ido, k, 4));
        double tr1 = hsqt2 * (CCF(ido, k, 2) - CCF(ido, k, 4));
        CHF(ido, 1, k) = tr1 + CCF(ido, k, 1);
        CHF(ido, 3, k) = CCF(ido, k, 1) - tr1;
        CHF(1, 2, k) = ti1 - CCF(ido, k, 3);
        CHF(1, 4, k) = ti1 + CCF(ido, k, 3);
    }
}

static void radb2(int ido, int l1, const double *cc, double *ch, const double *wa1) { /* :204-254 */
    const int ip = 2;
    for (int k = 1; k <= l1; ++k) {
        CHB(1, k, 1) = CCB(1, 1, k) + CCB(ido, 2, k);
        CHB(1, k, 2) = CCB(1, 1, k) - CCB(ido, 2, k);
    }
    if (ido < 2) return;
    if (ido > 2) {
        for (int k = 1; k <= l1; ++k)
            for (int i = 3; i <= ido; i += 2) {
                int ic = ido + 2 - i;
                CHB(i - 1, k, 1) = CCB(i - 1, 1, k) + CCB(ic - 1, 2, k);
                double tr2 = CCB(i - 1, 1, k) - CCB(ic - 1, 2, k);
                CHB(i, k, 1) = CCB(i, 1, k) - CCB(ic, 2, k);
                double ti2 = CCB(i, 1, k) + CCB(ic, 2, k);
                CHB(i - 1, k, 2) = wa1[i - 3] * tr2 - wa1[i - 2] * ti2;
                CHB(i, k, 2) = wa1[i - 3] * ti2 + wa1[i - 2] * tr2;
            }
        if (ido % 2 == 1) return;
    }
    for (int k = 1; k <= l1; ++k) {
        CHB(ido, k, 1) = CCB(ido, 1, k) + CCB(ido, 1, k);
        CHB(ido, k, 2) = -(CCB(1, 2, k) + CCB(1, 2, k));
    }
}

static void radb3(int ido, int l1, const double *cc, double *ch, const double *wa1, const double *wa2) { /* :256-326 */
    const int ip = 3;
    const double taur = -.5f, taui = (double)(.5f * sqrtf(3.f));
    for (int k = 1; k <= l1; ++k) {
        double tr2 = CCB(ido, 2, k) + CCB(ido, 2, k);
        double cr2 = CCB(1, 1, k) + taur * tr2;
        CHB(1, k, 1) = CCB(1, 1, k) + tr2;
        double ci3 = taui * (CCB(1, 3, k) + CCB(1, 3, k));
        CHB(1, k, 2) = cr2 - ci3;
        CHB(1, k, 3) = cr2 + ci3;
    }
    if (ido == 1) return;
    for (int k = 1; k <= l1; ++k)
        for (int i = 3; i <= ido; i += 2) {
            int ic = ido + 2 - i;
            double tr2 = CCB(i - 1, 3, k) + CCB(ic - 1, 2, k);
            double cr2 = CCB(i - 1, 1, k) + taur * tr2;
            CHB(i - 1, k, 1) = CCB(i - 1, 1, k) + tr2;
            double ti2 = CCB(i, 3, k) - CCB(ic, 2, k);
            double ci2 = CCB(i, 1, k) + taur * ti2;
            CHB(i, k, 1) = CCB(i, 1, k) + ti2;
            double cr3 = taui * (CCB(i - 1, 3, k) - CCB(ic - 1, 2, k));
            double ci3 = taui * (CCB(i, 3, k) + CCB(ic, 2, k));
            double dr2 = cr2 - ci3, dr3 = cr2 + ci3, di2 = ci2 + cr3, di3 = ci2 - cr3;
            CHB(i - 1, k, 2) = wa1[i - 3] * dr2 - wa1[i - 2] * di2;
            CHB(i, k, 2) = wa1[i - 3] * di2 + wa1[i - 2] * dr2;
            CHB(i - 1, k, 3) = wa2[i - 3] * dr3 - wa2[i - 2] * di3;
            CHB(i, k, 3) = wa2[i - 3] * di3 + wa2[i - 2] * dr3;
        }
}

static void radb4(int ido, int l1, const double *cc, double *ch, const double *wa1, const double *wa2,
                  const double *wa3) { /* :328-424 */
    const int ip = 4;
    const double sqrt2 = (double)sqrtf(2.f);
    for (int k = 1; k <= l1; ++k) {
        double tr1 = CCB(1, 1, k) - CCB(ido, 4, k);
        double tr2 = CCB(1, 1, k) + CCB(ido, 4, k);
        double tr3 = CCB(ido, 2, k) + CCB(ido, 2, k);
        double tr4 = CCB(1, 3, k) + CCB(1, 3, k);
        CHB(1, k, 1) = tr2 + tr3;
        CHB(1, k, 2) = tr1 - tr4;
        CHB(1, k, 3) = tr2 - tr3;
        CHB(1, k, 4) = tr1 + tr4;
    }
    if (ido < 2) return;
    if (ido > 2) {
        for (int k = 1; k <= l1; ++k)
            for (int i = 3; i <= ido; i += 2) {
                int ic = ido + 2 - i;
                double ti1 = CCB(i, 1, k) + CCB(ic, 4, k);
                double ti2 = CCB(i, 1, k) - CCB(ic, 4, k);
                double ti3 = CCB(i, 3, k) - CCB(ic, 2, k);
                double tr4 = CCB(i, 3, k) + CCB(ic, 2, k);
                double tr1 = CCB(i - 1, 1, k) - CCB(ic - 1, 4, k);
                double tr2 = CCB(i - 1, 1, k) + CCB(ic - 1, 4, k);
                double ti4 = CCB(i - 1, 3, k) - CCB(ic - 1, 2, k);
                double tr3 = CCB(i - 1, 3, k) + CCB(ic - 1, 2, k);
                CHB(i - 1, k, 1) = tr2 + tr3;
                double cr3 = tr2 - tr3;
                CHB(i, k, 1) = ti2 + ti3;
                double ci3 = ti2 - ti3;
                double cr2 = tr1 - tr4, cr4 = tr1 + tr4, ci2 = ti1 + ti4, ci4 = ti1 - ti4;
                CHB(i - 1, k, 2) = wa1[i - 3] * cr2 - wa1[i - 2] * ci2;
                CHB(i, k, 2) = wa1[i - 3] * ci2 + wa1[i - 2] * cr2;
                CHB(i - 1, k, 3) = wa2[i - 3] * cr3 - wa2[i - 2] * ci3;
                CHB(i, k, 3) = wa2[i - 3] * ci3 + wa2[i - 2] * cr3;
                CHB(i - 1, k, 4) = wa3[i - 3] * cr4 - wa3[i - 2] * ci4;
                CHB(i, k, 4) = wa3[i - 3] * ci4 + wa3[i - 2] * cr4;
            }
        if (ido % 2 == 1) return;
    }
    for (int k = 1; k <= l1; ++k) {
        double ti1 = CCB(1, 2, k) + CCB(1, 4, k);
        double ti2 = CCB(1, 4, k) - CCB(1, 2, k);
        double tr1 = CCB(ido, 1, k) - CCB(ido, 3, k);
        double tr2 = CCB(ido, 1, k) + CCB(ido, 3, k);
        CHB(ido, k, 1) = tr2 + tr2;
        CHB(ido, k, 2) = sqrt2 * (tr1 - ti1);
        CHB(ido, k, 3) = ti2 + ti2;
        CHB(ido, k, 4) = -sqrt2 * (tr1 + ti1);
    }
}

/* rfftf1 / rfftb1 (fftpack.f90:136-202, 69-134), factors 2,3,4 only (all that N=96 needs). */
void orc_rfftf96(const orc_tables *t, double *c) {
    const int n = IX;
    double chbuf[IX];
    const int *ifac = t->ifac;
    const double *wa = t->work;
    int nf = ifac[1], na = 1, l2 = n, iw = n;
    for (int k1 = 1; k1 <= nf; ++k1) {
        int kh = nf - k1, ip = ifac[kh + 2], l1 = l2 / ip, ido = n / l2;
        iw -= (ip - 1) * ido;
        na = 1 - na;
        const double *in = na ? chbuf : c;
        double *out = na ? c : chbuf;
        if (ip == 4)
            radf4(ido, l1, in, out, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1);
        else if (ip == 2)
            radf2(ido, l1, in, out, wa + iw - 1);
        else
            radf3(ido, l1, in, out, wa + iw - 1, wa + iw + ido - 1);
        l2 = l1;
    }
    if (na != 1) memcpy(c, chbuf, sizeof chbuf);
}

void orc_rfftb96(const orc_tables *t, double *c) {
    const int n = IX;
    double chbuf[IX];
    const int *ifac = t->ifac;
    const double *wa = t->work;
    int nf = ifac[1], na = 0, l1 = 1, iw = 1;
    for (int k1 = 1; k1 <= nf; ++k1) {
        int ip = ifac[k1 + 1], l2 = ip * l1, ido = n / l2;
        const double *in = na ? chbuf : c;
        double *out = na ? c : chbuf;
        if (ip == 4)
            radb4(ido, l1, in, out, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1);
        else if (ip == 2)
            radb2(ido, l1, in, out, wa + iw - 1);
        else
            radb3(ido, l1, in, out, wa + iw - 1, wa + iw + ido - 1);
        na = 1 - na;
        l1 = l2;
        iw += (ip - 1) * ido;
    }
    if (na != 0) memcpy(c, chbuf, sizeof chbuf);
}

/* ------------------------------------------------------------------------------------------
 * spectral.f90:39-116
 * ---------------------------------------------------------------------------------------- */
#define S2(a, m, n) t->a[((m)-1) + MX * ((n)-1)]
static void init_spectral(orc_tables *t) {
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m) {
            int l = (m - 1) + (n - 1);
            S2(el2, m, n) = (double)(float)(l * (l + 1)) / (REARTH * REARTH);
            S2(el4, m, n) = S2(el2, m, n) * S2(el2, m, n);
            S2(trfilt, m, n) = (l <= TRUNC) ? 1.0 : 0.0;
        }
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m) S2(elm2, m, n) = (m == 1 && n == 1) ? 0.0 : 1.0 / S2(el2, m, n);
    for (int m = 1; m <= MX; ++m)
        for (int n = 1; n <= NX; ++n) {
            int m1 = m - 1, m2 = m1 + 1;
            double el1 = (double)(float)(m1 + n - 1);
            if (n == 1) {
                t->gradx[m - 1] = (double)(float)m1 / REARTH;
                S2(uvdx, m, 1) = -REARTH / (double)(float)(m1 + 1);
                S2(uvdym, m, 1) = 0.0;
                S2(vddym, m, 1) = 0.0;
                S2(gradym, m, 1) = 0.0; /* never set by the reference */
            } else {
                S2(uvdx, m, n) = -REARTH * (double)(float)m1 / (el1 * (el1 + 1));
                S2(gradym, m, n) = (el1 - 1.0) * EPSI(m2, n) / REARTH;
                S2(uvdym, m, n) = -REARTH * EPSI(m2, n) / el1;
                S2(vddym, m, n) = (el1 + 1) * EPSI(m2, n) / REARTH;
            }
            S2(gradyp, m, n) = (el1 + 2.0) * EPSI(m2, n + 1) / REARTH;
            S2(uvdyp, m, n) = -REARTH * EPSI(m2, n + 1) / (el1 + 1.0);
            S2(vddyp, m, n) = el1 * EPSI(m2, n + 1) / REARTH;
        }
}

static void init_fband(orc_tables *t) { /* longwave_radiation.f90:208-232 ; fband(100:400, 4) */
    static const double epslw = 0.05f;
    double eps1 = 1.0 - epslw;
#define FB(T, b) t->fband[((T)-100) + 301 * ((b)-1)]
    for (int jt = 200; jt <= 320; ++jt) {
        /* integer**2 then mixed real(4)*integer -> single precision products, widened on use */
        FB(jt, 2) = (double)(0.148f - 3.0e-6f * (float)((jt - 247) * (jt - 247))) * eps1;
        FB(jt, 3) = (double)(0.356f - 5.2e-6f * (float)((jt - 282) * (jt - 282))) * eps1;
        FB(jt, 4) = (double)(0.314f + 1.0e-5f * (float)((jt - 315) * (jt - 315))) * eps1;
        FB(jt, 1) = eps1 - (FB(jt, 2) + FB(jt, 3) + FB(jt, 4));
    }
    for (int jb = 1; jb <= 4; ++jb) {
        for (int jt = 100; jt <= 199; ++jt) FB(jt, jb) = FB(200, jb);
        for (int jt = 321; jt <= 400; ++jt) FB(jt, jb) = FB(320, jb);
    }
#undef FB
}

void orc_tables_init(orc_tables *t) {
    memset(t, 0, sizeof *t);
    init_geometry(t);
    init_legendre(t);
    fft_init(IX, t->work, t->ifac);
    init_spectral(t);
    init_fband(t);
}

/* ------------------------------------------------------------------------------------------
 * Legendre transforms, legendre.f90:130-221.  Real views in(2*mx, .), 1-based macros.
 * ---------------------------------------------------------------------------------------- */
void orc_legendre_inv(const orc_tables *t, const double *in, double *out) {
    double even[2 * MX], odd[2 * MX];
    for (int j = 1; j <= IY; ++j) {
        int j1 = IL + 1 - j;
        memset(even, 0, sizeof even);
        memset(odd, 0, sizeof odd);
        for (int n = 1; n <= NX; n += 2)
            for (int m = 1; m <= t->nsh2[n - 1]; ++m) even[m - 1] = even[m - 1] + in[(m - 1) + 2 * MX * (n - 1)] * CPOL(m, n, j);
        for (int n = 2; n <= NX; n += 2)
            for (int m = 1; m <= t->nsh2[n - 1]; ++m) odd[m - 1] = odd[m - 1] + in[(m - 1) + 2 * MX * (n - 1)] * CPOL(m, n, j);
        for (int m = 0; m < 2 * MX; ++m) {
            out[m + 2 * MX * (j1 - 1)] = even[m] + odd[m];
            out[m + 2 * MX * (j - 1)] = even[m] - odd[m];
        }
    }
}

void orc_legendre(const orc_tables *t, const double *in, double *out) {
    double even[2 * MX * IY], odd[2 * MX * IY];
    memset(out, 0, sizeof(double) * 2 * MX * NX);
    for (int j = 1; j <= IY; ++j) {
        int j1 = IL + 1 - j;
        for (int m = 0; m < 2 * MX; ++m) {
            even[m + 2 * MX * (j - 1)] = (in[m + 2 * MX * (j1 - 1)] + in[m + 2 * MX * (j - 1)]) * t->wt[j - 1];
            odd[m + 2 * MX * (j - 1)] = (in[m + 2 * MX * (j1 - 1)] - in[m + 2 * MX * (j - 1)]) * t->wt[j - 1];
        }
    }
    for (int n = 1; n <= TRUNC + 1; ++n) {
        const double *src = (n % 2 == 1) ? even : odd;
        for (int m = 1; m <= t->nsh2[n - 1]; ++m) {
            double s = 0.0;
            for (int j = 1; j <= IY; ++j) s = s + CPOL(m, n, j) * src[(m - 1) + 2 * MX * (j - 1)];
            out[(m - 1) + 2 * MX * (n - 1)] = s;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Fourier transforms, fourier.f90:63-123
 * ---------------------------------------------------------------------------------------- */
void orc_fourier_inv(const orc_tables *t, const double *in, double *out, int kcos) {
    double fvar[IX];
    for (int j = 0; j < IL; ++j) {
        const double *row = in + 2 * MX * j;
        fvar[0] = row[0];
        for (int m = 3; m <= 2 * MX; ++m) fvar[m - 2] = row[m - 1];
        for (int m = 2 * MX; m <= IX; ++m) fvar[m - 1] = 0.0;
        orc_rfftb96(t, fvar);
        if (kcos == 1)
            for (int i = 0; i < IX; ++i) out[i + IX * j] = fvar[i];
        else
            for (int i = 0; i < IX; ++i) out[i + IX * j] = fvar[i] * t->cosgr[j];
    }
}

void orc_fourier(const orc_tables *t, const double *in, double *out) {
    double fvar[IX];
    const double scale = (double)(1.0f / (float)IX); /* fourier.f90:113 */
    for (int j = 0; j < IL; ++j) {
        memcpy(fvar, in + IX * j, sizeof fvar);
        orc_rfftf96(t, fvar);
        double *row = out + 2 * MX * j;
        row[0] = fvar[0] * scale;
        row[1] = 0.0;
        for (int m = 3; m <= 2 * MX; ++m) row[m - 1] = fvar[m - 2] * scale;
    }
}

void orc_spec2grid(const orc_tables *t, const double *spec, double *grid, int kcos) { /* spectral.f90:251-261 */
    double four[2 * MX * IL];
    orc_legendre_inv(t, spec, four);
    orc_fourier_inv(t, four, grid, kcos);
}

void orc_grid2spec(const orc_tables *t, const double *grid, double *spec) { /* spectral.f90:263-273 */
    double four[2 * MX * IL];
    orc_fourier(t, grid, four);
    orc_legendre(t, four, spec);
}

void orc_spec2grid_batch(const orc_tables *t, const double *spec, double *grid, int kcos, int nfields) {
    for (int f = 0; f < nfields; ++f) orc_spec2grid(t, spec + (long)f * 2 * MX * NX, grid + (long)f * IX * IL, kcos);
}

void orc_grid2spec_batch(const orc_tables *t, const double *grid, double *spec, int nfields) {
    for (int f = 0; f < nfields; ++f) orc_grid2spec(t, grid + (long)f * IX * IL, spec + (long)f * 2 * MX * NX);
}

/* ------------------------------------------------------------------------------------------
 * Spectral-space operators, spectral.f90:134-317.  complex z(m,n) -> re at 2*idx, im at 2*idx+1.
 * Complex * real and complex * (0,1) follow the Fortran evaluation order; multiplying by the
 * imaginary unit is the full complex product (a+ib)(0+1i) = (a*0 - b*1) + (a*1 + b*0)i.
 * ---------------------------------------------------------------------------------------- */
#define IDX(m, n) (((m)-1) + MX * ((n)-1))
#define RE(z, m, n) (z)[2 * IDX(m, n)]
#define IM(z, m, n) (z)[2 * IDX(m, n) + 1]

static inline void times_i(double a, double b, double *re, double *im) {
    *re = a * 0.0 - b * 1.0;
    *im = a * 1.0 + b * 0.0;
}

void orc_vort2vel(const orc_tables *t, const double *vor, const double *div, double *ucos, double *vcos) {
    double zp[2 * MX * NX], zc[2 * MX * NX]; /* spectral.f90:198-199 */
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m) {
            double u = S2(uvdx, m, n);
            times_i(u * RE(vor, m, n), u * IM(vor, m, n), &RE(zp, m, n), &IM(zp, m, n));
            times_i(u * RE(div, m, n), u * IM(div, m, n), &RE(zc, m, n), &IM(zc, m, n));
        }
    for (int m = 1; m <= MX; ++m)
        for (int c = 0; c < 2; ++c) { /* c: 0 = real part, 1 = imaginary part */
#define Z(z, mm, nn) (z)[2 * IDX(mm, nn) + c]
            Z(ucos, m, 1) = Z(zc, m, 1) - S2(uvdyp, m, 1) * Z(vor, m, 2);
            Z(ucos, m, NX) = S2(uvdym, m, NX) * Z(vor, m, TRUNC + 1);
            Z(vcos, m, 1) = Z(zp, m, 1) + S2(uvdyp, m, 1) * Z(div, m, 2);
            Z(vcos, m, NX) = -S2(uvdym, m, NX) * Z(div, m, TRUNC + 1);
        }
    for (int n = 2; n <= TRUNC + 1; ++n)
        for (int m = 1; m <= MX; ++m)
            for (int c = 0; c < 2; ++c) {
                Z(vcos, m, n) = -S2(uvdym, m, n) * Z(div, m, n - 1) + S2(uvdyp, m, n) * Z(div, m, n + 1) + Z(zp, m, n);
                Z(ucos, m, n) = S2(uvdym, m, n) * Z(vor, m, n - 1) - S2(uvdyp, m, n) * Z(vor, m, n + 1) + Z(zc, m, n);
            }
}

void orc_vel2vort(const orc_tables *t, const double *ucos, const double *vcos, double *vor, double *div) {
    double zp[2 * MX * NX], zc[2 * MX * NX]; /* spectral.f90:168-171 */
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m) {
            double g = t->gradx[m - 1];
            times_i(g * RE(ucos, m, n), g * IM(ucos, m, n), &RE(zp, m, n), &IM(zp, m, n));
            times_i(g * RE(vcos, m, n), g * IM(vcos, m, n), &RE(zc, m, n), &IM(zc, m, n));
        }
    for (int m = 1; m <= MX; ++m)
        for (int c = 0; c < 2; ++c) {
            Z(vor, m, 1) = Z(zc, m, 1) - S2(vddyp, m, 1) * Z(ucos, m, 2);
            Z(vor, m, NX) = S2(vddym, m, NX) * Z(ucos, m, TRUNC + 1);
            Z(div, m, 1) = Z(zp, m, 1) + S2(vddyp, m, 1) * Z(vcos, m, 2);
            Z(div, m, NX) = -S2(vddym, m, NX) * Z(vcos, m, TRUNC + 1);
        }
    for (int n = 2; n <= TRUNC + 1; ++n)
        for (int m = 1; m <= MX; ++m)
            for (int c = 0; c < 2; ++c) {
                Z(vor, m, n) = S2(vddym, m, n) * Z(ucos, m, n - 1) - S2(vddyp, m, n) * Z(ucos, m, n + 1) + Z(zc, m, n);
                Z(div, m, n) = -S2(vddym, m, n) * Z(vcos, m, n - 1) + S2(vddyp, m, n) * Z(vcos, m, n + 1) + Z(zp, m, n);
            }
}

void orc_grid_vel2vort(const orc_tables *t, const double *ug, const double *vg, double *vor, double *div, int kcos) {
    double ug1[IX * IL], vg1[IX * IL], su[2 * MX * NX], sv[2 * MX * NX]; /* spectral.f90:218-248 */
    const double *scale = (kcos == 2) ? t->cosgr : t->cosgr2;
    for (int j = 0; j < IL; ++j)
        for (int i = 0; i < IX; ++i) {
            ug1[i + IX * j] = ug[i + IX * j] * scale[j];
            vg1[i + IX * j] = vg[i + IX * j] * scale[j];
        }
    orc_grid2spec(t, ug1, su);
    orc_grid2spec(t, vg1, sv);
    orc_vel2vort(t, su, sv, vor, div);
}

void orc_gradient(const orc_tables *t, const double *psi, double *psdx, double *psdy) { /* spectral.f90:275-296 */
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m) {
            double g = t->gradx[m - 1];
            times_i(g * RE(psi, m, n), g * IM(psi, m, n), &RE(psdx, m, n), &IM(psdx, m, n));
        }
    for (int m = 1; m <= MX; ++m)
        for (int c = 0; c < 2; ++c) {
            Z(psdy, m, 1) = S2(gradyp, m, 1) * Z(psi, m, 2);
            Z(psdy, m, NX) = -S2(gradym, m, NX) * Z(psi, m, TRUNC + 1);
        }
    for (int n = 2; n <= TRUNC + 1; ++n)
        for (int m = 1; m <= MX; ++m)
            for (int c = 0; c < 2; ++c)
                Z(psdy, m, n) = -S2(gradym, m, n) * Z(psi, m, n - 1) + S2(gradyp, m, n) * Z(psi, m, n + 1);
}

void orc_laplacian(const orc_tables *t, const double *in, double *out, int inverse) { /* spectral.f90:140-155 */
    const double *tab = inverse ? t->elm2 : t->el2;
    for (int k = 0; k < MX * NX; ++k) {
        out[2 * k] = -in[2 * k] * tab[k];
        out[2 * k + 1] = -in[2 * k + 1] * tab[k];
    }
}

void orc_truncate(const orc_tables *t, double *f) { /* spectral.f90:134-138 */
    for (int k = 0; k < MX * NX; ++k) {
        f[2 * k] = f[2 * k] * t->trfilt[k];
        f[2 * k + 1] = f[2 * k + 1] * t->trfilt[k];
    }
}

void orc_grid_filter(const orc_tables *t, const double *fg1, double *fg2) { /* spectral.f90:299-317 */
    double fsp[2 * MX * NX];
    orc_grid2spec(t, fg1, fsp);
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m)
            if (m + n - 2 > TRUNC) RE(fsp, m, n) = IM(fsp, m, n) = 0.0;
    orc_spec2grid(t, fsp, fg2, 1);
}
