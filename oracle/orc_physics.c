/* TEST INFRASTRUCTURE -- CPU oracle (see speedy_oracle.h).  NOT PART OF THE PRODUCT.
 *
 * Column physics of the reference restated in plain C, array-at-a-time like the Fortran (one member,
 * (ix, il[, kx]) column-major arrays), same operation order, same single-precision-seeded constants.
 * Default-real literals of the reference appear here as `float` literals: C's usual arithmetic
 * conversions then reproduce Fortran's mixed-kind rules (float op float stays float, float op double
 * widens the float operand first).
 */
#include "speedy_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IX ORC_IX
#define IL ORC_IL
#define KX ORC_KX
#define NG (IX * IL)

/* 3-D (ix,il,kx) and higher arrays are addressed by (point p = i + ix*j, level k 1-based) */
#define A3(a, p, k) (a)[(p) + NG * ((k)-1)]
#define A4(a, p, k, b) (a)[(p) + NG * (((k)-1) + KX * ((b)-1))]
#define WVI(k, c) t->wvi[((k)-1) + 8 * ((c)-1)]
#define FBAND(T, b) t->fband[((T)-100) + 301 * ((b)-1)]

/* physical_constants.f90:16-30, mod_radcon.f90:11-16 */
static const double P0 = 1.e+5f, CP = 1004.0f, GRAV = 9.81f, ALHC = 2501.0f, SBC = 5.67e-8f;
#define AKAP ((double)(2.0f / 7.0f))
#define RGAS (AKAP * CP)
static const double EPSLW = 0.05f, EMISFC = 0.98f;

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }
/* Real-exponent powers as the flang -O2 build of the reference evaluates them (established by bitwise comparison with
 * the reference library): x**3.0 becomes (x*x)*x, x**4.0 stays a libm pow() call (correctly rounded in glibc). */
static inline double pow3(double x) { return (x * x) * x; }
static inline double pow4(double x) { return pow(x, 4.0); }
static inline int nint_(double x) { return (int)lround(x); } /* Fortran nint: half away from zero */

/* ---------------------------------------------------------------- humidity.f90:44-78 */
void orc_qsat(const double *ta, const double *ps, double sig, double *qsat, int n) {
    const double e0 = 6.108e-3, c1 = 17.269f, c2 = 21.875f, t0 = 273.16f, t1 = 35.86f, t2 = 7.66f;
    for (int p = 0; p < n; ++p) {
        if (ta[p] >= t0)
            qsat[p] = e0 * exp(c1 * (ta[p] - t0) / (ta[p] - t1));
        else
            qsat[p] = e0 * exp(c2 * (ta[p] - t0) / (ta[p] - t2));
    }
    if (sig <= 0.0) {
        for (int p = 0; p < n; ++p) qsat[p] = 622.0f * qsat[p] / (ps[0] - 0.378f * qsat[p]);
    } else {
        for (int p = 0; p < n; ++p) qsat[p] = 622.0f * qsat[p] / (sig * ps[p] - 0.378f * qsat[p]);
    }
}

/* ---------------------------------------------------------------- convection.f90:170-253 */
static const double PSMIN = 0.8f, TRCNV = 6.0f, RHBL = 0.9f, RHIL = 0.7f, ENTMAX = 0.5f, SMF = 0.8f;

static void diagnose_convection(const orc_tables *t, const double *psa, const double *se, const double *qa,
                                const double *qsat, int *itop, double *qdif) {
    const int nl1 = KX - 1, nlp = KX + 1;
    double msthr = 0;
    const double rlhc = 1.0 / ALHC;
    double *mss = (double *)malloc(sizeof(double) * NG * (KX + 1));
#define MSS(p, k) mss[(p) + NG * (k)]
    for (int k = 2; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) MSS(p, k) = A3(se, p, k) + ALHC * A3(qsat, p, k);
    for (int i = 0; i < IX; ++i)
        for (int j = 0; j < IL; ++j) { /* i outer, j inner as in the reference: msthr carries over between columns */
            const int p = i + IX * j;
            itop[p] = nlp;
            if (psa[p] > PSMIN) {
                double mse0 = A3(se, p, KX) + ALHC * A3(qa, p, KX);
                double mse1 = A3(se, p, nl1) + ALHC * A3(qa, p, nl1);
                mse1 = dmin(mse0, mse1);
                double mss0 = dmax(mse0, MSS(p, KX));
                int ktop1 = KX, ktop2 = KX;
                for (int k = KX - 3; k >= 3; --k) {
                    double mss2 = MSS(p, k) + WVI(k, 2) * (MSS(p, k + 1) - MSS(p, k));
                    if (mss0 > mss2) ktop1 = k;
                    if (mse1 > mss2) {
                        ktop2 = k;
                        msthr = mss2;
                    }
                }
                if (ktop1 < KX) {
                    double qthr0 = RHBL * A3(qsat, p, KX), qthr1 = RHBL * A3(qsat, p, nl1);
                    int lqthr = (A3(qa, p, KX) > qthr0 && A3(qa, p, nl1) > qthr1);
                    if (ktop2 < KX) {
                        itop[p] = ktop1;
                        qdif[p] = dmax(A3(qa, p, KX) - qthr0, (mse0 - msthr) * rlhc);
                    } else if (lqthr) {
                        itop[p] = ktop1;
                        qdif[p] = A3(qa, p, KX) - qthr0;
                    }
                }
            }
        }
#undef MSS
    free(mss);
}

/* ---------------------------------------------------------------- convection.f90:27-158 */
void orc_convection(const orc_tables *t, const double *psa, const double *se, const double *qa, const double *qsat,
                    int *itop, double *cbmf, double *precnv, double *dfse, double *dfqa) {
    const int nl1 = KX - 1, nlp = KX + 1;
    const double fqmax = 5.0f;
    const double fm0 = P0 * t->dhs[KX - 1] / (GRAV * TRCNV * 3600.0f);
    const double rdps = 2.0f / (1.0f - PSMIN);
    double entr[KX + 1], sentr = 0.0;
    double *qdif = (double *)calloc(NG, sizeof(double));
    memset(dfse, 0, sizeof(double) * NG * KX);
    memset(dfqa, 0, sizeof(double) * NG * KX);
    memset(cbmf, 0, sizeof(double) * NG);
    memset(precnv, 0, sizeof(double) * NG);
    for (int k = 2; k <= nl1; ++k) {
        double d = dmax(0.0f, t->fsg[k - 1] - 0.5f);
        entr[k] = d * d;
        sentr = sentr + entr[k];
    }
    sentr = ENTMAX / sentr;
    for (int k = 2; k <= nl1; ++k) entr[k] = entr[k] * sentr;

    diagnose_convection(t, psa, se, qa, qsat, itop, qdif);

    for (int p = 0; p < NG; ++p) {
        if (itop[p] == nlp) continue;
        int k = KX, k1 = k - 1;
        double qmax = dmax(1.01f * A3(qa, p, k), A3(qsat, p, k));
        double sb = A3(se, p, k1) + WVI(k1, 2) * (A3(se, p, k) - A3(se, p, k1));
        double qb = A3(qa, p, k1) + WVI(k1, 2) * (A3(qa, p, k) - A3(qa, p, k1));
        qb = dmin(qb, A3(qa, p, k));
        double fpsa = psa[p] * dmin(1.0f, (psa[p] - PSMIN) * rdps);
        double fmass = fm0 * fpsa * dmin(fqmax, qdif[p] / (qmax - qb));
        cbmf[p] = fmass;
        double fus = fmass * A3(se, p, k), fuq = fmass * qmax;
        double fds = fmass * sb, fdq = fmass * qb;
        A3(dfse, p, k) = fds - fus;
        A3(dfqa, p, k) = fdq - fuq;
        for (k = KX - 1; k >= itop[p] + 1; --k) {
            k1 = k - 1;
            A3(dfse, p, k) = fus - fds;
            A3(dfqa, p, k) = fuq - fdq;
            double enmass = entr[k] * psa[p] * cbmf[p];
            fmass = fmass + enmass;
            fus = fus + enmass * A3(se, p, k);
            fuq = fuq + enmass * A3(qa, p, k);
            sb = A3(se, p, k1) + WVI(k1, 2) * (A3(se, p, k) - A3(se, p, k1));
            qb = A3(qa, p, k1) + WVI(k1, 2) * (A3(qa, p, k) - A3(qa, p, k1));
            fds = fmass * sb;
            fdq = fmass * qb;
            A3(dfse, p, k) = A3(dfse, p, k) + fds - fus;
            A3(dfqa, p, k) = A3(dfqa, p, k) + fdq - fuq;
            double delq = RHIL * A3(qsat, p, k) - A3(qa, p, k);
            if (delq > 0.0) {
                double fsq = SMF * cbmf[p] * delq;
                A3(dfqa, p, k) = A3(dfqa, p, k) + fsq;
                A3(dfqa, p, KX) = A3(dfqa, p, KX) - fsq;
            }
        }
        k = itop[p];
        double qsatb = A3(qsat, p, k) + WVI(k, 2) * (A3(qsat, p, k + 1) - A3(qsat, p, k));
        precnv[p] = dmax(fuq - fmass * qsatb, 0.0);
        A3(dfse, p, k) = fus - fds + ALHC * precnv[p];
        A3(dfqa, p, k) = fuq - fdq - precnv[p];
    }
    free(qdif);
}

/* ---------------------------------------------------------------- large_scale_condensation.f90:33-96 */
void orc_lsc(const orc_tables *t, const double *psa, const double *qa, const double *qsat, int *itop, double *precls,
             double *dtlsc, double *dqlsc) {
    const double trlsc = 4.0f, rhlsc = 0.9f, drhlsc = 0.1f, rhblsc = 0.95f;
    const double qsmax = 10.0f;
    const double rtlsc = 1.0f / (trlsc * 3600.0f);
    const double tfact = ALHC / CP;
    const double prg = P0 / GRAV;
    for (int p = 0; p < NG; ++p) {
        A3(dtlsc, p, 1) = 0.0;
        A3(dqlsc, p, 1) = 0.0;
        precls[p] = 0.0;
    }
    for (int k = 2; k <= KX; ++k) {
        double sig2 = t->fsg[k - 1] * t->fsg[k - 1];
        double rhref = rhlsc + drhlsc * (sig2 - 1.0f);
        if (k == KX) rhref = dmax(rhref, rhblsc);
        double dqmax = qsmax * sig2 * rtlsc;
        for (int p = 0; p < NG; ++p) {
            double dqa = rhref * A3(qsat, p, k) - A3(qa, p, k);
            if (dqa < 0.0) {
                itop[p] = k < itop[p] ? k : itop[p];
                A3(dqlsc, p, k) = dqa * rtlsc;
                A3(dtlsc, p, k) = tfact * dmin(-A3(dqlsc, p, k), dqmax * (psa[p] * psa[p]));
            } else {
                A3(dqlsc, p, k) = 0.0;
                A3(dtlsc, p, k) = 0.0;
            }
        }
    }
    for (int k = 2; k <= KX; ++k) {
        double pfact = t->dhs[k - 1] * prg;
        for (int p = 0; p < NG; ++p) precls[p] = precls[p] - pfact * A3(dqlsc, p, k);
    }
    for (int p = 0; p < NG; ++p) precls[p] = precls[p] * psa[p];
}

/* ---------------------------------------------------------------- shortwave_radiation.f90:325-404 */
static const double SOLC = 342.0f, RHCL1 = 0.30f, RHCL2 = 1.00f, QACL = 0.20f, WPCL = 0.2f, PMAXCL = 10.0f,
                    CLSMAX = 0.60f, CLSMINL = 0.15f, GSE_S0 = 0.25f, GSE_S1 = 0.40f, ALBCL = 0.43f, ALBCLS = 0.50f,
                    EPSSW = 0.020f, ABSDRY = 0.033f, ABSAER = 0.033f, ABSWV1 = 0.022f, ABSWV2 = 15.000f,
                    ABSCL1 = 0.015f, ABSCL2 = 0.15f, ABLWIN = 0.3f, ABLWV1 = 0.7f, ABLWV2 = 50.0f, ABLCL1 = 12.0f,
                    ABLCL2 = 0.6f;

void orc_clouds(const double *qa, const double *rh, const double *precnv, const double *precls, const int *iptop,
                const double *gse, const double *fmask, int *icltop, double *cloudc, double *clstr,
                double *qcloud_equiv) {
    const int nl1 = KX - 1, nlp = KX + 1;
    const double rrcl = 1.f / (RHCL2 - RHCL1);
    (void)SOLC; (void)EPSSW;
    for (int p = 0; p < NG; ++p) {
        if (A3(rh, p, nl1) > RHCL1) {
            cloudc[p] = A3(rh, p, nl1) - RHCL1;
            icltop[p] = nl1;
        } else {
            cloudc[p] = 0.0;
            icltop[p] = nlp;
        }
    }
    for (int k = 3; k <= KX - 2; ++k)
        for (int p = 0; p < NG; ++p) {
            double drh = A3(rh, p, k) - RHCL1;
            if (drh > cloudc[p] && A3(qa, p, k) > QACL) {
                cloudc[p] = drh;
                icltop[p] = k;
            }
        }
    for (int p = 0; p < NG; ++p) {
        double pr1 = dmin(PMAXCL, 86.4f * (precnv[p] + precls[p]));
        double c = dmin(1.0f, cloudc[p] * rrcl);
        cloudc[p] = dmin(1.0f, WPCL * sqrt(pr1) + c * c);
        icltop[p] = iptop[p] < icltop[p] ? iptop[p] : icltop[p];
    }
    for (int p = 0; p < NG; ++p) qcloud_equiv[p] = A3(qa, p, nl1);
    const double clfact = 1.2f;
    const double rgse = 1.0f / (GSE_S1 - GSE_S0);
    for (int p = 0; p < NG; ++p) {
        double fstab = dmax(0.0f, dmin(1.0f, rgse * (gse[p] - GSE_S0)));
        clstr[p] = fstab * dmax(CLSMAX - clfact * cloudc[p], 0.0f);
        double clstrl = dmax(clstr[p], CLSMINL) * A3(rh, p, KX);
        clstr[p] = clstr[p] + fmask[p] * (clstrl - clstr[p]);
    }
}

/* ---------------------------------------------------------------- shortwave_radiation.f90:50-214 */
void orc_shortwave(const orc_tables *t, orc_phys_io *io, const double *psa, const double *qa, const int *icltop,
                   const double *cloudc, const double *clstr) {
    const int nl1 = KX - 1;
    const double fband2 = 0.05f, fband1 = 1.0f - fband2;
    double *tau2 = io->rad_tau2, *flux = io->rad_flux, *ttr = io->tt_rsw;
    double *acloud = (double *)malloc(sizeof(double) * NG), *psaz = (double *)malloc(sizeof(double) * NG);
#define TAU(p, k, b) A4(tau2, p, k, b)
#define FLX(p, b) flux[(p) + NG * ((b)-1)]
    memset(tau2, 0, sizeof(double) * NG * KX * 4);
    for (int p = 0; p < NG; ++p) {
        if (icltop[p] <= KX) TAU(p, icltop[p], 3) = ALBCL * cloudc[p];
        TAU(p, KX, 3) = ALBCLS * clstr[p];
    }
    for (int p = 0; p < NG; ++p) {
        psaz[p] = psa[p] * io->zenit_correction[p];
        acloud[p] = cloudc[p] * dmin(ABSCL1 * io->qcloud_equiv[p], ABSCL2);
        TAU(p, 1, 1) = exp(-psaz[p] * t->dhs[0] * ABSDRY);
    }
    for (int k = 2; k <= nl1; ++k) {
        double abs1 = ABSDRY + ABSAER * (t->fsg[k - 1] * t->fsg[k - 1]);
        for (int p = 0; p < NG; ++p) {
            if (k >= icltop[p])
                TAU(p, k, 1) = exp(-psaz[p] * t->dhs[k - 1] * (abs1 + ABSWV1 * A3(qa, p, k) + acloud[p]));
            else
                TAU(p, k, 1) = exp(-psaz[p] * t->dhs[k - 1] * (abs1 + ABSWV1 * A3(qa, p, k)));
        }
    }
    {
        double abs1 = ABSDRY + ABSAER * (t->fsg[KX - 1] * t->fsg[KX - 1]);
        for (int p = 0; p < NG; ++p) TAU(p, KX, 1) = exp(-psaz[p] * t->dhs[KX - 1] * (abs1 + ABSWV1 * A3(qa, p, KX)));
    }
    for (int k = 2; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) TAU(p, k, 2) = exp(-psaz[p] * t->dhs[k - 1] * ABSWV2 * A3(qa, p, k));

    for (int p = 0; p < NG; ++p) {
        io->tsr[p] = io->flux_solar_in[p];
        FLX(p, 1) = io->flux_solar_in[p] * fband1;
        FLX(p, 2) = io->flux_solar_in[p] * fband2;
        /* stratosphere, k = 1 and 2 */
        A3(ttr, p, 1) = FLX(p, 1);
        FLX(p, 1) = TAU(p, 1, 1) * (FLX(p, 1) - io->flux_ozone_upper[p] * psa[p]);
        A3(ttr, p, 1) = A3(ttr, p, 1) - FLX(p, 1);
        A3(ttr, p, 2) = FLX(p, 1);
        FLX(p, 1) = TAU(p, 2, 1) * (FLX(p, 1) - io->flux_ozone_lower[p] * psa[p]);
        A3(ttr, p, 2) = A3(ttr, p, 2) - FLX(p, 1);
    }
    for (int k = 3; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) {
            TAU(p, k, 3) = FLX(p, 1) * TAU(p, k, 3);
            FLX(p, 1) = FLX(p, 1) - TAU(p, k, 3);
            A3(ttr, p, k) = FLX(p, 1);
            FLX(p, 1) = TAU(p, k, 1) * FLX(p, 1);
            A3(ttr, p, k) = A3(ttr, p, k) - FLX(p, 1);
        }
    for (int k = 2; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) {
            A3(ttr, p, k) = A3(ttr, p, k) + FLX(p, 2);
            FLX(p, 2) = TAU(p, k, 2) * FLX(p, 2);
            A3(ttr, p, k) = A3(ttr, p, k) - FLX(p, 2);
        }
    for (int p = 0; p < NG; ++p) {
        io->ssrd[p] = FLX(p, 1) + FLX(p, 2);
        FLX(p, 1) = FLX(p, 1) * io->alb_surface[p];
        io->ssr[p] = io->ssrd[p] - FLX(p, 1);
    }
    for (int k = KX; k >= 1; --k)
        for (int p = 0; p < NG; ++p) {
            A3(ttr, p, k) = A3(ttr, p, k) + FLX(p, 1);
            FLX(p, 1) = TAU(p, k, 1) * FLX(p, 1);
            A3(ttr, p, k) = A3(ttr, p, k) - FLX(p, 1);
            FLX(p, 1) = FLX(p, 1) + TAU(p, k, 3);
        }
    for (int p = 0; p < NG; ++p) io->tsr[p] = io->tsr[p] - FLX(p, 1);

    /* 5. longwave transmissivities */
    const double co2 = io->air_absortivity_co2;
    for (int p = 0; p < NG; ++p) {
        TAU(p, 1, 1) = exp(-psa[p] * t->dhs[0] * ABLWIN);
        TAU(p, 1, 2) = exp(-psa[p] * t->dhs[0] * co2);
        TAU(p, 1, 3) = 1.0;
        TAU(p, 1, 4) = 1.0;
    }
    for (int k = 2; k <= KX; k += KX - 2)
        for (int p = 0; p < NG; ++p) {
            TAU(p, k, 1) = exp(-psa[p] * t->dhs[k - 1] * ABLWIN);
            TAU(p, k, 2) = exp(-psa[p] * t->dhs[k - 1] * co2);
            TAU(p, k, 3) = exp(-psa[p] * t->dhs[k - 1] * ABLWV1 * A3(qa, p, k));
            TAU(p, k, 4) = exp(-psa[p] * t->dhs[k - 1] * ABLWV2 * A3(qa, p, k));
        }
    for (int p = 0; p < NG; ++p) acloud[p] = cloudc[p] * ABLCL2;
    for (int k = 3; k <= nl1; ++k)
        for (int p = 0; p < NG; ++p) {
            double deltap = psa[p] * t->dhs[k - 1];
            double acloud1 = (k < icltop[p]) ? acloud[p] : ABLCL1 * cloudc[p];
            TAU(p, k, 1) = exp(-deltap * (ABLWIN + acloud1));
            TAU(p, k, 2) = exp(-deltap * co2);
            TAU(p, k, 3) = exp(-deltap * dmax(ABLWV1 * A3(qa, p, k), acloud[p]));
            TAU(p, k, 4) = exp(-deltap * dmax(ABLWV2 * A3(qa, p, k), acloud[p]));
        }
    const double eps1 = EPSLW / (t->dhs[0] + t->dhs[1]);
    for (int p = 0; p < NG; ++p) {
        io->rad_strat_corr[p] = io->stratospheric_correction[p] * psa[p];
        io->rad_strat_corr[p + NG] = eps1 * psa[p];
    }
#undef TAU
#undef FLX
    free(acloud);
    free(psaz);
}

/* ---------------------------------------------------------------- longwave_radiation.f90:16-121 */
void orc_lw_down(const orc_tables *t, const double *ta, double *fsfcd, double *dfabs, double *rad_flux,
                 const double *rad_tau2, double *rad_st4a) {
    const int nl1 = KX - 1, nband = 4;
    const double anis = 1.0f;
#define ST4(p, k, c) rad_st4a[(p) + NG * (((k)-1) + KX * ((c)-1))]
#define TAU(p, k, b) A4(rad_tau2, p, k, b)
#define FLX(p, b) rad_flux[(p) + NG * ((b)-1)]
    for (int k = 1; k <= nl1; ++k)
        for (int p = 0; p < NG; ++p) ST4(p, k, 1) = A3(ta, p, k) + WVI(k, 2) * (A3(ta, p, k + 1) - A3(ta, p, k));
    for (int p = 0; p < NG; ++p) {
        ST4(p, 1, 2) = 0.75f * A3(ta, p, 1) + 0.25f * ST4(p, 1, 1);
        ST4(p, 2, 2) = 0.50f * A3(ta, p, 2) + 0.25f * (ST4(p, 1, 1) + ST4(p, 2, 1));
    }
    for (int k = 3; k <= nl1; ++k)
        for (int p = 0; p < NG; ++p) ST4(p, k, 2) = 0.5f * anis * dmax(ST4(p, k, 1) - ST4(p, k - 1, 1), 0.0f);
    for (int p = 0; p < NG; ++p) ST4(p, KX, 2) = anis * dmax(A3(ta, p, KX) - ST4(p, nl1, 1), 0.0f);
    for (int k = 1; k <= 2; ++k)
        for (int p = 0; p < NG; ++p) {
            ST4(p, k, 1) = SBC * pow4(ST4(p, k, 2));
            ST4(p, k, 2) = 0.0;
        }
    for (int k = 3; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) {
            double st3a = SBC * pow3(A3(ta, p, k));
            ST4(p, k, 1) = st3a * A3(ta, p, k);
            ST4(p, k, 2) = 4.0f * st3a * ST4(p, k, 2);
        }
    memset(fsfcd, 0, sizeof(double) * NG);
    memset(dfabs, 0, sizeof(double) * NG * KX);
    for (int jb = 1; jb <= 2; ++jb)
        for (int p = 0; p < NG; ++p) {
            double emis = 1.0f - TAU(p, 1, jb);
            double brad = FBAND(nint_(A3(ta, p, 1)), jb) * (ST4(p, 1, 1) + emis * ST4(p, 1, 2));
            FLX(p, jb) = emis * brad;
            A3(dfabs, p, 1) = A3(dfabs, p, 1) - FLX(p, jb);
        }
    for (int jb = 3; jb <= nband; ++jb)
        for (int p = 0; p < NG; ++p) FLX(p, jb) = 0.0;
    for (int jb = 1; jb <= nband; ++jb)
        for (int k = 2; k <= KX; ++k)
            for (int p = 0; p < NG; ++p) {
                double emis = 1.0f - TAU(p, k, jb);
                double brad = FBAND(nint_(A3(ta, p, k)), jb) * (ST4(p, k, 1) + emis * ST4(p, k, 2));
                A3(dfabs, p, k) = A3(dfabs, p, k) + FLX(p, jb);
                FLX(p, jb) = TAU(p, k, jb) * FLX(p, jb) + emis * brad;
                A3(dfabs, p, k) = A3(dfabs, p, k) - FLX(p, jb);
            }
    for (int jb = 1; jb <= nband; ++jb)
        for (int p = 0; p < NG; ++p) fsfcd[p] = fsfcd[p] + EMISFC * FLX(p, jb);
    for (int p = 0; p < NG; ++p) {
        double corlw = EPSLW * EMISFC * ST4(p, KX, 1);
        A3(dfabs, p, KX) = A3(dfabs, p, KX) - corlw;
        fsfcd[p] = fsfcd[p] + corlw;
    }
}

/* ---------------------------------------------------------------- longwave_radiation.f90:124-205 */
void orc_lw_up(const orc_tables *t, const double *ta, const double *ts, const double *fsfcd, const double *fsfcu,
               double *fsfc, double *ftop, double *dfabs, double *rad_flux, const double *rad_tau2,
               const double *rad_st4a, const double *rad_strat_corr) {
    const int nband = 4;
    const double refsfc = 1.0f - EMISFC;
    for (int p = 0; p < NG; ++p) fsfc[p] = fsfcu[p] - fsfcd[p];
    for (int jb = 1; jb <= nband; ++jb)
        for (int p = 0; p < NG; ++p) FLX(p, jb) = FBAND(nint_(ts[p]), jb) * fsfcu[p] + refsfc * FLX(p, jb);
    for (int p = 0; p < NG; ++p) A3(dfabs, p, KX) = A3(dfabs, p, KX) + EPSLW * fsfcu[p];
    for (int jb = 1; jb <= nband; ++jb)
        for (int k = KX; k >= 2; --k)
            for (int p = 0; p < NG; ++p) {
                double emis = 1.0f - TAU(p, k, jb);
                double brad = FBAND(nint_(A3(ta, p, k)), jb) * (ST4(p, k, 1) - emis * ST4(p, k, 2));
                A3(dfabs, p, k) = A3(dfabs, p, k) + FLX(p, jb);
                FLX(p, jb) = TAU(p, k, jb) * FLX(p, jb) + emis * brad;
                A3(dfabs, p, k) = A3(dfabs, p, k) - FLX(p, jb);
            }
    for (int jb = 1; jb <= 2; ++jb)
        for (int p = 0; p < NG; ++p) {
            double emis = 1.0f - TAU(p, 1, jb);
            double brad = FBAND(nint_(A3(ta, p, 1)), jb) * (ST4(p, 1, 1) - emis * ST4(p, 1, 2));
            A3(dfabs, p, 1) = A3(dfabs, p, 1) + FLX(p, jb);
            FLX(p, jb) = TAU(p, 1, jb) * FLX(p, jb) + emis * brad;
            A3(dfabs, p, 1) = A3(dfabs, p, 1) - FLX(p, jb);
        }
    for (int p = 0; p < NG; ++p) {
        double corlw1 = t->dhs[0] * rad_strat_corr[p + NG] * ST4(p, 1, 1) + rad_strat_corr[p];
        double corlw2 = t->dhs[1] * rad_strat_corr[p + NG] * ST4(p, 2, 1);
        A3(dfabs, p, 1) = A3(dfabs, p, 1) - corlw1;
        A3(dfabs, p, 2) = A3(dfabs, p, 2) - corlw2;
        ftop[p] = corlw1 + corlw2;
    }
    for (int jb = 1; jb <= nband; ++jb)
        for (int p = 0; p < NG; ++p) ftop[p] = ftop[p] + FLX(p, jb);
#undef ST4
#undef TAU
#undef FLX
}

/* ---------------------------------------------------------------- surface_fluxes.f90:40-320 (lfluxland = .true.) */
void orc_surface_fluxes(const orc_tables *t, const double *psa, const double *ua, const double *va, const double *ta,
                        const double *qa, const double *rh, const double *phi, const double *phi0,
                        const double *fmask, const double *forog, const double *tsea, const double *ssrd,
                        const double *slrd, double *ustr, double *vstr, double *shf, double *evap, double *slru,
                        double *hfluxn, double *tsfc, double *tskin, double *u0, double *v0, double *t0,
                        const double *alb_land, const double *alb_sea, const double *snowc, const double *land_temp,
                        const double *soil_avail_water) {
    const double fwind0 = 0.95f, ftemp0 = 1.0f, fhum0 = 0.0f, cdl = 2.4e-3f, cds = 1.0e-3f, chl = 1.2e-3f,
                 chs = 0.9e-3f, vgust = 5.0f, ctday = 1.0e-2f, dtheta = 3.0f, fstab = 0.67f, clambda = 7.0f,
                 clambsn = 7.0f;
    const int nl1 = KX - 1;
    const double esbc = EMISFC * SBC;
    const double ghum0 = 1.0f - fhum0;
    (void)ghum0; (void)rh;
    double *buf = (double *)malloc(sizeof(double) * NG * 14);
    double *t1l = buf, *t1s = buf + NG, *t2l = buf + 2 * NG, *t2s = buf + 3 * NG, *q1l = buf + 4 * NG,
           *q1s = buf + 5 * NG, *qs0l = buf + 6 * NG, *qs0s = buf + 7 * NG, *den0 = buf + 8 * NG, *den1 = buf + 9 * NG,
           *den2 = buf + 10 * NG, *dtskin = buf + 11 * NG, *one = buf + 12 * NG, *tmp = buf + 13 * NG;
#define P3(a, p, c) (a)[(p) + NG * ((c)-1)]
    const double gtemp0 = 1.0f - ftemp0, rcp = 1.0f / CP;
    for (int p = 0; p < NG; ++p) {
        u0[p] = fwind0 * A3(ua, p, KX);
        v0[p] = fwind0 * A3(va, p, KX);
    }
    for (int p = 0; p < NG; ++p) {
        double dt1 = WVI(KX, 2) * (A3(ta, p, KX) - A3(ta, p, nl1));
        t1l[p] = A3(ta, p, KX) + dt1;
        t1s[p] = t1l[p] - phi0[p] * dt1 / (RGAS * 288.0f * t->sigl[KX - 1]);
        t2s[p] = A3(ta, p, KX) + rcp * A3(phi, p, KX);
        t2l[p] = t2s[p] - rcp * phi0[p];
    }
    for (int p = 0; p < NG; ++p) {
        if (A3(ta, p, KX) > A3(ta, p, nl1)) {
            t1l[p] = ftemp0 * t1l[p] + gtemp0 * t2l[p];
            t1s[p] = ftemp0 * t1s[p] + gtemp0 * t2s[p];
        } else {
            t1l[p] = A3(ta, p, KX);
            t1s[p] = A3(ta, p, KX);
        }
        t0[p] = t1s[p] + fmask[p] * (t1l[p] - t1s[p]);
    }
    for (int p = 0; p < NG; ++p)
        den0[p] = (P0 * psa[p] / (RGAS * t0[p])) * sqrt(u0[p] * u0[p] + v0[p] * v0[p] + vgust * vgust);
    for (int j = 0; j < IL; ++j)
        for (int i = 0; i < IX; ++i) {
            int p = i + IX * j;
            tskin[p] = land_temp[p] + ctday * sqrt(t->coa[j]) * ssrd[p] * (1.0f - alb_land[p]) * psa[p];
        }
    double rdth = fstab / dtheta;
    const double astab = 0.5f;
    for (int p = 0; p < NG; ++p) {
        double dthl;
        if (tskin[p] > t2l[p])
            dthl = dmin(dtheta, tskin[p] - t2l[p]);
        else
            dthl = dmax(-dtheta, astab * (tskin[p] - t2l[p]));
        den1[p] = den0[p] * (1.0f + dthl * rdth);
    }
    for (int p = 0; p < NG; ++p) {
        double cdldv = cdl * den0[p] * forog[p];
        P3(ustr, p, 1) = -cdldv * A3(ua, p, KX);
        P3(vstr, p, 1) = -cdldv * A3(va, p, KX);
    }
    const double chlcp = chl * CP;
    for (int p = 0; p < NG; ++p) P3(shf, p, 1) = chlcp * den1[p] * (tskin[p] - t1l[p]);
    for (int p = 0; p < NG; ++p) q1l[p] = A3(qa, p, KX); /* fhum0 = 0 branch */
    for (int p = 0; p < NG; ++p) one[p] = 1.0;
    (void)one;
    orc_qsat(tskin, psa, 1.0, qs0l, NG);
    for (int p = 0; p < NG; ++p) P3(evap, p, 1) = chl * den1[p] * dmax(0.0f, soil_avail_water[p] * qs0l[p] - q1l[p]);
    for (int p = 0; p < NG; ++p) {
        double tsk3 = pow3(tskin[p]);
        tmp[p] = 4.0f * esbc * tsk3; /* dslr */
        P3(slru, p, 1) = esbc * tsk3 * tskin[p];
        P3(hfluxn, p, 1) = ssrd[p] * (1.0f - alb_land[p]) + slrd[p] - (P3(slru, p, 1) + P3(shf, p, 1) + ALHC * P3(evap, p, 1));
    }
    /* skin-temperature energy balance, surface_fluxes.f90:216-244 */
    for (int p = 0; p < NG; ++p) {
        double clamb = clambda + snowc[p] * (clambsn - clambda);
        P3(hfluxn, p, 1) = P3(hfluxn, p, 1) - clamb * (tskin[p] - land_temp[p]);
        dtskin[p] = tskin[p] + 1.0f;
    }
    orc_qsat(dtskin, psa, 1.0, qs0s, NG);
    for (int p = 0; p < NG; ++p) {
        if (P3(evap, p, 1) > 0.0)
            qs0s[p] = soil_avail_water[p] * (qs0s[p] - qs0l[p]);
        else
            qs0s[p] = 0.0;
    }
    for (int p = 0; p < NG; ++p) {
        double clamb = clambda + snowc[p] * (clambsn - clambda);
        dtskin[p] = P3(hfluxn, p, 1) / (clamb + tmp[p] + chl * den1[p] * (CP + ALHC * qs0s[p]));
        tskin[p] = tskin[p] + dtskin[p];
        P3(shf, p, 1) = P3(shf, p, 1) + chlcp * den1[p] * dtskin[p];
        P3(evap, p, 1) = P3(evap, p, 1) + chl * den1[p] * qs0s[p] * dtskin[p];
        P3(slru, p, 1) = P3(slru, p, 1) + tmp[p] * dtskin[p];
        P3(hfluxn, p, 1) = clamb * (tskin[p] - land_temp[p]);
    }
    rdth = fstab / dtheta;
    for (int p = 0; p < NG; ++p) {
        double dths;
        if (tsea[p] > t2s[p])
            dths = dmin(dtheta, tsea[p] - t2s[p]);
        else
            dths = dmax(-dtheta, astab * (tsea[p] - t2s[p]));
        den2[p] = den0[p] * (1.0f + dths * rdth);
    }
    for (int p = 0; p < NG; ++p) q1s[p] = A3(qa, p, KX);
    for (int p = 0; p < NG; ++p) {
        double cdsdv = cds * den2[p];
        P3(ustr, p, 2) = -cdsdv * A3(ua, p, KX);
        P3(vstr, p, 2) = -cdsdv * A3(va, p, KX);
    }
    /* sea surface */
    for (int p = 0; p < NG; ++p) P3(shf, p, 2) = chs * CP * den2[p] * (tsea[p] - t1s[p]);
    orc_qsat(tsea, psa, 1.0, qs0s, NG);
    for (int p = 0; p < NG; ++p) P3(evap, p, 2) = chs * den2[p] * (qs0s[p] - q1s[p]);
    for (int p = 0; p < NG; ++p) {
        P3(slru, p, 2) = esbc * pow4(tsea[p]);
        P3(hfluxn, p, 2) = ssrd[p] * (1.0f - alb_sea[p]) + slrd[p] - P3(slru, p, 2) + P3(shf, p, 2) + ALHC * P3(evap, p, 2);
    }
    for (int p = 0; p < NG; ++p) {
        P3(ustr, p, 3) = P3(ustr, p, 2) + fmask[p] * (P3(ustr, p, 1) - P3(ustr, p, 2));
        P3(vstr, p, 3) = P3(vstr, p, 2) + fmask[p] * (P3(vstr, p, 1) - P3(vstr, p, 2));
        P3(shf, p, 3) = P3(shf, p, 2) + fmask[p] * (P3(shf, p, 1) - P3(shf, p, 2));
        P3(evap, p, 3) = P3(evap, p, 2) + fmask[p] * (P3(evap, p, 1) - P3(evap, p, 2));
        P3(slru, p, 3) = P3(slru, p, 2) + fmask[p] * (P3(slru, p, 1) - P3(slru, p, 2));
        tsfc[p] = tsea[p] + fmask[p] * (land_temp[p] - tsea[p]);
        tskin[p] = tsea[p] + fmask[p] * (tskin[p] - tsea[p]);
        t0[p] = t1s[p] + fmask[p] * (t1l[p] - t1s[p]);
    }
#undef P3
    free(buf);
}

/* ---------------------------------------------------------------- vertical_diffusion.f90:30-146 */
void orc_vdiff(const orc_tables *t, const double *se, const double *rh, const double *qa, const double *qsat,
               const double *phi, const int *icnv, double *ut, double *vt, double *tt, double *qt) {
    const double trshc = 6.0f, trvdi = 24.0f, trvds = 6.0f, redshc = 0.5f, rhgrad = 0.5f, segrad = 0.1f;
    const int nl1 = KX - 1;
    const double cshc = t->dhs[KX - 1] / 3600.0f;
    const double cvdi = (t->sigh[nl1] - t->sigh[1]) / ((nl1 - 1) * 3600.0f);
    const double fshcq = cshc / trshc, fshcse = cshc / (trshc * CP);
    const double fvdiq = cvdi / trvdi, fvdise = cvdi / (trvds * CP);
    double rsig[KX + 1], rsig1[KX + 1];
    for (int k = 1; k <= nl1; ++k) {
        rsig[k] = 1.0f / t->dhs[k - 1];
        rsig1[k] = 1.0f / (1.0f - t->sigh[k]);
    }
    rsig[KX] = 1.0f / t->dhs[KX - 1];
    memset(ut, 0, sizeof(double) * NG * KX);
    memset(vt, 0, sizeof(double) * NG * KX);
    memset(tt, 0, sizeof(double) * NG * KX);
    memset(qt, 0, sizeof(double) * NG * KX);
    double drh0 = rhgrad * (t->fsg[KX - 1] - t->fsg[nl1 - 1]);
    double fvdiq2 = fvdiq * t->sigh[nl1];
    for (int p = 0; p < NG; ++p) {
        double dmse = A3(se, p, KX) - A3(se, p, nl1) + ALHC * (A3(qa, p, KX) - A3(qsat, p, nl1));
        double drh = A3(rh, p, KX) - A3(rh, p, nl1);
        double fcnv = 1.0f;
        if (dmse >= 0.0) {
            if (icnv[p] > 0) fcnv = redshc;
            double fluxse = fcnv * fshcse * dmse;
            A3(tt, p, nl1) = fluxse * rsig[nl1];
            A3(tt, p, KX) = -fluxse * rsig[KX];
            if (drh >= 0.0) {
                double fluxq = fcnv * fshcq * A3(qsat, p, KX) * drh;
                A3(qt, p, nl1) = fluxq * rsig[nl1];
                A3(qt, p, KX) = -fluxq * rsig[KX];
            }
        } else if (drh > drh0) {
            double fluxq = fvdiq2 * A3(qsat, p, nl1) * drh;
            A3(qt, p, nl1) = fluxq * rsig[nl1];
            A3(qt, p, KX) = -fluxq * rsig[KX];
        }
    }
    for (int k = 3; k <= KX - 2; ++k) {
        if (t->sigh[k] > 0.5f) {
            drh0 = rhgrad * (t->fsg[k] - t->fsg[k - 1]);
            fvdiq2 = fvdiq * t->sigh[k];
            for (int p = 0; p < NG; ++p) {
                double drh = A3(rh, p, k + 1) - A3(rh, p, k);
                if (drh >= drh0) {
                    double fluxq = fvdiq2 * A3(qsat, p, k) * drh;
                    A3(qt, p, k) = A3(qt, p, k) + fluxq * rsig[k];
                    A3(qt, p, k + 1) = A3(qt, p, k + 1) - fluxq * rsig[k + 1];
                }
            }
        }
    }
    for (int k = 1; k <= nl1; ++k)
        for (int p = 0; p < NG; ++p) {
            double se0 = A3(se, p, k + 1) + segrad * (A3(phi, p, k) - A3(phi, p, k + 1));
            if (A3(se, p, k) < se0) {
                double fluxse = fvdise * (se0 - A3(se, p, k));
                A3(tt, p, k) = A3(tt, p, k) + fluxse * rsig[k];
                for (int k1 = k + 1; k1 <= KX; ++k1) A3(tt, p, k1) = A3(tt, p, k1) - fluxse * rsig1[k];
            }
        }
}

/* ---------------------------------------------------------------- physics.f90:14-256 (from line 107) */
void orc_physics(const orc_tables *t, orc_phys_io *io) {
    const size_t n3 = (size_t)NG * KX;
    double *w = (double *)malloc(sizeof(double) * (n3 * 13 + NG * 16));
    double *qg = w, *se = qg + n3, *rh = se + n3, *qsat = rh + n3, *tt_cnv = qsat + n3, *qt_cnv = tt_cnv + n3,
           *tt_lsc = qt_cnv + n3, *qt_lsc = tt_lsc + n3, *tt_rlw = qt_lsc + n3, *ut_pbl = tt_rlw + n3,
           *vt_pbl = ut_pbl + n3, *tt_pbl = vt_pbl + n3, *qt_pbl = tt_pbl + n3;
    double *psg = qt_pbl + n3, *rps = psg + NG, *gse = rps + NG, *ts = gse + NG, *tskin = ts + NG, *u0 = tskin + NG,
           *v0 = u0 + NG, *t0 = v0 + NG, *cloudc = t0 + NG, *clstr = cloudc + NG;
    int *iptop = (int *)malloc(sizeof(int) * NG * 3), *icnv = iptop + NG, *icltop = icnv + NG;
    const double *tg = io->tg, *phig = io->phig;

    for (int p = 0; p < NG; ++p) { /* physics.f90:107-108 */
        psg[p] = exp(io->pslg[p]);
        rps[p] = 1.0f / psg[p];
    }
    for (size_t q = 0; q < n3; ++q) { /* :110-111 */
        qg[q] = dmax(io->qg_in[q], 0.0f);
        se[q] = CP * tg[q] + phig[q];
    }
    for (int k = 1; k <= KX; ++k) { /* :113-116, humidity.f90:17-27 */
        orc_qsat(tg + NG * (k - 1), psg, t->fsg[k - 1], qsat + NG * (k - 1), NG);
        for (int p = 0; p < NG; ++p) A3(rh, p, k) = A3(qg, p, k) / A3(qsat, p, k);
    }
    orc_convection(t, psg, se, qg, qsat, iptop, io->cbmf, io->precnv, tt_cnv, qt_cnv); /* :123-125 */
    for (int k = 2; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) {
            A3(tt_cnv, p, k) = A3(tt_cnv, p, k) * rps[p] * t->grdscp[k - 1];
            A3(qt_cnv, p, k) = A3(qt_cnv, p, k) * rps[p] * t->grdsig[k - 1];
        }
    for (int p = 0; p < NG; ++p) icnv[p] = KX - iptop[p];
    orc_lsc(t, psg, qg, qsat, iptop, io->precls, tt_lsc, qt_lsc); /* :135-136 */
    for (size_t q = 0; q < n3; ++q) {
        io->ttend[q] = io->ttend[q] + tt_cnv[q] + tt_lsc[q];
        io->qtend[q] = io->qtend[q] + qt_cnv[q] + qt_lsc[q];
    }
    if (io->compute_shortwave) { /* :151-169 */
        for (int p = 0; p < NG; ++p)
            gse[p] = (A3(se, p, KX - 1) - A3(se, p, KX)) / (A3(phig, p, KX - 1) - A3(phig, p, KX));
        orc_clouds(qg, rh, io->precnv, io->precls, iptop, gse, io->fmask_land, icltop, cloudc, clstr, io->qcloud_equiv);
        orc_shortwave(t, io, psg, qg, icltop, cloudc, clstr);
        for (int k = 1; k <= KX; ++k)
            for (int p = 0; p < NG; ++p) A3(io->tt_rsw, p, k) = A3(io->tt_rsw, p, k) * rps[p] * t->grdscp[k - 1];
    } else {
        for (int p = 0; p < NG; ++p) {
            icltop[p] = 0;
            cloudc[p] = clstr[p] = 0.0;
        }
    }
    orc_lw_down(t, tg, io->slrd, tt_rlw, io->rad_flux, io->rad_tau2, io->rad_st4a); /* :172-174 */
    orc_surface_fluxes(t, psg, io->ug, io->vg, tg, qg, rh, phig, io->phis0, io->fmask_land, io->forog, io->sst_am,
                       io->ssrd, io->slrd, io->ustr, io->vstr, io->shf, io->evap, io->slru, io->hfluxn, ts, tskin, u0, v0,
                       t0, io->alb_land, io->alb_sea, io->snowc, io->land_temp, io->soil_avail_water); /* :177-185 */
    orc_lw_up(t, tg, ts, io->slrd, io->slru + 2 * NG, io->slr, io->olr, tt_rlw, io->rad_flux, io->rad_tau2, io->rad_st4a,
              io->rad_strat_corr); /* :202-206 */
    for (int k = 1; k <= KX; ++k)
        for (int p = 0; p < NG; ++p) A3(tt_rlw, p, k) = A3(tt_rlw, p, k) * rps[p] * t->grdscp[k - 1];
    for (size_t q = 0; q < n3; ++q) io->ttend[q] = io->ttend[q] + io->tt_rsw[q] + tt_rlw[q]; /* :211 */
    orc_vdiff(t, se, rh, qg, qsat, phig, icnv, ut_pbl, vt_pbl, tt_pbl, qt_pbl); /* :218-220 */
    for (int p = 0; p < NG; ++p) { /* :223-226 */
        A3(ut_pbl, p, KX) = A3(ut_pbl, p, KX) + io->ustr[p + 2 * NG] * rps[p] * t->grdsig[KX - 1];
        A3(vt_pbl, p, KX) = A3(vt_pbl, p, KX) + io->vstr[p + 2 * NG] * rps[p] * t->grdsig[KX - 1];
        A3(tt_pbl, p, KX) = A3(tt_pbl, p, KX) + io->shf[p + 2 * NG] * rps[p] * t->grdscp[KX - 1];
        A3(qt_pbl, p, KX) = A3(qt_pbl, p, KX) + io->evap[p + 2 * NG] * rps[p] * t->grdsig[KX - 1];
    }
    for (size_t q = 0; q < n3; ++q) { /* :228-231 */
        io->utend[q] = io->utend[q] + ut_pbl[q];
        io->vtend[q] = io->vtend[q] + vt_pbl[q];
        io->ttend[q] = io->ttend[q] + tt_pbl[q];
        io->qtend[q] = io->qtend[q] + qt_pbl[q];
    }
    if (io->iptop) memcpy(io->iptop, iptop, sizeof(int) * NG);
    if (io->icltop) memcpy(io->icltop, icltop, sizeof(int) * NG);
#define COPY(dst, src) if (io->dst) memcpy(io->dst, src, sizeof(double) * NG)
    COPY(ts, ts); COPY(tskin, tskin); COPY(u0, u0); COPY(v0, v0); COPY(t0, t0); COPY(cloudc, cloudc); COPY(clstr, clstr);
#undef COPY
    free(iptop);
    free(w);
}

/* ---------------------------------------------------------------- physics.f90:89-101 + the column schemes */
void orc_physics_from_spectral(const orc_tables *t, orc_state *s, int j1, double *utend, double *vtend, double *ttend,
                               double *qtend) {
    const size_t n3 = (size_t)NG * KX, ns2 = 2 * 31 * 32, lev = ns2 * KX;
    double *g = (double *)malloc(sizeof(double) * (n3 * 5 + NG));
    double *ug = g, *vg = ug + n3, *tg = vg + n3, *qg = tg + n3, *phig = qg + n3, *pslg = phig + n3;
    double ucos[2 * 31 * 32], vcos[2 * 31 * 32];
    for (int k = 0; k < KX; ++k) {
        const size_t o = ns2 * k + lev * (j1 - 1);
        orc_vort2vel(t, s->vor + o, s->div + o, ucos, vcos);
        orc_spec2grid(t, ucos, ug + (size_t)NG * k, 2);
        orc_spec2grid(t, vcos, vg + (size_t)NG * k, 2);
        orc_spec2grid(t, s->t + o, tg + (size_t)NG * k, 1);
        orc_spec2grid(t, s->tr + o, qg + (size_t)NG * k, 1);
        orc_spec2grid(t, s->phi + ns2 * k, phig + (size_t)NG * k, 1);
    }
    orc_spec2grid(t, s->ps + ns2 * (j1 - 1), pslg, 1);
    s->ph.ug = ug; s->ph.vg = vg; s->ph.tg = tg; s->ph.qg_in = qg; s->ph.phig = phig; s->ph.pslg = pslg;
    s->ph.utend = utend; s->ph.vtend = vtend; s->ph.ttend = ttend; s->ph.qtend = qtend;
    orc_physics(t, &s->ph);
    s->ph.ug = s->ph.vg = s->ph.tg = s->ph.qg_in = s->ph.phig = s->ph.pslg = NULL;
    free(g);
}
