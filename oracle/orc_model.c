/* TEST INFRASTRUCTURE -- CPU oracle (see speedy_oracle.h).  NOT PART OF THE PRODUCT.
 *
 * The rest of the model around the hot path, restated in plain C so that the oracle is a WHOLE model that can be run beside
 * the reference from boundary fields to any date:
 *   model calendar                         speedy.f90/model_control.f90:79-185
 *   time interpolation                     interpolation.f90:16-93
 *   daily forcing                          forcing.f90:15-117, shortwave_radiation.f90:218-322
 *   land / sea / sea-ice coupling          coupler.f90, land_model.f90:151-215, sea_model.f90:193-383
 *   initialisation                         initialization.f90:13-91, boundaries.f90:22-37, prognostics.f90:29-120,
 *                                          time_stepping.f90:13-27, surface_fluxes.f90:324-334
 *   one model step                         speedy.f90:20-74 (do_single_step)
 * on top of orc_step / orc_physics / the transforms (orc_dynamics.c, orc_physics.c, orc_spectral.c) and the boundary-field
 * preprocessing (orc_surface.c).  Pinned bit for bit against the flang-compiled reference: whole runs of 36 ... 360 steps,
 * a month crossing with SST anomalies and the CO2 trend, leap February 1980, the 1982/83 year end, both coupling flags off
 * (tests/test_model_oracle.py on tests/golden/run.npz, run10.npz, anomaly.npz, calendar.npz).
 *
 * Default-real literals of the reference are float literals widened to double; fp32 sub-expressions are evaluated in float.
 * sin and cos of one argument come from one sincos call, as in the compiled reference (both flang and gfortran merge the pair).
 */
#define _GNU_SOURCE
#include "speedy_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IX ORC_IX
#define IL ORC_IL
#define KX ORC_KX
#define MX ORC_MX
#define NX ORC_NX
#define NG (IX * IL)
#define NS (MX * NX)

static const double GRAV = 9.81f, CP = 1004.0f, GAMMA = 6.0f, HSCALE = 7.5f, HSHUM = 2.5f, REFRH1 = 0.7f;
#define AKAP ((double)(2.0f / 7.0f))
#define RGAS (AKAP * CP)
static const double ALBSEA = 0.07f, ALBICE = 0.60f, ALBSN = 0.60f, EMISFC = 0.98f, ALHC = 2501.0f, SBC = 5.67e-8f;
static const double DELT = 86400.0f / 36;

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* ------------------------------------------------------------------------------------------------ the model object */
struct orc_model {
    orc_tables T;
    orc_dyn_tables D;
    orc_state S;
    int planes; /* sst_anom(ix, il, 0:n_months+1) */
    /* calendar, model_control.f90:37-47 */
    int year, month, day, hour, minute, month_idx, imont1;
    double tmonth, tyear;
    /* run control */
    int current_step, initialized, land_coupling_flag, sst_anomaly_coupling_flag, increase_co2;
    double ablco2_ref, fmean;
    /* boundary fields and what initialisation derives from them */
    double *orog, *fmask_orig, *alb0, *veg_high, *veg_low, *stl12, *snowd12, *soil_wc_l1, *soil_wc_l2, *soil_wc_l3, *sst12,
        *sea_ice_frac12, *sst_anom, *soilw12, *phi0, *fmask_sea, *bmask_land, *bmask_sea, *rhcapl, *cdland, *rhcaps, *rhcapi,
        *cdsea, *cdice, *hfseacl;
    /* slab models */
    double *stlcl_obs, *snowdcl_obs, *soilwcl_obs, *stl_lm, *snow_depth, *sstcl_ob, *sicecl_ob, *ticecl_ob, *sstan_ob, *sst_om,
        *tice_om, *sice_om, *sstan_am, *sice_am, *tice_am, *ssti_om;
    /* name -> array, for the test harness */
    struct {
        const char *name;
        double *ptr;
        long n;
    } reg[96];
    int nreg;
};

static double *take(orc_model *m, const char *name, long n) {
    double *p = (double *)calloc((size_t)n, sizeof(double));
    m->reg[m->nreg].name = name;
    m->reg[m->nreg].ptr = p;
    m->reg[m->nreg].n = n;
    m->nreg += 1;
    return p;
}

orc_model *orc_model_new(int n_anom_planes) {
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    orc_tables_init(&m->T);
    orc_dyn_tables_init(&m->T, &m->D);
    m->planes = n_anom_planes;
    m->land_coupling_flag = m->sst_anomaly_coupling_flag = 1; /* model_state_def.py:378-418 */
    m->S.ph.air_absortivity_co2 = 6.0;                        /* model_state_def.py: air_absortivity_co2 */
    m->S.ph.compute_shortwave = 1;
    m->month_idx = 1;
    orc_state *s = &m->S;
    orc_phys_io *ph = &s->ph;
    const long lev = 2L * NS * KX;
#define TAKE(field, n) field = take(m, #field, (n))
#define TAKE_AS(target, name, n) target = take(m, name, (n))
    TAKE_AS(s->vor, "vor", 2 * lev); TAKE_AS(s->div, "div", 2 * lev); TAKE_AS(s->t, "t", 2 * lev); TAKE_AS(s->tr, "tr", 2 * lev);
    TAKE_AS(s->ps, "ps", 4L * NS); TAKE_AS(s->phi, "phi", lev); TAKE_AS(s->phis, "phis", 2L * NS);
    TAKE_AS(s->tcorh, "tcorh", 2L * NS); TAKE_AS(s->qcorh, "qcorh", 2L * NS);
    double **in2d[] = {(double **)&ph->fmask_land, (double **)&ph->phis0, (double **)&ph->forog, (double **)&ph->sst_am,
                       (double **)&ph->alb_land, (double **)&ph->alb_sea, (double **)&ph->snowc, (double **)&ph->land_temp,
                       (double **)&ph->soil_avail_water, (double **)&ph->flux_solar_in, (double **)&ph->flux_ozone_upper,
                       (double **)&ph->flux_ozone_lower, (double **)&ph->zenit_correction, (double **)&ph->stratospheric_correction,
                       (double **)&ph->alb_surface};
    static const char *in2d_names[] = {"fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc", "land_temp",
                                       "soil_avail_water", "flux_solar_in", "flux_ozone_upper", "flux_ozone_lower",
                                       "zenit_correction", "stratospheric_correction", "alb_surface"};
    for (int i = 0; i < 15; ++i) *in2d[i] = take(m, in2d_names[i], NG);
    TAKE_AS(ph->precnv, "precnv", NG); TAKE_AS(ph->precls, "precls", NG); TAKE_AS(ph->cbmf, "cbmf", NG);
    TAKE_AS(ph->slrd, "slrd", NG); TAKE_AS(ph->slr, "slr", NG); TAKE_AS(ph->olr, "olr", NG);
    TAKE_AS(ph->slru, "slru", 3L * NG); TAKE_AS(ph->ustr, "ustr", 3L * NG); TAKE_AS(ph->vstr, "vstr", 3L * NG);
    TAKE_AS(ph->shf, "shf", 3L * NG); TAKE_AS(ph->evap, "evap", 3L * NG); TAKE_AS(ph->hfluxn, "hfluxn", 3L * NG);
    TAKE_AS(ph->rad_st4a, "rad_st4a", 2L * NG * KX); TAKE_AS(ph->rad_flux, "rad_flux", 4L * NG);
    TAKE_AS(ph->tt_rsw, "tt_rsw", (long)NG * KX); TAKE_AS(ph->rad_tau2, "rad_tau2", 4L * NG * KX);
    TAKE_AS(ph->rad_strat_corr, "rad_strat_corr", 2L * NG);
    TAKE_AS(ph->tsr, "tsr", NG); TAKE_AS(ph->ssrd, "ssrd", NG); TAKE_AS(ph->ssr, "ssr", NG);
    TAKE_AS(ph->qcloud_equiv, "qcloud_equiv", NG);
    TAKE(m->orog, NG); TAKE(m->fmask_orig, NG); TAKE(m->alb0, NG); TAKE(m->veg_high, NG); TAKE(m->veg_low, NG);
    TAKE(m->stl12, 12L * NG); TAKE(m->snowd12, 12L * NG); TAKE(m->soil_wc_l1, 12L * NG); TAKE(m->soil_wc_l2, 12L * NG);
    TAKE(m->soil_wc_l3, 12L * NG); TAKE(m->sst12, 12L * NG); TAKE(m->sea_ice_frac12, 12L * NG);
    TAKE(m->sst_anom, (long)n_anom_planes * NG); TAKE(m->soilw12, 12L * NG);
    TAKE(m->phi0, NG); TAKE(m->fmask_sea, NG); TAKE(m->bmask_land, NG); TAKE(m->bmask_sea, NG); TAKE(m->rhcapl, NG);
    TAKE(m->cdland, NG); TAKE(m->rhcaps, NG); TAKE(m->rhcapi, NG); TAKE(m->cdsea, NG); TAKE(m->cdice, NG); TAKE(m->hfseacl, NG);
    TAKE(m->stlcl_obs, NG); TAKE(m->snowdcl_obs, NG); TAKE(m->soilwcl_obs, NG); TAKE(m->stl_lm, NG); TAKE(m->snow_depth, NG);
    TAKE(m->sstcl_ob, NG); TAKE(m->sicecl_ob, NG); TAKE(m->ticecl_ob, NG); TAKE(m->sstan_ob, NG); TAKE(m->sst_om, NG);
    TAKE(m->tice_om, NG); TAKE(m->sice_om, NG); TAKE(m->sstan_am, NG); TAKE(m->sice_am, NG); TAKE(m->tice_am, NG);
    TAKE(m->ssti_om, NG);
#undef TAKE
#undef TAKE_AS
    /* registry names of the reference where the struct member is spelled differently */
    for (int i = 0; i < m->nreg; ++i)
        if (strncmp(m->reg[i].name, "m->", 3) == 0) m->reg[i].name += 3;
    return m;
}

void orc_model_free(orc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->nreg; ++i) free(m->reg[i].ptr);
    free(m);
}

double *orc_model_field(orc_model *m, const char *name, long *n_doubles) {
    for (int i = 0; i < m->nreg; ++i)
        if (strcmp(m->reg[i].name, name) == 0) {
            if (n_doubles) *n_doubles = m->reg[i].n;
            return m->reg[i].ptr;
        }
    return NULL;
}

/* flags / scalars by name; -> 0, or -1 for an unknown name */
int orc_model_set_scalar(orc_model *m, const char *name, double v) {
    if (!strcmp(name, "land_coupling_flag")) m->land_coupling_flag = v != 0;
    else if (!strcmp(name, "sst_anomaly_coupling_flag")) m->sst_anomaly_coupling_flag = v != 0;
    else if (!strcmp(name, "increase_co2")) m->increase_co2 = v != 0;
    else if (!strcmp(name, "air_absortivity_co2")) m->S.ph.air_absortivity_co2 = v;
    else return -1;
    return 0;
}

double orc_model_get_scalar(const orc_model *m, const char *name) {
    if (!strcmp(name, "current_step")) return m->current_step;
    if (!strcmp(name, "air_absortivity_co2")) return m->S.ph.air_absortivity_co2;
    if (!strcmp(name, "ablco2_ref")) return m->ablco2_ref;
    if (!strcmp(name, "compute_shortwave")) return m->S.ph.compute_shortwave;
    return NAN;
}

/* ------------------------------------------------------------------------------------------------ calendar */
static const int NDAYCAL[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31}; /* ncal365, model_control.f90:60 */

static void update_forcing_params(orc_model *m) { /* :162-185; every right-hand side is a default-real expression */
    int before = 0;
    for (int k = 1; k < m->month; ++k) before += NDAYCAL[k - 1];
    m->imont1 = m->month;
    m->tmonth = (double)(((float)m->day - 0.5f) / (float)NDAYCAL[m->month - 1]);
    m->tyear = (double)(((float)(before + m->day) - 0.5f) / (float)365);
}

static void initialize_control(orc_model *m, int y, int mo, int d, int h, int mi) { /* :79-111 */
    m->year = y; m->month = mo; m->day = d; m->hour = h; m->minute = mi;
    m->month_idx = 1;
    update_forcing_params(m);
}

static void advance_date(orc_model *m) { /* :114-160 */
    m->minute += 24 * 60 / 36;
    if (m->minute >= 60) {
        m->minute %= 60;
        m->hour += 1;
    }
    if (m->hour >= 24) {
        m->hour %= 24;
        m->day += 1;
    }
    if (m->year % 4 == 0 && m->month == 2) {
        if (m->day > 29) {
            m->day = 1;
            m->month += 1;
            m->month_idx += 1;
        }
    } else if (m->day > NDAYCAL[m->month - 1]) {
        m->day = 1;
        m->month += 1;
        m->month_idx += 1;
    }
    if (m->month > 12) {
        m->month = 1;
        m->year += 1;
    }
    update_forcing_params(m);
}

void orc_model_calendar(const orc_model *m, int *ymdhm, int *month_idx, int *imont1, double *tmonth, double *tyear) {
    ymdhm[0] = m->year; ymdhm[1] = m->month; ymdhm[2] = m->day; ymdhm[3] = m->hour; ymdhm[4] = m->minute;
    *month_idx = m->month_idx;
    *imont1 = m->imont1;
    *tmonth = m->tmonth;
    *tyear = m->tyear;
}

/* ------------------------------------------------------------------------------------------------ interpolation.f90 */
static void forint(int imon, const double *for12, double *for1, double tmonth) { /* :40-58 */
    int imon2;
    double wmon;
    if (tmonth <= 0.5f) {
        imon2 = imon == 1 ? 12 : imon - 1;
        wmon = 0.5f - tmonth;
    } else {
        imon2 = imon == 12 ? 1 : imon + 1;
        wmon = tmonth - 0.5f;
    }
    const double *a = for12 + (size_t)(imon - 1) * NG, *b = for12 + (size_t)(imon2 - 1) * NG;
    for (int p = 0; p < NG; ++p) for1[p] = a[p] + wmon * (b[p] - a[p]);
}

static void forin5(int imon, const double *for12, double *for1, double tmonth) { /* :61-93 */
    int im2 = imon - 2, im1 = imon - 1, ip1 = imon + 1, ip2 = imon + 2;
    if (im2 < 1) im2 += 12;
    if (im1 < 1) im1 += 12;
    if (ip1 > 12) ip1 -= 12;
    if (ip2 > 12) ip2 -= 12;
    const double c0 = (double)(1.0f / 12.0f);
    const double t0 = c0 * tmonth, t1 = c0 * (1.0f - tmonth), t2 = 0.25f * tmonth * (1 - tmonth);
    const double wm2 = -t1 + t2, wm1 = -c0 + 8 * t1 - 6 * t2, w0 = 7 * c0 + 10 * t2, wp1 = -c0 + 8 * t0 - 6 * t2, wp2 = -t0 + t2;
    const double *f2 = for12 + (size_t)(im2 - 1) * NG, *f1 = for12 + (size_t)(im1 - 1) * NG, *f0 = for12 + (size_t)(imon - 1) * NG,
                 *g1 = for12 + (size_t)(ip1 - 1) * NG, *g2 = for12 + (size_t)(ip2 - 1) * NG;
    for (int p = 0; p < NG; ++p) for1[p] = wm2 * f2[p] + wm1 * f1[p] + w0 * f0[p] + wp1 * g1[p] + wp2 * g2[p];
}

static void monthly_interp(int month_idx, const double *in, double *out, double frac) { /* :16-37; planes 0 .. n_months+1 */
    int imon2;
    double wmon;
    if (frac <= 0.5f) {
        imon2 = month_idx - 1;
        wmon = 0.5f - frac;
    } else {
        imon2 = month_idx + 1;
        wmon = frac - 0.5f;
    }
    const double *a = in + (size_t)month_idx * NG, *b = in + (size_t)imon2 * NG;
    for (int p = 0; p < NG; ++p) out[p] = a[p] + wmon * (b[p] - a[p]);
}

/* ------------------------------------------------------------------------------------------------ land_model.f90:151-215 */
static void couple_land_atm(orc_model *m, int day) {
    orc_phys_io *ph = &m->S.ph;
    double *land_temp = (double *)ph->land_temp, *soil_avail_water = (double *)ph->soil_avail_water;
    forin5(m->imont1, m->stl12, m->stlcl_obs, m->tmonth);
    forint(m->imont1, m->snowd12, m->snowdcl_obs, m->tmonth);
    forint(m->imont1, m->soilw12, m->soilwcl_obs, m->tmonth);
    if (day == 0) {
        memcpy(m->stl_lm, m->stlcl_obs, sizeof(double) * NG);
        memcpy(land_temp, m->stlcl_obs, sizeof(double) * NG);
    } else if (m->land_coupling_flag) {
        for (int p = 0; p < NG; ++p) { /* run_land_model, :194-215 */
            double tanom = m->stl_lm[p] - m->stlcl_obs[p];
            tanom = m->cdland[p] * (tanom + m->rhcapl[p] * ph->hfluxn[p]);
            m->stl_lm[p] = tanom + m->stlcl_obs[p];
        }
        memcpy(land_temp, m->stl_lm, sizeof(double) * NG);
    } else {
        memcpy(land_temp, m->stlcl_obs, sizeof(double) * NG);
    }
    memcpy(m->snow_depth, m->snowdcl_obs, sizeof(double) * NG);
    memcpy(soil_avail_water, m->soilwcl_obs, sizeof(double) * NG);
}

/* ------------------------------------------------------------------------------------------------ sea_model.f90:193-383
 * with the reference's compile-time settings sea_coupling_flag = 0, ice_coupling_flag = 1 (:20-23) */
static void run_sea_model(orc_model *m) { /* :313-383 */
    const orc_phys_io *ph = &m->S.ph;
    const double sstfr = 273.2f - 1.8f, beta = 1.0f;
    for (int p = 0; p < NG; ++p) {
        const double hfl2 = ph->hfluxn[NG + p];
        /* x**4.0 with a real exponent: libm pow in the flang build (orc_physics.c has the same finding for the longwave scheme) */
        const double difice = (ALBSEA - ALBICE) * ph->ssrd[p] + EMISFC * SBC * (pow(sstfr, 4.0) - pow(m->tice_am[p], 4.0)) +
                              ph->shf[NG + p] + ph->evap[NG + p] * ALHC;
        const double hflux_i = hfl2 + difice * (1.0f - m->sice_am[p]);
        double hflux = hfl2 - m->hfseacl[p] - m->sicecl_ob[p] * (hflux_i + beta * (sstfr - m->tice_om[p]));
        double tanom = m->sst_om[p] - m->sstcl_ob[p];
        tanom = m->cdsea[p] * (tanom + m->rhcaps[p] * hflux);
        m->sst_om[p] = tanom + m->sstcl_ob[p];
        hflux = hflux_i + beta * (sstfr - m->tice_om[p]);
        tanom = m->tice_om[p] - m->ticecl_ob[p];
        const double anom0 = 20.f;
        const double cdis = m->cdice[p] * (anom0 / (anom0 + fabs(tanom)));
        tanom = cdis * (tanom + m->rhcapi[p] * hflux);
        m->tice_om[p] = tanom + m->ticecl_ob[p];
        m->sice_om[p] = m->sicecl_ob[p];
    }
}

static void couple_sea_atm(orc_model *m, int day) {
    orc_phys_io *ph = &m->S.ph;
    double *sst_am = (double *)ph->sst_am;
    forin5(m->imont1, m->sst12, m->sstcl_ob, m->tmonth);
    forint(m->imont1, m->sea_ice_frac12, m->sicecl_ob, m->tmonth);
    if (m->sst_anomaly_coupling_flag) monthly_interp(m->month_idx, m->sst_anom, m->sstan_ob, m->tmonth);
    const double sstfr = 273.2f - 1.8f; /* single-precision subtraction, :229 */
    for (int p = 0; p < NG; ++p) {
        if (m->sstcl_ob[p] > sstfr) {
            m->sicecl_ob[p] = dmin(0.5f, m->sicecl_ob[p]);
            m->ticecl_ob[p] = sstfr;
            if (m->sicecl_ob[p] > 0.0) m->sstcl_ob[p] = sstfr + (m->sstcl_ob[p] - sstfr) / (1.0f - m->sicecl_ob[p]);
        } else {
            m->sicecl_ob[p] = dmax(0.5f, m->sicecl_ob[p]);
            m->ticecl_ob[p] = sstfr + (m->sstcl_ob[p] - sstfr) / m->sicecl_ob[p];
            m->sstcl_ob[p] = sstfr;
        }
    }
    if (day == 0) {
        memcpy(m->sst_om, m->sstcl_ob, sizeof(double) * NG);
        memcpy(m->tice_om, m->ticecl_ob, sizeof(double) * NG);
        memcpy(m->sice_om, m->sicecl_ob, sizeof(double) * NG);
        for (int p = 0; p < NG; ++p) m->sst_om[p] = 0.0; /* sea_coupling_flag <= 0, :261 */
    } else {
        run_sea_model(m); /* ice_coupling_flag > 0 */
    }
    for (int p = 0; p < NG; ++p) {
        m->sstan_am[p] = m->sst_anomaly_coupling_flag ? m->sstan_ob[p] : 0.0;
        sst_am[p] = m->sstcl_ob[p] + m->sstan_am[p];
        m->sice_am[p] = m->sice_om[p];
        m->tice_am[p] = m->tice_om[p];
        sst_am[p] = sst_am[p] + m->sice_am[p] * (m->tice_am[p] - sst_am[p]);
        m->ssti_om[p] = m->sst_om[p] + m->sice_am[p] * (m->tice_am[p] - m->sst_om[p]);
    }
}

/* ------------------------------------------------------------------------------------------------ daily forcing */
static void zonal_average_fields(orc_model *m, double tyear) { /* shortwave_radiation.f90:218-322 */
    const orc_tables *t = &m->T;
    orc_phys_io *ph = &m->S.ph;
    const float pih = asinf(1.0f);
    const double epssw = 0.020f, solc = 342.0f;
    const double alpha = (double)(4.0f * pih) * (tyear + (double)(10.0f / 365.0f));
    const double coz1 = 1.0f * dmax(0.0, cos(alpha - 0.0)), coz2 = 1.8f, azen = 1.0f, fs0 = 6.0f;
    const double rzen = -cos(alpha) * 23.45f * pih / 90.0f;
    double topsr[IL];
    { /* solar(), :277-322 */
        const double csol = 4.0f * solc, pigr = (double)(2.0f * pih), al = 2.0f * pigr * tyear;
        double ca1, sa1, cdecl, sdecl;
        sincos(al, &sa1, &ca1);
        const double ca2 = ca1 * ca1 - sa1 * sa1, sa2 = 2.f * sa1 * ca1, ca3 = ca1 * ca2 - sa1 * sa2, sa3 = sa1 * ca2 + sa2 * ca1;
        const double decl = 0.006918f - 0.399912f * ca1 + 0.070257f * sa1 - 0.006758f * ca2 + 0.000907f * sa2 - 0.002697f * ca3 +
                            0.001480f * sa3;
        const double fdis = 1.000110f + 0.034221f * ca1 + 0.001280f * sa1 + 0.000719f * ca2 + 0.000077f * sa2;
        sincos(decl, &sdecl, &cdecl);
        const double tdecl = sdecl / cdecl, csolp = csol / pigr;
        for (int j = 0; j < IL; ++j) {
            const double ch0 = dmin(1.0, dmax(-1.0, -tdecl * t->sia[j] / t->coa[j]));
            const double h0 = acos(ch0), sh0 = sin(h0);
            topsr[j] = csolp * fdis * (h0 * t->sia[j] * sdecl + sh0 * t->coa[j] * cdecl);
        }
    }
    double cz, sz;
    sincos(rzen, &sz, &cz);
    for (int j = 0; j < IL; ++j) {
        const double flat2 = 1.5f * (t->sia[j] * t->sia[j]) - 0.5f;
        const double q = 1.0f - (t->coa[j] * cz + t->sia[j] * sz);
        const double zen = 1.0f + azen * (q * q); /* (...)**nzen, nzen = 2 */
        const double o3u = 0.5f * epssw, o3l = 0.4f * epssw * (1.0f + coz1 * t->sia[j] + coz2 * flat2);
        for (int i = 0; i < IX; ++i) {
            const int p = i + IX * j;
            ((double *)ph->flux_solar_in)[p] = topsr[j];
            ((double *)ph->zenit_correction)[p] = zen;
            ((double *)ph->flux_ozone_upper)[p] = topsr[j] * o3u * zen;
            ((double *)ph->flux_ozone_lower)[p] = topsr[j] * o3l * zen;
            ((double *)ph->stratospheric_correction)[p] = dmax(fs0 - topsr[j], 0.0);
        }
    }
}

static void set_forcing(orc_model *m, int imode) { /* forcing.f90:15-102 */
    orc_phys_io *ph = &m->S.ph;
    if (imode == 0) {
        const double hdrag = 2000.0f, rhdrag = 1.0f / (GRAV * hdrag); /* set_orog_land_sfc_drag, surface_fluxes.f90:324-334 */
        for (int p = 0; p < NG; ++p) ((double *)ph->forog)[p] = 1.0f + rhdrag * (1.0f - exp(-dmax(ph->phis0[p], 0.0) * rhdrag));
        m->ablco2_ref = ph->air_absortivity_co2;
    }
    zonal_average_fields(m, m->tyear);
    for (int p = 0; p < NG; ++p) {
        const double snowc = dmin(1.0, m->snow_depth[p] / 60.0f);
        const double alb_land = m->alb0[p] + snowc * (ALBSN - m->alb0[p]);
        const double alb_sea = ALBSEA + m->sice_am[p] * (ALBICE - ALBSEA);
        ((double *)ph->snowc)[p] = snowc;
        ((double *)ph->alb_land)[p] = alb_land;
        ((double *)ph->alb_sea)[p] = alb_sea;
        ((double *)ph->alb_surface)[p] = alb_sea + ph->fmask_land[p] * (alb_land - alb_sea);
    }
    if (m->increase_co2) {
        const double del_co2 = 0.005f;
        ph->air_absortivity_co2 = m->ablco2_ref * exp(del_co2 * (m->year + m->tyear - 1950));
    }
    const double gamlat = GAMMA / (1000.f * GRAV), pexp = 1.f / (RGAS * gamlat);
    double *corh = (double *)malloc(sizeof(double) * NG * 6), *tsfc = corh + NG, *tref = tsfc + NG, *psfc = tref + NG,
           *qref = psfc + NG, *qsfc = qref + NG;
    for (int p = 0; p < NG; ++p) corh[p] = gamlat * ph->phis0[p];
    orc_grid2spec(&m->T, corh, m->S.tcorh);
    for (int p = 0; p < NG; ++p) {
        tsfc[p] = ph->fmask_land[p] * ph->land_temp[p] + m->fmask_sea[p] * ph->sst_am[p];
        tref[p] = tsfc[p] + corh[p];
        psfc[p] = pow(tsfc[p] / tref[p], pexp);
    }
    const double one = psfc[0] / psfc[0]; /* get_qsat(tref, psfc / psfc, -1): sig <= 0 takes ps(1, 1) as the pressure */
    orc_qsat(tref, &one, -1.0, qref, NG);
    orc_qsat(tsfc, psfc, 1.0, qsfc, NG);
    for (int p = 0; p < NG; ++p) corh[p] = REFRH1 * (qref[p] - qsfc[p]);
    orc_grid2spec(&m->T, corh, m->S.qcorh);
    free(corh);
}

/* ------------------------------------------------------------------------------------------------ initialisation */
static int initialize_from_rest_state(orc_model *m) { /* prognostics.f90:29-120 */
    const orc_tables *t = &m->T;
    orc_state *s = &m->S;
    const size_t lev = (size_t)2 * NS * KX;
    double surfs[2 * NS], *surfg = (double *)malloc(sizeof(double) * NG);
    const double gam1 = GAMMA / (1000.0f * GRAV);
    orc_grid2spec(t, s->ph.phis0, s->phis);
    memset(s->vor, 0, sizeof(double) * lev); /* time level 1 only */
    memset(s->div, 0, sizeof(double) * lev);
    memset(s->tr, 0, sizeof(double) * lev);
    const double tref = 288.0f, ttop = 216.0f, gam2 = gam1 / tref, rgam = RGAS * gam1, rgamr = 1.0f / rgam;
    const double sqrt2 = (double)sqrtf(2.0f);
    memset(s->t, 0, sizeof(double) * 2 * 2 * NS); /* levels 1 and 2 */
    for (int q = 0; q < 2 * NS; ++q) surfs[q] = -gam1 * s->phis[q];
    s->t[0] = sqrt2 * ttop;
    s->t[1] = 0.0 * ttop;
    s->t[2 * NS] = sqrt2 * ttop;
    s->t[2 * NS + 1] = 0.0 * ttop;
    surfs[0] = sqrt2 * tref - gam1 * s->phis[0];
    surfs[1] = 0.0 * tref - gam1 * s->phis[1];
    for (int k = 2; k < KX; ++k) {
        const double f = pow(t->fsg[k], rgam);
        for (int q = 0; q < 2 * NS; ++q) s->t[(size_t)2 * NS * k + q] = surfs[q] * f;
    }
    /* log(1.013) in default real, as the flang build evaluates it: 0x3c539ee9, one fp32 ulp above the correctly rounded value
     * that glibc's logf (and gcc's constant folder) return -- 5e-8 of ln ps; pinned by tests/golden/run.npz (time level 1
     * after initialisation is the untouched rest state) */
    const double rlog0 = 0.012916237115859985;
    for (int p = 0; p < NG; ++p) surfg[p] = rlog0 + rgamr * log(1.0f - gam2 * s->ph.phis0[p]);
    orc_grid2spec(t, surfg, s->ps);
    orc_truncate(t, s->ps);
    const double esref = 17.0f, qref = REFRH1 * 0.622f * esref, qexp = HSCALE / HSHUM;
    for (int p = 0; p < NG; ++p) surfg[p] = qref * exp(qexp * surfg[p]);
    orc_grid2spec(t, surfg, surfs);
    orc_truncate(t, surfs);
    for (int k = 2; k < KX; ++k) {
        const double f = pow(t->fsg[k], qexp);
        for (int q = 0; q < 2 * NS; ++q) s->tr[(size_t)2 * NS * k + q] = surfs[q] * f;
    }
    free(surfg);
    return orc_check_diagnostics(t, s, 1, NULL);
}

/* initialize_state (initialization.f90:13-91) for boundary fields already in the model.  -> the reference's error code */
int orc_model_init(orc_model *m, int year, int month, int day, int hour, int minute) {
    orc_phys_io *ph = &m->S.ph;
    initialize_control(m, year, month, day, hour, minute);
    m->current_step = 0;
    orc_dyn_tables_init(&m->T, &m->D);
    for (int p = 0; p < NG; ++p) m->phi0[p] = GRAV * m->orog[p]; /* initialize_boundaries, boundaries.f90:22-37 */
    orc_grid_filter(&m->T, m->phi0, (double *)ph->phis0);
    if (initialize_from_rest_state(m) != 0) return -2;
    /* initialize_coupler, coupler.f90:13-31 (land_model_init and sea_model_init: orc_surface.c) */
    orc_land_sea_init(&m->T, m->planes, m->fmask_orig, m->alb0, m->veg_high, m->veg_low, m->soil_wc_l1, m->soil_wc_l2, m->stl12,
                      m->snowd12, m->sst12, m->sea_ice_frac12, m->sst_anom, m->soilw12, (double *)ph->fmask_land, m->bmask_land,
                      m->fmask_sea, m->bmask_sea, m->rhcapl, m->cdland, m->rhcaps, m->rhcapi, m->cdsea, m->cdice, &m->fmean);
    memset(m->hfseacl, 0, sizeof(double) * NG);
    couple_land_atm(m, 0);
    couple_sea_atm(m, 0);
    set_forcing(m, 0);
    /* first_step, time_stepping.f90:13-27 */
    orc_dyn_set_time_step(&m->T, &m->D, 0.5 * DELT);
    orc_step(&m->T, &m->D, &m->S, 1, 1, 0.5 * DELT);
    orc_dyn_set_time_step(&m->T, &m->D, DELT);
    orc_step(&m->T, &m->D, &m->S, 1, 2, DELT);
    orc_dyn_set_time_step(&m->T, &m->D, 2 * DELT);
    m->initialized = 1;
    return 0;
}

/* do_single_step, speedy.f90:20-74.  -> 0, -1 (not initialised), -2 (diagnostics out of range) */
int orc_model_step(orc_model *m) {
    if (!m->initialized) return -1;
    if (m->current_step % 36 == 0) set_forcing(m, 1);
    m->S.ph.compute_shortwave = m->current_step % 3 == 0;
    orc_step(&m->T, &m->D, &m->S, 2, 2, 2 * DELT);
    m->current_step += 1;
    if (orc_check_diagnostics(&m->T, &m->S, 2, NULL) != 0) return -2;
    advance_date(m);
    const int day = 1 + m->current_step / 36;
    couple_land_atm(m, day);
    couple_sea_atm(m, day);
    return 0;
}
