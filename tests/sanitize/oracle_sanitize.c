/* AddressSanitizer / UndefinedBehaviorSanitizer driver for the CPU oracle (oracle/orc_*.c, test infrastructure): tables,
 * transforms and spectral operators on a synthetic field, the whole column physics on a synthetic member (shortwave and
 * ordinary step).  Compiled by tests/test_sanitizers.py with gcc -fsanitize=address,undefined -fno-sanitize-recover=all. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "speedy_oracle.h"

#define NG (96 * 48)
static double *arr(size_t n, double lo, double hi, unsigned seed) {
    double *a = malloc(n * sizeof(double));
    for (size_t i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        a[i] = lo + (hi - lo) * ((seed >> 8) / 16777216.0);
    }
    return a;
}

int main(void) {
    orc_tables *t = malloc(sizeof(orc_tables));
    orc_tables_init(t);
    double acc = 0.0;
    /* transforms and operators */
    double *spec = calloc(2 * 31 * 32, sizeof(double)), *grid = malloc(NG * sizeof(double)), *back = malloc(2 * 31 * 32 * sizeof(double));
    for (int n = 0; n < 32; ++n)
        for (int m = 0; m < 31; ++m)
            if (m + n <= 30) {
                spec[2 * (m + 31 * n)] = 1.0 / (1 + m + n);
                spec[2 * (m + 31 * n) + 1] = m ? 0.5 / (1 + m + n) : 0.0;
            }
    orc_spec2grid(t, spec, grid, 2);
    orc_grid2spec(t, grid, back);
    double *u = malloc(2 * 31 * 32 * sizeof(double)), *v = malloc(2 * 31 * 32 * sizeof(double));
    orc_vort2vel(t, spec, back, u, v);
    orc_vel2vort(t, u, v, spec, back);
    orc_gradient(t, spec, u, v);
    orc_laplacian(t, spec, u, 0);
    orc_laplacian(t, u, v, 1);
    orc_truncate(t, v);
    orc_grid_filter(t, grid, grid);
    for (int i = 0; i < NG; ++i) acc += grid[i];
    /* column physics on a synthetic member */
    orc_phys_io io;
    memset(&io, 0, sizeof(io));
    double *tg = malloc(8 * NG * sizeof(double)), *qg = malloc(8 * NG * sizeof(double)), *phig = malloc(8 * NG * sizeof(double));
    const double fsg[8] = {0.025, 0.095, 0.2, 0.34, 0.51, 0.685, 0.835, 0.95};
    unsigned seed = 7;
    for (int k = 0; k < 8; ++k)
        for (int p = 0; p < NG; ++p) {
            seed = seed * 1664525u + 1013904223u;
            const double r = (seed >> 8) / 16777216.0, lat = -1.5 + 3.0 * (p / 96) / 47.0;
            const double tsfc = 288.0 - 40.0 * sin(lat) * sin(lat);
            tg[p + NG * k] = fmax(205.0, tsfc * pow(fsg[k], 0.19) + 2.0 * r - 1.0);
            qg[p + NG * k] = 12.0 * exp(-(1 - fsg[k]) * 6.0) * cos(lat) * cos(lat) * (0.3 + 0.8 * r);
            phig[p + NG * k] = 287.0 * 260.0 * log(1.0 / fsg[k]) + 50.0 * (r - 0.5);
        }
    io.tg = tg; io.qg_in = qg; io.phig = phig;
    io.ug = arr(8 * NG, -20, 20, 1); io.vg = arr(8 * NG, -10, 10, 2); io.pslg = arr(NG, -0.16, 0.03, 3);
    io.utend = arr(8 * NG, -1e-5, 1e-5, 4); io.vtend = arr(8 * NG, -1e-5, 1e-5, 5); io.ttend = arr(8 * NG, -1e-5, 1e-5, 6);
    io.qtend = arr(8 * NG, -1e-6, 1e-6, 7);
    io.fmask_land = arr(NG, 0, 1, 8); io.phis0 = arr(NG, 0, 3000, 9); io.forog = arr(NG, 1, 1.3, 10); io.sst_am = arr(NG, 271, 303, 11);
    io.alb_land = arr(NG, 0.1, 0.5, 12); io.alb_sea = arr(NG, 0.07, 0.3, 13); io.snowc = arr(NG, 0, 1, 14);
    io.land_temp = arr(NG, 240, 310, 15); io.soil_avail_water = arr(NG, 0, 1, 16);
    io.flux_solar_in = arr(NG, 0, 450, 17); io.flux_ozone_upper = arr(NG, 0, 8, 18); io.flux_ozone_lower = arr(NG, 0, 8, 19);
    io.zenit_correction = arr(NG, 1, 1.8, 20); io.stratospheric_correction = arr(NG, 0, 6, 21); io.alb_surface = arr(NG, 0.07, 0.5, 22);
    io.air_absortivity_co2 = 6.0;
    io.precnv = calloc(NG, 8); io.precls = calloc(NG, 8); io.cbmf = calloc(NG, 8); io.slrd = calloc(NG, 8); io.slr = calloc(NG, 8);
    io.olr = calloc(NG, 8); io.slru = calloc(3 * NG, 8); io.ustr = calloc(3 * NG, 8); io.vstr = calloc(3 * NG, 8);
    io.shf = calloc(3 * NG, 8); io.evap = calloc(3 * NG, 8); io.hfluxn = calloc(3 * NG, 8); io.rad_st4a = calloc(16 * NG, 8);
    io.rad_flux = calloc(4 * NG, 8); io.tt_rsw = calloc(8 * NG, 8); io.rad_tau2 = calloc(32 * NG, 8);
    io.rad_strat_corr = calloc(2 * NG, 8); io.tsr = calloc(NG, 8); io.ssrd = calloc(NG, 8); io.ssr = calloc(NG, 8);
    io.qcloud_equiv = calloc(NG, 8);
    io.iptop = calloc(NG, sizeof(int)); io.icltop = calloc(NG, sizeof(int));
    io.ts = calloc(NG, 8); io.tskin = calloc(NG, 8); io.u0 = calloc(NG, 8); io.v0 = calloc(NG, 8); io.t0 = calloc(NG, 8);
    io.cloudc = calloc(NG, 8); io.clstr = calloc(NG, 8);
    for (int sw = 1; sw >= 0; --sw) {
        io.compute_shortwave = sw;
        orc_physics(t, &io);
        for (int p = 0; p < NG; ++p) acc += io.olr[p] + io.precnv[p] + io.ttend[p + 7 * NG];
    }
    /* dynamics tables */
    orc_dyn_tables *d = malloc(sizeof(orc_dyn_tables));
    orc_dyn_tables_init(t, d);
    orc_dyn_set_time_step(t, d, 2400.0);
    acc += d->xj[5] + d->elz[991];
    /* boundary-field preprocessing of init: holes, rows without a valid point, five anomaly planes */
    {
        double *in2[4], *in12[6], *out2[10], *soilw = calloc(12 * NG, 8), *anom = arr(5 * NG, -1.0, 1.0, 77u), fmean = 0.0;
        for (int k = 0; k < 4; ++k) in2[k] = arr(NG, -0.2, 1.0, 31u + k);
        for (int k = 0; k < 6; ++k) in12[k] = arr(12 * NG, k < 2 ? -0.2 : -80.0, k < 2 ? 0.8 : 310.0, 41u + k);
        for (int k = 0; k < 10; ++k) out2[k] = calloc(NG, 8);
        for (int i = 0; i < 96; ++i) in12[2][i + 96 * 23] = in12[4][i + 96 * 47 + 5 * NG] = -1.0;
        orc_land_sea_init(t, 5, in2[0], in2[1], in2[2], in2[3], in12[0], in12[1], in12[2], in12[3], in12[4], in12[5], anom, soilw,
                          out2[0], out2[1], out2[2], out2[3], out2[4], out2[5], out2[6], out2[7], out2[8], out2[9], &fmean);
        for (int p = 0; p < NG; ++p) acc += in12[2][p] + in12[4][p + 11 * NG] + soilw[p] + out2[0][p] + out2[9][p] + anom[p + 4 * NG];
        acc += fmean;
    }
    /* the whole model (orc_model.c): a synthetic planet -- a mountain, a continent, a seasonal cycle, SST anomalies over 5 planes --
     * initialised in late December and stepped across the year end with the CO2 trend on (daily forcing, coupling, calendar) */
    {
        orc_model *m = orc_model_new(5);
        long n = 0;
        double *orog = orc_model_field(m, "orog", &n), *lsm = orc_model_field(m, "fmask_orig", NULL), *alb = orc_model_field(m, "alb0", NULL);
        double *vegh = orc_model_field(m, "veg_high", NULL), *vegl = orc_model_field(m, "veg_low", NULL);
        double *stl = orc_model_field(m, "stl12", NULL), *snow = orc_model_field(m, "snowd12", NULL), *sst = orc_model_field(m, "sst12", NULL);
        double *ice = orc_model_field(m, "sea_ice_frac12", NULL), *sw1 = orc_model_field(m, "soil_wc_l1", NULL),
               *sw2 = orc_model_field(m, "soil_wc_l2", NULL), *anom = orc_model_field(m, "sst_anom", NULL);
        if (n != NG || orc_model_field(m, "no_such_field", NULL) != NULL) return 2;
        for (int j = 0; j < 48; ++j)
            for (int i = 0; i < 96; ++i) {
                const int p = i + 96 * j;
                const double lat = -87.0 + 174.0 * j / 47.0, land = (i > 20 && i < 50 && j > 10 && j < 40) ? 1.0 : 0.0;
                orog[p] = land * 1500.0 * exp(-0.01 * ((i - 35) * (i - 35) + (j - 25) * (j - 25)));
                lsm[p] = land;
                alb[p] = 0.2 + 0.3 * (fabs(lat) > 70.0);
                vegh[p] = 0.3 * land;
                vegl[p] = 0.4 * land;
                for (int mo = 0; mo < 12; ++mo) {
                    const double season = cos(6.283185307 * (mo - 0.5) / 12.0) * (lat > 0 ? -1.0 : 1.0);
                    const size_t q = (size_t)mo * NG + p;
                    stl[q] = 288.0 - 45.0 * (lat / 90.0) * (lat / 90.0) + 8.0 * season;
                    sst[q] = 271.5 + 28.0 * (1.0 - (lat / 90.0) * (lat / 90.0)) + 2.0 * season;
                    snow[q] = fabs(lat) > 55.0 ? 40.0 : 0.0;
                    ice[q] = fabs(lat) > 72.0 ? 0.8 : 0.0;
                    sw1[q] = 0.25;
                    sw2[q] = 0.22;
                }
                for (int pl = 0; pl < 5; ++pl) anom[(size_t)pl * NG + p] = 0.5 * sin(0.1 * i + pl);
            }
        if (orc_model_set_scalar(m, "increase_co2", 1.0) != 0 || orc_model_set_scalar(m, "no_such_flag", 1.0) != -1) return 3;
        if (orc_model_step(m) != -1) return 4;
        if (orc_model_init(m, 1983, 12, 31, 0, 0) != 0) return 5;
        int rc = 0, ymdhm[5], month_idx, imont1;
        double tmonth, tyear;
        for (int s = 0; s < 40 && rc == 0; ++s) rc = orc_model_step(m);
        orc_model_calendar(m, ymdhm, &month_idx, &imont1, &tmonth, &tyear);
        if (rc != 0 || ymdhm[0] != 1984 || ymdhm[1] != 1 || ymdhm[2] != 1 || month_idx != 2 || imont1 != 1) return 6;
        const double *olr = orc_model_field(m, "olr", NULL), *sst_am = orc_model_field(m, "sst_am", NULL);
        for (int p = 0; p < NG; ++p) acc += olr[p] + sst_am[p];
        acc += orc_model_get_scalar(m, "air_absortivity_co2") + tyear + tmonth;
        orc_model_free(m);
    }
    printf("oracle sanitize ok %.6e\n", acc);
    return isfinite(acc) ? 0 : 1;
}
