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
    printf("oracle sanitize ok %.6e\n", acc);
    return isfinite(acc) ? 0 : 1;
}
