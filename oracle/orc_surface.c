/* TEST INFRASTRUCTURE -- CPU oracle, NOT PART OF THE PRODUCT (see speedy_oracle.h).
 *
 * Plain-C restatement of the boundary-field preprocessing of the reference's initialisation:
 * fill_missing_values / check_surface_fields (speedy.f90/boundaries.f90:41-114), land_model_init
 * (land_model.f90:23-148) and sea_model_init (sea_model.f90:33-192).  Pinned bit for bit against
 * tests/golden/init.npz (the reference's own init on the example boundary file and on variants of
 * it with extra missing-value patterns; oracle/gen_golden_init.py), tests/test_oracle_init.py.
 * Default-real literals and sub-expressions of the reference are evaluated in float here, as the
 * reference's compiler does; build with -ffp-contract=off.
 */
#include <math.h>
#include <string.h>

#include "speedy_oracle.h"

enum { NG = ORC_IX * ORC_IL };

/* MAX / MIN as the reference's compiler evaluates them: one ordered comparison and a select (max(-0.0, 0.0) = +0.0) */
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* boundaries.f90:69-113.  `fmean` is a SAVEd local of the reference (:77, initialised to 0): it lives as long as the
 * process and is carried from row to row, from month to month and from the land call sequence into the sea one; it is read
 * before it is set only when the first row visited (j = il/2) has no valid point at all. */
static void fill_missing_values(double *sf, double fmis, double *fmean) {
    double sf2[ORC_IX + 2];
    int j1 = 0, j2, j3;
    for (int hemisphere = 1; hemisphere <= 2; ++hemisphere) {
        if (hemisphere == 1) {
            j1 = ORC_IL / 2; j2 = 1; j3 = -1;
        } else {
            j1 = j1 + 1; j2 = ORC_IL; j3 = 1;
        }
        for (int j = j1; j3 > 0 ? j <= j2 : j >= j2; j += j3) {
            double *row = sf + (size_t)ORC_IX * (j - 1);
            int nmis = 0;
            for (int i = 1; i <= ORC_IX; ++i) {
                sf2[i] = row[i - 1];
                if (row[i - 1] < fmis) {
                    ++nmis;
                    sf2[i] = 0.0;
                }
            }
            if (nmis < ORC_IX) {
                double s = 0.0;
                for (int i = 1; i <= ORC_IX; ++i) s += sf2[i];
                *fmean = s / (double)(float)(ORC_IX - nmis);
            }
            for (int i = 1; i <= ORC_IX; ++i)
                if (row[i - 1] < fmis) sf2[i] = *fmean;
            sf2[0] = sf2[ORC_IX];
            sf2[ORC_IX + 1] = sf2[1];
            for (int i = 1; i <= ORC_IX; ++i)
                if (row[i - 1] < fmis) row[i - 1] = 0.5f * (sf2[i - 1] + sf2[i + 1]);
        }
    }
}

/* boundaries.f90:41-64 (the count of out-of-range values is never used by the reference) */
static void check_surface_fields(const double *fmask, int nf, double fset, double *field) {
    for (int jf = 0; jf < nf; ++jf)
        for (int p = 0; p < NG; ++p)
            if (!(fmask[p] > 0.0)) field[(size_t)jf * NG + p] = fset;
}

void orc_land_sea_init(const orc_tables *t, int n_anom_planes, const double *fmask_orig, const double *alb0,
                       const double *veg_high, const double *veg_low, const double *soil_wc_l1, const double *soil_wc_l2,
                       double *stl12, double *snowd12, double *sst12, double *sea_ice_frac12, double *sst_anom,
                       double *soilw12, double *fmask_land, double *bmask_land, double *fmask_sea, double *bmask_sea,
                       double *rhcapl, double *cdland, double *rhcaps, double *rhcapi, double *cdsea, double *cdice,
                       double *fmean) {
    const double thrsh = 0.1f;
    const double delt = 86400.0f / 36; /* params.f90:33 */
    /* ---- land_model_init, land_model.f90:60-71: fractional and binary land masks */
    for (int p = 0; p < NG; ++p) {
        fmask_land[p] = fmask_orig[p];
        if (fmask_land[p] >= thrsh) {
            bmask_land[p] = 1.0;
            if (fmask_orig[p] > (1.0f - thrsh)) fmask_land[p] = 1.0;
        } else {
            bmask_land[p] = 0.0;
            fmask_land[p] = 0.0;
        }
    }
    /* :74-80 land-surface temperature and snow depth */
    for (int month = 0; month < 12; ++month) fill_missing_values(stl12 + (size_t)month * NG, 0.0, fmean);
    check_surface_fields(bmask_land, 12, 273.0, stl12);
    check_surface_fields(bmask_land, 12, 0.0, snowd12);
    /* :86-110 soil water availability from the two top layers and the vegetation fraction */
    const double swcap = 0.30f, swwil = 0.17f;
    const int idep2 = 3;
    const double swwil2 = idep2 * swwil;
    const double rsw = 1.0f / (swcap + idep2 * (swcap - swwil));
    for (int month = 0; month < 12; ++month)
        for (int p = 0; p < NG; ++p) {
            const size_t q = (size_t)month * NG + p;
            const double veg = dmax(0.0, veg_high[p] + 0.8f * veg_low[p]);
            const double swroot = idep2 * soil_wc_l2[q];
            soilw12[q] = dmin(1.0, rsw * (soil_wc_l1[q] + veg * dmax(0.0, swroot - swwil2)));
        }
    check_surface_fields(bmask_land, 12, 0.0, soilw12);
    /* :119-148 heat capacity and dissipation time of the soil / land-ice layer */
    const double tdland = 40.f, flandmin = (double)(1.f / 3.f);
    const double hcapl = 1.0f * 2.50e+6f, hcapli = 5.0f * 1.93e+6f;
    for (int p = 0; p < NG; ++p) {
        const double dmask = fmask_land[p] < flandmin ? 0.0 : 1.0;
        rhcapl[p] = alb0[p] < 0.4f ? delt / hcapl : delt / hcapli;
        cdland[p] = dmask * tdland / (1.f + dmask * tdland);
    }
    /* ---- sea_model_init, sea_model.f90:93-105: fractional and binary sea masks */
    for (int p = 0; p < NG; ++p) {
        fmask_sea[p] = 1.0f - fmask_orig[p];
        if (fmask_sea[p] >= thrsh) {
            bmask_sea[p] = 1.0;
            if (fmask_sea[p] > (1.0f - thrsh)) fmask_sea[p] = 1.0;
        } else {
            bmask_sea[p] = 0.0;
            fmask_sea[p] = 0.0;
        }
    }
    /* :108-124 SST, sea-ice concentration, SST anomalies (their first three planes) */
    for (int month = 0; month < 12; ++month) fill_missing_values(sst12 + (size_t)month * NG, 0.0, fmean);
    check_surface_fields(bmask_sea, 12, 273.0, sst12);
    for (size_t q = 0; q < (size_t)12 * NG; ++q) sea_ice_frac12[q] = dmax(sea_ice_frac12[q], 0.0);
    check_surface_fields(bmask_sea, 12, 0.0, sea_ice_frac12);
    if (n_anom_planes >= 3) check_surface_fields(bmask_sea, 3, 0.0, sst_anom);
    /* :146-187 heat capacities of the mixed layer and of sea ice by latitude; global domain: the smoothed mask is 1 */
    const float pih = asinf(1.0f);
    const double crad = (double)(pih / 90.f);
    const double depth_ml = 60.f, dept0_ml = 40.f, depth_ice = 2.5f, dept0_ice = 1.5f, tdsst = 90.f, tdice = 30.0f;
    const double fseamin = (double)(1.f / 3.f);
    for (int j = 0; j < ORC_IL; ++j) {
        const double deglat_s = t->radang[j] * 90.0f / pih; /* :108 */
        const double coslat = cos(crad * deglat_s);
        const double hcaps = 4.18e+6f * (depth_ml + (dept0_ml - depth_ml) * ((coslat * coslat) * coslat));
        const double hcapi = 1.93e+6f * (depth_ice + (dept0_ice - depth_ice) * (coslat * coslat));
        for (int i = 0; i < ORC_IX; ++i) {
            const int p = i + ORC_IX * j;
            const double dmask = fmask_sea[p] < fseamin ? 0.0 : 1.0;
            rhcaps[p] = delt / hcaps;
            rhcapi[p] = delt / hcapi;
            cdsea[p] = dmask * tdsst / (1.f + dmask * tdsst);
            cdice[p] = dmask * tdice / (1.f + dmask * tdice);
        }
    }
}
