// Host-side pieces of the model that run once per day or once per run: calendar (model_control.f90), time interpolation
// weights (interpolation.f90), zonally averaged radiative forcing (shortwave_radiation.f90:218-322) and the member-independent
// constants of land_model_init / sea_model_init (land_model.f90:23-149, sea_model.f90:33-192).
// Compiled with floating-point contraction off; fp32 sub-expressions of the reference are evaluated in float.
#pragma once
#include <array>
#include <vector>

#include "tables.hpp"

namespace spd {

struct Calendar {  // Datetime_t + ControlParams_t, model_control.f90:19-47
    int year = 1982, month = 1, day = 1, hour = 0, minute = 0;
    int month_idx = 1;
    int imont1 = 1;
    double tmonth = 0.0, tyear = 0.0;
    void set(int y, int mo, int d, int h, int mi);  // initialize_control, :79-111
    void advance();                                 // advance_date, :114-160 (one 40-minute step)
    void update_forcing_params();                   // :162-185
};

struct TimeInterp {  // weights of forin5 (5-point, mean conserving) and forint (linear), interpolation.f90:39-93
    int m5[5];       // 0-based month indices im2, im1, imon, ip1, ip2
    double w5[5];
    int l0, l1;      // forint: for12(:, l0) + wlin * (for12(:, l1) - for12(:, l0))
    double wlin;
    // monthly_interp for the SST anomaly (no wrap-around): planes month_idx and month_idx -+ 1
    int a0, a1;
    double wan;
};
TimeInterp time_interp(const Calendar &c);

struct ZonalForcing {  // one value per latitude, shortwave_radiation.f90:218-322
    std::array<double, 48> flux_solar_in, flux_ozone_upper, flux_ozone_lower, zenit_correction, stratospheric_correction;
};
ZonalForcing zonal_average_fields(const HostTables &t, double tyear);

// land_model_init + sea_model_init (land_model.f90:23-148, sea_model.f90:33-192) run on the device, one workgroup per monthly plane
// and member (land_sea_init_kernel, surface.hip).  What does not depend on the member's fields is evaluated here, once, with the host's
// arithmetic -- the reference's default-real literals in float, no contraction, the host's cos() for the latitude-dependent
// heat capacities -- and handed to the kernel by value, so that the device only compares, selects, adds and multiplies.
struct LandSeaConsts {
    double thrsh, one_minus_thrsh, one;         // mask threshold 0.1 (default real), 1.0 - thrsh, 1.0
    double alb_thr, veg_low_weight, idep2;      // 0.4, 0.8 (default real), 3
    double swwil2, rsw;                         // land_model.f90:93-94
    double flandmin, fseamin;                   // 1./3. (default real)
    double rhcapl[2];                           // delt / hcapl (alb0 < 0.4), delt / hcapli
    double cdland[2], cdsea[2], cdice[2];       // d td / (1 + d td) for the domain mask d = 0, 1
    double rhcaps_row[48], rhcapi_row[48];      // delt / hcaps(j), delt / hcapi(j)
};
LandSeaConsts land_sea_consts(const HostTables &t);

// set_orog_land_sfc_drag, surface_fluxes.f90:324-334
void orog_land_sfc_drag(const std::vector<double> &phis0, std::vector<double> &forog);

}  // namespace spd
