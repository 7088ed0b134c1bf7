// Device pointers of the surface / coupler state (all [M][48][96] unless noted) and small by-value argument structs.
#pragma once
#include "device_tables.hpp"
#include "surface_host.hpp"

namespace spd {

struct SurfacePtrs {
    // monthly climatologies after land_model_init / sea_model_init: [M][12][48][96]
    double *stl12, *snowd12, *soilw12, *sst12, *sea_ice_frac12;
    double *sst_anom;  // [M][n_months+2][48][96]
    // land
    double *stlcl_obs, *snowdcl_obs, *soilwcl_obs, *stl_lm, *land_temp, *snow_depth, *soil_avail_water, *cdland, *rhcapl;
    // sea / ice
    double *sstcl_ob, *sicecl_ob, *ticecl_ob, *sstan_ob, *sst_om, *tice_om, *sice_om, *sst_am, *sstan_am, *sice_am, *tice_am,
        *ssti_om, *cdsea, *cdice, *rhcaps, *rhcapi, *hfseacl, *fmask_sea;
    // fluxes read by the slab models (physics outputs): hfluxn, shf, evap are [M][3][48][96]
    const double *hfluxn, *shf, *evap, *ssrd;
    // daily forcing
    double *flux_solar_in, *flux_ozone_upper, *flux_ozone_lower, *zenit_correction, *stratospheric_correction;
    double *snowc, *alb_land, *alb_sea, *alb_surface;
    const double *alb0, *fmask_land, *phis0;
};

// land_sea_init_kernel: the boundary fields of every member as the host stored them, and what land_model_init / sea_model_init
// make of them.  [M][48][96], monthly fields [M][12][48][96], sst_anom [M][anom_planes][48][96].
struct LandSeaPtrs {
    const double *fmask_orig, *alb0, *veg_high, *veg_low, *soil_wc_l1, *soil_wc_l2;  // read
    double *stl12, *snowd12, *sst12, *sea_ice_frac12, *sst_anom;                      // cleaned in place
    double *soilw12, *fmask_land, *bmask_land, *fmask_sea, *bmask_sea, *rhcapl, *cdland, *rhcaps, *rhcapi, *cdsea, *cdice;  // written
    int anom_planes;
};

// multi_copy_kernel: up to kCopyListMax device-to-device copies in one launch (the arrays of one member, model to model)
constexpr int kCopyListMax = 160;
struct CopyList {
    const char *src[kCopyListMax];
    char *dst[kCopyListMax];
    unsigned bytes[kCopyListMax];  // multiples of 16; both pointers 16-byte aligned
    int n;
};

struct ZonalDevice {
    double v[5][48];  // flux_solar_in, flux_ozone_upper, flux_ozone_lower, zenit_correction, stratospheric_correction
};

struct RestPtrs {
    double *vor, *div, *t, *tr, *ps;
    const double *phis, *spec_ps, *spec_q, *trfilt;
};

struct RestConsts {
    double gam1, tref, ttop, sqrt2;
    double fsg_rgam[8], fsg_qexp[8];
};

}  // namespace spd
