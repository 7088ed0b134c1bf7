/* TEST INFRASTRUCTURE -- CPU oracle for the pySPEEDY hot path.  NOT PART OF THE PRODUCT.
 *
 * Plain-C restatement of the reference algorithm for the spectral-transform path
 * (speedy.f90/legendre.f90, fourier.f90, fftpack.f90, spectral.f90, geometry.f90), of the
 * per-column physics (physics.f90 and the scheme files it calls) and -- orc_dynamics.c, orc_surface.c,
 * orc_model.c -- of everything around them, up to the whole model (initialisation, do_single_step).
 * Each function cites the reference file:line it follows.  Parity is PINNED: tests/test_oracle_golden.py,
 * test_schemes_oracle.py, test_physics_oracle.py, test_step_oracle.py, test_oracle_init.py and
 * test_model_oracle.py compare every function, and whole runs, BIT FOR BIT with golden vectors captured from
 * the flang-compiled reference itself (oracle/build_ref.sh -> oracle/_ref/libspeedy_ref.so, vectors by
 * oracle/gen_golden*.py).
 *
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may use this library,
 * and only as the checker / baseline -- never on the product path.
 *
 * All arrays are Fortran column-major, exactly as the reference lays them out.
 */
#ifndef SPEEDY_ORACLE_H
#define SPEEDY_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_IX = 96, ORC_IL = 48, ORC_IY = 24, ORC_KX = 8, ORC_MX = 31, ORC_NX = 32, ORC_TRUNC = 30 };

typedef struct orc_tables {
    /* geometry.f90:16-47 */
    double hsg[9], dhs[8], fsg[8], dhsr[8], fsgr[8];
    double radang[48], coriol[48], sia[48], coa[48], sia_half[24], coa_half[24], cosgr[48], cosgr2[48];
    double sigl[8], sigh[9], grdsig[8], grdscp[8], wvi[16];
    /* legendre.f90:14-32 */
    double epsi[32 * 33], repsi[32 * 33], cpol[62 * 32 * 24], wt[24];
    int nsh2[32];
    /* fourier.f90:21-22 (slots rffti1 never writes are kept at 0 here; the reference leaves them unset) */
    double work[96];
    int ifac[15];
    /* spectral.f90:14-18 (gradym(:,1) is never set by the reference; 0 here) */
    double el2[31 * 32], elm2[31 * 32], el4[31 * 32], trfilt[31 * 32];
    double gradym[31 * 32], gradyp[31 * 32], uvdx[31 * 32], uvdym[31 * 32], uvdyp[31 * 32];
    double vddym[31 * 32], vddyp[31 * 32], gradx[31];
    /* longwave_radiation.f90:208-232, fband(100:400,4) */
    double fband[301 * 4];
} orc_tables;

void orc_tables_init(orc_tables *t);

/* transforms: real views, input(2*mx, nx|il), grid(ix, il) */
void orc_legendre_inv(const orc_tables *t, const double *in /*62x32*/, double *out /*62x48*/);
void orc_legendre(const orc_tables *t, const double *in /*62x48*/, double *out /*62x32*/);
void orc_fourier_inv(const orc_tables *t, const double *in /*62x48*/, double *out /*96x48*/, int kcos);
void orc_fourier(const orc_tables *t, const double *in /*96x48*/, double *out /*62x48*/);
void orc_spec2grid(const orc_tables *t, const double *spec /*complex 31x32*/, double *grid, int kcos);
void orc_grid2spec(const orc_tables *t, const double *grid, double *spec);
void orc_rfftf96(const orc_tables *t, double *c);
void orc_rfftb96(const orc_tables *t, double *c);

/* spectral-space operators: complex(31,32) as interleaved doubles */
void orc_vort2vel(const orc_tables *t, const double *vor, const double *div, double *ucos, double *vcos);
void orc_vel2vort(const orc_tables *t, const double *ucos, const double *vcos, double *vor, double *div);
void orc_grid_vel2vort(const orc_tables *t, const double *ug, const double *vg, double *vor, double *div, int kcos);
void orc_gradient(const orc_tables *t, const double *psi, double *psdx, double *psdy);
void orc_laplacian(const orc_tables *t, const double *in, double *out, int inverse);
void orc_truncate(const orc_tables *t, double *f);
void orc_grid_filter(const orc_tables *t, const double *fg1, double *fg2);

/* batched helpers used by the CPU baseline */
void orc_spec2grid_batch(const orc_tables *t, const double *spec, double *grid, int kcos, int nfields);
void orc_grid2spec_batch(const orc_tables *t, const double *grid, double *spec, int nfields);

/* ---- column physics (physics.f90:14-256).  One member; arrays (ix, il[, kx[, n]]) column-major. ---- */
typedef struct orc_phys_io {
    /* inputs: grid-point state (time level 1), physics.f90:89-101 */
    const double *ug, *vg, *tg, *qg_in, *phig, *pslg;
    /* in/out: tendencies, physics.f90:31-34 */
    double *utend, *vtend, *ttend, *qtend;
    /* surface / forcing inputs, physics.f90:177-185 */
    const double *fmask_land, *phis0, *forog, *sst_am, *alb_land, *alb_sea, *snowc, *land_temp, *soil_avail_water;
    /* shortwave inputs, shortwave_radiation.f90:88-168,212 */
    const double *flux_solar_in, *flux_ozone_upper, *flux_ozone_lower, *zenit_correction, *stratospheric_correction,
        *alb_surface;
    double air_absortivity_co2;
    int compute_shortwave;
    /* outputs every step */
    double *precnv, *precls, *cbmf, *slrd, *slr, *olr;
    double *slru, *ustr, *vstr, *shf, *evap, *hfluxn; /* (ix,il,3) */
    double *rad_st4a /* (ix,il,kx,2) */, *rad_flux /* (ix,il,4) */;
    /* persisted shortwave state (written on SW steps, read otherwise) */
    double *tt_rsw /* (ix,il,kx) */, *rad_tau2 /* (ix,il,kx,4) */, *rad_strat_corr /* (ix,il,2) */;
    double *tsr, *ssrd, *ssr, *qcloud_equiv;
    /* optional diagnostics (may be NULL) */
    int *iptop, *icltop;
    double *ts, *tskin, *u0, *v0, *t0, *cloudc, *clstr;
} orc_phys_io;

void orc_physics(const orc_tables *t, orc_phys_io *io);

/* individual schemes, same argument meaning as the reference routines */
void orc_qsat(const double *ta, const double *ps, double sig, double *qsat, int n);
void orc_convection(const orc_tables *t, const double *psa, const double *se, const double *qa, const double *qsat,
                    int *itop, double *cbmf, double *precnv, double *dfse, double *dfqa);
void orc_lsc(const orc_tables *t, const double *psa, const double *qa, const double *qsat, int *itop, double *precls,
             double *dtlsc, double *dqlsc);
void orc_clouds(const double *qa, const double *rh, const double *precnv, const double *precls, const int *iptop,
                const double *gse, const double *fmask, int *icltop, double *cloudc, double *clstr,
                double *qcloud_equiv);
void orc_shortwave(const orc_tables *t, orc_phys_io *io, const double *psa, const double *qa, const int *icltop,
                   const double *cloudc, const double *clstr);
void orc_lw_down(const orc_tables *t, const double *ta, double *fsfcd, double *dfabs, double *rad_flux,
                 const double *rad_tau2, double *rad_st4a);
void orc_lw_up(const orc_tables *t, const double *ta, const double *ts, const double *fsfcd, const double *fsfcu,
               double *fsfc, double *ftop, double *dfabs, double *rad_flux, const double *rad_tau2,
               const double *rad_st4a, const double *rad_strat_corr);
void orc_surface_fluxes(const orc_tables *t, const double *psa, const double *ua, const double *va, const double *ta,
                        const double *qa, const double *rh, const double *phi, const double *phi0,
                        const double *fmask, const double *forog, const double *tsea, const double *ssrd,
                        const double *slrd, double *ustr, double *vstr, double *shf, double *evap, double *slru,
                        double *hfluxn, double *tsfc, double *tskin, double *u0, double *v0, double *t0,
                        const double *alb_land, const double *alb_sea, const double *snowc, const double *land_temp,
                        const double *soil_avail_water);
void orc_vdiff(const orc_tables *t, const double *se, const double *rh, const double *qa, const double *qsat,
               const double *phi, const int *icnv, double *ut, double *vt, double *tt, double *qt);


/* ---- callers of the hot path ("next #1"): dynamics, implicit solver, time stepping ---- */
typedef struct orc_dyn_tables {
    /* horizontal_diffusion.f90:16-30 */
    double dmp[31 * 32], dmpd[31 * 32], dmps[31 * 32], dmp1[31 * 32], dmp1d[31 * 32], dmp1s[31 * 32];
    double tcorv[8], qcorv[8];
    /* implicit.f90:20-22 (dt-dependent parts are set by orc_dyn_set_time_step) */
    double tref[8], tref2[8], tref3[8], dhsx[8];
    double xc[64], xd[64], xj[64 * 64], elz[31 * 32]; /* xj(kx, kx, mx+nx+1) */
    /* geopotential.f90:16-31 */
    double xgeop1[8], xgeop2[8];
} orc_dyn_tables;

typedef struct orc_state {
    /* prognostic spectral state, complex: vor/div/t/tr (mx,nx,kx,2), ps (mx,nx,2), phi (mx,nx,kx), phis (mx,nx) */
    double *vor, *div, *t, *tr, *ps, *phi, *phis;
    double *tcorh, *qcorh; /* complex (mx,nx), forcing.f90:84,101 */
    orc_phys_io ph;        /* physics-side state; the grid-field / tendency pointers are filled per call */
} orc_state;

void orc_dyn_tables_init(const orc_tables *t, orc_dyn_tables *d);
void orc_dyn_set_time_step(const orc_tables *t, orc_dyn_tables *d, double dt);
void orc_geopotential(const orc_tables *t, const orc_dyn_tables *d, const double *tt, const double *phis, double *phi);
void orc_physics_from_spectral(const orc_tables *t, orc_state *s, int j1, double *utend, double *vtend, double *ttend,
                               double *qtend);
void orc_get_tendencies(const orc_tables *t, const orc_dyn_tables *d, orc_state *s, double *vordt, double *divdt,
                        double *tdt, double *psdt, double *trdt, int j2);
void orc_step(const orc_tables *t, const orc_dyn_tables *d, orc_state *s, int j1, int j2, double dt);
int orc_check_diagnostics(const orc_tables *t, const orc_state *s, int time_lev, double *diag);

/* ---- boundary-field preprocessing of the initialisation (orc_surface.c): land_model_init + sea_model_init.
 * Monthly fields (ix, il, 12) and sst_anom (ix, il, n_anom_planes) are cleaned in place; *fmean is the SAVEd running mean
 * of fill_missing_values (boundaries.f90:77), 0 in a fresh process. */
void orc_land_sea_init(const orc_tables *t, int n_anom_planes, const double *fmask_orig, const double *alb0,
                       const double *veg_high, const double *veg_low, const double *soil_wc_l1, const double *soil_wc_l2,
                       double *stl12, double *snowd12, double *sst12, double *sea_ice_frac12, double *sst_anom,
                       double *soilw12, double *fmask_land, double *bmask_land, double *fmask_sea, double *bmask_sea,
                       double *rhcapl, double *cdland, double *rhcaps, double *rhcapi, double *cdsea, double *cdice,
                       double *fmean);

/* ---- the whole model (orc_model.c): calendar, interpolation, daily forcing, land / sea / ice coupling, initialisation and
 * do_single_step around the pieces above.  Boundary fields are written into the model's arrays by name (orc_model_field:
 * orog, fmask_orig, alb0, veg_high, veg_low, stl12, snowd12, soil_wc_l1..3, sst12, sea_ice_frac12, sst_anom) before
 * orc_model_init; every state array of the reference that the run touches can be read back the same way. */
typedef struct orc_model orc_model;
orc_model *orc_model_new(int n_anom_planes /* n_months + 2 */);
void orc_model_free(orc_model *m);
double *orc_model_field(orc_model *m, const char *name, long *n_doubles);
int orc_model_set_scalar(orc_model *m, const char *name, double value); /* land_coupling_flag, sst_anomaly_coupling_flag, increase_co2, air_absortivity_co2 */
double orc_model_get_scalar(const orc_model *m, const char *name);      /* current_step, air_absortivity_co2, ablco2_ref, compute_shortwave */
int orc_model_init(orc_model *m, int year, int month, int day, int hour, int minute); /* initialize_state; 0 or -2 */
int orc_model_step(orc_model *m);                                                      /* do_single_step; 0, -1, -2 */
void orc_model_calendar(const orc_model *m, int *ymdhm /* 5 */, int *month_idx, int *imont1, double *tmonth, double *tyear);

#ifdef __cplusplus
}
#endif
#endif
