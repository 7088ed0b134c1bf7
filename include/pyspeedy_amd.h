/* pyspeedy_amd -- C ABI of the MI355X-native SPEEDY hot path (spectral transforms + column physics).
 *
 * Drop-in boundary, operator level ("inner boundary"): every entry point replaces
 * one type-bound procedure of the reference's ModSpectral_t (speedy.f90/spectral.f90:19-31) or the
 * column-physics driver (speedy.f90/physics.f90:14), with an explicit batch count added.  A Fortran
 * host binds these with ISO_C_BINDING (INTEGRATION.md shows the interface block); the Python host in
 * pyspeedy_amd/ binds them with ctypes.
 *
 * Conventions
 *   - All field pointers are DEVICE pointers (hipMalloc'ed memory on the handle's device) unless the
 *     name ends in _host.  Nothing is retained after a call returns; all work is ordered on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream).  No call synchronises the device.
 *   - Layout: batch slowest, the reference's Fortran order inside one field:
 *       spectral field  complex(8) (mx=31, nx=32)  -> 992 complex = 15872 B, index m + 31*n, re/im interleaved
 *       Fourier plane   real(8)    (2*mx=62, il=48)-> 23808 B, index r + 62*j
 *       grid field      real(8)    (ix=96, il=48)  -> 36864 B, index i + 96*j   (j=0 southernmost)
 *   - Return value: 0 = success, negative = SPD_E_* below.  No exceptions, no process exit.
 *   - Thread safety: calls on different handles are independent; a handle may be shared by host threads
 *     (it is immutable after spd_create).
 */
#ifndef PYSPEEDY_AMD_H
#define PYSPEEDY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPD_IX 96
#define SPD_IL 48
#define SPD_IY 24
#define SPD_KX 8
#define SPD_MX 31
#define SPD_NX 32
#define SPD_TRUNC 30

#define SPD_OK 0
#define SPD_E_ARG (-1)    /* bad argument (null pointer, negative count, unknown name) */
#define SPD_E_DEVICE (-2) /* HIP runtime error (no device, launch failure, ...) */
#define SPD_E_SIZE (-3)   /* caller buffer too small */

typedef struct spd_context *spd_handle;

/* ---- lifecycle ------------------------------------------------------------------------------
 * spd_create: builds the transform / geometry / radiation tables on the host exactly as the reference's
 * ModGeometry_initialize (geometry.f90:67), ModLegendre_initialize (legendre.f90:38), rffti1
 * (fftpack.f90:1), ModSpectral_initialize (spectral.f90:39) and radset (longwave_radiation.f90:208) do,
 * and uploads them to device `device`.  Replaces state%mod_geometry/mod_spectral%initialize
 * (initialization.f90:41-44). */
int spd_create(spd_handle *out, int device);
int spd_destroy(spd_handle h);
int spd_device(spd_handle h);
const char *spd_last_error(void);
const char *spd_version(void);

/* Host copy of a table by the reference's name ("sia_half", "cpol", "work", "el2", "fband", ...).
 * Doubles, Fortran order.  Integer tables ("nsh2", "ifac") are returned converted to double.
 * Returns the element count, or a negative error.  With buf == NULL only the count is returned.
 * h may be NULL: the tables are then built on the host without touching any device. */
long spd_get_table_host(spd_handle h, const char *name, double *buf_host, size_t buf_elems);

/* ---- spectral transforms (spectral.f90:251-273, legendre.f90:130-221, fourier.f90:63-123) ---- */
/* spec2grid: kcos == 1 -> no scaling, otherwise multiply row j by cosgr(j) (fourier.f90:87-91). */
int spd_spec2grid(spd_handle h, const double *spec, double *grid, int kcos, int nfields, void *stream);
int spd_grid2spec(spd_handle h, const double *grid, double *spec, int nfields, void *stream);
/* the two stages separately (same kernels, one stage disabled); `four` is the Fourier plane */
int spd_legendre_inv(spd_handle h, const double *spec, double *four, int nfields, void *stream);
int spd_legendre(spd_handle h, const double *four, double *spec, int nfields, void *stream);
int spd_fourier_inv(spd_handle h, const double *four, double *grid, int kcos, int nfields, void *stream);
int spd_fourier(spd_handle h, const double *grid, double *four, int nfields, void *stream);

/* ---- spectral-space operators (spectral.f90:134-317) ------------------------------------------ */
int spd_vort2vel(spd_handle h, const double *vor, const double *div, double *ucos, double *vcos, int nfields,
                 void *stream);
int spd_vel2vort(spd_handle h, const double *ucos, const double *vcos, double *vor, double *div, int nfields,
                 void *stream);
/* grid_vel2vort: kcos == 2 -> pre-multiply by cosgr, otherwise by cosgr2 (spectral.f90:229-243) */
int spd_grid_vel2vort(spd_handle h, const double *ug, const double *vg, double *vor, double *div, int kcos,
                      int nfields, void *stream);
int spd_gradient(spd_handle h, const double *psi, double *psdx, double *psdy, int nfields, void *stream);
int spd_laplacian(spd_handle h, const double *in, double *out, int inverse, int nfields, void *stream);
int spd_truncate(spd_handle h, double *field, int nfields, void *stream);
int spd_grid_filter(spd_handle h, const double *fg1, double *fg2, int nfields, void *stream);

/* ---- column physics (physics.f90:14-256 from line 107 on; the 41 spec2grid calls of lines 89-101 are
 *      issued by the caller through spd_vort2vel / spd_spec2grid so that they can be batched) ----------
 * One call processes `nmembers` ensemble members; every array below is member-major:
 * [nmembers][...reference shape...].  Shapes in comments are the reference's (Fortran order). */
typedef struct spd_physics_args {
    /* grid-point state at time level 1 (physics.f90:89-101) */
    const double *ug, *vg, *tg, *qg, *phig; /* (ix,il,kx) */
    const double *pslg;                     /* (ix,il)  log surface pressure */
    /* dynamics tendencies, updated in place (physics.f90:31-34) */
    double *utend, *vtend, *ttend, *qtend; /* (ix,il,kx) */
    /* surface and forcing fields (physics.f90:177-185) */
    const double *fmask_land, *phis0, *forog, *sst_am, *alb_land, *alb_sea, *snowc, *land_temp, *soil_avail_water;
    /* daily shortwave forcing (shortwave_radiation.f90:88-168, 212) -- read only when compute_shortwave != 0 */
    const double *flux_solar_in, *flux_ozone_upper, *flux_ozone_lower, *zenit_correction, *stratospheric_correction,
        *alb_surface;
    /* outputs written every step (ModelState_t fields of the same names) */
    double *precnv, *precls, *cbmf, *slrd, *slr, *olr;   /* (ix,il) */
    double *slru, *ustr, *vstr, *shf, *evap, *hfluxn;    /* (ix,il,3) ; hfluxn planes 1:2 written */
    double *rad_st4a;                                     /* (ix,il,kx,2) */
    double *rad_flux;                                     /* (ix,il,4) */
    /* radiation state that persists between shortwave steps (written when compute_shortwave != 0) */
    double *tt_rsw;         /* (ix,il,kx) */
    double *rad_tau2;       /* (ix,il,kx,4) */
    double *rad_strat_corr; /* (ix,il,2) */
    double *tsr, *ssrd, *ssr, *qcloud_equiv; /* (ix,il) */
    /* optional diagnostics, may be NULL: iptop/icltop as doubles would lose nothing but stay int32 */
    int32_t *iptop, *icltop;                  /* (ix,il) */
    double *ts, *tskin, *u0, *v0, *t0, *cloudc, *clstr; /* (ix,il) */
    double air_absortivity_co2; /* state%air_absortivity_co2 */
    int32_t compute_shortwave;  /* state%compute_shortwave (speedy.f90:53) */
    int32_t reserved;
} spd_physics_args;

int spd_physics(spd_handle h, const spd_physics_args *args, int nmembers, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PYSPEEDY_AMD_H */
