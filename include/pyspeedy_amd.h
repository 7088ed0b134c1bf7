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
#define SPD_E_TIMEOUT (-4) /* a collective did not complete inside its bound; work of unknown state is left on the devices it ran on */

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

/* Host-only (no device is touched): the model calendar, model_control.f90:79-185 -- initialize_control at the given start date,
 * then `nsteps` times advance_date (one 40-minute step each: February has 29 days when mod(year, 4) == 0, :135-142; the month
 * counter month_idx keeps counting across the year end, :153-157) with update_forcing_params (:162-185: imont1, tmonth, tyear in
 * the reference's default-real arithmetic).  Row 0 of every output is the state after initialize_control, row s the state
 * after s steps; ymdhm is [nsteps + 1][5] (year, month, day, hour, minute).  Any output pointer may be NULL.  This is the
 * calendar spd_model_step advances on the host beside the device state. */
int spd_calendar_walk(int year, int month, int day, int hour, int minute, int nsteps, int32_t *ymdhm, int32_t *month_idx,
                      int32_t *imont1, double *tmonth, double *tyear);
/* Host-only: the zonally uniform daily forcing of set_forcing / get_zonal_average_fields (forcing.f90:84-101,
 * shortwave_radiation.f90:218-322) for a fraction of the year `tyear`: out[5][48] = flux_solar_in, flux_ozone_upper,
 * flux_ozone_lower, zenit_correction, stratospheric_correction by latitude (south to north), as the model uploads them once
 * per simulated day. */
int spd_daily_forcing_host(double tyear, double *out);

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
    int32_t fp32; /* != 0: column arithmetic in single precision (cfg 5); inputs / outputs stay double */
    /* SPPT pattern (ix,il,kx), values outside [-1, 1] are clipped; NULL = off (physics.f90:234-248, sppt_on = .false.) */
    const double *sppt_pattern;
} spd_physics_args;

int spd_physics(spd_handle h, const spd_physics_args *args, int nmembers, void *stream);

/* ---- streaming-rate probe (measurement infrastructure; no counterpart in the reference -- SURVEY.md section 8d: "report
 *      measured copy / triad bandwidth on the box and use the spec peak for the contract fraction") -------------------------
 * What a kernel that only moves bytes reaches on this device in a given SHAPE, with kernels of this library (csrc/
 * stream_probe.hip): one wavefront per 64-thread workgroup; a wavefront requests `in_flight` rows (64 lanes x lane_bytes,
 * contiguous) of each of its `reads` input streams back to back, adds them and stores `in_flight` rows to each of its
 * `writes` output streams, and repeats until it has touched about `rows_per_wave` rows of all its streams together (1: a
 * plain copy kernel's one row and out; 243: the column kernel's life).  The streams lie far apart in memory.  The probe
 * allocates total_bytes itself, launches 2 + reps times and times every launch by the time stamps of its dispatch packet.
 * The step's kernels are priced against 8 TB/s in bench.py's `roofline`; this is the ceiling to read those fractions against. */
typedef struct spd_stream_probe_args {
    int32_t reads, writes;  /* streams: 1:1 (copy), 2:1 (the column kernel's mix), 3:2, 1:0 (read only), 0:1 (write only) */
    int32_t lane_bytes;     /* 8 (the column kernel: one double per lane) or 16 (the transforms' staging) */
    int32_t in_flight;      /* rows per stream requested before the first use: 1, 2, 4, 8, 16 */
    int32_t nontemporal;    /* 0 / 1: the non-temporal hint on loads and stores (csrc/stream_store.hpp) */
    int32_t waves_per_simd; /* 1 ... 8 wavefronts per SIMD, held there by dynamic LDS (2: the column kernel at 256 VGPRs) */
    int32_t rows_per_wave;  /* rows of all streams together a wavefront works through before it ends */
    int32_t reps;           /* timed launches, 1 ... 1000 */
    int32_t layout;         /* 0: a wavefront walks a contiguous chunk of its own in every stream; 1: the column kernel's -- row r of every
                             * wavefront lies in array r of the stream, consecutive wavefronts side by side inside each array */
    int32_t reserved;
    uint64_t total_bytes;   /* bytes one launch moves, reads + writes (rounded down to whole wavefronts) */
} spd_stream_probe_args;
/* mean and minimum duration of a launch in microseconds; optionally the bytes a launch really moved and its workgroups */
int spd_stream_probe(spd_handle h, const spd_stream_probe_args *args, double *mean_us, double *min_us, uint64_t *bytes_moved,
                     uint64_t *workgroups);

/* ---- ensemble model object: device-resident state of M members and the model time step ------------------------
 * Replaces, for M members at once, the reference's ModelState_t container (model_state.f90) and
 * time_stepping.f90:38-147 `step` (get_tendencies -> horizontal diffusion -> leapfrog + Robert/Williams filter), i.e. what
 * speedy_driver.f90.j2:43-79 (`step`, `parallel_step`) reach through do_single_step.  The state never leaves HBM; every
 * array is member-major with the reference's Fortran order inside a member, so get/set are plain copies.
 * Variable names are the registry names of registry/model_state_def.py (vor, div, t, tr, ps, phi, phis, precnv, ...,
 * rad_tau2, ...); two arrays the registry does not expose are added: tcorh, qcorh (mod_implicit%tcorh/qcorh,
 * forcing.f90:84,101). */
typedef struct spd_model *spd_model_handle;

int spd_model_create(spd_handle h, int nmembers, spd_model_handle *out);
/* (waits for the device; the model's memory block is kept by its context for the next model of the same size, up to 1 GiB per
 * context, and returned to the device with the context) */
int spd_model_destroy(spd_model_handle m);
int spd_model_members(spd_model_handle m);
/* device memory of the model, all members: bytes reserved (a few large blocks the arrays are carved from) and bytes in use */
int spd_model_memory(spd_model_handle m, size_t *bytes_reserved, size_t *bytes_used);
/* bytes of one member's copy of `name`, or a negative error */
long spd_model_var_bytes(spd_model_handle m, const char *name);
/* host <-> device copy of one member's array (get_<v>/set_<v> of speedy_driver.f90.j2:250-334); member = -1 in
 * spd_model_set broadcasts the same host array to every member.  Synchronous. */
int spd_model_set(spd_model_handle m, const char *name, int member, const void *host_buf, size_t bytes);
int spd_model_get(spd_model_handle m, const char *name, int member, void *host_buf, size_t bytes);
/* Device base pointer of a registry array ([nmembers][...]), for zero-copy users.  Contract:
 *  - the address stays valid (and stays the array of that name) until spd_model_destroy, except "sst_anom" after
 *    spd_model_init_sst_anom and the SPPT arrays before spd_model_set_sppt;
 *  - taking the address of "phi" pins the geopotential to one buffer: the model stops alternating between two (the
 *    look-ahead it otherwise uses for launches of up to 8 members), which costs such ensembles 1-3 % per step;
 *  - the kernels of a step are stream-ordered on the stream given to spd_model_step: read or write through the pointer
 *    on that stream, or synchronise first;
 *  - the call itself drops what the model had derived from the state (look-ahead geopotential, the day's interpolated
 *    climatologies).  A caller that WRITES through a pointer it obtained earlier -- spectral t / phis, a climatology or
 *    anomaly field, anything -- must call spd_model_invalidate before the next step, or the step uses the stale values. */
void *spd_model_device_ptr(spd_model_handle m, const char *name);
/* the state was changed behind the model's back (a write through a device pointer): drop everything derived from it */
int spd_model_invalidate(spd_model_handle m);
int spd_model_set_co2(spd_model_handle m, double air_absortivity_co2);
/* current value (the daily forcing raises it when increase_co2 is set, forcing.f90:52-57) */
double spd_model_co2(spd_model_handle m);
/* ModImplicit_set_time_step (implicit.f90:83-218): rebuilds the dt-dependent tables (8x8 inversions on the host) */
int spd_model_set_time_step(spd_model_handle m, double dt);
/* time_stepping.f90 `step(state, j1, j2, dt)` for all members; j1, j2 = 1 or 2 as in the reference */
int spd_model_step_dynamics(spd_model_handle m, int j1, int j2, double dt, int compute_shortwave, void *stream);
/* check_diagnostics (diagnostics.f90:16-76) for every member: error_codes_host[i] = 0 or -2; diag_host may be NULL or
 * receive [nmembers][3][kx] (eddy KE of vor, of div, global-mean T).  Synchronises `stream`. */
int spd_model_check(spd_model_handle m, int time_level, int32_t *error_codes_host, double *diag_host, void *stream);
/* The same check without stalling the launch pipeline: _begin enqueues it and returns a slot (0 or 1, whichever is
 * free; at most two in flight: a third _begin fails with SPD_E_ARG until one has been ended), _end waits for that slot only.  Begin the check of
 * step k, launch step k + 1, then end the check of step k. */
int spd_model_check_begin(spd_model_handle m, int time_level, void *stream);
int spd_model_check_end(spd_model_handle m, int slot, int32_t *error_codes_host);
/* spd_model_check_begin without a launch of its own: the check is put off and rides in the first launch of the NEXT
 * spd_model_step on this stream (`members` more workgroups of its spectral -> grid launch), on the state exactly as it is now --
 * for hosts that collect the check of step k after they have enqueued step k + 1 (spd_parallel_step_begin / _end: the range check
 * then costs the step's stream nothing, 5 % of the step at 64 members, 11 % at one).  Anything else that would read or change the
 * state first -- spd_model_set, the export transforms, member copies, a call of several steps in member groups, spd_model_check_end
 * itself -- makes it a launch of its own there and then.  Returns the slot for spd_model_check_end. */
int spd_model_check_defer(spd_model_handle m, int time_level, void *stream);
/* launches the check spd_model_check_defer put off under `slot` (-1: whichever one is waiting) now, if it is still waiting for a
 * step to carry it; a check put off under another slot keeps waiting for its step.  Afterwards spd_model_check_end of that slot
 * is a pure wait (a host may then make that call outside the lock it serialises its other calls on this model with) */
int spd_model_check_settle(spd_model_handle m, int slot);
/* how many of the model's begun / deferred range checks went out as launches of their own and how many rode in a step's launch */
int spd_model_check_counts(spd_model_handle m, int32_t *alone, int32_t *rode);
int spd_model_checks_in_flight(spd_model_handle m); /* 0, 1 or 2: checks begun and not yet ended */

/* initialize_state (initialization.f90:13-91) for every member from the boundary fields stored beforehand with
 * spd_model_set (orog, fmask_orig, alb0, veg_high, veg_low, stl12, snowd12, soil_wc_l1, soil_wc_l2, sst12,
 * sea_ice_frac12 and optionally sst_anom, i.e. what pyspeedy/speedy.py:279-296 sets): land/sea preprocessing, spectrally
 * truncated orography, resting reference atmosphere, coupler initialisation, forcing, first_step. */
int spd_model_init(spd_model_handle m, int year, int month, int day, int hour, int minute, void *stream);
/* do_single_step (speedy.f90:20-74) `nsteps` times for all members: daily forcing, shortwave every third step, leapfrog
 * step, date advance, land/sea coupling.  Stream-ordered, no synchronisation; returns SPD_E_ARG when the state was not
 * initialised (the reference's error code -1).  spd_model_check runs the reference's per-step range check on demand.
 * A call of several steps on a model of 20 members or more issues the members in 2 or 3 groups on streams of the model's own
 * (forked from and joined to `stream` around the call).  The FIRST such call creates them and spends a few milliseconds making
 * sure they sit on different hardware queues (it measures: HIP's hand-out depends on every stream the process created before,
 * and two group streams on one queue run one after the other); that first call blocks the host for that long. */
int spd_model_step(spd_model_handle m, int nsteps, void *stream);
/* The same `nsteps` steps with the range check of EVERY step (diagnostics.f90:16-76, which the reference's time loop runs after each
 * do_single_step: pyspeedy/speedy.py:396-405) recorded by the device: for hosts that know they will not look at the state before
 * nsteps steps have passed (no callback due before).  One call instead of nsteps -- with the member groups and rounds of the
 * multi-step plan -- and still every step's code: the check of step k rides in the spectral -> grid launch of step k + 1, the last
 * one is a launch of its own.  _begin enqueues everything and returns; _end waits and reports per member the first step of the call
 * whose check failed (0-based; -1: none) and, in `accepted` ([members][7], may be NULL), the model's step counter, date (year,
 * month, day, hour, minute) and month index after the member's last ACCEPTED step (the one before its first failure, else the
 * last of the call).  The device does not stop at a failed check: the steps behind it run on a state the model does not accept;
 * after a failure the only defined continuation is spd_model_init.  One such call may be in flight per model; nsteps <= 4096. */
int spd_model_step_checked_begin(spd_model_handle m, int nsteps, void *stream);
int spd_model_step_checked_end(spd_model_handle m, int32_t *first_failed_step, int32_t *accepted);
int spd_model_current_step(spd_model_handle m);
int spd_model_get_date(spd_model_handle m, int *ymdhm /* 5 ints */);
/* declare a state loaded through spd_model_set as initialised at the START of a run: step counter and date as given, month
 * index 1, CO2 reference = the current absorptivity (what set_forcing(imode = 0) does, forcing.f90:40) */
int spd_model_mark_initialized(spd_model_handle m, int current_step, int year, int month, int day, int hour, int minute);
/* Everything the model keeps on the host between steps (ControlParams_t / Datetime_t of model_control.f90:19-47, the
 * registry scalars of model_state_def.py:305-423 and the SPPT generator position).  Together with the registry arrays
 * (spd_model_get / _set) this IS the state of a run: _get + the arrays make a checkpoint, _set + the arrays resume it
 * bit for bit at any step -- after a month boundary (month_idx selects the sst_anom planes), with the CO2 trend on
 * (ablco2_ref is the untrended reference value) and with SPPT on (the AR(1) pattern sppt_spec continues at sppt_step).
 * spd_model_set_control marks the model initialised and resets nothing. */
typedef struct spd_model_control {
    int32_t current_step;
    int32_t year, month, day, hour, minute; /* model date */
    int32_t month_idx;                      /* months started since the run began, + 1: plane of sst_anom */
    int32_t land_coupling_flag, sst_anomaly_coupling_flag, increase_co2;
    int32_t sppt_on, sppt_first, physics_fp32;
    int32_t reserved;
    int64_t sppt_step, sppt_first_member_id;
    uint64_t sppt_seed;
    double air_absortivity_co2, ablco2_ref;
} spd_model_control;
int spd_model_get_control(spd_model_handle m, spd_model_control *out);
int spd_model_set_control(spd_model_handle m, const spd_model_control *in);
/* measurement hook: HIP events on the launch stream for the kernels of every step.  level 0 = off, 1 = the dominant kernel
 * only (the 77*M-field spectral->grid launch), 2 = every kernel of the step.  The events of a step kernel are attached to its
 * dispatch (hipExtLaunchKernel): they hold the kernel's own begin / end time stamps, as rocprofv3's kernel trace does, and no
 * marker packets sit between the launches (only the daily forcing, three launches, is bracketed by recorded events).  While the level
 * is not 0 the step is issued as ONE member group on the caller's stream (the serial plan), whatever "member_groups" says:
 * kernels of overlapping groups share the GPU and their durations would not be their own.
 * _read synchronises the events of the spectral->grid launches and returns their mean time in ms;
 * _read_kernels returns, per kernel id SPD_K_*, mean and minimum bracket time in ms, the number of brackets and the units
 * (fields for the transforms, members otherwise) one bracket processed; all four arrays hold SPD_K_COUNT entries. */
#define SPD_K_GEOPOTENTIAL 0  /* geopotential_kernel */
#define SPD_K_SPEC2GRID 1     /* spec2grid_table_kernel, 91 (77 pruned) fields per member */
#define SPD_K_COLUMN_SW 2     /* fused grid-point dynamics + column physics, shortwave step */
#define SPD_K_COLUMN 3        /* the same on the two steps out of three without shortwave */
#define SPD_K_GRID2SPEC 4     /* grid2spec_table_kernel, 73 fields per member */
#define SPD_K_SPECTRAL_STEP 5 /* spectral_step_kernel */
#define SPD_K_COUPLER 6       /* coupler_kernel */
#define SPD_K_FORCING 7       /* daily: forcing_kernel + the two 1-field-per-member transforms of tcorh / qcorh */
#define SPD_K_SPPT 8          /* SPPT on: AR(1) update + 8 fields per member spectral->grid */
#define SPD_K_DYN_GRID 9      /* split mode: dyn_grid_kernel */
#define SPD_K_PHYSICS_SW 10   /* split mode: physics_kernel, shortwave step */
#define SPD_K_PHYSICS 11      /* split mode: physics_kernel, other steps */
#define SPD_K_COUNT 12
int spd_model_profile(spd_model_handle m, int level);
int spd_model_profile_read(spd_model_handle m, double *mean_ms, int *launches, int *fields_per_launch);
int spd_model_profile_read_kernels(spd_model_handle m, double *mean_ms, double *min_ms, int *launches, int *units);
/* how the step is configured (environment switches read at spd_model_create): cfg[0] = spectral->grid transforms per member
 * and step (77, or the reference's 91 with PYSPEEDY_AMD_PRUNE_DEAD=0), cfg[1] = 1 when every step stores the diagnostics-only
 * physics outputs (PYSPEEDY_AMD_DIAG_EVERY_STEP=1; default 0: only the last step of a multi-step call does), cfg[2] = member
 * groups stepped on separate streams (PYSPEEDY_AMD_CHUNKS / "member_groups"; the configured number), cfg[3] = 1 for separate dynamics / physics launches, cfg[4] = 1
 * when spectral_step_kernel also computes the next step's geopotential, cfg[5] = 1 when it carries the land / sea-ice coupling,
 * cfg[6] = 1 for fp32 arithmetic in the column physics, cfg[7] = 1 while the arrays only the column physics reads back are stored
 * as fp32 (spd_model_set_physics_precision) */
int spd_model_get_config(spd_model_handle m, int32_t *cfg /* 8 */);
/* the streams the member groups of multi-step calls are issued on: how many the model has created so far, and whether each was
 * measured to run side by side with the others when it was created (0: after several replacements two of them still shared a
 * hardware queue -- their groups then run one after the other; PYSPEEDY_AMD_STREAMS_APART=2 reports the measurements) */
int spd_model_group_streams(spd_model_handle m, int32_t *created, int32_t *apart);
/* The launch-plan switches that can change on a live model, by name (the environment variables of README.md set the same
 * fields when the model is created; none of them changes the state a step leaves behind):
 *   "diag_every_step"      0 / 1   store the diagnostics-only physics outputs on every step of a multi-step call
 *   "coupler_in_spectral"  0 / 1   land / sea-ice coupling as tail blocks of spectral_step_kernel or as a launch of its own
 *   "spectral_early"      -1 / 0 / 1   spectral_step_kernel with all loads up front: automatic (up to 8 members) / never / always
 *   "split_dyn"            0 / 1   separate launches for grid-point dynamics and column physics
 *   "member_groups"        1 ... 4 the members are stepped in that many groups on separate HIP streams (default: 1 below 20
 *                                  members, 2 for 20 ... 23 and from 64 up, 3 for 24 ... 63; always 1 while spd_model_profile is on, for calls of a single step and with split_dyn)
 *   "block_members"        0, n    (default 32, PYSPEEDY_AMD_BLOCK_MEMBERS) from 4 n members up a multi-step spd_model_step takes the
 *                                  members in rounds of member_groups x n, a round through all steps of the call before the next
 *                                  starts (the cross-step hand-over of a group's spectral state then stays in the Infinity Cache;
 *                                  bitwise the same state); 0: everybody together
 *   "physics_storage32"    0 / 1   (default 1, PYSPEEDY_AMD_PHYS_STORE32) with spd_model_set_physics_precision(m, 1): keep the arrays
 *                                  only the column physics reads back as fp32 in memory (1) or as fp64 (0: same arithmetic, same
 *                                  bits in the state, 13 % more bytes in the column kernel); converts the arrays when it changes
 *   "prepare_multi_step"   1       create the streams of the member groups now instead of at the first multi-step call (write-only)
 *   "fail_launch_after"    n, -1   fault injection for tests: the (n + 1)-th launch sequence of a member group from now on fails like a
 *                                  device error (-1: off); spd_model_init clears it (write-only)
 * Returns SPD_E_ARG for an unknown name or a value outside the list.  What is fixed at creation (the pruned transform
 * table, the geopotential fold) is read from the environment only. */
int spd_model_set_option(spd_model_handle m, const char *name, int32_t value);
/* ... and read back, by the same names */
int spd_model_get_option(spd_model_handle m, const char *name, int32_t *value);
/* BASELINE cfg 5: fp32 != 0 runs the arithmetic of the column physics (physics.f90:107-256 and the schemes it calls) in
 * single precision; the model state, the grid-point dynamics and the tendencies handed to the transforms stay fp64 (the
 * physics increment is formed in fp32 and added to the fp64 dynamics tendency).  Not bitwise comparable with the reference:
 * tests/test_cfg5_gpu.py states the error bounds.  Default 0.
 * The arrays that only the column physics reads back change their STORAGE with it: its grid-point inputs at the physics' time
 * level (the work arrays t/q/phi/u/v_grid_phys, pslg_phys), the radiation state a shortwave step leaves for the next two steps
 * (tt_rsw, rad_tau2, rad_strat_corr) and the diagnostics-only outputs rad_st4a, rad_flux, precnv, precls, cbmf, slrd, slr, olr,
 * slru, ustr, vstr are kept as fp32 in the first half of their allocations (the fp32 kernel narrows each of these values before
 * it uses it and computes each one it stores in fp32: nothing is lost, 13 % of the column kernel's and 11 % of the
 * spectral -> grid launch's bytes are).  spd_model_get / _set (and the driver's spd_get / spd_set) keep speaking fp64 and
 * convert; spd_model_device_ptr hands out the array as stored: ask spd_model_var_storage (8 or 4 bytes per element).
 * Switching converts the arrays in place (synchronises the device); spd_model_set_control with another physics_fp32 switches
 * the model first, and spd_model_copy_member makes the receiving model take over the source's physics precision (its members
 * then all run with it: members of one model share their control block). */
int spd_model_set_physics_precision(spd_model_handle m, int fp32);
int spd_model_var_storage(spd_model_handle m, const char *name);
/* registry scalars land_coupling_flag, sst_anomaly_coupling_flag, increase_co2 (model_state_def.py:305-418) */
int spd_model_set_flags(spd_model_handle m, int land_coupling_flag, int sst_anomaly_coupling_flag, int increase_co2);

/* Grid-space views of the prognostic state in output units, members [first, first + count)
 * (prognostics.f90:125-219; `transform_spectral2grid`, `transform_grid2spectral`, `apply_grid_filter` of
 * speedy_driver.f90.j2:94-125).  Variables u_grid, v_grid, t_grid, q_grid (kg/kg), phi_grid (m), ps_grid (Pa). */
int spd_model_spectral2grid(spd_model_handle m, int first, int count, void *stream);
int spd_model_grid2spectral(spd_model_handle m, int first, int count, void *stream);
int spd_model_grid_filter(spd_model_handle m, int first, int count, void *stream);
/* One grid-space registry variable ((ix, il) or (ix, il, kx) per member, e.g. "t_grid" after spd_model_spectral2grid) of the
 * members [first, first + count) as a NetCDF-3 file carries it -- float32, BIG-endian, vertical levels bottom-up (the reference's
 * export conventions, pyspeedy/speedy.py:415-477) -- into dst_device[count][levels][48][96] (4 bytes each), stream-ordered.  For
 * hosts that write files: narrowing, level order and byte order happen on the GPU, and what crosses PCIe is the file's payload. */
int spd_model_export_pack(spd_model_handle m, const char *name, int first, int count, void *dst_device, size_t dst_bytes,
                          void *stream);
/* modelstate_init_sst_anom (speedy_driver.f90.j2:225-237): sst_anom(ix, il, 0:n_months+1) per member, zero-filled */
int spd_model_init_sst_anom(spd_model_handle m, int n_months);
/* Stochastically perturbed parametrisation tendencies (sppt.f90; compile-time off and non-functional in the reference:
 * PARITY UNPINNED, see csrc/sppt.hip).  Deterministic: the noise is a function of (seed, first_member_id + member, step,
 * level, coefficient).  While on, every step advances the AR(1) spectral pattern (registry names sppt_spec, sppt_pattern). */
int spd_model_set_sppt(spd_model_handle m, int on, uint64_t seed, int64_t first_member_id);
/* device-to-device copy of every registered variable of one member into a member of another model on the same GPU */
int spd_model_copy_member(spd_model_handle dst, int dst_member, spd_model_handle src, int src_member, void *stream);
/* the named registry variables only; the two models may live on different GPUs (device-to-device over xGMI,
 * hipMemcpyPeerAsync): how one process hands the shared boundary fields to the members it keeps on its other devices */
int spd_model_copy_vars(spd_model_handle dst, int dst_member, spd_model_handle src, int src_member, const char *const *names,
                        int n_names, void *stream);
/* the same copies enqueued only: no device is synchronised first, so the caller must have made sure that nothing in flight on
 * either device still uses the arrays (spd_broadcast_boundary synchronises every device once, then enqueues all its copies);
 * `stream` is a stream of the DESTINATION device, which is the current device when the call returns */
int spd_model_copy_vars_enqueue(spd_model_handle dst, int dst_member, spd_model_handle src, int src_member,
                                const char *const *names, int n_names, void *stream);
/* ONE collective broadcast of the named (fp64) variables between device models that live on different GPUs of this process:
 * member members[root] of models[root] into member members[i] of every other models[i] -- RCCL (ncclBroadcast in one group call,
 * single-process communicators) over xGMI, on each device's null stream; the caller synchronises the devices before and
 * after.  RCCL (librccl.so.1) is loaded when this is first called; SPD_E_DEVICE with the reason when it cannot be.  n = 1 is a
 * broadcast to nobody (it still initialises the communicator).  Every (variable, model) pair is validated before RCCL is
 * touched (SPD_E_ARG / SPD_E_SIZE: nothing was enqueued).  RCCL's initialisation, its group call and the completion of the
 * broadcasts are each waited for PYSPEEDY_AMD_RCCL_TIMEOUT seconds (default 30) at most: an initialisation that does not come
 * back gives SPD_E_DEVICE (nothing enqueued: the caller may copy point to point instead, and RCCL is not tried again in this
 * process), a group call or broadcast that does not come back SPD_E_TIMEOUT (nothing may be queued behind it). */
int spd_model_broadcast_vars(const spd_model_handle *models, const int *members, int n, int root, const char *const *names,
                             int n_names);

#ifdef __cplusplus
}
#endif
#endif /* PYSPEEDY_AMD_H */
