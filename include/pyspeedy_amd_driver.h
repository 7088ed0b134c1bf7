/* pyspeedy_amd -- outer boundary: the procedures of the reference's f2py module `speedy_driver`
 * (registry/templates/speedy_driver.f90.j2), one C entry point per procedure, same argument meaning.
 *
 * The reference hands Python 64-bit "containers" holding the bits of a Fortran pointer (speedy_driver.f90.j2:38-40, 222);
 * here a container is a 64-bit key into the library's own table.  State containers own device memory on the HIP device that
 * is current when they are created; nothing is retained from caller buffers; every call is synchronous (it returns after
 * the GPU work it started has finished), exactly like the Fortran routines it replaces.  All functions return SPD_OK or a
 * negative SPD_E_* code (pyspeedy_amd.h) -- that is the status of the CALL; the model's own error code of
 * init / step / check (error_codes.f90:7-9: 0 success, -1 state not initialised, -2 variables out of range) is written
 * to `error_code`, as the reference's `intent(out) :: error_code` arguments are.  One more value exists for parallel_step
 * over several device models: -3 for the members of a model whose step could not be issued or checked at all (the call then
 * returns that model's SPD_E_* code after the other models have been stepped; their members have their usual codes).
 *
 * Batching.  The reference steps an ensemble with an OpenMP loop over independent containers (parallel_step, :58-79).
 * Here spd_parallel_step advances the members with one set of kernel launches per DEVICE MODEL: the first time it is handed
 * independent, initialised containers it gathers them (device-to-device copies, once) into batched device models -- per
 * device, and per set of containers that agree in date, step counter, control flags and anomaly length; 32 or more of them
 * on one device become two models -- and rebinds the containers to the members of those models; get / set / check / transforms
 * keep working per container.  A call over several models enqueues the step and range check of every model (each on a stream
 * of its own) before it waits for any, and does not hold the library's lock while it waits: devices, models and host threads
 * that step other containers work side by side.  spd_step on a single member of a batch, or a parallel_step over a different
 * grouping, takes that batch apart again first (correct, but it gives the batching up).
 * spd_modelstate_init_ensemble creates n containers that are batched from the start.
 *
 * Current device.  Entry points that touch a container living on another GPU switch the calling thread's HIP device while
 * they work and restore the device that was current when they were called before they return: the caller's current device
 * (and with it e.g. torch.cuda.current_device()) never changes under it, whatever devices an ensemble is spread over.
 *
 * A step that went out without its check.  Everything that can refuse a step (a free check slot, the control block) is asked
 * before the step is enqueued.  If a DEVICE error strikes between the enqueue of the step and the hand-over of its codes, the
 * members get -3, their dates stay, and the device model is marked: its state has moved on while date and codes say it has not,
 * and every later step of it fails (SPD_E_ARG, code -3) until its members are initialised again (spd_init).  In the begin / end
 * form a member whose check of step k fails (-2) gets the date from before step k back even if step k + 1 has been begun on
 * it meanwhile: after -2 the state is outside the model's accepted range, and the only defined continuation is a new spd_init.
 *
 * Threads.  Every entry point may be called from any host thread; calls on the SAME container (or on containers that share a
 * device model) must not overlap in time -- the reference's `!f2py threadsafe` contract.
 *
 * Date: as in the reference the CONTROL container owns the model date (ControlParams_t%model_datetime, month_idx,
 * model_control.f90:38-47): step / parallel_step take the date from it and advance it; a member whose step fails keeps its
 * date (speedy.f90:57-71 returns before advance_date).
 */
#ifndef PYSPEEDY_AMD_DRIVER_H
#define PYSPEEDY_AMD_DRIVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ModelState interface (speedy_driver.f90.j2:216-248) ---- */
int spd_modelstate_init(int64_t *state_cnt);
/* extension: n containers that are batched from the start.  Up to 31 members of a device form ONE device model; from 32 up they
 * are kept as TWO (first half / second half; parallel_step enqueues both before it waits for either, which is worth 8 % per step
 * at 64 members; PYSPEEDY_AMD_DRIVER_SPLIT=0 keeps one model, =n splits from n members).  With a device placement of k > 1
 * devices: member e of n on device e k / n, one or two device models per device.  A host that reaches for the device arrays
 * (spd_driver_model) must therefore go through the containers, not assume that the first container's model holds everybody. */
int spd_modelstate_init_ensemble(int64_t *state_cnts, int32_t n_members);
/* the same with the number of devices as an argument (0: the calling thread's current device; k: devices 0 .. k-1 in blocks):
 * the process-wide placement below is neither read nor changed */
int spd_modelstate_init_ensemble_on(int64_t *state_cnts, int32_t n_members, int32_t n_devices);
/* ... and as ONE device model per device whatever the number of members (n_devices < 0: the process-wide placement, 0: the current
 * device, k: devices 0 .. k-1 in blocks): for hosts that hand over many steps at once (spd_parallel_steps_begin; what SpeedyEns
 * creates), whose device model forms its member groups, offsets and rounds itself.  Either way the grouping follows the host: a
 * multi-step call over the two halves of a device merges them into one model, single steps over a model that was made or merged
 * whole halve it -- device-to-device copies, a few milliseconds, once per change of habit (csrc/driver.cpp: regroup). */
int spd_modelstate_init_ensemble_whole(int64_t *state_cnts, int32_t n_members, int32_t n_devices);
int spd_modelstate_init_sst_anom(int64_t state_cnt, int32_t n_months);    /* sst_anom(ix, il, 0:n_months+1), zero-filled */
int spd_modelstate_close(int64_t state_cnt);

/* ---- extension: one process, several GPUs.  The reference's ensemble is ONE process that hands all its containers to
 *      parallel_step (speedy_driver.f90.j2:58-79, an OpenMP loop over members).  By default a container lives on the HIP
 *      device that is current in the calling thread.  spd_set_device_placement(k) (or PYSPEEDY_AMD_DEVICES=k|all in the
 *      environment) makes spd_modelstate_init spread containers over devices 0 .. k-1 round-robin in creation order;
 *      spd_modelstate_init_on names the device; spd_set_device_placement(0) goes back to the current device.
 *      spd_parallel_step over containers of several devices gathers and steps them per device and enqueues every device's
 *      step and range check before it waits for any, so the GPUs work side by side.  spd_broadcast_boundary copies the shared
 *      boundary fields (orog, fmask_orig, alb0, veg_high, veg_low, stl12, snowd12, soil_wc_l1..3, sst12, sea_ice_frac12 and,
 *      when the lengths agree, sst_anom) of container `root` into all the others, device to device: ONE RCCL broadcast over
 *      xGMI to the other GPUs, local copies on each of them. ---- */
int spd_device_count(int32_t *n_devices);
int spd_set_device_placement(int32_t n_devices);
int spd_modelstate_init_on(int64_t *state_cnt, int32_t device);
int spd_modelstate_device(int64_t state_cnt, int32_t *device);
int spd_broadcast_boundary(const int64_t *state_cnts, int32_t n, int32_t root);
/* what the last spd_broadcast_boundary did: how many other GPUs received the fields through the ONE collective broadcast (RCCL
 * over xGMI: the first container of the list on each of them; 0 when the transport was point-to-point), copies that crossed to
 * another device one by one (hipMemcpyPeerAsync: the transport without RCCL, PYSPEEDY_AMD_BROADCAST=peer; and single fields a
 * collective could not carry for everybody) and copies that stayed on a device (the other containers of a GPU take the fields from
 * its first one) */
int spd_broadcast_boundary_stats(int32_t *peer_copies, int32_t *local_copies, int32_t *collective_devices);
/* ... and in words (valid until this thread's next call): "one RCCL broadcast to 7 other device(s)", "peer copies
 * (PYSPEEDY_AMD_BROADCAST=peer)", "peer copies, because: <why the collective was not used>" -- RCCL not loadable, or its
 * initialisation did not come back within PYSPEEDY_AMD_RCCL_TIMEOUT seconds (default 30; spd_model_broadcast_vars bounds every
 * wait on RCCL) --, "local copies only (one device)", or "failed: ..." when the collective was enqueued and did not complete
 * inside its bound (spd_broadcast_boundary then returns SPD_E_TIMEOUT: nothing can be queued behind it) */
const char *spd_broadcast_boundary_note(void);

/* ---- Datetime interface (:163-210) ---- */
int spd_create_datetime(int32_t year, int32_t month, int32_t day, int32_t hour, int32_t minute, int64_t *datetime_cnt);
int spd_get_datetime(int64_t datetime_cnt, int32_t *year, int32_t *month, int32_t *day, int32_t *hour, int32_t *minute);
int spd_close_datetime(int64_t datetime_cnt);

/* ---- ControlParams interface (:131-158); the extra getter exposes model_datetime, which the reference's Python layer
 *      mirrors on its own (speedy.py:396-405) ---- */
int spd_controlparams_init(int64_t *control_cnt, int64_t start_datetime_cnt, int64_t end_datetime_cnt);
int spd_controlparams_close(int64_t control_cnt);
int spd_controlparams_get_model_datetime(int64_t control_cnt, int32_t *ymdhm /* 5 */, int32_t *month_idx);

/* ---- Speedy interface (:29-125) ---- */
int spd_init(int64_t state_cnt, int64_t control_cnt, int32_t *error_code);
/* extension: spd_init for every container of the list, with the containers that are all the members of a not yet initialised
 * device model (spd_modelstate_init_ensemble) and share their start date initialised together in one pass (same states, bit for
 * bit; 256 members in 20 ms instead of 2 s) */
int spd_init_ensemble(const int64_t *state_cnts, const int64_t *control_cnts, int32_t *error_codes, int32_t n_members);
int spd_step(int64_t state_cnt, int64_t control_cnt, int32_t *error_code);
int spd_parallel_step(const int64_t *state_cnts, const int64_t *control_cnts, int32_t *error_codes, int32_t n_members);
/* Extension: the same step with its range check overlapped.  _begin enqueues the step and its check for all containers and
 * returns a token; _end waits for that check only and hands out the codes.  A host loop that begins step k + 1 before ending
 * step k never leaves the GPU waiting for the host; the price is that the code of step k is seen after step k + 1 has been
 * enqueued.  The model dates in the control containers advance at _begin and are put back at _end for a member whose check
 * failed.  At most two steps may be in flight per device model.  The check of a step has no launch of its own in this form:
 * the first launch of the NEXT step of the same containers carries it (it looks at the state as its own step left it; whatever
 * else touches the state first -- spd_set, a regrouping, the _end that collects it -- sends it out there and then). */
int spd_parallel_step_begin(const int64_t *state_cnts, const int64_t *control_cnts, int32_t n_members, int64_t *token);
int spd_parallel_step_end(int64_t token, int32_t *error_codes /* n_members of the matching _begin */);
/* Extension: n_steps steps (1 ... 4096) of the same containers as ONE call, for the stretch of a time loop in which nothing looks at
 * the state (the reference's Speedy.run / SpeedyEns.run between two due callbacks, pyspeedy/speedy.py:396-405, 572-586).  Every
 * device model takes the steps as one multi-step device call -- member groups on streams of their own, large ensembles in rounds --
 * and the range check of EVERY step is recorded on the device.  _begin enqueues and returns a token, the model dates move n_steps
 * steps; _end waits and hands out, per member, the code of the first step whose check failed (0 when none did; the reference's loop
 * stops at the first code) and in steps_done (may be NULL) the steps it completed before that one (n_steps when none failed).  For a
 * member with -2 the reference's text goes to stderr with the step counter of the failing step, and its date is the one after its
 * last accepted step (speedy.f90:57-71 returns before advance_date).  The device does NOT stop at the failed check: the member's
 * state is what the remaining steps made of a state outside the accepted range -- as after every -2, the only defined
 * continuation is a new spd_init.  No single step (spd_parallel_step_begin) may be in flight on the same containers. */
int spd_parallel_steps_begin(const int64_t *state_cnts, const int64_t *control_cnts, int32_t n_members, int32_t n_steps, int64_t *token);
int spd_parallel_steps_end(int64_t token, int32_t *error_codes /* n_members */, int32_t *steps_done /* n_members or NULL */);
int spd_check(int64_t state_cnt, int32_t *error_code); /* diagnostics on time level 1 */
int spd_transform_spectral2grid(int64_t state_cnt);
int spd_transform_grid2spectral(int64_t state_cnt);
int spd_apply_grid_filter(int64_t state_cnt);

/* ---- registry access: get_<v> / set_<v> / get_<v>_shape / is_array_<v> (:250-334), driven by the variable's name instead
 *      of one generated procedure per variable.  Buffers are HOST memory in the reference's shape and (Fortran) order;
 *      element types as the f2py getters return them: complex(8) -> 2 doubles, real(8) -> double, lon / lat / lev ->
 *      float, integer and logical scalars -> int32_t.  `bytes` must be exactly the size of the variable. ---- */
#define SPD_T_FLOAT64 0
#define SPD_T_COMPLEX128 1
#define SPD_T_FLOAT32 2
#define SPD_T_INT32 3
#define SPD_T_LOGICAL 4 /* int32_t 0 / 1 */
int spd_get(int64_t state_cnt, const char *name, void *buf_host, size_t bytes);
int spd_set(int64_t state_cnt, const char *name, const void *buf_host, size_t bytes);
/* shape in the reference's order; ndim = 0 for scalars; all extents 0 while a run-length array (sst_anom) is unallocated */
int spd_get_shape(int64_t state_cnt, const char *name, int32_t *shape /* up to 5 */, int32_t *ndim);
int spd_is_array(const char *name, int32_t *is_array);
/* the registry itself, without a device: entry `index` (0 .. count-1); returns the number of entries */
int spd_registry_entry(int32_t index, char *name /* 32 bytes */, int32_t *dtype, int32_t *ndim, int32_t *shape /* 5 */,
                       int32_t *is_read_only);
/* the batched device model behind a container (spd_model_handle of pyspeedy_amd.h, owned by the driver: do not destroy it)
 * and the container's member index in it (an ensemble of 32 or more containers of one device lives in TWO device models, see
 * spd_modelstate_init_ensemble: ask per container) -- for zero-copy access to the state (spd_model_device_ptr) and the extensions of
 * the model level (SPPT, physics precision, profiling).  The binding changes when parallel_step gathers or splits. */
int spd_driver_model(int64_t state_cnt, void **model, int32_t *member, int32_t *members_in_model);
/* how many device models are alive and how many members the container's model holds (tests / diagnostics of batching) */
int spd_driver_stats(int64_t state_cnt, int32_t *models_alive, int32_t *members_in_model);
/* Host-side order in which parallel_step (begin / end) handled its device models when there were several: spd_driver_trace(1)
 * starts recording, _read copies up to `capacity` (kind, group) pairs and returns how many there are; kind 1 = step + check
 * of the group enqueued, 2 = the host starts waiting for the group, 3 = its codes are back. */
int spd_driver_trace(int32_t on);
int spd_driver_trace_read(int32_t *pairs, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* PYSPEEDY_AMD_DRIVER_H */
