"""Ensemble model object: the device-resident state of M members and the model time step (`step` of
speedy.f90/time_stepping.f90:38-147 for all members at once).

Host arrays use the reference's shapes and Fortran order (what the f2py getters of speedy_driver.f90.j2:250-334
return); on the device every variable is member-major with that same order inside a member, so get/set are plain copies.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .registry import REGISTRY

# reference shapes of the device-resident registry variables (registry.py); sst_anom follows the model's allocation
SHAPES = {n: (v.dtype, v.shape) for n, v in REGISTRY.items() if v.where == "device"}
SHAPES["sst_anom"] = (np.float64, (96, 48, 3))  # right after creation: n_months = 1

DELT = 86400.0 / 36  # params.f90:33

# kernel ids of spd_model_profile_read_kernels (SPD_K_* of include/pyspeedy_amd.h)
KERNEL_NAMES = ("geopotential", "spec2grid", "column_sw", "column", "grid2spec", "spectral_step", "coupler", "forcing",
                "sppt", "dyn_grid", "physics_sw", "physics")

# boundary-condition file variable -> registry variable (pyspeedy/speedy.py:279-296)
BC_MAP = (("orog", "orog"), ("fmask_orig", "lsm"), ("alb0", "alb"), ("veg_high", "vegh"), ("veg_low", "vegl"),
          ("stl12", "stl"), ("snowd12", "snowd"), ("soil_wc_l1", "swl1"), ("soil_wc_l2", "swl2"), ("soil_wc_l3", "swl3"),
          ("sst12", "sst"), ("sea_ice_frac12", "icec"))


class EnsembleModel:
    def __init__(self, spectral, nmembers):
        self.sp = spectral
        self.nmembers = int(nmembers)
        self._lib = _lib.lib()
        self._m = C.c_void_p()
        self.n_months = 1  # sst_anom holds n_months + 2 planes
        self._shapes = dict(SHAPES)  # per model: set_sppt adds the two SPPT arrays
        self._owned = True
        with torch.cuda.device(spectral.device):
            check(self._lib.spd_model_create(spectral.handle, self.nmembers, C.byref(self._m)), "spd_model_create")

    @classmethod
    def borrowed(cls, handle, nmembers, device, n_months=1):
        """A view of a device model that something else owns (the C driver's containers, speedy_driver.device_model): the
        same methods, nothing is created and nothing destroyed."""
        import types
        self = cls.__new__(cls)
        self.sp = types.SimpleNamespace(device=device)
        self.nmembers, self.n_months = int(nmembers), int(n_months)
        self._lib, self._m, self._owned = _lib.lib(), handle, False
        self._shapes = dict(SHAPES)
        return self

    def close(self):
        if getattr(self, "_m", None) is not None and self._m:
            if self._owned:
                self._lib.spd_model_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- registry access (speedy_driver.f90.j2:250-334) --------------------------------------------------
    def variables(self):
        """Names of the registry arrays of this model (the reference's + tcorh / qcorh [+ the SPPT arrays when SPPT is on])."""
        return tuple(self._shapes)

    def shape(self, name):
        dtype, shape = self._shapes[name]
        return (dtype, (96, 48, self.n_months + 2)) if name == "sst_anom" else (dtype, shape)

    def set(self, name, value, member=-1):
        """Copy a host array (reference shape) into one member, or into every member when member == -1."""
        dtype, shape = self.shape(name)
        a = np.asarray(value, dtype=dtype)
        if a.shape != shape:
            raise ValueError("Array shape missmatch: %s expects %s, got %s" % (name, shape, a.shape))  # speedy.py:153
        flat = np.ascontiguousarray(a.ravel(order="F"))
        check(self._lib.spd_model_set(self._m, name.encode(), int(member), flat.ctypes.data_as(C.c_void_p), flat.nbytes),
              "spd_model_set(%s)" % name)

    def get(self, name, member=0):
        dtype, shape = self.shape(name)
        flat = np.empty(int(np.prod(shape)), dtype=dtype)
        check(self._lib.spd_model_get(self._m, name.encode(), int(member), flat.ctypes.data_as(C.c_void_p), flat.nbytes),
              "spd_model_get(%s)" % name)
        return flat.reshape(shape, order="F")

    # work arrays of the step that are registered with the C model without being registry variables of the reference: the
    # grid-point inputs of the column physics (physics.f90:89-101) as the last step left them
    WORK_ARRAYS = {"u_grid_phys": (96, 48, 8), "v_grid_phys": (96, 48, 8), "t_grid_phys": (96, 48, 8), "q_grid_phys": (96, 48, 8),
                   "phi_grid_phys": (96, 48, 8), "pslg_phys": (96, 48)}

    def device_view(self, name):
        """Zero-copy torch view [nmembers, *reversed reference shape] of a registry variable in HBM (C order == the
        reference's Fortran order inside a member).  For on-device post-processing such as ensemble statistics.

        The view stays valid for the life of the model (spd_model_device_ptr, include/pyspeedy_amd.h): after every later
        step it shows the variable as that step left it.  Taking the view drops what the model had derived from the state;
        WRITING through a view taken earlier must be followed by `invalidate()` before the next step.  A view of "phi"
        pins the geopotential to one buffer (small ensembles lose the 1-3 % of the look-ahead geopotential)."""
        dtype, shape = (np.float64, self.WORK_ARRAYS[name]) if name in self.WORK_ARRAYS else self.shape(name)
        ptr = self._lib.spd_model_device_ptr(self._m, name.encode())
        if not ptr:
            raise KeyError(name)
        if self._lib.spd_model_var_storage(self._m, name.encode()) == 4:
            # cfg 5 (set_physics_precision(True)): what only the column physics reads back is STORED as float32; the view shows the
            # array as it is in memory (a view taken before the precision was switched shows bytes that no longer mean anything)
            dtype = np.float32

        class _Blob:  # CUDA array interface v2 (understood by torch.as_tensor on ROCm builds as well)
            __cuda_array_interface__ = {"shape": (self.nmembers,) + tuple(reversed(shape)),
                                        "typestr": np.dtype(dtype).str, "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_Blob(), device=self.sp.device)

    def invalidate(self):
        """The state was written through a device view taken earlier: drop what the model derived from it (look-ahead
        geopotential, the day's interpolated climatologies)."""
        check(self._lib.spd_model_invalidate(self._m), "spd_model_invalidate")

    def set_sppt(self, on=True, seed=0, first_member_id=0):
        """Switch the deterministic SPPT scheme (csrc/sppt.hip) on or off; `first_member_id` = global id of member 0 of
        this shard, so that an ensemble gives the same noise however it is split over GPUs."""
        check(self._lib.spd_model_set_sppt(self._m, int(bool(on)), int(seed), int(first_member_id)), "spd_model_set_sppt")
        if on:
            self._shapes.setdefault("sppt_spec", (np.complex128, (31, 32, 8)))
            self._shapes.setdefault("sppt_pattern", (np.float64, (96, 48, 8)))

    @property
    def co2(self):
        return float(self._lib.spd_model_co2(self._m))

    def set_co2(self, value):
        check(self._lib.spd_model_set_co2(self._m, float(value)), "spd_model_set_co2")

    # ---- lifecycle (pyspeedy/speedy.py:217-301, 375-405) ----------------------------------------------------
    def set_bc(self, bc, start_date=(1982, 1, 1, 0, 0), member=-1):
        """Load the 12 boundary fields (mapping like example_bc.nc, dims (lon, lat[, month])) and run the reference's
        `init`: land/sea preprocessing, rest atmosphere, coupler, forcing, first_step.  Zero SST anomaly unless
        `sst_anom` was set before."""
        for state_name, bc_name in BC_MAP:
            self.set(state_name, np.asarray(bc[bc_name], dtype=np.float64), member)
        self.init(start_date)

    def init(self, start_date=(1982, 1, 1, 0, 0)):
        """The reference's `init` (initialize_state) for every member, from the boundary fields stored with set()."""
        check(self._lib.spd_model_init(self._m, *[int(v) for v in start_date], self._stream()), "spd_model_init")

    def init_sst_anom(self, n_months):
        """modelstate_init_sst_anom: (re)allocate sst_anom with n_months + 2 zero-filled planes per member."""
        check(self._lib.spd_model_init_sst_anom(self._m, int(n_months)), "spd_model_init_sst_anom")
        self.n_months = int(n_months)

    def mark_initialized(self, current_step, date):
        """Declare a state loaded through set() / copy_member_from() as initialised at `date` = (y, m, d, h, mi)."""
        check(self._lib.spd_model_mark_initialized(self._m, int(current_step), *[int(v) for v in date]),
              "spd_model_mark_initialized")
        self.set_time_step(2 * DELT)

    # ---- checkpoint / resume: the registry arrays plus the host-side control block are the whole state of a run -------
    def control(self):
        """spd_model_control: step counter, date, month index, coupling / CO2 flags, CO2 reference, SPPT generator position."""
        c = _lib.ModelControl()
        check(self._lib.spd_model_get_control(self._m, C.byref(c)), "spd_model_get_control")
        return c

    def state_dict(self, member=0):
        """Everything needed to continue a member's run elsewhere, bit for bit (numpy arrays in the reference's shapes plus
        the control block as `__control__/<field>` scalars)."""
        out = {n: self.get(n, member) for n in self._shapes if n != "sppt_pattern"}  # (the pattern is recomputed every step)
        c = self.control()
        for name, _ in c._fields_:
            out["__control__/" + name] = np.asarray(getattr(c, name))
        return out

    def load_state_dict(self, state, member=-1):
        """Inverse of state_dict (member = -1: every member gets the same state); marks the model initialised and resets
        nothing: month index, CO2 reference, flags and the SPPT position continue where the checkpoint left them."""
        c = _lib.ModelControl()
        for name, _ in c._fields_:
            setattr(c, name, state["__control__/" + name].item())
        if state["sst_anom"].shape[2] != self.n_months + 2:
            self.init_sst_anom(state["sst_anom"].shape[2] - 2)
        if c.sppt_on:
            self.set_sppt(True, c.sppt_seed, c.sppt_first_member_id)
        for n in self._shapes:
            if n in state:
                self.set(n, state[n], member)
        check(self._lib.spd_model_set_control(self._m, C.byref(c)), "spd_model_set_control")
        self.set_time_step(2 * DELT)

    def copy_member_from(self, src, src_member, dst_member):
        """Device-to-device copy of every registry variable of one member of `src` into one of this model's members."""
        check(self._lib.spd_model_copy_member(self._m, int(dst_member), src._m, int(src_member), self._stream()),
              "spd_model_copy_member")

    def sync(self):
        torch.cuda.current_stream().synchronize()

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- grid-space views of the prognostic variables (prognostics.f90:125-219) -------------------------
    def _range(self, first, count):
        return int(first), int(self.nmembers - first if count is None else count)

    def spectral2grid(self, first=0, count=None):
        check(self._lib.spd_model_spectral2grid(self._m, *self._range(first, count), self._stream()), "spd_model_spectral2grid")

    def grid2spectral(self, first=0, count=None):
        check(self._lib.spd_model_grid2spectral(self._m, *self._range(first, count), self._stream()), "spd_model_grid2spectral")

    def grid_filter(self, first=0, count=None):
        check(self._lib.spd_model_grid_filter(self._m, *self._range(first, count), self._stream()), "spd_model_grid_filter")

    def run(self, nsteps):
        """`nsteps` model steps (40 simulated minutes each) for every member; asynchronous."""
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_model_step(self._m, int(nsteps), stream), "spd_model_step")

    def run_checked(self, nsteps):
        """`nsteps` steps as ONE device call with the range check of EVERY step recorded by the device
        (spd_model_step_checked_begin / _end); waits for it.  -> (first_failed, accepted): per member the first step of the call
        (0-based) whose check failed, -1 for none, and [members, 7] the model's step counter, date (y, m, d, h, min) and month index
        after the member's last accepted step."""
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_model_step_checked_begin(self._m, int(nsteps), stream), "spd_model_step_checked_begin")
        failed = np.zeros(self.nmembers, dtype=np.int32)
        accepted = np.zeros((self.nmembers, 7), dtype=np.int32)
        check(self._lib.spd_model_step_checked_end(self._m, failed.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   accepted.ctypes.data_as(C.POINTER(C.c_int32))), "spd_model_step_checked_end")
        return failed, accepted

    @property
    def current_step(self):
        return int(self._lib.spd_model_current_step(self._m))

    @property
    def current_date(self):
        buf = (C.c_int * 5)()
        check(self._lib.spd_model_get_date(self._m, buf), "spd_model_get_date")
        return tuple(buf)

    def set_physics_precision(self, fp32):
        """BASELINE cfg 5: run the arithmetic of the column physics in single precision AND keep what only the column physics
        reads back -- its grid-point inputs at the physics' time level, the radiation state persisted between shortwave steps
        (tt_rsw, rad_tau2, rad_strat_corr), the diagnostics-only outputs (rad_st4a, rad_flux, precnv, precls, cbmf, slrd, slr,
        olr, slru, ustr, vstr) -- in memory as float32.  State, dynamics and tendencies stay fp64; get / set of the narrowed
        variables still speak float64 (converted on the way); device_view shows them as float32.  Switching converts the
        arrays in place."""
        check(self._lib.spd_model_set_physics_precision(self._m, int(bool(fp32))), "spd_model_set_physics_precision")

    def memory(self):
        """Device memory of the model in bytes: (reserved, in use) -- spd_model_memory."""
        reserved, used = C.c_size_t(0), C.c_size_t(0)
        check(self._lib.spd_model_memory(self._m, C.byref(reserved), C.byref(used)), "spd_model_memory")
        return reserved.value, used.value

    def config(self):
        """How the step is configured: dict(inv_per_member, diag_every_step, chunks, split_dyn) -- spd_model_get_config."""
        cfg = (C.c_int32 * 8)()
        check(self._lib.spd_model_get_config(self._m, cfg), "spd_model_get_config")
        created, apart = C.c_int32(0), C.c_int32(1)
        check(self._lib.spd_model_group_streams(self._m, C.byref(created), C.byref(apart)), "spd_model_group_streams")
        block = C.c_int32(0)
        check(self._lib.spd_model_get_option(self._m, b"block_members", C.byref(block)), "spd_model_get_option")
        streams, M = cfg[2], self.nmembers
        # (as spd_model_step forms them: rounds of `chunks` x `block_members` members from 4 x block_members members up; not while
        # profiling or with separate dynamics / physics launches, which step everybody as one group)
        rounds = 1
        if block.value > 0 and M >= 4 * block.value and not cfg[3]:
            rounds = (M + max(streams, 1) * block.value - 1) // (max(streams, 1) * block.value)
        return dict(inv_per_member=cfg[0], diag_every_step=bool(cfg[1]), chunks=cfg[2], split_dyn=bool(cfg[3]),
                    fold_geo=bool(cfg[4]), coupler_in_spectral=bool(cfg[5]), physics_fp32=bool(cfg[6]),
                    physics_storage32=bool(cfg[7]), group_streams=created.value, group_streams_apart=bool(apart.value),
                    block_members=block.value, rounds=rounds)

    def set_option(self, name, value):
        """A launch-plan switch of the live model by name (spd_model_set_option: diag_every_step, coupler_in_spectral,
        spectral_early, split_dyn, member_groups, block_members, physics_storage32); none of them changes the state a step leaves behind.
        ValueError for an unknown name."""
        rc = self._lib.spd_model_set_option(self._m, name.encode(), int(value))
        if rc == _lib.SPD_E_ARG:
            raise ValueError("unknown option or value out of range: %s = %r" % (name, value))
        check(rc, "spd_model_set_option")

    def profile(self, level=1):
        """HIP-event brackets on the launch stream: 0 off, 1 the spectral->grid launch of every step, 2 every kernel."""
        check(self._lib.spd_model_profile(self._m, int(level)), "spd_model_profile")

    def profile_read_kernels(self):
        """{kernel id (KERNEL_NAMES): (mean ms, min ms, brackets, units per bracket)} since profile(2)."""
        n = len(KERNEL_NAMES)
        mean, mn = np.zeros(n), np.zeros(n)
        cnt, units = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        check(self._lib.spd_model_profile_read_kernels(self._m, mean.ctypes.data_as(C.c_void_p), mn.ctypes.data_as(C.c_void_p),
                                                       cnt.ctypes.data_as(C.c_void_p), units.ctypes.data_as(C.c_void_p)),
              "spd_model_profile_read_kernels")
        return {KERNEL_NAMES[k]: (float(mean[k]), float(mn[k]), int(cnt[k]), int(units[k])) for k in range(n) if cnt[k]}

    def profile_read(self):
        """(mean launch time in ms, number of launches, fields per launch) of the spectral->grid kernel since profile(True)."""
        ms, n, f = C.c_double(), C.c_int(), C.c_int()
        check(self._lib.spd_model_profile_read(self._m, C.byref(ms), C.byref(n), C.byref(f)), "spd_model_profile_read")
        return ms.value, n.value, f.value

    def set_flags(self, land_coupling_flag=True, sst_anomaly_coupling_flag=True, increase_co2=False):
        check(self._lib.spd_model_set_flags(self._m, int(land_coupling_flag), int(sst_anomaly_coupling_flag),
                                            int(increase_co2)), "spd_model_set_flags")

    # ---- time stepping -------------------------------------------------------------------------------------
    def set_time_step(self, dt):
        check(self._lib.spd_model_set_time_step(self._m, float(dt)), "spd_model_set_time_step")

    def step_dynamics(self, j1, j2, dt, compute_shortwave):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_model_step_dynamics(self._m, int(j1), int(j2), float(dt), int(bool(compute_shortwave)), stream),
              "spd_model_step_dynamics")

    def check_begin(self, time_level=2):
        """Enqueue the range check of the current state; returns a token for check_end (at most two may be in flight)."""
        slot = self._lib.spd_model_check_begin(self._m, int(time_level), self._stream())
        if slot < 0:
            check(slot, "spd_model_check_begin")
        return slot

    def check_defer(self, time_level=2):
        """check_begin without a launch of its own: the check rides in the next single-step run() on this stream, or goes out when
        anything else would touch the state first (spd_model_check_defer).  Returns the token for check_end."""
        slot = self._lib.spd_model_check_defer(self._m, int(time_level), self._stream())
        if slot < 0:
            check(slot, "spd_model_check_defer")
        return slot

    def check_counts(self):
        """(range checks launched on their own, range checks carried by a step's launch)"""
        alone, rode = C.c_int32(0), C.c_int32(0)
        check(self._lib.spd_model_check_counts(self._m, C.byref(alone), C.byref(rode)), "spd_model_check_counts")
        return alone.value, rode.value

    def check_end(self, token):
        codes = np.zeros(self.nmembers, dtype=np.int32)
        check(self._lib.spd_model_check_end(self._m, int(token), codes.ctypes.data_as(C.c_void_p)), "spd_model_check_end")
        return codes

    def check(self, time_level=2, with_diag=False):
        """diagnostics.f90 range check; returns int32 codes per member (0 ok, -2 out of range) [and the diagnostics]."""
        codes = np.zeros(self.nmembers, dtype=np.int32)
        diag = np.zeros((self.nmembers, 3, 8)) if with_diag else None
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.spd_model_check(self._m, int(time_level), codes.ctypes.data_as(C.c_void_p),
                                        diag.ctypes.data_as(C.c_void_p) if with_diag else None, stream), "spd_model_check")
        return (codes, diag) if with_diag else codes
