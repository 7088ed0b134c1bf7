"""ctypes binding of the C ABI declared in include/pyspeedy_amd.h.

The HIP library is the product: if it is missing or cannot be loaded this module raises -- there is no
CPU fallback (the CPU oracle under oracle/ is test infrastructure and is never imported from here).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYSPEEDY_AMD_LIB: another build of the same library (A/B measurements of kernel variants in one session)
LIB_PATH = os.environ.get("PYSPEEDY_AMD_LIB") or os.path.join(_HERE, "libpyspeedy_amd.so")
CSRC = os.path.join(_HERE, "csrc")

IX, IL, IY, KX, MX, NX, TRUNC = 96, 48, 24, 8, 31, 32, 30
NSPEC, NFOUR, NGRID = MX * NX, 2 * MX * IL, IX * IL

SPD_OK, SPD_E_ARG, SPD_E_DEVICE, SPD_E_SIZE = 0, -1, -2, -3


class SpeedyHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into pyspeedy_amd/libpyspeedy_amd.so (hipcc cross-compiles)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.run(cmd, check=True)
    if not os.path.isfile(LIB_PATH):
        raise SpeedyHipError("build did not produce " + LIB_PATH)


_PHYS_PTR_FIELDS = [
    "ug", "vg", "tg", "qg", "phig", "pslg", "utend", "vtend", "ttend", "qtend",
    "fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc", "land_temp", "soil_avail_water",
    "flux_solar_in", "flux_ozone_upper", "flux_ozone_lower", "zenit_correction", "stratospheric_correction",
    "alb_surface",
    "precnv", "precls", "cbmf", "slrd", "slr", "olr",
    "slru", "ustr", "vstr", "shf", "evap", "hfluxn", "rad_st4a", "rad_flux",
    "tt_rsw", "rad_tau2", "rad_strat_corr", "tsr", "ssrd", "ssr", "qcloud_equiv",
    "iptop", "icltop", "ts", "tskin", "u0", "v0", "t0", "cloudc", "clstr",
]


class PhysicsArgs(C.Structure):
    """Mirror of spd_physics_args (include/pyspeedy_amd.h)."""
    _fields_ = [(n, C.c_void_p) for n in _PHYS_PTR_FIELDS] + [
        ("air_absortivity_co2", C.c_double), ("compute_shortwave", C.c_int32), ("fp32", C.c_int32),
        ("sppt_pattern", C.c_void_p)]


class ModelControl(C.Structure):
    """Mirror of spd_model_control (include/pyspeedy_amd.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "current_step", "year", "month", "day", "hour", "minute", "month_idx", "land_coupling_flag",
        "sst_anomaly_coupling_flag", "increase_co2", "sppt_on", "sppt_first", "physics_fp32", "reserved")] + [
        ("sppt_step", C.c_int64), ("sppt_first_member_id", C.c_int64), ("sppt_seed", C.c_uint64),
        ("air_absortivity_co2", C.c_double), ("ablco2_ref", C.c_double)]


class StreamProbeArgs(C.Structure):
    """Mirror of spd_stream_probe_args (include/pyspeedy_amd.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "reads", "writes", "lane_bytes", "in_flight", "nontemporal", "waves_per_simd", "rows_per_wave", "reps", "layout",
        "reserved")] + [
        ("total_bytes", C.c_uint64)]


_SIGNATURES = {
    "spd_stream_probe": (C.c_int, [C.c_void_p, C.POINTER(StreamProbeArgs), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "spd_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "spd_destroy": (C.c_int, [C.c_void_p]),
    "spd_device": (C.c_int, [C.c_void_p]),
    "spd_last_error": (C.c_char_p, []),
    "spd_version": (C.c_char_p, []),
    "spd_get_table_host": (C.c_long, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "spd_calendar_walk": (C.c_int, [C.c_int] * 6 + [C.c_void_p] * 5),
    "spd_daily_forcing_host": (C.c_int, [C.c_double, C.c_void_p]),
    "spd_spec2grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spd_grid2spec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "spd_legendre_inv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "spd_legendre": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "spd_fourier_inv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spd_fourier": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "spd_vort2vel": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]),
    "spd_vel2vort": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]),
    "spd_grid_vel2vort": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]),
    "spd_gradient": (C.c_int, [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]),
    "spd_laplacian": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spd_truncate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "spd_grid_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "spd_physics": (C.c_int, [C.c_void_p, C.POINTER(PhysicsArgs), C.c_int, C.c_void_p]),
    "spd_model_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "spd_model_destroy": (C.c_int, [C.c_void_p]),
    "spd_model_members": (C.c_int, [C.c_void_p]),
    "spd_model_memory": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "spd_model_var_bytes": (C.c_long, [C.c_void_p, C.c_char_p]),
    "spd_model_var_storage": (C.c_int, [C.c_void_p, C.c_char_p]),
    "spd_model_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]),
    "spd_model_get": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]),
    "spd_model_device_ptr": (C.c_void_p, [C.c_void_p, C.c_char_p]),
    "spd_model_invalidate": (C.c_int, [C.c_void_p]),
    "spd_model_checks_in_flight": (C.c_int, [C.c_void_p]),
    "spd_model_set_co2": (C.c_int, [C.c_void_p, C.c_double]),
    "spd_model_co2": (C.c_double, [C.c_void_p]),
    "spd_model_set_time_step": (C.c_int, [C.c_void_p, C.c_double]),
    "spd_model_step_dynamics": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "spd_model_check": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spd_model_check_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "spd_model_check_end": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "spd_model_check_defer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "spd_model_check_settle": (C.c_int, [C.c_void_p, C.c_int]),
    "spd_model_check_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_model_init": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "spd_model_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "spd_model_current_step": (C.c_int, [C.c_void_p]),
    "spd_model_get_date": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spd_model_mark_initialized": (C.c_int, [C.c_void_p] + [C.c_int] * 6),
    "spd_model_get_control": (C.c_int, [C.c_void_p, C.POINTER(ModelControl)]),
    "spd_model_set_control": (C.c_int, [C.c_void_p, C.POINTER(ModelControl)]),
    "spd_model_set_flags": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "spd_model_spectral2grid": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spd_model_grid2spectral": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spd_model_grid_filter": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spd_model_export_pack": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "spd_model_init_sst_anom": (C.c_int, [C.c_void_p, C.c_int]),
    "spd_model_set_sppt": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_int64]),
    "spd_model_copy_member": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    # outer boundary (include/pyspeedy_amd_driver.h): the procedures of speedy_driver.f90.j2
    "spd_modelstate_init": (C.c_int, [C.POINTER(C.c_int64)]),
    "spd_modelstate_init_ensemble": (C.c_int, [C.POINTER(C.c_int64), C.c_int32]),
    "spd_modelstate_init_sst_anom": (C.c_int, [C.c_int64, C.c_int32]),
    "spd_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "spd_set_device_placement": (C.c_int, [C.c_int32]),
    "spd_modelstate_init_on": (C.c_int, [C.POINTER(C.c_int64), C.c_int32]),
    "spd_modelstate_device": (C.c_int, [C.c_int64, C.POINTER(C.c_int32)]),
    "spd_broadcast_boundary": (C.c_int, [C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "spd_driver_trace": (C.c_int, [C.c_int32]),
    "spd_driver_trace_read": (C.c_int, [C.POINTER(C.c_int32), C.c_int32]),
    "spd_model_copy_vars": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_void_p]),
    "spd_model_copy_vars_enqueue": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_void_p]),
    "spd_model_broadcast_vars": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int]),
    "spd_modelstate_init_ensemble_on": (C.c_int, [C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "spd_broadcast_boundary_stats": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_broadcast_boundary_note": (C.c_char_p, []),
    "spd_modelstate_close": (C.c_int, [C.c_int64]),
    "spd_create_datetime": (C.c_int, [C.c_int32] * 5 + [C.POINTER(C.c_int64)]),
    "spd_get_datetime": (C.c_int, [C.c_int64] + [C.POINTER(C.c_int32)] * 5),
    "spd_close_datetime": (C.c_int, [C.c_int64]),
    "spd_controlparams_init": (C.c_int, [C.POINTER(C.c_int64), C.c_int64, C.c_int64]),
    "spd_controlparams_close": (C.c_int, [C.c_int64]),
    "spd_controlparams_get_model_datetime": (C.c_int, [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_init": (C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_int32)]),
    "spd_init_ensemble": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32]),
    "spd_step": (C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_int32)]),
    "spd_parallel_step": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32]),
    "spd_parallel_step_begin": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int64)]),
    "spd_parallel_step_end": (C.c_int, [C.c_int64, C.POINTER(C.c_int32)]),
    "spd_modelstate_init_ensemble_whole": (C.c_int, [C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "spd_parallel_steps_begin": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "spd_parallel_steps_end": (C.c_int, [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_model_step_checked_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "spd_model_step_checked_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_driver_model": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_check": (C.c_int, [C.c_int64, C.POINTER(C.c_int32)]),
    "spd_transform_spectral2grid": (C.c_int, [C.c_int64]),
    "spd_transform_grid2spectral": (C.c_int, [C.c_int64]),
    "spd_apply_grid_filter": (C.c_int, [C.c_int64]),
    "spd_get": (C.c_int, [C.c_int64, C.c_char_p, C.c_void_p, C.c_size_t]),
    "spd_set": (C.c_int, [C.c_int64, C.c_char_p, C.c_void_p, C.c_size_t]),
    "spd_get_shape": (C.c_int, [C.c_int64, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_is_array": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "spd_registry_entry": (C.c_int, [C.c_int32, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32)]),
    "spd_driver_stats": (C.c_int, [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_model_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "spd_model_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "spd_model_profile_read_kernels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spd_model_set_physics_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "spd_model_get_config": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "spd_model_group_streams": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spd_model_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "spd_model_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load the HIP library (once).  Raises SpeedyHipError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise SpeedyHipError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C pyspeedy_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
        # PyTorch ships its own HIP / ROCr runtime libraries.  They must be in the process BEFORE this library is loaded, so that
        # its libamdhip64 dependency binds to the copy torch initialises: with the opposite order the process holds two ROCr
        # runtimes, only the first of which can open the GPU ("no ROCm-capable device is detected" in the other).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as exc:
            raise SpeedyHipError("cannot load %s: %s" % (LIB_PATH, exc))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != SPD_OK:
        msg = lib().spd_last_error()
        raise SpeedyHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
