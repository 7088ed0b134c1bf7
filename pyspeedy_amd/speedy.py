"""`Speedy` and `SpeedyEns`: the user-facing model objects of pySPEEDY (pyspeedy/speedy.py:41-597) over the MI355X backend.

Same constructor arguments, properties and methods as the reference classes -- set_params, set_bc, run(callbacks),
state access with model["var"], get_shape, spectral2grid / grid2spectral, check, to_dataframe, get_current_step -- written
against `speedy_driver` (the `_speedy` function set) exactly as the reference is.  Differences, all on the host side:

* to_dataframe() returns a `pyspeedy_amd.dataset.Dataset` (xarray is not a dependency); it carries the same variables,
  dimension order (time, [ens,] lev, lat, lon; lev top-down reversed), float32 dtype and attributes, and writes NetCDF-3.
* Boundary conditions: the packaged example_bc (converted from the reference's example_bc.nc), an .npz / NetCDF-3 file, the
  reference's own NetCDF-4 / HDF5 files when netCDF4, h5py or xarray is importable (none is required; the error names the
  converter otherwise), an xarray.Dataset, or a mapping of arrays.  The reference's default SST-anomaly file is not distributed with it (.MISSING_LARGE_BLOBS),
  so `sst_anomaly=None` means zero anomalies; a mapping / file with `ssta` (lon, lat, time) and `time` works as upstream.
* SpeedyEns members are slots of ONE batched device model (speedy_driver.modelstate_init_ensemble): `SpeedyEns.run`
  advances all members with one set of kernel launches per step.  A member of an ensemble cannot `run()` on its own.
"""
import os
from datetime import datetime, timedelta

import numpy as np

from . import speedy_driver as _speedy
from .dataset import Dataset, Variable, open_dataset
from .registry import DEFAULT_OUTPUT_VARS, REGISTRY, model_state_def
from .speedy_driver import ERROR_CODES

PACKAGE_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# boundary-condition file variable -> state variable (pyspeedy/speedy.py:279-296)
_BC_FIELDS = (("orog", "orog"), ("fmask_orig", "lsm"), ("alb0", "alb"), ("veg_high", "vegh"), ("veg_low", "vegl"),
              ("stl12", "stl"), ("snowd12", "snowd"), ("soil_wc_l1", "swl1"), ("soil_wc_l2", "swl2"), ("soil_wc_l3", "swl3"),
              ("sst12", "sst"), ("sea_ice_frac12", "icec"))

_DT_STEP = timedelta(seconds=3600 * 24 / 36)
MODEL_STATE_DEF = model_state_def()  # pyspeedy/speedy.py:36-37 loads the same table from model_state.json


def example_bc_file():
    """Path of the packaged example boundary conditions (ERA-interim 1979-2008 climatology of the original SPEEDY)."""
    return os.path.join(PACKAGE_DATA_DIR, "example_bc.npz")


def example_sst_anomaly_file():
    """Path where the reference keeps its example SST anomalies (pyspeedy/__init__.py:33).  The file is not distributed with the
    reference (its repository lists it under .MISSING_LARGE_BLOBS) and is not packaged here either: `set_bc(sst_anomaly=None)`
    means zero anomalies; pass this path (or any file / mapping with `ssta` and `time`) once you have the data."""
    return os.path.join(PACKAGE_DATA_DIR, "sst_anomaly.nc")


def _add_months(date, months):
    y, m = divmod(date.year * 12 + date.month - 1 + months, 12)
    return date.replace(year=y, month=m + 1)


_last_file = [None, None]  # (path, mtime, size) and the fields of the file read last
_CACHE_LIMIT_BYTES = 32 << 20  # a boundary-condition file is 7 MB as float64; a multi-decade SST-anomaly record is not kept


_HDF5_SIGNATURE = b"\x89HDF\r\n\x1a\n"


def _read_hdf5(path):
    """name -> array of every variable of a NetCDF-4 / HDF5 file -- the format of the reference's own boundary files
    (pyspeedy/data/example_bc.nc, read there with xr.load_dataset(bc_file, engine="netcdf4"), speedy.py:277) -- through whichever
    of netCDF4, h5py or xarray can be imported; none of them is a dependency of this package.  `time` comes back as datetime64."""
    reasons = []
    try:
        import netCDF4
        with netCDF4.Dataset(str(path), "r") as f:
            out = {}
            for name, v in f.variables.items():
                values = np.asarray(v[...])
                if name == "time" and hasattr(v, "units"):
                    dates = netCDF4.num2date(values, v.units, getattr(v, "calendar", "standard"), only_use_cftime_datetimes=False)
                    values = np.asarray([np.datetime64(d) for d in np.atleast_1d(dates)], dtype="datetime64[s]")
                out[name] = values
            return out
    except ImportError as exc:
        reasons.append("netCDF4: %s" % exc)
    try:
        import h5py
        from .dataset import _decode_time
        with h5py.File(str(path), "r") as f:
            out = {}
            for name, v in f.items():
                if not isinstance(v, h5py.Dataset):
                    continue
                values = np.asarray(v[...])
                units = v.attrs.get("units")
                units = units.decode() if isinstance(units, bytes) else units
                if name == "time" and isinstance(units, str) and " since " in units:
                    values = _decode_time(values, units)
                out[name] = values
            return out
    except ImportError as exc:
        reasons.append("h5py: %s" % exc)
    try:
        import xarray
        ds = xarray.load_dataset(str(path))
        return {str(name): np.asarray(v.values) for name, v in ds.variables.items()}
    except ImportError as exc:
        reasons.append("xarray: %s" % exc)
    raise RuntimeError(
        "%s is a NetCDF-4 / HDF5 file, and none of netCDF4, h5py and xarray can be imported here (%s).  Install one of them, or "
        "convert the file once with `python tools/convert_bc.py %s` under an interpreter that has one: pyspeedy_amd itself reads .npz "
        "and NetCDF-3 classic files." % (path, "; ".join(reasons), path))


def _is_hdf5(path):
    with open(path, "rb") as fh:
        return fh.read(8) == _HDF5_SIGNATURE


def _dataset_like(source):
    """an xarray.Dataset (or anything shaped like one: a mapping of names to objects with `.values`), without importing xarray"""
    variables = getattr(source, "variables", None)
    return variables is not None and hasattr(variables, "items") and not isinstance(source, Dataset)


def _load_fields(source):
    """Mapping name -> array from an .npz file, a NetCDF-3 or NetCDF-4 / HDF5 file, a Dataset (ours or xarray's) or a plain mapping.  A FILE's arrays come back as
    float64 in Fortran order and READ-ONLY (this module only copies them into state containers).  The file read last is kept
    while it is small (a boundary-condition file: the reference's `for member in ens: member.set_bc()` reads the same file once
    per member, and decompressing it costs more than everything else a member's set_bc does); anything larger than
    _CACHE_LIMIT_BYTES -- an SST-anomaly record -- is read again when it is asked for again and pins no host memory."""
    if isinstance(source, (str, os.PathLike)):
        if not os.path.isfile(source):
            raise RuntimeError("The boundary conditions file does not exist.\nFile: %s" % source)
        st = os.stat(source)
        key = (os.path.realpath(source), st.st_mtime_ns, st.st_size)
        if _last_file[0] == key:
            return dict(_last_file[1])
        if str(source).endswith(".npz"):
            with np.load(source) as z:
                fields = {k: z[k] for k in z.files}
        elif _is_hdf5(source):  # the reference's own files (speedy.py:277, 321-331)
            fields = _read_hdf5(source)
        else:
            fields = {k: np.asarray(v.values) for k, v in open_dataset(source).variables.items()}
        # (as the state containers take them: float64, Fortran order -- converting costs as much as a member's 12 transfers)
        fields = {k: np.asfortranarray(v, dtype=np.float64) if v.dtype.kind == "f" else v for k, v in fields.items()}
        for v in fields.values():
            v.setflags(write=False)
        if sum(v.nbytes for v in fields.values()) <= _CACHE_LIMIT_BYTES:
            _last_file[0], _last_file[1] = key, fields
        elif _last_file[0] is not None and _last_file[0][0] == key[0]:
            _last_file[0], _last_file[1] = None, None
        return dict(fields)
    if isinstance(source, Dataset):
        return {k: v.values for k, v in source.variables.items()}
    if _dataset_like(source):  # an xarray.Dataset, as the reference's set_bc accepts for the SST anomalies (speedy.py:321-331)
        return {str(k): np.asarray(v.values) for k, v in source.variables.items()}
    return dict(source)


class Speedy:
    """One SPEEDY model instance."""

    def __init__(self, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 2), member=None, _state_cnt=None):
        self._start_date = None
        self._end_date = None
        self._model_date = None
        self._control_cnt = None
        self.member_id = member
        self.is_ensemble_member = self.member_id is not None
        self._state_cnt = _speedy.modelstate_init() if _state_cnt is None else _state_cnt
        self.set_params(start_date=start_date, end_date=end_date)
        self._initialized_bc = False
        self._initialized_ssta = False
        self.current_date = self.start_date

    def __del__(self):
        try:
            _speedy.modelstate_close(self._state_cnt)
            if self._control_cnt is not None:
                _speedy.controlparams_close(self._control_cnt)
            for cnt in (self._start_date, self._end_date, self._model_date):
                self._dealloc_date(cnt)
        except Exception:
            pass

    def set_params(self, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 2)):
        """Set the control parameters (start and end date of the simulation)."""
        self.start_date = start_date
        self.end_date = end_date
        if self.start_date > self.end_date:
            raise ValueError("The start date should be lower than the en date.")
        if self._control_cnt is not None:
            _speedy.controlparams_close(self._control_cnt)
        self._control_cnt = _speedy.controlparams_init(self._start_date, self._end_date)
        self.current_date = start_date
        self.n_months = ((self.end_date.year - self.start_date.year) * 12 + (self.end_date.month - self.start_date.month) + 1)

    # ---- dates ---------------------------------------------------------------------------------------------
    @staticmethod
    def _dealloc_date(container):
        if container is not None:
            _speedy.close_datetime(container)

    @staticmethod
    def _get_date(container):
        return datetime(*_speedy.get_datetime(container))

    def _date(self, which):
        """the date held by the container `which` (as the reference: a driver-side datetime object); the value is remembered on the
        Python side, so that a time loop that compares dates every step does not ask the driver for it every time"""
        cached = self.__dict__.get("_value" + which)
        return cached if cached is not None else self._get_date(getattr(self, which))

    def _assign_date(self, which, value):
        setattr(self, which, self._set_date(getattr(self, which), value))
        # (what the container holds, as the reference's getter returns it: year ... minute, no seconds, no time zone)
        self.__dict__["_value" + which] = datetime(value.year, value.month, value.day, value.hour, value.minute)

    @staticmethod
    def _set_date(container, value):
        Speedy._dealloc_date(container)
        if not isinstance(value, datetime):
            raise TypeError("The input value is not a datetime object.")
        return _speedy.create_datetime(value.year, value.month, value.day, value.hour, value.minute)

    start_date = property(lambda self: self._date("_start_date"), lambda self, v: self._assign_date("_start_date", v))
    end_date = property(lambda self: self._date("_end_date"), lambda self, v: self._assign_date("_end_date", v))
    current_date = property(lambda self: self._date("_model_date"), lambda self, v: self._assign_date("_model_date", v))

    # ---- state access --------------------------------------------------------------------------------------
    def __getitem__(self, var_name):
        getter = getattr(_speedy, "get_%s" % var_name, None)
        if getter is None:
            raise AttributeError("The state variable '%s' does not exist." % var_name)
        if var_name == "sst_anom":
            return getter(self._state_cnt, self.n_months)
        return getter(self._state_cnt)

    def get_shape(self, var_name):
        getter = getattr(_speedy, "get_%s_shape" % var_name, None)
        if getter is None:
            raise AttributeError("The 'get-shape' method for the state variable '%s' does not exist." % var_name)
        return tuple(getter(self._state_cnt))

    def __setitem__(self, var_name, value):
        setter = getattr(_speedy, "set_%s" % var_name, None)
        if setter is None:
            raise AttributeError("The setter for the state variable '%s' does not exist." % var_name)
        if getattr(_speedy, "is_array_%s" % var_name)():
            value = np.asarray(value)
            if self.get_shape(var_name) != value.shape:
                raise ValueError("Array shape missmatch")
            if var_name == "sst_anom":
                return setter(self._state_cnt, value, self.n_months)
        return setter(self._state_cnt, value)

    def get_current_step(self):
        inside_run = self.__dict__.get("_step_in_run")  # (the time loop counts along: hooks ask for the step several times a step)
        return self["current_step"] if inside_run is None else inside_run

    # ---- boundary conditions and initialisation ----------------------------------------------------------
    def set_bc(self, bc_file=None, sst_anomaly=None):
        """Load the boundary conditions and initialise the model (the reference's set_bc, speedy.py:217-301).

        bc_file: None (packaged example), a path (.npz or NetCDF-3) or a mapping with `orog, lsm, alb, vegh, vegl`
        (lon, lat) and `stl, snowd, swl1, swl2, swl3, sst, icec` (lon, lat, month).
        sst_anomaly: None (zero anomalies) or a path / mapping with `ssta` (lon, lat, time) and `time` (datetime64,
        first day of each month) covering one month before the start to one month after the end of the run."""
        if self._initialized_bc:
            raise RuntimeError("The model was already initialized. Create a new instance if you need different boundary conditions.")
        self._load_bc(bc_file, sst_anomaly)
        self._init_from_loaded_bc()

    def _load_bc(self, bc_file, sst_anomaly):
        """the boundary fields and SST anomalies into the state container (no initialisation yet)"""
        self._set_sst_anomalies(sst_anomaly)
        fields = _load_fields(example_bc_file() if bc_file is None else bc_file)
        for state_name, file_name in _BC_FIELDS:
            self[state_name] = np.asarray(fields[file_name], dtype=np.float64)

    def _init_from_loaded_bc(self):
        """the reference's `init` on boundary fields that are in the container already (set here, or handed over device to
        device by SpeedyEns.set_bc)"""
        code = _speedy.init(self._state_cnt, self._control_cnt)
        if code < 0:
            raise RuntimeError(ERROR_CODES[code])
        self.spectral2grid()
        self._initialized_bc = True

    def _set_sst_anomalies(self, sst_anomaly=None):
        if self._initialized_ssta:
            raise RuntimeError("The SST anomaly was already initialized. Create a new instance if you need different boundary conditions.")
        # the model interpolates in a 3-month window: months from (start - 1 month) to (end + 1 month), speedy.py:338-372
        first = _add_months(self.start_date.replace(day=1, hour=0, minute=0, second=0, microsecond=0), -1)
        last = _add_months(self.end_date.replace(day=1, hour=0, minute=0, second=0, microsecond=0), 1)
        expected_months = (last.year - first.year) * 12 + (last.month - first.month) + 1
        _speedy.modelstate_init_sst_anom(self._state_cnt, expected_months - 2)
        self.n_months = expected_months - 2
        if sst_anomaly is not None:
            fields = _load_fields(sst_anomaly)
            times = np.asarray(fields["time"], dtype="datetime64[s]")
            keep = (times >= np.datetime64(first, "s")) & (times <= np.datetime64(last + timedelta(days=1), "s"))
            missing = expected_months - int(keep.sum())
            if missing > 0:
                raise RuntimeError("%d months are missing in the SST anomalies for the period: %s , %s.\n"
                                   % (missing, first.strftime("%Y/%m/%d"), last.strftime("%Y/%m/%d")))
            self["sst_anom"] = np.asarray(fields["ssta"], dtype=np.float64)[:, :, keep][:, :, :expected_months]
        self._initialized_ssta = True

    # ---- run -------------------------------------------------------------------------------------------------
    def run(self, callbacks=None):
        """Run from `start_date` to `end_date`, calling every callback after each 40-minute step.  The reference's range
        check runs after every step as upstream.  On steps where no callback is due its result is collected one step later,
        so that the GPU is never left waiting for the host; on a step where a callback IS due the check of that step is
        collected first: as in the reference (speedy.py:398-405) no callback ever sees a state that failed the check."""
        callbacks = list(callbacks or [])
        if not self._initialized_bc:
            raise RuntimeError("The SPEEDY model was not initialized. Call the `set_bc` method to initialize the model.")
        self.current_date = self.start_date
        end_date = self.end_date
        self._step_in_run = self["current_step"]
        _own(callbacks, True)
        intervals, spinups = _hook_intervals(callbacks), _hook_spinups(callbacks)
        if intervals is not None:  # the hooks' schedule is known: the steps between two due hooks are one device call
            rest = []  # what the hooks of the last boundary left to be done once the next stretch is on the device
            try:
                while self.current_date < end_date:
                    k = _stretch(self._step_in_run, intervals, self.current_date, end_date, spinups)
                    before, step_before = self.current_date, self._step_in_run
                    token = _speedy.parallel_steps_begin([self._state_cnt], [self._control_cnt], k)
                    try:
                        _do_rest(rest)
                        self._step_in_run += k
                        self.current_date += k * _DT_STEP
                        due = _callbacks_due(callbacks, self)
                        early = _act_ahead(due, self)  # (hooks that only enqueue device work do so behind the stretch, now)
                    except BaseException:
                        _end_quietly(token)
                        raise
                    codes, done = _speedy.parallel_steps_end(token)
                    if (codes < 0).any():  # the date of the step before the one that failed, as the reference's loop leaves it
                        self._step_in_run = step_before + int(done[0])
                        self.current_date = before + int(done[0]) * _DT_STEP
                        _raise_step_failure(codes)  # (what hooks enqueued ahead is dropped: nothing of it was written)
                    if early is None:
                        _act(due, self, rest)
                    else:
                        rest.extend(early)
            finally:
                self._step_in_run = None
                _finish_all(callbacks, rest)
            return
        pending = None  # the range check of a step is collected after the next step has been enqueued (GPU never idles)
        first_date, first_step, accepted = self.current_date, self._step_in_run, 0  # accepted: steps whose check has come back good
        try:
            while self.current_date < end_date:
                token = _speedy.parallel_step_begin([self._state_cnt], [self._control_cnt])
                previous, pending = pending, token
                accepted += self._collect(previous)
                self._step_in_run += 1
                self.current_date += _DT_STEP
                due = _callbacks_due(callbacks, self)
                if due:
                    previous, pending = pending, None
                    accepted += self._collect(previous)
                    _act(due, self, None)
            previous, pending = pending, None
            accepted += self._collect(previous)
        except RuntimeError:
            if first_step is not None and self._step_in_run is not None and accepted < self._step_in_run - first_step:
                # a step failed its check: the reference's loop raises before it advances the date (speedy.py:398-401), and the
                # loop above had moved it on while the check was still out
                self.current_date = first_date + accepted * _DT_STEP
            raise
        finally:
            self._step_in_run = None
            self._drain(pending)  # (a step that was begun behind the one that failed: end it, its code no longer matters)
            _finish_all(callbacks)

    @staticmethod
    def _collect(token):
        """end a step that was begun (None: nothing to end) -> the number of steps it accepted (0 or 1); raises on a failed check"""
        if token is None:
            return 0
        codes = _speedy.parallel_step_end(token)
        if (codes < 0).any():
            _raise_step_failure(codes)
        return 1

    @staticmethod
    def _drain(token):
        if token is not None:
            try:
                _speedy.parallel_step_end(token)
            except Exception:
                pass

    def grid2spectral(self):
        """Transform the grid u, v, t, q, ps and phi fields to the spectral domain."""
        _speedy.transform_grid2spectral(self._state_cnt)

    def spectral2grid(self):
        """Transform the spectral prognostic fields to u, v, t, q, phi and ps on the grid."""
        _speedy.transform_spectral2grid(self._state_cnt)

    def check(self):
        code = _speedy.check(self._state_cnt)
        if code < 0:
            raise RuntimeError(ERROR_CODES[code])

    # ---- export ----------------------------------------------------------------------------------------------
    def to_dataframe(self, variables=None, packed=False, slot=0, buffers=None, wait=True):
        """Current model state as a Dataset following the export conventions of the reference (speedy.py:415-477).
        packed=True (extension, what XarrayExporter asks for): the data variables come as they go into a NetCDF-3 file -- float32,
        big-endian, narrowed and ordered on the GPU -- and alias a buffer that the next packed call with the same `slot` (of the same
        `buffers` dict, when the caller brings its own) overwrites.  wait=False (packed only): see SpeedyEns.to_dataframe."""
        variables = DEFAULT_OUTPUT_VARS if variables is None else variables
        if packed:
            for var in variables:
                _exportable(var)
            arrays = _speedy.ensemble_export_arrays([self._state_cnt], list(variables), slot=slot, buffers=buffers, wait=wait)
            arrays, ready = arrays if not wait else (arrays, [])
            members = [self.member_id] if self.is_ensemble_member else None
            frame = _build_dataset(self, arrays, members, self.current_date, packed=True)
            frame.ready = ready
            return frame
        self.spectral2grid()
        arrays = {}
        for var in variables:
            _exportable(var)
            values = self[var]  # (lon, lat[, lev]) -> ([lev,] lat, lon)
            arrays[var] = values.transpose(*range(values.ndim - 1, -1, -1))[None]
        members = [self.member_id] if self.is_ensemble_member else None
        return _build_dataset(self, arrays, members, self.current_date)


    def snapshot_on_device(self, variables=None):
        """Extension: see SpeedyEns.snapshot_on_device (None for a member of a device model that holds other members as well)."""
        variables = DEFAULT_OUTPUT_VARS if variables is None else variables
        for var in variables:
            _exportable(var)
        tensors = _speedy.ensemble_export_tensors([self._state_cnt], list(variables))
        if tensors is None:
            return None
        return PendingFrame(self, tensors, [self.member_id] if self.is_ensemble_member else None, self.current_date)


# The time loops below take the steps between two due callbacks as ONE device call (speedy_driver.parallel_steps_begin / _end): the
# device then runs its multi-step plan -- member groups on streams of their own, large ensembles in rounds -- instead of being
# asked once per 40 simulated minutes, and the range check the reference makes after every step (speedy.py:398-400) is still
# made after every step, by the device.  That needs the schedule of the hooks: known for a BaseCallback with the stock gating
# (`interval` / `spinup_date`: it can only act where current_step is a multiple of its interval); a plain callable, a hook that
# overrides __call__ or skip_flag, or an interval that is not a positive integer may act at every step, and the loop then asks
# step by step as before.  A stretch is at most _MAX_STRETCH steps long (ten model days).
_MAX_STRETCH = 360


def _hook_intervals(callbacks):
    """[interval of every hook] when the schedule of all of them is known, else None"""
    from .callbacks import BaseCallback
    intervals = []
    for cb in callbacks:
        stock = (isinstance(cb, BaseCallback) and type(cb).__call__ is BaseCallback.__call__
                 and type(cb).skip_flag is BaseCallback.skip_flag)
        interval = getattr(cb, "interval", None)
        if not stock or isinstance(interval, bool) or not isinstance(interval, (int, np.integer)) or interval < 1:
            return None
        intervals.append(int(interval))
    return intervals


def _hook_spinups(callbacks):
    """[spinup_date of every hook] (None: acts from the start), beside _hook_intervals"""
    return [getattr(cb, "spinup_date", None) for cb in callbacks]


def _stretch(step, intervals, current_date, end_date, spinups=None):
    """steps until the next one at which a hook may act, the end of the run or _MAX_STRETCH, whichever comes first.  A hook that is
    still spinning up (callbacks.py:52-70: silent while the model's date is before its `spinup_date`) may act at the first multiple
    of its interval whose date is not: the multiples before that one do not end a stretch."""
    remaining = -((current_date - end_date) // _DT_STEP)  # ceil((end - current) / dt)
    k = min(int(remaining), _MAX_STRETCH)
    for n, interval in enumerate(intervals):
        k_hook = interval - step % interval
        spinup = spinups[n] if spinups else None
        if spinup is not None and current_date + k_hook * _DT_STEP < spinup:
            first = step + int(-((current_date - spinup) // _DT_STEP))  # the first step whose date is not before the spin-up date
            k_hook = first + (-first) % interval - step
        k = min(k, k_hook)
    return max(k, 1)


def _raise_step_failure(codes):
    raise RuntimeError("".join("Member%d: %s\n" % (n, ERROR_CODES[int(c)]) for n, c in enumerate(codes))
                       if len(codes) > 1 else ERROR_CODES[int(codes[0])])


def _act(due, model, rest):
    """let the due hooks act; what one of them leaves to be done without the model's state (a callable returned by `fire`,
    callbacks.BaseCallback.fire) goes on the list `rest` -- or, without a list, happens at once"""
    for act in due:
        left = act(model)
        if callable(left):
            if rest is None:
                left()
            else:
                rest.append(left)


def _do_rest(rest):
    while rest:
        rest.pop(0)()


def _act_ahead(due, model):
    """The hooks due at the END of the stretch that has just been handed to the device, fired before the host waits for it --
    possible when every one of them says (`acts_ahead(model)`, callbacks.XarrayExporter) that all its `fire` does with the model's
    state is to ENQUEUE device work: that work then runs right behind the stretch's last step, in stream order, instead of after the
    host has woken up, looked at the range checks and walked through the hook.  Returns what the hooks left to be done (the list
    for `rest`) -- to be thrown away by the caller if a step of the stretch turns out to have failed its check: as in the reference
    (speedy.py:398-405) nothing a hook produces ever comes from a state that failed it -- or None: some hook needs the state on
    the host, all of them act after the stretch has been waited for, as before."""
    if not due:
        return []
    for act in due:
        ahead = getattr(getattr(act, "__self__", None), "acts_ahead", None)
        if not callable(ahead) or not ahead(model):
            return None
    left = []
    _act(due, model, left)
    return left


def _end_quietly(token):
    """a stretch that was begun is ended on every way out of the loop (its device model takes no other call before that); the
    exception that is under way is the one to report"""
    try:
        _speedy.parallel_steps_end(token)
    except Exception:
        pass


def _own(callbacks, yes):
    """tell the hooks that a run owns them (XarrayExporter then writes behind the time loop: the run will call finish())"""
    for cb in callbacks:
        if hasattr(cb, "_in_run"):
            cb._in_run = yes


def _finish(callbacks):
    """end of a run: hooks that work in the background (XarrayExporter's file writer) complete what they hold"""
    failure = None
    for cb in callbacks:
        done = getattr(cb, "finish", None)
        if callable(done):
            try:
                done()
            except BaseException as exc:  # noqa: B902 -- every hook gets its turn; the first failure is reported
                failure = failure or exc
    if failure is not None:
        raise failure


def _finish_all(callbacks, rest=None):
    """_finish from the `finally` of a time loop, after what the hooks of the last boundary left to be done (every item gets its
    turn, and the hooks' finish() is called whatever one of them did): a failure of the run itself is not replaced by what a
    writer could not do -- the writer's error is attached to it (__context__ does that) and the run's exception travels on;
    without one, the first failure of the clean-up is raised"""
    import sys
    running = sys.exc_info()[1]
    _own(callbacks, False)
    failure = None
    while rest:
        try:
            rest.pop(0)()
        except BaseException as exc:  # noqa: B902
            failure = failure or exc
    try:
        _finish(callbacks)
    except BaseException as exc:  # noqa: B902
        failure = failure or exc
    if failure is None:
        return
    if running is None:
        raise failure
    if running.__context__ is None and failure is not running:
        running.__context__ = failure


def _callbacks_due(callbacks, model):
    """What acts at this step, in order: for a hook with the reference's gating (BaseCallback: skip_flag / fire) the gate is asked
    ONCE and its `fire` is what is due; anything else -- a plain callable, a hook that overrides __call__ -- always acts, through
    its own __call__."""
    from .callbacks import BaseCallback
    due = []
    for cb in callbacks:
        if isinstance(cb, BaseCallback) and type(cb).__call__ is BaseCallback.__call__:
            if not cb.skip_flag(model):
                due.append(cb.fire)
        else:
            due.append(cb)
    return due


def _exportable(var):
    meta = REGISTRY[var]
    if meta.nc_dims is None or meta.where != "device":
        raise ValueError("'%s' cannot be exported: not a grid-space array" % var)
    return meta


def _export_coords(model):
    """lon, lat and lev as a Dataset carries them.  Tables of the geometry, the same for every step: asked for once per model object
    -- each `model[c]` is a synchronous copy from the device, which an export that otherwise only enqueues work would have to
    wait for."""
    cached = model.__dict__.setdefault("_export_coords", {})
    for c in ("lon", "lat", "lev"):
        if c not in cached:
            vals = model[c][::-1] if c == "lev" else model[c]
            cached[c] = np.ascontiguousarray(vals, dtype=np.float32)
            cached[c].setflags(write=False)
    return cached


def _build_dataset(model, arrays, members, date, packed=False):
    """arrays: var -> [member, (lev,) lat, lon] float64 in model level order -- or, packed, float32 with the levels already
    bottom-up (speedy_driver.ensemble_export_arrays); members: list of ids or None (single run); model: the Speedy object the
    coordinates are asked of, or the dict _export_coords made of them."""
    lead = ("time", "ens") if members is not None else ("time",)
    data = {}
    for var, values in arrays.items():
        meta = _exportable(var)
        dims = tuple(reversed(meta.nc_dims))
        if not packed:
            if "lev" in dims:
                values = values[:, ::-1]  # vertical levels increasing with height (lev coordinate reversed)
            values = values.astype(np.float32)  # one pass: reversed view -> contiguous float32
        values = values[None] if members is not None else values  # -> (time, ens, ...) or (time, ...)
        attrs = {"long_name": meta.long_name, "standard_name": var}
        if meta.units is not None:
            attrs["units"] = meta.units
        data[meta.alt_name] = Variable(lead + dims, values, attrs)
    coords = {}
    cached = model if isinstance(model, dict) else _export_coords(model)
    for c, axis in (("lon", "X"), ("lat", "Y"), ("lev", None)):
        meta = REGISTRY[c]
        attrs = {"long_name": meta.long_name, "standard_name": c}
        if meta.units is not None:
            attrs["units"] = meta.units
        if axis:
            attrs["axis"] = axis
        coords[c] = Variable((c,), cached[c], attrs)
    coords["time"] = Variable(("time",), np.array([np.datetime64(date, "s")]), {"axis": "T", "standard_name": "time"})
    if members is not None:
        coords["ens"] = Variable(("ens",), np.array(members, dtype=np.int32))
    return Dataset(data, coords)


class PendingFrame:
    """A snapshot of an ensemble that is still on the GPU (SpeedyEns.snapshot_on_device): `resolve()` copies it out and returns the
    Dataset `to_dataframe` would have returned at the time it was taken; `nbytes` is what it holds on the device until then."""

    def __init__(self, model, tensors, members, date):
        # (the coordinates, not the model: a hook that holds frames is copied with them -- callbacks.BaseCallback.copy -- and a
        # model object owns its containers)
        self._coords, self._tensors, self._members, self._date, self._frame = _export_coords(model), tensors, members, date, None
        self.nbytes = sum(t.numel() * t.element_size() for t in tensors.values())

    def resolve(self):
        if self._frame is None:
            arrays = {name: t.cpu().numpy() for name, t in self._tensors.items()}
            self._frame = _build_dataset(self._coords, arrays, self._members, self._date, packed=True)
            self._tensors, self.nbytes = None, 0
        return self._frame


class SpeedyEns:
    """Ensemble of Speedy members that live in one batched device model -- or, with `devices=k`, in one batched model on
    each of the GPUs 0 .. k-1 of this process (members in blocks, member e of n on device e k / n): `run` then drives all
    devices with one parallel_step per model step.  `devices=None` keeps the process-wide placement
    (speedy_driver.set_device_placement / PYSPEEDY_AMD_DEVICES; by default the current device)."""

    def __init__(self, num_of_members, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 2), devices=None):
        self.n_members = int(num_of_members)
        # (ONE device model per GPU: `run` hands the steps between two due callbacks over as one call, and the model's own multi-step
        # plan forms the member groups and rounds; a host that steps one by one gets two models per GPU from 32 members up)
        cnts = _speedy.modelstate_init_ensemble(self.n_members, devices=None if devices is None else int(devices), whole=True)
        self.members = [Speedy(start_date=start_date, end_date=end_date, member=i, _state_cnt=c) for i, c in enumerate(cnts)]
        self.current_date = self.members[0].current_date

    def __iter__(self):
        return iter(self.members)

    def __len__(self):
        return self.n_members

    def set_params(self, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 2)):
        for member in self:
            member.set_params(start_date=start_date, end_date=end_date)
        self.current_date = start_date

    def set_bc(self, bc_file=None, sst_anomaly=None):
        """Extension (the reference initialises an ensemble with `for member in ens: member.set_bc()`, which still works): load
        the boundary conditions ONCE, into member 0, hand them to every other member device to device -- across the GPUs of a
        `devices=k` ensemble with one RCCL broadcast over xGMI, on a GPU with local copies (speedy_driver.broadcast_boundary) --
        and initialise every member.  The file is read and its 3.4 MB cross PCIe once instead of once per member; the members
        are bitwise what `member.set_bc(bc_file, sst_anomaly)` gives each of them."""
        for member in self:
            if member._initialized_bc:
                raise RuntimeError("The model was already initialized. Create a new instance if you need different boundary conditions.")
        self.members[0]._load_bc(bc_file, sst_anomaly)
        for member in self.members[1:]:
            member._set_sst_anomalies(None)  # (allocates the same number of months: the anomalies travel with the other fields)
        _speedy.broadcast_boundary([m._state_cnt for m in self], 0)
        codes = _speedy.init_ensemble([m._state_cnt for m in self], [m._control_cnt for m in self])  # one pass per device model
        if (codes < 0).any():
            raise RuntimeError("".join("Member%d: %s\n" % (n, ERROR_CODES[int(c)]) for n, c in enumerate(codes)))
        for member in self:
            member.spectral2grid()
            member._initialized_bc = True

    def to_dataframe(self, variables=None, packed=False, slot=0, buffers=None, wait=True):
        """All members along the `ens` dimension: one batched spectral -> grid conversion and one device-to-host copy per
        variable (the device layout [member][lev][lat][lon] is already the export order).  packed=True: see Speedy.to_dataframe;
        with wait=False (packed only) the Dataset comes back as soon as the transforms and pack kernels are enqueued and carries
        `ready`: its arrays hold the payload once `synchronize()` of every item of that list has returned -- which is also what
        copies it out of the device (speedy_driver.ensemble_export_arrays; to be called from the thread that writes the file)."""
        variables = DEFAULT_OUTPUT_VARS if variables is None else variables
        for var in variables:
            _exportable(var)
        cnts = [m._state_cnt for m in self]
        ready = []
        if packed and not wait:
            arrays, ready = _speedy.ensemble_export_arrays(cnts, list(variables), slot=slot, buffers=buffers, wait=False)
        else:
            # (not packed: float32 with the levels bottom-up all the same, formed on the GPU -- in the host's byte order and in
            # memory of the Dataset's own, for hooks that keep what they are given)
            arrays = (_speedy.ensemble_export_arrays(cnts, list(variables), slot=slot, buffers=buffers) if packed else
                      _speedy.ensemble_grid_arrays(cnts, list(variables), narrow=True))
        frame = _build_dataset(self.members[0], arrays, [m.member_id for m in self], self.current_date, packed=True)
        frame.ready = ready
        return frame

    def snapshot_on_device(self, variables=None):
        """Extension: what `to_dataframe(variables)` would return, kept on the GPU until it is read -- a PendingFrame whose
        `resolve()` is that Dataset -- or None when the ensemble does not live in one device model (callers then ask
        `to_dataframe`).  Only enqueues device work (speedy_driver.ensemble_export_tensors); for hooks that keep a time series
        (callbacks.ModelCheckpoint)."""
        variables = DEFAULT_OUTPUT_VARS if variables is None else variables
        for var in variables:
            _exportable(var)
        tensors = _speedy.ensemble_export_tensors([m._state_cnt for m in self], list(variables))
        if tensors is None:
            return None
        return PendingFrame(self.members[0], tensors, [m.member_id for m in self], self.current_date)

    def _device_models(self):
        """[(EnsembleModel view, member_id of the container that is member 0 of that model)] for the device models the
        ensemble lives in (one per GPU; two from 32 members of a GPU up once it has been stepped one step at a time)."""
        found = {}
        for m in self.members:
            model, index = _speedy.device_model(m._state_cnt)
            entry = found.setdefault(model._m.value, [model, None])
            if index == 0:
                entry[1] = m.member_id
        return [(model, first) for model, first in found.values()]

    def set_sppt(self, on=True, seed=0):
        """BASELINE cfg 5: switch the deterministic SPPT scheme (csrc/sppt.hip; compile-time off and non-functional in the
        reference, parity unpinned) on or off for every member.  The noise of a member is keyed by its `member_id`, so it does
        not depend on how the ensemble is grouped into device models or sharded over GPUs."""
        for model, first in self._device_models():
            model.set_sppt(on, seed=seed, first_member_id=0 if first is None else first)

    def set_physics_precision(self, fp32):
        """BASELINE cfg 5: fp32 arithmetic in the column physics of every member (state and dynamics stay fp64)."""
        for model, _ in self._device_models():
            model.set_physics_precision(fp32)

    def device_view(self, name, spectral2grid=False):
        """The registry variable `name` of all members as one device tensor [member, ...] (speedy_driver.ensemble_device_view):
        zero-copy while the ensemble lives in one device model, gathered on the device otherwise (32 or more members are kept
        as two models, `devices=k` as one or two per GPU)."""
        return _speedy.ensemble_device_view([m._state_cnt for m in self], name, spectral2grid)

    def run(self, callbacks=None):
        """Advance every member from the start to the end date; all members step together (parallel_step)."""
        callbacks = list(callbacks or [])
        end_date = self.members[0].end_date
        state_cnts = np.array([m._state_cnt for m in self], dtype=np.int64)
        control_cnts = np.array([m._control_cnt for m in self], dtype=np.int64)
        for member in self:
            if not member._initialized_bc:
                raise RuntimeError("The SPEEDY model was not initialized. Call the `set_bc` method of every member.")
        step = self.members[0]["current_step"]
        _own(callbacks, True)
        intervals, spinups = _hook_intervals(callbacks), _hook_spinups(callbacks)
        if intervals is not None:  # (see Speedy.run)
            rest = []
            try:
                while self.current_date < end_date:
                    k = _stretch(step, intervals, self.current_date, end_date, spinups)
                    before = self.current_date
                    token = _speedy.parallel_steps_begin(state_cnts, control_cnts, k)
                    try:
                        _do_rest(rest)
                        step += k
                        self.current_date += k * _DT_STEP
                        for member in self:
                            member.current_date = self.current_date
                            member._step_in_run = step
                        due = _callbacks_due(callbacks, self)
                        early = _act_ahead(due, self)  # (see Speedy.run)
                    except BaseException:
                        _end_quietly(token)
                        raise
                    codes, done = _speedy.parallel_steps_end(token)
                    if (codes < 0).any():
                        # The reference's loop (speedy.py:572-586) stops at the first step any member fails; it has moved the
                        # ENSEMBLE's date past that step by then and not yet handed it to the members, who keep the date before it.
                        first = int(done[codes < 0].min())
                        for member in self:
                            member.current_date = before + first * _DT_STEP
                        self.current_date = before + (first + 1) * _DT_STEP
                        _raise_step_failure(codes)
                    if early is None:
                        _act(due, self, rest)
                    else:
                        rest.extend(early)
            finally:
                for member in self:
                    member._step_in_run = None
                _finish_all(callbacks, rest)
            return
        pending = None
        first_date, first_step, accepted = self.current_date, step, 0
        try:
            while self.current_date < end_date:
                token = _speedy.parallel_step_begin(state_cnts, control_cnts)
                previous, pending = pending, token
                accepted += Speedy._collect(previous)
                step += 1
                self.current_date += _DT_STEP
                for member in self:
                    member.current_date = self.current_date
                    member._step_in_run = step
                due = _callbacks_due(callbacks, self)
                if due:
                    previous, pending = pending, None
                    accepted += Speedy._collect(previous)
                    _act(due, self, None)
            previous, pending = pending, None
            accepted += Speedy._collect(previous)
        except RuntimeError:
            if accepted < step - first_step:  # a step failed its check: the dates as the reference's loop leaves them (see above)
                for member in self:
                    member.current_date = first_date + accepted * _DT_STEP
                self.current_date = first_date + (accepted + 1) * _DT_STEP
            raise
        finally:
            for member in self:
                member._step_in_run = None
            Speedy._drain(pending)
            _finish_all(callbacks)

    def get_current_step(self):
        return self.members[0].get_current_step()
