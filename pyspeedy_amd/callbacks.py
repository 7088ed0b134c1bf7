"""Per-step hooks for `Speedy.run` / `SpeedyEns.run`, API-compatible with the reference's pyspeedy/callbacks.py:31-255
(BaseCallback, DiagnosticCheck, ModelCheckpoint, XarrayExporter; same constructor keywords, `interval` / `spinup_date` gating,
`skip_flag`, `print_msg`, `copy`).

Design: gating lives in ONE place.  `BaseCallback.__call__` decides whether the hook is due at this step and, if so, hands the
model to `fire()`; the concrete hooks only implement `fire()`.  Output goes through pyspeedy_amd.dataset (NetCDF-3 classic,
the format of the reference's own fixtures); for an ensemble one file holds all members along `ens`.
"""
import copy as _copy
import os

from . import dataset as _dataset
from .registry import DEFAULT_OUTPUT_VARS
from .speedy import Speedy


class BaseCallback:
    """interval: fire every `interval` model steps; spinup_date: stay silent before that date; verbose: print progress."""

    def __init__(self, *args, **kwargs):
        self.verbose = kwargs.pop("verbose", False)
        self.interval = kwargs.pop("interval", 1)
        self.spinup_date = kwargs.pop("spinup_date", None)

    # -- gating ------------------------------------------------------------------------------------------------
    def skip_flag(self, model_instance):
        """True when nothing is due at this step: still spinning up, or not a multiple of `interval`."""
        spinning_up = self.spinup_date is not None and model_instance.current_date < self.spinup_date
        return spinning_up or model_instance.get_current_step() % self.interval != 0

    def __call__(self, model_instance):
        if not self.skip_flag(model_instance):
            rest = self.fire(model_instance)
            if callable(rest):  # (called by hand there is no time loop to hand the rest to: it happens here)
                rest()

    def fire(self, model_instance):
        """What the hook does when it is due (nothing in the base class).  It may return a callable: the part of its work that no
        longer needs the model's state (a file header, handing data to a writer).  `Speedy.run` / `SpeedyEns.run` call it after they
        have handed the next stretch of steps to the device, so that it runs beside the GPU instead of in front of it."""

    # -- helpers -----------------------------------------------------------------------------------------------
    def print_msg(self, msg):
        if self.verbose:
            print(msg)

    def copy(self):
        return _copy.deepcopy(self)


class DiagnosticCheck(BaseCallback):
    """Range check of the prognostic variables (diagnostics.f90) every `interval` steps; raises RuntimeError on failure."""

    def __init__(self, interval=36):
        super().__init__(interval=interval)

    def fire(self, model_instance):
        members = [model_instance] if isinstance(model_instance, Speedy) else list(model_instance)
        if len(members) > 1:
            # one check per device model first; only when that finds something (or cannot be made) the reference's loop over the
            # members (callbacks.py:96-112), which reports and raises as the reference does
            from . import speedy_driver as _speedy
            try:
                if not _speedy.ensemble_check([member._state_cnt for member in members]).any():
                    return
            except RuntimeError:
                pass
        for member in members:
            member.check()


class _GridOutput(BaseCallback):
    """Common part of the two output hooks: which variables, how often, where."""

    def __init__(self, interval, verbose, spinup_date, variables, output_dir):
        super().__init__(verbose=verbose, interval=interval, spinup_date=spinup_date)
        self.variables = DEFAULT_OUTPUT_VARS if variables is None else variables
        self.output_dir = output_dir
        self.history_interval = interval

    def snapshot(self, model_instance):
        return model_instance.to_dataframe(variables=self.variables)


class ModelCheckpoint(_GridOutput):
    """Accumulates the selected grid-space variables as a time series in memory (`dataframe`).  The reference joins every new
    snapshot to the series at once (callbacks.py:175-180, an outer merge: a copy of everything kept so far per output); here the
    snapshots are kept as they come and joined when `dataframe` is read -- the same Dataset, one copy.  While a `SpeedyEns.run`
    owns the hook, the snapshots of an ensemble that lives in one device model stay ON THE GPU until then (float32, 48 MB per day
    of 64 members with the default variables; `device_bytes`, 8 GiB by default, is how much of them may wait there -- beyond it the
    oldest are copied out): taking one only enqueues device work, and the run may ask for it while the steps that lead to it are
    still on the device (`acts_ahead`; dropped, like every hook's output, if one of those steps fails its range check)."""

    def __init__(self, interval=36, verbose=False, spinup_date=None, variables=None, output_dir="./", device_bytes=8 << 30):
        super().__init__(interval, verbose, spinup_date, variables, output_dir)
        self.device_bytes = device_bytes
        self._frames = []
        self._in_run = False  # set by the time loops of speedy.py around the run that owns this hook

    @property
    def dataframe(self):
        if not self._frames:
            return None
        if len(self._frames) > 1 or not isinstance(self._frames[0], _dataset.Dataset):
            frames = [f if isinstance(f, _dataset.Dataset) else f.resolve() for f in self._frames]
            self._frames = [frames[0] if len(frames) == 1 else _dataset.concat(frames, "time")]
        return self._frames[0]

    @dataframe.setter
    def dataframe(self, value):
        self._frames = [] if value is None else [value]

    def _on_device(self, model_instance):
        """whether this output is taken on the GPU: inside a run, of a model that offers it, with the stock `snapshot`"""
        return (self._in_run and self.device_bytes > 0 and callable(getattr(model_instance, "snapshot_on_device", None))
                and type(self).snapshot is _GridOutput.snapshot)

    def acts_ahead(self, model_instance):
        if not self._on_device(model_instance):
            return False
        from . import speedy_driver as _speedy
        cnts = [member._state_cnt for member in ([model_instance] if isinstance(model_instance, Speedy) else model_instance)]
        # (an ensemble spread over several device models takes the host path, which waits for the state: not ahead of it)
        return _speedy.whole_device_model(cnts) and _speedy.on_default_streams(cnts)

    def fire(self, model_instance):
        frame = model_instance.snapshot_on_device(self.variables) if self._on_device(model_instance) else None
        if frame is None:
            frame = self.snapshot(model_instance)

        def keep():  # (not before the state has passed its range check: the time loop calls this, or drops it)
            self._frames.append(frame)
            waiting = [f for f in self._frames if not isinstance(f, _dataset.Dataset)]
            while waiting and sum(f.nbytes for f in waiting) > self.device_bytes:
                waiting.pop(0).resolve()
        return keep


class XarrayExporter(_GridOutput):
    """Writes the selected grid-space variables to `output_dir/<model date formatted with filename_fmt>`.

    The file's payload (float32, big-endian, levels bottom-up) is formed on the GPU and copied out as such
    (`to_dataframe(packed=True)`).  `background=None` (the default): while a `Speedy.run` / `SpeedyEns.run` owns the hook, `fire`
    only enqueues the transforms and the pack kernels (the run may call it while the steps that lead to this output are still on
    the device, `acts_ahead`); the copy to pinned memory, the header and the file follow once the next stretch of steps has been
    handed to the device -- for an ensemble of 8 members or more in a thread of this exporter (two output buffers alternate, a
    third output waits for the first file to be finished), for anything smaller in the time loop's own thread.  The run calls
    `finish()` when it ends, also when it ends with an exception: every file is on disk then (and what the writer could not do is
    raised).  Called by hand, outside a run, the hook writes inside the call, as the reference's exporter does: the file is
    complete when it returns.  `background=True` defers wherever it is called (the caller owes a `finish()`), `background=False`
    never does."""

    def __init__(self, interval=36, verbose=False, spinup_date=None, variables=None, output_dir="./",
                 filename_fmt="%Y-%m-%d_%H%M.nc", background=None):
        super().__init__(interval, verbose, spinup_date, variables, output_dir)
        self.filename_fmt = filename_fmt
        self.background = background
        self._buffers = {}            # this exporter's own two pinned output buffers (speedy_driver.ensemble_export_arrays)
        self._pending = [None, None]  # per output buffer: the thread that is writing from it
        self._turn = 0
        self._failure = None
        self._in_run = False  # set by the time loops of speedy.py around the run that owns this hook

    def _deferred(self):
        """whether the file is finished after `fire` has returned (the run that owns the hook, or the caller, owes a finish())"""
        return self._in_run if self.background is None else bool(self.background)

    def acts_ahead(self, model_instance):
        """True when all `fire` will do with the model's state is enqueue device work (the transforms and the pack kernels): the
        time loop may then call it while the stretch of steps that ends at this output is still running on the device
        (speedy._act_ahead), and drops what it returns if one of those steps fails its range check."""
        if not (self._in_run and self._deferred()):
            return False
        from . import speedy_driver as _speedy
        members = [model_instance] if isinstance(model_instance, Speedy) else list(model_instance)
        return _speedy.on_default_streams([member._state_cnt for member in members])  # (what is enqueued must be ORDERED behind the steps)

    def fire(self, model_instance):
        target = os.path.join(self.output_dir, model_instance.current_date.strftime(self.filename_fmt))
        os.makedirs(self.output_dir, exist_ok=True)
        self.print_msg("Saving model output at: %s." % target)
        if not self._deferred():
            model_instance.to_dataframe(variables=self.variables, packed=True, buffers=self._buffers).to_netcdf(target)
            return
        slot = self._turn
        self._turn = 1 - slot
        self._wait(slot)  # (the buffer this output goes into may still be on its way to disk)
        # (wait=False: the transforms and pack kernels are enqueued, nothing else; whoever writes the file waits for them and has
        # the payload copied to this slot's pinned buffer -- by an SDMA engine, beside the next stretch of the time loop)
        frame = model_instance.to_dataframe(variables=self.variables, packed=True, slot=slot, buffers=self._buffers, wait=False)
        ready = list(getattr(frame, "ready", ()))
        if getattr(model_instance, "n_members", 1) < 8:
            # (a single model's day is 0.8 MB: handing it to a thread costs more than writing it -- the time loop's own thread
            # does, once the next stretch is on the device)
            def write_now():
                for copy in ready:
                    copy.synchronize()
                _dataset.write_prepared(target, _dataset.prepare_netcdf(frame))
            return write_now
        import threading

        def hand_over():
            # (the header is made here, not in the writer: a thread that runs Python code competes with the time loop for the
            # interpreter lock, one that only waits and writes bytes does not)
            prepared = _dataset.prepare_netcdf(frame)

            def write():
                try:
                    for copy in ready:
                        copy.synchronize()
                    _dataset.write_prepared(target, prepared)
                except BaseException as exc:  # noqa: B902 -- handed to the thread that owns the exporter
                    self._failure = exc
            self._pending[slot] = threading.Thread(target=write, name="pyspeedy_amd-export", daemon=False)
            self._pending[slot].start()
        # (what needed the state is enqueued; the rest is the time loop's to call once the next stretch is on the device)
        return hand_over

    def _wait(self, slot):
        thread, self._pending[slot] = self._pending[slot], None
        if thread is not None:
            thread.join()
        if self._failure is not None:
            failure, self._failure = self._failure, None
            raise failure

    def finish(self):
        """Every file handed to the writer so far is complete when this returns."""
        for slot in (0, 1):
            self._wait(slot)

    def copy(self):
        self.finish()  # (threads do not copy, and a copy gets buffers of its own)
        buffers, self._buffers = self._buffers, {}
        try:
            return super().copy()
        finally:
            self._buffers = buffers

    def __del__(self):
        try:
            for thread in self._pending:
                if thread is not None:
                    thread.join()
        except Exception:
            pass


NetcdfExporter = XarrayExporter
