"""Callbacks: objects called with the model instance after every time step (pyspeedy/callbacks.py:31-255).

BaseCallback (interval / spin-up gating), DiagnosticCheck, ModelCheckpoint (time series kept in memory) and
XarrayExporter (one NetCDF file per output time).  Names and constructor arguments follow the reference; the files are
NetCDF-3 classic written by pyspeedy_amd.dataset (the reference's own fixtures are in that format).
"""
import copy
import os

from .dataset import concat
from .registry import DEFAULT_OUTPUT_VARS
from .speedy import Speedy


class BaseCallback:
    def __init__(self, *args, **kwargs):
        """interval: apply every `interval` time steps; verbose: print progress; spinup_date: ignore calls before it."""
        self.verbose = kwargs.pop("verbose", False)
        self.interval = kwargs.pop("interval", 1)
        self.spinup_date = kwargs.pop("spinup_date", None)

    def skip_flag(self, model_instance):
        """True when this time step is skipped: still in the spin-up period, or not a multiple of `interval`."""
        if self.spinup_date is not None and model_instance.current_date < self.spinup_date:
            return True
        return model_instance.get_current_step() % self.interval != 0

    def print_msg(self, msg):
        if self.verbose:
            print(msg)

    def copy(self):
        return copy.deepcopy(self)

    def __call__(self, model_instance):
        pass


class DiagnosticCheck(BaseCallback):
    """Check that the prognostic variables are inside their accepted ranges (diagnostics.f90) every `interval` steps."""

    def __init__(self, interval=36):
        super().__init__(interval=interval)

    def __call__(self, model_instance):
        if self.skip_flag(model_instance):
            return
        members = [model_instance] if isinstance(model_instance, Speedy) else model_instance
        for member in members:
            member.check()  # raises RuntimeError when a range test fails


class ModelCheckpoint(BaseCallback):
    """Keep a time series of selected grid-space variables in memory (`dataframe`), one entry every `interval` steps."""

    def __init__(self, interval=36, verbose=False, spinup_date=None, variables=None, output_dir="./"):
        self.variables = DEFAULT_OUTPUT_VARS if variables is None else variables
        self.output_dir = output_dir
        self.history_interval = interval
        super().__init__(verbose=verbose, interval=interval, spinup_date=spinup_date)
        self.dataframe = None

    def __call__(self, model_instance):
        if self.skip_flag(model_instance):
            return
        snapshot = model_instance.to_dataframe(variables=self.variables)
        self.dataframe = snapshot if self.dataframe is None else concat((self.dataframe, snapshot), "time")


class XarrayExporter(BaseCallback):
    """Write selected grid-space variables to `output_dir/<filename_fmt of the model date>` every `interval` steps.
    For an ensemble the file holds all members along the `ens` dimension."""

    def __init__(self, interval=36, verbose=False, spinup_date=None, variables=None, output_dir="./",
                 filename_fmt="%Y-%m-%d_%H%M.nc"):
        self.variables = DEFAULT_OUTPUT_VARS if variables is None else variables
        self.output_dir = output_dir
        self.filename_fmt = filename_fmt
        self.history_interval = interval
        super().__init__(verbose=verbose, interval=interval, spinup_date=spinup_date)

    def __call__(self, model_instance):
        if self.skip_flag(model_instance):
            return
        snapshot = model_instance.to_dataframe(variables=self.variables)
        os.makedirs(self.output_dir, exist_ok=True)
        path = os.path.join(self.output_dir, model_instance.current_date.strftime(self.filename_fmt))
        self.print_msg("Saving model output at: %s." % path)
        snapshot.to_netcdf(path)


NetcdfExporter = XarrayExporter
