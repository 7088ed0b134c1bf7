#!/usr/bin/env python3
"""The reference's ensemble example (examples/Ensemble_forecast.ipynb, its model and post-processing cells) with the import
changed and nothing else:

    from pyspeedy import SpeedyEns                   ->  from pyspeedy_amd import SpeedyEns
    from pyspeedy.callbacks import ...               ->  from pyspeedy_amd.callbacks import ...

Ten members, initialised member by member as the notebook does (set_bc, a perturbation of t_grid, grid2spectral), daily
checkpoints after a spin-up month, the diagnostic check every 160 steps; then the spread statistics of the notebook on the
checkpoint dataframe.  All members advance with one set of kernel launches per step (the reference runs one member per thread).
examples/ensemble_one_process.py / ensemble_multi_gpu.py are the same forecast at scale, with the library's extensions.

    python examples/ensemble_forecast.py [--members 10] [--end 1980-02-29] [--spinup 1980-02-01]
"""
import argparse
import os
import sys
from datetime import datetime

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspeedy_amd import SpeedyEns  # noqa: E402
from pyspeedy_amd.callbacks import DiagnosticCheck, ModelCheckpoint  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--members", type=int, default=10)
ap.add_argument("--end", default="1980-02-29")
ap.add_argument("--spinup", default="1980-02-01")
args = ap.parse_args()
np.random.seed(0)  # (the notebook draws unseeded perturbations)

# Definitions
number_of_members = args.members
start_date = datetime(1980, 1, 1)  # Simulation start date (datetime object).
end_date = datetime.strptime(args.end, "%Y-%m-%d")  # Simulation end date.
spinup_date = datetime.strptime(args.spinup, "%Y-%m-%d")  # End of spinup period.

# Create an instance of the speedy model.
model_ens = SpeedyEns(number_of_members, start_date=start_date, end_date=end_date)
# At this point, each ensemble member contains an "empty" (not initialized) the model state.  To initialize them we iterate
# over each member, set the boundary conditions, and add a random perturbation.
for member in model_ens:
    # Set the default boundary conditions derived from the ERA reanalysis.
    member.set_bc()
    # Add a perturbation to the temperature field in the grid space (not in the spectral one)
    member["t_grid"] += np.random.normal(0.0, 0.01, member["t_grid"].shape)
    # The prognostic variables used for the model integration are in the spectral space: convert the grid variables.
    member.grid2spectral()

# The callback that will store the forecast in a dataframe with selected variables ("dataframe" attribute).
model_checkpoints = ModelCheckpoint(interval=36, verbose=True, variables=None, spinup_date=spinup_date)
# A diagnostic check: if some diagnostic values are out of range, an exception is raised and the model stops.
diag_checks = DiagnosticCheck(interval=160)

# Run the model passing our callbacks.
model_ens.run(callbacks=[model_checkpoints, diag_checks])
# After the ensemble model is run, the model state contains the values from the last integration step.

ens_dataset = model_checkpoints.dataframe
print(ens_dataset)

spr_ds = ens_dataset.var(dim="ens").mean(dim=["lev", "lat", "lon"]).apply(np.sqrt)
# Copy attributes from the ens_dataset
for var in spr_ds:
    spr_ds[var].attrs.update(**ens_dataset[var].attrs)
print("Domain-averaged ensemble spread by day:")
for var in spr_ds:
    print("  %-4s %s" % (var, " ".join("%.4g" % v for v in spr_ds[var].values)))

spr_ds = ens_dataset.std(dim="ens").apply(np.sqrt).isel(lev=0)  # keep first level (surface)
for var in spr_ds:
    spr_ds[var].attrs.update(**ens_dataset[var].attrs)
print("Spread at the lowest level on the last day, t: max %.4g at a grid point" % spr_ds["t"].values[-1].max())
