#!/usr/bin/env python3
"""The reference's first example (examples/My_first_forecast.ipynb, its model cell) with the import changed and nothing else:

    from pyspeedy import Speedy                      ->  from pyspeedy_amd import Speedy
    from pyspeedy.callbacks import ...               ->  from pyspeedy_amd.callbacks import ...

One member from 1980-01-01 to 1980-02-29, the first month as spin-up, daily NetCDF files from XarrayExporter and the daily
checkpoints ModelCheckpoint keeps in memory; then the fields the notebook plots, read from the model state (`model["t_grid"]`,
`model["ps_grid"]`, `model["lon"]`, `model["lat"]`) and summarised instead of plotted.

    python examples/my_first_forecast.py [--end 1980-02-29] [--spinup 1980-02-01] [--out ./data]
"""
import argparse
import os
import sys
from datetime import datetime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspeedy_amd import Speedy  # noqa: E402
from pyspeedy_amd.callbacks import ModelCheckpoint, XarrayExporter  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--end", default="1980-02-29")
ap.add_argument("--spinup", default="1980-02-01")
ap.add_argument("--out", default="./data")
args = ap.parse_args()

start_date = datetime(1980, 1, 1)  # Simulation start date (datetime object).
end_date = datetime.strptime(args.end, "%Y-%m-%d")  # Simulation end date.
spinup_date = datetime.strptime(args.spinup, "%Y-%m-%d")  # End of spinup period.

# Create an instance of the speedy model.  At this point, the model state is "empty".
model = Speedy(start_date=start_date, end_date=end_date)
# To initialize the model, we need to define its boundary conditions first (the default ones derived from the ERA reanalysis).
model.set_bc()

# A "callback" is an object that performs user defined actions at each time step.
my_exporter = XarrayExporter(
    output_dir=args.out,  # Output directory where the model output will be stored
    interval=36,  # Every how many time steps we will save the output file. 36 -> once per day.
    verbose=True,  # Print progress messages
    variables=None,  # Which variables to output. If none, save the most commonly used variables.
    spinup_date=spinup_date,  # End of spinup period
)
# This one keeps a dataframe with selected variables with different model times ("checkpoints") in its "dataframe" attribute.
model_checkpoints = ModelCheckpoint(interval=36, verbose=True, variables=None, spinup_date=spinup_date)

print("Exported variables:")
print(my_exporter.variables)

# Run the model. We pass the a list of callbacks
model.run(callbacks=[my_exporter, model_checkpoints])
# After the model is run, the model state will keep the last values of the last integration step.

print(model_checkpoints.dataframe)
lon, lat = model["lon"], model["lat"]
t_surface = model["t_grid"][:, :, -1] - 273.15  # [lon, lat, lev], the vertical dimension sorted in decreasing height
ps = model["ps_grid"] / 100
print("SPEEDY Gaussian grid: %d longitudes %.2f .. %.2f, %d latitudes %.3f .. %.3f" % (lon.size, lon[0], lon[-1], lat.size, lat[0], lat[-1]))
print("Temperature [C] at the lowest level: min %.1f  mean %.1f  max %.1f" % (t_surface.min(), t_surface.mean(), t_surface.max()))
print("Pressure [hPa] at the surface:       min %.1f  mean %.1f  max %.1f" % (ps.min(), ps.mean(), ps.max()))
print("model date %s, %d files in %s" % (model.current_date, len([f for f in os.listdir(args.out) if f.endswith(".nc")]), args.out))
