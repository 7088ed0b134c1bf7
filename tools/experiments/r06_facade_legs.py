"""bench.py's facade leg on its own: SpeedyEns(64).run(), Speedy().run(), with and without daily files.

    python tools/experiments/r06_facade_legs.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

out = bench.facade_leg()
print(json.dumps(out, indent=1))
