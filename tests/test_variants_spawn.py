"""Kernel variants that are chosen once per process (launch-bounds builds of the column kernel) must give the same bits: a
member's trajectory must not depend on which variant stepped it.  The device code is built with -ffp-contract=on for exactly
that (csrc/Makefile); with the compiler's default the 1- and 2-wave builds of the column kernel differed in 35 of 92 registry
variables after 40 steps.  Each variant runs in a process of its own (tests/dump_state.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _state(tmp_path, tag, **env):
    out = str(tmp_path / (tag + ".npz"))
    run = subprocess.run([sys.executable, os.path.join(HERE, "dump_state.py"), out], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, **env))
    assert run.returncode == 0, run.stderr[-2000:]
    return np.load(out)


@pytest.mark.gpu
def test_launch_bounds_builds_of_the_column_kernel_agree_bitwise(tmp_path):
    a = _state(tmp_path, "w2", PYSPEEDY_AMD_PHYS_WAVES="2")
    b = _state(tmp_path, "w1", PYSPEEDY_AMD_PHYS_WAVES="1")
    c = _state(tmp_path, "split", PYSPEEDY_AMD_SPLIT_DYN="1")
    assert len(a.files) > 80
    for n in a.files:
        assert np.array_equal(a[n], b[n]), ("1-wave build", n)
    # separate dynamics / physics launches write the dynamics' tendencies to memory and read them back: same values
    differing = [n for n in a.files if not np.array_equal(a[n], c[n])]
    assert not differing, ("split launches", differing[:5])


@pytest.mark.gpu
def test_launch_bounds_builds_of_the_fp32_column_kernel_agree_bitwise(tmp_path):
    """cfg 5: the 2- and 3-wave builds of the fp32 kernel are chosen by the size of the launch (physics.hip: physics_waves32),
    so a member's trajectory must not depend on which one stepped it; and keeping the physics-only arrays as fp64 instead of
    fp32 (PYSPEEDY_AMD_PHYS_STORE32=0) must not change a bit either."""
    a = _state(tmp_path, "f32w2", DUMP_STATE_CFG5="1", PYSPEEDY_AMD_PHYS_WAVES32="2")
    b = _state(tmp_path, "f32w3", DUMP_STATE_CFG5="1", PYSPEEDY_AMD_PHYS_WAVES32="3")
    c = _state(tmp_path, "f32w3s64", DUMP_STATE_CFG5="1", PYSPEEDY_AMD_PHYS_WAVES32="3", PYSPEEDY_AMD_PHYS_STORE32="0")
    plain = _state(tmp_path, "f64")
    assert len(a.files) > 80
    for other, what in ((b, "3-wave build"), (c, "fp64 storage")):
        differing = [n for n in a.files if not np.array_equal(a[n], other[n])]
        assert not differing, (what, differing[:5])
    assert not np.array_equal(a["t"], plain["t"])  # (the switch did switch something)
