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
