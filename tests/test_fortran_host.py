"""The ISO_C_BINDING interface module (include/pyspeedy_amd_c.f90) and the Fortran host example built on it.

CPU tier: both compile with amdflang (the interface block is valid Fortran and matches the example's calls).
GPU tier: the Fortran program drives a whole 1-day run through the C ABI -- set the boundary fields, init, 36 steps, check,
spectral2grid, get -- and its t_grid / ps_grid equal the reference-generated golden export (tests/golden/export.npz) to 1e-10
of the field's max norm, the tolerance of the 36-step parity runs."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLANG = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
NAMES = ("orog", "lsm", "alb", "vegh", "vegl", "stl", "snowd", "swl1", "swl2", "swl3", "sst", "icec")

needs_flang = pytest.mark.skipif(not os.path.exists(FLANG), reason="amdflang not available")


def compile_module(workdir):
    subprocess.run([FLANG, "-c", os.path.join(ROOT, "include", "pyspeedy_amd_c.f90"), "-o", "pyspeedy_amd_c.o"],
                   cwd=workdir, check=True, capture_output=True, text=True)


@needs_flang
def test_interface_module_and_example_compile(tmp_path):
    compile_module(tmp_path)
    assert (tmp_path / "pyspeedy_amd_c.mod").exists()
    for prog in ("fortran_host", "fortran_ensemble_host"):
        subprocess.run([FLANG, "-c", "-I.", os.path.join(ROOT, "examples", prog + ".f90"), "-o", prog + ".o"],
                       cwd=tmp_path, check=True, capture_output=True, text=True)


@needs_flang
@pytest.mark.gpu
def test_fortran_host_runs_the_model(tmp_path, hip_lib):
    import pyspeedy_amd
    compile_module(tmp_path)
    libdir = os.path.join(ROOT, "pyspeedy_amd")
    subprocess.run([FLANG, "-I.", os.path.join(ROOT, "examples", "fortran_host.f90"), "pyspeedy_amd_c.o", "-L" + libdir,
                    "-lpyspeedy_amd", "-Wl,-rpath," + libdir, "-o", "fortran_host"], cwd=tmp_path, check=True,
                   capture_output=True, text=True)
    with np.load(pyspeedy_amd.example_bc_file()) as bc, open(tmp_path / "bc.bin", "wb") as fh:
        for n in NAMES:
            fh.write(np.asarray(bc[n], dtype=np.float64).tobytes(order="F"))
    run = subprocess.run([str(tmp_path / "fortran_host"), "bc.bin", "out.bin", "36"], cwd=tmp_path, capture_output=True,
                         text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "steps 36" in run.stdout
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float64)
    t_grid = out[:96 * 48 * 8].reshape((96, 48, 8), order="F")
    ps_grid = out[96 * 48 * 8:].reshape((96, 48), order="F")
    gold = np.load(os.path.join(ROOT, "tests", "golden", "export.npz"))
    for got, name in ((t_grid, "d1_t_grid"), (ps_grid, "d1_ps_grid")):
        ref = gold[name]
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), name


@needs_flang
@pytest.mark.gpu
def test_fortran_host_with_the_reference_call_sites_steps_three_containers_as_one_batch(tmp_path, hip_lib):
    """examples/fortran_ensemble_host.f90: modelstate_init x 3, set_<v>, controlparams_init, init, 36 x parallel_step over the
    three independent containers, check, transform_spectral2grid, get_t_grid -- member 1 (unperturbed) equals the reference
    run, member 3 (SST + 0.5 K) differs, and the three containers ended up in ONE device model."""
    import pyspeedy_amd
    compile_module(tmp_path)
    libdir = os.path.join(ROOT, "pyspeedy_amd")
    subprocess.run([FLANG, "-I.", os.path.join(ROOT, "examples", "fortran_ensemble_host.f90"), "pyspeedy_amd_c.o", "-L" + libdir,
                    "-lpyspeedy_amd", "-Wl,-rpath," + libdir, "-o", "fortran_ensemble_host"], cwd=tmp_path, check=True,
                   capture_output=True, text=True)
    with np.load(pyspeedy_amd.example_bc_file()) as bc, open(tmp_path / "bc.bin", "wb") as fh:
        for n in NAMES:
            fh.write(np.asarray(bc[n], dtype=np.float64).tobytes(order="F"))
    run = subprocess.run([str(tmp_path / "fortran_ensemble_host"), "bc.bin", "out.bin", "36"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "members in one device model 3  device models alive 1" in run.stdout, run.stdout
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float64).reshape((2, 96 * 48 * 8))
    t1, t3 = (o.reshape((96, 48, 8), order="F") for o in out)
    ref = np.load(os.path.join(ROOT, "tests", "golden", "export.npz"))["d1_t_grid"]
    assert np.abs(t1 - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(t3 - ref).max() > 1e-3
