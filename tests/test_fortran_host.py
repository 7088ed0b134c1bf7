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


def compile_reference_named_module(workdir):
    subprocess.run([FLANG, "-c", "-I.", os.path.join(ROOT, "include", "speedy_driver_amd.f90"), "-o", "speedy_driver_amd.o"],
                   cwd=workdir, check=True, capture_output=True, text=True)


@needs_flang
def test_interface_module_and_example_compile(tmp_path):
    compile_module(tmp_path)
    assert (tmp_path / "pyspeedy_amd_c.mod").exists()
    for prog in ("fortran_host", "fortran_ensemble_host"):
        subprocess.run([FLANG, "-c", "-I.", os.path.join(ROOT, "examples", prog + ".f90"), "-o", prog + ".o"],
                       cwd=tmp_path, check=True, capture_output=True, text=True)


def test_the_reference_named_module_is_current_and_complete():
    """include/speedy_driver_amd.f90 (module `speedy_driver`, the reference's own procedure names forwarding to the C ABI) is
    what tools/gen_fortran_driver.py generates from the registry, fits gfortran's 132-column free form, and declares every
    procedure of registry/templates/speedy_driver.f90.j2: the fixed ones by name, and get_ / set_ / get_<v>_shape / is_array_
    for every registry variable."""
    import re
    import sys
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fortran_driver.py"), "--check"], check=True)
    text = open(os.path.join(ROOT, "include", "speedy_driver_amd.f90")).read()
    assert max(len(ln) for ln in text.splitlines()) <= 132
    subs = set(re.findall(r"^\s*subroutine\s+(\w+)\s*\(", text, flags=re.M))
    fixed = {"init", "step", "parallel_step", "check", "transform_spectral2grid", "transform_grid2spectral", "apply_grid_filter",
             "controlparams_init", "controlparams_close", "create_datetime", "get_datetime", "close_datetime", "modelstate_init",
             "modelstate_init_sst_anom", "modelstate_close"}
    assert fixed <= subs
    assert {"parallel_step_begin", "parallel_step_end", "modelstate_init_ensemble", "broadcast_boundary"} <= subs  # extensions
    sys.path.insert(0, ROOT)
    import pyspeedy_amd.registry as R
    for name, v in R.REGISTRY.items():
        want = {"get_" + name, "set_" + name, "is_array_" + name} | ({"get_%s_shape" % name} if v.shape is not None else set())
        assert want <= subs, name
    assert "!f2py threadsafe" in text


@needs_flang
def test_reference_named_module_and_its_host_compile(tmp_path):
    compile_module(tmp_path)
    compile_reference_named_module(tmp_path)
    assert (tmp_path / "speedy_driver.mod").exists()
    subprocess.run([FLANG, "-c", "-I.", os.path.join(ROOT, "examples", "fortran_reference_api_host.f90"), "-o", "host.o"],
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


@needs_flang
@pytest.mark.gpu
def test_a_host_written_against_the_references_own_module_runs_unchanged(tmp_path, hip_lib):
    """examples/fortran_reference_api_host.f90 uses `speedy_driver` and nothing but the reference's procedure names
    (modelstate_init, set_orog ..., init, parallel_step, check, transform_spectral2grid, get_t_grid, get_current_step,
    get_sst_anom_shape, is_array_t_grid).  Linked against include/speedy_driver_amd.f90 it steps two members for a day on the
    GPU: member 1 equals the reference-generated golden, member 2 (SST + 0.5 K) differs."""
    import pyspeedy_amd
    compile_module(tmp_path)
    compile_reference_named_module(tmp_path)
    libdir = os.path.join(ROOT, "pyspeedy_amd")
    subprocess.run([FLANG, "-I.", os.path.join(ROOT, "examples", "fortran_reference_api_host.f90"), "speedy_driver_amd.o",
                    "pyspeedy_amd_c.o", "-L" + libdir, "-lpyspeedy_amd", "-Wl,-rpath," + libdir, "-o", "ref_api_host"],
                   cwd=tmp_path, check=True, capture_output=True, text=True)
    with np.load(pyspeedy_amd.example_bc_file()) as bc, open(tmp_path / "bc.bin", "wb") as fh:
        for n in NAMES:
            fh.write(np.asarray(bc[n], dtype=np.float64).tobytes(order="F"))
    run = subprocess.run([str(tmp_path / "ref_api_host"), "bc.bin", "out.bin", "36"], cwd=tmp_path, capture_output=True,
                         text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "steps 36  t_grid is an array T  end year 1982" in run.stdout, run.stdout
    out = np.fromfile(tmp_path / "out.bin", dtype=np.float64).reshape((2, 96 * 48 * 8))
    t1, t2 = (o.reshape((96, 48, 8), order="F") for o in out)
    ref = np.load(os.path.join(ROOT, "tests", "golden", "export.npz"))["d1_t_grid"]
    assert np.abs(t1 - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(t2 - ref).max() > 1e-3
    # the same host with its time loop in the overlapped form (parallel_step_begin / parallel_step_end of the same module):
    # the same bits
    run = subprocess.run([str(tmp_path / "ref_api_host"), "bc.bin", "out2.bin", "36", "overlapped"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "steps 36" in run.stdout
    assert np.array_equal(np.fromfile(tmp_path / "out2.bin", dtype=np.float64), out.ravel())
