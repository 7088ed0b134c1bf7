"""A plain-C host of the outer boundary (examples/c_host.c, include/pyspeedy_amd_driver.h).

CPU tier: the two headers are C (gcc -std=c99 -Wall -Wextra -pedantic -Werror), the example links against the library and,
without a device, stops with the library's message -- there is no CPU fallback.
GPU tier: the program steps four independent containers for one simulated day, once from ONE host thread and once from TWO
threads that each own two containers (the reference's parallel_step is `!f2py threadsafe`; the library gives its lock up while
it waits for the GPU): member 0 equals the reference-generated golden export, and the two runs are bitwise identical."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("orog", "lsm", "alb", "vegh", "vegl", "stl", "snowd", "swl1", "swl2", "swl3", "sst", "icec")
needs_gcc = pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")


def build(tmp_path, name="c_host"):
    libdir = os.path.join(ROOT, "pyspeedy_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", name + ".c"), "-L" + libdir, "-lpyspeedy_amd", "-lpthread",
                    "-Wl,-rpath," + libdir, "-o", name], cwd=tmp_path, check=True, capture_output=True, text=True)
    return str(tmp_path / name)


@needs_gcc
def test_the_headers_are_plain_c_and_the_host_fails_loudly_without_a_device(tmp_path, hip_lib):
    exe = build(tmp_path)
    build(tmp_path, "c_ensemble_host")  # (the host of the library's extensions: plain C too)
    if os.path.exists("/dev/kfd"):
        pytest.skip("the rest needs a machine without a GPU")
    run = subprocess.run([exe, os.devnull, "out.bin", "1"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert run.returncode == 1 and "no HIP device" in run.stderr and "modelstate_init" in run.stderr


@needs_gcc
@pytest.mark.gpu
def test_c_host_steps_four_containers_from_one_and_from_two_threads(tmp_path, hip_lib):
    import pyspeedy_amd
    exe = build(tmp_path)
    with np.load(pyspeedy_amd.example_bc_file()) as bc, open(tmp_path / "bc.bin", "wb") as fh:
        for n in NAMES:
            fh.write(np.asarray(bc[n], dtype=np.float64).tobytes(order="F"))
    outs = []
    for threads in (1, 2):
        run = subprocess.run([exe, "bc.bin", "out%d.bin" % threads, "36", "4", str(threads)], cwd=tmp_path, capture_output=True,
                             text=True, timeout=600)
        assert run.returncode == 0, run.stdout + run.stderr
        assert "model date 1982-01-02 00:00" in run.stdout, run.stdout
        assert ("members in the first device model %d" % (4 // threads)) in run.stdout, run.stdout
        outs.append(np.fromfile(tmp_path / ("out%d.bin" % threads), dtype=np.float64).reshape((4, 96 * 48 * 8)))
    ref = np.load(os.path.join(ROOT, "tests", "golden", "export.npz"))["d1_t_grid"]
    t0 = outs[0][0].reshape((96, 48, 8), order="F")
    assert np.abs(t0 - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(outs[0][3] - outs[0][0]).max() > 1e-3  # member 3: SST + 0.75 K
    assert np.array_equal(outs[0], outs[1])  # the same trajectories whether one thread steps all four or two threads two each
    # ... and through the library's extensions (examples/c_ensemble_host.c): containers batched from the start, the boundary file
    # into container 0 only and handed on device to device, one initialisation per device model, the overlapped time loop
    ens = build(tmp_path, "c_ensemble_host")
    run = subprocess.run([ens, "bc.bin", "out_ens.bin", "36", "4"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "model date 1982-01-02 00:00 members in the first device model 4" in run.stdout, run.stdout
    assert "0 peer copies, 3 local copies" in run.stdout or "GPUs reached collectively" in run.stdout, run.stdout
    assert np.array_equal(np.fromfile(tmp_path / "out_ens.bin", dtype=np.float64).reshape((4, 96 * 48 * 8)), outs[0])
