"""CPU tier: sanitizer builds of the host-side C++ of the product and of the CPU oracle, each run through a program that
exercises its entry points (tests/sanitize/).
* AddressSanitizer + UndefinedBehaviorSanitizer: tables.cpp, surface_host.cpp, the oracle (oracle/orc_*.c);
* AddressSanitizer + UndefinedBehaviorSanitizer (with leak detection) AND ThreadSanitizer: the outer boundary's host logic,
  csrc/driver.cpp -- containers, placement, gathering / splitting, the kept plan, pending-step tokens and their error paths --
  built unchanged against a stub of driver_backend.hpp and of the spd_model_* functions (tests/sanitize/driver_stub.cpp): four
  host threads step disjoint container sets through spd_parallel_step and _begin / _end, regroup, replace containers, a fifth
  keeps the container tables busy, a container is closed while its owner waits for it.
GPU AddressSanitizer is not available on the pool: device code is covered by the parity tests only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def run(cmd, cwd):
    p = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600, env=ENV)
    assert p.returncode == 0, "%s\n%s\n%s" % (" ".join(cmd), p.stdout[-3000:], p.stderr[-6000:])
    return p.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_tables_and_surface_preprocessing_under_asan_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "pyspeedy_amd", "csrc")
    run(["g++", "-std=c++17", "-ffp-contract=off", *SAN, "-I" + csrc, os.path.join(ROOT, "tests", "sanitize", "host_sanitize.cpp"),
         os.path.join(csrc, "tables.cpp"), os.path.join(csrc, "surface_host.cpp"), "-o", "host_sanitize"], tmp_path)
    assert "host sanitize ok" in run([str(tmp_path / "host_sanitize")], tmp_path)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_under_asan_ubsan(tmp_path):
    orc = os.path.join(ROOT, "oracle")
    run(["gcc", "-std=c11", "-ffp-contract=off", *SAN, "-I" + orc, os.path.join(ROOT, "tests", "sanitize", "oracle_sanitize.c"),
         os.path.join(orc, "orc_spectral.c"), os.path.join(orc, "orc_physics.c"), os.path.join(orc, "orc_dynamics.c"),
         os.path.join(orc, "orc_surface.c"), os.path.join(orc, "orc_model.c"), "-lm", "-o",
         "oracle_sanitize"], tmp_path)
    assert "oracle sanitize ok" in run([str(tmp_path / "oracle_sanitize")], tmp_path)


DRIVER_SOURCES = [os.path.join(ROOT, "tests", "sanitize", "driver_sanitize.cpp"), os.path.join(ROOT, "tests", "sanitize", "driver_stub.cpp"),
                  os.path.join(ROOT, "pyspeedy_amd", "csrc", "driver.cpp")]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_driver_logic_under_asan_ubsan(tmp_path):
    run(["g++", "-std=c++17", "-pthread", "-Wall", "-Werror", *SAN, *DRIVER_SOURCES, "-o", "driver_asan"], tmp_path)
    p = subprocess.run([str(tmp_path / "driver_asan"), "all"], cwd=tmp_path, capture_output=True, text=True, timeout=600,
                       env=dict(ENV, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert p.returncode == 0 and "driver sanitize ok" in p.stdout, p.stdout[-3000:] + p.stderr[-6000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_driver_logic_under_tsan(tmp_path):
    run(["g++", "-std=c++17", "-pthread", "-Wall", "-Werror", "-fsanitize=thread", "-g", "-O1", *DRIVER_SOURCES, "-o", "driver_tsan"],
        tmp_path)
    # the enqueue of a step goes out on one host thread per device (driver.cpp: issue_all): the default (the two pretend devices
    # of the stub: the calling thread and one worker), a worker per device model, and everything from the calling thread
    for mode in ("1", "2", "0"):
        p = subprocess.run([str(tmp_path / "driver_tsan"), "all"], cwd=tmp_path, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1", PYSPEEDY_AMD_ISSUE_THREADS=mode))
        assert p.returncode == 0 and "driver sanitize ok" in p.stdout and "ThreadSanitizer" not in p.stderr, \
            "PYSPEEDY_AMD_ISSUE_THREADS=" + mode + "\n" + p.stdout[-3000:] + p.stderr[-8000:]
