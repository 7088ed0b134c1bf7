"""CPU tier: AddressSanitizer + UndefinedBehaviorSanitizer builds of the host-side C++ of the product (tables.cpp,
surface_host.cpp) and of the CPU oracle (oracle/orc_*.c), each run through a driver that exercises its entry points
(tests/sanitize/).  GPU AddressSanitizer is not available on the pool: device code is covered by the parity tests only."""
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
         os.path.join(orc, "orc_spectral.c"), os.path.join(orc, "orc_physics.c"), os.path.join(orc, "orc_dynamics.c"), "-lm", "-o",
         "oracle_sanitize"], tmp_path)
    assert "oracle sanitize ok" in run([str(tmp_path / "oracle_sanitize")], tmp_path)
