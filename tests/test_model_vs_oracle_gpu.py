"""GPU tier: the whole model step of DIFFERENT states against the CPU oracle's whole model (oracle/orc_model.c, bit for bit the
reference: tests/test_model_oracle.py) on the same seeded inputs -- six containers stepped through `spd_step`, each with its own
temperature perturbation (numpy default_rng(seed = member)), its own start date (through the coupling flags and the CO2 trend for two
of them) and 12 model steps (four shortwave steps, a midnight coupling for the member started at 20:00).  The goldens pin
unperturbed trajectories; this pins perturbed ones, at other dates and flag settings: every member against an oracle run of its own
(observed 6.4e-14).  Tolerance 1e-11 of each field's
max norm (12 steps, fp64; one step is held to 1e-12 in tests/test_step_gpu.py)."""
from datetime import datetime

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPEC = ("vor", "div", "t", "tr", "ps")
SURF = ("land_temp", "sst_am", "tice_am", "sice_am", "snowc", "alb_surface", "olr", "precnv", "precls", "tsr", "ssrd", "hfluxn", "shf")


def perturbation(seed):
    rng = np.random.default_rng(seed)
    f = 1.0 + 2e-4 * rng.standard_normal((31, 32, 8, 1))
    f[0] = 1.0  # (the zonal-mean coefficients keep a zero imaginary part)
    return f


def test_six_different_members_against_an_oracle_run_each(oracle, golden_dir):
    from pyspeedy_amd.speedy import Speedy
    bc = np.load(golden_dir + "/../../pyspeedy_amd/data/example_bc.npz")
    cases = [  # (start, flags)
        (datetime(1982, 1, 1), {}),
        (datetime(1982, 1, 1), {}),
        (datetime(1982, 6, 30, 20, 0), {}),
        (datetime(1982, 6, 30, 20, 0), {"increase_co2": True}),
        (datetime(1980, 2, 29), {"land_coupling_flag": False}),
        (datetime(1982, 12, 31, 18, 0), {"increase_co2": True}),
    ]
    worst = 0.0
    for seed, (start, flags) in enumerate(cases):
        end = datetime(start.year + 1, 1, 2)
        gpu = Speedy(start_date=start, end_date=end)
        cpu = oracle.Model(n_months=gpu.n_months)
        cpu.set_bc(bc)
        for k, v in flags.items():
            gpu[k] = v
            cpu.set(k, int(v))
        gpu.set_bc()
        assert cpu.init(start.year, start.month, start.day, start.hour, start.minute) == 0
        if seed:  # member 0 stays on the unperturbed trajectory
            f = perturbation(seed)
            gpu["t"] = gpu["t"] * f
            cpu.set("t", cpu.get("t") * f)
        from pyspeedy_amd import speedy_driver as drv
        for _ in range(12):
            assert drv.step(gpu._state_cnt, gpu._control_cnt) == 0 and cpu.step() == 0
        date, month_idx = drv.get_model_datetime(gpu._control_cnt)
        assert (list(date), month_idx) == cpu.calendar()[:2]
        for name in SPEC + SURF:
            ref, got = cpu.get(name), np.asarray(gpu[name])
            scale = np.abs(ref).max()
            err = np.abs(got - ref.reshape(got.shape)).max() / (scale if scale > 0 else 1.0)
            assert err <= 1e-11, (seed, name, err)
            worst = max(worst, err)
        assert abs(gpu["air_absortivity_co2"] - cpu.get("air_absortivity_co2")) <= 1e-14
    print("six members x 12 steps against the oracle: worst scaled error %.2e" % worst)
