"""GPU tier: the device-resident model step (spd_model_step_dynamics = time_stepping.f90 `step`) against the reference's
own do_single_step outputs (tests/golden/step.npz) for an ensemble of identical and of perturbed members.
Tolerance: scaled max error <= 1e-12 after one step, <= 2e-12 after two (fp64)."""
import numpy as np
import pytest
import torch

from test_step_oracle import DELT, STEP_2D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir + "/step.npz")


def load_initial(model, gold, member=-1):
    for n in ("vor", "div", "t", "ps", "phis") + STEP_2D:
        model.set(n, gold["s0_" + n], member)
    model.set("tr", gold["s0_tr"], member)
    model.set("tcorh", gold["tab_tcorh"], member)
    model.set("qcorh", gold["tab_qcorh"], member)
    model.set_co2(float(gold["air_absortivity_co2"]))


def err(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)


def test_two_steps_against_reference(spectral, gold):
    from pyspeedy_amd.model import EnsembleModel
    M = 3
    model = EnsembleModel(spectral, M)
    load_initial(model, gold)
    model.set_time_step(2 * DELT)
    model.step_dynamics(2, 2, 2 * DELT, True)
    for member in (0, M - 1):
        for n in ("vor", "div", "t", "tr", "ps", "olr", "precnv", "ssrd", "tsr"):
            e = err(model.get(n, member), gold["s1_" + n])
            assert e <= 1e-12, (n, member, e)
    codes, diag = model.check(2, with_diag=True)
    assert (codes == 0).all() and 200 < diag[0, 2, 0] < 230
    for n in ("sst_am", "land_temp", "soil_avail_water", "snowc", "alb_land", "alb_sea", "alb_surface"):
        model.set(n, gold["s1_" + n])
    model.step_dynamics(2, 2, 2 * DELT, False)
    for n in ("vor", "div", "t", "tr", "ps", "olr", "precnv"):
        e = err(model.get(n, 1), gold["s2_" + n])
        assert e <= 2e-12, (n, e)
    model.close()


def test_members_are_independent(spectral, gold, oracle):
    """A perturbed member evolves like the oracle says; its neighbours are untouched (bitwise equal to each other)."""
    from pyspeedy_amd.model import EnsembleModel
    model = EnsembleModel(spectral, 4)
    load_initial(model, gold)
    rng = np.random.default_rng(5)
    t = gold["s0_t"].copy()
    t[:, :, :, 0] *= (1.0 + 1e-4 * rng.standard_normal((31, 32, 8)))
    t[0, :, :, :] = t[0, :, :, :].real
    model.set("t", t, member=2)
    model.set_time_step(2 * DELT)
    model.step_dynamics(2, 2, 2 * DELT, True)
    arr = {n: gold["s0_" + n] for n in ("vor", "div", "tr", "ps", "phis") + STEP_2D}
    arr.update(t=t, tcorh=gold["tab_tcorh"], qcorh=gold["tab_qcorh"])
    st = oracle.ModelState(arr, True, float(gold["air_absortivity_co2"]))
    oracle.step(st, oracle.dyn_tables(2 * DELT), 2, 2, 2 * DELT)
    for n in ("vor", "div", "t", "tr", "ps"):
        assert err(model.get(n, 2), st.a[n]) <= 1e-12, n
        assert np.array_equal(model.get(n, 0), model.get(n, 3)), n
        assert not np.array_equal(model.get(n, 0), model.get(n, 2)), n
    model.close()


def test_diagnostics_error_code(spectral, gold):
    """pyspeedy/tests/test_speedy.py:117-128: zero temperature -> error code -2, for that member only."""
    from pyspeedy_amd.model import EnsembleModel
    model = EnsembleModel(spectral, 2)
    load_initial(model, gold)
    model.set("t", np.zeros((31, 32, 8, 2), dtype=np.complex128), member=1)
    codes = model.check(1)
    assert codes.tolist() == [0, -2]
    model.close()


def test_registry_errors(spectral):
    from pyspeedy_amd import SpeedyHipError
    from pyspeedy_amd.model import EnsembleModel
    model = EnsembleModel(spectral, 1)
    with pytest.raises(ValueError):
        model.set("vor", np.zeros((31, 32, 8), dtype=np.complex128))
    with pytest.raises(SpeedyHipError):
        model.step_dynamics(2, 2, 2 * DELT, True)  # set_time_step not called yet
    with pytest.raises(SpeedyHipError):
        model.set("vor", np.zeros((31, 32, 8, 2), dtype=np.complex128), member=5)
    model.close()
