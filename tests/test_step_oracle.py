"""CPU tier: oracle restatement of the dynamics / implicit solver / time stepping against a full reference model step
(tests/golden/step.npz: state of the example_bc run before step 42 and after steps 42 and 43).  Bitwise."""
import numpy as np
import pytest

DELT = 86400.0 / 36
STEP_2D = ("fmask_land", "phis0", "forog", "sst_am", "alb_land", "alb_sea", "snowc", "land_temp", "soil_avail_water",
           "flux_solar_in", "flux_ozone_upper", "flux_ozone_lower", "zenit_correction", "stratospheric_correction",
           "alb_surface")


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir + "/step.npz")


def test_dt_dependent_tables_bitwise(oracle, gold):
    d = oracle.dyn_tables(2 * DELT)
    for k in ("dmp", "dmpd", "dmps", "dmp1", "dmp1d", "dmp1s", "tcorv", "qcorv", "tref", "tref2", "tref3", "dhsx", "xc",
              "xd", "xj", "elz"):
        assert np.array_equal(oracle.dyn_table(d, k), gold["tab_" + k]), k


def initial_state(oracle, gold):
    arr = {n: gold["s0_" + n] for n in ("vor", "div", "t", "tr", "ps", "phis") + STEP_2D}
    arr["tcorh"], arr["qcorh"] = gold["tab_tcorh"], gold["tab_qcorh"]
    return oracle.ModelState(arr, True, float(gold["air_absortivity_co2"]))


def test_two_model_steps_bitwise(oracle, gold):
    st = initial_state(oracle, gold)
    d = oracle.dyn_tables(2 * DELT)
    oracle.step(st, d, 2, 2, 2 * DELT)  # step 42: mod(42, 3) == 0 -> shortwave step
    for n in ("vor", "div", "t", "tr", "ps", "olr", "precnv", "ssrd", "tsr"):
        assert np.array_equal(st.a[n], gold["s1_" + n]), n
    rc, diag = oracle.check_diagnostics(st, 2)
    assert rc == 0 and 200 < diag[0, 2] < 230
    for n in ("sst_am", "land_temp", "soil_avail_water", "snowc", "alb_land", "alb_sea", "alb_surface"):
        st.a[n][...] = gold["s1_" + n]  # what the reference's per-step coupler left for the next step
    st.set_shortwave(False)
    oracle.step(st, d, 2, 2, 2 * DELT)  # step 43 reuses the radiation state persisted by step 42
    for n in ("vor", "div", "t", "tr", "ps", "olr", "precnv"):
        assert np.array_equal(st.a[n], gold["s2_" + n]), n


def test_diagnostics_flag_out_of_range(oracle, gold):
    """pyspeedy/tests/test_speedy.py:117-128 (test_exceptions): a zeroed temperature must give error code -2."""
    st = initial_state(oracle, gold)
    st.a["t"][...] = 0
    rc, _ = oracle.check_diagnostics(st, 1)
    assert rc == -2
