"""GPU tier, end to end: boundary conditions -> init -> 36 / 108 model steps on the GPU against the reference Fortran's own
run (tests/golden/run.npz), i.e. the scenario of the reference's test_speedy_run (pyspeedy/tests/test_speedy.py:27-50:
start 1982-01-01, default boundary conditions, 1 and 3 days) -- with zero SST anomaly because sst_anomaly.nc is absent
upstream.

Tolerances (scaled max norm, fp64): after init 1e-11, after 36 steps 1e-10, after 108 steps 1e-9.  SURVEY.md section 8c
measured 1.2e-14 (36 steps) between two builds of the reference itself and no error growth during the first 10 days."""
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPEC = ("vor", "div", "t", "tr", "ps")


def err(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)


@pytest.fixture(scope="module")
def run(golden_dir):
    return np.load(golden_dir + "/run.npz")


@pytest.fixture(scope="module")
def bc(golden_dir):
    return np.load(golden_dir + "/../../pyspeedy_amd/data/example_bc.npz")


def compare(model, run, tag, tol, member=0):
    worst = 0.0
    for n in SPEC + ("phis", "tcorh", "qcorh"):
        e = err(model.get(n, member), run[tag + n])
        assert e <= tol, (tag, n, e)
        worst = max(worst, e)
    for n in ("land_temp", "sst_am", "stl_lm", "tice_om", "sst_om", "snowc", "alb_surface", "soil_avail_water", "phis0",
              "forog", "fmask_land", "olr"):
        e = err(model.get(n, member), run[tag + n])
        assert e <= tol, (tag, n, e)
        worst = max(worst, e)
    return worst


def test_one_and_three_day_forecast(spectral, run, bc):
    from pyspeedy_amd.model import EnsembleModel
    model = EnsembleModel(spectral, 2)
    model.set_bc(bc)
    assert model.current_step == 0 and model.current_date == (1982, 1, 1, 0, 0)
    w0 = compare(model, run, "d0_", 1e-11)
    model.run(36)
    assert model.current_step == 36 and model.current_date == (1982, 1, 2, 0, 0)
    assert (model.check(2) == 0).all()
    w1 = compare(model, run, "d1_", 1e-10)
    model.run(72)
    assert model.current_date == (1982, 1, 4, 0, 0)
    w3 = compare(model, run, "d3_", 1e-9, member=1)
    print("scaled max errors: init %.2e, day 1 %.2e, day 3 %.2e" % (w0, w1, w3))
    # the two members saw identical inputs: bitwise identical trajectories (deterministic kernels)
    for n in SPEC:
        assert np.array_equal(model.get(n, 0), model.get(n, 1))
    model.close()


def test_step_without_init_is_refused(spectral):
    """speedy.f90:41-44: error when the state was not initialised."""
    from pyspeedy_amd import SpeedyHipError
    from pyspeedy_amd.model import EnsembleModel
    model = EnsembleModel(spectral, 1)
    with pytest.raises(SpeedyHipError):
        model.run(1)
    model.close()


def test_exceptions_like_reference(spectral, bc):
    """pyspeedy/tests/test_speedy.py:117-128 test_exceptions: zero temperature -> check() reports -2."""
    from pyspeedy_amd.model import EnsembleModel
    model = EnsembleModel(spectral, 1)
    model.set_bc(bc)
    assert model.check(1).tolist() == [0]
    model.set("t", np.zeros((31, 32, 8, 2), dtype=np.complex128))
    assert model.check(1).tolist() == [-2]
    model.close()


def test_members_of_a_large_batch_equal_single_member_runs(spectral, bc):
    """27 members (not a multiple of the 8 XCDs) with different initial perturbations, 15 steps (5 shortwave steps): every
    sampled member is bitwise identical to a one-member model started from the same state -- batching, the descriptor-table
    order and the position inside a launch do not enter the arithmetic."""
    from pyspeedy_amd.model import EnsembleModel
    M = 27
    ens = EnsembleModel(spectral, M)
    ens.set_bc(bc)
    t0 = ens.get("t", 0)
    perturbed = {}
    for i in range(1, M):
        rng = np.random.default_rng(100 + i)
        t = t0 * (1.0 + 1e-4 * rng.standard_normal((31, 32, 8, 1)))
        t[0] = t[0].real
        perturbed[i] = t
        ens.set("t", t, member=i)
    ens.run(15)
    assert (ens.check(2) == 0).all()
    for i in (0, 1, 7, 8, 13, 26):
        single = EnsembleModel(spectral, 1)
        single.set_bc(bc)
        if i:
            single.set("t", perturbed[i])
        single.run(15)
        for n in SPEC + ("phi", "land_temp", "sst_am", "olr", "precnv", "hfluxn", "rad_tau2"):
            assert np.array_equal(ens.get(n, i), single.get(n, 0)), (i, n)
        single.close()
    assert not np.array_equal(ens.get("t", 1), ens.get("t", 2))


def test_thousands_of_members_on_one_gpu(spectral, bc):
    """Sized for the 288 GB of the card: 12 288 members in ONE device model (227 GB; descriptor tables of 946 176 fields per
    launch; the largest array holds 1.8e9 elements) when the card is free, 4096 otherwise -- initialised and stepped together.
    Members at the far end of every array stay bit for bit on the trajectory of a one-member model, perturbed or not, and every
    member passes the check."""
    import torch
    from pyspeedy_amd.model import EnsembleModel
    free = torch.cuda.mem_get_info()[0]
    M = 12288 if free > (250 << 30) else 4096
    if free < 100 << 30:
        pytest.skip("needs 100 GB of free device memory")
    ens = EnsembleModel(spectral, M)
    ens.set_bc(bc)
    reserved, used = ens.memory()
    assert 17.0e6 * M < used <= reserved < 21.0e6 * M
    t0 = ens.get("t", 0)
    rng = np.random.default_rng(7)
    bump = t0 * (1.0 + 1e-4 * rng.standard_normal((31, 32, 8, 1)))
    bump[0] = bump[0].real
    for i in (1900, M - 1):
        ens.set("t", bump, member=i)
    ens.run(6)  # two shortwave steps
    assert (ens.check(2) == 0).all()
    plain, bumped = EnsembleModel(spectral, 1), EnsembleModel(spectral, 1)
    for single, t in ((plain, None), (bumped, bump)):
        single.set_bc(bc)
        if t is not None:
            single.set("t", t)
        single.run(6)
    for i, single in ((0, plain), (1899, plain), (1900, bumped), (M // 2, plain), (M - 2, plain), (M - 1, bumped)):
        for n in SPEC + ("phi", "land_temp", "sst_am", "olr", "precnv", "hfluxn", "rad_tau2", "tt_rsw"):
            assert np.array_equal(ens.get(n, i), single.get(n, 0)), (M, i, n)
    print("%d members in one device model, %.1f GB" % (M, used / 1e9))
    ens.close()
    plain.close()
    bumped.close()


def test_initialisation_with_distinct_boundary_sets(spectral, bc):
    """spd_model_init preprocesses every member's boundary fields on the device.  Six members, two of
    them with their own SST / soil climatologies: every member is bitwise the one-member model initialised from the same fields,
    after the initialisation and after 7 steps."""
    from pyspeedy_amd.model import BC_MAP, EnsembleModel
    M = 6
    fields = {i: {k: np.asarray(bc[k], dtype=np.float64).copy() for k in bc.files} for i in range(M)}
    fields[2]["sst"] += 0.7
    fields[5]["swl1"] *= 0.9
    fields[5]["stl"] += 0.25
    ens = EnsembleModel(spectral, M)
    for state_name, bc_name in BC_MAP:
        ens.set(state_name, fields[0][bc_name], -1)
        for i in (2, 5):
            ens.set(state_name, fields[i][bc_name], i)
    ens.init((1982, 1, 1, 0, 0))
    names = SPEC + ("sst12", "stl12", "soilw12", "fmask_land", "fmask_sea", "rhcapl", "cdsea", "land_temp", "sst_am", "phis0")
    singles = {}
    for i in (0, 2, 3, 5):
        singles[i] = EnsembleModel(spectral, 1)
        singles[i].set_bc(fields[i])
        for n in names:
            assert np.array_equal(ens.get(n, i), singles[i].get(n, 0)), (i, n, "after init")
    assert not np.array_equal(ens.get("sst12", 2), ens.get("sst12", 0)) and np.isfinite(ens.get("sst12", 2)).all()
    ens.run(7)
    for i, single in singles.items():
        single.run(7)
        for n in names + ("olr", "rad_tau2"):
            assert np.array_equal(ens.get(n, i), single.get(n, 0)), (i, n, "after 7 steps")
        single.close()
    ens.close()


def test_time_step_tables_are_shared_per_context_and_survive_a_host_that_keeps_changing_the_step(spectral, bc):
    """The dynamics tables of a time step exist once per context (a model that changes its step switches pointers).  A context
    keeps 64 of them; a model that asks for a 65th gets a private set, rebuilt in place at every change.  Either way the run is
    the same, bit for bit."""
    import pyspeedy_amd
    from pyspeedy_amd.model import DELT, EnsembleModel
    ref = EnsembleModel(spectral, 2)
    ref.set_bc(bc)
    ref.run(7)
    crowded = pyspeedy_amd.ModSpectral(0)  # (a context of its own: the session's keeps its three steps)
    other = EnsembleModel(crowded, 1)
    for k in range(64):
        other.set_time_step(100.0 + k)  # fills the context
    m = EnsembleModel(crowded, 2)
    m.set_bc(bc)                        # delt / 2, delt, 2 delt: three private rebuilds
    m.set_time_step(123.0)              # ... a fourth, and back
    m.set_time_step(2 * DELT)
    other.set_time_step(2 * DELT)       # (two models with private sets of the same step do not share them)
    m.run(7)
    for n in SPEC + ("phi", "olr", "land_temp", "rad_tau2"):
        assert np.array_equal(m.get(n, 1), ref.get(n, 1)), n
    with pytest.raises(Exception):
        m.set_time_step(0.0)
    for model in (m, other, ref):
        model.close()
    crowded.close()


def test_a_deferred_range_check_rides_in_the_next_step_and_sees_the_state_it_was_deferred_on(spectral, bc):
    """spd_model_check_defer: the check of step k has no launch of its own -- it rides in the spectral -> grid launch of step k + 1
    -- and looks at the state as step k left it, whatever happens next: another step (it rides), a host write (it goes out on
    its own first), nothing at all (it goes out when it is collected).  Same codes as check_begin in every case, same state."""
    from pyspeedy_amd.model import EnsembleModel
    a, b = EnsembleModel(spectral, 3), EnsembleModel(spectral, 3)
    for m in (a, b):
        m.set_bc(bc)
        m.run(2)
    hot = a.get("t", 1).copy()
    hot[0, 0, :, :] = 500.0 * np.sqrt(2.0)  # a global-mean temperature of 500 K: outside diagnostics.f90's 180 ... 320 K

    def drive(m, defer):
        begin = m.check_defer if defer else m.check_begin
        codes = []
        m.run(1)
        t0 = begin()              # check of step 3 ...
        m.run(1)                  # ... rides in step 4 (deferred) / is a launch behind step 3 (begun)
        t1 = begin()
        codes.append(m.check_end(t0))
        m.set("t", hot, 1)        # a host write: the put-off check of step 4 goes out first and still sees a healthy member 1
        codes.append(m.check_end(t1))
        m.run(1)                  # step 5 starts from the hot state
        t2 = begin()
        codes.append(m.check_end(t2))  # nothing came to carry it: it goes out here
        m.run(1)
        t3 = begin()
        m.run(2)                  # a call of several steps: the put-off check rides in its first step (one member group here)
        codes.append(m.check_end(t3))
        return [c.tolist() for c in codes]

    begun, deferred = drive(a, False), drive(b, True)
    assert begun == deferred == [[0, 0, 0], [0, 0, 0], [0, -2, 0], [0, -2, 0]], (begun, deferred)
    assert a.check_counts() == (4, 0) and b.check_counts() == (2, 2), (a.check_counts(), b.check_counts())
    for n in SPEC + ("phi", "olr"):
        assert np.array_equal(a.get(n, 0), b.get(n, 0)) and np.array_equal(a.get(n, 2), b.get(n, 2)), n
    for m in (a, b):
        m.close()


def _through_a_file(snapshot):
    with tempfile.TemporaryDirectory() as tmp:  # as a checkpoint would travel
        np.savez(os.path.join(tmp, "ckpt.npz"), **snapshot)
        return dict(np.load(os.path.join(tmp, "ckpt.npz")))


def _assert_same_state(resumed, ref):
    a, b = resumed.control(), ref.control()
    for name, _ in a._fields_:
        assert getattr(a, name) == getattr(b, name), name
    for n in ref.variables():
        assert np.array_equal(resumed.get(n, 0), ref.get(n, 0)), n


def test_restart_from_the_registry_is_bitwise(spectral, bc):
    """The registry IS the model state: copying every registry array of a running model into a fresh one (plus the
    host-side control block) and continuing gives bitwise the same trajectory as the uninterrupted run -- across a day
    boundary (daily forcing) and shortwave / non-shortwave steps."""
    from pyspeedy_amd.model import EnsembleModel
    ref = EnsembleModel(spectral, 1)
    ref.set_bc(bc)
    ref.run(31)
    snapshot = _through_a_file(ref.state_dict(0))
    ref.run(17)  # crosses step 36 (new day: forcing) and several shortwave steps
    resumed = EnsembleModel(spectral, 1)
    resumed.load_state_dict(snapshot)
    resumed.run(17)
    assert resumed.current_step == ref.current_step == 48 and resumed.current_date == ref.current_date
    _assert_same_state(resumed, ref)


def test_restart_after_a_month_boundary_with_sst_anomalies_and_co2_trend(spectral, bc):
    """What a day-0 checkpoint cannot show: the month index (which sst_anom planes the coupler reads) and the untrended CO2
    reference must travel with the checkpoint.  Run from 30 January into February with two months of non-zero anomalies
    and increase_co2 on, checkpoint in February, continue both."""
    from pyspeedy_amd.model import EnsembleModel

    def fresh():
        m = EnsembleModel(spectral, 1)
        m.init_sst_anom(2)
        return m
    ref = fresh()
    ref.set("sst_anom", np.random.default_rng(11).normal(0.0, 0.8, (96, 48, 4)))
    ref.set_flags(True, True, True)
    ref.set_bc(bc, start_date=(1982, 1, 30, 0, 0))
    ref.run(36 * 3 + 7)  # 2 February, 04:40
    c = ref.control()
    assert (c.month, c.day, c.month_idx) == (2, 2, 2) and c.increase_co2 == 1 and c.air_absortivity_co2 != c.ablco2_ref
    snapshot = _through_a_file(ref.state_dict(0))
    ref.run(36 + 5)  # another day boundary: the CO2 trend is applied again, from the reference value
    resumed = EnsembleModel(spectral, 1)
    resumed.load_state_dict(snapshot)
    resumed.run(36 + 5)
    _assert_same_state(resumed, ref)
    assert resumed.co2 == ref.co2
    # and the planes matter: a resume that restarted the month count would read different anomalies
    assert np.abs(ref.get("sstan_am", 0)).max() > 0.1


def test_restart_with_sppt_and_fp32_physics(spectral, bc):
    """The AR(1) pattern and the position of the counter-based generator are part of the checkpoint (cfg 5)."""
    from pyspeedy_amd.model import EnsembleModel
    ref = EnsembleModel(spectral, 1)
    ref.set_bc(bc)
    ref.set_sppt(True, seed=5, first_member_id=3)
    ref.set_physics_precision(True)
    ref.run(10)
    snapshot = _through_a_file(ref.state_dict(0))
    assert "sppt_spec" in snapshot and snapshot["__control__/sppt_step"] == 10
    ref.run(9)
    resumed = EnsembleModel(spectral, 1)
    resumed.load_state_dict(snapshot)
    resumed.run(9)
    _assert_same_state(resumed, ref)
    assert np.abs(ref.get("sppt_pattern", 0)).max() > 0.0


def test_ten_day_forecast(spectral, bc, golden_dir):
    """360 steps against the reference Fortran (tests/golden/run10.npz): still round-off level, no drift.
    Tolerance 1e-9 of each field's max norm (observed ~1e-13); precipitation fields, which are sums of thresholded
    column processes, 1e-6 of their maximum."""
    from pyspeedy_amd.model import EnsembleModel
    g = np.load(golden_dir + "/run10.npz")
    model = EnsembleModel(spectral, 1)
    model.set_bc(bc)
    model.run(360)
    assert (model.check(2) == 0).all() and model.current_date == (1982, 1, 11, 0, 0)
    worst = 0.0
    for n in ("vor", "div", "t", "tr", "ps"):
        e = err(model.get(n, 0)[..., 0], g[n])
        worst = max(worst, e)
        assert e <= 1e-9, (n, e)
    for n in ("land_temp", "sst_am", "tice_am", "snowc", "olr", "tsr"):
        e = err(model.get(n, 0), g[n])
        assert e <= 1e-9, (n, e)
    for n in ("precnv", "precls"):
        assert err(model.get(n, 0), g[n]) <= 1e-6, n
    print("10-day scaled max error of the spectral state: %.2e" % worst)


def test_member_groups_on_separate_streams_are_bitwise(spectral, bc, monkeypatch):
    """PYSPEEDY_AMD_CHUNKS: stepping the members in 3 overlapping groups (uneven: 4 + 3 + 3) gives bitwise the same state
    as the single-stream step, including across the daily forcing and with get / check right after an asynchronous run."""
    from pyspeedy_amd.model import EnsembleModel
    states = []
    for chunks in ("1", "3"):
        monkeypatch.setenv("PYSPEEDY_AMD_CHUNKS", chunks)
        model = EnsembleModel(spectral, 10)
        model.set_bc(bc)
        t0 = model.get("t", 0)
        for i in range(1, 10):
            t = t0 * (1.0 + 1e-4 * np.random.default_rng(i).standard_normal((31, 32, 8, 1)))
            t[0] = t[0].real
            model.set("t", t, member=i)
        model.run(40)
        assert (model.check(2) == 0).all()
        states.append({n: [model.get(n, i) for i in (0, 3, 4, 6, 9)] for n in SPEC + ("sst_am", "olr", "tcorh")})
        model.close()
    for n, per_member in states[0].items():
        for a, b in zip(per_member, states[1][n]):
            assert np.array_equal(a, b), n


def test_rounds_of_a_large_ensemble_are_bitwise(spectral, bc):
    """From 128 members up a multi-step call takes the members in rounds of 64 -- a round through ALL steps of the call before the
    next one starts, with the host side of the step (calendar, step counter, geopotential buffer, SPPT counter, CO2) rewound in
    between (option block_members).  Bitwise the state of the call that steps everybody together: 150 members (three uneven
    rounds), 50 steps across a midnight and the January / February boundary, SST anomalies, the CO2 trend -- and, second pass, SPPT
    with the fp32 column physics -- then a second call on top (the rewound state must end where a single pass ends)."""
    from pyspeedy_amd.model import EnsembleModel
    M = 150
    i_ = np.arange(96)[:, None, None]
    ssta = 1.5 * np.sin(2 * np.pi * i_ / 96 + 0.7 * np.arange(4)[None, None, :]) * np.ones((1, 48, 1)) - 0.4
    for cfg5 in (False, True):
        states = []
        for block in (0, 32):
            model = EnsembleModel(spectral, M)
            model.init_sst_anom(2)
            model.set("sst_anom", ssta)
            model.set_flags(increase_co2=True)
            model.set_bc(bc, start_date=(1982, 1, 31, 8, 0))
            model.set_option("block_members", block)
            assert model.config()["block_members"] == block
            t0 = model.get("t", 0)
            for i in (1, 63, 64, 100, M - 1):
                t = t0 * (1.0 + 1e-4 * np.random.default_rng(i).standard_normal((31, 32, 8, 1)))
                t[0] = t[0].real
                model.set("t", t, member=i)
            if cfg5:
                model.set_sppt(True, seed=11)
                model.set_physics_precision(True)
            model.run(50)
            model.run(7)
            assert model.current_step == 57 and model.current_date == (1982, 2, 1, 22, 0)
            assert (model.check(2) == 0).all()
            states.append({n: [model.get(n, i) for i in (0, 1, 49, 50, 63, 64, 100, M - 1)] for n in model.variables()
                           if n not in ("lon", "lat", "lev")})
            c = model.control()
            states[-1]["__control__"] = [np.array([float(getattr(c, name)) for name, _ in c._fields_])]
            assert model.config()["rounds"] == (3 if block else 1)
            model.close()
        for n, per_member in states[0].items():
            for a, b in zip(per_member, states[1][n]):
                assert np.array_equal(a, b), (cfg5, n)


def test_multi_step_calls_leave_the_same_state_as_single_steps(spectral, bc, monkeypatch):
    """Inside a multi-step spd_model_step call only the last step stores the physics outputs no later kernel reads, and the
    coupler re-uses the day's interpolated climatologies: every registry variable after run(n) equals, bit for bit, the
    one-step-per-call run (which stores everything every step) and the run with PYSPEEDY_AMD_DIAG_EVERY_STEP=1."""
    from pyspeedy_amd.model import EnsembleModel
    states = []
    for mode in ("single", "multi", "every"):
        if mode == "every":
            monkeypatch.setenv("PYSPEEDY_AMD_DIAG_EVERY_STEP", "1")
        model = EnsembleModel(spectral, 2)
        model.set_bc(bc)
        if mode == "single":
            for _ in range(41):
                model.run(1)
        else:
            model.run(30)
            model.run(11)  # ends on a non-shortwave step after a day boundary
        states.append({n: model.get(n, 1) for n in model.variables()})
        model.close()
    for n in states[0]:
        assert np.array_equal(states[0][n], states[1][n]), n
        assert np.array_equal(states[0][n], states[2][n]), n
    assert np.abs(states[0]["olr"]).max() > 100.0 and np.abs(states[0]["rad_st4a"]).max() > 0.0


def test_pruning_the_unused_transforms_changes_nothing(spectral, bc, monkeypatch):
    """PYSPEEDY_AMD_PRUNE_DEAD=1 drops the 14 inverse transforms per member-step whose results nothing reads (u, v above the
    lowest level at the physics' time level): every registry variable stays bitwise identical."""
    from pyspeedy_amd.model import SHAPES, EnsembleModel
    states = []
    for flag in ("0", "1"):
        monkeypatch.setenv("PYSPEEDY_AMD_PRUNE_DEAD", flag)
        model = EnsembleModel(spectral, 3)
        model.set_bc(bc)
        model.run(40)
        states.append({n: model.get(n, 2) for n in SHAPES})
        model.close()
    for n in SHAPES:
        assert np.array_equal(states[0][n], states[1][n]), n


def test_both_forms_of_the_spectral_step_kernel_are_bitwise(spectral, bc, monkeypatch):
    """spectral_step_kernel<..., EARLY> (all loads up front; small launches) and the form that loads where the values are
    needed (large launches) run the same arithmetic: forcing either one leaves every registry variable bitwise identical, with
    and without the geopotential of the next step folded into the kernel."""
    from pyspeedy_amd.model import SHAPES, EnsembleModel
    for fold in ("0", "1"):
        monkeypatch.setenv("PYSPEEDY_AMD_FOLD_GEO", fold)
        states = []
        for early in ("0", "1"):
            monkeypatch.setenv("PYSPEEDY_AMD_SPECTRAL_EARLY", early)
            model = EnsembleModel(spectral, 3)
            model.set_bc(bc)
            model.run(40)
            states.append({n: model.get(n, 1) for n in SHAPES})
            model.close()
        for n in SHAPES:
            assert np.array_equal(states[0][n], states[1][n]), (fold, n)


def test_launch_plan_options_by_name(spectral, bc):
    """spd_model_set_option: the switches the environment variables set at creation can be flipped on a live model; the state
    a step leaves behind does not depend on them; unknown names and values outside the list are refused."""
    from pyspeedy_amd.model import SHAPES, EnsembleModel
    a, b = EnsembleModel(spectral, 2), EnsembleModel(spectral, 2)
    for m in (a, b):
        m.set_bc(bc)
    b.set_option("diag_every_step", 1)
    b.set_option("coupler_in_spectral", 0)
    b.set_option("spectral_early", 0)
    cfg = b.config()
    assert cfg["diag_every_step"] and not cfg["coupler_in_spectral"] and a.config()["coupler_in_spectral"]
    a.run(12)
    b.run(5)
    b.set_option("coupler_in_spectral", 1)
    b.set_option("spectral_early", -1)
    b.run(7)
    for n in SHAPES:
        assert np.array_equal(a.get(n, 1), b.get(n, 1)), n
    for name, value in (("no_such_switch", 1), ("diag_every_step", 2), ("spectral_early", 5)):
        with pytest.raises(ValueError):
            b.set_option(name, value)
    a.close()
    b.close()


def test_device_views_held_across_steps_stay_valid(spectral, bc):
    """include/pyspeedy_amd.h, spd_model_device_ptr: a view taken once shows the variable after every later step -- also `phi`
    of a small ensemble, which the step otherwise keeps in two alternating buffers (taking its address pins it) --, and a
    WRITE through a view that was taken earlier takes effect once spd_model_invalidate has been called: the model then
    equals one whose temperature was set through spd_model_set."""
    import torch
    from pyspeedy_amd.model import EnsembleModel
    a, b = EnsembleModel(spectral, 2), EnsembleModel(spectral, 2)
    for m in (a, b):
        m.set_bc(bc)
        m.run(3)
    assert a.config()["fold_geo"]
    phi, t = a.device_view("phi"), a.device_view("t")
    assert not a.config()["fold_geo"] and b.config()["fold_geo"]  # pinned: the look-ahead geopotential is off for `a` only
    for _ in range(4):  # the held views follow the model step by step; the pinned model stays bitwise on the other's course
        a.run(1)
        b.run(1)
        torch.cuda.synchronize()
        for member in (0, 1):
            assert np.array_equal(phi[member].cpu().numpy(), b.get("phi", member).transpose(2, 1, 0)), "phi"
            assert np.array_equal(t[member].cpu().numpy(), b.get("t", member).transpose(3, 2, 1, 0)), "t"
    # a write through the view taken seven steps ago + invalidate == spd_model_set (which invalidates by itself)
    new_t = b.get("t", 1) * (1.0 + 1e-3)
    b.set("t", new_t, member=1)
    t[1].copy_(torch.from_numpy(np.ascontiguousarray(new_t.transpose(3, 2, 1, 0))).to(t.device))
    a.invalidate()
    a.run(5)
    b.run(5)
    for n in SPEC + ("phi", "sst_am", "olr"):
        for member in (0, 1):
            assert np.array_equal(a.get(n, member), b.get(n, member)), n
    a.close()
    b.close()


def test_the_default_plan_overlaps_member_groups_and_profiling_is_serial(spectral, bc):
    """spd_model_create: member groups on separate streams from 20 members up (3 groups at 33 members, 2 at 64), one group
    below; spd_model_set_option
    ("member_groups") changes it on a live model; the states are bitwise the same whatever the grouping; while
    spd_model_profile is on the step is issued as one group (per-kernel durations must be the kernel's own)."""
    from pyspeedy_amd.model import EnsembleModel
    assert "PYSPEEDY_AMD_CHUNKS" not in os.environ
    small = EnsembleModel(spectral, 8)
    assert small.config()["chunks"] == 1
    small.close()
    a, b = EnsembleModel(spectral, 33), EnsembleModel(spectral, 33)
    assert a.config()["chunks"] == 3
    b.set_option("member_groups", 1)
    assert b.config()["chunks"] == 1
    for m in (a, b):
        m.set_bc(bc)
        t0 = m.get("t", 0)
        for i in (1, 16, 17, 32):
            t = t0 * (1.0 + 1e-4 * np.random.default_rng(i).standard_normal((31, 32, 8, 1)))
            t[0] = t[0].real
            m.set("t", t, member=i)
    a.run(38)
    b.run(38)
    a.profile(2)  # (serial plan for the profiled steps: the two models keep agreeing)
    a.run(3)
    prof = a.profile_read_kernels()
    a.profile(0)
    assert prof["spec2grid"][2] == 3 and prof["spec2grid"][3] == 77 * 33  # ONE launch per step over all members
    b.run(3)
    assert (a.check(2) == 0).all()
    for n in SPEC + ("sst_am", "olr", "land_temp"):
        for member in (0, 16, 17, 32):
            assert np.array_equal(a.get(n, member), b.get(n, member)), n
    with pytest.raises(ValueError):
        a.set_option("member_groups", 5)
    a.close()
    b.close()


def test_export_pack_entry_point(spectral, bc):
    """spd_model_export_pack through the C ABI: a grid variable of a member range as a NetCDF-3 file carries it (float32, big-endian,
    levels bottom-up) against numpy on the values spd_model_get returns; what it refuses."""
    import ctypes as C
    import torch
    from pyspeedy_amd.model import EnsembleModel
    model = EnsembleModel(spectral, 5)
    model.set_bc(bc)
    model.run(4)
    model.spectral2grid()
    L = model._lib
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, levels in (("t_grid", 8), ("ps_grid", 1), ("olr", 1), ("tt_rsw", 8)):
        n = 3 * levels * 4608
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        rc = L.spd_model_export_pack(model._m, name.encode(), 1, 3, C.c_void_p(out.data_ptr()), n * 4, stream)
        assert rc == 0, L.spd_last_error()
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(">f4").reshape(3, levels, 48, 96)
        for k, member in enumerate((1, 2, 3)):
            ref = model.get(name, member)  # (lon, lat[, lev]) fp64, levels top-down
            ref = ref.transpose(2, 1, 0)[::-1] if levels == 8 else ref.T[None]
            np.testing.assert_array_equal(got[k], ref.astype(np.float32))
    small = torch.zeros(16, dtype=torch.int32, device="cuda")
    assert L.spd_model_export_pack(model._m, b"t_grid", 0, 5, C.c_void_p(small.data_ptr()), 64, stream) == -3  # SPD_E_SIZE
    assert L.spd_model_export_pack(model._m, b"vor", 0, 1, C.c_void_p(small.data_ptr()), 64, stream) == -1     # spectral: refused
    assert L.spd_model_export_pack(model._m, b"no_such", 0, 1, C.c_void_p(small.data_ptr()), 64, stream) == -1
    assert L.spd_model_export_pack(model._m, b"t_grid", 4, 2, C.c_void_p(small.data_ptr()), 64, stream) == -1  # member range
    assert L.spd_model_export_pack(model._m, b"t_grid", 2, 0, C.c_void_p(small.data_ptr()), 0, stream) == 0
    model.close()


def _oracle_first_failure(bc, nmax=40):
    """A member that leaves the accepted range a few steps into its run: SST + 80 K, the global mean of the two lowest levels
    + 30 K.  The step at which it does is what the CPU oracle's whole model says (oracle/orc_model.c)."""
    import oracle as orc
    hot = {k: bc[k] for k in bc.files}
    hot["sst"] = bc["sst"] + 80.0
    cpu = orc.Model()
    cpu.set_bc(hot)
    assert cpu.init(1982, 1, 1) == 0
    t = cpu.get("t").copy()
    t[0, 0, 6:8, :] += 30.0 * np.sqrt(2.0)
    cpu.set("t", t)
    for k in range(nmax):
        if cpu.step() != 0:
            return k
    raise AssertionError("the perturbed oracle member never left the accepted range")


@pytest.mark.parametrize("plan", ["one_group", "two_groups", "rounds"])
def test_checked_multi_step_call_records_every_steps_range_check(spectral, bc, plan):
    """spd_model_step_checked_begin / _end: k steps as ONE device call, the range check of every step recorded by the device
    (the check of step i rides in the spectral -> grid launch of step i + 1, the last one is a launch of its own).  The state is
    bitwise the state of k single steps; a member that leaves the accepted range in the MIDDLE of the call is reported with the
    step at which the CPU oracle's whole model reports it, its last accepted step counter and date; the others with -1 -- in the
    serial plan, with two member groups on streams of their own, and with the members taken in rounds."""
    from pyspeedy_amd.model import EnsembleModel
    M, K, bad = 8, 30, 5
    f = _oracle_first_failure(bc)
    assert 0 < f < K - 1, f
    hot = {k: bc[k] for k in bc.files}
    hot["sst"] = bc["sst"] + 80.0
    states = []
    for checked in (False, True):
        model = EnsembleModel(spectral, M)
        model.set_bc(bc)
        # (member `bad` gets boundary fields of its own and everybody is initialised again)
        from pyspeedy_amd.model import BC_MAP
        for state_name, file_name in BC_MAP:
            model.set(state_name, np.asarray(hot[file_name], dtype=np.float64), member=bad)
        model.init((1982, 1, 1, 0, 0))
        t = model.get("t", bad)
        t[0, 0, 6:8, :] += 30.0 * np.sqrt(2.0)
        model.set("t", t, member=bad)
        if plan == "one_group":
            model.set_option("member_groups", 1)
        else:
            model.set_option("member_groups", 2)
            model.set_option("block_members", 1 if plan == "rounds" else 0)
        if checked:
            alone0, rode0 = model.check_counts()
            failed, accepted = model.run_checked(K)
            alone1, rode1 = model.check_counts()
            assert model.config()["rounds"] == (4 if plan == "rounds" else 1)
            expect = np.full(M, -1)
            expect[bad] = f
            assert (failed == expect).all(), (failed, f)
            groups = 1 if plan == "one_group" else 2
            rounds = 4 if plan == "rounds" else 1
            assert rode1 - rode0 == (K - 1) * groups * rounds and alone1 - alone0 == groups * rounds
            for i in range(M):
                steps = f if i == bad else K
                total_minutes = 40 * steps
                assert accepted[i, 0] == steps
                assert tuple(accepted[i, 1:6]) == (1982, 1, 1 + total_minutes // 1440, (total_minutes % 1440) // 60, total_minutes % 60)
            # a second call on top: the codes of the new call are its own
            failed2, _ = model.run_checked(3)
            assert failed2[bad] in (-1, 0, 1, 2) and (np.delete(failed2, bad) == -1).all()
            with pytest.raises(Exception):
                model.run_checked(5000)
        else:
            codes = []
            for _ in range(K):
                model.run(1)
                codes.append(model.check(2))
            codes = np.array(codes)
            assert (codes[:, bad][:f] == 0).all() and codes[f, bad] == -2
            assert (np.delete(codes, bad, axis=1) == 0).all()
            model.run(3)
        states.append({n: [model.get(n, i) for i in range(M) if i != bad] for n in model.variables() if n not in ("lon", "lat", "lev")})
        model.close()
    for n, per_member in states[0].items():
        for a, b in zip(per_member, states[1][n]):
            assert np.array_equal(a, b), n


def test_a_device_error_in_the_middle_of_a_step_leaves_the_model_unusable_until_it_is_initialised_again(spectral, bc):
    """A launch of step k fails after another member group's launches of the same step went out (fault injection: option
    fail_launch_after): whatever the plan -- two groups and ONE round here, the case the rounds-only rule of round 5 missed --
    the model refuses to be stepped, read or transformed until spd_model_init has rebuilt its state, and then runs as a new one."""
    from pyspeedy_amd._lib import SpeedyHipError
    from pyspeedy_amd.model import EnsembleModel
    model = EnsembleModel(spectral, 4)
    model.set_bc(bc)
    model.set_option("member_groups", 2)
    model.set_option("block_members", 0)
    model.run(2)
    model.set_option("fail_launch_after", 5)  # the 6th per-group launch sequence: group 1 of the third step of the call
    with pytest.raises(SpeedyHipError):
        model.run(6)
    for call in (lambda: model.run(1), lambda: model.run_checked(2), lambda: model.check(2), lambda: model.spectral2grid(),
                 lambda: model.get("t", 0)):
        with pytest.raises(SpeedyHipError, match="initialised again"):
            call()
    with pytest.raises(KeyError):
        model.device_view("t")
    model.init((1982, 1, 1, 0, 0))
    model.run(5)
    fresh = EnsembleModel(spectral, 4)
    fresh.set_bc(bc)
    fresh.set_option("member_groups", 2)
    fresh.set_option("block_members", 0)
    fresh.run(5)
    for n in SPEC:
        assert np.array_equal(model.get(n, 3), fresh.get(n, 3)), n
    model.close()
    fresh.close()
