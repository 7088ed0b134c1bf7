"""GPU tier: the outer C boundary (include/pyspeedy_amd_driver.h) called the way the reference's Python layer calls its
f2py module -- modelstate_init, set_<v> of the 12 boundary fields, controlparams_init, init, step / parallel_step, check,
transform_spectral2grid, get_<v> -- against the reference-generated goldens, and the batching of independent containers."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BC_MAP = (("orog", "orog"), ("fmask_orig", "lsm"), ("alb0", "alb"), ("veg_high", "vegh"), ("veg_low", "vegl"),
          ("stl12", "stl"), ("snowd12", "snowd"), ("soil_wc_l1", "swl1"), ("soil_wc_l2", "swl2"), ("soil_wc_l3", "swl3"),
          ("sst12", "sst"), ("sea_ice_frac12", "icec"))


class Driver:
    """The `_speedy` function set over the C ABI, for the tests."""

    def __init__(self, lib):
        self.L = lib

    def ok(self, rc):
        assert rc == 0, self.L.spd_last_error()

    def date(self, *ymdhm):
        c = C.c_int64()
        self.ok(self.L.spd_create_datetime(*ymdhm, C.byref(c)))
        return c.value

    def control(self, start, end):
        c = C.c_int64()
        self.ok(self.L.spd_controlparams_init(C.byref(c), self.date(*start), self.date(*end)))
        return c.value

    def state(self):
        c = C.c_int64()
        self.ok(self.L.spd_modelstate_init(C.byref(c)))
        return c.value

    def set(self, cnt, name, value):
        a = np.ascontiguousarray(np.asarray(value).ravel(order="F"))
        self.ok(self.L.spd_set(cnt, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def shape(self, cnt, name):
        shp, nd = (C.c_int32 * 5)(), C.c_int32()
        self.ok(self.L.spd_get_shape(cnt, name.encode(), shp, C.byref(nd)))
        return tuple(shp[:nd.value])

    def get(self, cnt, name, dtype=np.float64):
        shape = self.shape(cnt, name)
        a = np.zeros(int(np.prod(shape)) if shape else 1, dtype=dtype)
        self.ok(self.L.spd_get(cnt, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a.reshape(shape, order="F") if shape else a[0]

    def set_bc(self, cnt, bc, perturb=None):
        for state_name, bc_name in BC_MAP:
            self.set(cnt, state_name, np.asarray(bc[bc_name], dtype=np.float64))
        if perturb is not None:
            self.set(cnt, "sst12", np.asarray(bc["sst"], dtype=np.float64) + perturb)

    def init(self, state, control):
        code = C.c_int32(99)
        self.ok(self.L.spd_init(state, control, C.byref(code)))
        return code.value

    def step(self, state, control):
        code = C.c_int32(99)
        self.ok(self.L.spd_step(state, control, C.byref(code)))
        return code.value

    def parallel_step(self, states, controls):
        n = len(states)
        codes = (C.c_int32 * n)(*([99] * n))
        self.ok(self.L.spd_parallel_step((C.c_int64 * n)(*states), (C.c_int64 * n)(*controls), codes, n))
        return list(codes)

    def stats(self, cnt):
        alive, members = C.c_int32(), C.c_int32()
        self.ok(self.L.spd_driver_stats(cnt, C.byref(alive), C.byref(members)))
        return alive.value, members.value

    def model_date(self, control):
        now, midx = (C.c_int32 * 5)(), C.c_int32()
        self.ok(self.L.spd_controlparams_get_model_datetime(control, now, C.byref(midx)))
        return tuple(now), midx.value

    def close(self, *states):
        for s in states:
            self.ok(self.L.spd_modelstate_close(s))


@pytest.fixture(scope="module")
def drv(spectral, hip_lib):
    return Driver(hip_lib)


@pytest.fixture(scope="module")
def bc():
    import pyspeedy_amd
    with np.load(pyspeedy_amd.example_bc_file()) as z:
        return {k: z[k] for k in z.files}


START, END = (1982, 1, 1, 0, 0), (1982, 1, 4, 0, 0)


def test_one_container_like_the_reference_python_layer(drv, bc, golden_dir):
    """modelstate_init -> set_<v> x 12 -> init -> 36 x step -> transform_spectral2grid -> get_<v>: equals the reference run
    (tests/golden/export.npz, 1 day) at the 36-step tolerance."""
    st, ctl = drv.state(), drv.control(START, END)
    assert drv.shape(st, "sst_anom") == (0, 0, 0)  # not allocated yet (get_sst_anom_shape of the reference returns zeros)
    drv.ok(drv.L.spd_modelstate_init_sst_anom(st, 1))
    assert drv.shape(st, "sst_anom") == (96, 48, 3) and drv.shape(st, "vor") == (31, 32, 8, 2)
    drv.set_bc(st, bc)
    code = C.c_int32(5)
    drv.ok(drv.L.spd_check(st, C.byref(code)))
    assert code.value == -1 and drv.step(st, ctl) == -1  # E_STATE_NOT_INITIALIZED
    assert drv.init(st, ctl) == 0
    for _ in range(36):
        assert drv.step(st, ctl) == 0
    assert drv.model_date(ctl) == ((1982, 1, 2, 0, 0), 1)
    assert drv.get(st, "current_step", np.int32) == 36
    drv.ok(drv.L.spd_transform_spectral2grid(st))
    gold = np.load(os.path.join(golden_dir, "export.npz"))
    for name in ("t_grid", "u_grid", "ps_grid"):
        got, ref = drv.get(st, name), gold["d1_" + name]
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), name
    # tables and scalars through the same name-driven access
    lat = drv.get(st, "lat", np.float32)
    assert lat.dtype == np.float32 and abs(lat[0] + 87.2165) < 1e-3 and abs(lat[-1] - 87.2165) < 1e-3
    assert drv.get(st, "air_absortivity_co2") == 6.0 and drv.get(st, "land_coupling_flag", np.int32) == 1
    v = np.zeros(8)
    assert drv.L.spd_set(st, b"xgeop1", v.ctypes.data_as(C.c_void_p), v.nbytes) < 0  # read-only table
    assert drv.L.spd_get(st, b"olr", v.ctypes.data_as(C.c_void_p), v.nbytes) < 0  # wrong size: "Array shape missmatch"
    drv.close(st)


def test_parallel_step_gathers_independent_containers_into_one_batched_model(drv, bc):
    """Three containers created and initialised one by one (different SSTs) are stepped with parallel_step: after the first
    call they are the members of ONE device model (one set of launches per step) and every member's trajectory is bitwise
    what the same container gives when it is stepped alone with step()."""
    alive0, _ = drv.stats(0)
    pert = [None, 0.3 * np.ones((96, 48, 12)), -0.2 * np.ones((96, 48, 12))]
    batched = [drv.state() for _ in range(3)]
    alone = [drv.state() for _ in range(3)]
    ctl_b = [drv.control(START, END) for _ in range(3)]
    ctl_a = [drv.control(START, END) for _ in range(3)]
    for group, ctls in ((batched, ctl_b), (alone, ctl_a)):
        for s, c, p in zip(group, ctls, pert):
            drv.set_bc(s, bc, p)
            assert drv.init(s, c) == 0
    assert drv.stats(batched[0]) == (alive0 + 6, 1)
    for _ in range(40):  # across a day boundary
        assert drv.parallel_step(batched, ctl_b) == [0, 0, 0]
    assert drv.stats(batched[0]) == (alive0 + 4, 3) and drv.stats(batched[2])[1] == 3
    for s, c in zip(alone, ctl_a):
        for _ in range(40):
            assert drv.step(s, c) == 0
    assert drv.model_date(ctl_b[1]) == drv.model_date(ctl_a[1]) == ((1982, 1, 2, 2, 40), 1)
    for sb, sa in zip(batched, alone):
        for name, dt in (("vor", np.complex128), ("t", np.complex128), ("ps", np.complex128), ("tr", np.complex128)):
            assert np.array_equal(drv.get(sb, name, dt), drv.get(sa, name, dt)), name
        for name in ("olr", "precnv", "land_temp", "sst_am", "hfluxn", "rad_tau2"):
            assert np.array_equal(drv.get(sb, name), drv.get(sa, name)), name
    assert not np.array_equal(drv.get(batched[0], "sst_am"), drv.get(batched[1], "sst_am"))
    # stepping ONE member of the batch on its own takes the batch apart again -- and stays on the same trajectory
    assert drv.step(batched[1], ctl_b[1]) == 0 and drv.step(alone[1], ctl_a[1]) == 0
    assert drv.stats(batched[1])[1] == 1 and drv.stats(batched[0])[1] == 1
    assert np.array_equal(drv.get(batched[1], "t", np.complex128), drv.get(alone[1], "t", np.complex128))
    # members 0 and 2 are one step behind member 1 now: parallel_step over all three batches the two that agree (a partial
    # gather) and steps member 1 as a second device model in the same call -- still on the trajectories of the lone runs
    codes = drv.parallel_step(batched, ctl_b)
    assert codes == [0, 0, 0] and drv.stats(batched[0])[1] == 2 and drv.stats(batched[2])[1] == 2 and drv.stats(batched[1])[1] == 1
    for s, c in ((alone[0], ctl_a[0]), (alone[2], ctl_a[2]), (alone[1], ctl_a[1])):
        assert drv.step(s, c) == 0
    for sb, sa in zip(batched, alone):
        assert np.array_equal(drv.get(sb, "t", np.complex128), drv.get(sa, "t", np.complex128))
        assert np.array_equal(drv.get(sb, "sst_am"), drv.get(sa, "sst_am"))
    drv.close(*batched, *alone)
    assert drv.stats(0)[0] == alive0


def test_ensemble_containers_are_batched_from_the_start(drv, bc):
    n = 4
    cnts = (C.c_int64 * n)()
    drv.ok(drv.L.spd_modelstate_init_ensemble(cnts, n))
    states = list(cnts)
    controls = [drv.control(START, END) for _ in range(n)]
    assert drv.stats(states[0])[1] == n
    for i, (s, c) in enumerate(zip(states, controls)):
        drv.set_bc(s, bc, 0.1 * i * np.ones((96, 48, 12)))
        assert drv.init(s, c) == 0
    single, cs = drv.state(), drv.control(START, END)
    drv.set_bc(single, bc, 0.2 * np.ones((96, 48, 12)))
    assert drv.init(single, cs) == 0
    for _ in range(9):
        assert drv.parallel_step(states, controls) == [0] * n
        assert drv.step(single, cs) == 0
    assert drv.stats(states[0])[1] == n
    assert np.array_equal(drv.get(states[2], "t", np.complex128), drv.get(single, "t", np.complex128))
    # a scalar that differs from the rest of the batch makes its member leave the batch
    one = np.array([1], dtype=np.int32)
    drv.ok(drv.L.spd_set(states[3], b"increase_co2", one.ctypes.data_as(C.c_void_p), 4))
    assert drv.stats(states[3])[1] == 1 and drv.get(states[3], "increase_co2", np.int32) == 1
    assert drv.get(states[0], "increase_co2", np.int32) == 0
    drv.close(single, *states)


def test_a_failing_member_keeps_its_date_and_reports_minus_two(drv, bc):
    """speedy.f90:57-71: the range check runs before advance_date; a member that leaves the accepted range returns -2 and its
    control container keeps the date, the others advance -- in the synchronous form and in the overlapped begin / end form."""
    states = [drv.state() for _ in range(2)]
    controls = [drv.control(START, END) for _ in range(2)]
    for s, c in zip(states, controls):
        drv.set_bc(s, bc)
        assert drv.init(s, c) == 0
    assert drv.parallel_step(states, controls) == [0, 0]
    t = drv.get(states[1], "t", np.complex128)
    t[0, 0, :, :] = 500.0 * np.sqrt(2.0)  # global-mean temperature of 500 K: outside 180 ... 320 K (diagnostics.f90:57-66)
    drv.set(states[1], "t", t)
    before = drv.model_date(controls[1])
    assert drv.parallel_step(states, controls) == [0, -2]
    assert drv.model_date(controls[1]) == before
    assert drv.model_date(controls[0])[0] == (1982, 1, 1, 1, 20)
    # overlapped form: the date moves at _begin and is put back at _end for the failing member
    n = 2
    token = C.c_int64()
    drv.ok(drv.L.spd_parallel_step_begin((C.c_int64 * n)(*states), (C.c_int64 * n)(*controls), n, C.byref(token)))
    codes = (C.c_int32 * n)()
    drv.ok(drv.L.spd_parallel_step_end(token, codes))
    assert list(codes) == [0, -2]
    assert drv.model_date(controls[1]) == before and drv.model_date(controls[0])[0] == (1982, 1, 1, 2, 0)
    assert drv.L.spd_parallel_step_end(token, codes) < 0  # a token is good for one _end
    drv.close(*states)


def _trace(drv):
    n = drv.L.spd_driver_trace_read(None, 0)
    buf = (C.c_int32 * (2 * max(n, 1)))()
    drv.L.spd_driver_trace_read(buf, n)
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]


def test_several_device_models_in_one_parallel_step_are_enqueued_before_any_wait(drv, bc):
    """speedy_driver.f90.j2:58-79 hands ALL containers of a one-process ensemble to one parallel_step.  Here the containers
    of such a call may live in several device models (one per GPU; on this one-GPU box: members with different SST-anomaly
    lengths, which cannot share a model).  The call batches what it can -- two 2-member models -- and enqueues the step and
    range check of every model before it waits for the first; the trajectories are bitwise those of containers stepped alone."""
    alive0, _ = drv.stats(0)
    pert = [None, 0.3 * np.ones((96, 48, 12)), -0.2 * np.ones((96, 48, 12)), 0.1 * np.ones((96, 48, 12))]
    months = [1, 1, 3, 3]
    together, alone = [drv.state() for _ in range(4)], [drv.state() for _ in range(4)]
    ctl_t, ctl_a = [drv.control(START, END) for _ in range(4)], [drv.control(START, END) for _ in range(4)]
    for group, ctls in ((together, ctl_t), (alone, ctl_a)):
        for s, c, p, nm in zip(group, ctls, pert, months):
            drv.ok(drv.L.spd_modelstate_init_sst_anom(s, nm))
            drv.set_bc(s, bc, p)
            assert drv.init(s, c) == 0
    drv.ok(drv.L.spd_driver_trace(1))
    assert drv.parallel_step(together, ctl_t) == [0, 0, 0, 0]
    assert [drv.stats(s)[1] for s in together] == [2, 2, 2, 2] and drv.stats(0)[0] == alive0 + 2 + 4
    # both models enqueued, then the waits (with a host thread per device model -- PYSPEEDY_AMD_ISSUE_THREADS=2, the rehearsal of
    # the one-thread-per-GPU enqueue -- the two enqueues may be recorded in either order)
    tr = _trace(drv)
    assert sorted(tr[:2]) == [(1, 0), (1, 1)] and tr[2:] == [(2, 0), (3, 0), (2, 1), (3, 1)]
    for _ in range(38):
        assert drv.parallel_step(together, ctl_t) == [0, 0, 0, 0]
    # ... and in the overlapped form
    drv.ok(drv.L.spd_driver_trace(1))
    n = 4
    token, codes = C.c_int64(), (C.c_int32 * n)()
    drv.ok(drv.L.spd_parallel_step_begin((C.c_int64 * n)(*together), (C.c_int64 * n)(*ctl_t), n, C.byref(token)))
    assert sorted(_trace(drv)) == [(1, 0), (1, 1)]
    drv.ok(drv.L.spd_parallel_step_end(token, codes))
    assert list(codes) == [0, 0, 0, 0] and _trace(drv)[2:] == [(2, 0), (3, 0), (2, 1), (3, 1)]
    drv.ok(drv.L.spd_driver_trace(0))
    for s, c in zip(alone, ctl_a):
        for _ in range(40):
            assert drv.step(s, c) == 0
    assert drv.model_date(ctl_t[3]) == drv.model_date(ctl_a[3])
    for st, sa in zip(together, alone):
        for name, dt in (("vor", np.complex128), ("t", np.complex128), ("ps", np.complex128), ("tr", np.complex128)):
            assert np.array_equal(drv.get(st, name, dt), drv.get(sa, name, dt)), name
        for name in ("olr", "land_temp", "sst_am", "rad_tau2"):
            assert np.array_equal(drv.get(st, name), drv.get(sa, name)), name
    drv.close(*together, *alone)
    assert drv.stats(0)[0] == alive0


def test_a_device_model_that_cannot_be_stepped_does_not_stop_the_others(drv, bc):
    """Two device models in one call; one of them already has two range checks in flight (two _begin without _end), so a
    third cannot be issued: its members report -3, the call returns that model's error AFTER the other model has been
    stepped, dates move only for the members that were stepped."""
    a, b = drv.state(), drv.state()
    ca, cb = drv.control(START, END), drv.control(START, END)
    drv.ok(drv.L.spd_modelstate_init_sst_anom(b, 3))  # (different anomaly length: the two cannot share a model)
    for s, c in ((a, ca), (b, cb)):
        drv.set_bc(s, bc)
        assert drv.init(s, c) == 0
    one = lambda v: (C.c_int64 * 1)(v)
    t1, t2, t3 = C.c_int64(), C.c_int64(), C.c_int64()
    drv.ok(drv.L.spd_parallel_step_begin(one(a), one(ca), 1, C.byref(t1)))
    drv.ok(drv.L.spd_parallel_step_begin(one(a), one(ca), 1, C.byref(t2)))
    date_a, date_b = drv.model_date(ca), drv.model_date(cb)
    drv.ok(drv.L.spd_parallel_step_begin((C.c_int64 * 2)(a, b), (C.c_int64 * 2)(ca, cb), 2, C.byref(t3)))
    assert drv.model_date(ca) == date_a and drv.model_date(cb)[0] == (1982, 1, 1, 0, 40)
    codes = (C.c_int32 * 2)()
    assert drv.L.spd_parallel_step_end(t3, codes) < 0 and b"in flight" in drv.L.spd_last_error()
    assert list(codes) == [-3, 0]
    assert drv.model_date(ca) == date_a and drv.model_date(cb)[0] == (1982, 1, 1, 0, 40) and date_b[0] == (1982, 1, 1, 0, 0)
    c1 = (C.c_int32 * 1)()
    for t in (t1, t2):
        drv.ok(drv.L.spd_parallel_step_end(t, c1))
        assert c1[0] == 0
    assert drv.parallel_step([a, b], [ca, cb]) == [0, 0]  # everything works again
    assert drv.get(a, "current_step", np.int32) == 3 and drv.get(b, "current_step", np.int32) == 2
    drv.close(a, b)


def test_device_placement_and_boundary_broadcast(drv, bc):
    """One process, several GPUs: the placement interface on the one device of this box (device ids, round-robin placement of
    single containers, block placement of an ensemble, refusal of devices that do not exist) and the device-to-device
    broadcast of the boundary fields, after which the receiving members run exactly like members that were set field by
    field."""
    ndev = C.c_int32()
    drv.ok(drv.L.spd_device_count(C.byref(ndev)))
    assert ndev.value >= 1
    assert drv.L.spd_set_device_placement(ndev.value + 1) < 0
    bad = C.c_int64()
    assert drv.L.spd_modelstate_init_on(C.byref(bad), ndev.value) < 0 and b"no such HIP device" in drv.L.spd_last_error()
    drv.ok(drv.L.spd_set_device_placement(ndev.value))
    try:
        root, other = drv.state(), drv.state()
        ens = (C.c_int64 * 3)()
        drv.ok(drv.L.spd_modelstate_init_ensemble(ens, 3))
    finally:
        drv.ok(drv.L.spd_set_device_placement(0))
    named = C.c_int64()
    drv.ok(drv.L.spd_modelstate_init_on(C.byref(named), 0))
    dev = C.c_int32(-1)
    for cnt, expect in ((root, 0), (other, 1 % ndev.value), (named.value, 0), (ens[0], 0), (ens[2], 2 * ndev.value // 3)):
        drv.ok(drv.L.spd_modelstate_device(cnt, C.byref(dev)))
        assert dev.value == expect
    cnts = [root, other, named.value] + list(ens)
    drv.set_bc(root, bc, 0.2 * np.ones((96, 48, 12)))
    drv.ok(drv.L.spd_broadcast_boundary((C.c_int64 * len(cnts))(*cnts), len(cnts), 0))
    peer, local, coll = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
    drv.ok(drv.L.spd_broadcast_boundary_stats(C.byref(peer), C.byref(local), C.byref(coll)))
    # the fields cross to another GPU once per GPU, whatever the number of containers there, all GPUs in ONE RCCL broadcast
    # (here: one GPU, nothing crosses)
    assert coll.value + peer.value == ndev.value - 1 and coll.value + peer.value + local.value == len(cnts) - 1
    by_hand = drv.state()
    drv.set_bc(by_hand, bc, 0.2 * np.ones((96, 48, 12)))
    for name in ("orog", "sst12", "soil_wc_l3", "sea_ice_frac12", "alb0"):
        for cnt in cnts[1:]:
            assert np.array_equal(drv.get(cnt, name), drv.get(by_hand, name)), name
    controls = [drv.control(START, END) for _ in cnts]
    ch = drv.control(START, END)
    for s, c in zip(cnts + [by_hand], controls + [ch]):
        assert drv.init(s, c) == 0
    for _ in range(4):
        assert drv.parallel_step(cnts, controls) == [0] * len(cnts)
        assert drv.step(by_hand, ch) == 0
    for cnt in (other, ens[1]):
        assert np.array_equal(drv.get(cnt, "t", np.complex128), drv.get(by_hand, "t", np.complex128))
    drv.close(by_hand, *cnts)


def test_init_of_a_batched_member_never_resets_the_other_members(drv, bc):
    """A batched model has one date and one step counter.  Initialising a member again after the batch has stepped, or with
    another start date than its fellow members, takes that member out of the batch first; the others keep their date."""
    n = 3
    ens = (C.c_int64 * n)()
    drv.ok(drv.L.spd_modelstate_init_ensemble(ens, n))
    states = list(ens)
    controls = [drv.control(START, END) for _ in range(n)]
    for s, c in zip(states, controls):
        drv.set_bc(s, bc)
        assert drv.init(s, c) == 0
    for _ in range(3):
        assert drv.parallel_step(states, controls) == [0] * n
    assert drv.stats(states[0])[1] == n
    assert drv.init(states[1], controls[1]) == 0  # start member 1 over
    assert drv.stats(states[1])[1] == 1 and drv.stats(states[0])[1] == 1
    assert drv.get(states[0], "current_step", np.int32) == 3 and drv.get(states[1], "current_step", np.int32) == 0
    assert drv.model_date(controls[1])[0] == START and drv.model_date(controls[0])[0] == (1982, 1, 1, 2, 0)
    # a member with its own start date is on its own from the beginning
    ens2 = (C.c_int64 * 2)()
    drv.ok(drv.L.spd_modelstate_init_ensemble(ens2, 2))
    c0, c1 = drv.control(START, END), drv.control((1982, 2, 1, 0, 0), (1982, 2, 4, 0, 0))
    for s in ens2:
        drv.set_bc(s, bc)
    assert drv.init(ens2[0], c0) == 0 and drv.init(ens2[1], c1) == 0
    assert drv.stats(ens2[0])[1] == 1 and drv.stats(ens2[1])[1] == 1
    assert drv.parallel_step(list(ens2), [c0, c1]) == [0, 0]
    assert drv.model_date(c0)[0] == (1982, 1, 1, 0, 40) and drv.model_date(c1)[0] == (1982, 2, 1, 0, 40)
    drv.close(*states, *ens2)


@pytest.mark.parametrize("fp32", [False, True])
def test_random_groupings_keep_every_container_on_the_trajectory_of_a_lone_container(drv, bc, fp32):
    """A randomised sequence (fixed seed) of parallel_step / begin-end over changing subsets of six containers, state writes and
    a scalar change -- gathers, partial gathers, splits and two-model calls in whatever order they come -- against six twin
    containers that are only ever stepped alone with step(): every container ends bitwise on its twin's trajectory, with its
    twin's date and step counter.  fp32: the same with BASELINE cfg 5's fp32 column physics in every container, whose fp32-stored
    arrays have to travel with their members through every gather and split."""
    rng = np.random.default_rng(20260410)
    n = 6
    world = [drv.state() for _ in range(n)]
    twins = [drv.state() for _ in range(n)]
    cw = [drv.control(START, END) for _ in range(n)]
    ct = [drv.control(START, END) for _ in range(n)]
    for i in range(n):
        for s, c in ((world[i], cw[i]), (twins[i], ct[i])):
            drv.set_bc(s, bc, 0.05 * i * np.ones((96, 48, 12)))
            if fp32:
                model = C.c_void_p()
                drv.ok(drv.L.spd_driver_model(s, C.byref(model), None, None))
                drv.ok(drv.L.spd_model_set_physics_precision(model, 1))
            assert drv.init(s, c) == 0
    steps, largest = [0] * n, 0
    for op in range(60):
        kind = rng.integers(0, 10)
        # mostly subsets of the containers that stand at the same date (they can share a device model: gathers, then splits when
        # the next subset differs), sometimes any subset (several dates, hence several device models in one call)
        counts = {c: steps.count(c) for c in steps}
        modal = max(counts, key=lambda c: (counts[c], -c))
        pool = [i for i in range(n) if steps[i] == modal] if rng.random() < 0.75 else list(range(n))
        k = int(rng.integers(1, len(pool) + 1))
        subset = sorted(rng.choice(pool, size=k, replace=False).tolist())
        if kind <= 7:
            for i in subset:
                steps[i] += 1 if kind <= 5 else 2
        if kind <= 5:  # synchronous parallel_step over a subset
            assert drv.parallel_step([world[i] for i in subset], [cw[i] for i in subset]) == [0] * k
            for i in subset:
                assert drv.step(twins[i], ct[i]) == 0
        elif kind <= 7:  # the overlapped form, two steps deep
            ids, ctl = (C.c_int64 * k)(*[world[i] for i in subset]), (C.c_int64 * k)(*[cw[i] for i in subset])
            t1, t2, codes = C.c_int64(), C.c_int64(), (C.c_int32 * k)()
            drv.ok(drv.L.spd_parallel_step_begin(ids, ctl, k, C.byref(t1)))
            drv.ok(drv.L.spd_parallel_step_begin(ids, ctl, k, C.byref(t2)))
            for t in (t1, t2):
                drv.ok(drv.L.spd_parallel_step_end(t, codes))
                assert list(codes) == [0] * k
            for i in subset:
                assert drv.step(twins[i], ct[i]) == 0 and drv.step(twins[i], ct[i]) == 0
        elif kind == 8:  # the host writes a member's temperature
            i = subset[0]
            t = drv.get(world[i], "t", np.complex128)
            t[1:4, 1:4] *= 1.0 + 1e-6 * (op + 1)
            drv.set(world[i], "t", t)
            drv.set(twins[i], "t", t)
        else:  # a scalar of its own takes a member out of its batch
            i, one = subset[0], np.array([op % 2], dtype=np.int32)
            for s in (world[i], twins[i]):
                drv.ok(drv.L.spd_set(s, b"land_coupling_flag", one.ctypes.data_as(C.c_void_p), 4))
        largest = max(largest, max(drv.stats(s)[1] for s in world))
    sizes = sorted(drv.stats(s)[1] for s in world)
    print("device models of the six containers at the end:", sizes, "; largest model on the way:", largest)
    assert largest >= 3
    for i in range(n):
        assert drv.model_date(cw[i]) == drv.model_date(ct[i]), i
        assert drv.get(world[i], "current_step", np.int32) == drv.get(twins[i], "current_step", np.int32)
        for name, dt in (("vor", np.complex128), ("t", np.complex128), ("ps", np.complex128), ("tr", np.complex128)):
            assert np.array_equal(drv.get(world[i], name, dt), drv.get(twins[i], name, dt)), (i, name)
        for name in ("olr", "land_temp", "sst_am", "rad_tau2", "hfluxn", "tt_rsw", "precnv", "ustr"):
            assert np.array_equal(drv.get(world[i], name), drv.get(twins[i], name)), (i, name)
    if fp32:
        model = C.c_void_p()
        drv.ok(drv.L.spd_driver_model(world[0], C.byref(model), None, None))
        assert drv.L.spd_model_var_storage(model, b"rad_tau2") == 4 and drv.L.spd_model_var_storage(model, b"t") == 8
    drv.close(*world, *twins)


def test_a_synchronous_step_beside_a_pending_one_takes_the_free_check_slot(drv, bc):
    """(advisor, round 3) A pending begin holds one of a model's two check slots; synchronous steps beside it must take the OTHER
    slot every time, not alternate into the busy one -- a step that had been enqueued when its check was refused left state and
    date out of step.  Three synchronous steps between a begin and its end, against a twin stepped alone."""
    a, twin = drv.state(), drv.state()
    ca, ct = drv.control(START, END), drv.control(START, END)
    for s, c in ((a, ca), (twin, ct)):
        drv.set_bc(s, bc)
        assert drv.init(s, c) == 0
    ids, ctl, token, codes = (C.c_int64 * 1)(a), (C.c_int64 * 1)(ca), C.c_int64(), (C.c_int32 * 1)(99)
    drv.ok(drv.L.spd_parallel_step_begin(ids, ctl, 1, C.byref(token)))
    for _ in range(3):
        assert drv.parallel_step([a], [ca]) == [0]
    drv.ok(drv.L.spd_parallel_step_end(token, codes))
    assert list(codes) == [0]
    for _ in range(4):
        assert drv.step(twin, ct) == 0
    assert drv.model_date(ca) == drv.model_date(ct) == ((1982, 1, 1, 2, 40), 1)
    assert drv.get(a, "current_step", np.int32) == 4
    for name in ("vor", "t", "ps"):
        assert np.array_equal(drv.get(a, name, np.complex128), drv.get(twin, name, np.complex128)), name
    drv.close(a, twin)


def test_containers_come_and_go_cheaply_and_leave_nothing_behind(drv, bc, spectral):
    """A device model carves its arrays from one zero-filled block and creates its stream only when it is first stepped: a host with the reference's call sequence creates, initialises, gathers and closes one-member models by the
    hundred.  Fresh containers read back zeros whatever lived in that memory before; the trajectory of a container does not
    depend on how many came and went before it; spd_model_memory reports what a member costs."""
    import ctypes
    from pyspeedy_amd.model import EnsembleModel
    start, end = (1982, 1, 1, 0, 0), (1982, 1, 3, 0, 0)
    alive0 = C.c_int32()
    drv.ok(drv.L.spd_driver_stats(0, C.byref(alive0), None))

    def one_run(perturb):
        s, c = drv.state(), drv.control(start, end)
        assert not drv.get(s, "t", np.complex128).any() and not drv.get(s, "sst12").any() and not drv.get(s, "rad_tau2").any()  # zero-filled
        drv.set_bc(s, bc, perturb)
        assert drv.init(s, c) == 0
        for _ in range(5):
            assert drv.step(s, c) == 0
        t = drv.get(s, "t", np.complex128)
        drv.ok(drv.L.spd_modelstate_close(s))
        drv.ok(drv.L.spd_controlparams_close(c))
        return t

    first = one_run(0.5)
    for k in range(40):  # (never stepped: these never take a stream; their blocks dirty the memory the next ones get)
        s = drv.state()
        drv.set_bc(s, bc, 1.0 + k)
        drv.ok(drv.L.spd_modelstate_close(s))
    assert np.array_equal(one_run(0.5), first)
    alive = C.c_int32()
    drv.ok(drv.L.spd_driver_stats(0, C.byref(alive), None))
    assert alive.value == alive0.value
    for members in (1, 8):
        m = EnsembleModel(spectral, members)
        reserved, used = m.memory()
        assert 18.0e6 * members < used <= reserved < 21.5e6 * members + 2e6, (members, reserved, used)
        m.init_sst_anom(14)  # (16 planes per member instead of 3: a second block)
        assert m.memory()[1] >= used + members * 16 * 96 * 48 * 8
        m.close()
    assert ctypes.sizeof(ctypes.c_size_t) == 8


def test_the_overlapped_form_carries_its_range_checks_in_the_next_step(drv, bc):
    """spd_parallel_step_begin puts the range check of its step off (spd_model_check_defer): the next _begin's first launch
    carries it, the last one goes out when it is collected.  A member that leaves the accepted range in the middle of a pipelined
    loop is reported at the step it happened, and the codes and states are those of the synchronous loop."""
    import ctypes
    n, steps = 3, 6

    def setup():
        states = [drv.state() for _ in range(n)]
        controls = [drv.control(START, END) for _ in range(n)]
        for k, (s, c) in enumerate(zip(states, controls)):
            drv.set_bc(s, bc, 0.1 * k)
            assert drv.init(s, c) == 0
        return states, controls

    def spoil(states):
        t = drv.get(states[1], "t", np.complex128)
        t[0, 0, :, :] = 500.0 * np.sqrt(2.0)
        drv.set(states[1], "t", t)

    sync_states, sync_controls = setup()
    sync_codes = []
    for k in range(steps):
        if k == 3:
            spoil(sync_states)
        sync_codes.append(drv.parallel_step(sync_states, sync_controls))
    states, controls = setup()
    arr_s, arr_c = (C.c_int64 * n)(*states), (C.c_int64 * n)(*controls)
    codes, pending = [], None
    for k in range(steps):
        if k == 3:  # (a host write between two steps of the pipeline: collect what is in flight first, as a callback would)
            out = (C.c_int32 * n)()
            drv.ok(drv.L.spd_parallel_step_end(pending, out))
            codes.append(list(out))
            pending = None
            model, member, members = ctypes.c_void_p(), C.c_int32(), C.c_int32()
            drv.ok(drv.L.spd_driver_model(states[0], C.byref(model), C.byref(member), C.byref(members)))
            alone, rode = C.c_int32(), C.c_int32()
            drv.ok(drv.L.spd_model_check_counts(model, C.byref(alone), C.byref(rode)))
            # (the gathered model of the three containers: the checks of steps 1 and 2 rode in steps 2 and 3, that of step 3 went out
            # when it was collected just now)
            assert members.value == n and (alone.value, rode.value) == (1, 2), (members.value, alone.value, rode.value)
            spoil(states)
        token = C.c_int64()
        drv.ok(drv.L.spd_parallel_step_begin(arr_s, arr_c, n, C.byref(token)))
        if pending is not None:
            out = (C.c_int32 * n)()
            drv.ok(drv.L.spd_parallel_step_end(pending, out))
            codes.append(list(out))
        pending = token
    out = (C.c_int32 * n)()
    drv.ok(drv.L.spd_parallel_step_end(pending, out))
    codes.append(list(out))
    assert codes == sync_codes and codes[2] == [0, 0, 0] and codes[3] == [0, -2, 0], (codes, sync_codes)
    for k in (0, 2):
        assert np.array_equal(drv.get(states[k], "t", np.complex128), drv.get(sync_states[k], "t", np.complex128))
    drv.close(*states, *sync_states)


def test_ensemble_placement_by_argument_leaves_the_process_placement_alone(drv):
    """spd_modelstate_init_ensemble_on(cnts, n, k) takes the number of devices as an argument: the process-wide placement
    (spd_set_device_placement / PYSPEEDY_AMD_DEVICES) is neither read nor reset by it -- SpeedyEns(devices=k) used to switch it to
    k and back to 0, losing whatever the host had set.  Every call leaves the caller's current HIP device what it was."""
    import torch
    from pyspeedy_amd.speedy import SpeedyEns
    ndev = C.c_int32()
    drv.ok(drv.L.spd_device_count(C.byref(ndev)))
    before = torch.cuda.current_device()
    drv.ok(drv.L.spd_set_device_placement(ndev.value))
    try:
        ens = SpeedyEns(5, devices=1)
        assert [m._state_cnt > 0 for m in ens.members] == [True] * 5
        first, second = drv.state(), drv.state()  # the placement set above still rules single containers: round-robin
        dev = C.c_int32(-1)
        drv.ok(drv.L.spd_modelstate_device(second, C.byref(dev)))
        assert dev.value == 1 % ndev.value
        cnts = (C.c_int64 * 40)()
        assert drv.L.spd_modelstate_init_ensemble_on(cnts, 40, ndev.value + 1) < 0
        drv.ok(drv.L.spd_modelstate_init_ensemble_on(cnts, 40, 0))
        assert drv.stats(cnts[0])[1] == 20 and drv.stats(cnts[39])[1] == 20  # 32 or more on one device: two device models
        drv.close(first, second, *cnts)
        del ens
    finally:
        drv.ok(drv.L.spd_set_device_placement(0))
    assert torch.cuda.current_device() == before


def test_collective_broadcast_between_device_models_reaches_rccl(spectral, bc):
    """spd_model_broadcast_vars is ONE RCCL broadcast (ncclCommInitAll + ncclBroadcast in a group call, single-process
    communicators) of the boundary fields between device models on different GPUs -- what spd_broadcast_boundary uses across
    devices.  This box has one GPU: the one-rank form (a broadcast to nobody) still loads RCCL, creates the communicator and runs
    the collective on the device; two models on the same GPU are refused (same-device copies are spd_model_copy_vars)."""
    import pyspeedy_amd
    from pyspeedy_amd.model import EnsembleModel
    L = pyspeedy_amd.lib()
    a, b = EnsembleModel(spectral, 2), EnsembleModel(spectral, 1)
    a.set("sst12", np.asarray(bc["sst"], dtype=np.float64), 1)
    names = (C.c_char_p * 3)(b"orog", b"sst12", b"stl12")
    before = a.get("sst12", 1)
    rc = L.spd_model_broadcast_vars((C.c_void_p * 1)(a._m), (C.c_int * 1)(1), 1, 0, names, 3)
    assert rc == 0, L.spd_last_error()
    import torch
    torch.cuda.synchronize()
    assert np.array_equal(a.get("sst12", 1), before)
    rc = L.spd_model_broadcast_vars((C.c_void_p * 2)(a._m, b._m), (C.c_int * 2)(1, 0), 2, 0, names, 3)
    assert rc < 0 and b"one model per GPU" in L.spd_last_error()
    assert L.spd_model_broadcast_vars((C.c_void_p * 1)(a._m), (C.c_int * 1)(5), 1, 0, names, 3) < 0
    bad = (C.c_char_p * 1)(b"rad_tau2")  # (an array whose storage follows the physics precision does not travel this way)
    assert L.spd_model_broadcast_vars((C.c_void_p * 1)(a._m), (C.c_int * 1)(0), 1, 0, bad, 1) < 0
    a.close()
    b.close()


def test_an_ensemble_that_does_not_fit_is_refused_whole():
    """An ensemble is several device models (two from 32 members of a device up).  Here it is sized from the free memory of the GPU
    so that the FIRST model fits and the second cannot (model.hip: arena_alloc, 19 MB per member in one hipMalloc): the call fails
    with the allocation's error, no container of it exists (every id 0), the first model's memory is back, and the next ensemble
    comes up and steps.  (The all-or-nothing logic itself runs under ASan / LeakSanitizer with an injected failure:
    tests/sanitize/driver_sanitize.cpp.)"""
    import torch
    from pyspeedy_amd import speedy_driver as drv
    L = drv._L()
    torch.cuda.synchronize()
    alive0, _ = drv.driver_stats()
    free0, total = torch.cuda.mem_get_info()
    per_member = 19 << 20
    n = int(1.3 * free0 / per_member)  # two models of 0.65 x the free memory each
    assert n >= 64 and (n // 2) * per_member < 0.8 * free0
    arr = (C.c_int64 * n)(*([-1] * n))
    rc = L.spd_modelstate_init_ensemble(arr, n)
    assert rc == -2, rc  # SPD_E_DEVICE
    message = L.spd_last_error().decode()
    assert "hipMalloc" in message and "memory" in message.lower(), message
    assert not any(arr)
    assert drv.driver_stats()[0] == alive0
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 > free0 - (2 << 30), (free0, free1)  # (a context keeps up to 1 GiB of released blocks for the next model)
    from datetime import datetime
    from pyspeedy_amd.speedy import SpeedyEns
    ens = SpeedyEns(40, start_date=datetime(1982, 1, 1), end_date=datetime(1982, 1, 1, 2, 0))
    ens.set_bc()
    ens.run()
    assert ens.get_current_step() == 3 and drv.driver_stats()[0] == alive0 + 1  # (SpeedyEns: one device model per GPU)
    del ens
