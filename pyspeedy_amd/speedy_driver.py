"""The function set of the reference's f2py module `pyspeedy.speedy_driver.speedy_driver` (imported there as
`_speedy`; generated from registry/templates/speedy_driver.f90.j2) on top of the MI355X C ABI.

Same names, argument order and return conventions, so that `pyspeedy/speedy.py`-style host code runs unchanged:

    modelstate_init() -> state_cnt              modelstate_init_sst_anom(state_cnt, n_months)   modelstate_close(state_cnt)
    create_datetime(y, m, d, h, mi) -> cnt      get_datetime(cnt) -> (y, m, d, h, mi)           close_datetime(cnt)
    controlparams_init(start_cnt, end_cnt) -> control_cnt                                       controlparams_close(control_cnt)
    init(state_cnt, control_cnt) -> code        step(state_cnt, control_cnt) -> code            check(state_cnt) -> code
    parallel_step(state_cnts, control_cnts) -> int32 codes
    transform_spectral2grid(state_cnt)          transform_grid2spectral(state_cnt)              apply_grid_filter(state_cnt)
    get_<v>(state_cnt[, n_months])              set_<v>(state_cnt, value[, n_months])
    get_<v>_shape(state_cnt) -> int tuple (zeros while unallocated)                             is_array_<v>() -> bool

Containers are int64 handles (the reference hands out the bits of a Fortran pointer, speedy_driver.f90.j2:38-40); here
they index a table of Python objects.  Error codes (error_codes.f90:7-9): 0 ok, -1 state not initialised, -2 prognostic
variables out of range.  A state container created by modelstate_init() owns a one-member model on the current GPU.

Extension for ensembles (the reference steps members one by one under OpenMP, speedy_driver.f90.j2:58-79):
`modelstate_init_ensemble(n)` returns n state containers that are the members of ONE batched device model; parallel_step
on exactly those containers advances all of them with one set of kernel launches.  Members of a batched model share the
model date, so they cannot be stepped individually.
"""
import itertools
import threading

import numpy as np

from . import registry as R

ERROR_CODES = {0: "Run successful.",
               -1: "The model state was not initialized. Initialize it (init) before running the model.",
               -2: "Model variables out of the accepted range (diagnostics.f90)."}

_lock = threading.Lock()
_ids = itertools.count(1)
_objects = {}
_contexts = {}  # device index -> ModSpectral


def _register(obj):
    with _lock:
        cnt = next(_ids)
        _objects[cnt] = obj
    return cnt


def _lookup(cnt, kind):
    obj = _objects.get(int(cnt))
    if not isinstance(obj, kind):
        raise ValueError("container %r is not a live %s" % (cnt, kind.__name__))
    return obj


def _context():
    import torch
    from .spectral import ModSpectral
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None
    if dev is None:
        raise RuntimeError("pyspeedy_amd.speedy_driver needs a HIP device; there is no CPU fallback")
    with _lock:
        if dev not in _contexts:
            _contexts[dev] = ModSpectral(dev)
        return _contexts[dev]


# --------------------------------------------------------------------------------------------------------------------
# containers
# --------------------------------------------------------------------------------------------------------------------
class _Date:
    def __init__(self, y, m, d, h, mi):
        self.ymdhm = (int(y), int(m), int(d), int(h), int(mi))


class _Control:  # ControlParams_t, model_control.f90:32-47: start / end date; the running model date lives with the state
    def __init__(self, start, end):
        self.start, self.end = start.ymdhm, end.ymdhm


class _Batch:
    """One device model shared by its member containers."""

    def __init__(self, nmembers):
        from .model import EnsembleModel
        self.sp = _context()
        self.model = EnsembleModel(self.sp, nmembers)
        self.nmembers = nmembers
        self.initialized = [False] * nmembers
        self.n_months = 1  # allocation of sst_anom: n_months + 2 planes (3 right after modelstate_init)
        self.sst_anom_allocated = False
        self.refs = nmembers

    def release(self):
        self.refs -= 1
        if self.refs == 0:
            self.model.close()


class _State:
    def __init__(self, batch, member):
        self.batch, self.member = batch, member
        self.host = {}  # registry arrays of kind "host"
        self.scalars = {"increase_co2": False, "compute_shortwave": True, "air_absortivity_co2": 6.0,
                        "land_coupling_flag": True, "sst_anomaly_coupling_flag": True, "ablco2_ref": 6.0}


def modelstate_init():
    return _register(_State(_Batch(1), 0))


def modelstate_init_ensemble(nmembers):
    batch = _Batch(int(nmembers))
    return [_register(_State(batch, i)) for i in range(batch.nmembers)]


def modelstate_init_sst_anom(state_cnt, n_months):
    st = _lookup(state_cnt, _State)
    b = st.batch
    if not b.sst_anom_allocated or b.n_months != int(n_months):
        b.model.init_sst_anom(int(n_months))
        b.n_months = int(n_months)
        b.sst_anom_allocated = True


def modelstate_close(state_cnt):
    with _lock:
        st = _objects.pop(int(state_cnt), None)
    if isinstance(st, _State):
        st.batch.release()


def create_datetime(year, month, day, hour, minute):
    return _register(_Date(year, month, day, hour, minute))


def get_datetime(cnt):
    return _lookup(cnt, _Date).ymdhm


def close_datetime(cnt):
    with _lock:
        _objects.pop(int(cnt), None)


def controlparams_init(start_cnt, end_cnt):
    return _register(_Control(_lookup(start_cnt, _Date), _lookup(end_cnt, _Date)))


def controlparams_close(control_cnt):
    with _lock:
        _objects.pop(int(control_cnt), None)


# --------------------------------------------------------------------------------------------------------------------
# model control
# --------------------------------------------------------------------------------------------------------------------
def _push_flags(st, co2=True):
    s = st.scalars
    m = st.batch.model
    m.set_flags(s["land_coupling_flag"], s["sst_anomaly_coupling_flag"], s["increase_co2"])
    if co2:  # (the model raises its own value daily when increase_co2 is set: do not overwrite it with the mirror)
        m.set_co2(s["air_absortivity_co2"])


def init(state_cnt, control_cnt):
    """initialize_state (initialization.f90:13-91) from the boundary fields already stored with set_<v>."""
    st, ctl = _lookup(state_cnt, _State), _lookup(control_cnt, _Control)
    b = st.batch
    if b.nmembers == 1:
        _push_flags(st)
        b.model.init(ctl.start)
    else:
        # one member of a batched model: initialise a scratch one-member model from this member's boundary fields and
        # copy the resulting state into the member's slot
        from .model import EnsembleModel
        scratch = EnsembleModel(b.sp, 1)
        try:
            if b.sst_anom_allocated:
                scratch.init_sst_anom(b.n_months)
            scratch.copy_member_from(b.model, st.member, 0)
            s = st.scalars
            scratch.set_flags(s["land_coupling_flag"], s["sst_anomaly_coupling_flag"], s["increase_co2"])
            scratch.set_co2(s["air_absortivity_co2"])
            scratch.init(ctl.start)
            b.model.copy_member_from(scratch, 0, st.member)
            b.model.sync()
        finally:
            scratch.close()
        _push_flags(st)
        b.model.mark_initialized(0, ctl.start)
    b.initialized[st.member] = True
    return 0


def _step_batch(b):
    if not all(b.initialized):
        return np.full(b.nmembers, -1, dtype=np.int32)
    b.model.run(1)
    return b.model.check(2).astype(np.int32)


def _group(state_cnts, control_cnts):
    """The distinct device models behind the containers, each with the positions of its members in the argument list."""
    state_cnts = np.asarray(state_cnts, dtype=np.int64).ravel()
    if np.asarray(control_cnts).size != state_cnts.size:
        raise ValueError("parallel_step: one control container per state container")
    states = [_lookup(c, _State) for c in state_cnts]
    groups = []
    for st in states:
        b = st.batch
        if any(g[0] is b for g in groups):
            continue
        mine = [i for i, s in enumerate(states) if s.batch is b]
        if sorted(states[i].member for i in mine) != list(range(b.nmembers)):
            raise ValueError("parallel_step needs every member of a batched ensemble model exactly once")
        groups.append((b, mine, [states[i].member for i in mine]))
    return len(states), groups


def step(state_cnt, control_cnt):
    """do_single_step (speedy.f90:20-74) followed by the range check of diagnostics.f90."""
    st = _lookup(state_cnt, _State)
    _lookup(control_cnt, _Control)
    if st.batch.nmembers != 1:
        raise ValueError("a member of a batched ensemble model cannot be stepped on its own; use parallel_step")
    return int(_step_batch(st.batch)[0])


def parallel_step(state_cnts, control_cnts):
    n, groups = _group(state_cnts, control_cnts)
    codes = np.zeros(n, dtype=np.int32)
    for b, positions, members in groups:
        codes[positions] = _step_batch(b)[members]
    return codes


# Extension: the same step with the range check overlapped.  parallel_step_begin enqueues the step and its check and returns
# a token; parallel_step_end(token) waits for that check only.  A loop that begins step k + 1 before ending step k never
# leaves the GPU waiting for the host (the synchronous form costs ~0.1 ms per step at 64 members); the price is that the
# error code of step k is seen after step k + 1 has been enqueued.  At most two steps may be in flight per model.
def parallel_step_begin(state_cnts, control_cnts):
    n, groups = _group(state_cnts, control_cnts)
    pending = []
    for b, positions, members in groups:
        if not all(b.initialized):
            pending.append((b, positions, members, None))
        else:
            b.model.run(1)
            pending.append((b, positions, members, b.model.check_begin(2)))
    return _register((n, pending))


def parallel_step_end(token):
    with _lock:
        n, pending = _objects.pop(int(token))
    codes = np.zeros(n, dtype=np.int32)
    for b, positions, members, slot in pending:
        codes[positions] = -1 if slot is None else b.model.check_end(slot)[members]
    return codes


def check(state_cnt):
    st = _lookup(state_cnt, _State)
    if not st.batch.initialized[st.member]:
        return -1
    return int(st.batch.model.check(1)[st.member])


def transform_spectral2grid(state_cnt):
    st = _lookup(state_cnt, _State)
    st.batch.model.spectral2grid(st.member, 1)


def transform_grid2spectral(state_cnt):
    st = _lookup(state_cnt, _State)
    st.batch.model.grid2spectral(st.member, 1)


def ensemble_grid_arrays(state_cnt, names):
    """Extension: the grid-space variables `names` of ALL members of the batched model `state_cnt` belongs to, after one
    batched spectral2grid: dict name -> float64 array [member, (lev,) lat, lon] (one device-to-host copy per variable)."""
    b = _lookup(state_cnt, _State).batch
    b.model.spectral2grid()
    return {n: b.model.device_view(n).cpu().numpy() for n in names}


def apply_grid_filter(state_cnt):
    st = _lookup(state_cnt, _State)
    st.batch.model.grid_filter(st.member, 1)


# --------------------------------------------------------------------------------------------------------------------
# registry access: get_<v>, set_<v>, get_<v>_shape, is_array_<v> (generated per variable in the reference)
# --------------------------------------------------------------------------------------------------------------------
def _table(st, name):
    sp = st.batch.sp
    if name == "lon":  # initialization.f90:86
        return (np.float32(3.75) * np.arange(R.IX, dtype=np.float32)).astype(np.float32)
    if name == "lat":  # initialization.f90:87: real(radang) * 90.0 / asin(1.0), default real
        return (sp.table("radang").astype(np.float32) * np.float32(90.0) / np.arcsin(np.float32(1.0))).astype(np.float32)
    if name == "lev":  # initialization.f90:85
        return sp.table("fsg").astype(np.float32)
    if name == "deglat_s":  # sea_model.f90: grid latitudes in degrees
        return sp.table("radang") * 90.0 / np.arcsin(1.0)
    if name == "fband":
        return sp.table("fband").reshape((301, 4), order="F")
    if name in ("xgeop1", "xgeop2"):  # geopotential.f90:16-31
        hsg, fsg = sp.table("hsg"), sp.table("fsg")
        rgas = float(np.float32(2.0) / np.float32(7.0)) * 1004.0
        if name == "xgeop1":
            return rgas * np.log(hsg[1:] / fsg)
        out = np.zeros(R.KX)
        out[1:] = rgas * np.log(fsg[1:] / hsg[1:-1])
        return out
    raise KeyError(name)


def _get(name, state_cnt, n_months=None):
    st = _lookup(state_cnt, _State)
    v = R.REGISTRY[name]
    b = st.batch
    if v.where == "device":
        return b.model.get(name, st.member)
    if v.where == "table":
        return _table(st, name)
    if v.where == "host":
        return st.host.setdefault(name, np.zeros(R.shape_of(name), dtype=v.dtype, order="F")).copy(order="F")
    if name == "current_step":
        return b.model.current_step
    if name == "air_absortivity_co2":
        return b.model.co2
    return st.scalars[name]


def _set(name, state_cnt, value, n_months=None):
    st = _lookup(state_cnt, _State)
    v = R.REGISTRY[name]
    b = st.batch
    if v.where == "device":
        b.model.set(name, np.asarray(value), st.member)
    elif v.where == "host":
        a = np.asarray(value, dtype=v.dtype)
        if a.shape != R.shape_of(name):
            raise ValueError("Array shape missmatch")
        st.host[name] = np.array(a, order="F")
    elif v.where == "table":
        raise ValueError("'%s' is a read-only table of the device context" % name)
    elif name == "current_step":
        raise ValueError("current_step is advanced by the model")
    else:
        st.scalars[name] = v.dtype(value).item()
        _push_flags(st, co2=(name == "air_absortivity_co2"))


def _shape(name, state_cnt):
    st = _lookup(state_cnt, _State)
    if name == "sst_anom":
        return (R.IX, R.IL, st.batch.n_months + 2)
    return R.shape_of(name)


def __getattr__(attr):  # PEP 562: the per-variable functions
    for prefix, make in (("is_array_", lambda n: (lambda: R.is_array(n))),
                         ("get_", None), ("set_", lambda n: (lambda cnt, value, n_months=None: _set(n, cnt, value, n_months)))):
        if not attr.startswith(prefix):
            continue
        rest = attr[len(prefix):]
        if prefix == "get_":
            if rest.endswith("_shape") and rest[:-6] in R.REGISTRY:
                name = rest[:-6]
                return lambda cnt: _shape(name, cnt)
            if rest in R.REGISTRY:
                return lambda cnt, n_months=None: _get(rest, cnt, n_months)
        elif rest in R.REGISTRY:
            return make(rest)
    raise AttributeError("module 'speedy_driver' has no attribute %r" % attr)
