"""The function set of the reference's f2py module `pyspeedy.speedy_driver.speedy_driver` (imported there as
`_speedy`; generated from registry/templates/speedy_driver.f90.j2), bound to the C entry points of the same names in
libpyspeedy_amd.so (include/pyspeedy_amd_driver.h, csrc/driver.cpp).

Same names, argument order and return conventions, so that `pyspeedy/speedy.py`-style host code runs unchanged:

    modelstate_init() -> state_cnt              modelstate_init_sst_anom(state_cnt, n_months)   modelstate_close(state_cnt)
    create_datetime(y, m, d, h, mi) -> cnt      get_datetime(cnt) -> (y, m, d, h, mi)           close_datetime(cnt)
    controlparams_init(start_cnt, end_cnt) -> control_cnt                                       controlparams_close(control_cnt)
    init(state_cnt, control_cnt) -> code        step(state_cnt, control_cnt) -> code            check(state_cnt) -> code
    parallel_step(state_cnts, control_cnts) -> int32 codes
    transform_spectral2grid(state_cnt)          transform_grid2spectral(state_cnt)              apply_grid_filter(state_cnt)
    get_<v>(state_cnt[, n_months])              set_<v>(state_cnt, value[, n_months])
    get_<v>_shape(state_cnt) -> int tuple (zeros while unallocated)                             is_array_<v>() -> bool

Everything with a reference counterpart is ONE call into the C library: this module holds no model logic.  Containers are
the library's int64 keys.  Error codes (error_codes.f90:7-9): 0 ok, -1 state not initialised, -2 prognostic variables out of
range.  As in the reference the control container owns the model date (`get_model_datetime`).

parallel_step over independent containers: the library gathers them into one batched device model on the first call (one set
of kernel launches per step for all of them afterwards) -- see include/pyspeedy_amd_driver.h.

Extensions (no counterpart in the reference): `modelstate_init_ensemble(n)` (containers batched from the start),
`parallel_step_begin / parallel_step_end` (range check overlapped with the next step), `device_model(state_cnt)` (the batched
device model behind a container, for zero-copy access to the state), `ensemble_device_view`, `ensemble_grid_arrays`,
`ensemble_export_arrays` (a NetCDF file's payload formed on the GPU), `parallel_steps_begin / parallel_steps_end` (the steps
between two due callbacks as one device call), `ensemble_check` (one range check per device model), `on_default_streams`, the
one-process-several-GPUs placement functions.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import registry as R

ERROR_CODES = {0: "Run successful.",
               -1: "The model state was not initialized. Initialize it (init) before running the model.",
               -2: "Model variables out of the accepted range (diagnostics.f90).",
               -3: "The step of this member's device model could not be issued or checked (see the error of the call)."}


def _L():
    """The HIP library (raises if it was not built).  Calls that need a device fail in the library itself when there is none
    ("no HIP device": there is no CPU fallback); containers of dates and control parameters work without one."""
    return _lib.lib()


def _ok(rc, what):
    if rc == _lib.SPD_E_SIZE:
        raise ValueError("Array shape missmatch")  # pyspeedy/speedy.py:153
    if rc == _lib.SPD_E_ARG:
        msg = _lib.lib().spd_last_error()
        raise ValueError("%s: %s" % (what, msg.decode() if msg else "bad argument"))
    _lib.check(rc, what)


def _cnts(values):
    a = np.asarray(values, dtype=np.int64).ravel()
    return (C.c_int64 * a.size)(*a.tolist()), a.size


# --------------------------------------------------------------------------------------------------------------------
# containers
# --------------------------------------------------------------------------------------------------------------------
def modelstate_init():
    c = C.c_int64()
    _ok(_L().spd_modelstate_init(C.byref(c)), "modelstate_init")
    return c.value


def modelstate_init_ensemble(nmembers, devices=None, whole=False):
    """n containers batched from the start.  devices=None: the process-wide placement (set_device_placement /
    PYSPEEDY_AMD_DEVICES; by default the current device); devices=k: blocks over GPUs 0 .. k-1 (0: the current device)
    without touching that placement.  whole=True: ONE device model per device whatever the number of members (for hosts that hand
    over many steps at once: parallel_steps_begin); otherwise two from 32 members of a device up."""
    arr = (C.c_int64 * int(nmembers))()
    if whole:
        _ok(_L().spd_modelstate_init_ensemble_whole(arr, int(nmembers), -1 if devices is None else int(devices)), "modelstate_init_ensemble_whole")
    elif devices is None:
        _ok(_L().spd_modelstate_init_ensemble(arr, int(nmembers)), "modelstate_init_ensemble")
    else:
        _ok(_L().spd_modelstate_init_ensemble_on(arr, int(nmembers), int(devices)), "modelstate_init_ensemble_on")
    return list(arr)


# Extension: one process, several GPUs (include/pyspeedy_amd_driver.h).  The reference's ensemble is one process handing all
# containers to parallel_step; with a placement of k devices the containers are spread over GPUs 0 .. k-1 and one
# parallel_step drives them all (every device's step is enqueued before the host waits for any).
def device_count():
    n = C.c_int32()
    _ok(_L().spd_device_count(C.byref(n)), "device_count")
    return n.value


def set_device_placement(n_devices):
    """0: new containers live on the current HIP device (default); k: spread over devices 0 .. k-1 (single containers
    round-robin in creation order, the members of modelstate_init_ensemble in blocks)."""
    _ok(_L().spd_set_device_placement(int(n_devices)), "set_device_placement")


def modelstate_init_on(device):
    c = C.c_int64()
    _ok(_L().spd_modelstate_init_on(C.byref(c), int(device)), "modelstate_init_on")
    return c.value


def modelstate_device(state_cnt):
    d = C.c_int32()
    _ok(_L().spd_modelstate_device(int(state_cnt), C.byref(d)), "modelstate_device")
    return d.value


def broadcast_boundary(state_cnts, root=0):
    """The shared boundary fields of container state_cnts[root] into all the others, device to device."""
    s, n = _cnts(state_cnts)
    _ok(_L().spd_broadcast_boundary(s, n, int(root)), "broadcast_boundary")


def broadcast_boundary_stats():
    """What the last broadcast_boundary did: (copies that crossed to another device one by one, copies that stayed on a device,
    other GPUs reached by the one collective RCCL broadcast -- 0 when the transport was point-to-point)"""
    peer, local, coll = C.c_int32(), C.c_int32(), C.c_int32()
    _ok(_L().spd_broadcast_boundary_stats(C.byref(peer), C.byref(local), C.byref(coll)), "broadcast_boundary_stats")
    return peer.value, local.value, coll.value


def broadcast_boundary_note():
    """... and in words: the transport, with the reason when the collective was not used (spd_broadcast_boundary_note)."""
    return (_L().spd_broadcast_boundary_note() or b"").decode()


def driver_trace(on=True):
    _ok(_L().spd_driver_trace(int(bool(on))), "driver_trace")


def driver_trace_read():
    """[(kind, group), ...] in host order: 1 = step + check of a device model enqueued, 2 = waiting for it, 3 = codes back."""
    n = _L().spd_driver_trace_read(None, 0)
    buf = (C.c_int32 * (2 * max(n, 1)))()
    n = _L().spd_driver_trace_read(buf, n)
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]


def modelstate_init_sst_anom(state_cnt, n_months):
    _ok(_L().spd_modelstate_init_sst_anom(int(state_cnt), int(n_months)), "modelstate_init_sst_anom")


def modelstate_close(state_cnt):
    _ok(_lib.lib().spd_modelstate_close(int(state_cnt)), "modelstate_close")


def create_datetime(year, month, day, hour, minute):
    c = C.c_int64()
    _ok(_lib.lib().spd_create_datetime(int(year), int(month), int(day), int(hour), int(minute), C.byref(c)), "create_datetime")
    return c.value


def get_datetime(cnt):
    v = [C.c_int32() for _ in range(5)]
    _ok(_lib.lib().spd_get_datetime(int(cnt), *[C.byref(x) for x in v]), "get_datetime")
    return tuple(x.value for x in v)


def close_datetime(cnt):
    _ok(_lib.lib().spd_close_datetime(int(cnt)), "close_datetime")


def controlparams_init(start_cnt, end_cnt):
    c = C.c_int64()
    _ok(_lib.lib().spd_controlparams_init(C.byref(c), int(start_cnt), int(end_cnt)), "controlparams_init")
    return c.value


def controlparams_close(control_cnt):
    _ok(_lib.lib().spd_controlparams_close(int(control_cnt)), "controlparams_close")


def get_model_datetime(control_cnt):
    """ControlParams_t%model_datetime (year, month, day, hour, minute) and month_idx of a control container."""
    now, midx = (C.c_int32 * 5)(), C.c_int32()
    _ok(_lib.lib().spd_controlparams_get_model_datetime(int(control_cnt), now, C.byref(midx)), "get_model_datetime")
    return tuple(now), midx.value


# --------------------------------------------------------------------------------------------------------------------
# model control
# --------------------------------------------------------------------------------------------------------------------
def init(state_cnt, control_cnt):
    """initialize_state (initialization.f90:13-91) from the boundary fields already stored with set_<v>."""
    code = C.c_int32(0)
    _ok(_L().spd_init(int(state_cnt), int(control_cnt), C.byref(code)), "init")
    return code.value


def init_ensemble(state_cnts, control_cnts):
    """Extension: `init` for every container; the members of a device model that nobody has initialised yet are initialised
    together (spd_init_ensemble).  Returns the int32 codes."""
    s, n = _cnts(state_cnts)
    c, nc = _cnts(control_cnts)
    if n != nc:
        raise ValueError("init_ensemble: one control container per state container")
    codes = np.zeros(n, dtype=np.int32)
    _ok(_L().spd_init_ensemble(s, c, codes.ctypes.data_as(C.POINTER(C.c_int32)), n), "init_ensemble")
    return codes


def step(state_cnt, control_cnt):
    """do_single_step (speedy.f90:20-74) followed by the range check of diagnostics.f90."""
    code = C.c_int32(0)
    _ok(_L().spd_step(int(state_cnt), int(control_cnt), C.byref(code)), "step")
    return code.value


def parallel_step(state_cnts, control_cnts):
    s, n = _cnts(state_cnts)
    c, nc = _cnts(control_cnts)
    if n != nc:
        raise ValueError("parallel_step: one control container per state container")
    codes = np.zeros(n, dtype=np.int32)
    _ok(_L().spd_parallel_step(s, c, codes.ctypes.data_as(C.POINTER(C.c_int32)), n), "parallel_step")
    return codes


# Extension: the same step with the range check overlapped.  parallel_step_begin enqueues the step and its check and returns
# a token; parallel_step_end(token) waits for that check only.  A loop that begins step k + 1 before ending step k never
# leaves the GPU waiting for the host (the synchronous form costs ~0.1 ms per step at 64 members); the price is that the
# error code of step k is seen after step k + 1 has been enqueued.  At most two steps may be in flight per model.
_pending_sizes = {}


def parallel_step_begin(state_cnts, control_cnts):
    s, n = _cnts(state_cnts)
    c, nc = _cnts(control_cnts)
    if n != nc:
        raise ValueError("parallel_step: one control container per state container")
    token = C.c_int64()
    _ok(_L().spd_parallel_step_begin(s, c, n, C.byref(token)), "parallel_step_begin")
    _pending_sizes[token.value] = n
    return token.value


def parallel_step_end(token):
    codes = np.zeros(_pending_sizes.pop(int(token)), dtype=np.int32)
    _ok(_lib.lib().spd_parallel_step_end(int(token), codes.ctypes.data_as(C.POINTER(C.c_int32))), "parallel_step_end")
    return codes


# Extension: n_steps steps as ONE call (spd_parallel_steps_begin / _end): the stretch of a time loop in which no callback is due.
# Every device model takes them as one multi-step device call -- member groups on streams of their own, large ensembles in rounds
# -- and the range check of every step is recorded on the device.  parallel_steps_end(token) -> (codes, steps_done): per member the
# code of the FIRST step whose check failed (0: none) and the steps it completed before that one.
def parallel_steps_begin(state_cnts, control_cnts, n_steps):
    s, n = _cnts(state_cnts)
    c, nc = _cnts(control_cnts)
    if n != nc:
        raise ValueError("parallel_steps: one control container per state container")
    token = C.c_int64()
    _ok(_L().spd_parallel_steps_begin(s, c, n, int(n_steps), C.byref(token)), "parallel_steps_begin")
    _pending_sizes[token.value] = n
    return token.value


def parallel_steps_end(token):
    n = _pending_sizes.pop(int(token))
    codes, done = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    _ok(_lib.lib().spd_parallel_steps_end(int(token), codes.ctypes.data_as(C.POINTER(C.c_int32)), done.ctypes.data_as(C.POINTER(C.c_int32))),
        "parallel_steps_end")
    return codes, done


def check(state_cnt):
    code = C.c_int32(0)
    _ok(_L().spd_check(int(state_cnt), C.byref(code)), "check")
    return code.value


def on_default_streams(state_cnts):
    """True when, on every device the given containers live on, torch's current stream is the default (null) stream.  The models
    are stepped on blocking streams of the library's own, which order themselves against the null stream and against no other:
    work a host enqueues for a state that is still being computed (speedy._act_ahead) is in order only there."""
    import torch
    order, groups = _group_by_model(state_cnts)
    devices = {groups[k][0].sp.device for k in order}
    return all(torch.cuda.current_stream(d) == torch.cuda.default_stream(d) for d in devices)


def ensemble_check(state_cnts):
    """Extension: `check` for many containers at once -- ONE range check per device model instead of one per container (each of
    which checks its whole model and picks its member: 64 launches and waits for a 64-member ensemble).  int32 codes in the order
    of `state_cnts`; nothing is printed: a caller that finds a failure asks `check` of that container for the reference's report."""
    import torch
    order, groups = _group_by_model(state_cnts)
    codes = np.zeros(len(state_cnts), dtype=np.int32)
    for k in order:
        model, positions, members = groups[k]
        with torch.cuda.device(model.sp.device):
            codes[positions] = model.check(time_level=1)[members]  # (time level 1, as spd_check)
    return codes


def transform_spectral2grid(state_cnt):
    _ok(_L().spd_transform_spectral2grid(int(state_cnt)), "transform_spectral2grid")


def transform_grid2spectral(state_cnt):
    _ok(_L().spd_transform_grid2spectral(int(state_cnt)), "transform_grid2spectral")


def apply_grid_filter(state_cnt):
    _ok(_L().spd_apply_grid_filter(int(state_cnt)), "apply_grid_filter")


# --------------------------------------------------------------------------------------------------------------------
# extensions on the batched device model behind a container
# --------------------------------------------------------------------------------------------------------------------
def device_model(state_cnt):
    """(EnsembleModel view of the batched device model the container belongs to, member index of the container).  The
    model stays owned by the library; the binding changes when parallel_step gathers or splits containers."""
    import torch
    from .model import EnsembleModel
    handle, member, members = C.c_void_p(), C.c_int32(), C.c_int32()
    _ok(_L().spd_driver_model(int(state_cnt), C.byref(handle), C.byref(member), C.byref(members)), "device_model")
    n_months = _shape("sst_anom", state_cnt)[2] - 2
    return EnsembleModel.borrowed(handle, members.value, torch.device("cuda", modelstate_device(state_cnt)),
                                  max(n_months, 1)), member.value


def _group_by_model(state_cnts):
    """(order of first appearance, {model handle: (EnsembleModel view, positions in state_cnts, member indices)}): one
    spd_driver_model per container, one view per device model (a 64-member export asks once per simulated day)"""
    lib = _L()
    handle, member, members = C.c_void_p(), C.c_int32(), C.c_int32()
    groups, order = {}, []
    for pos, cnt in enumerate(state_cnts):
        _ok(lib.spd_driver_model(int(cnt), C.byref(handle), C.byref(member), C.byref(members)), "device_model")
        key = handle.value
        if key not in groups:
            groups[key] = (device_model(cnt)[0], [], [])
            order.append(key)
        groups[key][1].append(pos)
        groups[key][2].append(member.value)
    return order, groups


def ensemble_device_view(state_cnts, name, spectral2grid=False):
    """Extension: the registry variable `name` of the given containers as ONE device tensor [member, *reversed reference
    shape] in the order of `state_cnts`.  Zero-copy when the containers are, in that order, all the members of one device
    model; otherwise (the two models of 32 or more containers, several GPUs, a subset) the views of the models are gathered
    into a new tensor on the device of the first container.  `spectral2grid=True` refreshes the grid-space variables of every
    model involved first (one batched transform per model).  For on-device post-processing such as ensemble statistics."""
    import torch
    models, where = {}, []
    for cnt in state_cnts:
        model, member = device_model(cnt)
        key = model._m.value
        if key not in models:
            if spectral2grid:
                model.spectral2grid()
            models[key] = model.device_view(name)
        where.append((key, member))
    if len(models) == 1 and [m for _, m in where] == list(range(next(iter(models.values())).shape[0])):
        return next(iter(models.values()))
    device = models[where[0][0]].device
    return torch.stack([models[key][member].to(device) for key, member in where])


def ensemble_grid_arrays(state_cnts, names, narrow=False):
    """Extension: the grid-space variables `names` of the given containers (one container: of ALL members of the batched model
    it belongs to), after one batched spectral2grid per device model: dict name -> float64 array [member, (lev,) lat, lon]
    in the order of `state_cnts` (one device-to-host copy per variable and device model).  narrow=True: as a Dataset of the
    reference carries them (speedy.py:415-477) -- float32, vertical levels bottom-up --, narrowed and turned on the GPU: half the
    bytes cross PCIe and the host makes no pass of its own over them (IEEE rounding to nearest either way: the same values)."""
    import torch

    def fetch(model, name):
        v = model.device_view(name)
        if narrow:
            v = (v.flip(1) if v.ndim == 4 else v).to(torch.float32)
        return v.cpu().numpy()
    if np.ndim(state_cnts) == 0:
        model, _ = device_model(state_cnts)
        model.spectral2grid()
        return {n: fetch(model, n) for n in names}
    order, groups = _group_by_model(state_cnts)
    for k in order:
        groups[k][0].spectral2grid()
    if len(order) == 1 and groups[order[0]][2] == list(range(groups[order[0]][0].nmembers)):
        return {n: fetch(groups[order[0]][0], n) for n in names}  # (every member of ONE model, in order: the copy itself)
    out = {}
    for n in names:
        parts = {k: fetch(groups[k][0], n) for k in order}
        first = parts[order[0]]
        out[n] = np.empty((len(state_cnts),) + first.shape[1:], dtype=first.dtype)
        for k in order:
            out[n][groups[k][1]] = parts[k][groups[k][2]]
    return out


_export_stages = {}  # (device, slot) -> uint8 staging tensor on that device (grown on demand)
_export_copy_streams = {}  # device -> the stream the copies to pinned memory of a packed export run on when the caller does not wait


def _export_stage(pool, device, nbytes, slot=0):
    """the device staging area of a packed export: the process-wide one of (device, slot), or -- `pool`, the caller's own `buffers`
    dict -- that caller's (two exporters of one run must not pack into the same area: with wait=False the copy out of it comes later)"""
    import torch
    pool, key = (_export_stages, (device, slot)) if pool is None else (pool, ("stage", device, slot))
    have = pool.get(key)
    if have is None or have.numel() < nbytes:
        have = pool[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return have


def _export_copy_stream(device):
    import torch
    if device not in _export_copy_streams:
        _export_copy_streams[device] = torch.cuda.Stream(device=device)  # (non-blocking: it does not order itself against the step's streams)
    return _export_copy_streams[device]


class _CopyOut:
    """The copy of a packed output from its device staging area to the pinned buffer, for whoever writes the file: `synchronize()`
    waits for the pack kernels, THEN hands the copy to the runtime on a stream that has nothing pending, and waits for it.
    The order matters.  A hipMemcpyAsync enqueued behind a pending command of its stream (the wait for the pack kernels) is carried
    out by the runtime's blit kernel, and a kernel that writes host memory holds up every other kernel on the GPU for as long as
    it runs -- the next steps of the time loop lost the copy's whole 0.9 ms per simulated day at 64 members, with any number of
    workgroups (profiles/r06_export_copy_out.txt); on an idle stream the same call goes to an SDMA engine and costs the steps
    nothing.  Waiting first needs a host thread: the file writer's (callbacks.XarrayExporter), not the time loop's."""

    def __init__(self, device, packed, buf, stage, pieces, side):
        self.device, self.packed, self.buf, self.stage, self.pieces, self.side = device, packed, buf, stage, pieces, side

    def synchronize(self):
        import torch
        if self.pieces is None:
            return
        pieces, self.pieces = self.pieces, None
        self.packed.synchronize()
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            for start, nbytes in pieces:
                self.buf[start:start + nbytes].copy_(self.stage[start:start + nbytes], non_blocking=True)
        self.side.synchronize()


_export_buffers = {}  # (bytes, slot) -> pinned uint8 tensor, kept for the life of the process (a day's output of 64 members: 48 MB)


def ensemble_export_arrays(state_cnts, names, slot=0, buffers=None, wait=True):
    """Extension, for writing files: the grid-space variables `names` of the given containers as they go into a NetCDF-3 file --
    float32, BIG-endian, vertical levels bottom-up (the reference's export convention, speedy.py:415-477) -- dict name -> numpy
    array of dtype '>f4' [member, (lev,) lat, lon] in the order of `state_cnts`.  Narrowing, level reversal and byte order happen
    on the GPU; what crosses PCIe is the file's payload itself (half the bytes of the fp64 fields), into pinned host memory.
    The arrays alias a pinned buffer that the next call with the same `slot` overwrites: the process-wide one of that slot, or --
    `buffers`, a dict the caller owns (an exporter that writes its files in the background keeps two of its own) -- buffers[slot]
    (the device staging areas of that caller live in the same dict).
    wait=False: returns (arrays, copies) as soon as the transforms and the pack kernels are ENQUEUED -- nothing has been copied
    yet.  Every item of `copies` has a `synchronize()` that waits for the pack kernels, copies that device model's part of the
    payload to the pinned buffer and waits for it (_CopyOut: from a thread of the caller's, e.g. the one that writes the file,
    while the time loop goes on); the arrays hold the payload when all of them have returned.  The device staging area is per
    slot then: a slot must not be asked for again before that."""
    import torch
    order, groups = _group_by_model(state_cnts)
    n = len(state_cnts)
    first = groups[order[0]][0]
    # (from the registry, not from a device view: spd_model_device_ptr drops what the model derived from its state -- the day's
    # interpolated climatologies, the look-ahead geopotential -- and a packed export never writes the state)
    shapes = {name: tuple(reversed(first.shape(name)[1])) for name in names}
    sizes = {name: 4 * n * int(np.prod(shapes[name])) for name in names}
    total = sum(sizes.values())
    pool, key = (_export_buffers, (total, slot)) if buffers is None else (buffers, slot)
    if key not in pool or pool[key].numel() != total:
        pool[key] = torch.empty(total, dtype=torch.uint8, pin_memory=True)  # (allocated pinned: .pin_memory() would allocate twice and copy)
    buf = pool[key]
    offsets, at = {}, 0
    for name in names:
        offsets[name] = at
        at += sizes[name]
    events, asynchronous = [], not wait
    for k in order:
        model, positions, members = groups[k]
        model.spectral2grid()
        contiguous = positions == list(range(positions[0], positions[0] + len(positions)))
        run = members == list(range(members[0], members[0] + len(members)))  # consecutive members of the model, in order
        if contiguous and run:
            # the usual case (all members of a model, in order): one kernel per variable forms the payload on the device
            # (spd_model_export_pack), one copy takes it to the pinned buffer
            # (the staging area is laid out like the pinned buffer: the device models of a call write to disjoint parts of it, so
            # the copies of one -- which, unwaited for, run on another stream -- are not overtaken by the pack kernels of the next)
            stage = _export_stage(buffers, model.sp.device, total, slot if not wait else 0)
            with torch.cuda.device(model.sp.device):
                here = torch.cuda.current_stream()
                stream = C.c_void_p(here.cuda_stream)
                pieces = []
                for name in names:
                    per_member = sizes[name] // n
                    nbytes = per_member * len(members)
                    start = offsets[name] + positions[0] * per_member
                    _ok(_L().spd_model_export_pack(model._m, name.encode(), members[0], len(members),
                                                   C.c_void_p(stage.data_ptr() + start), nbytes, stream), "export_pack")
                    if wait:
                        buf[start:start + nbytes].copy_(stage[start:start + nbytes], non_blocking=True)
                    else:
                        pieces.append((start, nbytes))
                if not wait:  # the copies behind the pack kernels, on a stream that does not hold up what the caller enqueues next
                    packed = torch.cuda.Event()
                    packed.record(here)
                    pieces.sort()
                    merged = [list(pieces[0])]
                    for start, nbytes in pieces[1:]:  # (a model that holds every member in order: ONE copy of the whole payload)
                        if start == merged[-1][0] + merged[-1][1]:
                            merged[-1][1] += nbytes
                        else:
                            merged.append([start, nbytes])
                    events.append(_CopyOut(model.sp.device, packed, buf, stage, merged, _export_copy_stream(model.sp.device)))
            continue
        asynchronous = False  # (a selection of members in another order goes through torch ops on the current stream: waited for)
        index = torch.as_tensor(members, device=model.sp.device)
        for name in names:
            v = model.device_view(name).index_select(0, index)
            if v.ndim == 4:
                v = v.flip(1)  # lev increasing with height
            be = v.to(torch.float32).contiguous().view(torch.uint8).view(-1, 4).flip(1)  # byte order of the file
            per_member = sizes[name] // n
            if contiguous:
                start = offsets[name] + positions[0] * per_member
                buf[start:start + len(positions) * per_member].copy_(be.reshape(-1), non_blocking=True)
            else:
                flat = be.reshape(len(positions), per_member)
                for row, pos in enumerate(positions):
                    buf[offsets[name] + pos * per_member:offsets[name] + (pos + 1) * per_member].copy_(flat[row], non_blocking=True)
    if not asynchronous:
        for k in order:
            torch.cuda.synchronize(groups[k][0].sp.device)
        events = []
    host = buf.numpy()
    arrays = {name: host[offsets[name]:offsets[name] + sizes[name]].view(">f4").reshape((n,) + shapes[name]) for name in names}
    return arrays if wait else (arrays, events)


def whole_device_model(state_cnts):
    """True when the containers are, in order, all the members of ONE device model (what ensemble_export_tensors needs)"""
    order, groups = _group_by_model(state_cnts)
    return len(order) == 1 and groups[order[0]][2] == list(range(groups[order[0]][0].nmembers))


def ensemble_export_tensors(state_cnts, names):
    """Extension, for hooks that KEEP outputs: the grid-space variables `names` of the given containers as a Dataset of the
    reference carries them (float32, vertical levels bottom-up) in DEVICE tensors of their own -- dict name -> torch.float32
    [member, (lev,) lat, lon] -- or None when the containers are not, in order, all the members of one device model (the caller
    then takes the host path).  Only enqueues (one transform, one pack kernel per variable, one pass that turns the file's byte
    order back into the host's): the tensors may be asked for while the steps that lead to this state are still running on the
    device (speedy._act_ahead), and a day of 64 members costs 48 MB of the GPU's 288 GB until somebody reads it."""
    import torch
    order, groups = _group_by_model(state_cnts)
    if len(order) != 1:
        return None
    model, positions, members = groups[order[0]]
    if members != list(range(model.nmembers)):
        return None  # (as whole_device_model)
    n = len(state_cnts)
    shapes = {name: tuple(reversed(model.shape(name)[1])) for name in names}
    sizes = {name: 4 * n * int(np.prod(shapes[name])) for name in names}
    with torch.cuda.device(model.sp.device):
        packed = torch.empty(sum(sizes.values()), dtype=torch.uint8, device=model.sp.device)
        model.spectral2grid()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        at, where = 0, {}
        for name in names:
            _ok(_L().spd_model_export_pack(model._m, name.encode(), 0, n, C.c_void_p(packed.data_ptr() + at), sizes[name], stream),
                "export_pack")
            where[name] = at
            at += sizes[name]
        native = packed.view(-1, 4).flip(1).contiguous().view(torch.float32)  # (big-endian, as a file wants it -> the host's order)
    return {name: native[where[name] // 4:(where[name] + sizes[name]) // 4].view((n,) + shapes[name]) for name in names}


def driver_stats(state_cnt=0):
    """(device models alive, members in the model of `state_cnt`)"""
    alive, members = C.c_int32(), C.c_int32()
    _ok(_lib.lib().spd_driver_stats(int(state_cnt), C.byref(alive), C.byref(members)), "driver_stats")
    return alive.value, members.value


# --------------------------------------------------------------------------------------------------------------------
# registry access: get_<v>, set_<v>, get_<v>_shape, is_array_<v> (generated per variable in the reference)
# --------------------------------------------------------------------------------------------------------------------
def _shape(name, state_cnt):
    shp, nd = (C.c_int32 * 5)(), C.c_int32()
    _ok(_L().spd_get_shape(int(state_cnt), name.encode(), shp, C.byref(nd)), "get_%s_shape" % name)
    return tuple(shp[:nd.value])


def _ctype_of(v):
    """element type the C boundary uses for a registry entry: logicals and integers travel as int32"""
    return np.int32 if v.dtype in (np.bool_, np.int32) else v.dtype


def _get(name, state_cnt, n_months=None):
    v = R.REGISTRY[name]
    shape = _shape(name, state_cnt) if v.shape is not None else ()
    buf = np.zeros(int(np.prod(shape)) if shape else 1, dtype=_ctype_of(v))
    _ok(_L().spd_get(int(state_cnt), name.encode(), buf.ctypes.data_as(C.c_void_p), buf.nbytes), "get_" + name)
    if v.shape is None:
        return v.dtype(buf[0]).item()
    return buf.reshape(shape, order="F")


def _set(name, state_cnt, value, n_months=None):
    v = R.REGISTRY[name]
    if v.shape is None:
        a = np.array([v.dtype(value)], dtype=_ctype_of(v))
    else:
        a = np.asarray(value, dtype=v.dtype)
        if a.shape != _shape(name, state_cnt):
            raise ValueError("Array shape missmatch")
        a = np.ascontiguousarray(a.ravel(order="F"))
    _ok(_L().spd_set(int(state_cnt), name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes), "set_" + name)


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
