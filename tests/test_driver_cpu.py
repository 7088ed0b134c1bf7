"""CPU tier: the outer C boundary (include/pyspeedy_amd_driver.h) -- what works without a device: the registry table
compiled into the library equals pyspeedy_amd/registry.py (itself checked against the reference's model_state.json in
test_facade_cpu.py), datetime / control containers, argument checking."""
import ctypes as C

import numpy as np

DTYPES = {0: np.float64, 1: np.complex128, 2: np.float32, 3: np.int32, 4: np.bool_}


def registry_of(lib):
    name = C.create_string_buffer(32)
    dt, nd, ro = C.c_int32(), C.c_int32(), C.c_int32()
    shape = (C.c_int32 * 5)()
    count = lib.spd_registry_entry(0, name, C.byref(dt), C.byref(nd), shape, C.byref(ro))
    out = {}
    for i in range(count):
        assert lib.spd_registry_entry(i, name, C.byref(dt), C.byref(nd), shape, C.byref(ro)) == count
        out[name.value.decode()] = (DTYPES[dt.value], tuple(shape[:nd.value]), bool(ro.value))
    return out


def test_compiled_registry_equals_the_python_registry(hip_lib):
    import pyspeedy_amd.registry as R
    reg = registry_of(hip_lib)
    assert set(reg) == set(R.REGISTRY)
    for n, v in R.REGISTRY.items():
        dtype, shape, read_only = reg[n]
        assert np.dtype(dtype) == np.dtype(v.dtype), n
        want = () if v.shape is None else tuple(-1 if s == R.N_MONTHS else s for s in v.shape)
        assert shape == want, (n, shape, want)
        assert read_only == (v.where == "table"), n
        flag = C.c_int32(-1)
        assert hip_lib.spd_is_array(n.encode(), C.byref(flag)) == 0 and bool(flag.value) == R.is_array(n)
    assert hip_lib.spd_is_array(b"no_such_variable", C.byref(C.c_int32())) < 0


def test_datetime_and_control_containers(hip_lib):
    L = hip_lib
    d0, d1, ctl = C.c_int64(), C.c_int64(), C.c_int64()
    assert L.spd_create_datetime(1982, 1, 30, 0, 0, C.byref(d0)) == 0
    assert L.spd_create_datetime(1982, 3, 1, 0, 0, C.byref(d1)) == 0
    assert d0.value != d1.value
    got = [C.c_int32() for _ in range(5)]
    assert L.spd_get_datetime(d1, *[C.byref(g) for g in got]) == 0
    assert [g.value for g in got] == [1982, 3, 1, 0, 0]
    assert L.spd_controlparams_init(C.byref(ctl), d0, d1) == 0
    now, midx = (C.c_int32 * 5)(), C.c_int32()
    assert L.spd_controlparams_get_model_datetime(ctl, now, C.byref(midx)) == 0
    assert list(now) == [1982, 1, 30, 0, 0] and midx.value == 1  # initialize_control: model date = start date
    assert L.spd_close_datetime(d0) == 0 and L.spd_close_datetime(d1) == 0
    assert L.spd_get_datetime(d0, *[C.byref(g) for g in got]) < 0  # closed
    assert L.spd_controlparams_init(C.byref(ctl), d0, d1) < 0
    assert L.spd_controlparams_close(ctl) == 0
    assert L.spd_controlparams_get_model_datetime(ctl, now, None) < 0


def test_calls_on_dead_containers_fail_cleanly(hip_lib):
    L = hip_lib
    code = C.c_int32(7)
    assert L.spd_step(12345, 67890, C.byref(code)) < 0
    assert L.spd_check(12345, C.byref(code)) < 0
    assert L.spd_transform_spectral2grid(12345) < 0
    buf = np.zeros(4)
    assert L.spd_get(12345, b"olr", buf.ctypes.data_as(C.c_void_p), buf.nbytes) < 0
    assert L.spd_modelstate_close(12345) == 0  # closing twice is harmless, as deallocate on a freed container is not tested upstream
    assert b"container" in L.spd_last_error()


def test_multi_device_interface_without_a_device(hip_lib):
    """The one-process-several-GPUs extension on a machine that has no GPU: the device count is 0, no placement beyond it is
    accepted, naming a device fails with a message (never a CPU fallback), and the trace / broadcast entry points check their
    arguments.  (The machine with a GPU runs tests/test_driver_gpu.py::test_device_placement_and_boundary_broadcast.)"""
    import os
    import pytest
    if os.path.exists("/dev/kfd"):
        pytest.skip("needs a machine without a GPU")
    L = hip_lib
    n = C.c_int32(-1)
    assert L.spd_device_count(C.byref(n)) == 0 and n.value == 0
    assert L.spd_device_count(None) < 0
    assert L.spd_set_device_placement(0) == 0 and L.spd_set_device_placement(1) < 0 and L.spd_set_device_placement(-1) < 0
    cnt = C.c_int64(0)
    assert L.spd_modelstate_init_on(C.byref(cnt), 0) < 0 and b"no such HIP device" in L.spd_last_error()
    assert L.spd_modelstate_init(C.byref(cnt)) < 0  # no device, no state: there is no CPU fallback
    dev = C.c_int32()
    assert L.spd_modelstate_device(4242, C.byref(dev)) < 0
    ids = (C.c_int64 * 2)(4242, 4243)
    assert L.spd_broadcast_boundary(ids, 2, 5) < 0 and L.spd_broadcast_boundary(ids, 2, 0) < 0
    assert L.spd_driver_trace(1) == 0 and L.spd_driver_trace_read(None, 0) == 0 and L.spd_driver_trace(0) == 0
    codes = (C.c_int32 * 2)()
    assert L.spd_parallel_step(ids, ids, codes, 2) < 0 and b"container" in L.spd_last_error()
    assert L.spd_parallel_step(ids, ids, codes, 0) == 0  # an empty ensemble steps trivially
