"""Host-side pieces of the pySPEEDY-compatible facade that need no GPU: registry, the `_speedy` function-name surface,
date containers, the NetCDF-3 dataset layer (checked against the reference's own fixture file) and callback gating."""
import os
import tempfile
from datetime import datetime

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_registry_covers_the_reference_state():
    from pyspeedy_amd import registry as R
    # ModelState_t holds 112 entries (registry/model_state_def.py:121-495); the three polymorphic module instances
    # (mod_geometry, mod_spectral, mod_implicit) have no getters in the reference either
    assert len(R.REGISTRY) == 109 + 2  # + tcorh, qcorh: spectral corrections kept in the state by this implementation
    for name in ("vor", "div", "t", "tr", "ps", "phi", "phis", "u_grid", "ps_grid", "sst_anom", "current_step", "lon", "fband",
                 "snowcv", "sstom12", "rad_tau2", "hfluxn", "increase_co2"):
        assert name in R.REGISTRY, name
    assert R.shape_of("vor") == (31, 32, 8, 2) and R.REGISTRY["vor"].dtype == np.complex128
    assert R.shape_of("sst_anom", n_months=4) == (96, 48, 6)
    assert R.shape_of("current_step") == () and not R.is_array("current_step")
    assert R.REGISTRY["lat"].dtype == np.float32
    assert [R.REGISTRY[v].alt_name for v in R.DEFAULT_OUTPUT_VARS] == ["u", "v", "t", "q", "phi", "ps"]


def test_model_state_def_has_the_layout_of_the_references_json():
    """`pyspeedy.speedy.MODEL_STATE_DEF` is the reference's data/model_state.json; here the same mapping is generated from the
    registry.  Where the reference is at hand (build container) every entry is compared with it: Fortran type, dimension string,
    run-length dimension, export name and export dimensions, and the units of the default output variables."""
    import json
    import pyspeedy_amd
    from pyspeedy_amd import registry as R
    from pyspeedy_amd.error_codes import ERROR_CODES
    mine = pyspeedy_amd.MODEL_STATE_DEF
    assert set(mine) == set(R.REGISTRY) and mine["u_grid"]["alt_name"] == "u" and mine["sst_anom"]["time_dim"] == "n_months"
    assert set(mine["vor"]) == {"dtype", "dims", "desc", "time_dim", "units", "nc_dims", "alt_name", "std_name"}
    assert ERROR_CODES[0] == "Run successful." and "range" in ERROR_CODES[-2] and "Unexpected" in ERROR_CODES[-77]
    assert pyspeedy_amd.example_sst_anomaly_file().endswith("sst_anomaly.nc")
    path = "/root/reference/pyspeedy/data/model_state.json"
    if not os.path.isfile(path):
        pytest.skip("the reference is not on this machine")
    with open(path) as fh:
        ref = json.load(fh)
    squash = lambda s: None if s is None else s.replace(" ", "")
    for name, r in ref.items():
        if name in ("mod_geometry", "mod_spectral", "mod_implicit"):
            continue  # (polymorphic module instances: no getters in the reference either)
        m = mine[name]
        assert m["dtype"] == r["dtype"].replace("real(p)", "real(8)"), name
        want = squash(r["dims"])
        if want is not None:
            for sym, val in (("aux_dim", "3"), ("t_levs", "2"), ("ntr", "1"), ("100:400", "301")):
                want = want.replace(sym, val)
            want = want.replace(",1)", ")") if name == "tr" else want  # (tr: the single tracer's axis is dropped here)
        assert squash(m["dims"]) == want, (name, m["dims"], r["dims"])
        assert m["time_dim"] == r["time_dim"] and m["alt_name"] == r["alt_name"] and m["std_name"] == r["std_name"], name
        if name in R.DEFAULT_OUTPUT_VARS:
            assert m["nc_dims"] == r["nc_dims"] and m["units"] == r["units"], name


def test_driver_function_names():
    from pyspeedy_amd import registry as R
    from pyspeedy_amd import speedy_driver as drv
    for fn in ("modelstate_init", "modelstate_init_sst_anom", "modelstate_close", "create_datetime", "get_datetime",
               "close_datetime", "controlparams_init", "controlparams_close", "init", "step", "parallel_step", "check",
               "transform_spectral2grid", "transform_grid2spectral", "apply_grid_filter"):
        assert callable(getattr(drv, fn)), fn
    for name in R.REGISTRY:
        for pat in ("get_%s", "set_%s", "get_%s_shape", "is_array_%s"):
            assert callable(getattr(drv, pat % name)), pat % name
    assert drv.is_array_t() and not drv.is_array_air_absortivity_co2()
    with pytest.raises(AttributeError):
        drv.get_no_such_variable
    assert getattr(drv, "get_no_such_variable", None) is None  # what Speedy.__getitem__ relies on


def test_date_and_control_containers():
    from pyspeedy_amd import speedy_driver as drv
    a = drv.create_datetime(1982, 1, 1, 0, 0)
    b = drv.create_datetime(1982, 3, 5, 12, 40)
    assert drv.get_datetime(b) == (1982, 3, 5, 12, 40)
    c = drv.controlparams_init(a, b)
    assert c not in (a, b)
    drv.controlparams_close(c)
    drv.close_datetime(a)
    with pytest.raises(ValueError):
        drv.get_datetime(a)
    with pytest.raises(ValueError):
        drv.step(12345678, 1)  # unknown state container


def test_month_window_of_the_sst_anomalies():
    from pyspeedy_amd.speedy import _add_months
    assert _add_months(datetime(1982, 1, 1), -1) == datetime(1981, 12, 1)
    assert _add_months(datetime(1982, 12, 1), 1) == datetime(1983, 1, 1)
    assert _add_months(datetime(1982, 6, 1), 7) == datetime(1983, 1, 1)


def test_dataset_roundtrip_and_reference_fixture():
    from pyspeedy_amd.dataset import Dataset, Variable, assert_allclose, concat, open_dataset
    ref = open_dataset(os.path.join(GOLD, "reference_fixtures", "1982-01-02_0000.nc"))
    assert set(ref.keys()) == {"u", "v", "t", "q", "phi", "ps"}
    assert ref["u"].dims == ("time", "lev", "lat", "lon") and ref["ps"].dims == ("time", "lat", "lon")
    assert ref["time"].values[0] == np.datetime64("1982-01-02T00:00:00")
    assert ref["u"].attrs == {"units": "m/s", "long_name": "eastward_wind", "standard_name": "u_grid"}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "copy.nc")
        ref.to_netcdf(path)
        back = open_dataset(path)
    assert_allclose(back, ref, rtol=0, atol=0)
    for name in ref.variables:
        assert back[name].attrs == ref[name].attrs, name
    # ensemble / time concatenation
    def member(i, when):
        return Dataset({"x": Variable(("time", "ens", "lat"), np.full((1, 1, 3), float(i)))},
                       {"ens": Variable(("ens",), np.array([i], dtype=np.int32)),
                        "time": Variable(("time",), np.array([np.datetime64(when, "s")])),
                        "lat": Variable(("lat",), np.arange(3, dtype=np.float32))})
    ens = concat([member(2, "1982-01-02"), member(0, "1982-01-02"), member(1, "1982-01-02")], "ens")
    assert ens["x"].shape == (1, 3, 3) and list(ens["ens"].values) == [0, 1, 2]
    assert float(ens.sel(ens=2)["x"].values[0, 0]) == 2.0
    series = concat([member(0, "1982-01-02T00:40:00"), member(0, "1982-01-02T00:00:00")], "time")
    with tempfile.TemporaryDirectory() as tmp:
        series.to_netcdf(os.path.join(tmp, "s.nc"))
        s2 = open_dataset(os.path.join(tmp, "s.nc"))
    assert list(s2["time"].values) == [np.datetime64("1982-01-02T00:00:00"), np.datetime64("1982-01-02T00:40:00")]
    with pytest.raises(AssertionError):
        assert_allclose(ens, series)


def test_callback_gating():
    from pyspeedy_amd.callbacks import BaseCallback, ModelCheckpoint, XarrayExporter

    class Fake:
        def __init__(self, step, date):
            self.step, self.current_date = step, date

        def get_current_step(self):
            return self.step

    cb = BaseCallback(interval=36, spinup_date=datetime(1982, 1, 3))
    assert cb.skip_flag(Fake(36, datetime(1982, 1, 2)))        # still spinning up
    assert cb.skip_flag(Fake(73, datetime(1982, 1, 3, 0, 40)))  # not a multiple of the interval
    assert not cb.skip_flag(Fake(72, datetime(1982, 1, 3)))
    assert BaseCallback().interval == 1 and cb.copy() is not cb
    assert XarrayExporter().variables == ("u_grid", "v_grid", "t_grid", "q_grid", "phi_grid", "ps_grid")
    assert ModelCheckpoint(interval=4).history_interval == 4


def test_example_bc_is_packaged():
    import pyspeedy_amd
    with np.load(pyspeedy_amd.example_bc_file()) as z:
        assert z["orog"].shape == (96, 48) and z["sst"].shape == (96, 48, 12)
        assert {"orog", "lsm", "alb", "vegh", "vegl", "stl", "snowd", "swl1", "swl2", "swl3", "sst", "icec"} <= set(z.files)


def test_dataset_prints_what_it_holds():
    import numpy as np
    from pyspeedy_amd.dataset import Dataset
    d = Dataset({"t": (("time", "lev", "lat", "lon"), np.full((2, 8, 48, 96), 280.0, np.float32))},
                {"lon": (("lon",), np.arange(96.0) * 3.75), "time": (("time",), np.array(["1980-01-01", "1980-01-02"], dtype="datetime64[s]"))},
                {"title": "x"})
    text = repr(d)
    assert "Dimensions:" in text and "time: 2" in text and "lon: 96" in text
    assert "t            (time, lev, lat, lon) float32  280 .. 280" in text and "1980-01-02T00:00:00" in text and "Attributes: title" in text


def test_dataset_reductions_of_the_references_ensemble_notebook():
    """Ensemble_forecast.ipynb post-processes the checkpoint dataframe with `var(dim="ens").mean(dim=[...]).apply(np.sqrt)` and
    `std(dim="ens").apply(np.sqrt).isel(lev=0)` and copies attributes variable by variable."""
    import numpy as np
    from pyspeedy_amd.dataset import Dataset
    rng = np.random.default_rng(3)
    t = rng.normal(280.0, 5.0, (3, 4, 8, 6, 12)).astype(np.float32)
    ps = rng.normal(1e5, 500.0, (3, 4, 6, 12)).astype(np.float32)
    ds = Dataset({"t": (("time", "ens", "lev", "lat", "lon"), t, {"units": "K"}), "ps": (("time", "ens", "lat", "lon"), ps, {"units": "Pa"})},
                 {"time": (("time",), np.arange(3)), "ens": (("ens",), np.arange(4)), "lev": (("lev",), np.linspace(0.95, 0.025, 8)),
                  "lat": (("lat",), np.arange(6.0)), "lon": (("lon",), np.arange(12.0))})
    spr = ds.var(dim="ens").mean(dim=["lev", "lat", "lon"]).apply(np.sqrt)
    assert list(spr) == ["t", "ps"] and spr["t"].dims == ("time",) and set(spr.coords) == {"time"}
    assert np.allclose(spr["t"].values, np.sqrt(t.astype(np.float64).var(axis=1).mean(axis=(1, 2, 3))))
    assert np.allclose(spr["ps"].values, np.sqrt(ps.astype(np.float64).var(axis=1).mean(axis=(1, 2))))
    for var in spr:
        spr[var].attrs.update(**ds[var].attrs)
    assert spr["t"].attrs["units"] == "K"
    low = ds.std(dim="ens").apply(np.sqrt).isel(lev=0)
    assert low["t"].dims == ("time", "lat", "lon") and np.allclose(low["t"].values, np.sqrt(t.astype(np.float64).std(axis=1))[:, 0])
    assert low["ps"].dims == ("time", "lat", "lon")


def test_file_loader_keeps_small_files_only(tmp_path):
    """speedy._load_fields: the boundary file read last is kept (the reference's per-member set_bc re-reads it), a large record -- an
    SST-anomaly file -- is not, so that it pins no host memory after set_bc; arrays from files are float64, Fortran order, read-only."""
    from pyspeedy_amd import speedy as S
    small = S._load_fields(S.example_bc_file())
    assert S._last_file[0] is not None and S._last_file[0][0] == os.path.realpath(S.example_bc_file())
    again = S._load_fields(S.example_bc_file())
    assert again["orog"] is small["orog"] and small["orog"].dtype == np.float64 and small["orog"].flags.f_contiguous
    assert not small["orog"].flags.writeable
    with pytest.raises(ValueError):
        small["orog"][0, 0] = 1.0
    big = tmp_path / "ssta_record.npz"
    months = S._CACHE_LIMIT_BYTES // (96 * 48 * 8) + 8
    np.savez(big, ssta=np.zeros((96, 48, months), dtype=np.float32), time=np.arange(months).astype("datetime64[M]").astype("datetime64[s]"))
    rec = S._load_fields(str(big))
    assert rec["ssta"].shape == (96, 48, months) and rec["ssta"].dtype == np.float64
    assert S._last_file[0][0] == os.path.realpath(S.example_bc_file())  # (the small file is still the one that is kept)
    assert S._load_fields(str(big))["ssta"] is not rec["ssta"]
    assert S._load_fields({"a": 1}) == {"a": 1}
    with pytest.raises(RuntimeError):
        S._load_fields(str(tmp_path / "missing.npz"))


def test_what_is_due_at_a_step():
    """The time loops ask every hook's gate ONCE per step (speedy._callbacks_due): a hook with the reference's gating contributes its
    `fire` when it is due, a plain callable or a hook that overrides __call__ always acts through its own __call__."""
    from pyspeedy_amd.callbacks import BaseCallback
    from pyspeedy_amd.speedy import _callbacks_due

    class Fake:
        current_date = datetime(1982, 1, 5)
        asked = 0

        def get_current_step(self):
            Fake.asked += 1
            return 72

    class Counts(BaseCallback):
        fired = 0

        def fire(self, model_instance):
            Counts.fired += 1

    class OwnCall(BaseCallback):
        called = 0

        def __call__(self, model_instance):
            OwnCall.called += 1

    seen = []
    hooks = [Counts(interval=36), Counts(interval=5), OwnCall(interval=1000), seen.append, Counts(interval=36, spinup_date=datetime(1982, 2, 1))]
    model = Fake()
    due = _callbacks_due(hooks, model)
    assert len(due) == 3 and Fake.asked == 2  # (the gate of the hook that is still spinning up never gets to the step counter)
    for act in due:
        act(model)
    assert Counts.fired == 1 and OwnCall.called == 1 and seen == [model]
    assert _callbacks_due([], model) == []


def _hdf5_like_file(tmp_path):
    """a file that starts like the reference's boundary files do (NetCDF-4 is HDF5: the 8-byte signature)"""
    path = tmp_path / "example_bc.nc"
    path.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    return path


def test_the_references_hdf5_boundary_files_are_read_through_whatever_reader_is_there(tmp_path, monkeypatch):
    """pyspeedy/speedy.py:277 reads the boundary conditions with xr.load_dataset(bc_file, engine="netcdf4"): the reference's packaged
    example_bc.nc is NetCDF-4 / HDF5.  `_load_fields` recognises the signature and reads through netCDF4, h5py or xarray --
    whichever can be imported; none is a dependency.  Here: a stand-in `h5py` on sys.path (the other two made unimportable)."""
    import sys
    import textwrap
    from pyspeedy_amd import speedy as S
    fake = tmp_path / "fake_readers"
    fake.mkdir()
    (fake / "h5py.py").write_text(textwrap.dedent('''
        import numpy as np
        class Dataset:
            def __init__(self, values, attrs=None):
                self._values, self.attrs = values, dict(attrs or {})
            def __getitem__(self, key):
                assert key is Ellipsis
                return self._values
        class Group:
            pass
        class File:
            opened = []
            def __init__(self, path, mode):
                assert mode == "r"
                File.opened.append(path)
            def __enter__(self):
                return self
            def __exit__(self, *exc):
                return False
            def items(self):
                yield "orog", Dataset(np.full((96, 48), 3.0, dtype=np.float32))
                yield "sst", Dataset(np.arange(96 * 48 * 12, dtype=np.float32).reshape(96, 48, 12))
                yield "time", Dataset(np.array([0.0, 31.0]), {"units": b"days since 1982-01-01 00:00:00"})
                yield "a_group", Group()
    '''))
    monkeypatch.syspath_prepend(str(fake))
    for name in ("netCDF4", "xarray"):
        monkeypatch.setitem(sys.modules, name, None)  # (import of a name mapped to None raises ImportError)
    monkeypatch.delitem(sys.modules, "h5py", raising=False)
    monkeypatch.setattr(S, "_last_file", [None, None])
    path = _hdf5_like_file(tmp_path)
    fields = S._load_fields(str(path))
    assert set(fields) == {"orog", "sst", "time"}
    assert fields["orog"].dtype == np.float64 and fields["orog"].flags.f_contiguous and float(fields["orog"][5, 7]) == 3.0
    assert fields["sst"].shape == (96, 48, 12) and fields["sst"][1, 0, 0] == 48 * 12
    assert fields["time"][1] == np.datetime64("1982-02-01")
    import h5py
    assert h5py.File.opened == [str(path)]
    # ... and a NetCDF-3 file with the same name ending still goes through the package's own reader
    from pyspeedy_amd.dataset import Dataset, Variable
    classic = tmp_path / "classic.nc"
    Dataset({"orog": Variable(("lon", "lat"), np.ones((96, 48)))}, {}).to_netcdf(str(classic))
    assert not S._is_hdf5(str(classic)) and S._load_fields(str(classic))["orog"].shape == (96, 48)


def test_an_hdf5_file_without_any_reader_names_the_converter(tmp_path, monkeypatch):
    import sys
    from pyspeedy_amd import speedy as S
    for name in ("netCDF4", "h5py", "xarray"):
        monkeypatch.setitem(sys.modules, name, None)
    monkeypatch.setattr(S, "_last_file", [None, None])
    path = _hdf5_like_file(tmp_path)
    with pytest.raises(RuntimeError) as failure:
        S._load_fields(str(path))
    text = str(failure.value)
    assert "NetCDF-4 / HDF5" in text and "tools/convert_bc.py" in text and str(path) in text
    assert all(name in text for name in ("netCDF4", "h5py", "xarray"))
    assert os.path.isfile(os.path.join(os.path.dirname(GOLD), "..", "tools", "convert_bc.py"))


def test_an_xarray_dataset_is_accepted_for_the_sst_anomalies():
    """pyspeedy/speedy.py:321-331 takes a path or an xr.Dataset for the SST anomalies: anything shaped like one (a `variables`
    mapping of objects with `.values`) is read without importing xarray."""
    from pyspeedy_amd import speedy as S

    class Var:
        def __init__(self, values):
            self.values = values

    class FakeXarrayDataset:
        def __init__(self):
            self.variables = {"ssta": Var(np.zeros((96, 48, 3))), "time": Var(np.array(["1981-12-01", "1982-01-01", "1982-02-01"],
                                                                                      dtype="datetime64[ns]"))}
    fields = S._load_fields(FakeXarrayDataset())
    assert set(fields) == {"ssta", "time"} and fields["ssta"].shape == (96, 48, 3) and fields["time"].dtype.kind == "M"


def test_the_stretches_of_the_time_loops_follow_the_hooks_schedule():
    """Speedy.run / SpeedyEns.run hand the steps between two due callbacks to the device as one call.  The schedule is known for
    hooks with the reference's gating (BaseCallback: interval / spinup_date); anything else -- a plain callable, a hook that
    overrides __call__ or skip_flag, an interval that is not a positive integer -- may act at every step: no stretches then."""
    from datetime import timedelta
    from pyspeedy_amd import speedy as S
    from pyspeedy_amd.callbacks import BaseCallback, DiagnosticCheck, ModelCheckpoint, XarrayExporter

    class Own(BaseCallback):
        def skip_flag(self, model_instance):
            return False

    class Caller(BaseCallback):
        def __call__(self, model_instance):
            pass

    assert S._hook_intervals([]) == []
    assert S._hook_intervals([DiagnosticCheck(36), XarrayExporter(interval=6), ModelCheckpoint(interval=7)]) == [36, 6, 7]
    for odd in ([lambda m: None], [Own(interval=3)], [Caller(interval=3)], [BaseCallback(interval=0)], [BaseCallback(interval=2.0)],
                [BaseCallback(interval=True)], [DiagnosticCheck(36), print]):
        assert S._hook_intervals(odd) is None, odd
    start, dt = datetime(1982, 1, 1), timedelta(minutes=40)
    # from step 0: the next multiple of every interval, the end of the run, or ten model days
    assert S._stretch(0, [36, 7], start, start + 100 * dt) == 7
    assert S._stretch(7, [36, 7], start + 7 * dt, start + 100 * dt) == 7 and S._stretch(35, [36, 7], start, start + 100 * dt) == 1
    assert S._stretch(36, [36], start, start + 20 * dt) == 20 and S._stretch(0, [], start, start + 1000 * dt) == S._MAX_STRETCH == 360
    assert S._stretch(0, [36], start, start + timedelta(minutes=50)) == 2  # (an end date between two steps: the loop runs past it, as upstream)
    assert S._stretch(5, [36], start, start) == 1  # (never less than a step: the loop's own condition ends the run)
    # a hook that is still spinning up does not end a stretch: the first multiple of its interval whose date is not before its
    # spinup_date does (checked against the hook's own gate, step by step)
    spin = start + 100 * dt + timedelta(minutes=10)  # between steps 100 and 101: step 108 is the first of interval 36 that may act
    assert S._stretch(0, [36], start, start + 500 * dt, [spin]) == 108 and S._stretch(0, [36, 50], start, start + 500 * dt, [spin, None]) == 50
    assert S._stretch(100, [36], start + 100 * dt, start + 500 * dt, [spin]) == 8
    assert S._stretch(108, [36], start + 108 * dt, start + 500 * dt, [spin]) == 36
    assert S._stretch(0, [36], start, start + 500 * dt, [start + 72 * dt]) == 72  # (a date that IS a step's date: that step acts)
    assert S._hook_spinups([DiagnosticCheck(36), ModelCheckpoint(interval=7, spinup_date=spin)]) == [None, spin]
    hook = ModelCheckpoint(interval=36, spinup_date=spin)

    class At:
        def __init__(self, step):
            self.step, self.current_date = step, start + step * dt

        def get_current_step(self):
            return self.step
    step, acted = 0, []
    while step < 300:
        k = S._stretch(step, [36], start + step * dt, start + 300 * dt, [spin])
        assert not any(not hook.skip_flag(At(s)) for s in range(step + 1, step + k)), (step, k)  # nothing due inside the stretch
        step += k
        if not hook.skip_flag(At(step)):
            acted.append(step)
    assert acted == [108, 144, 180, 216, 252, 288]
    # hooks may leave what no longer needs the state to the time loop; called by hand it happens at once
    done = []

    class Late(BaseCallback):
        def fire(self, model_instance):
            done.append("fire")
            return lambda: done.append("rest")

    class Model:
        current_date = start

        def get_current_step(self):
            return 4

    Late(interval=2)(Model())
    assert done == ["fire", "rest"]
    rest = []
    S._act([Late(interval=2).fire, lambda m: done.append("plain")], Model(), rest)
    assert done == ["fire", "rest", "fire", "plain"] and len(rest) == 1
    S._do_rest(rest)
    assert done[-1] == "rest" and rest == []


def test_a_failing_writer_does_not_replace_the_runs_own_exception():
    """_finish_all runs in the `finally` of the time loops: every hook gets its finish(); when the run itself failed, that is the
    exception that travels on (the writer's is attached to it), when it did not, the writer's is raised."""
    from pyspeedy_amd import speedy as S

    class Hook:
        def __init__(self, fail):
            self.fail, self.finished, self._in_run = fail, 0, True

        def finish(self):
            self.finished += 1
            if self.fail:
                raise OSError("disk full")

    hooks = [Hook(True), Hook(False)]
    with pytest.raises(OSError):
        S._finish_all(hooks)
    assert [h.finished for h in hooks] == [1, 1] and not any(h._in_run for h in hooks)
    hooks = [Hook(True), Hook(False)]
    with pytest.raises(RuntimeError) as failure:
        try:
            raise RuntimeError("the model left the accepted range")
        finally:
            S._finish_all(hooks, [lambda: None])
    assert isinstance(failure.value.__context__, OSError) and [h.finished for h in hooks] == [1, 1]
    # what a hook left to the loop is done first, every item of it, and a failure there does not keep the writers from being joined
    hooks, done = [Hook(False)], []

    def broken():
        raise ValueError("header")
    with pytest.raises(ValueError):
        S._finish_all(hooks, [broken, lambda: done.append(1)])
    assert done == [1] and hooks[0].finished == 1


def test_hooks_that_only_enqueue_device_work_act_while_the_stretch_is_still_running(monkeypatch):
    """speedy._act_ahead: the hooks due at the end of a stretch fire before the host waits for it when ALL of them declare
    (`acts_ahead`) that they only enqueue device work; one that needs the state on the host keeps every hook behind the wait, in
    the order given.  XarrayExporter declares it only for what it writes in the background of a run that owns it."""
    from pyspeedy_amd import speedy as S
    from pyspeedy_amd.callbacks import BaseCallback, DiagnosticCheck, XarrayExporter
    done = []

    class Ahead(BaseCallback):
        def acts_ahead(self, model_instance):
            return True

        def fire(self, model_instance):
            done.append("enqueued")
            return lambda: done.append("written")

    class Model:
        n_members = 64

    assert S._act_ahead([], Model()) == []  # (nothing due: nothing to wait with)
    left = S._act_ahead([Ahead().fire, Ahead().fire], Model())
    assert done == ["enqueued", "enqueued"] and len(left) == 2
    S._do_rest(left)
    assert done[2:] == ["written", "written"]
    del done[:]
    assert S._act_ahead([Ahead().fire, DiagnosticCheck(36).fire], Model()) is None and done == []
    assert S._act_ahead([lambda m: None], Model()) is None
    # (the model's steps run on streams that order themselves against the null stream only: ahead of them only from there)
    from pyspeedy_amd import speedy_driver
    asked, on_default = [], [True]
    monkeypatch.setattr(speedy_driver, "on_default_streams", lambda cnts: asked.append(list(cnts)) or on_default[0])

    class Member:
        def __init__(self, cnt):
            self._state_cnt = cnt
    Model.__iter__ = lambda self: iter([Member(7), Member(8)])
    exporter = XarrayExporter(interval=36)
    assert not exporter.acts_ahead(Model())  # called by hand it writes inside the call
    exporter._in_run = True
    assert exporter.acts_ahead(Model()) and asked == [[7, 8]]
    on_default[0] = False
    assert not exporter.acts_ahead(Model())
    on_default[0] = True
    Model.n_members = 1
    assert exporter.acts_ahead(Model())  # (a single model's file is written by the time loop itself, behind the next stretch)
    Model.n_members = 64
    assert not XarrayExporter(background=False).acts_ahead(Model())
    by_hand = XarrayExporter(background=True)
    assert not by_hand.acts_ahead(Model())  # (outside a run there is no time loop to be ahead of)


def test_model_checkpoint_joins_its_snapshots_when_the_series_is_read():
    """callbacks.ModelCheckpoint keeps the snapshots as they come and joins them when `dataframe` is read (the reference merges at
    every output, callbacks.py:175-180: the same Dataset in the end); dataset.concat orders by the coordinate, and a series that is
    in order already is not copied a second time."""
    from pyspeedy_amd.callbacks import ModelCheckpoint
    from pyspeedy_amd.dataset import Dataset, Variable, concat

    def frame(day, value):
        return Dataset({"t": Variable(("time", "lat"), np.full((1, 3), value, dtype=np.float32))},
                       {"time": Variable(("time",), np.array([np.datetime64("1982-01-%02d" % day, "s")])),
                        "lat": Variable(("lat",), np.arange(3, dtype=np.float32))})

    class Model:
        def __init__(self):
            self.day = 0

        def to_dataframe(self, variables=None):
            self.day += 1
            return frame(self.day, float(self.day))

    class Step:
        """the part of a model the gating asks for, around the fake's to_dataframe"""
        current_date = datetime(1982, 1, 1)

        def __init__(self, model):
            self.to_dataframe = model.to_dataframe

        def get_current_step(self):
            return 0

    keep, model = ModelCheckpoint(interval=1), Model()
    assert keep.dataframe is None
    for _ in range(3):
        keep(Step(model))  # (by hand: what `fire` leaves to be done happens inside the call)
    assert len(keep._frames) == 3
    dropped = keep.fire(model)  # a time loop that finds the state refused by the range check drops what `fire` returned
    assert callable(dropped) and len(keep._frames) == 3
    model.day -= 1
    series = keep.dataframe
    assert series["t"].values.shape == (3, 3) and list(series["t"].values[:, 0]) == [1.0, 2.0, 3.0]
    assert keep.dataframe is series and len(keep._frames) == 1  # (joined once)
    keep(Step(model))
    assert keep.dataframe["t"].values.shape == (4, 3) and keep.copy().dataframe["t"].values.shape == (4, 3)
    keep.dataframe = None
    assert keep.dataframe is None and keep._frames == []
    mixed = concat([frame(3, 3.0), frame(1, 1.0), frame(2, 2.0)], "time")
    assert list(mixed["t"].values[:, 0]) == [1.0, 2.0, 3.0] and list(mixed["time"].values) == sorted(mixed["time"].values)


def test_model_checkpoint_keeps_an_ensembles_snapshots_on_the_device_while_a_run_owns_it(monkeypatch):
    """Inside a run ModelCheckpoint asks a model that offers it for snapshots that stay on the GPU (SpeedyEns.snapshot_on_device:
    only enqueues) -- so it may act ahead of a running stretch --, copies the oldest out when more than `device_bytes` wait there,
    and `dataframe` is the same series as ever.  By hand, with a snapshot() of the user's own, or with device_bytes=0: the host path."""
    from pyspeedy_amd import speedy_driver
    from pyspeedy_amd.callbacks import ModelCheckpoint
    from pyspeedy_amd.dataset import Dataset, Variable
    monkeypatch.setattr(speedy_driver, "on_default_streams", lambda cnts: True)
    whole = [True]
    monkeypatch.setattr(speedy_driver, "whole_device_model", lambda cnts: whole[0])
    resolved = []

    def frame(day):
        return Dataset({"t": Variable(("time", "lat"), np.full((1, 3), float(day), dtype=np.float32))},
                       {"time": Variable(("time",), np.array([np.datetime64("1982-01-%02d" % day, "s")])),
                        "lat": Variable(("lat",), np.arange(3, dtype=np.float32))})

    class Pending:
        def __init__(self, day):
            self.day, self.nbytes = day, 100

        def resolve(self):
            if self.nbytes:
                resolved.append(self.day)
                self.nbytes = 0
            return frame(self.day)

    class Member:
        _state_cnt = 5

    class Ens:
        day, host = 0, 0

        def __iter__(self):
            return iter([Member()])

        def snapshot_on_device(self, variables):
            self.day += 1
            return Pending(self.day)

        def to_dataframe(self, variables=None):
            self.day += 1
            self.host += 1
            return frame(self.day)

    keep, ens = ModelCheckpoint(interval=1, device_bytes=250), Ens()
    assert not keep.acts_ahead(ens)  # by hand: the host path, at once
    keep.fire(ens)()
    assert ens.host == 1 and isinstance(keep._frames[0], Dataset)
    keep._in_run = True
    whole[0] = False
    assert not keep.acts_ahead(ens)  # (spread over several device models: the host path, after the wait)
    whole[0] = True
    assert keep.acts_ahead(ens)
    for _ in range(3):
        keep.fire(ens)()
    assert ens.host == 1 and resolved == [2]  # (three of 100 bytes against 250: the oldest went to the host)
    assert list(keep.dataframe["t"].values[:, 0]) == [1.0, 2.0, 3.0, 4.0] and resolved == [2, 3, 4]

    class Own(ModelCheckpoint):
        def snapshot(self, model_instance):
            return model_instance.to_dataframe()
    own = Own(interval=1)
    own._in_run = True
    assert not own.acts_ahead(ens) and not ModelCheckpoint(device_bytes=0).acts_ahead(ens)
    own.fire(ens)()
    assert ens.host == 2
