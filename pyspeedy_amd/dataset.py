"""A small labelled-array container with NetCDF-3 classic I/O (scipy.io.netcdf_file) for the model output.

The reference exports through xarray + netCDF4 (pyspeedy/speedy.py:415-477, callbacks.py:169-255); neither is a dependency
here, and its own test fixtures are NetCDF-3 classic files, so this module writes that format directly, with the same
variable names, dimension order, dtypes and attributes: float32 data variables on (time, [ens,] lev, lat, lon), int32
`time` ("<unit> since <first time>", proleptic_gregorian) and `ens`, float32 coordinates.
"""
from datetime import datetime, timedelta

import numpy as np


class Variable:
    __slots__ = ("dims", "values", "attrs")

    def __init__(self, dims, values, attrs=None):
        self.dims = tuple(dims)
        self.values = np.asarray(values)
        if self.values.ndim != len(self.dims):
            raise ValueError("dims %s do not match an array of shape %s" % (self.dims, self.values.shape))
        self.attrs = dict(attrs or {})

    @property
    def shape(self):
        return self.values.shape


class Dataset:
    """data_vars / coords: name -> Variable or (dims, values[, attrs])."""

    def __init__(self, data_vars=None, coords=None, attrs=None):
        as_var = lambda v: v if isinstance(v, Variable) else Variable(*v)
        self.data_vars = {k: as_var(v) for k, v in (data_vars or {}).items()}
        self.coords = {k: as_var(v) for k, v in (coords or {}).items()}
        self.attrs = dict(attrs or {})

    def __repr__(self):
        """Dimensions, coordinates and data variables with their dimensions, types and value ranges (what a notebook shows)."""
        def line(name, v):
            vals = np.asarray(v.values)
            rng = ""
            if vals.size and vals.dtype.kind in "fiu":
                rng = "  %.6g .. %.6g" % (np.nanmin(vals), np.nanmax(vals))
            elif vals.size and vals.dtype.kind == "M":
                rng = "  %s .. %s" % (vals.min(), vals.max())
            return "    %-12s (%s) %s%s" % (name, ", ".join(v.dims), vals.dtype, rng)
        out = ["<pyspeedy_amd.Dataset>", "Dimensions:  (" + ", ".join("%s: %d" % kv for kv in self.dims.items()) + ")", "Coordinates:"]
        out += [line(k, v) for k, v in self.coords.items()]
        out += ["Data variables:"] + [line(k, v) for k, v in self.data_vars.items()]
        if self.attrs:
            out.append("Attributes: " + ", ".join(sorted(self.attrs)))
        return "\n".join(out)

    # ---- mapping-style access (what the reference's tests use of xarray) ----
    def keys(self):
        return self.data_vars.keys()

    @property
    def variables(self):
        out = dict(self.coords)
        out.update(self.data_vars)
        return out

    def __contains__(self, name):
        return name in self.data_vars or name in self.coords

    def __getitem__(self, name):
        return self.data_vars[name] if name in self.data_vars else self.coords[name]

    @property
    def dims(self):
        out = {}
        for v in self.variables.values():
            for d, n in zip(v.dims, v.shape):
                out[d] = n
        return out

    # ---- selection / combination ----
    def isel(self, **indexers):
        """Integer selection along dimensions; the selected dimensions are dropped."""
        def cut(v):
            idx = tuple(indexers.get(d, slice(None)) for d in v.dims)
            dims = tuple(d for d in v.dims if d not in indexers)
            return Variable(dims, v.values[idx], v.attrs)
        return Dataset({k: cut(v) for k, v in self.data_vars.items()},
                       {k: cut(v) for k, v in self.coords.items() if k not in indexers}, self.attrs)

    def sel(self, **labels):
        idx = {}
        for d, lab in labels.items():
            hits = np.nonzero(self.coords[d].values == lab)[0]
            if hits.size == 0:
                raise KeyError("%s=%r not found" % (d, lab))
            idx[d] = int(hits[0])
        return self.isel(**idx)

    # ---- reductions (the ones the reference's ensemble notebook applies to the checkpoint dataframe; xarray's defaults:
    # missing values skipped, population variance, attributes of the variables dropped) ----
    def _reduce(self, func, dim):
        dims = (dim,) if isinstance(dim, str) else tuple(dim) if dim is not None else None

        def red(v):
            hit = tuple(i for i, d in enumerate(v.dims) if dims is None or d in dims)
            if not hit or np.asarray(v.values).dtype.kind not in "fiu":
                return Variable(v.dims, v.values, v.attrs)
            return Variable(tuple(d for i, d in enumerate(v.dims) if i not in hit), func(np.asarray(v.values, dtype=np.float64), axis=hit))
        gone = set(self.dims) if dims is None else set(dims)
        return Dataset({k: red(v) for k, v in self.data_vars.items()},
                       {k: v for k, v in self.coords.items() if not gone.intersection(v.dims)}, self.attrs)

    def mean(self, dim=None):
        return self._reduce(np.nanmean, dim)

    def var(self, dim=None):
        return self._reduce(np.nanvar, dim)

    def std(self, dim=None):
        return self._reduce(np.nanstd, dim)

    def apply(self, func):
        """`func` on the values of every data variable (xarray's Dataset.apply / map)."""
        return Dataset({k: Variable(v.dims, func(v.values), v.attrs) for k, v in self.data_vars.items()}, self.coords, self.attrs)

    map = apply

    def __iter__(self):
        return iter(self.data_vars)

    def to_xarray(self):
        """The same data as an `xarray.Dataset` -- what the reference's `to_dataframe()` returns (pyspeedy/speedy.py:415-477) --
        for hosts that have xarray installed (it is not a dependency of this package; ImportError otherwise)."""
        import xarray as xr
        return xr.Dataset({k: (v.dims, v.values, v.attrs) for k, v in self.data_vars.items()},
                          coords={k: (v.dims, v.values, v.attrs) for k, v in self.coords.items()}, attrs=dict(self.attrs))

    def to_netcdf(self, path):
        write_netcdf(self, path)


def concat(datasets, dim):
    """Join datasets that differ only along `dim` (an existing dimension of the data variables), ordered by its coordinate."""
    datasets = list(datasets)
    first = datasets[0]
    labels = np.concatenate([d.coords[dim].values for d in datasets])
    order = np.argsort(labels, kind="stable")
    data = {}
    in_order = bool((order == np.arange(len(order))).all())  # (the usual case, a time series in the making: no second copy)
    for name, v in first.data_vars.items():
        ax = v.dims.index(dim)
        joined = np.concatenate([d.data_vars[name].values for d in datasets], axis=ax)
        data[name] = Variable(v.dims, joined if in_order else np.take(joined, order, axis=ax), v.attrs)
    coords = dict(first.coords)
    coords[dim] = Variable((dim,), labels[order], first.coords[dim].attrs)
    return Dataset(data, coords, first.attrs)


def assert_allclose(a, b, rtol=1e-5, atol=0.0):
    """xr.testing.assert_allclose for these datasets: same variables, dimensions and values within tolerance."""
    if set(a.keys()) != set(b.keys()):
        raise AssertionError("data variables differ: %s vs %s" % (sorted(a.keys()), sorted(b.keys())))
    for name in list(a.keys()) + [c for c in a.coords if c in b.coords]:
        va, vb = a[name], b[name]
        if va.dims != vb.dims or va.shape != vb.shape:
            raise AssertionError("%s: dims %s %s vs %s %s" % (name, va.dims, va.shape, vb.dims, vb.shape))
        if va.values.dtype.kind == "M":
            if not (va.values == vb.values).all():
                raise AssertionError("%s: time coordinates differ" % name)
            continue
        np.testing.assert_allclose(va.values, vb.values, rtol=rtol, atol=atol, err_msg=name)


# --------------------------------------------------------------------------------------------------------------------
# NetCDF-3 classic
# --------------------------------------------------------------------------------------------------------------------
_EPOCH_FMT = "%Y-%m-%d %H:%M:%S"


def _encode_time(values):
    """datetime64 / datetime values -> (int32 offsets, units string): the coarsest unit that keeps the offsets integral."""
    t = np.asarray(values, dtype="datetime64[s]")
    ref = t.min()
    secs = (t - ref).astype(np.int64)
    for unit, n in (("days", 86400), ("hours", 3600), ("minutes", 60), ("seconds", 1)):
        if (secs % n == 0).all():
            break
    ref_dt = ref.astype(datetime)
    return (secs // n).astype(np.int32), "%s since %s" % (unit, ref_dt.strftime(_EPOCH_FMT))


def _decode_time(offsets, units):
    unit, _, ref = units.partition(" since ")
    ref = ref.strip()
    fmt = _EPOCH_FMT if len(ref) > 10 else "%Y-%m-%d"
    ref_dt = datetime.strptime(ref, fmt)
    step = {"days": 86400, "hours": 3600, "minutes": 60, "seconds": 1}[unit]
    return np.array([np.datetime64(ref_dt + timedelta(seconds=int(o) * step), "s") for o in np.atleast_1d(offsets)])


def _prepare(ds):
    """(dims, [(name, dim names, big-endian array, attrs)]) with times encoded and dtypes narrowed as the files carry them."""
    out = []
    for name, v in ds.variables.items():
        vals, attrs = v.values, dict(v.attrs)
        if vals.dtype.kind == "M" or (vals.dtype == object and len(vals) and isinstance(vals.flat[0], datetime)):
            vals, units = _encode_time(vals)
            attrs.update(units=units, calendar="proleptic_gregorian")
        if vals.dtype.kind in "iu":
            vals = vals.astype(">i4")
        elif vals.dtype.kind == "f":
            vals = vals.astype(">f4", copy=False)  # (packed exports arrive in the file's byte order already)
        else:
            raise TypeError("%s: dtype %s cannot be written" % (name, vals.dtype))
        out.append((name, v.dims, vals, {k: a for k, a in attrs.items() if a is not None}))
    return ds.dims, out


def prepare_netcdf(ds):
    """Everything of a NetCDF-3 classic (CDF-1) file that takes Python work -- the header and the list of big-endian blocks behind
    it -- without touching the disk: (header bytes, [(contiguous array, padding bytes)]).  `write_prepared` then only moves bytes,
    which a background thread can do without holding the interpreter lock for more than a few calls."""
    import struct

    NC_DIMENSION, NC_VARIABLE, NC_ATTRIBUTE, NC_CHAR, NC_INT, NC_FLOAT, NC_DOUBLE = 10, 11, 12, 2, 4, 5, 6
    pad = lambda b: b + b"\0" * (-len(b) % 4)
    name = lambda s_: struct.pack(">i", len(s_.encode())) + pad(s_.encode())

    def attributes(attrs):
        if not attrs:
            return struct.pack(">ii", 0, 0)
        out = struct.pack(">ii", NC_ATTRIBUTE, len(attrs))
        for k, a in attrs.items():
            if isinstance(a, str):
                raw = a.encode()
                out += name(k) + struct.pack(">ii", NC_CHAR, len(raw)) + pad(raw)
            else:
                arr = np.atleast_1d(np.asarray(a))
                if arr.dtype.kind in "iu":
                    out += name(k) + struct.pack(">ii", NC_INT, arr.size) + arr.astype(">i4").tobytes()
                elif arr.dtype == np.float32:
                    out += name(k) + struct.pack(">ii", NC_FLOAT, arr.size) + arr.astype(">f4").tobytes()
                else:
                    out += name(k) + struct.pack(">ii", NC_DOUBLE, arr.size) + arr.astype(">f8").tobytes()
        return out

    dims, variables = _prepare(ds)
    dim_ids = {d: i for i, d in enumerate(dims)}
    head = b"CDF\x01" + struct.pack(">i", 0)
    head += struct.pack(">ii", NC_DIMENSION, len(dims)) if dims else struct.pack(">ii", 0, 0)
    for d, n in dims.items():
        head += name(d) + struct.pack(">i", int(n))
    head += attributes(ds.attrs)
    # variable headers need the data offsets, which need the header size: build them with a placeholder first
    def var_headers(begins):
        out = struct.pack(">ii", NC_VARIABLE, len(variables))
        for (vname, vdims, vals, attrs), begin in zip(variables, begins):
            out += name(vname) + struct.pack(">i", len(vdims)) + b"".join(struct.pack(">i", dim_ids[d]) for d in vdims)
            out += attributes(attrs)
            out += struct.pack(">iii", NC_INT if vals.dtype.kind == "i" else NC_FLOAT, vals.nbytes + (-vals.nbytes % 4), begin)
        return out
    size = len(head) + len(var_headers([0] * len(variables)))
    begins = []
    for _, _, vals, _ in variables:
        begins.append(size)
        size += vals.nbytes + (-vals.nbytes % 4)
    if size >= 2 ** 31:
        raise ValueError("dataset too large for the NetCDF-3 classic format (2 GiB offsets)")
    return head + var_headers(begins), [(np.ascontiguousarray(vals), b"\0" * (-vals.nbytes % 4)) for _, _, vals, _ in variables]


def write_prepared(path, prepared):
    header, blocks = prepared
    with open(str(path), "wb") as f:
        f.write(header)
        for vals, padding in blocks:
            vals.tofile(f)
            if padding:
                f.write(padding)


def write_netcdf(ds, path):
    """NetCDF-3 classic (CDF-1) writer: header, then every variable as one contiguous big-endian block.  Written directly
    (one tofile per variable) because it is on the path of every model output; files are read back by scipy / netCDF4 /
    xarray like any other classic file (tests/test_facade_cpu.py compares with scipy.io.netcdf_file)."""
    write_prepared(path, prepare_netcdf(ds))


def open_dataset(path):
    """Read a NetCDF-3 classic file (ours or the reference's fixtures) into a Dataset; `time` is decoded to datetime64."""
    from scipy.io import netcdf_file
    with netcdf_file(str(path), "r", mmap=False) as f:
        coords, data = {}, {}
        for name, nv in f.variables.items():
            attrs = {k: (a.decode() if isinstance(a, bytes) else a) for k, a in nv._attributes.items() if k != "_FillValue"}
            vals = np.array(nv.data).astype(nv.data.dtype.newbyteorder("="))
            if name == "time" and "units" in attrs:
                vals = _decode_time(vals, attrs.pop("units"))
                attrs.pop("calendar", None)
            var = Variable(nv.dimensions, vals, attrs)
            (coords if nv.dimensions == (name,) else data)[name] = var
        gattrs = {k: (a.decode() if isinstance(a, bytes) else a) for k, a in f._attributes.items()}
    return Dataset(data, coords, gattrs)
