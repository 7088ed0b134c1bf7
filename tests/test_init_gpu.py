"""GPU tier: the boundary-field preprocessing of init (land_model_init + sea_model_init with fill_missing_values and
check_surface_fields) runs on the device, one workgroup per monthly plane and member (land_sea_init_kernel, surface.hip).  Bit for bit:
  * against what the REFERENCE's init made of the same boundary-field sets (tests/golden/init.npz; oracle/init_cases.py has
    the sets: the example file, and the example with missing-value patterns the example itself never contains);
  * against the oracle (pinned on the same golden, tests/test_oracle_init.py) on random sets: random masks and fractions
    around the thresholds, random holes, whole rows without a valid point -- including the first row visited, whose mean is
    carried in from the previous plane, and, for the first plane of all, from the start value 0."""
import numpy as np
import pytest

import init_cases

pytestmark = pytest.mark.gpu

INPUTS = [name for name, _ in init_cases.BC_MAP] + ["sst_anom"]


@pytest.fixture(scope="module")
def bc(golden_dir):
    return np.load(golden_dir + "/../../pyspeedy_amd/data/example_bc.npz")


def init_members(spectral, sets):
    from pyspeedy_amd.model import EnsembleModel
    ens = EnsembleModel(spectral, len(sets))
    ens.init_sst_anom(init_cases.N_MONTHS)
    for member, fields in enumerate(sets):
        for name in INPUTS:
            ens.set(name, fields[name], member)
    ens.init(init_cases.START)
    return ens


def test_init_preprocessing_matches_the_reference_bitwise(spectral, bc, golden_dir):
    gold = np.load(golden_dir + "/init.npz")
    cases = ["example", "holes", "example", "holes", "holes"]  # (several workgroups, not in blocks)
    ens = init_members(spectral, [init_cases.CASES[c](bc) for c in cases])
    for member, case in enumerate(cases):
        for name in init_cases.OUTPUTS:
            got, ref = ens.get(name, member), gold[case + "_" + name]
            assert got.shape == ref.shape, (case, name)
            assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), (member, case, name)
    assert (ens.check(2) == 0).all()
    ens.close()


def random_set(bc, seed):
    rng = np.random.default_rng(seed)
    f = init_cases.example(bc)
    ix, il = init_cases.IX, init_cases.IL
    f32 = np.float32
    # masks and fractions on a coarse lattice of values that contains every threshold of the two routines, and their neighbours
    marks = np.array([0.0, 0.05, float(f32(0.1)), 0.1, float(np.nextafter(f32(0.1), f32(1))), 0.25, float(f32(1) / f32(3)), 1 / 3,
                      0.5, 2 / 3, 1.0 - float(f32(1) / f32(3)), 0.9, 1.0 - float(f32(0.1)), 0.95, 1.0])
    f["fmask_orig"][:] = marks[rng.integers(0, marks.size, (ix, il))]
    f["alb0"][:] = np.array([0.1, float(f32(0.4)), 0.4, float(np.nextafter(f32(0.4), f32(0))), 0.6])[rng.integers(0, 5, (ix, il))]
    f["veg_high"][:] = rng.uniform(-0.5, 1.0, (ix, il))
    f["veg_low"][:] = rng.uniform(-0.5, 1.0, (ix, il))
    f["soil_wc_l1"][:] = rng.uniform(0.0, 1.5, (ix, il, 12))
    f["soil_wc_l2"][:] = rng.uniform(0.0, 0.6, (ix, il, 12))
    f["snowd12"][:] = rng.uniform(-10.0, 400.0, (ix, il, 12))
    f["sea_ice_frac12"][:] = np.where(rng.random((ix, il, 12)) < 0.2, -0.0, rng.uniform(-0.5, 1.0, (ix, il, 12)))
    f["sst_anom"][:] = rng.standard_normal(f["sst_anom"].shape)
    for name, lo, hi in (("stl12", 220.0, 310.0), ("sst12", 270.0, 303.0)):
        a = rng.uniform(lo, hi, (ix, il, 12))
        a[rng.random((ix, il, 12)) < rng.uniform(0.0, 0.6)] = -999.0
        rows = rng.random((il, 12)) < 0.15  # rows without a valid point
        a[:, rows] = -1.0
        a[:, 23, rng.integers(0, 12)] = -1.0  # ... the first row visited among them
        f[name] = np.asfortranarray(a)
    if seed % 2:  # the first row of the first plane of the land sequence: takes the start value of the running mean, 0
        f["stl12"][:, 23, 0] = -1.0
    f["sst12"][:, 23, 0] = -1.0  # the first row of the sea sequence: takes the land sequence's last mean
    return f


def test_init_preprocessing_matches_the_oracle_on_random_sets(spectral, bc, oracle):
    sets = [random_set(bc, seed) for seed in range(11, 17)]
    ens = init_members(spectral, sets)
    for member, fields in enumerate(sets):
        ref, _ = oracle.land_sea_init(fields)
        for name in init_cases.OUTPUTS:
            got = ens.get(name, member)
            assert np.array_equal(got.view(np.uint64), ref[name].view(np.uint64)), (member, name)
        # the holes are gone, and the rows that had no valid point hold one value on the points inside the mask
        assert (ens.get("stl12", member) >= 0).all() and (ens.get("sst12", member) >= 0).all()
    ens.close()
