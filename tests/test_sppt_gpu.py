"""Deterministic SPPT (csrc/sppt.hip).  PARITY UNPINNED w.r.t. the reference (its sppt.f90 is compiled out and cannot work as
written); checked here: the HIP kernels against the independent numpy restatement (oracle/sppt_oracle.py), the statistics the
scheme is defined by (stationary variance, lag-1 autocorrelation phi, spectrum ~ exp(-L^2 el2 / 2)), reproducibility under
re-sharding of the members, and the tendency formula of physics.f90:234-248 through the physics kernel."""
import os
import sys

import numpy as np
import pytest

import physics_helpers as H  # tests/physics_helpers.py

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="module")
def bc():
    import pyspeedy_amd
    with np.load(pyspeedy_amd.example_bc_file()) as z:
        return {k: z[k] for k in z.files}


def make_model(spectral, bc, members, seed, first=0):
    from pyspeedy_amd.model import EnsembleModel
    m = EnsembleModel(spectral, members)
    m.set_bc(bc)
    m.set_sppt(True, seed=seed, first_member_id=first)
    return m


def test_pattern_matches_the_numpy_restatement(spectral, bc):
    import sppt_oracle as O
    el2 = spectral.table("el2")
    model = make_model(spectral, bc, 3, seed=1234, first=5)
    specs = [None] * 3
    for step in range(3):
        model.run(1)
        for i in range(3):
            specs[i] = O.advance(specs[i], el2, 1234, 5 + i, step)
            got = model.get("sppt_spec", i)  # (31, 32, 8) Fortran order: [m, n, k]
            ref = specs[i].reshape(8, 32, 31).transpose(2, 1, 0)
            scale = np.abs(ref).max()
            assert np.abs(got - ref).max() <= 1e-12 * scale, (step, i)
    # grid-space image = spec2grid of the spectral pattern (kcos = 1)
    import torch
    spec = torch.from_numpy(np.ascontiguousarray(model.get("sppt_spec", 1).transpose(2, 1, 0))).cuda()
    grid = spectral.spec2grid(spec, kcos=1).cpu().numpy()  # [8, 48, 96]
    got = model.get("sppt_pattern", 1).transpose(2, 1, 0)
    assert np.abs(got - grid).max() <= 1e-13 * np.abs(grid).max()


def test_statistics_and_resharding(spectral, bc):
    import sppt_oracle as O
    phi, f0, q = O.constants()
    assert abs(phi - np.exp(-1.0 / 9.0)) < 1e-15
    model = make_model(spectral, bc, 16, seed=7)
    series = []
    for _ in range(40):
        model.run(1)
        series.append(np.stack([model.get("sppt_spec", i) for i in range(16)]))
    s = np.stack(series)  # [time, member, m, n, k]
    el2 = spectral.table("el2").reshape(32, 31).T  # [m, n]
    sigma2 = (f0 * np.exp(-q * el2)) ** 2
    keep = sigma2 > 1e-8 * sigma2.max()
    # stationary variance of each coefficient: E|r|^2 = 2 sigma^2 / (1 - phi^2)  (re and im are independent)
    var = (np.abs(s) ** 2).mean(axis=(0, 1, 4))
    ratio = var[keep] / (2.0 * sigma2[keep] / (1.0 - phi * phi))
    assert 0.85 < ratio.mean() < 1.15 and abs(np.median(ratio) - 1.0) < 0.2
    # lag-1 autocorrelation = phi
    num = (s[1:] * np.conj(s[:-1])).real.sum(axis=(0, 1, 4))
    den = (np.abs(s[:-1]) ** 2).sum(axis=(0, 1, 4))
    ac = (num[keep] / den[keep])
    assert abs(np.average(ac, weights=sigma2[keep]) - phi) < 0.03
    # the first step is already drawn from the stationary distribution
    v0 = (np.abs(s[0]) ** 2).mean(axis=(0, 3))
    assert 0.7 < (v0[keep] / (2.0 * sigma2[keep] / (1.0 - phi * phi))).mean() < 1.3
    # grid-space pattern: clipped to [-1, 1] when applied; raw standard deviation of the order of the nominal 0.33
    grid = np.stack([model.get("sppt_pattern", i) for i in range(16)])
    assert 0.1 < grid.std() < 1.0
    # same global member ids -> same noise, however the ensemble is split: members 8..11 of this model == a 4-member shard
    shard = make_model(spectral, bc, 4, seed=7, first=8)
    shard.run(1)
    for i in range(4):
        np.testing.assert_array_equal(shard.get("sppt_spec", i), series[0][8 + i])
    other = make_model(spectral, bc, 1, seed=8)
    other.run(1)
    assert np.abs(other.get("sppt_spec", 0) - series[0][0]).max() > 0


def test_tendency_formula_through_the_physics_kernel(spectral):
    """spd_physics with and without a pattern on the same inputs: out = (1 + clip(r)) (out0 - dyn) + dyn, level by level
    for T and q, at the lowest level for u and v (physics.f90:239-246)."""
    import torch
    import sppt_oracle as O
    import pyspeedy_amd.physics as P
    phys = P.ColumnPhysics(spectral)
    base = H.synthetic_member(seed=11)
    M = 2
    dev = lambda n: torch.from_numpy(P.to_device_layout(base[n])).cuda()[None].expand(M, *P.shapes(1)[n][1:]).contiguous()
    fields = {n: dev(n) for n in P.STATE_IN_3D + P.STATE_IN_2D}
    forcing = {n: dev(n) for n in P.SURFACE_IN + P.SHORTWAVE_IN}
    rng = np.random.default_rng(3)
    pattern = torch.from_numpy(rng.normal(0.0, 0.6, (M, 8, 48, 96))).cuda()  # some values beyond +-1: clipped
    outs = []
    for pat in (None, pattern):
        tend = {n: dev(n) for n in P.TENDENCIES}
        dyn = {n: t.clone() for n, t in tend.items()}
        st = P.PhysicsState(M, spectral.device)
        phys(fields, tend, forcing, st, True, 0.3, sppt_pattern=pat)
        outs.append({n: t.cpu().numpy() for n, t in tend.items()})
    pat = pattern.cpu().numpy()
    dynh = {n: t.cpu().numpy() for n, t in dyn.items()}
    for n in ("ttend", "qtend"):
        ref = O.perturb(outs[0][n], dynh[n], pat)
        assert np.abs(outs[1][n] - ref).max() <= 1e-12 * np.abs(ref).max(), n
    for n in ("utend", "vtend"):
        ref = outs[0][n].copy()
        ref[:, 7] = O.perturb(outs[0][n][:, 7], dynh[n][:, 7], pat[:, 7])
        assert np.abs(outs[1][n] - ref).max() <= 1e-12 * np.abs(ref).max(), n


def test_model_runs_with_sppt_and_members_diverge(spectral, bc):
    model = make_model(spectral, bc, 4, seed=99)
    model.run(36)
    assert (model.check(2) == 0).all()
    t = [model.get("t", i) for i in range(4)]
    assert np.abs(t[0] - t[1]).max() > 1e-6  # identical initial states, different noise
    plain = make_model(spectral, bc, 1, seed=0)
    plain.set_sppt(False)
    plain.run(36)
    d = np.abs(plain.get("t", 0) - t[0]).max() / np.abs(t[0]).max()
    assert 1e-8 < d < 1e-1


def test_sppt_ensembles_step_in_member_groups_bitwise(spectral, bc):
    """The generator is keyed by global member ids, so a group of members advances exactly its own part of the pattern: an
    ensemble stepped as three member groups on three streams leaves bitwise the state of the same ensemble stepped as one."""
    grouped, serial = make_model(spectral, bc, 33, seed=5, first=100), make_model(spectral, bc, 33, seed=5, first=100)
    assert grouped.config()["chunks"] == 3
    serial.set_option("member_groups", 1)
    for m in (grouped, serial):
        m.set_physics_precision(True)  # cfg 5 as bench.py runs it
        m.run(7)
        m.run(1)  # (a call of one step is issued serially in both)
        m.run(4)
    for name in ("t", "vor", "tr", "ps", "sppt_spec", "sppt_pattern", "olr"):
        for member in (0, 16, 17, 32):
            assert np.array_equal(grouped.get(name, member), serial.get(name, member)), name
    assert (grouped.check(2) == 0).all()
    c1, c2 = grouped.control(), serial.control()
    assert c1.sppt_step == c2.sppt_step == 12  # (SPPT was switched on after the two steps of first_step)
    grouped.close()
    serial.close()
