"""GPU tier: the fused HIP column-physics kernel (through spd_physics) against the reference's own outputs
(golden snapshots) and against the CPU oracle on synthetic members.

Tolerance: scaled max error <= 1e-12 per output field (fp64; the device uses FMA contraction and its own exp()).
The physics has genuine discontinuities (nint() table index, threshold tests, integer cloud/convection tops): a
last-bit difference can flip one in principle, so integer diagnostics are compared exactly and any flipped column is
reported; none is tolerated on the committed vectors."""
import numpy as np
import pytest
import torch

from test_physics_oracle import load_snapshot

import physics_helpers as H  # tests/physics_helpers.py

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def phys(spectral):
    from pyspeedy_amd.physics import ColumnPhysics
    return ColumnPhysics(spectral)


def run_hip(phys, members, sw, co2, pre=None, diagnostics=True):
    """members: list of dicts in reference host layout; returns (tend dict of tensors, PhysicsState)."""
    import pyspeedy_amd.physics as P
    M = len(members)
    dev = lambda n: torch.from_numpy(np.stack([P.to_device_layout(m[n]) for m in members])).cuda()
    fields = {n: dev("qg_in" if n == "qg" and "qg_in" in members[0] else n) for n in P.STATE_IN_3D + P.STATE_IN_2D}
    tend = {n: dev(n) for n in P.TENDENCIES}
    forcing = {n: dev(n) for n in P.SURFACE_IN + P.SHORTWAVE_IN}
    st = P.PhysicsState(M, phys.device, diagnostics=diagnostics)
    if pre is not None:
        for n, arrs in pre.items():
            getattr(st, n).copy_(torch.from_numpy(np.stack([P.to_device_layout(a) for a in arrs])).cuda())
    phys(fields, tend, forcing, st, sw, co2)
    torch.cuda.synchronize()
    return tend, st


def compare(got, ref, name, tol=TOL):
    scale = max(np.abs(ref).max(), 1e-300)
    err = np.abs(got - ref).max() / scale
    assert err <= tol, "%s: scaled max error %.3e" % (name, err)
    return err


@pytest.mark.parametrize("name", ["physics_sw", "physics_nosw"])
def test_golden_snapshots(phys, golden_dir, name):
    import pyspeedy_amd.physics as P
    inp, pre, ref, sw, co2 = load_snapshot(golden_dir, name)
    pre_dev = {k: [v] for k, v in pre.items()} if not sw else None
    tend, st = run_hip(phys, [inp], sw, co2, pre=pre_dev)
    worst = 0.0
    for k, r in ref.items():
        t = tend[k][0] if k in tend else getattr(st, k)[0]
        got = P.from_device_layout(t)[::3]
        if k == "hfluxn":
            got, r = got[:, :, :2], r[:, :, :2]
        worst = max(worst, compare(got, r, name + "/" + k))
    print("worst scaled error", worst)


@pytest.mark.parametrize("sw", [True, False])
def test_against_oracle_synthetic_members(phys, oracle, sw):
    """Three different synthetic members in one launch (member-major batching), incl. integer diagnostics."""
    import pyspeedy_amd.physics as P
    members = [H.synthetic_member(seed=s) for s in (1, 2, 3)]
    refs, pres = [], {k: [] for k in oracle.PHYS_PERSIST_SHAPES}
    rng = np.random.default_rng(0)
    for m in members:
        o_in = {("qg_in" if k == "qg" else k): v for k, v in m.items()}
        if not sw:  # persisted radiation state from a previous shortwave step of the same member
            prev = oracle.physics(o_in, True, 0.3)
            for k in pres:
                o_in[k] = prev[k]
                pres[k].append(prev[k])
        refs.append(oracle.physics(o_in, sw, 0.3))
    tend, st = run_hip(phys, members, sw, 0.3, pre=None if sw else pres)
    flips = 0
    for i, ref in enumerate(refs):
        for k in P.TENDENCIES:
            compare(P.from_device_layout(tend[k][i]), ref[k], "member%d/%s" % (i, k))
        for k in list(oracle.PHYS_OUT_SHAPES) + list(oracle.PHYS_PERSIST_SHAPES) + list(P.DIAG_F):
            got, r = P.from_device_layout(getattr(st, k)[i]), ref[k]
            if k == "hfluxn":
                got, r = got[:, :, :2], r[:, :, :2]
            if k in ("cloudc", "clstr") and not sw:
                continue  # only defined on shortwave steps
            compare(got, r, "member%d/%s" % (i, k))
        flips += int((P.from_device_layout(st.iptop[i]) != ref["iptop"]).sum())
        if sw:
            flips += int((P.from_device_layout(st.icltop[i]) != ref["icltop"]).sum())
    assert flips == 0, "%d columns changed an integer cloud/convection top" % flips


def test_member_independence_and_determinism(phys):
    """Columns are independent: permuting members permutes results bit for bit; two runs are identical."""
    import pyspeedy_amd.physics as P
    members = [H.synthetic_member(seed=s) for s in (4, 5)]
    t1, s1 = run_hip(phys, members, True, 0.3)
    t2, s2 = run_hip(phys, members[::-1], True, 0.3)
    t3, s3 = run_hip(phys, members, True, 0.3)
    for k in P.TENDENCIES:
        assert torch.equal(t1[k][0], t2[k][1]) and torch.equal(t1[k][1], t2[k][0])
        assert torch.equal(t1[k], t3[k])
    assert torch.equal(s1.rad_tau2[0], s2.rad_tau2[1])


def test_full_ensemble_size_properties(phys):
    """64 members (BASELINE cfg 4 shard on one GPU): all finite, energy-like bounds, and replicated members agree."""
    import pyspeedy_amd.physics as P
    base = H.synthetic_member(seed=9)
    M = 64
    dev = lambda n: torch.from_numpy(P.to_device_layout(base[n])).cuda()[None].expand(M, *P.shapes(1)[n][1:]).contiguous()
    fields = {n: dev(n) for n in P.STATE_IN_3D + P.STATE_IN_2D}
    tend = {n: dev(n) for n in P.TENDENCIES}
    forcing = {n: dev(n) for n in P.SURFACE_IN + P.SHORTWAVE_IN}
    st = P.PhysicsState(M, phys.device)
    phys(fields, tend, forcing, st, True, 0.3)
    torch.cuda.synchronize()
    for k in P.TENDENCIES:
        assert torch.isfinite(tend[k]).all()
        assert torch.equal(tend[k][0], tend[k][M - 1])
    assert (st.precnv >= 0).all() and (st.precls >= 0).all()
    assert (st.rad_tau2 >= 0).all() and (st.rad_tau2 <= 1).all()
    assert (st.olr > 50).all() and (st.olr < 400).all()


def test_argument_checking(phys):
    import pyspeedy_amd.physics as P
    m = H.synthetic_member(seed=1)
    dev = lambda n: torch.from_numpy(P.to_device_layout(m[n])[None]).cuda()
    fields = {n: dev(n) for n in P.STATE_IN_3D + P.STATE_IN_2D}
    tend = {n: dev(n) for n in P.TENDENCIES}
    forcing = {n: dev(n) for n in P.SURFACE_IN}  # shortwave forcing missing
    st = P.PhysicsState(1, phys.device)
    from pyspeedy_amd import SpeedyHipError
    with pytest.raises(SpeedyHipError):
        phys(fields, tend, forcing, st, True, 0.3)
    fields["tg"] = fields["tg"].float()
    with pytest.raises(ValueError):
        phys(fields, tend, forcing, st, False, 0.3)
