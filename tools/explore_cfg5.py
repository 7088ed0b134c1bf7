"""Exploration for tests/test_cfg5_gpu.py: fp32 vs fp64 column physics (snapshots) and 1-day ensembles.  python tools/explore_cfg5.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyspeedy_amd, pyspeedy_amd.physics as P
from pyspeedy_amd.model import EnsembleModel
from test_physics_oracle import load_snapshot

sp = pyspeedy_amd.ModSpectral()
phys = P.ColumnPhysics(sp)
G = os.path.join(ROOT, "tests", "golden")

def run(inp, sw, co2, pre, fp32):
    dev = lambda n: torch.from_numpy(P.to_device_layout(inp[n])[None]).cuda()
    fields = {n: dev("qg_in" if n == "qg" and "qg_in" in inp else n) for n in P.STATE_IN_3D + P.STATE_IN_2D}
    tend = {n: dev(n) for n in P.TENDENCIES}
    forcing = {n: dev(n) for n in P.SURFACE_IN + P.SHORTWAVE_IN}
    st = P.PhysicsState(1, sp.device, diagnostics=True)
    if pre is not None:
        for n, a in pre.items():
            getattr(st, n).copy_(torch.from_numpy(P.to_device_layout(a)[None]).cuda())
    phys(fields, tend, forcing, st, sw, co2, fp32=fp32)
    torch.cuda.synchronize()
    return tend, st

for name in ("physics_sw", "physics_nosw"):
    inp, pre, ref, sw, co2 = load_snapshot(G, name)
    t64, s64 = run(inp, sw, co2, None if sw else pre, False)
    t32, s32 = run(inp, sw, co2, None if sw else pre, True)
    print("==", name)
    for k in ref:
        a = (t64[k][0] if k in t64 else getattr(s64, k)[0]).cpu().numpy()
        b = (t32[k][0] if k in t32 else getattr(s32, k)[0]).cpu().numpy()
        sc = max(np.abs(a).max(), 1e-300)
        d = np.abs(a - b)
        print("%-14s max|64| %.3e  maxerr/scale %.2e  rms err/scale %.2e  frac>1e-4: %.4f" % (k, sc, d.max() / sc, np.sqrt((d**2).mean()) / sc, (d / sc > 1e-4).mean()))
    for k in ("iptop", "icltop"):
        a, b = getattr(s64, k)[0].cpu().numpy(), getattr(s32, k)[0].cpu().numpy()
        print(k, "flipped columns:", int((a != b).sum()), "of", a.size)

bc = dict(np.load(os.path.join(ROOT, "pyspeedy_amd", "data", "example_bc.npz")))
def ensemble(M, fp32, sppt, days=1):
    m = EnsembleModel(sp, M)
    m.set_bc(bc)
    m.spectral2grid()
    tg = m.device_view("t_grid")
    noise = np.stack([np.random.default_rng(i).normal(0.0, 0.01, (96, 48, 8)).transpose(2, 1, 0) for i in range(M)])
    tg += torch.from_numpy(np.ascontiguousarray(noise)).cuda()
    m.grid2spectral()
    if sppt: m.set_sppt(True, seed=7, first_member_id=0)
    m.set_physics_precision(fp32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.run(36 * days)
    codes = m.check(2)
    el = time.perf_counter() - t0
    m.spectral2grid()
    out = {v: m.device_view(v).clone() for v in ("t_grid", "u_grid", "q_grid", "ps_grid")}
    pr = m.device_view("precnv").clone() + m.device_view("precls").clone()
    m.close()
    return out, pr, codes, el

for M in (32,):
    for sppt in (False, True):
        a, pa, ca, ea = ensemble(M, False, sppt)
        b, pb, cb, eb = ensemble(M, True, sppt)
        print("== M", M, "sppt", sppt, "codes", int((ca != 0).sum()), int((cb != 0).sum()), "ms/step fp64 %.3f fp32 %.3f" % (ea / 36 * 1e3, eb / 36 * 1e3))
        for v in a:
            ma, mb = a[v].mean(0), b[v].mean(0)
            sa, sb = a[v].std(0), b[v].std(0)
            dm = (ma - mb).abs()
            print("%-7s |mean| %.3e  max|dmean| %.3e  rms dmean %.3e  rms spread64 %.3e rms spread32 %.3e  max dmean/spread %.3f  member max diff %.3e" % (
                v, ma.abs().max().item(), dm.max().item(), dm.pow(2).mean().sqrt().item(), sa.pow(2).mean().sqrt().item(), sb.pow(2).mean().sqrt().item(),
                (dm / (sa + 1e-300)).max().item(), (a[v] - b[v]).abs().max().item()))
