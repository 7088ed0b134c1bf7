"""Time the physics kernel alone (64 members) for SW and non-SW steps.  Usage: python tools/perf_physics.py [M]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyspeedy_amd
import pyspeedy_amd.physics as P
sys.path.insert(0, os.path.join(ROOT, "tests"))
import physics_helpers as H  # noqa: E402
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sp = pyspeedy_amd.ModSpectral()
phys = P.ColumnPhysics(sp)
base = H.synthetic_member(seed=3)
dev = lambda n: torch.from_numpy(P.to_device_layout(base[n])).cuda()[None].expand(M, *P.shapes(1)[n][1:]).contiguous()
fields = {n: dev(n) for n in P.STATE_IN_3D + P.STATE_IN_2D}
tend = {n: dev(n) for n in P.TENDENCIES}
forcing = {n: dev(n) for n in P.SURFACE_IN + P.SHORTWAVE_IN}
st = P.PhysicsState(M, sp.device)
phys(fields, tend, forcing, st, True, 0.3)
for sw in (True, False):
    for _ in range(3): phys(fields, tend, forcing, st, sw, 0.3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): phys(fields, tend, forcing, st, sw, 0.3)
    b.record(); torch.cuda.synchronize()
    print("waves=%s M=%d sw=%d  %.1f us/launch  %.3f ns/column" % (os.environ.get("PYSPEEDY_AMD_PHYS_WAVES", "1"), M, sw, a.elapsed_time(b) / 20 * 1e3, a.elapsed_time(b) / 20 * 1e6 / (M * 4608)))
