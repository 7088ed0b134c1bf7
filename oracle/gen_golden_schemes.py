"""tests/golden/schemes_s<step>.npz: inputs and outputs of EVERY column-physics scheme of the reference, called one by one
through oracle/ref_shim.f90 in the order of get_physical_tendencies, in the running example_bc model: on ALL 96 x 48 columns
before model steps 36 and 37 (second day: convection, condensation, clouds active), and on every third longitude (the tests
tile them back to 96: the schemes are column-local) before steps 0, 1, 2, where the atmosphere still rests and the
columns of a latitude circle differ by the orography only.
TEST INFRASTRUCTURE; needs oracle/_ref/libspeedy_ref.so (build container only).

    python oracle/gen_golden_schemes.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refmodel as R  # noqa: E402
import schemes as S  # noqa: E402
from gen_golden_physics import physics_inputs_from_state  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
STEPS = (0, 1, 2, 36, 37)
SUBSAMPLED = (0, 1, 2)  # stored on every third longitude


def chain_inputs(m):
    st = physics_inputs_from_state(m)
    cp = float(np.float32(1004.0))
    inp = dict(tg=st["tg"], qg=np.maximum(st["qg_in"], 0.0), phig=st["phig"], ua=st["ug"], va=st["vg"])
    inp["psg"] = np.exp(st["pslg"])
    inp["se"] = cp * inp["tg"] + inp["phig"]
    inp["gse"] = (inp["se"][:, :, 6] - inp["se"][:, :, 7]) / (inp["phig"][:, :, 6] - inp["phig"][:, :, 7])
    for n in S.SURFACE + S.SHORTWAVE_IN:
        inp[n] = m.get(n)
    inp["air_absortivity_co2"] = np.float64(m.get("air_absortivity_co2"))
    return {k: np.asfortranarray(v) for k, v in inp.items()}


def main():
    bc = np.load(os.path.join(HERE, "..", "pyspeedy_amd", "data", "example_bc.npz"))
    import oracle as orc
    fsg = orc.table("fsg")
    for step in STEPS:
        m = R.RefModel()  # a model of its own per snapshot: the shortwave shim writes into the model state
        m.set_bc(bc)
        for _ in range(step):
            assert m.step() == 0
        inp = chain_inputs(m)
        sub = step in SUBSAMPLED
        if sub:  # make the sub-sampled problem self-contained: every third longitude, repeated three times
            inp = {k: (np.asfortranarray(np.repeat(v[::3], 3, axis=0)) if np.ndim(v) >= 2 else v) for k, v in inp.items()}
        res = S.run_chain(S.ReferenceBackend(m), inp, fsg)
        keep = (lambda a: np.ascontiguousarray(a[::3]) if np.ndim(a) >= 2 else a) if sub else (lambda a: a)
        data = {"in_" + k: keep(v) for k, v in inp.items()}
        data["every_third_longitude"] = np.int32(sub)
        for scheme, names in S.SCHEME_OUTPUTS:
            for n in names:
                a = res[scheme][n]
                data["%s_%s" % (scheme, n)] = keep(a[:, :, :2] if n == "hfluxn" else a)  # (plane 3 is never written)
        path = os.path.join(GOLD, "schemes_s%d.npz" % step)
        np.savez_compressed(path, **data)
        act = int((res["convection"]["precnv"] > 0).sum()), int((res["lsc"]["precls"] > 0).sum())
        print("%s: %d KiB raw, %d KiB on disk; convective / large-scale rain in %d / %d columns" % (
            os.path.basename(path), sum(v.nbytes for v in data.values()) // 1024, os.path.getsize(path) // 1024, *act))


if __name__ == "__main__":
    main()
