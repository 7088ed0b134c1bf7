"""CPU tier: the C restatement of every column-physics scheme against the reference's own routine, scheme by scheme, on the
reference's inputs (tests/golden/schemes_s*.npz from oracle/gen_golden_schemes.py: humidity, convection, large-scale
condensation, clouds, shortwave, longwave down, surface fluxes, longwave up, vertical diffusion; model steps 0, 1, 2, 36,
37).  Bitwise, every output, every column; each scheme is fed the REFERENCE's outputs of the schemes before it, so a
difference belongs to the scheme it shows up in."""
import os

import numpy as np
import pytest

STEPS = (0, 1, 2, 36, 37)


def load(golden_dir, step):
    g = np.load(os.path.join(golden_dir, "schemes_s%d.npz" % step))
    rep = (lambda a: np.asfortranarray(np.repeat(a, 3, axis=0)) if np.ndim(a) >= 2 else a) if int(g["every_third_longitude"]) \
        else (lambda a: np.asfortranarray(a))
    import schemes as S
    inp = {n: rep(g["in_" + n]) for n in S.CHAIN_INPUTS}
    ref = {scheme: {n: rep(g["%s_%s" % (scheme, n)]) for n in names} for scheme, names in S.SCHEME_OUTPUTS}
    return inp, ref


@pytest.mark.parametrize("step", STEPS)
def test_every_scheme_bitwise(oracle, golden_dir, step):
    import schemes as S
    inp, ref = load(golden_dir, step)
    got = S.run_chain(S.OracleBackend(), inp, oracle.table("fsg"), feed=ref)
    for scheme, names in S.SCHEME_OUTPUTS:
        for n in names:
            a, b = got[scheme][n], ref[scheme][n]
            if scheme == "surface_fluxes" and n == "hfluxn":
                a = a[:, :, :2]
            assert a.shape == b.shape, (scheme, n, a.shape, b.shape)
            bad = a != b
            assert not bad.any(), "step %d, %s/%s: %d values differ, max |diff| %g, first column %s" % (
                step, scheme, n, int(bad.sum()), np.abs(a.astype(float) - b.astype(float)).max(), np.argwhere(bad)[0])


def test_the_day_two_snapshots_exercise_the_branches(golden_dir):
    """All columns of steps 36 / 37: deep convection, both condensation branches, cloud tops over the whole range."""
    _, ref = load(golden_dir, 36)
    assert (ref["convection"]["precnv"] > 0).sum() > 150
    assert set(np.unique(ref["convection"]["itop"])) >= {3, 4, 5, 9}
    assert (ref["lsc"]["precls"] > 0).sum() > 2000
    assert len(np.unique(ref["clouds"]["icltop"])) >= 5
    assert (ref["vdiff"]["tt"] != 0).sum() > 1000
