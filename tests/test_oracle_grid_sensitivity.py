"""How far the GRID descriptor (s_desc_grid.cu:19-147) moves when its keypoint's orientation moves by one unit in the
last place -- measured on the oracle itself.

The grid descriptor snaps its 16 x 16 x 16 sample points to pixels (s_desc_grid.cu:77), so it is a step function of the
orientation: an ulp of the angle flips a sample point across a rounding boundary now and then, and a point that lands on
another pixel is a step of 1e-3 .. 1e-1 in the descriptor.  The loop descriptor weights its samples continuously and does
not move.  This is the evidence behind comparing grid descriptors of the HIP path with the oracle IN THE SAME FRAME
(tests/test_gpu_parity.py::test_grid_descriptor_in_the_oracles_frame feeds the HIP angles to the oracle and then demands
1e-3) instead of bounding the amplified orientation noise by a fitted bar."""
import numpy as np
import pytest

from popsift_amd.synth import synth


def moved_fraction(O, kw, spec, ulps, sigma_ulps=0):
    orc = O.Oracle(O.default_params(**kw), threads=8).run(synth(*spec))
    _, d0 = orc.fetch()
    d0 = d0.copy()
    orc.redo_descriptors(None, ulps, None, sigma_ulps)
    _, d1 = orc.fetch()
    rel = np.linalg.norm(d1 - d0, axis=1) / np.maximum(np.linalg.norm(d0, axis=1), 1e-20)
    return float((rel > 1e-3).mean()), float(rel.max()), len(d0)


@pytest.mark.parametrize("spec,label", [((28, 240, 180), "plane 480 x 360"), ((46, 90, 70), "plane 180 x 140"),
                                        ((45, 300, 24), "thin plane 600 x 48")])
def test_one_ulp_of_orientation_moves_grid_descriptors(oracle_mod, spec, label, capsys):
    up, up_max, n = moved_fraction(oracle_mod, dict(desc_mode=2), spec, +1)
    dn, dn_max, _ = moved_fraction(oracle_mod, dict(desc_mode=2), spec, -1)
    four, four_max, _ = moved_fraction(oracle_mod, dict(desc_mode=2), spec, +4)
    sg, sg_max, _ = moved_fraction(oracle_mod, dict(desc_mode=2), spec, 0, +1)
    with capsys.disabled():
        print("\n  grid, %s, %d descriptors: +1 ulp of orientation moves %.2f %% beyond 1e-3 (max %.1e), -1 ulp %.2f %% (max %.1e), "
              "+4 ulp %.2f %% (max %.1e); +1 ulp of sigma %.2f %% (max %.1e)" % (label, n, 100 * up, up_max, 100 * dn, dn_max,
                                                                               100 * four, four_max, 100 * sg, sg_max))
    assert n > 20
    # a step function: some descriptors jump, by far more than an ulp of rotation could move a continuous one
    assert up > 0 or dn > 0 or four > 0
    assert max(up_max, dn_max, four_max) > 1e-3


def test_one_ulp_of_orientation_does_not_move_loop_descriptors(oracle_mod):
    frac, mx, n = moved_fraction(oracle_mod, dict(), (28, 240, 180), +1)
    assert n > 20 and frac == 0.0 and mx < 1e-5
