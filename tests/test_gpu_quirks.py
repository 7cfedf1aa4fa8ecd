"""The crafted-plane cases of tests/quirks.py on the HIP path: planes uploaded through the C-ABI debug hook
(params.store_dog = 1 keeps DoG planes as data), keypoint stages re-run, results against the oracle's."""
import numpy as np
import pytest

import quirks as Q
from util import bits


def _both(O, hip, case, **kw):
    orc = O.Oracle(O.default_params(**Q.params_kw(**kw)), threads=1).run(np.zeros((Q.H, Q.W), np.uint8))
    Q.load_into_oracle(orc, case)
    ctx = hip.Context(hip.default_params(store_dog=1, **Q.params_kw(**kw)))
    ctx.submit(np.zeros((Q.H, Q.W), np.uint8)).wait()
    for l, p in enumerate(Q.dog_planes(**case["dog"])):
        ctx.upload_plane(0, 1, l, p)
    for l, p in enumerate(Q.gauss_planes(**case["gauss"])):
        ctx.upload_plane(0, 0, l, p)
    ctx.rerun_keypoint_stages()
    return orc.fetch(), ctx.fetch()


@pytest.mark.gpu
@pytest.mark.parametrize("name,case,kw,nori", [("no_peak", Q.NO_PEAK, {}, 4), ("truncation", Q.TRUNCATION, {}, 1),
                                               ("opencv_floor", Q.OPENCV_FLOOR, dict(sift_mode=1), None),
                                               ("popsift_drops", Q.OPENCV_FLOOR, dict(sift_mode=0), None)])
def test_reference_quirks_on_the_gpu(oracle_mod, gpu_hip, name, case, kw, nori):
    (fo, do), (fh, dh) = _both(oracle_mod, gpu_hip, case, **kw)
    assert len(fo) == len(fh) == (0 if name == "popsift_drops" else 1)
    if len(fo) == 0:
        return
    a, b = fo[0], fh[0]
    assert bits(a["xpos"]) == bits(b["xpos"]) and bits(a["ypos"]) == bits(b["ypos"])
    assert a["num_ori"] == b["num_ori"] and (nori is None or a["num_ori"] == nori)
    # The crafted planes are mirror-symmetric, so two of a keypoint's histogram peaks can be equal up to the rounding of
    # the sums (the reference's float atomicAdd arrives in no fixed order, s_orientation.cu:136): the orientations are
    # compared as a SET, and each descriptor with the descriptor of the same orientation.
    n = int(a["num_ori"])
    oa, ob = np.argsort(a["orientation"][:n], kind="stable"), np.argsort(b["orientation"][:n], kind="stable")
    np.testing.assert_allclose(b["orientation"][:n][ob], a["orientation"][:n][oa], atol=2e-6)
    assert do.shape == dh.shape and np.all(np.isfinite(dh))
    da = np.stack([do[a["desc_idx"][k]] for k in oa])
    db = np.stack([dh[b["desc_idx"][k]] for k in ob])
    rel = np.linalg.norm(da - db, axis=1) / np.maximum(np.linalg.norm(da, axis=1), 1e-20)
    assert rel.max() < 1e-3
