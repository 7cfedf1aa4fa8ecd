"""Known-answer tests for behaviours of the reference listed in SURVEY.md Appendix A.9 that no natural image decides:
the no-peak orientation case (s_orientation.cu:183-231), the integer truncation of the squared distance in the
orientation window (s_orientation.cu:123) and OpenCV mode's floorf(threshold) first contrast test (s_extrema.cu:150-153).
They pin the oracle's reading of the reference; tests/test_gpu_quirks.py holds the HIP path to the same answers."""
import numpy as np

import quirks as Q


def _run(O, case, **kw):
    orc = O.Oracle(O.default_params(**Q.params_kw(**kw)), threads=1).run(np.zeros((Q.H, Q.W), np.uint8))
    Q.load_into_oracle(orc, case)
    return orc.fetch()


def test_no_gradient_in_the_window_gives_four_orientations(oracle_mod):
    feats, desc = _run(oracle_mod, Q.NO_PEAK)
    assert len(feats) == 1
    f = feats[0]
    assert abs(f["xpos"] - 30.0) < 1e-4 and abs(f["ypos"] - 24.0) < 1e-4
    assert abs(f["sigma"] - 1.6 * 2.0 ** (1.0 / 3.0)) < 1e-5
    assert f["num_ori"] == 4 and len(desc) == 4
    assert np.all(f["orientation"] == np.float32(Q.NO_PEAK_ANGLE))
    assert np.all(np.isfinite(desc)) and np.all(desc[0] == desc[1]) and desc[0].max() > 0


def test_squared_distance_is_truncated_to_int(oracle_mod):
    feats, desc = _run(oracle_mod, Q.TRUNCATION)
    assert len(feats) == 1
    f = feats[0]
    assert abs(f["xpos"] - 30.03) < 1e-4 and abs(f["ypos"] - 24.0) < 1e-5
    # (21 - 30.03)^2 = 81.54 > 81 = rad^2, but (int)81.54 = 81 <= 81: the pixel counts, the keypoint has ONE orientation
    assert f["num_ori"] == 1, "the pixel at dx^2 = 81.54 must be inside the window (int truncation)"
    assert abs(abs(float(f["orientation"][0])) - np.pi) < 1e-5
    # the control: move the keypoint so that the same pixel has dx^2 = 82.8 -> 82 > 81: outside, no peak, 4 orientations
    far = dict(dog=dict(Q.TRUNCATION["dog"], xc=30.10), gauss=dict(Q.TRUNCATION["gauss"], xc=30.10))
    feats, _ = _run(oracle_mod, far)
    assert len(feats) == 1 and feats[0]["num_ori"] == 4


def test_opencv_mode_floors_the_first_contrast_threshold(oracle_mod):
    feats, _ = _run(oracle_mod, Q.OPENCV_FLOOR, sift_mode=1)
    assert len(feats) == 1, "1.58 >= floorf(1.7) = 1: the candidate must reach refinement in OpenCV mode"
    assert abs(feats[0]["xpos"] - 30.4) < 1e-3 and abs(feats[0]["ypos"] - 24.0) < 1e-4
    for mode in (0, 2):                       # PopSift: 1.6 * thr, VLFeat: 0.8 * 2 * thr = 2.72 > 1.58
        feats, _ = _run(oracle_mod, Q.OPENCV_FLOOR, sift_mode=mode)
        assert len(feats) == 0
