"""Matching on the GPU: popsift_hip_match_sets against the oracle, bit for bit."""
import numpy as np
import pytest

from popsift_amd.synth import synth

pytestmark = pytest.mark.gpu


def assert_same(mo, mh):
    for k in ("best", "second", "accept"):
        assert np.array_equal(mo[k], mh[k]), k
    assert np.array_equal(mo["dist_best"].view(np.uint32), mh["dist_best"].view(np.uint32))
    assert np.array_equal(mo["dist_second"].view(np.uint32), mh["dist_second"].view(np.uint32))


@pytest.mark.parametrize("nl,nr", [(1, 1), (1, 2), (3, 1), (31, 63), (32, 64), (33, 65), (100, 1000), (700, 129), (2500, 3100)])
def test_random_sets_match_the_oracle(oracle_mod, gpu_hip, nl, nr):
    rng = np.random.default_rng(nl * 7919 + nr)
    l = rng.random((nl, 128), np.float32)
    r = rng.random((nr, 128), np.float32)
    k = min(nl, nr) // 3
    r[rng.permutation(nr)[:k]] = l[rng.permutation(nl)[:k]] + rng.normal(0, 0.02, (k, 128)).astype(np.float32)
    if nr > 4:
        r[nr - 1] = r[1]                     # exact duplicates: ties go to the lower index
        r[nr // 2] = r[1]
    L, R = gpu_hip.DevFeatures.from_host(l), gpu_hip.DevFeatures.from_host(r)
    assert_same(oracle_mod.match(l, r), L.match(R))


def test_empty_sets(gpu_hip, oracle_mod):
    l = np.ones((5, 128), np.float32)
    L, E = gpu_hip.DevFeatures.from_host(l), gpu_hip.DevFeatures.from_host(np.zeros((0, 128), np.float32))
    m = L.match(E)
    assert_same(oracle_mod.match(l, np.zeros((0, 128), np.float32)), m)
    assert len(E.match(L)) == 0


def test_cloned_results_and_image_to_image_matching(oracle_mod, gpu_hip):
    """MatchingMode: two images extracted, cloned to device-resident sets, matched (match.cpp:255-273)."""
    a = synth(90, 320, 240)
    b = np.roll(a, (3, 5), axis=(0, 1))                  # the same scene shifted: most keypoints re-appear
    ca, cb = gpu_hip.Context().submit(a), gpu_hip.Context().submit(b)
    fa, da = ca.fetch()
    fb, db = cb.fetch()
    A, B = ca.clone_results(), cb.clone_results()
    assert A.info() == (0, len(fa), len(da)) and B.info() == (0, len(fb), len(db))
    xa, ra = A.download()
    assert np.array_equal(xa.view(np.uint32), da.view(np.uint32))
    # reverse map: descriptor -> feature (feat_to_ext_map)
    want = np.concatenate([[i] * int(f["num_ori"]) for i, f in enumerate(fa)])
    assert np.array_equal(ra, want)
    ca.submit(b)                                         # the clone is independent of its context
    ca.wait()
    assert np.array_equal(A.download()[0].view(np.uint32), da.view(np.uint32))
    m = A.match(B)
    assert_same(oracle_mod.match(da, db), m)
    acc = m["accept"] == 1
    assert acc.mean() > 0.3
    # accepted matches land on the shifted position
    pa = np.stack([fa["xpos"][ra], fa["ypos"][ra]], 1)[acc]
    pb = np.stack([fb["xpos"][B.download()[1]], fb["ypos"][B.download()[1]]], 1)[m["best"][acc]]
    ok = np.hypot(pb[:, 0] - pa[:, 0] - 5, pb[:, 1] - pa[:, 1] - 3) < 1.0
    assert ok.mean() > 0.9


def test_full_size_sets(oracle_mod, gpu_hip):
    """Two 1080p images (about 95 k descriptors each side is too slow for the oracle: 20 k x 95 k here)."""
    a = gpu_hip.Context().submit(synth(2, 1920, 1080))
    fa, da = a.fetch()
    b = gpu_hip.Context().submit(synth(102, 1920, 1080))
    fb, db = b.fetch()
    A, B = gpu_hip.DevFeatures.from_host(da[:20000]), b.clone_results()
    m = A.match(B)
    mo = oracle_mod.match(da[:2000], db, threads=16)
    assert_same(mo, m[:2000])
    # size-independent property on the rest: the reported distances are the true distances of the
    # reported indices and nothing is closer (spot check)
    idx = np.arange(2000, 20000, 37)
    d = ((da[idx].astype(np.float64) - db[m["best"][idx]]) ** 2).sum(-1)
    np.testing.assert_allclose(m["dist_best"][idx], d, rtol=1e-5)
    assert np.all(m["dist_best"] <= m["dist_second"])


@pytest.fixture
def match_path(gpu_hip):
    """popsift_hip_match_set_path for the duration of a test"""
    def set_path(p):
        assert gpu_hip.lib().popsift_hip_match_set_path(p) == 0
    yield set_path
    gpu_hip.lib().popsift_hip_match_set_path(gpu_hip.MATCH_AUTO)


@pytest.mark.parametrize("nl,nr", [(1, 1), (2, 3), (5, 4), (33, 5), (130, 129), (257, 1000), (1500, 2100)])
def test_screened_path_matches_the_oracle(oracle_mod, gpu_hip, match_path, nl, nr):
    """Matrix-core screening + exact re-rank (match_mfma.hip) forced for every size, against the oracle."""
    match_path(gpu_hip.MATCH_SCREEN)
    rng = np.random.default_rng(nl * 31 + nr)
    l = rng.random((nl, 128), np.float32)
    l /= np.linalg.norm(l, axis=1, keepdims=True)
    r = rng.random((nr, 128), np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    k = min(nl, nr) // 2
    r[rng.permutation(nr)[:k]] = l[rng.permutation(nl)[:k]] + rng.normal(0, 0.01, (k, 128)).astype(np.float32)
    L, R = gpu_hip.DevFeatures.from_host(l), gpu_hip.DevFeatures.from_host(r)
    assert_same(oracle_mod.match(l, r), L.match(R))


def test_screening_margin_cases(oracle_mod, gpu_hip, match_path):
    """Rows the screening pass cannot decide -- exact duplicates and near-ties in the right set, more of them than
    it tracks -- go to the exact kernel; large norms (norm_multi) scale the margin."""
    match_path(gpu_hip.MATCH_SCREEN)
    rng = np.random.default_rng(7)
    l = rng.random((300, 128), np.float32)
    r = rng.random((900, 128), np.float32)
    r[10:17] = l[5]                                        # seven exact copies of a left row: all distance 0
    r[100:106] = l[6] + 1e-7                               # six near-copies inside the margin
    r[200] = l[7]
    r[201] = l[7] + np.float32(3e-4)                       # best and second a hair apart
    r[300:303] = r[299]                                    # duplicates that are nobody's neighbour
    for scale in (1.0, 512.0):
        ls, rs = (l * scale).astype(np.float32), (r * scale).astype(np.float32)
        L, R = gpu_hip.DevFeatures.from_host(ls), gpu_hip.DevFeatures.from_host(rs)
        m = L.match(R)
        assert_same(oracle_mod.match(ls, rs), m)
        assert m["best"][5] == 10 and m["second"][5] == 11 and m["best"][7] == 200
    # and the two paths agree with each other on a bigger case
    l = rng.random((3000, 128), np.float32)
    r = rng.random((5000, 128), np.float32)
    L, R = gpu_hip.DevFeatures.from_host(l), gpu_hip.DevFeatures.from_host(r)
    a = L.match(R)
    match_path(gpu_hip.MATCH_EXACT)
    b = L.match(R)
    assert_same(a, b)
