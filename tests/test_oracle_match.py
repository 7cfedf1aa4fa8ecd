"""Matching (features.cu:157-221): the oracle's restatement of compute_distance."""
import numpy as np


def test_two_nearest_and_ratio(oracle_mod):
    rng = np.random.default_rng(1)
    l = rng.random((40, 128), np.float32)
    r = rng.random((90, 128), np.float32)
    r[17] = l[3] + 1e-3          # a clear match for left 3
    m = oracle_mod.match(l, r)
    d = ((l[:, None, :].astype(np.float64) - r[None, :, :]) ** 2).sum(-1)
    o = np.argsort(d, axis=1, kind="stable")
    assert np.array_equal(m["best"], o[:, 0]) and np.array_equal(m["second"], o[:, 1])
    np.testing.assert_allclose(m["dist_best"], d[np.arange(40), o[:, 0]], rtol=1e-5)
    assert m["best"][3] == 17 and m["accept"][3] == 1
    assert np.array_equal(m["accept"], (m["dist_best"] / m["dist_second"] < np.float32(0.8)).astype(np.int32))


def test_ties_and_degenerate_sizes(oracle_mod):
    l = np.zeros((2, 128), np.float32)
    r = np.zeros((5, 128), np.float32)
    r[0, 0] = 3.0
    m = oracle_mod.match(l, r)                     # distances 9,0,0,0,0: strict '<' keeps the first of equals
    assert m["best"].tolist() == [1, 1] and m["second"].tolist() == [2, 2]
    assert m["accept"].tolist() == [0, 0]          # 0/0 is NaN, NaN < 0.8 is false
    m = oracle_mod.match(l, r[:1])                 # one candidate: second stays (inf, index 0)
    assert m["best"].tolist() == [0, 0] and m["second"].tolist() == [0, 0]
    assert np.all(np.isinf(m["dist_second"])) and m["accept"].tolist() == [1, 1]   # 9/inf = 0 < 0.8
    m = oracle_mod.match(l, r[:0])                 # none: (inf, inf, 0, 0), inf/inf is NaN
    assert m["best"].tolist() == [0, 0] and m["accept"].tolist() == [0, 0]


def test_summation_order_is_the_shuffle_tree(oracle_mod):
    # values chosen so that float addition order matters: lane sums 1, 2^-24 x 30, 1 -> the tree
    # adds lane t and lane t+16 first; a left-to-right sum would give a different float
    l = np.zeros((1, 128), np.float32)
    r = np.zeros((1, 128), np.float32)
    lane = np.full(32, 2.0 ** -24, np.float32)
    lane[0] = 1.0
    lane[16] = 1.0
    r[0, ::4] = np.sqrt(lane)                       # lane t gets x = sqrt(lane[t]), y = z = w = 0
    m = oracle_mod.match(l, r)
    p = (r[0, ::4] * r[0, ::4]).astype(np.float32)
    t = p.copy()
    s = 16
    while s >= 1:
        t[:s] = t[:s] + t[s:2 * s]
        s //= 2
    seq = np.float32(0)
    for v in p:
        seq = np.float32(seq + v)
    assert m["dist_best"][0] == t[0] and t[0] != seq
