"""BitonicSort::Warp32<float>::sort64 (common/warp_bitonic_sort.h:35-78) as the oracle restates it, against an
independent emulation of the same network written here (32 lanes, shuffle_xor semantics: every lane reads its partner's
state from before the step), and against what the orientation stage needs from it: lanes 0 .. 3 hold the four largest
values in descending order."""
import numpy as np


def emulate_sort64(arr):
    """Line by line from the header; returns (x, y) index lists of the 32 lanes."""
    x, y = list(range(32)), list(range(32, 64))

    def shiftit(idx, shift, direction, increasing):
        out = []
        for t in range(32):
            o = t ^ (1 << shift)
            my_val, other_val = arr[idx[t]], arr[idx[o]]
            reverse = bool(t & (1 << direction))
            id_less = (t & (1 << shift)) == 0
            my_more = (my_val > other_val) if id_less else (my_val < other_val)
            must_swap = not (my_more ^ reverse ^ increasing)
            out.append(idx[o] if must_swap else idx[t])
        return out

    for outer in range(5):
        for inner in range(outer, -1, -1):
            x = shiftit(x, inner, outer + 1, False)
            y = shiftit(y, inner, outer + 1, True)
    for t in range(32):
        if arr[x[t]] < arr[y[t]]:
            x[t], y[t] = y[t], x[t]
    for outer in range(5):
        for inner in range(outer, -1, -1):
            x = shiftit(x, inner, outer + 1, False)
            y = shiftit(y, inner, outer + 1, False)
    return x, y


def cases():
    rng = np.random.default_rng(64)
    for i in range(400):
        kind = i % 4
        if kind == 0:
            v = rng.normal(size=64)
        elif kind == 1:                      # the orientation stage's shape: a few finite peaks, the rest -inf
            v = np.full(64, -np.inf)
            n = int(rng.integers(0, 7))
            v[rng.choice(36, n, replace=False)] = rng.uniform(1, 100, n)
        elif kind == 2:                      # exact ties between finite values
            v = rng.integers(0, 5, 64).astype(np.float64)
        else:                                # ties among a few peaks
            v = np.full(64, -np.inf)
            n = int(rng.integers(2, 6))
            v[rng.choice(36, n, replace=False)] = rng.integers(1, 3, n)
        yield v.astype(np.float32)


def test_c_restatement_equals_the_emulated_network(oracle_mod):
    for v in cases():
        got = oracle_mod.warp32_sort64(v)
        x, y = emulate_sort64(v.tolist())
        assert got.tolist() == x + y


def test_sort64_is_a_descending_sort(oracle_mod):
    for v in cases():
        idx = oracle_mod.warp32_sort64(v)
        assert sorted(idx.tolist()) == list(range(64))            # a permutation
        vals = v[idx[:32]]
        assert np.all(vals[:-1] >= vals[1:])                      # x of lanes 0..31: descending
        assert vals[-1] >= v[idx[32:]].max()                      # ... and they are the 32 largest


def test_tie_order_is_the_networks_not_lowest_bin_first(oracle_mod):
    """Known answer: two equal finite peaks at bins 3 and 20.  'Ties towards the lower bin' would give [3, 20]; whatever
    the network gives is pinned here through the emulation, and the orientation stage takes its angles in that order."""
    v = np.full(64, -np.inf, np.float32)
    v[3] = v[20] = 7.0
    x, _ = emulate_sort64(v.tolist())
    got = oracle_mod.warp32_sort64(v)
    assert got[:2].tolist() == x[:2] and set(x[:2]) == {3, 20}
