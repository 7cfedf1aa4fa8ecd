"""Grid filter on the GPU against the oracle (s_filtergrid.cu:109-322)."""
import numpy as np
import pytest

from popsift_amd.synth import synth
from util import compare_features

pytestmark = pytest.mark.gpu


def ekeys(e):
    return set(zip(e["octave"].tolist(), e["lpos"].tolist(), e["xpos"].tolist(), e["ypos"].tolist()))


def run_pair(O, hip, img, **kw):
    orc = O.Oracle(O.default_params(**kw), threads=8).run(img)
    ctx = hip.Context(hip.default_params(**kw)); ctx.submit(img); ctx.wait()
    return orc, ctx


@pytest.mark.parametrize("mode", [1, 2], ids=["largest_first", "smallest_first"])
@pytest.mark.parametrize("grid,fmax", [(2, 800), (1, 500), (5, 1200), (16, 900)])
def test_scale_ordered_filter_keeps_the_oracles_members(oracle_mod, gpu_hip, mode, grid, fmax):
    img = synth(31, 320, 240)
    orc, ctx = run_pair(oracle_mod, gpu_hip, img, filter_max_extrema=fmax, filter_sorting=mode, filter_grid_size=grid)
    eo, eh = orc.extrema(), ctx.extrema()
    assert len(eo) < 2000 and ekeys(eo) == ekeys(eh)
    assert orc.ext_counts() == list(ctx.report().ext_ct)[:orc.num_octaves]
    fo, do = orc.fetch()
    fh, dh = ctx.fetch()
    st = compare_features(fo, do, fh, dh)
    assert st["n_a"] == st["n_b"] == st["matched"] and st["missing"] == 0
    assert st["desc_bad"] <= max(1, st["n_desc"] // 500) and st["max_desc"] < 3e-2, st
    ctx.close()


@pytest.mark.parametrize("grid", [1, 2, 4])
def test_random_scale_keeps_the_same_number_per_cell(oracle_mod, gpu_hip, grid):
    # RandomScale keeps "the first ones" of every cell in an order that is arrival order in the
    # reference and compaction order here: only the per-cell counts are defined
    img = synth(32, 320, 240)
    orc, ctx = run_pair(oracle_mod, gpu_hip, img, filter_max_extrema=700, filter_sorting=0, filter_grid_size=grid)
    eo, eh = orc.extrema(), ctx.extrema()
    assert np.array_equal(np.bincount(eo["cell"], minlength=grid * grid), np.bincount(eh["cell"], minlength=grid * grid))
    assert orc.counts()[0] == ctx.wait()[0]
    full = gpu_hip.Context(gpu_hip.default_params(filter_grid_size=grid)).submit(img)
    assert ekeys(eh) <= ekeys(full.extrema())
    fh, dh = ctx.fetch()
    assert len(fh) == len(eh) and np.all(np.isfinite(dh))
    ctx.close()


def test_ten_percent_rule_leaves_the_list_alone(oracle_mod, gpu_hip):
    img = synth(33, 300, 220)
    plain = gpu_hip.Context().submit(img)
    n0 = plain.wait()[0]
    e0 = plain.extrema()
    for fmax in (int(n0 / 1.1) + 1, 10 * n0):                 # int(max * 1.1) >= total: not filtered
        ctx = gpu_hip.Context(gpu_hip.default_params(filter_max_extrema=fmax, filter_sorting=1)).submit(img)
        assert ctx.wait()[0] == n0 and ekeys(ctx.extrema()) == ekeys(e0)
    ctx = gpu_hip.Context(gpu_hip.default_params(filter_max_extrema=int(n0 / 1.1) - 2, filter_sorting=1)).submit(img)
    assert ctx.wait()[0] < n0                                 # just past the slack: filtered


def test_filter_at_full_size_and_context_reuse(oracle_mod, gpu_hip):
    # 1080p, 77 k extrema -> 20 k; then a small image through the same context
    img = synth(2, 1920, 1080)
    kw = dict(filter_max_extrema=20000, filter_sorting=1, filter_grid_size=4)
    orc, ctx = run_pair(oracle_mod, gpu_hip, img, **kw)
    eo, eh = orc.extrema(), ctx.extrema()
    assert 19000 < len(eo) <= 20000 + 32 and ekeys(eo) == ekeys(eh)
    small = synth(31, 320, 240)
    orc2 = oracle_mod.Oracle(oracle_mod.default_params(octaves=orc.num_octaves, **kw)).run(small)
    ctx.submit(small)
    assert ekeys(orc2.extrema()) == ekeys(ctx.extrema())
    ctx.close()


def test_filter_rejects_unsupported_grids(gpu_hip):
    with pytest.raises(gpu_hip.PopsiftHipError):
        gpu_hip.Context(gpu_hip.default_params(filter_max_extrema=100, filter_grid_size=65))
    with pytest.raises(gpu_hip.PopsiftHipError):
        gpu_hip.Context(gpu_hip.default_params(filter_max_extrema=100, filter_sorting=3))
