"""popsift_hip_submit_batch: several images of one size through every launch together (image index in blockIdx.y).
Every image has its own slot (planes, lists, result slabs), so its features and descriptors must equal, bit for bit, a
submit of its own -- whatever it is batched with, in whatever position, also after the buffers had to grow."""
import numpy as np
import pytest

from popsift_amd.synth import synth
from util import bits, sorted_features

pytestmark = pytest.mark.gpu


def _canon(feats, desc):
    f, d = sorted_features(feats, desc)
    return (bits(f["xpos"]), bits(f["ypos"]), bits(f["sigma"]), f["num_ori"].copy(), bits(f["orientation"]), bits(d))


def _same(a, b):
    return all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("kw", [dict(), dict(sift_mode=1, gauss_mode=3), dict(desc_mode=2), dict(desc_mode=4, norm_mode=1),
                                dict(filter_max_extrema=600, filter_sorting=1, filter_grid_size=3)],
                         ids=["default", "opencv", "grid", "notile_classic", "grid_filter"])
def test_batch_equals_single_submits(gpu_hip, kw):
    imgs = [synth(400 + k, 320, 240) for k in range(5)]
    single = gpu_hip.Context(gpu_hip.default_params(**kw))
    want = [_canon(*single.submit(im).fetch()) for im in imgs]
    assert len(want[0][0]) > (300 if "filter_max_extrema" in kw else 1000)
    ctx = gpu_hip.Context(gpu_hip.default_params(**kw))
    ctx.submit_batch(imgs)
    counts = ctx.wait_batch()
    assert len(counts) == 5
    for k in range(5):
        assert _same(_canon(*ctx.fetch_item(k)), want[k]), "image %d of the batch differs from its own submit" % k
    # another order and size of the batch, slots reused; then a single submit on the same context
    ctx.submit_batch([imgs[3], imgs[0]])
    assert len(ctx.wait_batch()) == 2
    assert _same(_canon(*ctx.fetch_item(0)), want[3]) and _same(_canon(*ctx.fetch_item(1)), want[0])
    assert _same(_canon(*ctx.submit(imgs[4]).fetch()), want[4])
    assert ctx.wait_batch() == [(len(want[4][0]), len(want[4][5]))]
    ctx.close()
    single.close()


def test_batch_of_float_images_and_a_change_of_size(gpu_hip):
    a = [synth(410 + k, 200, 150).astype(np.float32) / 256.0 for k in range(3)]
    b = [synth(420 + k, 333, 257) for k in range(4)]
    single = gpu_hip.Context()
    ctx = gpu_hip.Context()
    ctx.submit_batch(a)
    for k, im in enumerate(a):
        assert _same(_canon(*ctx.fetch_item(k)), _canon(*single.submit(im).fetch()))
    ctx.submit_batch(b)      # other size (the planes are laid out anew), other dtype, one more slot
    for k, im in enumerate(b):
        assert _same(_canon(*ctx.fetch_item(k)), _canon(*gpu_hip.Context().submit(im).fetch()))
    with pytest.raises(Exception):
        ctx.submit_batch([a[0], b[0]])
    ctx.close()


def test_batch_regrows_all_slots(gpu_hip):
    """tiny initial candidate / histogram capacities: the first wait grows the buffers of every slot and re-runs the
    keypoint stages of the whole batch"""
    imgs = [synth(430 + k, 400, 300) for k in range(3)]
    want = [_canon(*gpu_hip.Context().submit(im).fetch()) for im in imgs]
    ctx = gpu_hip.Context()
    ctx.debug_set(gpu_hip.DEBUG_CAND_CAP, 256)
    ctx.debug_set(gpu_hip.DEBUG_OHIST_CAP, 100)
    ctx.submit_batch(imgs)
    for k in range(3):
        assert _same(_canon(*ctx.fetch_item(k)), want[k])
    ctx.close()


def test_batch_argument_errors(gpu_hip):
    ctx = gpu_hip.Context()
    im = synth(1, 64, 48)
    with pytest.raises(Exception):
        ctx.submit_batch([im] * (gpu_hip.MAX_BATCH + 1))
    ctx.submit_batch([im, im])
    with pytest.raises(Exception):
        ctx.fetch_item(2)
    f0, d0 = ctx.fetch_item(0)
    f1, d1 = ctx.fetch_item(1)
    assert _same(_canon(f0, d0), _canon(f1, d1))
    # image 0 handed to the overlapped download (popsift_hip_fetch_begin = item 0): gone from the slab, image 1 is not
    pend = ctx.fetch_begin()
    with pytest.raises(Exception):
        ctx.fetch_item(0)
    assert _same(_canon(*ctx.fetch_item(1)), _canon(f1, d1))
    ctx.submit_batch([im, im, im])        # the next batch runs under the download
    assert _same(_canon(*pend.result()), _canon(f0, d0))
    assert _same(_canon(*ctx.fetch_item(2)), _canon(f0, d0))
    ctx.close()


def test_a_failed_allocation_inside_a_batch_is_recoverable(gpu_hip):
    """A batch grows the buffers of every slot it uses; when the n-th device allocation fails (DEBUG_FAIL_ALLOC) the
    submit returns ERR_OOM, and the context -- some slots sized, some not, some buffers freed -- must take the next
    batch as if nothing had happened: same results as single submits, for a failure at any of the first allocations of
    the second, third and fourth slot."""
    small = [synth(430 + k, 160, 120) for k in range(2)]
    big = [synth(440 + k, 400, 300) for k in range(4)]
    single = gpu_hip.Context()
    want_small = [_canon(*single.submit(im).fetch()) for im in small]
    want_big = [_canon(*single.submit(im).fetch()) for im in big]
    single.close()
    failures = 0
    for nth in (1, 2, 5, 9, 14, 17, 23, 30, 38):
        ctx = gpu_hip.Context()
        ctx.submit_batch(small)                       # two slots sized for the small images
        for k in range(2):
            assert _same(_canon(*ctx.fetch_item(k)), want_small[k])
        ctx.debug_set(gpu_hip.DEBUG_FAIL_ALLOC, nth)  # four slots, bigger planes: every slot allocates
        try:
            ctx.submit_batch(big)
        except gpu_hip.PopsiftHipError as e:
            assert e.status == gpu_hip.ERR_OOM
            failures += 1
        ctx.debug_set(gpu_hip.DEBUG_FAIL_ALLOC, 0)
        ctx.submit_batch(big)
        assert len(ctx.wait_batch()) == 4
        for k in range(4):
            assert _same(_canon(*ctx.fetch_item(k)), want_big[k]), (nth, k)
        ctx.submit_batch(small)
        for k in range(2):
            assert _same(_canon(*ctx.fetch_item(k)), want_small[k]), (nth, k)
        ctx.close()
    assert failures >= 5      # most of the chosen allocations exist (the later ones depend on the slots' buffer count)


def test_a_failed_submit_of_the_same_size_leaves_no_stale_results(gpu_hip):
    """Two images, then four of the SAME size with an allocation failure in a new slot: the geometry does not change, so
    nothing but the failed submit itself says that the context holds no results -- wait and fetch must answer ERR_STATE
    (not the previous batch's counts for slots whose kernels never ran), and the next submit must work."""
    imgs = [synth(450 + k, 240, 180) for k in range(4)]
    single = gpu_hip.Context()
    want = [_canon(*single.submit(im).fetch()) for im in imgs]
    single.close()
    failed = 0
    for nth in (1, 3, 6, 10):
        ctx = gpu_hip.Context()
        ctx.submit_batch(imgs[:2])
        assert len(ctx.wait_batch()) == 2
        ctx.debug_set(gpu_hip.DEBUG_FAIL_ALLOC, nth)       # slots 2 and 3 are new: they allocate
        try:
            ctx.submit_batch(imgs)
        except gpu_hip.PopsiftHipError as e:
            assert e.status == gpu_hip.ERR_OOM
            failed += 1
            for call in (ctx.wait_batch, lambda: ctx.fetch_item(0), lambda: ctx.fetch_item(3), ctx.wait):
                with pytest.raises(gpu_hip.PopsiftHipError) as ei:
                    call()
                assert ei.value.status == gpu_hip.ERR_STATE
        ctx.debug_set(gpu_hip.DEBUG_FAIL_ALLOC, 0)
        ctx.submit_batch(imgs)
        assert len(ctx.wait_batch()) == 4
        for k in range(4):
            assert _same(_canon(*ctx.fetch_item(k)), want[k]), (nth, k)
        ctx.close()
    assert failed >= 2
