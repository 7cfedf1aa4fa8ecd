"""popsift_hip_fetch_begin / popsift_hip_fetch_end: the download of image i under the kernels of image i+1 on ONE context
(two result slabs, include/popsift_hip.h).  The results must be the bytes the plain fetch delivers, whatever is submitted
in between, and the calls that would read the handed-over slab must refuse."""
import numpy as np
import pytest

from popsift_amd.synth import synth
from test_gpu_configs45 import digest


def _same(a, b):
    """the order of features differs from run to run (atomic compaction): order-independent digest of
    (position, scale, orientation, descriptor) records, as tests/cpp/host_batch_test.cpp computes it"""
    return len(a[0]) == len(b[0]) and len(a[1]) == len(b[1]) and digest(*a) == digest(*b)


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [True, False])
def test_download_overlaps_the_next_image(gpu_hip, pinned):
    imgs = [synth(31, 640, 480), synth(32, 800, 600), synth(33, 333, 257), synth(34, 800, 600)]
    ref_ctx = gpu_hip.Context()
    want = [ref_ctx.submit(im).fetch() for im in imgs]
    assert all(len(w[0]) > 500 for w in want)
    ctx = gpu_hip.Context()
    pend = None
    got = []
    for im in imgs:                       # sizes change: the planes are re-planned while a download is pending
        ctx.submit(im)
        if pend is not None:
            got.append(pend.result())
        pend = ctx.fetch_begin(pinned=pinned)
    got.append(pend.result())
    for k, (g, w) in enumerate(zip(got, want)):
        assert _same(g, w), "image %d" % k
    # the context still serves the plain path afterwards
    assert _same(ctx.submit(imgs[0]).fetch(), want[0])


@pytest.mark.gpu
def test_two_downloads_back_to_back_and_state_errors(gpu_hip):
    a, b = synth(41, 512, 384), synth(42, 512, 384)
    ref_ctx = gpu_hip.Context()
    wa, wb = ref_ctx.submit(a).fetch(), ref_ctx.submit(b).fetch()
    ctx = gpu_hip.Context()
    assert gpu_hip.lib().popsift_hip_fetch_end(ctx._h) == gpu_hip.ERR_STATE     # nothing pending
    pa = ctx.submit(a).fetch_begin()
    for call in (ctx.fetch, ctx.fetch_begin, ctx.clone_results):
        with pytest.raises(gpu_hip.PopsiftHipError) as e:      # the slab went to the pending download
            call()
        assert e.value.status == gpu_hip.ERR_STATE
    pb = ctx.submit(b).fetch_begin()      # waits for a's download first: its slab becomes the one b+1 writes
    assert _same(pb.result(), wb)
    assert _same(pa.result(), wa)
    # re-running the keypoint stages regenerates the results in the current slab
    ctx.submit(a).wait()
    p = ctx.fetch_begin()
    ctx.rerun_keypoint_stages()
    assert _same(ctx.fetch(), wa)
    assert _same(p.result(), wa)


@pytest.mark.gpu
def test_failed_slab_allocation_leaves_results_fetchable(gpu_hip):
    a = synth(43, 400, 300)
    ctx = gpu_hip.Context()
    want = ctx.submit(a).fetch()
    ctx.submit(a).wait()
    ctx.debug_set(gpu_hip.DEBUG_FAIL_ALLOC, 1)     # the second slab's first allocation fails
    with pytest.raises(gpu_hip.PopsiftHipError) as e:
        ctx.fetch_begin()
    assert e.value.status == gpu_hip.ERR_OOM
    assert _same(ctx.fetch(), want)
    assert _same(ctx.fetch_begin().result(), want)


@pytest.mark.gpu
def test_a_download_drained_behind_the_handles_back_still_delivers(gpu_hip):
    """popsift_hip_fetch_begin waits for the previous download BEFORE it allocates, so when it then fails (ERR_OOM) the
    predecessor's data has arrived although its handle was never told; the handle's result() then gets ERR_STATE from
    popsift_hip_fetch_end (nothing is pending any more) and must deliver all the same.  The drain is done here directly
    through the C call -- the state a failed fetch_begin leaves."""
    a, b = synth(44, 400, 300), synth(45, 640, 480)
    ref = gpu_hip.Context()
    wa, wb = ref.submit(a).fetch(), ref.submit(b).fetch()
    ctx = gpu_hip.Context()
    ctx.submit(a).wait()
    pa = ctx.fetch_begin()
    ctx.submit(b).wait()
    assert gpu_hip.lib().popsift_hip_fetch_end(ctx._h) == gpu_hip.OK          # drained, the handle does not know
    assert gpu_hip.lib().popsift_hip_fetch_end(ctx._h) == gpu_hip.ERR_STATE   # nothing pending
    assert _same(pa.result(), wa)
    assert _same(ctx.fetch(), wb)
