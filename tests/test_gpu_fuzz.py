"""A bounded, fixed-seed slice of the randomised differential test (tools/fuzz_parity.py runs it open-ended):
60 random sizes / parameter sets -- levels, sigma, all three SiftModes, both Gauss modes, upscale +1 / 0 / -1, forced
octaves, max_extrema caps, the grid filter, float images, every descriptor mode -- oracle vs HIP: planes and extrema
bit-exact, descriptors within the bars of fuzz_cases.check_case."""
import os

import numpy as np
import pytest

import fuzz_cases


@pytest.mark.gpu
def test_sixty_random_configurations(oracle_mod, gpu_hip):
    rng = np.random.default_rng(20261004)
    failures = []
    for case in range(60):
        kw, img = fuzz_cases.random_case(rng, case, max_w=420, max_h=320)
        ok, msg = fuzz_cases.check_case(oracle_mod, gpu_hip, kw, img, threads=min(os.cpu_count() or 4, 16))
        if not ok:
            failures.append((case, img.shape, kw, msg))
    assert not failures, failures


@pytest.mark.gpu
def test_random_configurations_through_the_march_kernels(oracle_mod, gpu_hip):
    """Cases 90 .. 109 of tools/fuzz_parity.py 400 50505 with every plane-to-plane level forced through the strip-march
    kernels in 32-row segments (BLUR_PATH = 2, BLUR_SEG = 32).  Case 99 is the one with two orientation peaks of exactly
    equal height pi apart: the reference's bitonic network (restated in the oracle) and the device order them differently,
    and the comparison pairs orientations as a set (util.compare_features)."""
    rng = np.random.default_rng(50505)
    failures = []
    for case in range(110):
        kw, img = fuzz_cases.random_case(rng, case)
        if case < 90:
            continue
        ok, msg = fuzz_cases.check_case(oracle_mod, gpu_hip, kw, img, threads=min(os.cpu_count() or 4, 16), debug=((8, 2), (9, 32)))
        if not ok:
            failures.append((case, img.shape, kw, msg))
    assert not failures, failures
