"""Randomised differential test: oracle vs HIP on random sizes / parameters (tests/fuzz_cases.py; a fixed 60-case
slice of it is tests/test_gpu_fuzz.py).  Open-ended here: python3 tools/fuzz_parity.py [n_cases] [seed] [max_w max_h], on the
GPU box.  FUZZ_DEBUG="8:2,9:64": popsift_hip_debug_set switches for every context (here: march kernels, 64-row segments)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import fuzz_cases
from oracle import oracle as O
from popsift_amd import _capi as hip

O.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
SIZE = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ()
DEBUG = tuple(tuple(int(v) for v in item.split(':')) for item in filter(None, os.environ.get('FUZZ_DEBUG', '').split(',')))
bad = 0
for case in range(n_cases):
    kw, img = fuzz_cases.random_case(rng, case, *SIZE)
    ok, msg = fuzz_cases.check_case(O, hip, kw, img, debug=DEBUG)
    bad += 0 if ok else 1
    print("%s case %2d %dx%d %s %s" % ("ok  " if ok else "FAIL", case, img.shape[1], img.shape[0], kw, msg), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
