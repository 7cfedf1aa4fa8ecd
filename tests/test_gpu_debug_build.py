"""The instrumented build of the HIP library (popsift_amd/csrc/trace.h: roctx ranges, stream synchronisation and error
check after every launch -- the reference's NVTX ranges and POP_SYNC_CHK, popsift.h:20-25, common/debug_macros.h:25-29)
gives the product library's results.  It is built by __graft_entry__.build() next to the product library and loaded
here in a child process through POPSIFT_HIP_LIB."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "popsift_amd", "libpopsift_hip_dbg.so")

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
ctx = hip.Context()
f, d = ctx.submit(synth(31, 320, 240)).fetch()
o = np.lexsort((f["sigma"], f["xpos"], f["ypos"], f["debug_octave"]))
f = f[o]
np.savez(sys.argv[2], pos=np.stack([f["xpos"], f["ypos"], f["sigma"]], 1), ori=f["orientation"],
         desc=np.array([d[r["desc_idx"][k]] for r in f for k in range(int(r["num_ori"]))], np.float32))
"""


@pytest.mark.gpu
def test_sync_check_build_matches_the_product(tmp_path):
    assert os.path.exists(DBG), "build it: make -C popsift_amd/csrc ROCTX=1 SYNC_CHECK=1 (__graft_entry__.build() does)"
    out = {}
    for tag, lib in (("product", None), ("dbg", DBG)):
        env = dict(os.environ)
        env.pop("POPSIFT_HIP_LIB", None)
        if lib:
            env["POPSIFT_HIP_LIB"] = lib
        path = str(tmp_path / (tag + ".npz"))
        subprocess.run([sys.executable, "-c", CHILD, ROOT, path], env=env, check=True, timeout=600)
        out[tag] = np.load(path)
    for k in ("pos", "ori", "desc"):
        a, b = out["product"][k], out["dbg"][k]
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
    assert len(out["product"]["desc"]) > 1000
