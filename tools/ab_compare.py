"""A/B check of two builds of libpopsift_hip.so on the same images: features, orientations and descriptors must be
bit-identical (or, with --tol, within a relative L2 tolerance).  Each build runs in its own child process
(POPSIFT_HIP_LIB is read when the binding is imported).

  python3 tools/ab_compare.py build_variants/old_tree popsift_amd/libpopsift_hip.so [--tol 1e-6] [--big]
(an argument that is a directory is the root of another checkout with its own built library and binding:
 `git archive <rev> popsift_amd include | tar -x -C build_variants/old_tree` + make in its csrc)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
cases = json.loads(sys.argv[3])
out = {}
for n, (spec, kw) in enumerate(cases):
    ctx = hip.Context(hip.default_params(**kw))
    ctx.submit(synth(*spec))
    feats, desc = ctx.fetch()
    order = np.lexsort((feats["sigma"], feats["xpos"], feats["ypos"], feats["debug_octave"]))
    f = feats[order]
    out["%d_pos" % n] = np.stack([f["xpos"], f["ypos"], f["sigma"]], 1)
    out["%d_nori" % n] = f["num_ori"]
    out["%d_ori" % n] = f["orientation"].copy()
    idx = np.array([r["desc_idx"][k] for r in f for k in range(int(r["num_ori"]))], np.int64)
    out["%d_desc" % n] = desc[idx] if len(idx) else np.zeros((0, 128), np.float32)
    out["%d_ms" % n] = np.float32(ctx.report().ms_device)
    ctx.close()
np.savez(sys.argv[2], **out)
"""


def run(lib, path, cases):
    """lib: a libpopsift_hip.so (loaded by this tree's binding) or the root of another checkout (its own binding)"""
    import json
    env = dict(os.environ)
    root = ROOT
    if os.path.isdir(lib):
        root = os.path.abspath(lib)
        env.pop("POPSIFT_HIP_LIB", None)
    else:
        env["POPSIFT_HIP_LIB"] = os.path.abspath(lib)
    subprocess.run([sys.executable, "-c", CHILD, root, path, json.dumps(cases)], env=env, check=True, timeout=900)
    return np.load(path)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tol = None
    if "--tol" in sys.argv:
        tol = float(sys.argv[sys.argv.index("--tol") + 1])
        args.remove(sys.argv[sys.argv.index("--tol") + 1])
    a_lib, b_lib = args[0], args[1]
    cases = [((5, 640, 480), {}), ((9, 333, 251), {"sift_mode": 1, "octaves": 3}), ((11, 400, 300), {"sift_mode": 2, "norm_mode": 1}),
             ((12, 257, 199), {"upscale_factor": 0.0}), ((13, 96, 64), {"levels": 5})]
    if "--big" in sys.argv:
        cases.append(((2, 1920, 1080), {}))
    import tempfile
    out = tempfile.mkdtemp(prefix="ab_compare_")  # ~100 MB of descriptors: not into gpurun_out (64 MiB merge limit)
    A = run(a_lib, os.path.join(out, "ab_a.npz"), cases)
    B = run(b_lib, os.path.join(out, "ab_b.npz"), cases)
    bad = 0
    for n in range(len(cases)):
        for k in ("pos", "nori", "ori"):
            x, y = A["%d_%s" % (n, k)], B["%d_%s" % (n, k)]
            same = x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))
            if not same:
                bad += 1
            print("case %d %-5s %s %s" % (n, k, x.shape, "identical" if same else "DIFFERENT"))
        x, y = A["%d_desc" % n], B["%d_desc" % n]
        if x.shape != y.shape:
            print("case %d desc shapes differ %s %s" % (n, x.shape, y.shape))
            bad += 1
            continue
        ident = np.array_equal(x.view(np.uint32), y.view(np.uint32))
        rel = np.linalg.norm(x - y, axis=1) / np.maximum(np.linalg.norm(x, axis=1), 1e-30) if len(x) else np.zeros(0)
        print("case %d desc  %s %s  max rel L2 %.3g   T_dev A %.3f ms  B %.3f ms" % (
            n, x.shape, "identical" if ident else "differ", rel.max() if len(rel) else 0.0, A["%d_ms" % n], B["%d_ms" % n]))
        if not ident and (tol is None or (len(rel) and rel.max() > tol)):
            bad += 1
    print("A/B: %s" % ("OK" if bad == 0 else "%d MISMATCHES" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
