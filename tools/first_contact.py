"""First GPU contact: stage-by-stage comparison of the HIP path with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
from oracle import oracle as O


def compare(img, tag, **kw):
    print("=== %s %s %s" % (tag, img.shape, kw), flush=True)
    orc = O.Oracle(O.default_params(**kw), threads=8)
    ctx = hip.Context(hip.default_params(**kw))
    t = time.time(); orc.run(img); t_or = time.time() - t
    t = time.time(); ctx.submit(img); nf, nd = ctx.wait(); t_hip = time.time() - t
    rep = ctx.report()
    print("oracle %.2fs counts %s | hip %.3fs counts (%d,%d) dev %.3f ms" % (t_or, orc.counts(), t_hip, nf, nd, rep.ms_device))
    print("ext_ct oracle", orc.ext_counts(), "hip", list(rep.ext_ct)[:rep.num_octaves])
    worst = 0
    for o in range(orc.num_octaves):
        for kind, n in ((0, orc.params.levels + 3), (1, orc.params.levels + 2)):
            for l in range(n):
                a = orc.plane(o, kind, l); b = ctx.plane(o, kind, l)
                nb = int((a.view(np.uint32) != b.view(np.uint32)).sum())
                md = float(np.abs(a - b).max())
                worst = max(worst, md)
                if nb:
                    ys, xs = np.nonzero(a.view(np.uint32) != b.view(np.uint32))
                    print("  oct %d kind %d lvl %d: %d/%d differ, max abs %.3g first at (%d,%d)" % (o, kind, l, nb, a.size, md, xs[0], ys[0]))
    print("planes worst abs diff", worst)
    # extrema sets
    eo = orc.extrema(); eh = ctx.extrema()
    key = lambda e: set(zip(e['octave'].tolist(), e['lpos'].tolist(), e['xpos'].tolist(), e['ypos'].tolist()))
    so, sh = key(eo), key(eh)
    print("extrema: oracle %d hip %d common(exact x,y) %d" % (len(so), len(sh), len(so & sh)))
    fo, do = orc.fetch(); fh, dh = ctx.fetch()
    # match features by (octave, x, y, sigma) exact
    ko = {(int(f['debug_octave']), float(f['xpos']), float(f['ypos'])): i for i, f in enumerate(fo)}
    n_match = n_ori_eq = 0; max_ang = 0; max_desc = 0; bad_desc = 0
    for j, f in enumerate(fh):
        k = (int(f['debug_octave']), float(f['xpos']), float(f['ypos']))
        if k not in ko: continue
        g = fo[ko[k]]; n_match += 1
        if g['num_ori'] != f['num_ori']: continue
        n_ori_eq += 1
        for q in range(f['num_ori']):
            da = abs(float(f['orientation'][q]) - float(g['orientation'][q]))
            max_ang = max(max_ang, da)
            a = do[g['desc_idx'][q]]; b = dh[f['desc_idx'][q]]
            rel = np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-12)
            max_desc = max(max_desc, rel)
            if rel > 1e-3: bad_desc += 1
    print("features: matched %d/%d, same num_ori %d, max |dtheta| %.3g, max desc rel L2 %.3g, >1e-3: %d" % (n_match, len(fo), n_ori_eq, max_ang, max_desc, bad_desc))
    ctx.close(); orc.close()


if __name__ == "__main__":
    print(hip.lib().popsift_hip_version(), "devices", hip.device_count(), flush=True)
    compare(synth(7, 200, 150), "small")
    compare(synth(1, 640, 480), "cfg1-vlfeat", octaves=3, sift_mode=2)
    compare(synth(5, 333, 257), "odd-opencv", sift_mode=1, gauss_mode=3)
    compare(synth(2, 1920, 1080), "cfg2")
