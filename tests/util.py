"""Shared helpers for the parity tests."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def feature_key(f):
    return (int(f["debug_octave"]), float(f["xpos"]), float(f["ypos"]))


def match_features(fa, fb):
    """Exact (octave, x, y) match.  Returns list of (ia, ib); duplicates matched in order."""
    from collections import defaultdict
    idx = defaultdict(list)
    for i, f in enumerate(fa):
        idx[feature_key(f)].append(i)
    pairs, missing = [], 0
    for j, f in enumerate(fb):
        lst = idx.get(feature_key(f))
        if lst:
            pairs.append((lst.pop(0), j))
        else:
            missing += 1
    return pairs, missing


def compare_features(fa, da, fb, db, tol_desc=1e-3, tol_ang=1e-3):
    """fa/da = oracle, fb/db = device.  Returns a dict of parity statistics."""
    pairs, missing = match_features(fa, fb)
    st = dict(n_a=len(fa), n_b=len(fb), matched=len(pairs), missing=missing, num_ori_diff=0,
              desc_bad=0, ang_bad=0, n_desc=0, max_desc=0.0, max_ang=0.0, max_sigma_rel=0.0, unexplained=0, offenders=[])
    for ia, ib in pairs:
        a, b = fa[ia], fb[ib]
        st["max_sigma_rel"] = max(st["max_sigma_rel"], abs(float(a["sigma"]) - float(b["sigma"])) / float(a["sigma"]))
        if a["num_ori"] != b["num_ori"]:
            st["num_ori_diff"] += 1
            continue
        for k in range(int(a["num_ori"])):
            st["n_desc"] += 1
            dth = abs(float(a["orientation"][k]) - float(b["orientation"][k]))
            dth = min(dth, abs(2 * np.pi - dth))
            st["max_ang"] = max(st["max_ang"], dth)
            if dth > tol_ang:
                st["ang_bad"] += 1
            x, y = da[a["desc_idx"][k]], db[b["desc_idx"][k]]
            rel = float(np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-20))
            st["max_desc"] = max(st["max_desc"], rel)
            if rel > tol_desc:
                st["desc_bad"] += 1
                # A descriptor is computed in the frame of its keypoint orientation: when the two sides' orientations differ
                # by dth (the orientation comes out of a HARD-binned histogram, s_orientation.cu:129, where an ulp of atan2
                # moves a sample to the neighbouring bin), the descriptors differ by a few times dth although each is right
                # for its frame.  Such an offender is "explained"; one whose orientations agree is not.
                explained = rel <= EXPLAIN_FACTOR * dth + tol_desc
                if not explained:
                    st["unexplained"] += 1
                st["offenders"].append(dict(octave=int(a["debug_octave"]), x=float(a["xpos"]), y=float(a["ypos"]), k=k,
                                            sigma=float(a["sigma"]), d_angle=dth, d_desc=rel, explained=explained))
    return st


EXPLAIN_FACTOR = 8.0  # relative L2 change of a descriptor per radian of frame rotation, generously (measured 2 .. 5)


def descriptor_parity(st, grid_mode=False, min_dim=None):
    """The descriptor bars of the GPU parity tests, one place for named cases, configs 4 / 5 and the fuzz runs.
    Returns (ok, message); the message names every offending keypoint.

    loop / iloop / notile / igrid (continuous in all inputs):
      - descriptors outside 1e-3 relative L2 whose keypoint orientation AGREES with the oracle's: <= max(1, n // 5000);
      - descriptors outside 1e-3 that are explained by an orientation difference (above): they are the orientation
        differences already bounded by ang_bad <= max(2, n // 2000), so the same bound;
      - nothing beyond 3e-2, angles within 3e-2.
    grid (grid_mode, min_dim = the smaller side of the OCTAVE-0 PLANE, i.e. of the image after the initial up- or
    down-scaling): s_desc_grid.cu:77 SNAPS the 4096 sample points of a descriptor to pixels, so an orientation that
    differs in its last bits (the reference sums its orientation histogram with float atomics in whatever order the
    hardware takes them, s_orientation.cu:136 -- no two runs of the reference agree in those bits either) moves a point
    across a rounding boundary now and then, and a point that lands on another pixel -- at a clamped border: on a quite
    different one -- is a step of 1e-2 in the descriptor.  The bars are the measured envelope of 1300 random cases at the
    round-3 kernels (tools/fuzz_parity.py 700 31337 and 600 2026) plus two descriptors of slack for small n:
      - planes of more than 200 pixels on both sides: <= 5 % outside 1e-3 (measured: 1.2 % of 246 568 descriptors, single
        cases up to 4.6 %),
      - 97 .. 200 on the smaller side (most keypoints of the upper octaves near a border): <= 8.3 % (measured 2.8 %, single
        cases up to 10 % at n ~ 50), both: all within 1e-1 (measured <= 8.3e-2), angles within 3e-2;
      - thin planes, <= 96 (EVERY keypoint at a clamped border): <= 40 % (measured 7.2 %, single cases up to 38 %), all within 2e-1.
    A third run (600 cases, seed 777001) after the bars were set: 0 failures (profiles/r03_fuzz_parity.txt)."""
    n = max(st["n_desc"], 1)
    expl = st["desc_bad"] - st["unexplained"]
    if grid_mode and min_dim is not None and min_dim <= 96:
        ok = st["desc_bad"] <= max(5, (2 * n) // 5) and st["max_desc"] < 2e-1
    elif grid_mode:
        frac = 12 if (min_dim is not None and min_dim <= 200) else 20
        ok = st["desc_bad"] <= max(3, n // frac + 2) and st["max_desc"] < 1e-1 and st["max_ang"] < 3e-2
    else:
        ok = (st["unexplained"] <= max(1, n // 5000) and expl <= max(2, n // 2000)
              and st["max_desc"] < 3e-2 and st["max_ang"] < 3e-2)
    msg = "%d of %d descriptors outside 1e-3 (%d with agreeing orientation, %d explained by an orientation difference), max %.2e; " \
          "angles: %d outside 1e-3 rad, max %.2e" % (st["desc_bad"], n, st["unexplained"], expl, st["max_desc"], st["ang_bad"], st["max_ang"])
    for o in st["offenders"][:20]:
        msg += "\n    octave %d (%.3f, %.3f) sigma %.3f ori %d: d_angle %.2e d_desc %.2e %s" % (
            o["octave"], o["x"], o["y"], o["sigma"], o["k"], o["d_angle"], o["d_desc"],
            "explained" if o["explained"] else "UNEXPLAINED")
    return ok, msg


def sorted_features(feats, desc):
    """Canonical order (device compaction order is arbitrary): by octave, y, x, sigma."""
    order = np.lexsort((feats["sigma"], feats["xpos"], feats["ypos"], feats["debug_octave"]))
    f = feats[order]
    rows = []
    for r in f:
        for k in range(int(r["num_ori"])):
            rows.append(desc[r["desc_idx"][k]])
    d = np.array(rows, np.float32).reshape(-1, 128)
    return f, d
