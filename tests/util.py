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
              desc_bad=0, ang_bad=0, n_desc=0, max_desc=0.0, max_ang=0.0, max_sigma_rel=0.0)
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
    return st


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
