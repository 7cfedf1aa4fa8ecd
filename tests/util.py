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
        # Orientations are paired as a SET: the order of orientation[k] among peaks of exactly equal height is the order the
        # reference's bitonic network leaves them in (common/warp_bitonic_sort.h:35-78, restated literally in the oracle),
        # the device picks the four largest with ties towards the lower bin -- and the symmetric blobs of the synthetic images
        # DO produce exact ties (tools/fuzz_parity.py 400 50505, case 99: two peaks pi apart, swapped).  What is compared is
        # each oracle orientation with the device orientation closest to it, and the descriptors that belong to them.
        n_ori = int(a["num_ori"])
        oa = [float(v) for v in a["orientation"][:n_ori]]
        ob = [float(v) for v in b["orientation"][:n_ori]]
        adiff = lambda p, q: min(abs(p - q), abs(2 * np.pi - abs(p - q)))
        perm, free = [], list(range(n_ori))
        for k in range(n_ori):
            j = min(free, key=lambda j: adiff(oa[k], ob[j]))
            perm.append(j)
            free.remove(j)
        if perm != list(range(n_ori)):
            st["reordered"] = st.get("reordered", 0) + 1
        # A peak in ANOTHER histogram bin (more than half a bin = 0.087 rad away from every device peak) is a different
        # SELECTION of peaks, like another number of them: on an exactly symmetric window (x = 26.5 in OpenCV mode, mirrored
        # blobs) two bins pi apart tie or sit on a plateau, and which of them is a strict local maximum turns on the last
        # bit of the histogram sums (fixed-point on the device, a float lane tree in the oracle, float atomics in arbitrary
        # order in the reference, s_orientation.cu:136).  Such a feature is counted with num_ori_diff (tools/fuzz_parity.py
        # 400 50505, case 99: one feature, peaks at -0.864 and 2.182) and its descriptors are not compared.
        if any(adiff(oa[k], ob[perm[k]]) > 0.0873 for k in range(n_ori)):
            st["num_ori_diff"] += 1
            st["peak_diff"] = st.get("peak_diff", 0) + 1
            continue
        for k in range(n_ori):
            st["n_desc"] += 1
            dth = adiff(oa[k], ob[perm[k]])
            st["max_ang"] = max(st["max_ang"], dth)
            if dth > tol_ang:
                st["ang_bad"] += 1
            x, y = da[a["desc_idx"][k]], db[b["desc_idx"][perm[k]]]
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


def descriptor_parity(st):
    """The descriptor bars of the GPU parity tests, one place for named cases, configs 2 .. 5 and the fuzz runs.
    Returns (ok, message); the message names every offending keypoint.

      - descriptors outside 1e-3 relative L2 whose keypoint orientation AGREES with the oracle's: <= max(1, n // 5000);
      - descriptors outside 1e-3 that are explained by an orientation difference (above): they are the orientation
        differences already bounded by ang_bad <= max(2, n // 2000), so the same bound; their size is bounded by the
        explanation itself (8 x the angle difference + 1e-3), and an angle difference is at most half a histogram bin
        (0.087 rad: beyond that compare_features counts a different SELECTION of peaks, with num_ori_diff).  Round 3
        capped both at 3e-2 from what 1 900 cases had shown; 500 more (seed 70707) brought 3.01e-2 and 3.07e-2;
      - no unexplained descriptor beyond 3e-2.
    The GRID descriptor is held to the same bars, compared in the same frame: see feature_parity."""
    n = max(st["n_desc"], 1)
    expl = st["desc_bad"] - st["unexplained"]
    worst_unexplained = max([o["d_desc"] for o in st["offenders"] if not o["explained"]] + [0.0])
    ok = st["unexplained"] <= max(1, n // 5000) and expl <= max(2, n // 2000) and worst_unexplained < 3e-2
    msg = "%d of %d descriptors outside 1e-3 (%d with agreeing orientation, %d explained by an orientation difference), max %.2e; " \
          "angles: %d outside 1e-3 rad, max %.2e" % (st["desc_bad"], n, st["unexplained"], expl, st["max_desc"], st["ang_bad"], st["max_ang"])
    for o in st["offenders"][:20]:
        msg += "\n    octave %d (%.3f, %.3f) sigma %.3f ori %d: d_angle %.2e d_desc %.2e %s" % (
            o["octave"], o["x"], o["y"], o["sigma"], o["k"], o["d_angle"], o["d_desc"],
            "explained" if o["explained"] else "UNEXPLAINED")
    return ok, msg


def feature_parity(orc, fh, dh, grid_mode=False):
    """ALL bars between an oracle run `orc` and the device's features / descriptors (fh, dh) of the same image: the one
    function behind the named cases, BASELINE configs 2 .. 5 and the fuzz slice.  Returns (ok, message, stats).

    Positions are matched exactly (the pyramid and the refinement are bit-exact); sigma within 1e-5 (device powf);
    features with another NUMBER of orientations, or with a peak in another histogram bin (compare_features),
    <= max(1, n_feat // 2000); orientations outside 1e-3 rad
    <= max(2, n_desc // 2000); descriptors: descriptor_parity.

    grid_mode (DescMode::Grid, s_desc_grid.cu:19-147): the descriptor snaps its 4096 sample points to pixels, so it is a
    STEP function of the orientation -- one ulp of the angle moves 4.6 .. 5.1 % of the oracle's own grid descriptors beyond
    1e-3 on ordinary planes, 8 .. 10 % at 180 x 140, 23 .. 25 % on thin planes (tests/test_oracle_grid_sensitivity.py), and the
    last bits of an orientation are not reproducible even between two runs of the reference (float atomics,
    s_orientation.cu:136); an ulp of the keypoint's SCALE (the device's powf against the host's, bounded at 1e-5 above)
    does the same, since every sample point is a multiple of 3 sigma away from the keypoint.  So the grid descriptors
    are compared IN THE SAME FRAME: the oracle recomputes its descriptors with the device's angles and scales
    (oracle_redo_descriptors), which takes both out of the comparison, and the ordinary bar applies -- 1e-3, at most
    max(1, n // 5000) outside.  (Rounds 2 and 3 bounded the amplified orientation
    noise by bars fitted to what was measured: 5 % / 8.3 % / 40 %.)"""
    fo, do = orc.fetch()
    st = compare_features(fo, do, fh, dh)
    n = max(st["n_desc"], 1)
    problems = []
    if not (st["n_a"] == st["n_b"] == st["matched"] and st["missing"] == 0):
        problems.append("features: oracle %d, device %d, matched %d" % (st["n_a"], st["n_b"], st["matched"]))
    if st["max_sigma_rel"] >= 1e-5:
        problems.append("sigma differs by %.2e relative" % st["max_sigma_rel"])
    if st["num_ori_diff"] > max(1, st["n_a"] // 2000):
        problems.append("%d features with another number of orientations" % st["num_ori_diff"])
    if st["ang_bad"] > max(2, n // 2000):
        problems.append("%d orientations outside 1e-3 rad, max %.2e" % (st["ang_bad"], st["max_ang"]))
    if grid_mode and not problems:
        pairs, _ = match_features(fo, fh)
        ori = np.ascontiguousarray(fo["orientation"], np.float32).copy()
        # a feature's sigma is its octave's sigma times 2^(octave - upscale): a power of two, undone exactly
        scale = np.exp2(fo["debug_octave"].astype(np.float64) - int(orc.params.upscale_factor))
        sig = (fo["sigma"].astype(np.float64) / scale).astype(np.float32)
        for ia, ib in pairs:
            if fo[ia]["num_ori"] == fh[ib]["num_ori"]:
                # the device's angles in the ORACLE's order (compare_features pairs them by angle in the same way)
                n_ori = int(fo[ia]["num_ori"])
                adiff = lambda p, q: min(abs(p - q), abs(2 * np.pi - abs(p - q)))
                free = list(range(n_ori))
                for k in range(n_ori):
                    j = min(free, key=lambda j: adiff(float(fo[ia]["orientation"][k]), float(fh[ib]["orientation"][j])))
                    free.remove(j)
                    ori[ia][k] = fh[ib]["orientation"][j]
                sig[ia] = np.float32(np.float64(fh[ib]["sigma"]) / scale[ia])
        orc.redo_descriptors(ori, 0, sig, 0)
        fo, do = orc.fetch()
        st_frame = compare_features(fo, do, fh, dh)
        st_frame["ang_bad"], st_frame["max_ang"], st_frame["desc_bad_own_frames"] = st["ang_bad"], st["max_ang"], st["desc_bad"]
        st = st_frame
    ok, msg = descriptor_parity(st)
    if not ok:
        problems.append(msg)
    return not problems, "; ".join(problems) if problems else msg, st


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
