"""BASELINE.json configs 4 and 5 on the GPU, through the drop-in C++ API.

config 4: a batch of 64 synthetic 1920x1080 images (seeds 100..163) enqueued into ONE PopSift object
          whose worker pool spans POPSIFT_DEVICES=all (every visible GPU x POPSIFT_CONTEXTS_PER_DEVICE contexts --
          src/popsift/popsift.cpp:139-213 semantics, src/application/main.cpp:304-326 usage).  Every job's features
          and descriptors must equal, bit for bit, a single-context C-ABI run of the same image (order-independent
          digest, tests/cpp/host_batch_test.cpp); three of them are compared with the CPU oracle directly.
config 5: the Oxford stand-in of SURVEY.md 8(d) -- seeds 200..211, each under six fixed homographies, 800x640,
          Config::setMode(OpenCV) + setGaussMode("opencv") -- the same digest check for all 72 images and the oracle
          for one warp of every seed.
Also: two workers on ONE card (POPSIFT_DEVICES=0,0) through MatchingMode, i.e. the multi-worker and
the set-to-set matching path of the host layer.
"""
import os
import subprocess

import numpy as np
import pytest

from popsift_amd.synth import oxford_like_stream, synth
from util import feature_parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "popsift_amd")
EXE = os.path.join(ROOT, "tests", "cpp", "host_batch_test.bin")

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)


def _build():
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_batch_test.cpp"),
                           os.path.join(PKG, "host", "pgmread.cpp"), "-o", EXE, "-L", PKG,
                           "-lpopsift", "-lpopsift_hip", "-pthread", "-Wl,-rpath," + PKG])


def _write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def digest(feats, desc):
    """The digest of tests/cpp/host_batch_test.cpp from C-ABI results (vectorised FNV-1a over 32-bit words)."""
    rows = []
    for k in range(4):
        sel = feats[feats["num_ori"] > k]
        if len(sel) == 0:
            continue
        head = np.stack([sel["xpos"].view(np.uint32), sel["ypos"].view(np.uint32), sel["sigma"].view(np.uint32),
                         sel["num_ori"].astype(np.uint32), np.full(len(sel), k, np.uint32),
                         np.ascontiguousarray(sel["orientation"][:, k]).view(np.uint32)], 1)
        rows.append(np.concatenate([head, desc[sel["desc_idx"][:, k]].view(np.uint32)], 1))
    if not rows:
        return 0
    words = np.concatenate(rows, 0).astype(np.uint64)
    h = np.full(len(words), FNV_OFFSET, np.uint64)
    with np.errstate(over="ignore"):
        for c in range(words.shape[1]):
            h = (h ^ words[:, c]) * FNV_PRIME
        return int(h.sum(dtype=np.uint64))


def _run_batch(tmp_path, imgs, extra=(), env=None, timeout=900, n_jobs=None):
    paths = []
    for i, im in enumerate(imgs):
        p = str(tmp_path / ("img_%03d.pgm" % i))
        _write_pgm(p, im)
        paths.append(p)
    e = dict(os.environ, POPSIFT_DEVICES="all")
    e.update(env or {})
    r = subprocess.run([EXE, str(tmp_path)] + list(extra) + paths, capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split() for l in r.stdout.splitlines() if len(l.split()) == 4]
    assert len(rows) == (n_jobs or len(imgs))
    return [(int(a), int(b), int(c), int(d, 16)) for a, b, c, d in rows]


def _load_dump(path):
    raw = np.fromfile(path, np.uint8)
    nf, nd = np.frombuffer(raw[:8], np.int32)
    rec = np.dtype([("xpos", np.float32), ("ypos", np.float32), ("sigma", np.float32), ("orientation", np.float32, (4,)),
                    ("debug_octave", np.int32), ("num_ori", np.int32), ("pad", np.int32, (2,)), ("desc_idx", np.int32, (4,))])
    feats = np.frombuffer(raw[8:8 + nf * rec.itemsize], rec)
    desc = np.frombuffer(raw[8 + nf * rec.itemsize:], np.float32).reshape(nd, 128)
    return feats, desc


def _oracle_check(oracle_mod, params, img, feats, desc):
    orc = oracle_mod.Oracle(params, threads=min(os.cpu_count() or 4, 16)).run(img)
    ok, msg, st = feature_parity(orc, feats, desc)
    assert ok, msg
    return st


@pytest.mark.gpu
def test_config4_batch_of_64_through_the_host_api(tmp_path, gpu_hip, oracle_mod):
    _build()
    seeds = list(range(100, 164))
    imgs = [synth(s, 1920, 1080) for s in seeds]
    rows = _run_batch(tmp_path, imgs, extra=["--dump", "0,21,63"], env={"POPSIFT_CONTEXTS_PER_DEVICE": "4"})
    ctx = gpu_hip.Context()
    for (i, nf, nd, dg), im in zip(rows, imgs):
        feats, desc = ctx.submit(im).fetch()
        assert (nf, nd) == (len(feats), len(desc)), i
        assert nf > 20000
        assert dg == digest(feats, desc), "image %d: host API result differs from the single-context C-ABI run" % i
    for i in (0, 21, 63):
        feats, desc = _load_dump(str(tmp_path / ("result_%d.bin" % i)))
        _oracle_check(oracle_mod, oracle_mod.default_params(), imgs[i], feats, desc)


@pytest.mark.gpu
def test_config5_affine_stream_opencv_mode(tmp_path, gpu_hip, oracle_mod):
    _build()
    stream = oxford_like_stream()
    assert len(stream) == 72
    imgs = [im for _, _, im in stream]
    dump = [6 * n + (n % 6) for n in range(12)]            # one warp of every seed, cycling through the homographies
    rows = _run_batch(tmp_path, imgs, extra=["--opencv", "--dump", ",".join(map(str, dump))])
    kw = dict(sift_mode=gpu_hip.SIFT_OPENCV, gauss_mode=gpu_hip.GAUSS_OPENCV_COMPUTE)
    ctx = gpu_hip.Context(gpu_hip.default_params(**kw))
    total = 0
    for (i, nf, nd, dg), im in zip(rows, imgs):
        feats, desc = ctx.submit(im).fetch()
        assert (nf, nd) == (len(feats), len(desc)), i
        assert dg == digest(feats, desc), "image %d: host API result differs from the single-context C-ABI run" % i
        total += nf
    assert total > 72 * 2000
    for i in dump:
        feats, desc = _load_dump(str(tmp_path / ("result_%d.bin" % i)))
        _oracle_check(oracle_mod, oracle_mod.default_params(**kw), imgs[i], feats, desc)


@pytest.mark.gpu
def test_eight_gpu_worker_count_on_one_card(tmp_path, gpu_hip):
    """The worker pool, the shared queue and the pinned pools at the size they have on an 8-GPU node, on one card:
    POPSIFT_DEVICES=0,0,0,0,0,0,0,0 x 4 contexts = 32 workers, 8 caller threads, 512 jobs (16 distinct 640x480 images,
    32 times over, at most 8 in flight per caller).  Every job's digest must equal the single-context C-ABI run of its
    image.  A rehearsal of the control path, not a scaling number."""
    _build()
    imgs = [synth(300 + k, 640, 480) for k in range(16)]
    rows = _run_batch(tmp_path, imgs, extra=["--callers", "8", "--repeat", "32", "--window", "8"],
                      env={"POPSIFT_DEVICES": "0,0,0,0,0,0,0,0", "POPSIFT_CONTEXTS_PER_DEVICE": "4"}, n_jobs=512)
    ctx = gpu_hip.Context()
    want = []
    for im in imgs:
        feats, desc = ctx.submit(im).fetch()
        want.append((len(feats), len(desc), digest(feats, desc)))
        assert len(feats) > 3000
    for i, nf, nd, dg in rows:
        assert (nf, nd, dg) == want[i % 16], "job %d (image %d) differs from the single-context run" % (i, i % 16)


@pytest.mark.gpu
def test_worker_takes_several_queued_jobs_per_submit(tmp_path, gpu_hip):
    """POPSIFT_BATCH=4: a worker takes up to four queued jobs of one size into ONE popsift_hip_submit_batch.  Two image
    sizes interleaved in runs of different length, 2 workers: every job's digest equals its own single-context run."""
    _build()
    imgs = []
    for k in range(26):
        imgs.append(synth(500 + k, 480, 360) if (k // 5) % 2 == 0 else synth(500 + k, 333, 257))
    rows = _run_batch(tmp_path, imgs, env={"POPSIFT_BATCH": "4", "POPSIFT_CONTEXTS_PER_DEVICE": "2", "POPSIFT_DEVICES": "0"})
    ctx = gpu_hip.Context()
    for (i, nf, nd, dg), im in zip(rows, imgs):
        feats, desc = ctx.submit(im).fetch()
        assert (nf, nd) == (len(feats), len(desc)), i
        assert dg == digest(feats, desc), "job %d differs from the single-context run" % i


MATCH_SRC = r"""
#include <popsift/features.h>
#include <popsift/popsift.h>
#include <popsift_hip.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../popsift_amd/host/pgmread.h"
static unsigned long long hash_desc(const float* d)
{
    unsigned long long h = 14695981039346656037ull;
    for (int q = 0; q < 128; q++) { uint32_t u; memcpy(&u, d + q, 4); h ^= u; h *= 1099511628211ull; }
    return h;
}
int main(int argc, char** argv)
{
    int w, h, w2, h2;
    unsigned char* a = readPGMfile(argv[1], w, h);
    unsigned char* b = readPGMfile(argv[2], w2, h2);
    if (!a || !b) return 2;
    popsift::Config config;
    PopSift sift(config, popsift::Config::MatchingMode);
    SiftJob* ja = sift.enqueue(w, h, a);
    SiftJob* jb = sift.enqueue(w2, h2, b);       /* two workers: the images are extracted by different contexts */
    popsift::FeaturesDev* fa = ja->getDev();
    popsift::FeaturesDev* fb = jb->getDev();
    if (!fa || !fb) return 3;
    std::vector<popsift::FeaturesDev::Match> m = fa->matchAndGet(fb);
    /* the device order of the descriptors differs from run to run: rows are identified by the descriptor bytes */
    std::vector<float> da((size_t)fa->getDescriptorCount() * 128), db((size_t)fb->getDescriptorCount() * 128);
    if (popsift_hip_devfeatures_download(fa->getHandle(), da.data(), 0) != POPSIFT_HIP_OK) return 4;
    if (popsift_hip_devfeatures_download(fb->getHandle(), db.data(), 0) != POPSIFT_HIP_OK) return 4;
    int acc = 0;
    for (size_t i = 0; i < m.size(); i++) acc += m[i].accept ? 1 : 0;
    std::printf("%d %d %zu %d\n", fa->getDescriptorCount(), fb->getDescriptorCount(), m.size(), acc);
    for (size_t i = 0; i < m.size(); i++)
        std::printf("m %016llx %016llx %016llx %d %.9g %.9g\n", hash_desc(&da[i * 128]), hash_desc(&db[(size_t)m[i].best * 128]),
                    hash_desc(&db[(size_t)m[i].second * 128]), m[i].accept ? 1 : 0, m[i].dist_best, m[i].dist_second);
    delete fa; delete fb; delete ja; delete jb;
    sift.uninit();
    return 0;
}
"""


def _hash_rows(d):
    """FNV-1a over the 128 words of every descriptor (hash_desc of MATCH_SRC), vectorised."""
    words = np.ascontiguousarray(d, np.float32).view(np.uint32).astype(np.uint64)
    h = np.full(len(words), FNV_OFFSET, np.uint64)
    with np.errstate(over="ignore"):
        for c in range(128):
            h = (h ^ words[:, c]) * FNV_PRIME
    return h


@pytest.mark.gpu
def test_two_workers_on_one_card_through_matching_mode(tmp_path, gpu_hip):
    """POPSIFT_DEVICES=0,0: two worker contexts on the same GPU, MatchingMode, FeaturesDev::match across the two sets;
    best / second / accept must equal the C-ABI matcher on two single-context extractions of the same images."""
    _build()
    src = tmp_path / "match_two.cpp"
    src.write_text(MATCH_SRC.replace("../../popsift_amd/host/pgmread.h", os.path.join(PKG, "host", "pgmread.h")))
    exe = str(tmp_path / "match_two.bin")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src),
                           os.path.join(PKG, "host", "pgmread.cpp"), "-o", exe, "-L", PKG, "-lpopsift", "-lpopsift_hip",
                           "-pthread", "-Wl,-rpath," + PKG])
    stream = oxford_like_stream(seeds=[200], W=640, H=480)
    ia, ib = stream[0][2], stream[1][2]
    pa, pb = str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm")
    _write_pgm(pa, ia)
    _write_pgm(pb, ib)
    r = subprocess.run([exe, pa, pb], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, POPSIFT_DEVICES="0,0", POPSIFT_CONTEXTS_PER_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    na, nb, nm, acc = map(int, lines[0].split())
    got = [l.split()[1:] for l in lines[1:] if l.startswith("m ")]
    ca, cb = gpu_hip.Context(), gpu_hip.Context()
    ca.submit(ia).wait()
    cb.submit(ib).wait()
    sa, sb = ca.clone_results(), cb.clone_results()
    assert (na, nb) == (sa.info()[2], sb.info()[2]) and nm == na and len(got) == na
    # the device order of descriptors differs between runs: rows are aligned through the descriptor bytes
    da, _ = sa.download()
    db, _ = sb.download()
    want = sa.match(sb)
    ha, hb = _hash_rows(da), _hash_rows(db)
    # a few descriptors occur twice (two orientations of one keypoint a hair apart): such rows have the same match by
    # construction (the search sees only the descriptor bytes), so any row of that content stands for all of them
    row_of = {int(h): i for i, h in enumerate(ha)}
    assert len(row_of) >= na - na // 500
    seen = []
    for hl, hbest, hsecond, accept, dbest, dsecond in got:
        i = row_of[int(hl, 16)]          # KeyError: the host API extracted a descriptor the C-ABI run does not have
        seen.append(int(hl, 16))
        w = want[i]
        assert int(hbest, 16) == int(hb[w["best"]]) and int(hsecond, 16) == int(hb[w["second"]]), "row %d: best / second differ" % i
        assert int(accept) == int(w["accept"]), "row %d: accept differs" % i
        assert np.float32(dbest) == w["dist_best"] and np.float32(dsecond) == w["dist_second"], "row %d: distances differ" % i
    assert sorted(seen) == sorted(int(h) for h in ha)      # the same multiset of left descriptors
    assert acc == int(want["accept"].sum())
    assert acc > 50, "the two views must share matches (zoom 1.12, 10 degrees)"
