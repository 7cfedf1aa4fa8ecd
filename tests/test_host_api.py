"""The C++ host layer (libpopsift.so: PopSift / popsift::Config / Features) -- the drop-in
boundary of SURVEY.md 8(b).  CPU part: API surface and host logic; GPU part: the demo
program (the reference's usage pattern, main.cpp:304-326) against the C-ABI results."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "popsift_amd")


def _build_host():
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host")], stdout=subprocess.DEVNULL)


def test_host_library_exports_the_class_api():
    _build_host()
    out = subprocess.check_output(["nm", "-DC", "--defined-only", os.path.join(PKG, "libpopsift.so")], text=True)
    for sym in ("PopSift::PopSift(popsift::Config const&, popsift::Config::ProcessingMode, PopSift::ImageMode)",
                "PopSift::PopSift(PopSift::ImageMode)", "PopSift::configure(popsift::Config const&, bool)",
                "PopSift::uninit()", "PopSift::enqueue(int, int, unsigned char const*)",
                "PopSift::enqueue(int, int, float const*)", "SiftJob::get()", "SiftJob::getBase()",
                "SiftJob::getHost()", "SiftJob::getDev()", "popsift::Config::Config()",
                "popsift::Config::setGaussMode(", "popsift::Config::setDescMode(", "popsift::Config::setNormMode(",
                "popsift::Config::equal(popsift::Config const&) const", "popsift::FeaturesHost::reset(int, int)",
                "popsift::FeaturesHost::print(", "popsift::Feature::print("):
        assert sym in out, sym


def test_host_api_semantics_without_gpu():
    _build_host()
    exe = os.path.join(ROOT, "tests", "cpp", "host_api_test.bin")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_api_test.cpp"), "-o", exe, "-L", PKG,
                           "-lpopsift", "-lpopsift_hip", "-pthread", "-Wl,-rpath," + PKG])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "host_api_test ok" in r.stdout


def _write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n# test\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


@pytest.mark.gpu
def test_demo_program_matches_c_abi(gpu_hip, tmp_path):
    """popsift-demo (PopSift::enqueue / SiftJob::get / Features::print) on three images of
    different sizes, in submission order, equals the direct C-ABI results."""
    from popsift_amd.synth import synth
    imgs = [synth(70, 200, 150), synth(71, 200, 150), synth(72, 200, 150)]
    names = []
    for i, im in enumerate(imgs):
        p = str(tmp_path / ("img%d.pgm" % i))
        _write_pgm(p, im)
        names.append(p)
    out = str(tmp_path / "features.txt")
    env = dict(os.environ, POPSIFT_CONTEXTS_PER_DEVICE="2")
    r = subprocess.run([os.path.join(PKG, "popsift-demo"), "-o", out] + names, capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    rows = np.loadtxt(out, ndmin=2)
    counts = [int(l.split(":")[1].split()[0]) for l in r.stderr.splitlines() if l.startswith("Number of feature")]
    dcounts = [int(l.rsplit(":", 1)[1]) for l in r.stderr.splitlines() if l.startswith("Number of feature")]
    assert len(counts) == 3
    ofs = 0
    for im, nf, nd in zip(imgs, counts, dcounts):
        feats, desc = gpu_hip.Context().submit(im).fetch()
        assert (nf, nd) == (len(feats), len(desc))
        block = rows[ofs:ofs + nd]
        ofs += nd
        # text rows are per (feature, orientation) in device compaction order -> compare as sorted sets
        want = []
        for f in feats:
            for k in range(int(f["num_ori"])):
                want.append(np.concatenate([[f["xpos"], f["ypos"], 1.0 / f["sigma"] ** 2], desc[f["desc_idx"][k]]]))
        want = np.array(want)
        got = block[:, [0, 1, 2] + list(range(5, 133))]
        assert got.shape == want.shape
        # printed with 6 (positions) / 3 (descriptor) significant digits: match rows by nearest position
        used = np.zeros(len(want), bool)
        for g in got:
            d = np.abs(want[:, 0] - g[0]) + np.abs(want[:, 1] - g[1]) + np.abs(want[:, 3:] - g[3:]).max(1)
            d[used] = np.inf
            j = int(np.argmin(d))
            used[j] = True
            np.testing.assert_allclose(g[:3], want[j, :3], rtol=2e-5)
            np.testing.assert_allclose(g[3:], want[j, 3:], atol=6e-4)
        assert used.all()
    assert ofs == len(rows)
