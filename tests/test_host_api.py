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
                "popsift::FeaturesHost::print(", "popsift::Feature::print(",
                "popsift::FeaturesDev::match(popsift::FeaturesDev*)", "popsift::FeaturesDev::reset(int, int)",
                "popsift::FeaturesDev::getReverseMap()", "popsift::cuda::device_prop_t::set(int, bool)"):
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


def _check_rows(rows, feats, desc):
    """rows of output-features.txt (x y 1/s^2 0 1/s^2 d0..d127 per orientation) against C-ABI results."""
    want = []
    for f in feats:
        for k in range(int(f["num_ori"])):
            want.append(np.concatenate([[f["xpos"], f["ypos"], 1.0 / f["sigma"] ** 2], desc[f["desc_idx"][k]]]))
    want = np.array(want)
    got = rows[:, [0, 1, 2] + list(range(5, 133))]
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


def _counts(stderr):
    lines = [l for l in stderr.splitlines() if l.startswith("Number of feature")]
    return [(int(l.split(":")[1].split()[0]), int(l.rsplit(":", 1)[1])) for l in lines]


DEMO = os.path.join(PKG, "popsift-demo")


@pytest.mark.gpu
def test_demo_program_matches_c_abi(gpu_hip, tmp_path):
    """popsift-demo with the reference's command line (main.cpp:48-149): one image per run writes
    output-features.txt; a directory run processes every file below it (main.cpp:149-166)."""
    from popsift_amd.synth import synth
    imgs = [synth(70, 200, 150), synth(71, 180, 150), synth(72, 200, 130)]
    d = tmp_path / "in"
    (d / "sub").mkdir(parents=True)
    names = [str(d / "img0.pgm"), str(d / "img1.pgm"), str(d / "sub" / "img2.pgm")]
    for p, im in zip(names, imgs):
        _write_pgm(p, im)
    env = dict(os.environ, POPSIFT_CONTEXTS_PER_DEVICE="2")
    want = [gpu_hip.Context().submit(im).fetch() for im in imgs]
    for p, (feats, desc) in zip(names, want):
        r = subprocess.run([DEMO, "--pgmread-loading", "-i", p], capture_output=True, text=True, timeout=300, env=env,
                           cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr
        assert r.stdout.splitlines()[0] == p                      # main.cpp:278
        assert _counts(r.stderr) == [(len(feats), len(desc))]
        _check_rows(np.loadtxt(str(tmp_path / "output-features.txt"), ndmin=2), feats, desc)
    r = subprocess.run([DEMO, "--input-file=" + str(d), "--dont-write", "--print-dev-info", "--print-time-info",
                        "--print-gauss-tables"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert sorted(_counts(r.stderr)) == sorted((len(f), len(x)) for f, x in want)
    assert "is directory" in r.stdout and "Choosing device 0" in r.stdout and "Device information:" in r.stdout
    assert "Warp size:             64" in r.stdout
    assert "    relative sigma" in r.stdout and "      5 27 3.09" in r.stdout      # level 5: 27 taps, sigma 3.090


@pytest.mark.gpu
def test_demo_modes_filter_and_log_dumps(gpu_hip, tmp_path):
    from popsift_amd.synth import synth
    im = synth(73, 160, 120)
    p = str(tmp_path / "a.pgm")
    _write_pgm(p, im)
    kw = dict(sift_mode=2, desc_mode=2, norm_mode=1, norm_multi=9, levels=4, filter_max_extrema=150,
              filter_grid_size=3, filter_sorting=1, upscale_factor=0.0)
    ctx = gpu_hip.Context(gpu_hip.default_params(**kw)).submit(im)
    feats, desc = ctx.fetch()
    r = subprocess.run([DEMO, "-i", p, "--vlfeat-mode", "--desc-mode", "grid", "--norm-mode=classic", "--norm-multi", "9",
                        "--levels", "4", "--filter-max-extrema", "150", "--filter-grid", "3", "--filter-sort", "down",
                        "--downsampling", "0", "--write-as-uchar", "--log", "--output-file", "f.txt"],
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert _counts(r.stderr) == [(len(feats), len(desc))]
    rows = np.loadtxt(str(tmp_path / "f.txt"), ndmin=2)
    assert rows.shape == (len(desc), 133) and np.all(rows[:, 5:] == np.round(rows[:, 5:]))   # rounded to int
    assert abs(rows[:, 5:].max() - np.round(desc).max()) <= 1
    # --log: sift_octave.cu:110-187 / sift_pyramid.cu:88-106 file set; the float dumps are the planes
    n_oct = ctx.report().num_octaves
    for o in range(n_oct):
        w, h = ctx.octave_dims(o)
        for l in range(4 + 3):
            raw = open(str(tmp_path / "dir-octave-dump" / ("pyramid-o-%d-l-%d.dump" % (o, l))), "rb").read()
            head = b"floats\n%d %d\n" % (w, h)
            assert raw.startswith(head)
            assert np.array_equal(np.frombuffer(raw[len(head):], np.float32).reshape(h, w), ctx.plane(o, 0, l))
            assert (tmp_path / "dir-octave" / ("pyramid-o-%d-l-%d.pgm" % (o, l))).exists()
        for l in range(4 + 2):
            for sub, ext in (("dir-dog", "pgm"), ("dir-dog-txt", "txt"), ("dir-dog-dump", "dump")):
                assert (tmp_path / sub / ("d-pyramid-o-%d-l-%d.%s" % (o, l, ext))).exists()
    txt = open(str(tmp_path / "dir-dog-txt" / "d-pyramid-o-0-l-1.txt")).read().split()
    w0, h0 = ctx.octave_dims(0)
    assert txt[:4] == ["P2", str(w0), str(h0), "255"]
    assert np.array_equal(np.array(txt[4:], int).reshape(h0, w0), ctx.plane(0, 1, 1).astype(np.int32) + 127)
    assert len(open(str(tmp_path / "dir-desc" / "desc-pyramid.txt")).read().splitlines()) == len(desc)
    assert len(open(str(tmp_path / "dir-fpt" / "desc-pyramid.txt")).read().splitlines()) == len(desc)


@pytest.mark.gpu
def test_match_program(gpu_hip, oracle_mod, tmp_path):
    """popsift-match (match.cpp:229-276): MatchingMode extraction of two images + FeaturesDev::match,
    one accept/reject line per left descriptor, against C-ABI extraction + the oracle's matcher."""
    import re
    from popsift_amd.synth import synth
    a = synth(95, 240, 180)
    b = np.roll(a, (2, 4), axis=(0, 1))
    pa, pb = str(tmp_path / "l.pgm"), str(tmp_path / "r.pgm")
    _write_pgm(pa, a)
    _write_pgm(pb, b)
    r = subprocess.run([os.path.join(PKG, "popsift-match"), "-l", pa, "--right=" + pb], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    fa, da = gpu_hip.Context().submit(a).fetch()
    fb, db = gpu_hip.Context().submit(b).fetch()
    lines = r.stdout.splitlines()
    assert lines[0] == pa + " <-> " + pb
    assert lines[1:5] == ["Number of features:    %d" % len(fa), "Number of descriptors: %d" % len(da),
                          "Number of features:    %d" % len(fb), "Number of descriptors: %d" % len(db)]
    rows = lines[5:]
    assert len(rows) == len(da)
    # feature order inside an octave is compaction order (differs run to run): compare as multisets of
    # (accept, left position, best position) through the descriptor -> feature maps
    mo = oracle_mod.match(da, db)
    rev_a = np.concatenate([[i] * int(f["num_ori"]) for i, f in enumerate(fa)])
    rev_b = np.concatenate([[i] * int(f["num_ori"]) for i, f in enumerate(fb)])
    pat = re.compile(r"(accept|reject) feat +(\d+) \[ *(\d+)\] matches feat +(\d+) \[ *(\d+)\] \( 2nd feat +(\d+) \[ *(\d+)\] \) dist ([0-9.]+) vs ([0-9.]+)")
    got = []
    for i, row in enumerate(rows):
        m = pat.match(row)
        assert m, row
        assert int(m.group(3)) == i
        got.append((m.group(1), m.group(8), m.group(9)))
    want = [("accept" if x["accept"] else "reject", "%.3f" % x["dist_best"], "%.3f" % x["dist_second"]) for x in mo]
    assert sorted(got) == sorted(want)
    assert sum(1 for g in got if g[0] == "accept") > len(got) // 4


@pytest.mark.gpu
def test_concurrent_callers_and_two_objects(gpu_hip):
    """tests/cpp/host_mt_test.cpp: four caller threads into one PopSift object, a MatchingMode object alongside."""
    _build_host()
    exe = os.path.join(ROOT, "tests", "cpp", "host_mt_test.bin")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_mt_test.cpp"), "-o", exe, "-L", PKG,
                           "-lpopsift", "-lpopsift_hip", "-pthread", "-Wl,-rpath," + PKG])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "host_mt_test ok" in r.stdout
    # eight callers into a pool of 2 x 2 contexts on one card (POPSIFT_DEVICES=0,0): more callers than workers, two
    # "devices" sharing the queue and the pinned pools
    r = subprocess.run([exe, "8", "9"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, POPSIFT_DEVICES="0,0", POPSIFT_CONTEXTS_PER_DEVICE="2"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "host_mt_test ok" in r.stdout


def test_demo_command_line_errors():
    _build_host()
    r = subprocess.run([DEMO, "--help"], capture_output=True, text=True)
    assert r.returncode == 1 and "--filter-max-extrema arg" in r.stdout and "-i [ --input-file ] arg" in r.stdout
    r = subprocess.run([DEMO, "--octaves", "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "'--input-file' is required" in r.stderr
    r = subprocess.run([DEMO, "--no-such-option", "-i", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "unrecognised option" in r.stderr
    r = subprocess.run([DEMO, "-i", "x", "--filter-sort", "sideways"], capture_output=True, text=True)
    assert r.returncode != 0 and "up, down or random" in r.stderr


def test_pgm_reader(tmp_path):
    """pgmread.cpp:38-253: P2/P3/P5/P6, 8 and 16 bit, comments, the gray conversion weights."""
    exe = os.path.join(ROOT, "tests", "cpp", "pgmread_test.bin")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(PKG, "host"),
                           os.path.join(ROOT, "tests", "cpp", "pgmread_test.cpp"), os.path.join(PKG, "host", "pgmread.cpp"),
                           "-o", exe])
    rng = np.random.default_rng(5)
    w, h = 7, 5

    def read(name, payload):
        p = str(tmp_path / name)
        open(p, "wb").write(payload)
        r = subprocess.run([exe, p], capture_output=True, text=True)
        if r.returncode:
            return None, r.stderr
        t = r.stdout.split()
        return np.array(t[2:], int).reshape(int(t[1]), int(t[0])), r.stderr

    g8 = rng.integers(0, 256, (h, w)).astype(np.uint8)
    rgb8 = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    gray = lambda c: ((4899 * c[..., 0].astype(np.int64) + 9617 * c[..., 1].astype(np.int64)
                      + 1868 * c[..., 2].astype(np.int64)) >> 14) & 255
    got, _ = read("a.pgm", b"P5\n# a comment\n%d %d\n255\n" % (w, h) + g8.tobytes())
    assert np.array_equal(got, g8)
    got, _ = read("b.pgm", b"P5 %d %d 255\n" % (w, h) + g8.tobytes())            # one-line header (Netpbm)
    assert np.array_equal(got, g8)
    got, _ = read("c.pgm", b"P2\n%d %d\n255\n" % (w, h) + " ".join(map(str, g8.ravel())).encode() + b"\n")
    assert np.array_equal(got, g8)
    v = rng.integers(0, 1001, (h, w))
    got, _ = read("d.pgm", b"P2\n%d %d\n1000\n" % (w, h) + "\n".join(map(str, v.ravel())).encode() + b"\n")
    assert np.array_equal(got, (v * 255.0 / 1000).astype(np.uint8))
    got, _ = read("e.ppm", b"P6\n%d %d\n255\n" % (w, h) + rgb8.tobytes())
    assert np.array_equal(got, gray(rgb8))
    got, _ = read("f.ppm", b"P3\n# c\n%d %d\n255\n" % (w, h) + " ".join(map(str, rgb8.ravel())).encode() + b"\n")
    assert np.array_equal(got, gray(rgb8))
    g16 = rng.integers(0, 4096, (h, w)).astype(np.uint16)
    got, _ = read("g.pgm", b"P5\n%d %d\n4095\n" % (w, h) + g16.tobytes())       # host byte order, as the reference reads it
    assert np.array_equal(got, (g16 * 255.0 / 4095).astype(np.uint8))
    rgb16 = rng.integers(0, 300, (h, w, 3)).astype(np.uint16)
    got, _ = read("h.ppm", b"P6\n%d %d\n299\n" % (w, h) + rgb16.tobytes())     # not rescaled (pgmread.cpp:226-246)
    assert np.array_equal(got, gray(rgb16))
    for name, payload, msg in (("i.pgm", b"P7\n1 1\n255\n\0", "can only contain"), ("j.pgm", b"P5\n0 4\n255\n", "meaningless"),
                               ("k.pgm", b"P2\n2 2\n255\n1 2 3", "too short"), ("l.pgm", b"P5\nabc\n", "WxH")):
        got, err = read(name, payload)
        assert got is None and msg in err
    got, err = read("missing.pgm", b"")
    os.remove(str(tmp_path / "missing.pgm"))
    r = subprocess.run([exe, str(tmp_path / "missing.pgm")], capture_output=True, text=True)
    assert r.returncode != 0 and "does not exist" in r.stderr
