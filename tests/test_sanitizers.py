"""Sanitizer builds of the CPU-side code (the GPU pool allows no GPU sanitizers): the oracle with its own driver under
ASan + UBSan, and the GPU-free semantics test of the C++ host layer (tests/cpp/host_api_test.cpp) with the host sources
compiled in under ASan + UBSan and under TSan (the reference has no such builds; its debug aid is the
sync-and-check macro of common/debug_macros.h:25-29, see `make SYNC_CHECK=1` in popsift_amd/csrc)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "popsift_amd")
HOST_SRCS = [os.path.join(PKG, "host", f) for f in ("popsift.cpp", "sift_conf.cpp", "features.cpp", "device_prop.cpp", "debug_dump.cpp")]


def test_oracle_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "san"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "oracle", "san", "oracle_san")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "oracle_san: ok" in r.stdout


@pytest.mark.parametrize("name,flags,env", [
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], {"ASAN_OPTIONS": "detect_leaks=0"}),
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1"}),
])
def test_host_layer_under_sanitizers(tmp_path, name, flags, env):
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc")], stdout=subprocess.DEVNULL)
    exe = str(tmp_path / ("host_api_%s.bin" % name))
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "host")]
                          + flags + [os.path.join(ROOT, "tests", "cpp", "host_api_test.cpp")] + HOST_SRCS
                          + ["-o", exe, "-L", PKG, "-lpopsift_hip", "-Wl,-rpath," + PKG])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "host_api_test ok" in r.stdout
