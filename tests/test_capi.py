"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/popsift_hip.h declares; no compute is attempted here."""
import ctypes as C
import subprocess
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "popsift_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(popsift_hip_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound(hip):
    names = declared_symbols()
    assert len(names) >= 20
    lib = hip.lib()
    bound = {n for n, _, _ in hip.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), "libpopsift_hip.so does not export %s" % n
        assert n in bound, "%s is declared in the header but not bound in _capi.py" % n
    assert bound <= set(names), "bindings for undeclared symbols: %s" % (bound - set(names))


def test_struct_layouts_match_the_header(hip, tmp_path):
    """sizeof / offsetof as a C compiler sees include/popsift_hip.h against the ctypes mirrors of the binding."""
    src = tmp_path / "layout.c"
    src.write_text(r"""
#include <stddef.h>
#include <stdio.h>
#include "popsift_hip.h"
int main(void)
{
    printf("%zu %zu %zu %zu %zu %zu\n", sizeof(popsift_hip_params), sizeof(popsift_hip_feature), sizeof(popsift_hip_extremum),
           sizeof(popsift_hip_report), sizeof(popsift_hip_match), sizeof(popsift_hip_device_info));
    printf("%zu %zu %zu %zu %zu\n", offsetof(popsift_hip_params, store_dog), offsetof(popsift_hip_report, ms_device),
           offsetof(popsift_hip_report, pyramid_pixels), offsetof(popsift_hip_report, ms_stage),
           offsetof(popsift_hip_feature, desc_idx));
    return 0;
}
""")
    exe = str(tmp_path / "layout.bin")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    sizes, offs = [list(map(int, l.split())) for l in subprocess.check_output([exe], text=True).splitlines()]
    assert sizes == [C.sizeof(hip.Params), hip.FEATURE_DTYPE.itemsize, hip.EXTREMUM_DTYPE.itemsize, C.sizeof(hip.Report),
                     hip.MATCH_DTYPE.itemsize, C.sizeof(hip.DeviceInfo)]
    assert offs == [hip.Params.store_dog.offset, hip.Report.ms_device.offset, hip.Report.pyramid_pixels.offset,
                    hip.Report.ms_stage.offset, hip.FEATURE_DTYPE.fields["desc_idx"][1]]
    assert hip.FEATURE_DTYPE.itemsize == 52          # 5 scalars + 4 angles + 4 indices


def test_default_params_are_the_reference_defaults(hip):
    p = hip.default_params()   # sift_conf.cu:17-39
    assert (p.octaves, p.levels, p.max_extrema, p.filter_grid_size) == (-1, 3, 100000, 2)
    assert abs(p.sigma - 1.6) < 1e-7 and p.edge_limit == 10.0 and abs(p.threshold - 0.04) < 1e-8
    assert p.upscale_factor == 1.0 and p.initial_blur == 0.5 and p.assume_initial_blur == 1
    assert (p.sift_mode, p.gauss_mode, p.desc_mode, p.norm_mode, p.norm_multi) == (0, 0, 0, 0, 0)


def test_version_and_error_strings(hip):
    lib = hip.lib()
    assert b"gfx950" in lib.popsift_hip_version()
    assert lib.popsift_hip_strerror(0) == b"ok"
    for code in range(-6, 0):
        assert lib.popsift_hip_strerror(code) not in (b"ok", b"unknown status")
    assert lib.popsift_hip_last_error(None) == b""


def test_invalid_arguments_are_rejected_without_a_gpu(hip):
    lib = hip.lib()
    h = C.c_void_p()
    assert lib.popsift_hip_ctx_create(0, None, C.byref(h)) == hip.ERR_INVALID
    for kw in (dict(sigma=2.5), dict(sigma=0.0), dict(levels=10), dict(gauss_mode=1), dict(desc_mode=5), dict(desc_mode=-1),
               dict(sift_mode=7), dict(norm_mode=3), dict(max_extrema=0), dict(edge_limit=0.0)):
        p = hip.default_params(**kw)
        assert lib.popsift_hip_ctx_create(0, C.byref(p), C.byref(h)) == hip.ERR_INVALID, kw
    assert lib.popsift_hip_ctx_destroy(None) == hip.OK
    assert lib.popsift_hip_wait(None, None, None) == hip.ERR_INVALID
    assert lib.popsift_hip_device_count(None) == hip.ERR_INVALID


def test_no_gpu_means_no_context_not_a_fallback(hip):
    if hip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hip.PopsiftHipError) as e:
        hip.Context()
    assert e.value.status == hip.ERR_NO_DEVICE


def test_missing_library_fails_loudly(monkeypatch, hip):
    import popsift_amd._capi as capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", os.path.join(ROOT, "popsift_amd", "no_such_lib.so"))
    with pytest.raises(ImportError):
        capi.lib()


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "popsift_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(import\s+oracle|from\s+oracle|#include\s*[\"<][^\">]*oracle|popsift_oracle|oracle_[a-z]+\s*\()", txt):
                    offenders.append(os.path.join(base, f))
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            if re.search(r"#include\s*[\"<][^\">]*oracle|oracle_[a-z]+\s*\(", open(os.path.join(base, f), errors="ignore").read()):
                offenders.append(os.path.join(base, f))
    assert offenders == [], offenders
