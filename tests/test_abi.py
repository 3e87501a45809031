"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/wm.h
declares, the Python mirror binds all of them, and host-side argument checking works without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "wm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wm_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib(wm):
    if not os.path.exists(wm.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(wm.LIB_PATH), "csrc")])
    return wm.lib()


def test_header_declares_expected_surface():
    syms = header_symbols()
    for need in ("wm_create", "wm_create_from_file", "wm_clone", "wm_reinit", "wm_destroy", "wm_embed", "wm_detect",
                 "wm_sync", "wm_strerror"):
        assert need in syms


def test_library_exports_every_header_symbol(built_lib, wm):
    syms = header_symbols()
    bound = {name for name, _, _ in wm.ABI}
    for s in syms:
        assert hasattr(built_lib, s), f"libwm_hip.so does not export {s}"
        assert s in bound, f"python mirror does not bind {s}"
    assert bound <= set(syms)


def test_strerror_and_version(built_lib, wm):
    assert wm.strerror(wm.WM_OK) == "ok"
    assert "p parameter" in wm.strerror(wm.WM_ERR_BAD_P)  # "Wrong p parameter" (Watermark.cpp:25)
    assert "W file total elements != image dimensions" in wm.strerror(wm.WM_ERR_W_SIZE)  # Watermark.cpp:71
    assert b"gfx950" in built_lib.wm_version()
    assert built_lib.wm_prof_kernel_count() >= 6
    names = [built_lib.wm_prof_kernel_name(i).decode() for i in range(built_lib.wm_prof_kernel_count())]
    assert "k_gram" in names and "k_detect" in names and "k_embed" in names


def test_argument_errors_need_no_gpu(built_lib, wm, tmp_path):
    ctx = C.c_void_p()
    import numpy as np
    w = np.zeros(64 * 64, np.float32)
    wp = w.ctypes.data_as(C.POINTER(C.c_float))
    # bad p / bad psnr are rejected before any device is touched (Watermark.cpp:24-25, main.cpp:96)
    assert built_lib.wm_create(C.byref(ctx), 0, 64, 64, 4, 40.0, wp) == wm.WM_ERR_BAD_P
    assert built_lib.wm_create(C.byref(ctx), 0, 64, 64, 3, -1.0, wp) == wm.WM_ERR_PSNR
    assert built_lib.wm_create(C.byref(ctx), 0, 0, 64, 3, 40.0, wp) == wm.WM_ERR_BAD_ARG
    # W file errors (Watermark.cpp:65-66,70-71)
    assert built_lib.wm_create_from_file(C.byref(ctx), 0, 64, 64, 3, 40.0, b"/nonexistent/w.dat") == wm.WM_ERR_W_OPEN
    f = tmp_path / "w_short.dat"
    f.write_bytes(b"\0" * 100)
    assert built_lib.wm_create_from_file(C.byref(ctx), 0, 64, 64, 3, 40.0, str(f).encode()) == wm.WM_ERR_W_SIZE
    assert not ctx.value


def test_no_cpu_fallback_without_device(built_lib, wm):
    """on a box without a GPU the engine refuses to construct (it must never fall back to the oracle)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    ctx = C.c_void_p()
    w = np.zeros(64 * 64, np.float32)
    rc = built_lib.wm_create(C.byref(ctx), 0, 64, 64, 3, 40.0, w.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == wm.WM_ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        wm.Watermark(64, 64, w.reshape(64, 64), 3, 40.0)


def test_product_does_not_reference_oracle():
    """the product tree must not include, import, link or load anything under oracle/"""
    bad = re.compile(r"#\s*include[^\n]*oracle|import[^\n]*oracle|libwm_oracle|oracle_lib|dlopen|-lwm_oracle|oracle/[^\n]*\.so")
    pkg = os.path.join(ROOT, "watermarking-gpu_amd")
    seen = 0
    for top in (pkg, os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(top):
            for fn in files:
                if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or fn == "Makefile":
                    txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                    assert not bad.search(txt), fn
                    seen += 1
    assert seen >= 5
