"""ctypes loader for the CPU oracle (oracle/libwm_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

MASK_ME, MASK_NVF = 0, 1
OK, UNSOLVABLE = 0, 1


class Opts(C.Structure):
    _fields_ = [("accum_f32", C.c_int), ("fp16_products", C.c_int), ("ref_arith", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libwm_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        u8p = C.POINTER(C.c_uint8)
        op = C.POINTER(Opts)
        L.wmo_strength_factor.restype = C.c_float
        L.wmo_strength_factor.argtypes = [C.c_float]
        L.wmo_gram.argtypes = [fp, C.c_int, C.c_int, dp, dp, op]
        L.wmo_solve.argtypes = [dp, dp, fp]
        L.wmo_solve_f32.argtypes = [dp, dp, fp]
        L.wmo_scaled_neighbors.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.wmo_scaled_neighbors.restype = None
        L.wmo_error_sequence.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.wmo_error_sequence.restype = None
        L.wmo_nvf_mask.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
        L.wmo_me_mask.argtypes = [fp, C.c_int, C.c_int, fp, fp, fp, fp, op]
        L.wmo_embed.argtypes = [fp, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, fp, fp, fp, op]
        L.wmo_detect.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, op]
        L.wmo_embed_u8.argtypes = [u8p, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, u8p, fp, op]
        L.wmo_detect_u8.argtypes = [u8p, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, op]
        L.wmo_rgb2gray.argtypes = [fp, C.c_int, C.c_int, fp]
        L.wmo_rgb2gray.restype = None
        _LIB = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _opts(accum_f32=False, fp16_products=False, ref_arith=False):
    """ref_arith: the reference's own arithmetic for the prediction system (half products, 64-lane f32 work-group sums,
    f32 fold, f32 LU -- wm_oracle.c wmo_opts)"""
    return C.byref(Opts(int(accum_f32), int(fp16_products), int(ref_arith)))


def _c32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def strength_factor(psnr):
    return float(lib().wmo_strength_factor(C.c_float(psnr)))


def gram(x, **kw):
    x = _c32(x)
    Rx = np.zeros((8, 8), np.float64)
    rx = np.zeros(8, np.float64)
    st = lib().wmo_gram(_f(x), x.shape[0], x.shape[1], _d(Rx), _d(rx), _opts(**kw))
    assert st == 0
    return Rx, rx


def solve(Rx, rx):
    Rx = np.ascontiguousarray(Rx, np.float64)
    rx = np.ascontiguousarray(rx, np.float64)
    c = np.zeros(8, np.float32)
    st = lib().wmo_solve(_d(Rx), _d(rx), _f(c))
    return st, c


def scaled_neighbors(x, c):
    x = _c32(x)
    c = _c32(c)
    out = np.empty_like(x)
    lib().wmo_scaled_neighbors(_f(x), x.shape[0], x.shape[1], _f(c), _f(out))
    return out


def error_sequence(x, c):
    x = _c32(x)
    c = _c32(c)
    out = np.empty_like(x)
    lib().wmo_error_sequence(_f(x), x.shape[0], x.shape[1], _f(c), _f(out))
    return out


def nvf_mask(x, p=3):
    x = _c32(x)
    m = np.empty_like(x)
    st = lib().wmo_nvf_mask(_f(x), x.shape[0], x.shape[1], p, _f(m))
    if st != 0:
        raise ValueError("bad p")
    return m


def me_mask(x, **kw):
    """returns (status, c, e, m, max|e|)"""
    x = _c32(x)
    c = np.zeros(8, np.float32)
    e = np.empty_like(x)
    m = np.empty_like(x)
    mx = C.c_float(0)
    st = lib().wmo_me_mask(_f(x), x.shape[0], x.shape[1], _f(c), _f(e), _f(m), C.byref(mx), _opts(**kw))
    return st, c, e, m, mx.value


def embed(gray, base, W, p=3, psnr=40.0, mask=MASK_ME, want_mask=False, **kw):
    """base: [rows,cols] or [channels,rows,cols] planar.  returns (status, out, a[, mask])"""
    gray = _c32(gray)
    base = _c32(base)
    W = _c32(W)
    ch = 1 if base.ndim == 2 else base.shape[0]
    out = np.empty_like(base)
    a = C.c_float(float("nan"))
    mo = np.empty_like(gray) if want_mask else None
    st = lib().wmo_embed(_f(gray), _f(base), ch, _f(W), gray.shape[0], gray.shape[1], p, C.c_float(psnr), mask,
                         _f(out), C.byref(a), _f(mo) if want_mask else None, _opts(**kw))
    if want_mask:
        return st, out, a.value, mo
    return st, out, a.value


def detect(img, W, p=3, mask=MASK_ME, **kw):
    img = _c32(img)
    W = _c32(W)
    corr = C.c_float(0)
    st = lib().wmo_detect(_f(img), _f(W), img.shape[0], img.shape[1], p, mask, C.byref(corr), _opts(**kw))
    return st, corr.value


def embed_u8(frame, W, p=3, psnr=40.0, mask=MASK_ME, **kw):
    frame = np.ascontiguousarray(frame, np.uint8)
    W = _c32(W)
    out = np.empty_like(frame)
    a = C.c_float(float("nan"))
    st = lib().wmo_embed_u8(_u8(frame), _f(W), frame.shape[0], frame.shape[1], p, C.c_float(psnr), mask, _u8(out),
                            C.byref(a), _opts(**kw))
    return st, out, a.value


def detect_u8(frame, W, p=3, mask=MASK_ME, **kw):
    frame = np.ascontiguousarray(frame, np.uint8)
    W = _c32(W)
    corr = C.c_float(0)
    st = lib().wmo_detect_u8(_u8(frame), _f(W), frame.shape[0], frame.shape[1], p, mask, C.byref(corr), _opts(**kw))
    return st, corr.value


def rgb2gray(rgb_planar):
    rgb = _c32(rgb_planar)
    g = np.empty(rgb.shape[1:], np.float32)
    lib().wmo_rgb2gray(_f(rgb), rgb.shape[1], rgb.shape[2], _f(g))
    return g
