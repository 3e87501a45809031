"""MI355X-native watermark engine: Python host-side mirror of the reference's `Watermark` class.

The reference is C++ (Watermark_GPU/Watermark.hpp:26-72); its C++ drop-in is include/Watermark.hpp.
This module is the same surface for Python callers (tests, bench.py): same class and method names,
argument order and error behaviour, over the C ABI of include/wm.h loaded with ctypes.  Arrays are
torch CUDA tensors (torch is plumbing here: device memory, streams, torch.distributed) standing in
for af::array: [rows, cols] grey, [3, rows, cols] planar RGB, or a batch [frames, rows, cols].

There is NO CPU fallback: without libwm_hip.so or without a HIP device every operation raises.
"""
import ctypes as C
import enum
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwm_hip.so")

WM_OK, WM_UNSOLVABLE = 0, 1
WM_ERR_BAD_P, WM_ERR_W_OPEN, WM_ERR_W_SIZE, WM_ERR_RUNTIME = -1, -2, -3, -4
WM_ERR_BAD_ARG, WM_ERR_NO_DEVICE, WM_ERR_ALLOC, WM_ERR_PSNR, WM_ERR_BUSY = -5, -6, -7, -8, -9
WM_SLOT_SYNC = -1
WM_F32, WM_U8 = 0, 1
WM_MEM_DEVICE, WM_MEM_HOST, WM_MEM_SLOT_OUT = 0, 1, 2


class MASK_TYPE(enum.IntEnum):
    """Watermark.hpp:10-14"""
    ME = 0
    NVF = 1


class wm_plane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("channels", C.c_int32),
                ("dtype", C.c_int32), ("mem", C.c_int32), ("frames", C.c_int32), ("pitch", C.c_int64),
                ("channel_stride", C.c_int64), ("frame_stride", C.c_int64)]


# every symbol include/wm.h declares: (name, restype, argtypes)
_P = C.POINTER
_ctx_p = C.c_void_p
ABI = [
    ("wm_create", C.c_int, [_P(_ctx_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P(C.c_float)]),
    ("wm_create_from_file", C.c_int, [_P(_ctx_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_char_p]),
    ("wm_create_generated", C.c_int, [_P(_ctx_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32]),
    ("wm_clone", C.c_int, [_ctx_p, _P(_ctx_p)]),
    ("wm_reinit", C.c_int, [_ctx_p, C.c_int, C.c_int, _P(C.c_float)]),
    ("wm_reinit_from_file", C.c_int, [_ctx_p, C.c_int, C.c_int, C.c_char_p]),
    ("wm_destroy", None, [_ctx_p]),
    ("wm_configure", C.c_int, [_ctx_p, C.c_int, C.c_int]),
    ("wm_set_fused", C.c_int, [_ctx_p, C.c_int]),
    ("wm_set_handover", C.c_int, [_ctx_p, C.c_int]),
    ("wm_fused_info", C.c_int, [_ctx_p, _P(C.c_int), _P(C.c_int), _P(C.c_ulonglong)]),
    ("wm_fused_lock_skips", C.c_ulonglong, [_ctx_p]),
    ("wm_fused_stamps", C.c_int, [_ctx_p, _P(C.c_ulonglong), C.c_int]),
    ("wm_fused_gram", C.c_int, [_ctx_p, _P(C.c_double)]),
    ("wm_selftest_nvf_quotient", C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint32, _P(C.c_ulonglong), _P(C.c_uint32)]),
    ("wm_set_rows_per_segment", C.c_int, [_ctx_p, C.c_int]),
    ("wm_embed", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), _P(wm_plane), _P(wm_plane), _P(C.c_float), _P(C.c_int), C.c_int]),
    ("wm_detect", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), _P(C.c_float), _P(C.c_int), C.c_int]),
    ("wm_embed_detect", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), _P(wm_plane), _P(wm_plane), _P(C.c_float), _P(C.c_float), _P(C.c_int), C.c_int]),
    ("wm_compute_mask", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), _P(wm_plane), _P(wm_plane), _P(C.c_float), _P(C.c_int), C.c_int]),
    ("wm_gram", C.c_int, [_ctx_p, _P(wm_plane), _P(C.c_double), C.c_int]),
    ("wm_band_configure", C.c_int, [_ctx_p, C.c_int, C.c_int, C.c_longlong]),
    ("wm_band_solve", C.c_int, [_ctx_p, _P(C.c_double), C.c_int, _P(C.c_int), C.c_int]),
    ("wm_band_stats", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), _P(C.c_double), C.c_int]),
    ("wm_band_embed", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), _P(wm_plane), _P(wm_plane), _P(C.c_double), _P(C.c_float), C.c_int]),
    ("wm_band_detect_sums", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), _P(C.c_double), C.c_int]),
    ("wm_band_gram_dev", C.c_int, [_ctx_p, _P(wm_plane), C.c_void_p, C.c_int]),
    ("wm_band_solve_dev", C.c_int, [_ctx_p, C.c_void_p, C.c_int, C.c_int]),
    ("wm_band_stats_dev", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), C.c_void_p, C.c_int]),
    ("wm_band_embed_dev", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), _P(wm_plane), _P(wm_plane), C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    ("wm_band_detect_sums_dev", C.c_int, [_ctx_p, C.c_int, _P(wm_plane), C.c_void_p, C.c_int]),
    ("wm_band_corr_dev", C.c_int, [_ctx_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    ("wm_sync", C.c_int, [_ctx_p, C.c_int]),
    ("wm_set_stream", C.c_int, [_ctx_p, C.c_int, C.c_void_p]),
    ("wm_get_stream", C.c_void_p, [_ctx_p, C.c_int]),
    ("wm_dev_alloc", C.c_void_p, [C.c_int, C.c_size_t]),
    ("wm_dev_free", None, [C.c_void_p]),
    ("wm_memcpy_h2d", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("wm_memcpy_d2h", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("wm_device_count", C.c_int, []),
    ("wm_host_alloc", C.c_void_p, [C.c_size_t]),
    ("wm_host_free", None, [C.c_void_p]),
    ("wm_membench", C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_double, _P(C.c_double), _P(C.c_int)]),
    ("wm_rows", C.c_int, [_ctx_p]),
    ("wm_cols", C.c_int, [_ctx_p]),
    ("wm_p", C.c_int, [_ctx_p]),
    ("wm_strength_factor", C.c_float, [_ctx_p]),
    ("wm_device", C.c_int, [_ctx_p]),
    ("wm_w_device", C.c_void_p, [_ctx_p]),
    ("wm_prof_enable", C.c_int, [_ctx_p, C.c_int]),
    ("wm_prof_reset", C.c_int, [_ctx_p]),
    ("wm_prof_kernel_count", C.c_int, []),
    ("wm_prof_kernel_name", C.c_char_p, [C.c_int]),
    ("wm_prof_get", C.c_int, [_ctx_p, C.c_int, _P(C.c_uint64), _P(C.c_double)]),
    ("wm_strerror", C.c_char_p, [C.c_int]),
    ("wm_last_error", C.c_char_p, [_ctx_p]),
    ("wm_version", C.c_char_p, []),
]

_lib = None


def lib():
    """loads libwm_hip.so; raises (loudly) if the HIP extension has not been built"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C watermarking-gpu_amd/csrc).  There is no CPU fallback.")
        # one HIP runtime per process: PyTorch ships its own libamdhip64 and this module hands torch tensors to the
        # library, so torch's runtime has to be the one libwm_hip.so binds to -- load torch first, whatever the
        # import order of the caller (a second runtime loaded afterwards sees no device)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def strerror(code):
    return lib().wm_strerror(code).decode()


def _raise(code, ctx=None):
    detail = ""
    if ctx:
        detail = lib().wm_last_error(ctx).decode()
    msg = strerror(code) + (": " + detail if detail else "")
    # the reference throws std::runtime_error for all of these (Watermark.cpp:24-25,65-66,70-71,111-113)
    raise RuntimeError(msg)


def plane_of(t, channels=1, batched=None):
    """wm_plane view of a torch CUDA tensor: [R,C], [3,R,C] (channels=3) or [F,R,C] / [F,3,R,C]"""
    import torch
    if not t.is_cuda:
        raise RuntimeError("plane tensors must live on the GPU (no CPU fallback)")
    if t.dtype == torch.float32:
        dt = WM_F32
    elif t.dtype == torch.uint8:
        dt = WM_U8
    else:
        raise RuntimeError(f"unsupported dtype {t.dtype}")
    if t.stride(-1) != 1:
        raise RuntimeError("innermost stride must be 1 (row-major planes)")
    nd = t.dim()
    base_nd = 2 if channels == 1 else 3
    if nd == base_nd:
        frames, fstride = 1, 0
    elif nd == base_nd + 1:
        frames, fstride = t.shape[0], t.stride(0)
    else:
        raise RuntimeError(f"bad tensor rank {nd} for channels={channels}")
    cstride = t.stride(-3) if channels > 1 else 0
    if channels > 1 and t.shape[-3] != channels:
        raise RuntimeError("channel dimension mismatch")
    return wm_plane(t.data_ptr(), t.shape[-2], t.shape[-1], channels, dt, WM_MEM_DEVICE, frames, t.stride(-2), cstride,
                    fstride)


class Watermark:
    """Functions for watermark computation and detection (Watermark.hpp:26-72).

    Watermark(rows, cols, randomMatrixPath, p, psnr): `randomMatrixPath` is the raw f32 W file
    (Watermark.cpp:62-75) or a numpy array [rows, cols].  The reference's `programs` argument
    (pre-built OpenCL programs) has no counterpart: kernels are compiled into libwm_hip.so.
    """

    def __init__(self, rows, cols, randomMatrixPath, p, psnr, device=0, nslots=2, max_frames=1):
        L = lib()
        self._ctx = _ctx_p()
        if isinstance(randomMatrixPath, (str, bytes, os.PathLike)):
            rc = L.wm_create_from_file(C.byref(self._ctx), device, rows, cols, p, psnr, os.fsencode(randomMatrixPath))
        else:
            w = np.ascontiguousarray(randomMatrixPath, dtype=np.float32)
            if w.size != rows * cols:
                _raise(WM_ERR_W_SIZE)
            rc = L.wm_create(C.byref(self._ctx), device, rows, cols, p, psnr, w.ctypes.data_as(_P(C.c_float)))
        if rc != WM_OK:
            self._ctx = _ctx_p()
            _raise(rc)
        if (nslots, max_frames) != (2, 1):
            self.configure(nslots, max_frames)

    @classmethod
    def generated(cls, rows, cols, seed, p, psnr, device=0, nslots=2, max_frames=1):
        """an engine whose W is generated on the device from `seed` (wm.h wm_create_generated; the matrix wm_genw writes)"""
        self = object.__new__(cls)
        self._ctx = _ctx_p()
        rc = lib().wm_create_generated(C.byref(self._ctx), device, rows, cols, p, psnr, seed & 0xFFFFFFFF)
        if rc != WM_OK:
            self._ctx = _ctx_p()
            _raise(rc)
        if (nslots, max_frames) != (2, 1):
            self.configure(nslots, max_frames)
        return self

    def watermark(self):
        """the engine's W as a numpy array [rows, cols] (downloaded)"""
        w = np.empty((self.rows, self.cols), np.float32)
        rc = lib().wm_memcpy_d2h(w.ctypes.data_as(C.c_void_p), lib().wm_w_device(self._ctx), w.nbytes)
        if rc != WM_OK:
            _raise(rc, self._ctx)
        return w

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            lib().wm_destroy(self._ctx)
            self._ctx = _ctx_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def copy(self):
        """copy constructor (Watermark.cpp:30-37): shares W, owns new scratch"""
        other = object.__new__(Watermark)
        other._ctx = _ctx_p()
        rc = lib().wm_clone(self._ctx, C.byref(other._ctx))
        if rc != WM_OK:
            _raise(rc, self._ctx)
        return other

    def reinitialize(self, randomMatrixPath, rows, cols):
        """Watermark.cpp:78-85"""
        if isinstance(randomMatrixPath, (str, bytes, os.PathLike)):
            rc = lib().wm_reinit_from_file(self._ctx, rows, cols, os.fsencode(randomMatrixPath))
        else:
            w = np.ascontiguousarray(randomMatrixPath, dtype=np.float32)
            if w.size != rows * cols:
                _raise(WM_ERR_W_SIZE)
            rc = lib().wm_reinit(self._ctx, rows, cols, w.ctypes.data_as(_P(C.c_float)))
        if rc != WM_OK:
            _raise(rc, self._ctx)

    def configure(self, nslots, max_frames):
        rc = lib().wm_configure(self._ctx, nslots, max_frames)
        if rc != WM_OK:
            _raise(rc, self._ctx)

    def set_fused(self, on):
        """one-frame synchronous calls as ONE launch with LDS-resident tiles (wm.h wm_set_fused); on by default"""
        rc = lib().wm_set_fused(self._ctx, 1 if on else 0)
        if rc != WM_OK:
            _raise(rc, self._ctx)

    def set_handover(self, on):
        """Gram hand-over from embed to a detector reading WM_MEM_SLOT_OUT (wm.h wm_set_handover); off by default"""
        rc = lib().wm_set_handover(self._ctx, 1 if on else 0)
        if rc != WM_OK:
            _raise(rc, self._ctx)

    def fused_info(self):
        """(active, workgroups, tile_rows, fallbacks)"""
        g, th, fb = C.c_int(), C.c_int(), C.c_ulonglong()
        act = lib().wm_fused_info(self._ctx, C.byref(g), C.byref(th), C.byref(fb))
        return bool(act), g.value, th.value, fb.value

    def set_rows_per_segment(self, rps):
        rc = lib().wm_set_rows_per_segment(self._ctx, rps)
        if rc != WM_OK:
            _raise(rc, self._ctx)

    # -- properties -------------------------------------------------------------------------
    @property
    def rows(self):
        return lib().wm_rows(self._ctx)

    @property
    def cols(self):
        return lib().wm_cols(self._ctx)

    @property
    def strengthFactor(self):
        return lib().wm_strength_factor(self._ctx)

    # -- the hot path -------------------------------------------------------------------------
    def makeWatermark(self, inputImage, outputImage, maskType, out=None):
        """Watermark.cpp:156-172.  inputImage: grey [R,C] (or batch [F,R,C]); outputImage: the base the
        watermark is added to ([R,C], [3,R,C] or batched).  Returns (watermarked, watermarkStrength);
        the reference returns the array and writes the strength through a float& argument.
        Unsolvable system: returns outputImage unchanged and strength None (reference leaves it unset)."""
        import torch
        rgb = outputImage.dim() - inputImage.dim() == 1
        pin = plane_of(inputImage, 1)
        pbase = plane_of(outputImage, 3 if rgb else 1)
        if out is None:
            out = torch.empty_like(outputImage)
        pout = plane_of(out, 3 if rgb else 1)
        frames = pin.frames
        a = (C.c_float * frames)(*([float("nan")] * frames))
        st = (C.c_int * frames)()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_embed(self._ctx, int(maskType), C.byref(pin), C.byref(pbase), C.byref(pout), a, st, WM_SLOT_SYNC)
        if rc < 0:
            _raise(rc, self._ctx)
        if inputImage.dim() == 2:
            return out, (None if st[0] != 0 else a[0])
        return out, [None if st[f] != 0 else a[f] for f in range(frames)]

    def detectWatermark(self, watermarkedImage, maskType):
        """Watermark.cpp:234-250; 0.0 for an unsolvable system"""
        import torch
        pimg = plane_of(watermarkedImage, 1)
        frames = pimg.frames
        corr = (C.c_float * frames)()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_detect(self._ctx, int(maskType), C.byref(pimg), corr, None, WM_SLOT_SYNC)
        if rc < 0:
            _raise(rc, self._ctx)
        if watermarkedImage.dim() == 2:
            return corr[0]
        return list(corr)

    def makeAndDetect(self, inputImage, outputImage, maskType, out=None):
        """makeWatermark followed by detectWatermark on its result (testForImage's pair, main.cpp:165-220) as one call
        (wm.h wm_embed_detect; grey output).  Returns (watermarked, strength or None, correlation)."""
        import torch
        pin, pbase = plane_of(inputImage, 1), plane_of(outputImage, 1)
        if out is None:
            out = torch.empty_like(outputImage)
        pout = plane_of(out, 1)
        frames = pin.frames
        a = (C.c_float * frames)(*([float("nan")] * frames))
        corr = (C.c_float * frames)()
        st = (C.c_int * frames)()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_embed_detect(self._ctx, int(maskType), C.byref(pin), C.byref(pbase), C.byref(pout), a, corr, st, WM_SLOT_SYNC)
        if rc < 0:
            _raise(rc, self._ctx)
        if inputImage.dim() == 2:
            return out, (None if st[0] != 0 else a[0]), corr[0]
        return out, [None if st[f] != 0 else a[f] for f in range(frames)], list(corr)

    # north_star aliases
    embed = makeWatermark
    detect = detectWatermark

    # -- asynchronous slot interface (frames in flight; wm.h) --------------------------------------
    # `inputImage` / `outputImage` / `out` / `image` may be torch tensors or wm_plane structs prepared once with
    # plane_of(): a streaming loop that reuses its frame buffers should pass planes (no per-call tensor walk)
    @staticmethod
    def _as_plane(t, channels):
        return t if isinstance(t, wm_plane) else plane_of(t, channels)

    def embed_async(self, inputImage, outputImage, out, maskType, slot, a_out=None, status_out=None):
        if isinstance(outputImage, wm_plane):
            ch = outputImage.channels
        else:
            ch = 3 if outputImage.dim() - inputImage.dim() == 1 else 1
        pin = self._as_plane(inputImage, 1)
        pbase = self._as_plane(outputImage, ch)
        pout = self._as_plane(out, ch)
        rc = lib().wm_embed(self._ctx, int(maskType), C.byref(pin), C.byref(pbase), C.byref(pout), a_out, status_out, slot)
        if rc < 0:
            _raise(rc, self._ctx)

    def detect_async(self, image, maskType, slot, corr_out=None, status_out=None):
        pimg = self._as_plane(image, 1)
        rc = lib().wm_detect(self._ctx, int(maskType), C.byref(pimg), corr_out, status_out, slot)
        if rc < 0:
            _raise(rc, self._ctx)

    def sync(self, slot):
        rc = lib().wm_sync(self._ctx, slot)
        if rc < 0:
            _raise(rc, self._ctx)
        return rc

    # -- parity-test building blocks (private in the reference: Watermark.cpp:96-114,176-218) -------------
    def computeMask(self, inputImage, maskType, want_error_sequence=False):
        """returns (mask, e or None, coefficients[8] or None, status)"""
        import torch
        pin = plane_of(inputImage, 1)
        m = torch.empty(inputImage.shape, dtype=torch.float32, device=inputImage.device)
        e = torch.empty_like(m) if want_error_sequence else None
        pm = plane_of(m, 1)
        pe = plane_of(e, 1) if e is not None else None
        frames = pin.frames
        coef = (C.c_float * (8 * frames))()
        st = (C.c_int * frames)()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_compute_mask(self._ctx, int(maskType), C.byref(pin), C.byref(pm), C.byref(pe) if pe else None, coef,
                                   st, WM_SLOT_SYNC)
        if rc < 0:
            _raise(rc, self._ctx)
        c = np.array(coef[:], dtype=np.float32).reshape(frames, 8)
        if inputImage.dim() == 2:
            return m, e, c[0], st[0]
        return m, e, c, list(st)

    def gram(self, image):
        """(Rx [8,8] f64, rx [8] f64) of a grey image: the sums the me kernel + af::sum produce"""
        import torch
        pimg = plane_of(image, 1)
        buf = (C.c_double * (44 * pimg.frames))()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_gram(self._ctx, C.byref(pimg), buf, 0)
        if rc < 0:
            _raise(rc, self._ctx)
        tot = np.array(buf[:44], dtype=np.float64)
        Rx = np.zeros((8, 8))
        k = 0
        for i in range(8):
            for j in range(i, 8):
                Rx[i, j] = Rx[j, i] = tot[k]
                k += 1
        return Rx, tot[36:].copy()

    # -- row-band building blocks (intra-frame sharding, wm.h wm_band_*; orchestration in bands.py) --------------
    def gram_totals(self, image):
        """the 44 Gram sums of `image` (of the owned rows in band mode) as a float64 array"""
        import torch
        pimg = plane_of(image, 1)
        buf = (C.c_double * (44 * pimg.frames))()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_gram(self._ctx, C.byref(pimg), buf, 0)
        if rc < 0:
            _raise(rc, self._ctx)
        return np.array(buf[:], dtype=np.float64)

    def band_configure(self, own_lo, own_hi, rows_global):
        rc = lib().wm_band_configure(self._ctx, int(own_lo), int(own_hi), int(rows_global))
        if rc < 0:
            _raise(rc, self._ctx)

    def band_solve(self, totals):
        t = np.ascontiguousarray(totals, dtype=np.float64)
        st = (C.c_int * 1)()
        rc = lib().wm_band_solve(self._ctx, t.ctypes.data_as(_P(C.c_double)), 1, st, 0)
        if rc < 0:
            _raise(rc, self._ctx)
        return st[0]

    def band_stats(self, image, maskType):
        import torch
        pimg = plane_of(image, 1)
        out = (C.c_double * 2)()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_band_stats(self._ctx, int(maskType), C.byref(pimg), out, 0)
        if rc < 0:
            _raise(rc, self._ctx)
        return out[0], out[1]

    def band_embed(self, image, base, out, maskType, max_e, ss):
        import torch
        ch = 3 if base.dim() - image.dim() == 1 else 1
        pin, pbase, pout = plane_of(image, 1), plane_of(base, ch), plane_of(out, ch)
        ms = (C.c_double * 2)(max_e, ss)
        a = (C.c_float * 1)()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_band_embed(self._ctx, int(maskType), C.byref(pin), C.byref(pbase), C.byref(pout), ms, a, 0)
        if rc < 0:
            _raise(rc, self._ctx)
        return a[0]

    def band_detect_sums(self, image, maskType):
        import torch
        pimg = plane_of(image, 1)
        out = (C.c_double * 3)()
        torch.cuda.current_stream().synchronize()
        rc = lib().wm_band_detect_sums(self._ctx, int(maskType), C.byref(pimg), out, 0)
        if rc < 0:
            _raise(rc, self._ctx)
        return out[0], out[1], out[2]

    # -- the same with the exchange resident in device memory (wm.h wm_band_*_dev): device tensors in, nothing synchronises.
    # The slot runs on torch's current stream (set_stream_current), so torch.distributed collectives order with the sweeps
    def set_stream_current(self, slot=0):
        import torch
        # torch's default stream has the handle 0, which wm_set_stream reads as "back to the slot's own stream": name the legacy
        # default stream by HIP's handle for it (hipStreamLegacy = 1)
        h = torch.cuda.current_stream().cuda_stream
        rc = lib().wm_set_stream(self._ctx, slot, C.c_void_p(h if h else 1))
        if rc < 0:
            _raise(rc, self._ctx)

    def _chk(self, rc):
        if rc < 0:
            _raise(rc, self._ctx)

    def band_gram_dev(self, image, totals):
        """totals: float64 CUDA tensor [44] (one frame): receives this band's Gram sums"""
        pimg = plane_of(image, 1)
        self._chk(lib().wm_band_gram_dev(self._ctx, C.byref(pimg), C.c_void_p(totals.data_ptr()), 0))

    def band_solve_dev(self, totals):
        self._chk(lib().wm_band_solve_dev(self._ctx, C.c_void_p(totals.data_ptr()), 1, 0))

    def band_stats_dev(self, image, maskType, max_sum):
        pimg = plane_of(image, 1)
        self._chk(lib().wm_band_stats_dev(self._ctx, int(maskType), C.byref(pimg), C.c_void_p(max_sum.data_ptr()), 0))

    def band_embed_dev(self, image, base, out, maskType, gathered, nparts, a_dev):
        ch = 3 if base.dim() - image.dim() == 1 else 1
        pin, pbase, pout = plane_of(image, 1), plane_of(base, ch), plane_of(out, ch)
        self._chk(lib().wm_band_embed_dev(self._ctx, int(maskType), C.byref(pin), C.byref(pbase), C.byref(pout), C.c_void_p(gathered.data_ptr()), nparts,
                                           C.c_void_p(a_dev.data_ptr()), 0))

    def band_detect_sums_dev(self, image, maskType, sums):
        pimg = plane_of(image, 1)
        self._chk(lib().wm_band_detect_sums_dev(self._ctx, int(maskType), C.byref(pimg), C.c_void_p(sums.data_ptr()), 0))

    def band_corr_dev(self, sums, corr):
        self._chk(lib().wm_band_corr_dev(self._ctx, C.c_void_p(sums.data_ptr()), 1, C.c_void_p(corr.data_ptr()), 0))

    # -- profiling ----------------------------------------------------------------------------
    def prof_enable(self, on=True):
        lib().wm_prof_enable(self._ctx, 1 if on else 0)

    def prof_reset(self):
        lib().wm_prof_reset(self._ctx)

    def prof_report(self):
        """{kernel name: (launches, total ms)} measured with hipEvents on the launch stream"""
        out = {}
        L = lib()
        for k in range(L.wm_prof_kernel_count()):
            n = C.c_uint64()
            ms = C.c_double()
            L.wm_prof_get(self._ctx, k, C.byref(n), C.byref(ms))
            if n.value:
                out[L.wm_prof_kernel_name(k).decode()] = (n.value, ms.value)
        return out
