"""Independent numpy restatement of the reference's watermark maths.

Written array-at-a-time from the formulas (not from oracle/wm_oracle.c) so that the two
restatements can check each other.  Sums in f64; element-wise maths in f32.
Reference: Watermark.cpp:156-250, kernels/{nvf,me_p3,scaled_neighbors_p3}.hpp.
"""
import numpy as np

# neighbour offsets in the reference's order x0..x8 without the centre (me_p3.hpp:46-54)
OFFS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def neighbours(x):
    """[8, R, C] replicate-padded neighbour planes"""
    xp = np.pad(x, 1, mode="edge")
    R, C = x.shape
    return np.stack([xp[1 + dr:1 + dr + R, 1 + dc:1 + dc + C] for dr, dc in OFFS])


def gram(x):
    n = neighbours(x.astype(np.float32)).astype(np.float64).reshape(8, -1)
    Rx = n @ n.T
    rx = n @ x.astype(np.float64).reshape(-1)
    return Rx, rx


def coefficients(x):
    Rx, rx = gram(x)
    return np.linalg.solve(Rx, rx).astype(np.float32)


def scaled_neighbors(x, c):
    n = neighbours(x.astype(np.float32))
    dot = np.zeros_like(x, dtype=np.float32)
    for k in range(8):
        # fused multiply-add emulated in f64 then rounded once (exact for f32 operands
        # up to double rounding, which is negligible for this cross-check)
        dot = (np.float64(c[k]) * n[k].astype(np.float64) + dot.astype(np.float64)).astype(np.float32)
    return dot


def error_sequence(x, c):
    return (x.astype(np.float32) - scaled_neighbors(x, c)).astype(np.float32)


def nvf_mask(x, p=3):
    pad = p // 2
    xp = np.pad(x.astype(np.float32), pad, mode="edge")
    R, C = x.shape
    s = np.zeros((R, C), np.float32)
    ss = np.zeros((R, C), np.float32)
    for i in range(p):
        for j in range(p):
            v = xp[i:i + R, j:j + C]
            s = (s + v).astype(np.float32)
            ss = (v.astype(np.float64) * v.astype(np.float64) + ss.astype(np.float64)).astype(np.float32)
    psq = np.float32(p * p)
    mean = (s / psq).astype(np.float32)
    var = ((ss / psq).astype(np.float32) - (mean * mean).astype(np.float32)).astype(np.float32)
    return (var / (np.float32(1) + var)).astype(np.float32)


def strength_factor(psnr):
    return np.float32(255.0) / np.sqrt(np.power(np.float32(10.0), np.float32(psnr) / np.float32(10.0), dtype=np.float32),
                                       dtype=np.float32)


def embed(gray, base, W, p=3, psnr=40.0, mask="ME"):
    gray = gray.astype(np.float32)
    if mask == "ME":
        c = coefficients(gray)
        e = error_sequence(gray, c)
        ae = np.abs(e)
        m = (ae / ae.max()).astype(np.float32)
    else:
        m = nvf_mask(gray, p)
    u = (m * W.astype(np.float32)).astype(np.float32)
    nrm = np.sqrt(np.sum(u.astype(np.float64) ** 2))
    a = np.float32(strength_factor(psnr) / np.float32(nrm / np.sqrt(float(gray.size))))
    y = (u.astype(np.float64) * np.float64(a) + base.astype(np.float64)).astype(np.float32)
    return np.clip(y, 0, 255).astype(np.float32), float(a)


def detect(img, W, p=3, mask="ME"):
    img = img.astype(np.float32)
    c = coefficients(img)
    ew = error_sequence(img, c)
    if mask == "ME":
        ae = np.abs(ew)
        m = (ae / ae.max()).astype(np.float32)
    else:
        m = nvf_mask(img, p)
    u = (m * W.astype(np.float32)).astype(np.float32)
    eu = error_sequence(u, c)
    d = np.sum(eu.astype(np.float64) * ew.astype(np.float64))
    nz = np.sqrt(np.sum(ew.astype(np.float64) ** 2))
    nu = np.sqrt(np.sum(eu.astype(np.float64) ** 2))
    return float(np.float32(d) / np.float32(nz * nu))
