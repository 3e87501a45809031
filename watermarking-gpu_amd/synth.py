"""Synthetic inputs for tests and bench.py: counter-based, no
<random> distributions, so the same (seed, frame, r, c) gives the same sample anywhere.

frame f:  x(r,c) = clamp(128 + 56 sin(2 pi r/97) cos(2 pi c/61) + 36 sin(2 pi (r+2c)/389)
                         + 24 n1 + 6 n2, 0, 255)
          n1 = hash noise box-filtered 4x4 (texture), n2 = white hash noise.
W:        two hash uniforms -> Box-Muller in f64 -> f32, row-major  (the reference's W is an
          N(0,1) matrix, CommonRandomMatrix/main.cpp:41-51; only its row-major f32 layout is
          contractual, Watermark.cpp:62-75).
"""
import numpy as np

SEED = 28390211  # samples/make_w.bat


def _mix(h):
    """32-bit finalizer (lowbias32) on uint64 lanes holding 32-bit values"""
    m = np.uint64(0xFFFFFFFF)
    h = h & m
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & m
    h ^= h >> np.uint64(16)
    return h


def hash_u32(seed, stream, r, c):
    """r, c broadcastable integer arrays -> uint32 hash (as uint64 array)"""
    r = np.asarray(r, np.uint64)
    c = np.asarray(c, np.uint64)
    h = _mix(np.uint64(seed & 0xFFFFFFFF) ^ _mix(np.uint64((stream * 0x9E3779B1) & 0xFFFFFFFF) + r))
    return _mix(h ^ _mix(c + np.uint64(0x85EBCA6B)))


def _uniform_pm1(seed, stream, r, c):
    return hash_u32(seed, stream, r, c).astype(np.float64) * (2.0 / 4294967296.0) - 1.0


def synth_frame(rows, cols, frame=0, seed=SEED, dtype=np.float32):
    """one luminance plane, f32 in [0,255] (image mode) or rounded to u8 (video mode)"""
    r = np.arange(rows + 3, dtype=np.int64)[:, None]
    c = np.arange(cols + 3, dtype=np.int64)[None, :]
    white = _uniform_pm1(seed, 2 * frame + 1, r, c)
    n1 = np.zeros((rows, cols))
    for i in range(4):
        for j in range(4):
            n1 += white[i:i + rows, j:j + cols]
    n1 *= np.sqrt(3.0) / 4.0  # unit variance
    rr = np.arange(rows, dtype=np.float64)[:, None]
    cc = np.arange(cols, dtype=np.float64)[None, :]
    n2 = _uniform_pm1(seed, 2 * frame + 2, r[:rows], c[:, :cols]) * np.sqrt(3.0)
    x = (128.0 + 56.0 * np.sin(2 * np.pi * rr / 97.0) * np.cos(2 * np.pi * cc / 61.0)
         + 36.0 * np.sin(2 * np.pi * (rr + 2 * cc) / 389.0) + 24.0 * n1 + 6.0 * n2)
    x = np.clip(x, 0.0, 255.0)
    if dtype == np.uint8:
        return np.rint(x).astype(np.uint8)
    return x.astype(np.float32)


def synth_watermark(rows, cols, seed=SEED):
    """W ~ N(0,1), f32 row-major [rows, cols]"""
    r = np.arange(rows, dtype=np.int64)[:, None]
    c = np.arange(cols, dtype=np.int64)[None, :]
    u1 = (hash_u32(seed, 0x5741, r, c).astype(np.float64) + 1.0) / 4294967297.0  # (0,1)
    u2 = hash_u32(seed, 0x5742, r, c).astype(np.float64) / 4294967296.0
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2)).astype(np.float32)


def synth_frames_torch(rows, cols, nframes, device, seed=SEED, dtype="f32", first_frame=0):
    """Same recipe as synth_frame(), generated on the GPU with torch (integer hash in int64 lanes) so that
    bench.py can build a batch of distinct 4K frames in milliseconds.  Returns [nframes, rows, cols]
    f32 (image mode) or u8 (video mode).  Values agree with synth_frame() up to the last ulp of sin/cos."""
    import torch

    M = 0xFFFFFFFF

    def mix(h):
        h = h & M
        h = h ^ (h >> 16)
        h = (h * 0x7FEB352D) & M
        h = h ^ (h >> 15)
        h = (h * 0x846CA68B) & M
        h = h ^ (h >> 16)
        return h

    def hash_u32(stream, r, c):
        h = mix((seed & M) ^ mix(((stream * 0x9E3779B1) & M) + r))
        return mix(h ^ mix(c + 0x85EBCA6B))

    r = torch.arange(rows + 3, device=device, dtype=torch.int64)[:, None]
    c = torch.arange(cols + 3, device=device, dtype=torch.int64)[None, :]
    rr = torch.arange(rows, device=device, dtype=torch.float64)[:, None]
    cc = torch.arange(cols, device=device, dtype=torch.float64)[None, :]
    smooth = (128.0 + 56.0 * torch.sin(2 * np.pi * rr / 97.0) * torch.cos(2 * np.pi * cc / 61.0)
              + 36.0 * torch.sin(2 * np.pi * (rr + 2 * cc) / 389.0))
    out = []
    for f in range(first_frame, first_frame + nframes):
        white = hash_u32(2 * f + 1, r, c).to(torch.float64) * (2.0 / 4294967296.0) - 1.0
        n1 = torch.zeros((rows, cols), device=device, dtype=torch.float64)
        for i in range(4):
            for j in range(4):
                n1 += white[i:i + rows, j:j + cols]
        n1 *= np.sqrt(3.0) / 4.0
        n2 = (hash_u32(2 * f + 2, r[:rows], c[:, :cols]).to(torch.float64) * (2.0 / 4294967296.0) - 1.0) * np.sqrt(3.0)
        x = (smooth + 24.0 * n1 + 6.0 * n2).clamp(0.0, 255.0)
        out.append(torch.round(x).to(torch.uint8) if dtype == "u8" else x.to(torch.float32))
    return torch.stack(out)
