"""Generates the committed golden fixtures from the reference's SAMPLE DATA (images + W files).

Run only in the build container (needs /root/reference and Pillow):
    python tests/golden/make_golden.py

Inputs are data files of the reference (MIT licence, kar-dim/Watermarking-GPU
Watermark_GPU/samples/): images/512.png, w_512.dat, images/720p.png, w_720p.dat.
Expected outputs are produced by the CPU oracle (oracle/wm_oracle.c) and cross-checked
here against the independent numpy restatement (tests/np_restatement.py); the reference
itself publishes no expected values (parity unpinned).
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import np_restatement as NP  # noqa: E402
import oracle_lib as O  # noqa: E402

REF = "/root/reference/Watermark_GPU/samples"


def write_ppm(path, rgb_u8):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb_u8.shape[1], rgb_u8.shape[0]))
        f.write(np.ascontiguousarray(rgb_u8).tobytes())


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def scalars(gray, base_rgb, W, tag, out):
    """harness flow of main.cpp:169-226: embed on RGB base, detect on grey of watermarked"""
    res = {}
    for mask, name in ((O.MASK_NVF, "NVF"), (O.MASK_ME, "ME")):
        st, y, a = O.embed(gray, gray, W, mask=mask)
        assert st == 0
        st, corr = O.detect(y, W, mask=mask)
        st, corr0 = O.detect(gray, W, mask=mask)
        ynp, anp = NP.embed(gray, gray, W, mask=name)
        cnp = NP.detect(ynp, W, mask=name)
        assert abs(a - anp) <= 1e-6 * abs(a), (a, anp)
        assert abs(corr - cnp) <= 1e-6, (corr, cnp)
        assert np.abs(y - ynp).max() <= 1e-3
        res[name] = {"a": a, "corr_gray": corr, "corr_unmarked": corr0, "y_sha256": sha(y)}
        if base_rgb is not None:
            st, yrgb, a2 = O.embed(gray, base_rgb, W, mask=mask)
            st, corr_rgb = O.detect(O.rgb2gray(yrgb), W, mask=mask)
            res[name]["corr_rgb_harness"] = corr_rgb
        # video-style: grey floored to u8, embed grey->grey, output truncated to u8 (main.cpp:355-357)
        g8 = gray.astype(np.uint8)
        st, y8, a8 = O.embed_u8(g8, W, mask=mask)
        st, c8 = O.detect_u8(y8, W, mask=mask)
        res[name]["video_u8"] = {"a": a8, "corr": c8, "y_sha256": sha(y8)}
    st, c, e, m, mx = O.me_mask(gray)
    res["coefficients"] = [float(v) for v in c]
    res["max_abs_e"] = mx
    Rx, rx = O.gram(gray)
    res["cond_Rx"] = float(np.linalg.cond(Rx))
    # the same pair in the reference's own arithmetic for the prediction system (oracle switch ref_arith: half products,
    # 64-lane f32 work-group sums, f32 fold, f32 LU -- me_p3.hpp:8-21,61-82, Watermark.cpp:148-149,203): what the stated
    # bracket "build vs the reference OpenCL path"  is measured against
    st, cr, er, mr, mxr = O.me_mask(gray, ref_arith=True)
    st, y_me, a_me = O.embed(gray, gray, W, mask=O.MASK_ME)
    st, yr, ar = O.embed(gray, gray, W, mask=O.MASK_ME, ref_arith=True)
    st, corr_exact = O.detect(y_me, W, mask=O.MASK_ME)
    st, corr_r = O.detect(y_me, W, mask=O.MASK_ME, ref_arith=True)
    res["reference_arith"] = {
        "coefficients": [float(v) for v in cr], "max_abs_dcoef": float(np.abs(cr - c).max()),
        "a_ME": ar, "rel_da_ME": abs(ar - a_me) / abs(a_me),
        "corr_ME_on_exact_y": corr_r, "abs_dcorr_ME": abs(corr_r - corr_exact),
        "y_rms_vs_exact": float(np.sqrt(((yr.astype(np.float64) - y_me) ** 2).mean())),
        "psnr_exact_dB": float(10 * np.log10(255.0 ** 2 / ((y_me.astype(np.float64) - gray) ** 2).mean())),
        "psnr_reference_arith_dB": float(10 * np.log10(255.0 ** 2 / ((yr.astype(np.float64) - gray) ** 2).mean())),
    }
    out[tag] = res


def big_sample(g8, W, tag, out):
    """scalar record of one large sample: the u8 plane through the video flow (embed_u8 / detect_u8, main.cpp:355-357) and the
    same integers as an f32 image (makeWatermark / detectWatermark on f32 planes), both masks; the prediction system's
    condition; the reference-arithmetic deviations (ME)"""
    gray = g8.astype(np.float32)
    res = {}
    for mask, name in ((O.MASK_NVF, "NVF"), (O.MASK_ME, "ME")):
        st, y, a = O.embed(gray, gray, W, mask=mask)
        assert st == 0
        st, corr = O.detect(y, W, mask=mask)
        st, corr0 = O.detect(gray, W, mask=mask)
        res[name] = {"a": a, "corr_gray": corr, "corr_unmarked": corr0, "y_sha256": sha(y)}
        st, y8, a8 = O.embed_u8(g8, W, mask=mask)
        st, c8 = O.detect_u8(y8, W, mask=mask)
        res[name]["video_u8"] = {"a": a8, "corr": c8, "y_sha256": sha(y8)}
    # independent restatement (numpy), ME only -- the expensive cross-check once per sample
    ynp, anp = NP.embed(gray, gray, W, mask="ME")
    assert abs(res["ME"]["a"] - anp) <= 1e-6 * abs(anp), (res["ME"]["a"], anp)
    assert abs(res["ME"]["corr_gray"] - NP.detect(ynp, W, mask="ME")) <= 1e-6
    st, c, e, m, mx = O.me_mask(gray)
    res["coefficients"] = [float(v) for v in c]
    res["max_abs_e"] = mx
    Rx, rx = O.gram(gray)
    res["cond_Rx"] = float(np.linalg.cond(Rx))
    st, cr, er, mr, mxr = O.me_mask(gray, ref_arith=True)
    st, y_me, a_me = O.embed(gray, gray, W, mask=O.MASK_ME)
    st, yr, ar = O.embed(gray, gray, W, mask=O.MASK_ME, ref_arith=True)
    st, corr_exact = O.detect(y_me, W, mask=O.MASK_ME)
    st, corr_r = O.detect(y_me, W, mask=O.MASK_ME, ref_arith=True)
    res["reference_arith"] = {
        "coefficients": [float(v) for v in cr], "max_abs_dcoef": float(np.abs(cr - c).max()),
        "a_ME": ar, "rel_da_ME": abs(ar - a_me) / abs(a_me),
        "corr_ME_on_exact_y": corr_r, "abs_dcorr_ME": abs(corr_r - corr_exact),
        "y_rms_vs_exact": float(np.sqrt(((yr.astype(np.float64) - y_me) ** 2).mean())),
    }
    out[tag] = res


def main():
    out = {}
    # --- full 512x512 pair (BASELINE config 1) ---
    im = np.asarray(Image.open(f"{REF}/images/512.png").convert("RGB"))
    write_ppm(f"{HERE}/512.ppm", im)  # binary PPM: usable by wm_app / settings.ini directly
    W = np.fromfile(f"{REF}/w_512.dat", np.float32)
    W.tofile(f"{HERE}/w_512.dat")
    W = W.reshape(512, 512)
    rgb = im.transpose(2, 0, 1).astype(np.float32)
    gray = O.rgb2gray(rgb)
    scalars(gray, rgb, W, "512", out)
    out["512"]["files"] = {"rgb": "512.ppm", "w": "w_512.dat", "rows": 512, "cols": 512}

    # --- non-square crop of the 720p pair: 96 rows x 200 cols (odd shape: cols % 64 != 0, % 16 != 0) ---
    im = np.asarray(Image.open(f"{REF}/images/720p.png").convert("RGB"))
    W7 = np.fromfile(f"{REF}/w_720p.dat", np.float32).reshape(720, 1280)
    r0, c0, R, Cc = 300, 500, 96, 200
    crop = np.ascontiguousarray(im[r0:r0 + R, c0:c0 + Cc])
    Wc = np.ascontiguousarray(W7[r0:r0 + R, c0:c0 + Cc])
    write_ppm(f"{HERE}/720p_crop.ppm", crop)
    Wc.tofile(f"{HERE}/w_720p_crop.dat")
    rgbc = crop.transpose(2, 0, 1).astype(np.float32)
    gc = O.rgb2gray(rgbc)
    scalars(gc, rgbc, Wc, "720p_crop", out)
    out["720p_crop"]["files"] = {"rgb": "720p_crop.ppm", "w": "w_720p_crop.dat", "rows": R, "cols": Cc,
                                 "crop_of": "720p.png rows 300:396 cols 500:700"}
    # full per-pixel expectations for the small crop (masks, error sequence, outputs)
    st, c, e, m, mx = O.me_mask(gc)
    nvf = O.nvf_mask(gc)
    st, y_me, a_me = O.embed(gc, gc, Wc, mask=O.MASK_ME)
    st, y_nvf, a_nvf = O.embed(gc, gc, Wc, mask=O.MASK_NVF)
    np.savez_compressed(f"{HERE}/720p_crop_expected.npz", gray=gc, e=e, m_me=m, m_nvf=nvf, y_me=y_me, y_nvf=y_nvf)

    # --- scalar-only records for the full 480p / 720p pairs (inputs stay in the reference) ---
    for tag, img, wf, shp in (("480p", "480p.png", "w_480p.dat", (480, 720)), ("720p", "720p.png", "w_720p.dat", (720, 1280))):
        im = np.asarray(Image.open(f"{REF}/images/{img}").convert("RGB"))
        assert im.shape[:2] == shp
        Wf = np.fromfile(f"{REF}/{wf}", np.float32).reshape(shp)
        rgbf = im.transpose(2, 0, 1).astype(np.float32)
        scalars(O.rgb2gray(rgbf), rgbf, Wf, tag, out)

    # --- the reference's LARGE sample images (samples/images/{1080p,4k,4k_non_divisible}.png): grey planes as the video path
    # forms them (0.299 R + 0.587 G + 0.114 B in f32, truncated to u8: main.cpp:142-144,355), stored losslessly (zlib) so that
    # the HIP path runs on the reference's own hardest inputs (cond(Rx) ~ 1.6e5 at 4K).  The reference ships no W for
    # them (make_w.bat only makes 512/480p/720p): W = this build's generator at the reference's seed (wm_genw <rows> <cols>
    # 28390211 = synth.synth_watermark), generated by the tests, not stored.
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    synth = importlib.import_module("watermarking-gpu_amd.synth")
    for tag, img, shp in (("1080p", "1080p.png", (1080, 1920)), ("4k", "4k.png", (2160, 3840)), ("4k_non_divisible", "4k_non_divisible.png", (2160, 3872))):
        im = np.asarray(Image.open(f"{REF}/images/{img}").convert("RGB"))
        assert im.shape[:2] == shp
        g8 = O.rgb2gray(im.transpose(2, 0, 1).astype(np.float32)).astype(np.uint8)
        np.savez_compressed(f"{HERE}/{tag}_gray_u8.npz", gray=g8)
        big_sample(g8, synth.synth_watermark(*shp), tag, out)
        out[tag]["files"] = {"gray_u8": f"{tag}_gray_u8.npz", "rows": shp[0], "cols": shp[1], "w": "synth.synth_watermark(rows, cols) == wm_genw rows cols 28390211",
                             "source": f"samples/images/{img}, grey as main.cpp:142-144, truncated to u8 as main.cpp:355"}

    out["_meta"] = {
        "psnr": 40.0, "p": 3,
        "generator": "tests/golden/make_golden.py (oracle/wm_oracle.c, exact-sum policy; cross-checked vs tests/np_restatement.py)",
        "note": "expected values are NOT produced by the reference binary (unbuildable here); parity unpinned",
    }
    with open(f"{HERE}/golden.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps({k: {m: v.get(m) for m in ("ME", "NVF")} for k, v in out.items() if k != "_meta"}, indent=1)[:3000])


if __name__ == "__main__":
    main()
