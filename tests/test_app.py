"""wm_app (C++ host side over the C ABI through include/Watermark.hpp): the reference's sample-application protocol.
CPU part: INI semantics and that the app builds and fails loudly without a GPU.  GPU part: image mode against the
golden harness correlations, video mode (raw yuv420p / y4m) against the oracle's video-frame contract."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frame, synth_watermark

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "watermarking-gpu_amd")
APP = os.path.join(PKG, "wm_app")


@pytest.fixture(scope="module")
def app():
    if not os.path.exists(APP):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(PKG, "csrc")])
        subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc", "app")])
    return APP


def write_ini(path, **kv):
    d = dict(image="", watermark="", video="", device=0, save="false", fps="true", p=3, psnr=40.0, loops=2, interval=1, out="",
             detection="true")
    d.update(kv)
    path.write_text(f"""[paths]
image = {d['image']}
watermark = {d['watermark']}
{'video = ' + d['video'] if d['video'] else '; video = none'}

[options]
OpenCL_Device = {d['device']}   ; keys are case-insensitive, inline comments allowed
save_watermarked_files_to_disk = {d['save']}
execution_time_in_fps = {d['fps']}

[parameters]
p = {d['p']}
psnr = {d['psnr']}
loops_for_test = {d['loops']}

[parameters_video]
watermark_interval = {d['interval']}
encode_watermark_file_path = {d['out']}
watermark_detection = {d['detection']}
""")


def test_app_rejects_bad_settings_without_gpu(app, tmp_path):
    r = subprocess.run([app, str(tmp_path / "missing.ini")], capture_output=True, text=True)
    assert r.returncode != 0 and "Could not load settings.ini file" in r.stdout  # main.cpp:66
    ini = tmp_path / "s.ini"
    write_ini(ini, p=5)
    r = subprocess.run([app, str(ini)], capture_output=True, text=True)
    assert r.returncode != 0 and "For now, only p=3 is allowed" in r.stdout  # main.cpp:89
    write_ini(ini, psnr=-3)
    r = subprocess.run([app, str(ini)], capture_output=True, text=True)
    assert r.returncode != 0 and "PSNR must be a positive number" in r.stdout  # main.cpp:96


def write_ppm(path, rgb_u8):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb_u8.shape[1], rgb_u8.shape[0]))
        f.write(np.ascontiguousarray(rgb_u8).tobytes())


def write_png(path, rgb_u8):
    """minimal PNG writer (filter 0 and a Paeth row) to exercise the app's zlib PNG reader"""
    import struct
    import zlib
    H, W, _ = rgb_u8.shape
    raw = bytearray()
    prev = np.zeros((W, 3), np.int32)
    for y in range(H):
        row = rgb_u8[y].astype(np.int32)
        if y % 2 == 0:
            raw += b"\x00" + rgb_u8[y].tobytes()
        else:  # filter 2 (Up)
            raw += b"\x02" + ((row - prev) % 256).astype(np.uint8).tobytes()
        prev = row

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw)))
                + chunk(b"IEND", b""))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["ppm", "png"])
def test_app_image_mode_golden(app, tmp_path, golden, fmt):
    """testForImage protocol on the reference's 512 sample pair: the two printed correlations are the golden
    harness values (embed on the RGB base, detect on the grey of the watermarked image, main.cpp:169-226)"""
    from conftest import GOLDEN
    info = golden["512"]["files"]
    rgb = np.fromfile(os.path.join(GOLDEN, info["rgb"]), np.uint8)[-512 * 512 * 3:].reshape(512, 512, 3)
    img = tmp_path / f"512.{fmt}"
    (write_ppm if fmt == "ppm" else write_png)(img, rgb)
    ini = tmp_path / "settings.ini"
    write_ini(ini, image=str(img), watermark=os.path.join(GOLDEN, info["w"]), save="true", loops=3)
    r = subprocess.run([app, str(ini)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    nvf = float(re.search(r"Correlation \[NVF\]: ([-0-9.]+)", r.stdout).group(1))
    me = float(re.search(r"Correlation \[ME\]: ([-0-9.]+)", r.stdout).group(1))
    assert nvf == pytest.approx(golden["512"]["NVF"]["corr_rgb_harness"], abs=1e-5)
    assert me == pytest.approx(golden["512"]["ME"]["corr_rgb_harness"], abs=1e-5)
    strengths = [float(v) for v in re.findall(r"Watermark strength \(parameter a\): ([-0-9.e+]+)", r.stdout)]
    assert strengths[0] == pytest.approx(golden["512"]["NVF"]["a"], rel=1e-4)
    assert strengths[1] == pytest.approx(golden["512"]["ME"]["a"], rel=1e-4)
    assert "FPS:" in r.stdout and "Calculation of ME mask with 512 rows and 512 columns" in r.stdout
    # saved files: <name>_W_NVF / <name>_W_ME (Utilities.cpp:7-11), u8 by truncation
    out = tmp_path / "512_W_ME.ppm"
    assert out.exists() and (tmp_path / "512_W_NVF.ppm").exists()
    planar = np.ascontiguousarray(rgb.transpose(2, 0, 1)).astype(np.float32)
    st, y, a = O.embed(O.rgb2gray(planar), planar, np.fromfile(os.path.join(GOLDEN, info["w"]), np.float32).reshape(512, 512))
    got = np.frombuffer(out.read_bytes()[-512 * 512 * 3:], np.uint8).reshape(512, 512, 3)
    diff = np.abs(got.astype(int) - y.transpose(1, 2, 0).astype(np.uint8).astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(96, 160), (98, 158), (130, 522)])
def test_app_video_mode_y4m(app, tmp_path, size):
    """raw yuv420p frames: every `watermark_interval`-th Y plane embedded (ME) with U/V passed through, then detected.
    Widths that are not multiples of 4 take the unaligned path (host staging re-pitches to a multiple of 4)."""
    R, C = size
    NF, interval = 7, 3
    W = synth_watermark(R, C)
    wfile = tmp_path / "w.dat"
    W.tofile(wfile)
    src = tmp_path / "in.y4m"
    ys, uvs = [], []
    with open(src, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420\n" % (C, R))
        for k in range(NF):
            y = synth_frame(R, C, frame=k, dtype=np.uint8)
            uv = (np.arange(2 * (R // 2) * (C // 2)) * (k + 1) % 251).astype(np.uint8)
            ys.append(y); uvs.append(uv)
            f.write(b"FRAME\n" + y.tobytes() + uv.tobytes())
    dst = tmp_path / "out.y4m"
    ini = tmp_path / "settings.ini"
    write_ini(ini, video=str(src), watermark=str(wfile), interval=interval, out=str(dst))
    r = subprocess.run([app, str(ini)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    data = dst.read_bytes()
    hdr_end = data.index(b"\n") + 1
    fsz = 6 + R * C + 2 * (R // 2) * (C // 2)
    assert len(data) == hdr_end + NF * fsz
    for k in range(NF):
        fr = data[hdr_end + k * fsz: hdr_end + (k + 1) * fsz]
        assert fr[:6] == b"FRAME\n"
        y = np.frombuffer(fr[6:6 + R * C], np.uint8).reshape(R, C)
        uv = np.frombuffer(fr[6 + R * C:], np.uint8)
        np.testing.assert_array_equal(uv, uvs[k])  # chroma untouched (main.cpp:359-386)
        if k % interval == 0:
            st, yo, a = O.embed_u8(ys[k], W)
            d = np.abs(y.astype(int) - yo.astype(int))
            assert d.max() <= 1 and (d != 0).mean() <= 1e-3
        else:
            np.testing.assert_array_equal(y, ys[k])
    # detection pass over the watermarked stream
    write_ini(ini, video=str(dst), watermark=str(wfile), interval=interval, out="", detection="true")
    r = subprocess.run([app, str(ini)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    found = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"Correlation for frame: (\d+): ([-0-9.e]+)", r.stdout)}
    assert sorted(found) == [0, 3, 6]
    for k, c in found.items():
        y = np.frombuffer(data[hdr_end + k * fsz + 6: hdr_end + k * fsz + 6 + R * C], np.uint8).reshape(R, C)
        assert c == pytest.approx(O.detect_u8(y, W)[1], abs=2e-5)
        assert c > 0.3


def test_genw_tool(app, tmp_path):
    """wm_genw: the CLI and file format of CommonRandomMatrix (CommonRandomMatrix/main.cpp:16-68); counter-based, so the
    output does not depend on the thread count and equals the Python generator used by the tests"""
    tool = os.path.join(PKG, "wm_genw")
    out = tmp_path / "w.dat"
    r = subprocess.run([tool, "96", "200", "28390211", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "96 x 200 = 19200 N(0,1) values (seed 28390211)" in r.stdout
    w = np.fromfile(out, np.float32)
    assert w.size == 96 * 200
    ref = synth_watermark(96, 200)
    np.testing.assert_allclose(w.reshape(96, 200), ref, rtol=0, atol=1e-6)
    assert abs(float(w.mean())) < 0.05 and abs(float(w.std()) - 1.0) < 0.05
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out1 = tmp_path / "w1.dat"
    subprocess.check_call([tool, "96", "200", "28390211", str(out1)], env=env, stdout=subprocess.DEVNULL)
    assert out.read_bytes() == out1.read_bytes()   # thread-count independent
    r = subprocess.run([tool, "96", "200"], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage:" in r.stderr
    for bad in (["0", "5", "1"], ["5", "32768", "1"], ["12x", "5", "1"], ["5", "5", "-3"]):
        r = subprocess.run([tool, *bad, str(out)], capture_output=True, text=True)
        assert r.returncode != 0 and "Usage:" in r.stderr, bad


@pytest.mark.gpu
def test_cpp_class_surface_selftest(app, tmp_path):
    """include/Watermark.hpp exercised from C++ (csrc/app/wm_selftest.cpp): constructor errors with the reference's
    messages (Watermark.cpp:24-25,65-66,70-71), deep copies sharing W (:30-51), reinitialize (:78-85), the embed()/
    detect() aliases, RGB bases (main.cpp:169-190), the unsolvable-system rule (:164-165,246-247)"""
    exe = os.path.join(PKG, "wm_selftest")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc", "app")])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 check(s) failed" in r.stdout and r.stdout.count("ok  ") >= 14


@pytest.mark.gpu
def test_genw_file_drives_the_engine(app, tmp_path):
    """SURVEY 8(f)3 end to end on the GPU box: a W file written by wm_genw (CommonRandomMatrix's CLI and format) is loaded
    by the engine through loadRandomMatrix's path (Watermark.cpp:62-75) and embeds / detects exactly like the same
    matrix handed over as an array, and like the oracle with that matrix"""
    import importlib
    import torch
    import oracle_lib as O
    from synth import synth_frame
    wm = importlib.import_module("watermarking-gpu_amd")
    R, Cc = 180, 516
    wpath = tmp_path / "w_gen.dat"
    subprocess.check_call([os.path.join(PKG, "wm_genw"), str(R), str(Cc), "424242", str(wpath)], stdout=subprocess.DEVNULL)
    W = np.fromfile(wpath, np.float32).reshape(R, Cc)
    assert abs(float(W.mean())) < 0.05 and abs(float(W.std()) - 1.0) < 0.05
    x = synth_frame(R, Cc, frame=1)
    xd = torch.from_numpy(x).cuda()
    e_file = wm.Watermark(R, Cc, str(wpath), 3, 40.0)
    e_arr = wm.Watermark(R, Cc, W, 3, 40.0)
    for mk, omk in ((wm.MASK_TYPE.ME, O.MASK_ME), (wm.MASK_TYPE.NVF, O.MASK_NVF)):
        y1, a1 = e_file.makeWatermark(xd, xd, mk)
        y2, a2 = e_arr.makeWatermark(xd, xd, mk)
        assert a1 == a2 and torch.equal(y1, y2)
        so, yo, ao = O.embed(x, x, W, mask=omk)
        assert a1 == pytest.approx(ao, rel=1e-4)
        c = e_file.detectWatermark(y1, mk)
        assert c == pytest.approx(O.detect(y1.cpu().numpy(), W, mask=omk)[1], abs=1e-5) and c > 0.3
    e_file.close(); e_arr.close()


@pytest.mark.gpu
def test_wm_stream_sharding_and_resequencer(app, tmp_path):
    """csrc/app/wm_stream.cpp: the C++ multi-GPU stream host (one context + thread per device, frame i -> device i mod G,
    in-order re-sequencer, RCCL score gather).  On the one-GPU box the device list repeats device 0: the sharding, the
    batching of each device's shard, the buffer pools and the re-sequencer are those of a multi-GPU run; outputs and scores
    must equal the single-device run byte for byte.  With one device listed once the scores travel through RCCL
    (ncclCommInitAll + ncclAllGather), which must not change them either."""
    exe = os.path.join(PKG, "wm_stream")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc", "app")])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(devices, gather, tag, frames=37, extra=()):
        out, sc = tmp_path / f"y_{tag}.raw", tmp_path / f"s_{tag}.txt"
        r = subprocess.run([exe, "--devices", devices, "--rows", "270", "--cols", "512", "--frames", str(frames), "--batch", "4", "--slots", "2",
                            "--gather", gather, "--out", str(out), "--scores", str(sc), *extra], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        info = json.loads(r.stdout.strip().splitlines()[-1])
        return out.read_bytes(), sc.read_text(), info
    y1, s1, i1 = run("0", "host", "g1")
    assert len(y1) == 37 * 270 * 512 and len(s1.splitlines()) == 37 and i1["mean_corr"] > 0.3
    for devices, tag in (("0,0", "g2"), ("0,0,0", "g3")):
        y, s, info = run(devices, "host", tag)
        assert info["gather"] == "host" and info["checksum"] == i1["checksum"]
        assert y == y1 and s == s1, f"devices={devices}: output or scores differ from the single-device run"
    yr, sr, ir = run("0", "rccl", "rccl")
    assert ir["gather"] == "rccl" and yr == y1 and sr == s1
    # watermark_interval (main.cpp:346,395): frames outside the interval pass through unmarked and unscored
    yi, si, ii = run("0,0", "host", "int3", extra=("--interval", "3"))
    rows = [l.split() for l in si.splitlines()]
    assert all((int(r[0]) % 3 == 0) == (r[3] == "1") for r in rows)
    n = 270 * 512
    for f in range(37):
        same = yi[f * n:(f + 1) * n] == y1[f * n:(f + 1) * n]
        assert same == (f % 3 == 0), f
    # --ring N: the source frames live in a pinned ring handed to the engine without a per-frame host copy; frame i has the
    # content of frame i mod N (N rounded down to devices x batch): the first N frames equal the plain run's, then they repeat;
    # --pin 1 on the one device must not change anything either (the placement of the worker thread)
    yg, sg, ig = run("0,0", "host", "ring", extra=("--ring", "16", "--pin", "1"))
    assert ig["ring_batches_per_device"] == 2
    for f in range(37):
        assert yg[f * n:(f + 1) * n] == y1[(f % 16) * n:(f % 16 + 1) * n], f
    assert [l.split()[2] for l in sg.splitlines()][:16] == [l.split()[2] for l in s1.splitlines()][:16]


@pytest.mark.gpu
def test_watermark_generated_on_the_device(app, tmp_path):
    """wm_create_generated: W filled on the GPU by the generator of wm_genw (closes SURVEY 8(f)3's "optional on-device
    generation"): equal to the file wm_genw writes and to synth_watermark up to the last ulp of the device's f64 log / cos,
    and an engine built on it embeds and detects like one built from the file"""
    import importlib
    import torch
    import oracle_lib as O
    from synth import synth_frame
    wm = importlib.import_module("watermarking-gpu_amd")
    R, Cc, seed = 200, 516, 28390211
    eng = wm.Watermark.generated(R, Cc, seed, 3, 40.0)
    Wd = eng.watermark()
    ref = synth_watermark(R, Cc, seed=seed)
    assert np.abs(Wd - ref).max() <= 1e-6 and (Wd != ref).mean() < 1e-3
    wpath = tmp_path / "w.dat"
    subprocess.check_call([os.path.join(PKG, "wm_genw"), str(R), str(Cc), str(seed), str(wpath)], stdout=subprocess.DEVNULL)
    Wf = np.fromfile(wpath, np.float32).reshape(R, Cc)
    assert np.abs(Wd - Wf).max() <= 1e-6
    x = synth_frame(R, Cc, frame=2)
    xd = torch.from_numpy(x).cuda()
    y, a = eng.makeWatermark(xd, xd, wm.MASK_TYPE.ME)
    so, yo, ao = O.embed(x, x, Wd)
    assert a == pytest.approx(ao, rel=1e-4)
    assert eng.detectWatermark(y, wm.MASK_TYPE.ME) == pytest.approx(O.detect(y.cpu().numpy(), Wd)[1], abs=1e-5)
    other = wm.Watermark.generated(R, Cc, seed + 1, 3, 40.0)
    assert abs(other.detectWatermark(y, wm.MASK_TYPE.ME)) < 0.05   # another key does not detect it
    eng.close(); other.close()

@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype,mask", [((1080, 1920), "f32", "ME"), ((720, 1280), "u8", "NVF"), ((300, 250), "f32", "ME")])
def test_single_call_timer_runs_on_both_paths(app, tmp_path, shape, dtype, mask):
    """csrc/app/wm_single.cpp (the bench line's `single_call` leg): Watermark::makeWatermark + detectWatermark per frame from
    C++, one synchronous call each.  Fusable shapes take the fused single-launch kernels (no fall-back to the sweeps),
    the others the sweeps; with WM_FUSED=0 the same calls give the same strength and correlation to the rounding of the
    partial sums' grouping."""
    exe = os.path.join(PKG, "wm_single")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc", "app")])
    R, Cc = shape

    def run(env_extra):
        r = subprocess.run([exe, str(R), str(Cc), "20", dtype, mask, str(tmp_path)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stdout + r.stderr
        return json.loads(r.stdout.strip().splitlines()[-1])

    a = run({})
    b = run({"WM_FUSED": "0"})
    fusable = Cc % 4 == 0 and Cc >= 256
    assert a["fused"] == (1 if fusable else 0) and b["fused"] == 0 and a["fallbacks"] == 0
    assert a["embed_us"] > 0 and a["detect_us"] > 0 and 0.0 < a["corr"] <= 1.0 and a["a"] > 0
    assert abs(a["a"] - b["a"]) <= 1e-4 * abs(b["a"]) and abs(a["corr"] - b["corr"]) <= 1e-5
