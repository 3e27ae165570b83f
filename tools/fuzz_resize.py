"""Randomised sweep of the .yuv full_screen_resize path against the CPU oracle (development aid, run on the GPU box):
    python tools/fuzz_resize.py [n_cases] [seed]
Random source / target sizes, modes, chroma formats, bit depths; test and reference of different sizes now and then."""
import os
import sys
import tempfile
import numpy as np
sys.path.insert(0, ".")
import colorvideovdp_amd as cv
from oracle import yuv_oracle as yo
from oracle.cvvdp_oracle import Oracle

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 20), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rng = np.random.default_rng(seed)


def planes(F, H, W, css, bd, noise):
    hc, wc = (H // 2, W // 2) if css == "420" else (H, W // 2) if css == "422" else (H, W)
    s = 2 ** (bd - 8)
    y, x = np.mgrid[0:H, 0:W]
    yc, xc = np.mgrid[0:hc, 0:wc]
    out = []
    for f in range(F):
        Y = 16 + 219 * (0.5 + 0.3 * np.sin(2 * np.pi * (2.5 * x / W + f / 20.0)) * np.cos(2 * np.pi * 1.5 * y / H)) + noise * rng.standard_normal((H, W))
        U = 128 + 112 * 0.6 * np.sin(2 * np.pi * (xc / wc + f / 15.0)) + 0.5 * noise * rng.standard_normal((hc, wc))
        V = 128 + 112 * 0.6 * np.cos(2 * np.pi * (yc / hc - f / 25.0)) + 0.5 * noise * rng.standard_normal((hc, wc))
        for p in (Y, U, V):
            out.append(np.clip(np.round(p * s), 0, 2 ** bd - 1).astype(np.uint8 if bd == 8 else np.uint16).ravel())
    return np.concatenate(out)


bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for k in range(n):
        css = str(rng.choice(["420", "422", "444"]))
        bd = int(rng.choice([8, 10]))
        mode = str(rng.choice(["nearest", "bilinear", "bicubic", "area"]))
        F = int(rng.choice([2, 3, 5]))
        fps = int(rng.choice([24, 30, 60]))
        disp = str(rng.choice(["standard_fhd", "standard_4k", "standard_hdr_pq"]))
        ws, hs = 2 * int(rng.integers(12, 90)), 2 * int(rng.integers(10, 60))
        W, H = int(rng.integers(20, 260)), int(rng.integers(18, 160))
        wr, hr = (ws, hs) if rng.random() < 0.6 else ((W - W % 2, H - H % 2) if rng.random() < 0.5 else (2 * int(rng.integers(12, 90)), 2 * int(rng.integers(10, 60))))
        if rng.random() < 0.2:
            W, H = wr, hr                                     # the reference already has the target size
        files, clips = {}, {}
        for tag, w, h, noise in (("test", ws, hs, 6.0), ("ref", wr, hr, 1.0)):
            arr = planes(F, h, w, css, bd, noise)
            files[tag] = os.path.join(tmp, f"{tag}{k}_{w}x{h}_{bd}b_{css}_709_{fps}fps.yuv")
            arr.tofile(files[tag])
            clips[tag] = yo.clip_to_rgb_resized(arr, yo.decode_video_props(files[tag]), F, H, W, mode)
        oj, os_ = Oracle(display_name=disp).predict(clips["test"], clips["ref"], dim_order="BCFHW", frames_per_second=fps)
        vs = cv.video_source_yuv_file(files["test"], files["ref"], display_photometry=disp, full_screen_resize=mode, resize_resolution=(W, H))
        j, s = cv.cvvdp(display_name=disp, block_frames=int(rng.choice([1, 2, 64]))).predict_video_source(vs)
        dq = np.abs(s["Q_per_ch"] - os_["Q_per_ch"]) / (np.abs(os_["Q_per_ch"]) * 2e-4 + 2e-6)
        dj = abs(float(j) - float(oj))
        ok = dj <= 1e-3 and dq.max() <= 1.0
        print(("ok  " if ok else "BAD ") + f"{k:3d} {ws}x{hs} / {wr}x{hr} -> {W}x{H} {mode} {css} {bd}b x{F} {disp}: dJOD {dj:.2e} Q err/tol {dq.max():.2f}", flush=True)
        bad += not ok
        for f in files.values():
            os.remove(f)
print("bad:", bad)
