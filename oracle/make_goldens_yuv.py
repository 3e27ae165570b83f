#!/usr/bin/env python3
"""Generate tests/golden/yuv_*.npz by running the REAL reference's .yuv reader and metric (container only).

Writes small synthetic planar Y'CbCr files (names carry the header, video_source_yuv.py:8-62) to a temporary
directory, runs /root/reference's video_source_yuv_file + cvvdp.predict_video_source on CPU (import shims as in
make_goldens.py) and stores the raw samples with the reference's outputs.  Fixtures are data only.

    python oracle/make_goldens_yuv.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import pycvvdp
from pycvvdp.video_source_yuv import video_source_yuv_file, YUVReader

OUT = os.path.join(HERE, "..", "tests", "golden")
CPU = torch.device("cpu")

CASES = [  # name, W, H, frames, fps, bit depth, chroma, colour space, display
    ("yuv420_8b_709_64x48x10_30", 64, 48, 10, 30, 8, "420", "709", "standard_fhd"),
    ("yuv420_10b_2020_80x56x6_60_pq", 80, 56, 6, 60, 10, "420", "2020", "standard_hdr_pq"),
    ("yuv422_8b_709_64x40x5_24", 64, 40, 5, 24, 8, "422", "709", "standard_4k"),
    ("yuv444_10b_709_48x40x4_50", 48, 40, 4, 50, 10, "444", "709", "standard_fhd"),
]


def planes(rng, F, H, W, css, bd, noise):
    """Limited-range planes with a few out-of-range codes (exercises the clips); flat sample array."""
    hc, wc = (H // 2, W // 2) if css == "420" else (H, W // 2) if css == "422" else (H, W)
    s = 2 ** (bd - 8)
    y, x = np.mgrid[0:H, 0:W]
    yc, xc = np.mgrid[0:hc, 0:wc]
    out = []
    for f in range(F):
        Y = 16 + 219 * (0.5 + 0.3 * np.sin(2 * np.pi * (2.5 * x / W + f / 20.0)) * np.cos(2 * np.pi * 1.5 * y / H)) + noise * rng.standard_normal((H, W))
        U = 128 + 112 * 0.6 * np.sin(2 * np.pi * (xc / wc + f / 15.0)) + 0.5 * noise * rng.standard_normal((hc, wc))
        V = 128 + 112 * 0.6 * np.cos(2 * np.pi * (yc / hc - f / 25.0)) + 0.5 * noise * rng.standard_normal((hc, wc))
        Y[0, :4] = [0, 5, 250, 255]          # codes outside the nominal range
        U[0, :2] = [0, 255]
        V[-1, -2:] = [3, 252]
        for p in (Y, U, V):
            out.append(np.clip(np.round(p * s), 0, 2 ** bd - 1).astype(np.uint8 if bd == 8 else np.uint16).ravel())
    return np.concatenate(out)


def main():
    rng = np.random.default_rng(20240607)
    with tempfile.TemporaryDirectory() as tmp:
        for name, W, H, F, fps, bd, css, cs, disp in CASES:
            ref = planes(rng, F, H, W, css, bd, noise=1.0)
            tst = planes(rng, F, H, W, css, bd, noise=6.0)     # same pattern, stronger noise
            fn = {}
            for tag, arr in (("ref", ref), ("test", tst)):
                fn[tag] = os.path.join(tmp, f"{tag}_{W}x{H}_{bd}b_{css}_{cs}_{fps}fps.yuv")
                arr.tofile(fn[tag])
            vs = video_source_yuv_file(fn["test"], fn["ref"], display_photometry=disp)
            assert vs.get_video_size() == [H, W, F] and vs.get_frames_per_second() == fps
            met = pycvvdp.cvvdp(display_name=disp, device=CPU, quiet=True)
            with torch.no_grad():
                jod, stats = met.predict_video_source(vs)
            rd = YUVReader(fn["test"])
            rgb0 = rd.get_frame_rgb_tensor(0, CPU).numpy()
            rgbL = rd.get_frame_rgb_tensor(F - 1, CPU).numpy()
            np.savez_compressed(os.path.join(OUT, name + ".npz"), test=tst, ref=ref, width=W, height=H, frames=F, fps=fps,
                                bit_depth=bd, chroma_ss=css, color_space=cs, display=disp,
                                fname_test=os.path.basename(fn["test"]), fname_ref=os.path.basename(fn["ref"]),
                                rgb_first=rgb0, rgb_last=rgbL, jod=np.float32(jod.item()),
                                Q_per_ch=stats["Q_per_ch"], rho_band=stats["rho_band"])
            print(name, "JOD", float(jod), "Q_per_ch", stats["Q_per_ch"].shape)


if __name__ == "__main__":
    main()
