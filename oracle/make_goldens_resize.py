#!/usr/bin/env python3
"""Generate tests/golden/resize_*.npz: the REAL reference's full_screen_resize of .yuv sources (container only).

Small synthetic planar Y'CbCr files (made like those of make_goldens_yuv.py) go through /root/reference's
video_source_yuv_file(full_screen_resize=mode, resize_resolution=(W, H)) + cvvdp.predict_video_source on the CPU.  Stored: the raw
samples of both files, the reference's resized R'G'B' of the first frame of each side (captured where _get_frame hands it to the
display model), JOD and Q_per_ch.  Fixtures are data only.

    python oracle/make_goldens_resize.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)

import numpy as np
import torch

import pycvvdp
from pycvvdp.video_source_yuv import video_source_yuv_file
from make_goldens_yuv import planes

OUT = os.path.join(HERE, "..", "tests", "golden")
CPU = torch.device("cpu")

CASES = [  # name, (Wt, Ht) test, (Wr, Hr) reference, target (W, H), mode, frames, fps, bit depth, chroma, colour space, display
    ("resize_bilinear_up_420_8b", (48, 32), (48, 32), (96, 64), "bilinear", 5, 30, 8, "420", "709", "standard_fhd"),
    ("resize_bicubic_test_only_444_10b_pq", (40, 24), (80, 48), (80, 48), "bicubic", 4, 60, 10, "444", "2020", "standard_hdr_pq"),
    ("resize_nearest_down_422_8b", (64, 48), (64, 48), (40, 30), "nearest", 4, 24, 8, "422", "709", "standard_4k"),
    ("resize_area_down_420_10b", (64, 48), (64, 48), (40, 28), "area", 4, 50, 10, "420", "709", "standard_fhd"),
    ("resize_bicubic_down_up_420_8b", (72, 40), (36, 20), (54, 30), "bicubic", 3, 30, 8, "420", "709", "standard_fhd"),
]


def main():
    rng = np.random.default_rng(20240929)
    with tempfile.TemporaryDirectory() as tmp:
        for name, (wt, ht), (wr, hr), (W, H), mode, F, fps, bd, css, cs, disp in CASES:
            ref = planes(rng, F, hr, wr, css, bd, noise=1.0)
            tst = planes(rng, F, ht, wt, css, bd, noise=6.0)
            fn = {}
            for tag, arr, w, h in (("ref", ref, wr, hr), ("test", tst, wt, ht)):
                fn[tag] = os.path.join(tmp, f"{tag}_{w}x{h}_{bd}b_{css}_{cs}_{fps}fps.yuv")
                arr.tofile(fn[tag])
            vs = video_source_yuv_file(fn["test"], fn["ref"], display_photometry=disp, full_screen_resize=mode, resize_resolution=(W, H))
            assert vs.get_video_size() == [H, W, F]
            seen = []
            inner = vs.apply_dm_and_color_transform

            def spy(frame, colorspace, inner=inner, seen=seen):
                seen.append(frame.detach().clone())
                return inner(frame, colorspace)
            vs.apply_dm_and_color_transform = spy
            vs.get_test_frame(0, CPU, "DKLd65")
            vs.get_reference_frame(0, CPU, "DKLd65")
            vs.apply_dm_and_color_transform = inner
            rgb_t, rgb_r = (x.numpy()[0, :, 0] for x in seen)              # [3, H, W]
            assert rgb_t.shape == (3, H, W) and rgb_r.shape == (3, H, W)
            met = pycvvdp.cvvdp(display_name=disp, device=CPU, quiet=True)
            with torch.no_grad():
                jod, stats = met.predict_video_source(vs)
            np.savez_compressed(os.path.join(OUT, name + ".npz"), test=tst, ref=ref, width=W, height=H, frames=F, fps=fps, mode=mode,
                                bit_depth=bd, chroma_ss=css, color_space=cs, display=disp,
                                fname_test=os.path.basename(fn["test"]), fname_ref=os.path.basename(fn["ref"]),
                                rgb_test_first=rgb_t, rgb_ref_first=rgb_r, jod=np.float32(jod.item()),
                                Q_per_ch=stats["Q_per_ch"], rho_band=stats["rho_band"])
            print(name, "JOD", float(jod), "Q_per_ch", stats["Q_per_ch"].shape)


if __name__ == "__main__":
    main()
