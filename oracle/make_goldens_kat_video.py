#!/usr/bin/env python3
"""tests/golden/kat_tree_blur_over_time.npz: the reference's video example (examples/ex_blur_over_time.py) run through the
real reference: example_media/tree.jpg (a data file of the reference) repeated for 240 frames at 30 fps, the test clip
blurred with a sigma that ramps 0.01 -> 2 -> 0.01 (scipy gaussian_filter, ex_utils.py:27-41), display standard_4k.

The example's docstring says 8.829 JOD; the current reference code (v0.5.6 + blur shim) gives 8.093 on these samples --
the docstring predates the current calibration (the image examples' docstrings do match, see make_goldens_kat*.py).
The fixture pins the code's answer: the decoded image (PIL), JOD, Q_per_ch.  Container only.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
from PIL import Image
from scipy.ndimage import gaussian_filter

import pycvvdp


def main():
    img = np.asarray(Image.open("/root/reference/example_media/tree.jpg").convert("RGB"))
    N, fps = 240, 30
    ref = np.repeat(img[..., np.newaxis], N, axis=3)
    sig = np.concatenate((np.linspace(0.01, 2, N // 2), np.linspace(2, 0.01, N // 2)))
    test = np.zeros_like(ref)
    for f, s in enumerate(sig):
        for c in range(3):
            test[..., c, f] = gaussian_filter(ref[..., c, f], s, mode="nearest", truncate=2.0)
    met = pycvvdp.cvvdp(display_name="standard_4k", heatmap=None, device=torch.device("cpu"), quiet=True)
    with torch.no_grad():
        jod, stats = met.predict(test, ref, dim_order="HWCF", frames_per_second=fps)
    print("reference JOD:", float(jod), "(docstring: 8.829)")
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "kat_tree_blur_over_time.npz"), img=img, frames=N, fps=fps,
                        jod=np.float32(jod.item()), Q_per_ch=stats["Q_per_ch"], rho_band=stats["rho_band"], docstring_jod=np.float32(8.829))


if __name__ == "__main__":
    main()
