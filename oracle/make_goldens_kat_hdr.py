#!/usr/bin/env python3
"""tests/golden/kat_nancy_church.npz: the reference's documented HDR known-answer case (examples/ex_hdr_images.py:13-17,
"Blur - Quality: 8.696 JOD"): linear-EOTF display photometry (L_peak 4000, contrast 1e6, E_ambient 100) on the
standard_hdr_linear geometry, float32 input in absolute cd/m^2.

example_media/nancy_church.hdr (Radiance RGBE, a data file of the reference) is decoded by the small reader below and
stored as its RGBE bytes (3 MB instead of 9 MB of float32); rgbe_to_float() below is the format's definition and is
repeated in tests/conftest.py.  The blurred test image follows the example's recipe (ex_utils.py:27-41).  Container only.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
from scipy.ndimage import gaussian_filter

import pycvvdp


def read_rgbe(path):
    """Radiance .hdr (-Y H +X W, new-style RLE or flat) -> uint8 [H, W, 4]."""
    d = open(path, "rb").read()
    assert d.startswith(b"#?RADIANCE")
    hend = d.index(b"\n\n") + 2
    lend = d.index(b"\n", hend)
    dims = d[hend:lend].split()
    assert dims[0] == b"-Y" and dims[2] == b"+X"
    H, W = int(dims[1]), int(dims[3])
    p = lend + 1
    out = np.zeros((H, W, 4), dtype=np.uint8)
    for y in range(H):
        if d[p] == 2 and d[p + 1] == 2 and ((d[p + 2] << 8) | d[p + 3]) == W:
            p += 4
            for c in range(4):
                x = 0
                while x < W:
                    n = d[p]
                    p += 1
                    if n > 128:
                        n -= 128
                        out[y, x:x + n, c] = d[p]
                        p += 1
                    else:
                        out[y, x:x + n, c] = np.frombuffer(d[p:p + n], dtype=np.uint8)
                        p += n
                    x += n
        else:
            out[y] = np.frombuffer(d[p:p + 4 * W], dtype=np.uint8).reshape(W, 4)
            p += 4 * W
    return out


def rgbe_to_float(rgbe):
    """Radiance RGBE -> float32 RGB: mantissa * 2^(e - 136), 0 where e == 0."""
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(np.float32)
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def main():
    rgbe = read_rgbe("/root/reference/example_media/nancy_church.hdr")
    img = rgbe_to_float(rgbe)
    L_peak = 4000
    ref = (img / img.max() * L_peak * 4).astype(np.float32)                      # ex_hdr_images.py:29
    test = np.zeros_like(ref)
    for c in range(3):
        test[..., c] = gaussian_filter(ref[..., c], 2, mode="nearest", truncate=2.0)
    disp = pycvvdp.vvdp_display_photo_eotf(L_peak, contrast=1000000, source_colorspace="BT.709", EOTF="linear", E_ambient=100)
    met = pycvvdp.cvvdp(display_name="standard_hdr_linear", display_photometry=disp, heatmap="threshold", device=torch.device("cpu"), quiet=True)
    with torch.no_grad():
        jod, stats = met.predict(test, ref, dim_order="HWC")
    print("reference JOD:", float(jod), "(documented: 8.696)", ref.shape, float(ref.max()))
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "kat_nancy_church.npz"), rgbe=rgbe, jod=np.float32(jod.item()),
                        Q_per_ch=stats["Q_per_ch"], rho_band=stats["rho_band"], documented_jod=np.float32(8.696))


if __name__ == "__main__":
    main()
