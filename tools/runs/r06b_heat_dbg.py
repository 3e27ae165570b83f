"""Where do the two heat-map finishing kernels differ?  (round 6b debugging aid; run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import colorvideovdp_amd as cv
rng = np.random.default_rng(77)
H, W, F = 71, 1028, 3
yy, xx = np.mgrid[0:H, 0:W]
base = 8.0 + 235.0 * (0.5 + 0.5 * np.sin(xx / 37.0 + yy / 23.0)) * (xx / W)
r = np.clip(base[None, None, :, :] + rng.normal(0, 2.0, (F, 3, H, W)), 0, 255).round().astype(np.uint8)
t = np.clip(r.astype(np.float32) + rng.normal(0, 6.0, r.shape) * (xx > W // 3), 0, 255).round().astype(np.uint8)
t, r = (torch.from_numpy(np.ascontiguousarray(a.transpose(1, 0, 2, 3))[None]) for a in (t, r))
for mode in ("threshold", "supra-threshold"):
    out = {}
    for layout in (0, 1):
        m = cv.cvvdp(display_name="standard_4k", heatmap=mode)
        m.fuse_mode, m.band_layout = 2, layout
        _, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=30)
        out[layout] = st["heatmap"].numpy().copy()
    a, b = out[0].view(np.uint16).astype(np.int32), out[1].view(np.uint16).astype(np.int32)
    bad = np.argwhere(a != b)
    print(mode, "differing", bad.shape[0])
    for idx in bad[:40]:
        i = tuple(idx)
        print("  ", i, out[0][i], out[1][i], "all channels new", out[0][0, :, i[2], i[3], i[4]], "old", out[1][0, :, i[2], i[3], i[4]])
