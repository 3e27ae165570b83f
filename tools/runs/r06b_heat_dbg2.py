"""Which intermediate of the two heat-map finishing kernels differs?  Libraries variants/heat_dbg<n>.so put the fp32 bits of stage n
(heatmap.hip HEAT_DEBUG_STAGE: 1 tone-mapped luminance, 2 map value d, 3 colour-map fraction, 4 reconstructed q, 5 log luminance, 6 tone-curve fraction, 7 b - node; build them with tools/build_variant.sh heat_dbg<n> heatmap.hip -DHEAT_DEBUG_STAGE=<n>) into planes 0 / 1.  Run on the GPU box:
    CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/variants/heat_dbg1.so python tools/runs/r06b_heat_dbg2.py"""
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
out = {}
for layout in (0, 1):
    m = cv.cvvdp(display_name="standard_4k", heatmap="threshold")
    m.fuse_mode, m.band_layout = 2, layout
    _, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=30)
    h = st["heatmap"].numpy().view(np.uint16).astype(np.uint32)
    out[layout] = (h[0, 0] | (h[0, 1] << 16)).view(np.float32)
a, b = out[0], out[1]
bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
print(os.environ.get("CVVDP_LIB", "in-tree"), "differing", bad.shape[0], "of", a.size)
for idx in bad[:6]:
    i = tuple(idx)
    print("  ", tuple(int(v) for v in i), repr(float(a[i])), repr(float(b[i])))
for i in ((0, 0, 0), (0, 0, 1), (0, 0, 2), (0, 40, 700)):
    print("  at", i, repr(float(a[i])), repr(float(b[i])))
