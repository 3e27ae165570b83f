"""A strip whose lane 63 holds the image's last four columns (W = 240 k + 248): fused against unfused route, raw heat map and Q_per_ch.
python tools/edge_exact_check.py [W=728] [H=120]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import colorvideovdp_amd as cv
from colorvideovdp_amd import _capi
from test_gpu_parity import _fuse_clip

W, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 728), (int(sys.argv[2]) if len(sys.argv) > 2 else 120)
t, r = _fuse_clip(W, H, 3, W + H)
t, r = torch.as_tensor(t).cuda(), torch.as_tensor(r).cuda()
out = {}
for name, mode in (("fused", 1), ("unfused", 2)):
    m = cv.cvvdp(display_name="standard_fhd", heatmap="raw")
    m.fuse_mode = mode
    j, s = m.predict(t, r, dim_order="BCFHW", frames_per_second=60)
    out[name] = (float(j), s["Q_per_ch"], s["heatmap"].float().numpy())
print(_capi.LIB_PATH)
q1, q2 = out["fused"][1], out["unfused"][1]
print("Q_per_ch max rel diff fused vs unfused: %.3e" % float(np.max(np.abs(q1 - q2) / (np.abs(q2) + 1e-9))))
d = np.abs(out["fused"][2] - out["unfused"][2])[0, 0]          # [F, H, W]
cols = d.max(axis=(0, 1))
print("raw heat map |fused - unfused| max per column, last 16 columns:", np.array2string(cols[W - 16:], precision=5))
print("max over all other columns: %.5f" % float(cols[:W - 16].max()))
