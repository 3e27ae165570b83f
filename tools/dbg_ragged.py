"""Debug aid: per-level comparison of pyramid levels and per-pixel D against the oracle for one image shape."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import colorvideovdp_amd as cv
from colorvideovdp_amd import _capi
from oracle import cvvdp_oracle as orc

H, W = int(sys.argv[1]), int(sys.argv[2])
F = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
heat = sys.argv[5] if len(sys.argv) > 5 else None
disp = sys.argv[6] if len(sys.argv) > 6 else "standard_fhd"
rng = np.random.default_rng(1)
y, x = np.mgrid[0:H, 0:W]
ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (4.3 * x / W + f / 9.0) + c) * np.cos(2 * np.pi * 2.7 * y / H) for c in range(3)]) for f in range(F)], axis=1)[None]
test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
ref = ref.astype(np.float32)
m = cv.cvvdp(display_name=disp, heatmap=heat)
m.debug_dump = True
jod, st = m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
o = orc.Oracle(display_name=disp, keep=True, heatmap=heat)
oj, ost = o.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
nch = 3 if F == 1 else 4
print("JOD", float(jod), float(oj))
d = o.dbg
for l in range(len(d["gpyr"])):
    h, w = d["gpyr"][l].shape[-2:]
    buf = m.debug_buffer(_capi.BUF_GPYR, l).cpu().numpy().reshape(2 * nch, -1, h, w)[:, F - 1]
    want = d["gpyr"][l][0, :, 0].numpy()
    e = np.abs(buf - want)
    print(f"level {l} {h}x{w}: gpyr max err {e.max():.3e} at col {np.unravel_index(e.argmax(), e.shape)}")
    if l < len(d["D"]):
        dd = m.debug_buffer(_capi.BUF_DDUMP, l).cpu().numpy().reshape(4, -1, h, w)[:nch, F - 1]
        wd = d["D"][l][0, :, 0].numpy()
        e = np.abs(dd - wd) / (np.abs(wd) + 1e-3)
        print(f"          D rel err max {e.max():.3e} at {np.unravel_index(e.argmax(), e.shape)}; per-column max:", np.round(e.max(axis=(0, 1)), 3)[-12:], "rows:", np.round(e.max(axis=(0, 2)), 3)[:6], np.round(e.max(axis=(0, 2)), 3)[-6:])
