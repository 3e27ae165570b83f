"""Where does the benched 4K x 64 f32 clip differ most from the real reference's Q_per_ch (tests/golden/bench_4k64_f32.npz)?"""
import sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
import colorvideovdp_amd as cv
from conftest import load_golden
name = sys.argv[1] if len(sys.argv) > 1 else "bench_4k64_f32"
g = load_golden(name)
W, H, F, fps, disp, dtype = int(g["width"]), int(g["height"]), int(g["frames"]), float(g["fps"]), str(g["display"]), str(g["dtype"])
clip = bench.ResidentClip(F, 0, F, H, W, fps, dtype, torch.device("cuda"), gen="cpu")
assert (clip.checksum_test, clip.checksum_ref) == (int(g["checksum_test"]), int(g["checksum_ref"]))
j, s = cv.cvvdp(display_name=disp).predict_video_source(clip)
q, qr = s["Q_per_ch"], g["Q_per_ch"]
rel = np.abs(q - qr) / (np.abs(qr) + 1e-12)
print("JOD", float(j), float(g["jod"]), " max rel", rel.max(), " mean rel", rel.mean())
print("per band max rel err:", np.round(rel.max(axis=(0, 1, 2)), 7))
print("per channel max rel err:", np.round(rel.max(axis=(0, 2, 3)), 7))
print("per band mean |Q|:", np.round(np.abs(qr).mean(axis=(0, 1, 2)), 5))
idx = np.dstack(np.unravel_index(np.argsort(rel.ravel())[::-1][:8], rel.shape))[0]
for i in idx:
    i = tuple(i)
    print("  [b,c,f,band]", i, "hip", q[i], "ref", qr[i], "rel", rel[i])
