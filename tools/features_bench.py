"""Features mode against the normal path: extract_features() vs predict_video_source() on the same resident clip.
    python tools/features_bench.py [WxH] [frames]        (run on the GPU box from the repo root)"""
import sys
import time
import torch
sys.path.insert(0, ".")
import bench
import colorvideovdp_amd as cv

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
clip = bench.ResidentClip(F, 0, F, H, W, 60, "u8", torch.device("cuda"))
m = cv.cvvdp(display_name="standard_4k" if W >= 3000 else "standard_fhd")


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


t_norm, _ = timed(lambda: m.predict_video_source(clip))
t_feat, (feats, _) = timed(lambda: m.extract_features(clip))
m.profile(True)
m.extract_features(clip)
torch.cuda.synchronize()
print("extract_features kernels (ms):", {k: round(v[0], 3) for k, v in m.profile_read().items()}, "fused levels", m.fused_levels)
m.profile(False)
print(f"{W}x{H} x {F} frames, cells of {int(-(-m.pix_per_deg // 1))} px: predict {t_norm * 1e3:.2f} ms, extract_features {t_feat * 1e3:.2f} ms "
      f"({t_feat / t_norm:.2f} x), {len(feats)} bands, band 0 features {tuple(feats[0].shape)}")
