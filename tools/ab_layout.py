#!/usr/bin/env python3
"""A/B of the fused band kernel's wave layouts on the bench clip (GPU box): band_layout 0 (front / back waves, band4s.hip) against
1 (one wave per channel, k_band4f<4, 0>): HIP-event time of the level-0 launch pair, the other kernel families, the step, and whether
Q_per_ch is bit-identical.   python tools/ab_layout.py [workload=4k64] [steps=10] [dtype=f32]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
import colorvideovdp_amd as cv

wl = sys.argv[1] if len(sys.argv) > 1 else "4k64"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
W, H, frames, fps, display, _per, wl_dtype, _heat = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
clip = bench.ResidentClip(frames, 0, frames, H, W, fps, dtype, dev, gen="gpu")
qs = {}
for rep in range(2):
    for layout in (1, 0):
        m = cv.cvvdp(display_name=display, device=dev)
        m.band_layout = layout
        t_end = time.time() + 1.0
        while time.time() < t_end:
            m.predict_video_source(clip)
        torch.cuda.synchronize()
        m.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            jod, st = m.predict_video_source(clip)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        prof = m.profile_read()
        m.profile(False)
        qs[layout] = st["Q_per_ch"]
        print(f"layout {layout} ({'front/back waves' if layout == 0 else 'one wave per channel'}): step {dt:7.3f} ms   " +
              "  ".join(f"{k} {v[0] / steps:6.3f}" for k, v in prof.items()) + f"   JOD {float(jod):.5f}  fused levels {m.fused_levels}", flush=True)
print("Q_per_ch: bit-identical", bool(np.array_equal(qs[0], qs[1])), " max relative difference %.2e (two separately compiled kernels: the last bit)" % float(np.max(np.abs(qs[0] - qs[1]) / np.maximum(np.abs(qs[1]), 1e-30))))
