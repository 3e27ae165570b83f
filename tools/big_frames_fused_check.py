"""Fused against unfused route on frames larger than the bench clip (run on the GPU box): time, fused levels and the distance of the two
routes' Q_per_ch -- 8K, 5K, DCI 4K and a ragged (W % 4 == 2) 3414-wide frame.   python tools/big_frames_fused_check.py"""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import bench, colorvideovdp_amd as cv
dev = torch.device("cuda")
for (W, H, F) in ((7680, 4320, 24), (5120, 2880, 32), (4096, 2160, 40), (3414, 1920, 40)):
    clip = bench.ResidentClip(F, 0, F, H, W, 60, "u8", dev)
    qs = {}
    for mode in (0, 2):
        m = cv.cvvdp(display_name="standard_4k")
        m.fuse_mode = mode
        for _ in range(2):
            jod, st = m.predict_video_source(clip)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            jod, st = m.predict_video_source(clip)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        qs[mode] = st["Q_per_ch"]
        print(f"{W}x{H}x{F} fuse_mode {mode}: fused levels {m.fused_levels}, {dt*1e3:.2f} ms, {W*H*F/dt/1e9:.2f} Gpixel/s, JOD {float(jod):.5f}", flush=True)
    r = np.abs(qs[0] - qs[2]) / (np.abs(qs[2]) * 5e-5 + 5e-7)
    print("   fused vs unfused Q_per_ch, worst entry in units of rtol 5e-5 / atol 5e-7:", float(r.max()))
    del clip
    torch.cuda.empty_cache()
