import torch, time, sys
sys.path.insert(0, ".")
import colorvideovdp_amd as cv, bench
clip = bench.ResidentClip(32, 0, 32, 2160, 3840, 60, "u8", torch.device("cuda"))
for mode in ("raw", "supra-threshold"):
    m = cv.cvvdp(display_name="standard_4k", heatmap=mode)
    jod, st = m.predict_video_source(clip); torch.cuda.synchronize()
    t0 = time.time(); jod, st = m.predict_video_source(clip); torch.cuda.synchronize(); dt = time.time() - t0
    print(mode, float(jod), tuple(st["heatmap"].shape), "%.1f ms  %.0f Mpix/s" % (dt * 1e3, 3840 * 2160 * 32 / dt / 1e6))
