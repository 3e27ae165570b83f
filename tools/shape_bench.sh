#!/bin/bash
# throughput on other shapes (64 frames, u8, resident): tools/shape_bench.sh "1366x768 1360x768 854x480 848x480 ..."
# per shape: wall time of 10 unprofiled calls, then the per-kernel-family milliseconds of 5 profiled calls (HIP events; the
# edge strips of a ragged width run in the main stream while profiling, so the families add up to a little more than the wall time)
python - "$@" <<'PY'
import os, sys, time, torch
sys.path.insert(0, ".")
import bench, colorvideovdp_amd as cv
dev = torch.device("cuda")
for shape in (sys.argv[1:] or ["1366x768", "1360x768", "854x480", "848x480", "1920x1080", "2560x1440"]):
    W, H = (int(v) for v in shape.split("x"))
    clip = bench.ResidentClip(64, 0, 64, H, W, 60, "u8", dev)
    m = cv.cvvdp(display_name="standard_fhd")
    m.fuse_mode = int(os.environ.get("SHAPE_FUSE_MODE", "0"))      # 0: the product's own choice; 1 / 2: the test hook (fused wherever possible / never)
    for _ in range(3):
        m.predict_video_source(clip)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        jod, _ = m.predict_video_source(clip)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    m.profile(True)
    for _ in range(5):
        m.predict_video_source(clip)
    torch.cuda.synchronize()
    prof = {k: round(v[0] / 5, 3) for k, v in m.profile_read().items() if v[0] > 0}
    m.profile(False)
    print(f"{shape} (fused levels {m.fused_levels}): {dt * 1e3:.3f} ms per 64 frames, {W * H * 64 / dt / 1e9:.2f} Gpixel/s, JOD {float(jod):.4f}  kernels {prof}")
PY
