#!/bin/bash
# Number of fused levels against clip length / frame size (dev build: make dev; run on the GPU box):
#   tools/fuse_short_sweep.sh "3840x2160x8 3840x2160x16 ..."
# per clip: the product's own choice, then 0..3 leading levels forced (fuse_mode = 1 capped by the dev knob CVVDP_FUSE_LEVELS)
for clip in ${1:-3840x2160x8 3840x2160x16 3840x2160x32 2560x1440x16 2560x1440x32 1920x1080x16 1920x1080x32 1360x768x64}; do
  for cap in own 0 1 2 3; do
    CLIP=$clip CAP=$cap CVVDP_FUSE_LEVELS=$([ $cap = own ] && echo 8 || echo $cap) CVVDP_DEV_KNOBS=1 CVVDP_LIB=$PWD/colorvideovdp_amd/libcvvdp_hip_dev.so python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, torch
sys.path.insert(0, ".")
import bench, colorvideovdp_amd as cv
W, H, F = (int(v) for v in os.environ["CLIP"].split("x"))
cap = os.environ["CAP"]
clip = bench.ResidentClip(F, 0, F, H, W, 60, "u8", torch.device("cuda"))
m = cv.cvvdp(display_name="standard_4k")
m.fuse_mode = 0 if cap == "own" else (2 if cap == "0" else 1)
for _ in range(4):
    m.predict_video_source(clip)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(12):
    m.predict_video_source(clip)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12
print(f"{os.environ['CLIP']} {cap:>3}: fused levels {m.fused_levels}, {dt * 1e3:.3f} ms")
PY
  done
done
