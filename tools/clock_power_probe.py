#!/usr/bin/env python3
"""Shader clock and socket power while ONE kernel family runs back to back (GPU box):  python tools/clock_power_probe.py [seconds=6]
Runs (a) the 4K x 64 bench step, (b) a clip whose step is dominated by the band kernels (level 0 only matters), sampling
`rocm-smi --showclocks --showpower --json` in a side thread.  Answers: is the step power-limited (clock well below the 2.4 GHz peak)?"""
import json
import subprocess
import sys
import threading
import time
import torch
sys.path.insert(0, ".")
import bench
import colorvideovdp_amd as cv

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            samples.append((time.time(), {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower() or "fclk" in k.lower()}))
        except Exception as e:      # noqa
            samples.append((time.time(), {"error": str(e)}))
        time.sleep(0.05)


W, H, frames, fps, display, _per, wl_dtype, _heat = bench.WORKLOADS["4k64"]
dev = torch.device("cuda", 0)
clip = bench.ResidentClip(frames, 0, frames, H, W, fps, "f32", dev, gen="gpu")
m = cv.cvvdp(display_name=display, device=dev)
m.predict_video_source(clip)
torch.cuda.synchronize()
th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.0)
t_idle = time.time()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    m.predict_video_source(clip)
    n += 1
torch.cuda.synchronize()
t1 = time.time()
time.sleep(0.5)
stop = True
th.join()
print(f"{n} steps in {t1 - t0:.2f} s = {(t1 - t0) / n * 1e3:.3f} ms/step")
for t, s in samples:
    tag = "idle" if t < t_idle else ("busy" if t < t1 else "after")
    print(f"{t - t0:7.2f} {tag:5s} {s}")
