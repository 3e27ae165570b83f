"""PCIe-inclusive rate: predict() on clips that live in host memory (numpy), the reference's normal calling convention.
    python tools/host_input_bench.py [frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import colorvideovdp_amd as cv

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 3840, 2160
frames = [bench.synth_frame(f, H, W, torch.device("cuda")) for f in range(F)]
t8 = torch.stack([a for a, _ in frames], dim=1)[None].cpu().numpy()   # [1,3,F,H,W] uint8
r8 = torch.stack([b for _, b in frames], dim=1)[None].cpu().numpy()
del frames
m = cv.cvvdp(display_name="standard_4k")
for name, t, r in (("u8 numpy", t8, r8), ("f32 numpy", t8.astype(np.float32) / 255, r8.astype(np.float32) / 255),
                   ("u8 pinned torch", torch.from_numpy(t8).pin_memory(), torch.from_numpy(r8).pin_memory())):
    m.predict(t, r, dim_order="BCFHW", frames_per_second=60); torch.cuda.synchronize()
    t0 = time.perf_counter()
    jod, _ = m.predict(t, r, dim_order="BCFHW", frames_per_second=60); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = 2 * (t.nbytes if isinstance(t, np.ndarray) else t.numel() * t.element_size())
    print("%-16s JOD %.4f  %.1f ms  %.0f Mpix/s  (%.1f GB over PCIe: %.1f GB/s)" % (name, float(jod), dt * 1e3, W * H * F / dt / 1e6, nbytes / 1e9, nbytes / dt / 1e9))
