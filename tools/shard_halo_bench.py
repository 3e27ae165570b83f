"""Cost of the temporal halo of a frame-range shard on one GPU: scores frames [16, 80) of a 4K clip the way rank 1 of a
sharded job does (16 real halo frames in front), next to frames [0, 64) (rank 0: padding).  python tools/shard_halo_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import colorvideovdp_amd as cv

dev = torch.device("cuda")
clip = bench.ResidentClip(128, 0, 80, 2160, 3840, 60, "f32", dev)
m = cv.cvvdp(display_name="standard_4k")
for first in (0, 16, 0, 16):
    m._score_range(clip, first, 64); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        m._score_range(clip, first, 64)
    torch.cuda.synchronize()
    print("frames [%d, %d): %.2f ms per 64 frames" % (first, first + 64, (time.perf_counter() - t0) / 4 * 1e3))
