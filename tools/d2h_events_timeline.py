"""HIP-event timeline of the heat-map D2H stream of configs[4] with a host sink (rocprofv3's --memory-copy-trace records
no row for these copies on this stack, profiles/r06_d2h_timeline.txt): per piece, when its kernels were done, when its copy started and ended
(ms from the step's start), the copy's rate, and how long the copy engine sat idle before it.
    python tools/d2h_events_timeline.py [frames]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import colorvideovdp_amd as cv
from colorvideovdp_amd.heatmap_writers import HeatmapFrameMeans

F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
clip = bench.ResidentClip(F, 0, F, 4320, 7680, 60, "u8", dev, gen="gpu", pq_range=True)
m = cv.cvvdp(display_name="standard_hdr_pq", heatmap="supra-threshold")
sink = HeatmapFrameMeans(uint8=True)
for _ in range(2):
    m.predict_video_source(clip, heatmap_sink=sink)
torch.cuda.synchronize()
for rep in range(2):
    m.d2h_trace = []
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    m.predict_video_source(clip, heatmap_sink=sink)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    tr, m.d2h_trace = m.d2h_trace, None
    print(f"# run {rep}: {F} frames of 8K, 8-bit RGB heat map to the host: {wall:.1f} ms wall; temporal block {m.last_block_frames}, pieces of {m.last_score_frames}")
    print("%6s %4s %9s %12s %11s %10s %8s %9s" % ("frame", "n", "MB", "kernels_done", "copy_start", "copy_end", "GB/s", "idle_ms"))
    prev, busy, idle = None, 0.0, 0.0
    for f0, n, nb, ek, ea, eb, *rest in tr:
        k, a, b = e0.elapsed_time(ek), e0.elapsed_time(ea), e0.elapsed_time(eb)
        gap = 0.0 if prev is None else max(0.0, a - prev)
        print("%6d %4d %9.0f %12.2f %11.2f %10.2f %8.1f %9.2f" % (f0, n, nb / 1e6, k, a, b, nb / 1e6 / max(b - a, 1e-6), gap))
        prev, busy, idle = b, busy + (b - a), idle + gap
        if rest:
            print("        host (ms from the step's start): fetch entered %.2f, buffer allocated +%.2f, heat-map kernels queued +%.2f, sink flushed +%.2f, copy queued +%.2f"
                  % tuple([(rest[0][0] - t0) * 1e3] + [(rest[0][i] - rest[0][i - 1]) * 1e3 for i in range(1, len(rest[0]))]))
    tot = sum(t[2] for t in tr)
    print(f"# copies: {tot / 1e9:.2f} GB, engine busy {busy:.1f} ms = {tot / 1e6 / busy:.1f} GB/s while copying; first copy starts at {e0.elapsed_time(tr[0][4]):.1f} ms; "
          f"idle between copies {idle:.1f} ms; last copy done at {prev:.1f} ms")
