"""Heat-map + distogram path at BASELINE.json config sizes (run on the GPU box):
    python tools/heatmap_bench.py [4k|8k] [frames]
4k: 3840x2160 sRGB u8, standard_4k; 8k: 7680x4320 PQ (codes mapped into [0.10, 0.75]), standard_hdr_pq -- the
per-GPU share of configs[4] (256 frames over 8 GPUs = 32 frames + 16 halo frames)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import colorvideovdp_amd as cv

which = sys.argv[1] if len(sys.argv) > 1 else "4k"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W, H, disp = (3840, 2160, "standard_4k") if which == "4k" else (7680, 4320, "standard_hdr_pq")
dev = torch.device("cuda")
clip = bench.ResidentClip(frames, 0, frames, H, W, 60, "u8", dev)
if which == "8k":   # PQ content: keep the codes inside [0.10, 0.75] (about 0.3 .. 1000 cd/m^2), SURVEY 8(d)
    for a in (clip.test, clip.ref):
        a.copy_((a.float() * (0.65) + 0.10 * 255).round().to(torch.uint8))
for mode in ("raw", "supra-threshold"):
    m = cv.cvvdp(display_name=disp, heatmap=mode)
    torch.cuda.reset_peak_memory_stats()
    jod, st = m.predict_video_source(clip); torch.cuda.synchronize()
    del st                                         # releases the page-locked heat-map buffer for the next call
    t0 = time.time(); jod, st = m.predict_video_source(clip); torch.cuda.synchronize(); dt = time.time() - t0
    print(which, mode, "JOD %.4f" % float(jod), tuple(st["heatmap"].shape), st["heatmap"].dtype, "block", m.last_block_frames,
          "%.1f ms  %.0f Mpix/s  peak %.1f GB" % (dt * 1e3, W * H * frames / dt / 1e6, torch.cuda.max_memory_allocated() / 1e9))
if os.environ.get("DISTOGRAM", "1") == "1":
    try:
        m.export_distogram(st, "/tmp/distogram.png", jod_max=10)
        print("distogram written:", os.path.getsize("/tmp/distogram.png"), "bytes")
    except Exception as e:   # matplotlib is optional on the box
        print("distogram skipped:", repr(e))
if os.environ.get("BREAKDOWN", "0") == "1":
    m = cv.cvvdp(display_name=disp, heatmap="supra-threshold")
    m.predict_video_source(clip); torch.cuda.synchronize()
    m.profile(True)
    t0 = time.time(); m.predict_video_source(clip); torch.cuda.synchronize(); dt = time.time() - t0
    print("wall %.1f ms; kernel families (ms):" % (dt * 1e3), {k: round(v[0], 2) for k, v in m.profile_read().items()})
