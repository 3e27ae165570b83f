"""Tuning experiment: frame-range lanes on separate HIP streams of ONE GPU (two handles / workspaces)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
import colorvideovdp_amd as cv

dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
clip = bench.ResidentClip(n, 0, n, 2160, 3840, 60, "f32", dev)


def run(ranges, reps=3, block=None):
    ms = [cv.cvvdp(display_name="standard_4k", block_frames=block) for _ in ranges]
    ss = [torch.cuda.Stream() for _ in ranges]
    def once():
        qs = []
        for m, s, (a, c) in zip(ms, ss, ranges):
            with torch.cuda.stream(s):
                qs.append(m._score_range(clip, a, c)[0])
        return qs
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        qs = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, torch.cat(qs, dim=2)

t1, q1 = run([(0, n)])
print("1 lane : %.2f ms  %.0f Mpix/s" % (t1 * 1e3, 3840 * 2160 * n / t1 / 1e6))
for lanes in (2, 3, 4):
    base, extra = divmod(n, lanes)
    r, a = [], 0
    for i in range(lanes):
        c = base + (1 if i < extra else 0); r.append((a, c)); a += c
    t, q = run(r)
    print("%d lanes: %.2f ms  %.0f Mpix/s  exact=%s" % (lanes, t * 1e3, 3840 * 2160 * n / t / 1e6, bool(torch.equal(q, q1))))
