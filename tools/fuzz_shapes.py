"""Randomised shape sweep of the HIP path against the CPU oracle (development aid, run on the GPU box):
    python tools/fuzz_shapes.py [n_cases] [seed] [fuse_heat]
Random widths / heights (16..700 x 16..400, all residues mod 16), frame counts, frame rates, displays, padding and heat-map modes;
every second video case without a heat map forces the fused band kernels (test hook fuse_mode = 1) wherever a level supports them;
with a third argument the heat-map video cases run on the fused kernels too (k_band4s_heat / k_band4f_heat)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import colorvideovdp_amd as cv
from oracle import cvvdp_oracle as orc

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 30), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
from tools import fuzz_cases
for cs in fuzz_cases.cases(seed, n):
    k, W, H, F, fps, disp, pad, heat, dt, B = (cs[x] for x in ("k", "W", "H", "F", "fps", "display", "padding", "heatmap", "dtype", "B"))
    test, ref = fuzz_cases.as_input(cs["test"]), fuzz_cases.as_input(cs["ref"])
    o = orc.Oracle(display_name=disp, temp_padding=pad, heatmap=heat)
    oj, os_ = o.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    m = cv.cvvdp(display_name=disp, temp_padding=pad, heatmap=heat, block_frames=cs["block_frames"])
    fm = cs["fuse_mode"]
    if len(sys.argv) > 3 and heat and F > 1:
        fm = 1
    m.fuse_mode = fm
    j, s = m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    dq = np.abs(s["Q_per_ch"] - os_["Q_per_ch"]) / (np.abs(os_["Q_per_ch"]) * 2e-4 + 2e-6)
    dj = float(np.max(np.abs(np.atleast_1d(j.cpu().numpy()) - np.atleast_1d(np.asarray(oj)))))
    msg = f"{k:3d} {W}x{H}x{F} B{B} {dt} C{test.shape[1]} fps {fps} {disp} {pad} heat {heat} fused {m.fused_levels if F > 1 else 0}: dJOD {dj:.2e}  Q err/tol {dq.max():.2f}"
    ok = dj <= 1e-3 and dq.max() <= 1.0
    if heat:
        a = s["heatmap"].numpy().astype(np.float32)
        b = os_["heatmap"]
        b = (b.numpy() if torch.is_tensor(b) else b).astype(np.float32)
        d = np.abs(a - b)
        msg += f"  heat frac>2e-3 {(d > 2e-3).mean():.1e} max {d.max():.1e}"
        ok = ok and (d > 2e-3).mean() < 1e-3 and d.max() < 2e-2
    print(("ok  " if ok else "BAD ") + msg, flush=True)
    if not ok:                                         # keep the case for a closer look (oracle / reference / HIP side by side)
        import os
        os.makedirs("gpurun_out/fuzz_bad", exist_ok=True)
        tt, rr = (x.numpy() if torch.is_tensor(x) else x for x in (test, ref))
        np.savez_compressed(f"gpurun_out/fuzz_bad/seed{seed}_case{k}.npz", test=tt, ref=rr, fps=fps, display=disp, padding=pad, heat=str(heat),
                            q_hip=s["Q_per_ch"], q_oracle=os_["Q_per_ch"], jod_hip=float(np.atleast_1d(j.cpu().numpy())[0]))
    bad += not ok
print("bad:", bad)
