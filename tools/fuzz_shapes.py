"""Randomised shape sweep of the HIP path against the CPU oracle (development aid, run on the GPU box):
    python tools/fuzz_shapes.py [n_cases] [seed]
Random widths / heights (16..700 x 16..400, all residues mod 16), frame counts, frame rates, displays, padding and heat-map modes;
every second video case without a heat map forces the fused band kernels (test hook fuse_mode = 1) wherever a level supports them."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import colorvideovdp_amd as cv
from oracle import cvvdp_oracle as orc

n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 30), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rng = np.random.default_rng(seed)
bad = 0
for k in range(n):
    W, H = int(rng.integers(16, 700)), int(rng.integers(16, 400))
    F = int(rng.choice([1, 1, 2, 3, 7]))
    fps = 0 if F == 1 else int(rng.choice([24, 30, 50, 60, 120]))
    disp = str(rng.choice(["standard_fhd", "standard_4k", "standard_hdr_pq"]))
    pad = str(rng.choice(["replicate", "symmetric"]))
    heat = str(rng.choice(["none", "none", "raw", "threshold", "supra-threshold"]))
    heat = None if heat == "none" else heat
    y, x = np.mgrid[0:H, 0:W]
    ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (3.1 * x / W + f / 9.0) + c) * np.cos(2 * np.pi * 2.3 * y / H) for c in range(3)])
                    for f in range(F)], axis=1)[None]
    test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1)
    dt = str(rng.choice(["u8", "u8", "u16", "f32", "f16"]))
    B = 1 if heat else int(rng.choice([1, 1, 2]))
    if B == 2:                                        # a batch with a broadcast reference (video_source.py:247-252)
        test = np.concatenate([test, np.clip(test + 0.02 * rng.standard_normal(test.shape), 0, 1)], axis=0)
    if rng.random() < 0.15:                           # luminance-only content
        test, ref = test[:, :1], ref[:, :1]
    if dt == "u8":
        test, ref = np.round(test * 255).astype(np.uint8), np.round(np.clip(ref, 0, 1) * 255).astype(np.uint8)
    elif dt == "u16":
        test, ref = np.round(test * 65535).astype(np.uint16), np.round(np.clip(ref, 0, 1) * 65535).astype(np.uint16)
    elif dt == "f16":
        test, ref = torch.tensor(test.astype(np.float16)), torch.tensor(np.clip(ref, 0, 1).astype(np.float16))
    else:
        test, ref = test.astype(np.float32), np.clip(ref, 0, 1).astype(np.float32)
    o = orc.Oracle(display_name=disp, temp_padding=pad, heatmap=heat)
    oj, os_ = o.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    m = cv.cvvdp(display_name=disp, temp_padding=pad, heatmap=heat, block_frames=int(rng.choice([1, 2, 64])))
    fm = 1 if (F > 1 and not heat and rng.random() < 0.5) else 0
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
