import sys
import numpy as np, torch
sys.path.insert(0, ".")
import colorvideovdp_amd as cv
from oracle import cvvdp_oracle as orc
rng = np.random.default_rng(5)
bad = 0
for (H, W) in [(4, 4), (5, 7), (8, 8), (9, 16), (16, 9), (12, 33), (33, 12), (7, 64), (64, 7), (15, 15), (6, 6), (3, 9), (2, 2), (1, 8)]:
    for F, fps in ((1, 0), (3, 30)):
        ref = rng.random((1, 3, F, H, W)).astype(np.float32)
        test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
        try:
            oj, os_ = orc.Oracle(display_name="standard_fhd").predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
            o_err = None
        except Exception as e:
            o_err = type(e).__name__ + ": " + str(e)[:60]
        try:
            j, s = cv.cvvdp(display_name="standard_fhd").predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
            h_err = None
        except Exception as e:
            h_err = type(e).__name__ + ": " + str(e)[:60]
        if o_err or h_err:
            print(f"{W}x{H}x{F}: oracle {o_err} | hip {h_err}")
            continue
        dq = np.abs(s["Q_per_ch"] - os_["Q_per_ch"]) / (np.abs(os_["Q_per_ch"]) * 2e-4 + 2e-6)
        dj = abs(float(j) - float(oj))
        ok = dj <= 1e-3 and dq.max() <= 1.0 and s["Q_per_ch"].shape == os_["Q_per_ch"].shape
        bad += not ok
        print(("ok  " if ok else "BAD ") + f"{W}x{H}x{F}: bands {s['Q_per_ch'].shape[-1]} dJOD {dj:.1e} Q err/tol {dq.max():.2f}")
print("bad:", bad)
