#!/usr/bin/env python3
"""gpurun_out/observed/*.json (left by a GPU run of tests/test_fuzz_goldens.py and the heat-map checks of tests/test_gpu_parity.py)
-> tests/golden/observed_bounds.json: per fixture 1.5 x the observed error, with floors so that a bound is never tighter than
rounding noise (Q_per_ch: 0.3 of the generic tolerance = 6e-5 relative; heat map: one fp16 ulp near 1 for the maximum, 1e-5 of the
pixels for the fraction, 1e-5 for the mean).
Usage: tools/make_observed_bounds.py"""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"fuzz": {}, "heatmap": {}, "rule": "bound = max(1.5 x observed, floor); floors: q_err_over_tol 0.3, heat-map max 1e-3, frac 1e-5, mean 1e-5"}
for p in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "observed", "*.json"))):
    kind, name = os.path.basename(p)[:-5].split("__", 1)
    v = json.load(open(p))
    if kind == "fuzz":
        out["fuzz"][name] = {"q_err_over_tol": round(max(1.5 * v["q_err_over_tol"], 0.3), 4), "observed": v}
    elif kind == "heatmap":
        out["heatmap"][name] = {"frac_gt_2e-3": max(1.5 * v["frac_gt_2e-3"], 1e-5), "max": max(1.5 * v["max"], 1e-3),
                                "mean": max(1.5 * v["mean"], 1e-5), "observed": v}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "observed_bounds.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
