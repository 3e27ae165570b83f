"""How close to the real reference are the 8-bit full-size fixtures with / without the per-code EOTF table?
    python tools/lut_parity.py ; CVVDP_NO_EOTF_LUT=1 python tools/lut_parity.py"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import colorvideovdp_amd as cv
from conftest import fullsize_cases, fullsize_inputs, load_golden
for name in fullsize_cases():
    g = load_golden(name)
    inp = fullsize_inputs(g)
    if inp is None:
        print(name, "inputs not reproducible here"); continue
    j, s = cv.cvvdp(display_name=str(g["display"])).predict(inp[0], inp[1], dim_order="BCFHW", frames_per_second=float(g["fps"]))
    q, qr = s["Q_per_ch"], g["Q_per_ch"]
    rel = np.abs(q - qr) / (np.abs(qr) + 1e-8)
    print(f"{name} {inp[0].dtype} {g['display']}: LUT {'off' if os.environ.get('CVVDP_NO_EOTF_LUT') == '1' else 'on'}  dJOD {abs(float(j) - float(g['jod'])):.2e}  Q rel err max {rel.max():.2e} mean {rel.mean():.2e}")
