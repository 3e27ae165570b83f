#!/usr/bin/env python3
"""Container-only development aid (imports the REAL reference from /root/reference, like make_goldens*.py): score a case that
tools/fuzz_shapes.py dumped on the GPU box (gpurun_out/fuzz_bad/seedN_caseK.npz: inputs, the HIP path's and the oracle's Q_per_ch)
with the reference itself, to see which of the two the reference sides with.

    python oracle/ref_vs_fuzz_case.py gpurun_out/fuzz_bad/seed13_case19.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import pycvvdp


def as_float(a):
    if a.dtype == np.uint8:
        return torch.tensor(a.astype(np.float32) / 255.0)
    if a.dtype == np.uint16:
        return torch.tensor(a.astype(np.float32) / 65535.0)
    return torch.tensor(a.astype(np.float32))


d = dict(np.load(sys.argv[1]))
fps, disp, pad = int(d["fps"]), str(d["display"]), str(d["padding"])
m = pycvvdp.cvvdp(display_name=disp, device=torch.device("cpu"), temp_padding=pad)
j, s = m.predict(as_float(d["test"]), as_float(d["ref"]), dim_order="BCFHW", frames_per_second=fps)
q = s["Q_per_ch"].numpy() if torch.is_tensor(s["Q_per_ch"]) else np.asarray(s["Q_per_ch"])
qh, qo = d["q_hip"], d["q_oracle"]
tol = np.abs(q) * 2e-4 + 2e-6                      # the tolerance of tests/test_gpu_parity.py
print(f"JOD: reference {float(j):.6f}, HIP {float(d['jod_hip']):.6f}")
print(f"Q_per_ch, worst entry in units of the tolerance: oracle vs reference {(np.abs(qo - q) / tol).max():.3f}, HIP vs reference {(np.abs(qh - q) / tol).max():.3f}")
r = np.abs(qh - q) / tol
for i in np.argwhere(r > 0.8)[:10]:
    i = tuple(int(v) for v in i)
    print(f"  [batch, channel, frame, band] = {i}: reference {q[i]:.7g}, oracle {qo[i]:.7g}, HIP {qh[i]:.7g}")
