"""Side-by-side intermediates (Gaussian pyramid, per-pixel D) of HIP and oracle for a case dumped by tools/fuzz_shapes.py:
    python tools/dbg_case.py gpurun_out/fuzz_bad/seedN_caseK.npz [frame]
The intermediates are those of the LAST frame; with a frame index the clip is cut after that frame (the temporal filter is causal,
so the frame scores as it did in the whole clip)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import colorvideovdp_amd as cv
from colorvideovdp_amd import _capi
from oracle import cvvdp_oracle as orc

d = dict(np.load(sys.argv[1]))
test, ref, fps, disp, pad = d["test"], d["ref"], int(d["fps"]), str(d["display"]), str(d["padding"])
if len(sys.argv) > 2:
    test, ref = test[:, :, :int(sys.argv[2]) + 1], ref[:, :, :int(sys.argv[2]) + 1]
F = test.shape[2]
m = cv.cvvdp(display_name=disp, temp_padding=pad)
m.debug_dump = True
j, s = m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
o = orc.Oracle(display_name=disp, temp_padding=pad, keep=True)
oj, os_ = o.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
dbg = o.dbg
L = len(dbg["gpyr"])
nch = 4 if fps > 0 else 3
for l in range(L):
    H, W = dbg["gpyr"][l].shape[-2:]
    buf = m.debug_buffer(_capi.BUF_GPYR, l).cpu().numpy().reshape(2 * nch, -1, H, W)[:, F - 1]
    want = dbg["gpyr"][l][0, :, 0].numpy()
    err = np.abs(buf - want) / (np.abs(want) + 1e-6)
    dd = m.debug_buffer(_capi.BUF_DDUMP, l).cpu().numpy().reshape(nch, -1, H, W)[:, F - 1]
    wd = dbg["D"][l][0, :, 0].numpy()
    ed = np.abs(dd - wd) / (np.abs(wd) + 1e-5)
    print(f"level {l} {W}x{H}: gpyr max rel err per plane {np.round(err.reshape(2 * nch, -1).max(1), 7)}")
    print(f"          D max rel err per channel {np.round(ed.reshape(nch, -1).max(1), 6)}  mean |D| {np.round(np.abs(wd).reshape(nch, -1).mean(1), 5)}")
q, qo = s["Q_per_ch"], os_["Q_per_ch"]
print("Q rel err [ch, band] (last frame):\n", np.round(np.abs(q - qo)[0, :, F - 1] / (np.abs(qo[0, :, F - 1]) + 1e-9), 6))
