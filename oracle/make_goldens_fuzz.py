#!/usr/bin/env python3
"""tests/golden/fuzz_seed<S>_case<K>.npz: the REAL reference on the thin class of the randomised sweep.

tools/fuzz_shapes.py (560 cases on the final round-3 build) left five cases outside the Q_per_ch tolerance of the parity tests
(1.04-2.6 x rtol 2e-4): the coarsest Laplacian bands -- a few dozen pixels -- of luminance-only clips, where g_l - expand(g_{l+1}) is
a 1e-4 difference of its operands, and one PQ fp16 clip.  JOD agreed to 1e-6 on all of them, so nothing in tests/ saw them.  This
recipe has the reference itself score those five cases and five neighbours of the same kind that pass, so that tests/test_fuzz_goldens.py
can hold the HIP path to a stated, per-fixture bound there (tests/golden/observed_bounds.json) and the coarse-band Laplacian
(lpyr_dec.py:386-408) cannot drift unnoticed.

The inputs are not stored: case K of seed S is a pure function of (S, K) (tools/fuzz_cases.py replays the sweep's generator); the
fixtures hold checksums of the inputs and the reference's outputs.  Container only (imports /root/reference).

    python oracle/make_goldens_fuzz.py [seedS_caseK ...]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, ".."))

import numpy as np
import torch

import pycvvdp
from tools import fuzz_cases

OUT = os.path.join(HERE, "..", "tests", "golden")
# (seed, case, why)
CASES = [
    (13, 0, "outlier r3: 1.20 x tol, band 6 of a luminance-only u8 clip, 120 fps, symmetric padding, supra-threshold heat map"),
    (21, 38, "outlier r3: 1.09 x tol, band 6 of a luminance-only u16 clip"),
    (35, 29, "outlier r3: 2.60 x tol (the worst of the sweep), band 6 of a luminance-only f32 clip, 120 fps"),
    (41, 23, "outlier r3: 1.04 x tol, band 6 of a luminance-only u8 clip"),
    (44, 38, "outlier r3: 1.15 x tol, PQ fp16 colour clip"),
    (13, 12, "neighbour: luminance-only f32, 120 fps, fused band kernels"),
    (21, 35, "neighbour: luminance-only f32, 7 frames, fused band kernels"),
    (41, 0, "neighbour: PQ fp16 colour clip, fused band kernels"),
    (41, 38, "neighbour: luminance-only f16 with a supra-threshold heat map"),
    (44, 23, "neighbour: luminance-only u8, 120 fps, fused band kernels"),
]


def main():
    only = set(sys.argv[1:])
    for seed, k, why in CASES:
        name = f"fuzz_seed{seed}_case{k}"
        if only and name[5:] not in only and name not in only:
            continue
        c = fuzz_cases.case(seed, k)
        met = pycvvdp.cvvdp(display_name=c["display"], device=torch.device("cpu"), temp_padding=c["padding"], quiet=True)
        with torch.no_grad():
            jod, stats = met.predict(fuzz_cases.as_input(c["test"]), fuzz_cases.as_input(c["ref"]), dim_order="BCFHW", frames_per_second=c["fps"])
        q = stats["Q_per_ch"]
        q = q.numpy() if torch.is_tensor(q) else np.asarray(q)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, case=k, why=why, width=c["W"], height=c["H"], frames=c["F"], fps=c["fps"],
                            display=c["display"], padding=c["padding"], heatmap=str(c["heatmap"]), dtype=c["dtype"], batch=c["B"],
                            channels=c["test"].shape[1], block_frames=c["block_frames"], fuse_mode=c["fuse_mode"],
                            jod=np.asarray(jod.detach().cpu().numpy(), dtype=np.float32), Q_per_ch=q.astype(np.float32), rho_band=stats["rho_band"],
                            checksum_test=np.uint64(fuzz_cases.checksum(c["test"])), checksum_ref=np.uint64(fuzz_cases.checksum(c["ref"])),
                            numpy_version=np.__version__)
        print(name, np.atleast_1d(jod.detach().cpu().numpy()), q.shape, flush=True)


if __name__ == "__main__":
    main()
