#!/usr/bin/env python3
"""tests/golden/fullsize_*.npz: outputs of the REAL reference on the first frames of bench.py's synthetic 4K clip.

The inputs are not stored (50 MB): bench.synth_frame(f, H, W, "cpu") regenerates them bit-exactly on any host with the
same torch build (CPU generator), so the fixture holds only the reference's JOD and Q_per_ch.  Because the temporal
filter is causal, the features of a k-frame prefix equal those of the full clip, so this pins the 4K
workload against the reference itself rather than against the oracle.  Container only (imports /root/reference).

    python oracle/make_goldens_fullsize.py
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
import bench

OUT = os.path.join(HERE, "..", "tests", "golden")


def main():
    cases = (("fullsize_4k_3f", 3840, 2160, 3, 60, "standard_4k"), ("fullsize_fhd_4f", 1920, 1080, 4, 60, "standard_fhd"),
             ("fullsize_8k_pq_2f", 7680, 4320, 2, 60, "standard_hdr_pq"))
    only = sys.argv[1:]
    for name, W, H, F, fps, disp in cases:
        if only and name not in only:
            continue
        frames = [bench.synth_frame(f, H, W, "cpu") for f in range(F)]
        t = torch.stack([a for a, _ in frames], dim=1)[None]   # [1,3,F,H,W] uint8
        r = torch.stack([b for _, b in frames], dim=1)[None]
        met = pycvvdp.cvvdp(display_name=disp, device=torch.device("cpu"), quiet=True)
        with torch.no_grad():
            jod, stats = met.predict(t.numpy(), r.numpy(), dim_order="BCFHW", frames_per_second=fps)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), width=W, height=H, frames=F, fps=fps, display=disp,
                            jod=np.float32(jod.item()), Q_per_ch=stats["Q_per_ch"], rho_band=stats["rho_band"],
                            checksum_test=np.int64(t.to(torch.int64).sum().item()), checksum_ref=np.int64(r.to(torch.int64).sum().item()),
                            torch_version=torch.__version__)
        print(name, float(jod), stats["Q_per_ch"].shape)


if __name__ == "__main__":
    main()
