#!/usr/bin/env python3
"""tests/golden/deep_8k_pqrange_heat_17f.npz: the REAL reference on the first 17 frames of configs[4]'s clip EXACTLY AS bench.py --workload
8k256pq MAKES IT (7680x4320, standard_hdr_pq, uint8 codes mapped into the PQ range [0.10, 0.75], bench.ResidentClip(pq_range=True)) with
its supra-threshold heat map and distogram arrays.

deep_8k_pq_heat_17f (make_goldens_8k17.py) holds the heat map of the FULL-RANGE codes on the PQ display, deep_8k_pq_80f
(make_goldens_8k80.py) the scores -- no heat map -- of the PQ-range clip: until round 6 no heat-map frame of the clip configs[4] is
timed on had met the reference.  The temporal filter is causal and the tone curve per frame, so the heat-map frames 0..16 and the
scores of a 17-frame clip are those of the 256-frame clip.  Stored as in make_goldens_8k17.py.  Container only (imports /root/reference
through oracle/ref_shims); about 25 minutes on 8 cores.

    python oracle/make_goldens_8k17_pqrange.py
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)

import numpy as np
import torch

import pycvvdp
import bench
from make_goldens_outputs import distogram_arrays

OUT = os.path.join(HERE, "..", "tests", "golden")
W, H, F, FPS, DISP, HEAT = 7680, 4320, 17, 60, "standard_hdr_pq", "supra-threshold"
KEEP = (0, 8, 16)


def main():
    t0 = time.time()
    t = torch.empty((1, 3, F, H, W), dtype=torch.uint8)
    r = torch.empty((1, 3, F, H, W), dtype=torch.uint8)
    cs_t = cs_r = 0
    for f in range(F):
        a, b = bench.synth_frame(f, H, W, "cpu")
        a, b = ((x.float() * 0.65 + 0.10 * 255).round().to(torch.uint8) for x in (a, b))     # bench.ResidentClip(pq_range=True): codes in [0.10, 0.75] = the clip bench.py --workload 8k256pq times
        cs_t += int(a.to(torch.int64).sum())
        cs_r += int(b.to(torch.int64).sum())
        t[0, :, f], r[0, :, f] = a, b
    print(f"frames made {time.time() - t0:.0f} s", flush=True)
    met = pycvvdp.cvvdp(display_name=DISP, device=torch.device("cpu"), quiet=True, heatmap=HEAT)
    with torch.no_grad():
        jod, stats = met.predict(t.numpy(), r.numpy(), dim_order="BCFHW", frames_per_second=FPS)
    print(f"reference done {time.time() - t0:.0f} s  jod {float(jod):.5f}", flush=True)
    hm = stats["heatmap"]                                   # [1,3,F,H,W] fp16
    st = {k: v for k, v in stats.items() if k != "heatmap"}
    q = stats["Q_per_ch"].copy()
    d_auto = distogram_arrays(met, st, None)
    d_10 = distogram_arrays(met, st, 10)
    np.savez_compressed(os.path.join(OUT, "deep_8k_pqrange_heat_17f.npz"), width=W, height=H, frames=F, fps=FPS, display=DISP, dtype="u8",
                        jod=np.float32(jod.item()), Q_per_ch=q, rho_band=stats["rho_band"],
                        checksum_test=np.int64(cs_t), checksum_ref=np.int64(cs_r), torch_version=torch.__version__,
                        reference_seconds=np.float32(time.time() - t0), heatmap_mode=HEAT, heatmap_frames=np.array(KEEP),
                        heatmap_frame_means=hm[0].float().mean(dim=(0, 2, 3)).numpy().astype(np.float32),
                        heatmap_ds=hm[0][:, list(KEEP), ::16, ::16].numpy().astype(np.float16),
                        disto_auto=d_auto, disto_10=d_10)
    print("saved", q.shape, d_auto.shape, f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
