#!/usr/bin/env python3
"""tests/golden/bench_*.npz: outputs of the REAL reference on the FULL clips bench.py times (not prefixes).

    bench_fhd64_f32   1920x1080 x 64 frames @60, standard_fhd, fp32 input   (BASELINE.json configs[1], bench --workload fhd64)
    bench_4k64_f32    3840x2160 x 64 frames @60, standard_4k,  fp32 input   (BASELINE.json metric clip, bench --workload 4k64)
    bench_4k256_u8    3840x2160 x 256 frames @60, standard_4k, uint8 input  (configs[2]; u8 so the clip fits host memory)
    bench_4k40_u8_120fps  3840x2160 x 40 frames @120, standard_4k, uint8 input (the 31-tap temporal filters of `bench.py --fps 120` at full size)
    bench_8k_pq_heat_2f  7680x4320 x 2 frames @60, standard_hdr_pq, supra-threshold heat map (configs[4]'s outputs;
                         the fp16 heat map is stored subsampled: every 16th pixel of both frames + its mean)

The inputs are not stored: bench.synth_frame(f, H, W, "cpu") regenerates them bit-exactly (CPU generator), verified by
checksum in the tests.  The fixtures hold the reference's JOD, Q_per_ch and rho_band (a few KB each).  Container only
(imports /root/reference through oracle/ref_shims).

    python oracle/make_goldens_bench.py [case ...]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, ".."))

import numpy as np
import torch

import pycvvdp
import bench

OUT = os.path.join(HERE, "..", "tests", "golden")

CASES = (
    # name, W, H, F, fps, display, dtype, heatmap
    ("bench_fhd64_f32", 1920, 1080, 64, 60, "standard_fhd", "f32", None),
    ("bench_8k_pq_heat_2f", 7680, 4320, 2, 60, "standard_hdr_pq", "u8", "supra-threshold"),
    ("bench_4k64_f32", 3840, 2160, 64, 60, "standard_4k", "f32", None),
    ("bench_4k256_u8", 3840, 2160, 256, 60, "standard_4k", "u8", None),
    ("bench_4k40_u8_120fps", 3840, 2160, 40, 120, "standard_4k", "u8", None),      # bench --fps 120 (31-tap filters), 40 frames: longer than the filter
)


def main():
    only = sys.argv[1:]
    for name, W, H, F, fps, disp, dtype, heat in CASES:
        if only and name not in only:
            continue
        t0 = time.time()
        tdt = torch.float32 if dtype == "f32" else torch.uint8
        t = torch.empty((1, 3, F, H, W), dtype=tdt)
        r = torch.empty((1, 3, F, H, W), dtype=tdt)
        cs_t = cs_r = 0
        for f in range(F):
            a, b = bench.synth_frame(f, H, W, "cpu")
            cs_t += int(a.to(torch.int64).sum())
            cs_r += int(b.to(torch.int64).sum())
            if dtype == "f32":
                a, b = a.float() / 255, b.float() / 255      # exactly what bench.ResidentClip stores
            t[0, :, f], r[0, :, f] = a, b
        met = pycvvdp.cvvdp(display_name=disp, device=torch.device("cpu"), quiet=True, heatmap=heat)
        with torch.no_grad():
            if dtype == "f32":
                jod, stats = met.predict(t, r, dim_order="BCFHW", frames_per_second=fps)
            else:
                jod, stats = met.predict(t.numpy(), r.numpy(), dim_order="BCFHW", frames_per_second=fps)
        extra = {}
        if heat is not None:
            hm = stats["heatmap"]                              # [1,3,F,H,W] fp16
            extra = dict(heatmap_mode=heat, heatmap_mean=np.float32(hm.float().mean().item()),
                         heatmap_ds=hm[0, :, :, ::16, ::16].numpy().astype(np.float16))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), width=W, height=H, frames=F, fps=fps, display=disp, dtype=dtype,
                            jod=np.float32(jod.item()), Q_per_ch=stats["Q_per_ch"], rho_band=stats["rho_band"],
                            checksum_test=np.int64(cs_t), checksum_ref=np.int64(cs_r), torch_version=torch.__version__,
                            reference_seconds=np.float32(time.time() - t0), **extra)
        print(name, float(jod), stats["Q_per_ch"].shape, f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
