#!/usr/bin/env python3
"""tests/golden/outputs.npz: what the REAL reference's host-side output functions produce on two committed golden cases
(container only):

  * export_distogram (cvvdp_metric.py:1158-1218): the per-channel arrays it hands to matplotlib's imshow, captured by
    wrapping Axes.imshow (so no line of the reference is restated), for jod_max = None and 10, video and image;
  * write_features_to_json (:1112-1127): the JSON text;
  * loss() (:294-298) on the video case;
  * a source with is_temporally_filtered = True (:470-488): the reference is fed the temporally filtered 'DKLd65_trans'
    frames it computed itself for the video case (captured from read_block_of_frames) -- Q_per_ch / JOD of that run.

    python oracle/make_goldens_outputs.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")

import matplotlib
matplotlib.use("Agg")
import matplotlib.axes
import numpy as np
import torch

import pycvvdp

G = os.path.join(HERE, "..", "tests", "golden")
CPU = torch.device("cpu")
VIDEO, IMAGE = "vid_u8_72x128x12_60_fhd", "img_u8_256x256_fhd"


def run(case, **kw):
    g = np.load(os.path.join(G, case + ".npz"), allow_pickle=False)
    meta = eval(str(g["meta"]))
    m = pycvvdp.cvvdp(display_name=meta["display"], device=CPU, quiet=True, temp_padding=meta["temp_padding"], **kw)
    with torch.no_grad():
        jod, stats = m.predict(g["test"], g["ref"], dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    return m, jod, stats, g, meta


def distogram_arrays(m, stats, jod_max):
    shown = []
    orig = matplotlib.axes.Axes.imshow

    def spy(self, X, *a, **k):
        shown.append(np.array(X, dtype=np.float32))
        return orig(self, X, *a, **k)

    matplotlib.axes.Axes.imshow = spy
    try:
        with tempfile.TemporaryDirectory() as d:
            # (a fresh copy: on the CPU the reference scales stats['Q_per_ch'] in place -- torch.as_tensor shares the array)
            m.export_distogram({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in stats.items()}, os.path.join(d, "d.png"), jod_max=jod_max)
            size = os.path.getsize(os.path.join(d, "d.png"))
    finally:
        matplotlib.axes.Axes.imshow = orig
    assert size > 0
    return np.stack(shown)


def main():
    out = {}
    for tag, case in (("vid", VIDEO), ("img", IMAGE)):
        m, jod, stats, g, meta = run(case)
        stats = {k: v for k, v in stats.items() if k != "heatmap"}
        out[f"{tag}_case"] = case
        out[f"{tag}_disto_auto"] = distogram_arrays(m, stats, None)
        out[f"{tag}_disto_10"] = distogram_arrays(m, stats, 10)
        with tempfile.TemporaryDirectory() as d:
            m.write_features_to_json(stats, os.path.join(d, "f.json"))
            out[f"{tag}_features_json"] = open(os.path.join(d, "f.json"), encoding="utf-8").read()
        if tag == "vid":
            with torch.no_grad():
                out["vid_loss"] = np.float32(m.loss(g["test"], g["ref"], dim_order=meta["dim_order"], frames_per_second=meta["fps"]).item())
            out["vid_jod"] = np.float32(jod.item())

    # ---- pre-filtered source: capture the reference's own temporally filtered planes, then feed them back
    m, jod, stats, g, meta = run(VIDEO)
    planes = []
    orig_read = m.read_block_of_frames

    def tap(*a, **k):
        R = orig_read(*a, **k)
        planes.append(R.clone())
        return R

    m.read_block_of_frames = tap
    with torch.no_grad():
        m.predict(g["test"], g["ref"], dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    R = torch.cat(planes, dim=2)                     # [1, 8, F, H, W]: T0, R0, T1, R1, ...
    T_filt, R_filt = R[:, 0::2], R[:, 1::2]          # [1, 4, F, H, W] each

    class Prefiltered:
        is_temporally_filtered = True

        def get_video_size(self):
            return (R.shape[3], R.shape[4], R.shape[2])

        def get_frames_per_second(self):
            return meta["fps"]

        def get_batch_size(self):
            return 1

        def get_test_frame(self, f, device, colorspace):
            assert colorspace == "DKLd65_trans"
            return T_filt[:, :, f:f + 1].to(device)

        def get_reference_frame(self, f, device, colorspace):
            assert colorspace == "DKLd65_trans"
            return R_filt[:, :, f:f + 1].to(device)

    m2 = pycvvdp.cvvdp(display_name=meta["display"], device=CPU, quiet=True, temp_padding=meta["temp_padding"])
    with torch.no_grad():
        jod2, stats2 = m2.predict_video_source(Prefiltered())
    assert abs(float(jod2) - float(jod)) < 1e-6, (float(jod2), float(jod))
    out["prefiltered_test"] = T_filt[0].numpy().astype(np.float32)      # [4, F, H, W]
    out["prefiltered_ref"] = R_filt[0].numpy().astype(np.float32)
    out["prefiltered_jod"] = np.float32(jod2.item())
    out["prefiltered_Q_per_ch"] = stats2["Q_per_ch"]
    np.savez_compressed(os.path.join(G, "outputs.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") and getattr(v, "ndim", 0) else (v if not isinstance(v, str) else len(v))) for k, v in out.items()})


if __name__ == "__main__":
    main()
