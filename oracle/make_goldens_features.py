#!/usr/bin/env python3
"""tests/golden/features.npz: the per-band pooled T/R/D statistics the reference's ML heads consume
(pycvvdp/cvvdp_ml_metric.py:77-107 cvvdp_feature_pooling, :302-390 process_block_of_frames), produced by the REAL reference
(container only) on committed golden inputs: a minimal subclass of cvvdp_ml_base (no network, random_init) whose
extract_features() is the reference's own.

    python oracle/make_goldens_features.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import pycvvdp
from pycvvdp.cvvdp_ml_metric import cvvdp_ml_base
from pycvvdp.video_source import video_source_array

G = os.path.join(HERE, "..", "tests", "golden")
CASES = ("vid_u8_135x240x18_60_fhd_raw", "img_u8_256x256_fhd")


class FeaturesOnly(cvvdp_ml_base):
    def get_nets_to_load(self):
        return []

    def do_pooling_and_jods(self, features):
        return torch.zeros(1)


def main():
    out = {}
    for k, case in enumerate(CASES):
        g = np.load(os.path.join(G, case + ".npz"), allow_pickle=False)
        meta = eval(str(g["meta"]))
        m = FeaturesOnly(random_init=True, display_name=meta["display"], device=torch.device("cpu"), quiet=True, temp_padding=meta["temp_padding"])
        vs = video_source_array(g["test"], g["ref"], meta["fps"], dim_order=meta["dim_order"], display_photometry=m.display_photometry)
        with torch.no_grad():
            feats, _ = m.extract_features(vs)
        out[f"case{k}"] = case
        out[f"case{k}_bands"] = len(feats)
        for bb, f in enumerate(feats):
            out[f"case{k}_band{bb}"] = f.numpy().astype(np.float32)          # [B, F, H', W', C, 6]
        print(case, [tuple(f.shape) for f in feats], "feature_size", int(np.ceil(m.pix_per_deg)))
    np.savez_compressed(os.path.join(G, "features.npz"), **out)


if __name__ == "__main__":
    main()
