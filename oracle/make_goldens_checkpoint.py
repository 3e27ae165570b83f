#!/usr/bin/env python3
"""tests/golden/checkpoint.npz: the REAL reference's `update_from_checkpoint` (cvvdp_metric.py:231-243) and `save_to_config`
(:1129-1154) on a synthetic calibration checkpoint (container only).

A `state_dict` with `params.*` entries that move every kind of trained parameter (scalars, per-channel lists, the temporal filters'
sigma, the 4x4 cross-channel weights) plus entries without the prefix (ignored by the reference) is loaded into the reference
metric; stored: the state dict itself (so the test rebuilds the very same checkpoint file), JOD / Q_per_ch of a committed video and
image case before and after, and the JSON text `save_to_config` writes before and after.

    python oracle/make_goldens_checkpoint.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import pycvvdp

G = os.path.join(HERE, "..", "tests", "golden")
CPU = torch.device("cpu")
VIDEO, IMAGE = "vid_u8_72x128x12_60_fhd", "img_u8_256x256_fhd"

STATE = {
    "params.mask_c": torch.tensor(-0.55, dtype=torch.float32),
    "params.mask_p": torch.tensor(2.1, dtype=torch.float32),
    "params.mask_q": torch.tensor([1.2, 3.0, 3.5, 3.9], dtype=torch.float32),
    "params.ch_trans_w": torch.tensor(0.9, dtype=torch.float32),
    "params.ch_chrom_w": torch.tensor(1.1, dtype=torch.float32),
    "params.baseband_weight": torch.tensor([0.01, 1.5, 4.0, 20.0], dtype=torch.float32),
    "params.d_max": torch.tensor(2.4, dtype=torch.float32),
    "params.jod_a": torch.tensor(0.05, dtype=torch.float32),
    "params.jod_exp": torch.tensor(0.9, dtype=torch.float32),
    "params.sensitivity_correction": torch.tensor(-1.0, dtype=torch.float32),
    "params.image_int": torch.tensor(0.6, dtype=torch.float32),
    "params.sigma_tf": torch.tensor([5.0, 13.0, 7.0, 0.13], dtype=torch.float32),
    "params.beta_tf": torch.tensor([1.3, 1.1, 0.95, 0.19], dtype=torch.float32),
    "params.xcm_weights": torch.linspace(-3.0, 2.0, 16, dtype=torch.float32),
    "head.weight": torch.ones(3, dtype=torch.float32),           # no 'params.' prefix: the reference skips it
    "epoch_marker": torch.tensor(7.0),
}


def score(m, case):
    g = np.load(os.path.join(G, case + ".npz"), allow_pickle=False)
    meta = eval(str(g["meta"]))
    m.set_display_model(meta["display"])
    m.temp_padding = meta["temp_padding"]
    with torch.no_grad():
        jod, stats = m.predict(g["test"], g["ref"], dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    return np.float32(jod.item()), stats["Q_per_ch"].astype(np.float32)


def saved(m, comment):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "out.json")
        m.save_to_config(p, comment)
        return open(p, encoding="utf-8").read()


def main():
    out = {"vid_case": VIDEO, "img_case": IMAGE, "state_keys": np.array(list(STATE))}
    for k, v in STATE.items():
        out["state/" + k] = v.numpy()
    m = pycvvdp.cvvdp(display_name="standard_fhd", device=CPU, quiet=True)
    out["saved_before"] = saved(m, "as shipped")
    for tag, case in (("vid", VIDEO), ("img", IMAGE)):
        out[f"{tag}_jod_before"], out[f"{tag}_q_before"] = score(m, case)
    with tempfile.TemporaryDirectory() as d:
        ck = os.path.join(d, "ck.ckpt")
        torch.save({"state_dict": STATE, "epoch": 3}, ck)
        m.update_from_checkpoint(ck)
    out["saved_after"] = saved(m, "after the checkpoint")
    for tag, case in (("vid", VIDEO), ("img", IMAGE)):
        out[f"{tag}_jod_after"], out[f"{tag}_q_after"] = score(m, case)
    np.savez_compressed(os.path.join(G, "checkpoint.npz"), **out)
    for tag in ("vid", "img"):
        print(tag, out[f"{tag}_jod_before"], "->", out[f"{tag}_jod_after"])


if __name__ == "__main__":
    main()
