#!/usr/bin/env python3
"""GPU-box diagnostic: run every golden case through the HIP path and print per-stage errors against
the CPU oracle and the reference vectors.  Not a test (never fails); used to localise a mismatch in one
gpurun call.  Usage: python tools/gpu_diag.py [case ...]"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import colorvideovdp_amd as cv
from colorvideovdp_amd import _capi
from conftest import golden_cases, load_golden
from oracle import cvvdp_oracle as orc


def err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    i = np.unravel_index(np.argmax(d), d.shape) if d.size else ()
    return "max|d|=%.3e rel=%.3e at %s (got %.6g want %.6g)" % (d.max(), d.max() / (np.abs(b).max() + 1e-30), i, a[i], b[i])


def make_metric(meta, **kw):
    if "custom_photometry" in meta:
        ph = cv.vvdp_display_photo_eotf(**meta["custom_photometry"])
        ge = cv.vvdp_display_geometry(**meta["custom_geometry"])
        return cv.cvvdp(display_photometry=ph, display_geometry=ge, heatmap=meta["heatmap"], temp_padding=meta["temp_padding"], **kw)
    return cv.cvvdp(display_name=meta["display"], heatmap=meta["heatmap"], temp_padding=meta["temp_padding"], **kw)


def inputs(g):
    t, r = g["test"], g["ref"]
    if t.dtype == np.float16:
        return torch.tensor(t), torch.tensor(r)
    return t, r


def run_case(name, stages):
    g = load_golden(name)
    meta = g["meta"]
    t, r = inputs(g)
    m = make_metric(meta)
    m.debug_dump = stages
    t0 = time.time()
    jod, stats = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    torch.cuda.synchronize()
    print("== %s  (%.2fs)" % (name, time.time() - t0))
    print("   JOD got %s want %s  |d|=%.2e" % (jod.cpu().numpy(), g["jod"], np.abs(jod.cpu().numpy() - g["jod"]).max()))
    print("   Q_per_ch vs reference:", err(stats["Q_per_ch"], g["Q_per_ch"]))
    if "heatmap" in g:
        hm = stats["heatmap"].numpy().astype(np.float32)
        print("   heatmap vs reference:", err(hm, g["heatmap"].astype(np.float32)), " frac>2e-3: %.2e" % (np.abs(hm - g["heatmap"].astype(np.float32)) > 2e-3).mean())
    if not stages:
        return
    # per-stage comparison against the oracle's intermediates of the LAST block (block = whole clip here)
    o = orc.Oracle(display_name=meta.get("display"), heatmap=meta["heatmap"], temp_padding=meta["temp_padding"], keep=True,
                   photometry=meta.get("custom_photometry"), geometry=meta.get("custom_geometry"))
    ojod, ostats = o.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    print("   Q_per_ch vs oracle   :", err(stats["Q_per_ch"], ostats["Q_per_ch"]))
    d = o.dbg  # last frame only (oracle block = 1)
    B = d["R"].shape[0]
    npl = d["R"].shape[1]
    nfr = stats["Q_per_ch"].shape[2]
    L = len(d["gpyr"])
    items_cap = None
    for l in range(L):
        H, W = d["gpyr"][l].shape[-2:]
        buf = m.debug_buffer(_capi.BUF_GPYR, l).cpu().numpy()
        items_cap = buf.size // (npl * H * W)
        buf = buf.reshape(npl, items_cap, H, W)
        item0 = (min(nfr, items_cap // B) - 1) * B  # last frame of the last block
        # which frame is the last block's last frame?  with block = whole clip it is frame nfr-1
        got = buf[:, item0:item0 + B].transpose(1, 0, 2, 3)
        print("   gpyr[%d] %dx%d:" % (l, H, W), err(got, d["gpyr"][l][:, :, 0].numpy()))
    for l in range(L):
        H, W = d["D"][l].shape[-2:]
        nchn = d["D"][l].shape[1]
        buf = m.debug_buffer(_capi.BUF_DDUMP, l).cpu().numpy().reshape(4, items_cap, H, W)
        item0 = (min(nfr, items_cap // B) - 1) * B
        got = buf[:nchn, item0:item0 + B].transpose(1, 0, 2, 3)
        print("   D[%d] %dx%d:" % (l, H, W), err(got, d["D"][l][:, :, 0].numpy()))


def main():
    names = sys.argv[1:] or golden_cases()
    print("device:", torch.cuda.get_device_name(0))
    for n in names:
        try:
            run_case(n, stages=n in ("img_u8_64x96_fhd_thr", "vid_u8_72x128x12_60_fhd", "img_u8_256x256_fhd", "vid_u16_67x121x20_30_4k_sym"))
        except Exception:
            print("== %s FAILED" % n)
            traceback.print_exc()


if __name__ == "__main__":
    main()
