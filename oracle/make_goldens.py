#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (container only).

Imports /root/reference/pycvvdp with the import shims in oracle/ref_shims (torchvision,
ffmpeg, imageio are absent from this image) on CPU, feeds it small
seeded inputs and stores inputs + outputs (+ selected intermediates).  The fixtures are
data only; this script is the committed recipe that made them.

    python oracle/make_goldens.py            # writes tests/golden/
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import pycvvdp
from pycvvdp.display_model import vvdp_display_photo_eotf, vvdp_display_geometry

OUT = os.path.join(HERE, "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
CPU = torch.device("cpu")


def pattern(rng, F, H, W, C=3, amp=0.18):
    """Smooth moving plaid + low-pass noise in [0,1], shape [F,H,W,C] float64."""
    y, x = np.mgrid[0:H, 0:W]
    out = np.empty((F, H, W, C))
    base = rng.random((H // 4 + 2, W // 4 + 2, C))
    base = np.kron(base, np.ones((4, 4, 1)))[:H, :W]
    for f in range(F):
        for c in range(C):
            out[f, :, :, c] = 0.5 + 0.25 * np.sin(2 * np.pi * (3 * x / W + f / 30.0)) * np.cos(2 * np.pi * 2 * y / H + 0.7 * c) \
                + amp * (np.roll(base[:, :, c], f, axis=1) - 0.5)
    return np.clip(out, 0, 1)


def distort(rng, ref, sigma=0.03):
    t = ref + sigma * rng.standard_normal(ref.shape)
    t[..., ref.shape[-2] // 2:, :] = 0.5 * (t[..., ref.shape[-2] // 2:, :] + np.roll(t, 1, axis=-2)[..., ref.shape[-2] // 2:, :])
    for f in range(0, t.shape[0], 4):
        t[f] *= 0.95
    return np.clip(t, 0, 1)


def quant(a, dtype):
    if dtype == "u8":
        return np.round(a * 255).astype(np.uint8)
    if dtype == "u16":
        return np.round(a * 65535).astype(np.uint16)
    if dtype == "f16":
        return a.astype(np.float16)
    return a.astype(np.float32)


class Tap:
    """Records intermediates of the last processed block of the reference metric."""

    def __init__(self, m):
        self.m = m
        self.rec = {}
        orig_pb = m.process_block_of_frames
        orig_mask = m.apply_masking_model
        orig_sens = m.csf.sensitivity
        tap = self

        def pb(R, vid_sz, temp_ch, lpyr, is_image):
            tap.rec = {"R": R.clone(), "S": [], "D": []}
            g = lpyr.gaussian_pyramid_dec(R, lpyr.height + 1, 0.4)
            tap.rec["gpyr"] = [x.clone() for x in g]
            c, l = lpyr.decompose(R)
            tap.rec["contrast"] = [x.clone() for x in c]
            tap.rec["logL"] = [x.clone() for x in l]
            tap._S = []
            out = orig_pb(R, vid_sz, temp_ch, lpyr, is_image)
            return out

        def mask(T, R, S):
            D = orig_mask(T, R, S)
            tap.rec["S"].append(S.clone())
            tap.rec["D"].append(D.clone())
            return D

        m.process_block_of_frames = pb
        m.apply_masking_model = mask


def run_case(name, test, ref, dim_order, fps, display="standard_fhd", heatmap=None, temp_padding="replicate",
             photometry=None, geometry=None, intermediates=False, extra=None):
    kw = dict(display_name=display, heatmap=heatmap, quiet=True, device=CPU, temp_padding=temp_padding)
    if photometry is not None:
        kw["display_photometry"] = photometry
    if geometry is not None:
        kw["display_geometry"] = geometry
    m = pycvvdp.cvvdp(**kw)
    tap = Tap(m) if intermediates else None
    with torch.no_grad():
        q, stats = m.predict(test, ref, dim_order=dim_order, frames_per_second=fps)
    d = {"test": test.numpy() if torch.is_tensor(test) else test, "ref": ref.numpy() if torch.is_tensor(ref) else ref,
         "jod": q.numpy().astype(np.float32), "Q_per_ch": stats["Q_per_ch"], "rho_band": np.asarray(stats["rho_band"]),
         "ppd": np.float64(m.pix_per_deg), "info": np.array(m.get_info_string())}
    meta = dict(dim_order=dim_order, fps=fps, display=display, heatmap=heatmap or "none", temp_padding=temp_padding)
    if extra:
        meta.update(extra)
    d["meta"] = np.array(repr(meta))
    if heatmap not in (None, "none"):
        d["heatmap"] = stats["heatmap"].numpy()
    if fps > 0:
        d["taps"] = np.stack([f.numpy() for f in m.F])
    if tap is not None:
        r = tap.rec
        d["i_R"] = r["R"].numpy()
        for i, x in enumerate(r["gpyr"]):
            d["i_g%d" % i] = x.numpy()
        for i, x in enumerate(r["contrast"]):
            d["i_c%d" % i] = x.numpy()
        for i, x in enumerate(r["logL"]):
            d["i_l%d" % i] = x.numpy()
        for i, x in enumerate(r["S"]):
            d["i_S%d" % i] = x.numpy()
        for i, x in enumerate(r["D"]):
            d["i_D%d" % i] = x.numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-28s JOD=%s  Q_per_ch%s  %.0f KB" % (name, np.array2string(d["jod"], precision=5), d["Q_per_ch"].shape, os.path.getsize(path) / 1024))


def main():
    rng = np.random.default_rng(20240928)

    # ---- setup vectors: temporal taps, DKL matrices, CSF rows, band frequencies ----
    m = pycvvdp.cvvdp(display_name="standard_4k", quiet=True, device=CPU)
    setup = {}
    for fps in (24, 25, 30, 50, 60, 120):
        F, _ = m.get_temporal_filters(fps)
        setup["taps_%d" % fps] = np.stack([f.numpy() for f in F])
    for disp in ("standard_4k", "standard_hdr_pq"):
        dm = pycvvdp.vvdp_display_photometry.load(disp, [])
        x = torch.ones((1, 3, 1, 2, 2)) * 0.5
        dm.source_2_target_colorspace(x, "DKLd65")
        M = torch.as_tensor(pycvvdp.display_model.LMS2006_to_DKLd65, dtype=torch.float32) @ \
            torch.as_tensor(pycvvdp.display_model.XYZ_to_LMS2006, dtype=torch.float32) @ dm.rgb2xyz
        setup["dkl_" + disp] = M.numpy()
        setup["black_" + disp] = np.array(dm.get_black_level())
    rhos = [37.70, 12.17, 3.0424, 0.38029, 0.1, 0.05, 80.0]
    rows = []
    for rho in rhos:
        for oo, cc in ((0, 0), (0, 1), (0, 2), (5, 0)):
            m.csf.sensitivity(rho, oo, torch.zeros(1), cc, None)
            key = "o%d_c%d_rho%s" % (0 if oo == 0 else 1, cc, rho)
            rows.append(m.csf.logS_rho[key].numpy())
    setup["csf_rhos"] = np.array(rhos)
    setup["csf_rows"] = np.stack(rows).reshape(len(rhos), 4, -1)
    q = torch.linspace(-3.0, 4.5, 97)
    setup["csf_query"] = q.numpy()
    setup["csf_S"] = np.stack([m.csf.sensitivity(3.0424, oo, q, cc, None).numpy() for oo, cc in ((0, 0), (0, 1), (0, 2), (5, 0))])
    sizes = [(1920, 1080, "standard_fhd"), (3840, 2160, "standard_4k"), (7680, 4320, "standard_hdr_pq"), (256, 256, "standard_fhd"),
             (97, 131, "standard_4k"), (240, 135, "standard_fhd"), (1280, 720, "sdr_fhd_24"), (1440, 1600, "standard_hmd"), (2532, 1170, "iphone_12_pro")]
    bf = []
    for W, H, disp in sizes:
        g = pycvvdp.vvdp_display_geometry.load(disp)
        lp = pycvvdp.lpyr_dec.lpyr_dec(W, H, g.get_ppd(), CPU)
        bf.append(np.concatenate([[W, H, g.get_ppd(), lp.get_band_count()], lp.get_freqs(), np.zeros(12 - len(lp.get_freqs()))]))
    setup["band_sizes"] = np.stack(bf)
    setup["band_displays"] = np.array([s[2] for s in sizes])
    # photometry forward on a ramp for every EOTF
    ramp = torch.linspace(-0.1, 1.1, 61).view(1, 1, 1, 1, -1).repeat(1, 3, 1, 1, 1) * torch.tensor([1.0, 0.9, 0.8]).view(1, 3, 1, 1, 1)
    for disp in ("standard_4k", "standard_hdr_pq", "standard_hdr_hlg", "standard_hdr_linear", "standard_phone"):
        dm = pycvvdp.vvdp_display_photometry.load(disp, [])
        inp = ramp * (1000.0 if disp == "standard_hdr_linear" else 1.0)
        setup["fwd_" + disp] = dm.source_2_target_colorspace(inp, "DKLd65").numpy()
    dm = vvdp_display_photo_eotf(Y_peak=120, contrast=800, source_colorspace="Adobe RGB (1998)", E_ambient=80, exposure=1)
    setup["fwd_gamma22"] = dm.source_2_target_colorspace(ramp, "DKLd65").numpy()
    dm = vvdp_display_photo_eotf(Y_peak=300, contrast=2000, source_colorspace="sRGB", E_ambient=10, exposure=0.7)
    setup["fwd_srgb_exp07"] = dm.source_2_target_colorspace(ramp, "DKLd65").numpy()
    setup["fwd_ramp"] = ramp.numpy()
    np.savez_compressed(os.path.join(OUT, "setup.npz"), **setup)
    print("setup.npz written")

    # ---- images ----
    r = pattern(rng, 1, 64, 96)[0]
    t = distort(rng, r[None])[0]
    run_case("img_u8_64x96_fhd_thr", quant(t, "u8"), quant(r, "u8"), "HWC", 0, "standard_fhd", heatmap="threshold", intermediates=True)

    r = pattern(rng, 1, 97, 131)[0]
    t = distort(rng, r[None], 0.05)[0]
    run_case("img_f32_97x131_4k_supra", quant(t, "f32"), quant(r, "f32"), "HWC", 0, "standard_4k", heatmap="supra-threshold")

    r = pattern(rng, 1, 40, 56)[0]
    t2 = np.stack([distort(rng, r[None], s)[0] for s in (0.02, 0.08)])  # [B,H,W,C]
    run_case("img_f16_batch2_40x56", torch.tensor(quant(t2, "f16")), torch.tensor(quant(r[None], "f16")), "BHWC", 0, "standard_fhd")

    r = pattern(rng, 1, 256, 256)[0]
    t = distort(rng, r[None], 0.04)[0]
    run_case("img_u8_256x256_fhd", quant(t, "u8"), quant(r, "u8"), "HWC", 0, "standard_fhd")

    r = pattern(rng, 1, 48, 80, C=1)[0, :, :, 0]
    t = distort(rng, r[None, :, :, None], 0.04)[0, :, :, 0]
    run_case("img_u16_lum_48x80", quant(t, "u16"), quant(r, "u16"), "HW", 0, "standard_4k")

    # linear HDR image (absolute cd/m^2)
    r = pattern(rng, 1, 50, 70)[0]
    t = distort(rng, r[None], 0.03)[0]
    run_case("img_f32_linear_50x70", (t ** 2.2 * 800 + 0.01).astype(np.float32), (r ** 2.2 * 800 + 0.01).astype(np.float32), "HWC", 0, "standard_hdr_linear")

    r = pattern(rng, 1, 50, 70)[0]
    t = distort(rng, r[None], 0.03)[0]
    run_case("img_u16_hlg_50x70", quant(t, "u16"), quant(r, "u16"), "HWC", 0, "standard_hdr_hlg")

    ph = vvdp_display_photo_eotf(Y_peak=120, contrast=800, source_colorspace="Adobe RGB (1998)", E_ambient=80)
    ge = vvdp_display_geometry((1920, 1200), distance_m=0.5, diagonal_size_inches=24)
    run_case("img_u8_gamma22_custom", quant(t, "u8"), quant(r, "u8"), "HWC", 0, photometry=ph, geometry=ge,
             extra=dict(custom_photometry=dict(Y_peak=120, contrast=800, source_colorspace="Adobe RGB (1998)", E_ambient=80),
                        custom_geometry=dict(resolution=(1920, 1200), distance_m=0.5, diagonal_size_inches=24)))

    # ---- videos ----
    r = pattern(rng, 12, 72, 128)
    t = distort(rng, r)
    run_case("vid_u8_72x128x12_60_fhd", quant(t, "u8"), quant(r, "u8"), "FHWC", 60, "standard_fhd", intermediates=True)

    r = pattern(rng, 20, 67, 121)
    t = distort(rng, r)
    run_case("vid_u16_67x121x20_30_4k_sym", quant(t, "u16"), quant(r, "u16"), "FHWC", 30, "standard_4k", temp_padding="symmetric")

    r = pattern(rng, 9, 36, 64)
    t = distort(rng, r)
    run_case("vid_u8_36x64x9_60_sym_short", quant(t, "u8"), quant(r, "u8"), "FHWC", 60, "standard_fhd", temp_padding="symmetric")

    r = pattern(rng, 18, 135, 240)
    t = distort(rng, r)
    run_case("vid_u8_135x240x18_60_fhd_raw", quant(t, "u8"), quant(r, "u8"), "FHWC", 60, "standard_fhd", heatmap="raw")

    r = pattern(rng, 6, 64, 80) * 0.65 + 0.10
    t = distort(rng, r, 0.01)
    run_case("vid_f32_64x80x6_60_pq_supra", quant(t.transpose(3, 0, 1, 2), "f32"), quant(r.transpose(3, 0, 1, 2), "f32"), "CFHW", 60,
             "standard_hdr_pq", heatmap="supra-threshold")

    r = pattern(rng, 5, 60, 90, C=1)[..., 0]
    t = distort(rng, r[..., None])[..., 0]
    run_case("vid_u8_lum_60x90x5_24", quant(t, "u8"), quant(r, "u8"), "FHW", 24, "standard_4k")

    r = pattern(rng, 3, 40, 48)  # exactly 3 frames + heatmap: Q3 does not trigger at block=1 (CPU), kept as a regression pin
    t = distort(rng, r)
    run_case("vid_u8_40x48x3_120_thr", quant(t, "u8"), quant(r, "u8"), "FHWC", 120, "standard_fhd", heatmap="threshold")

    # identical clips: Q_per_ch must be exactly 0 everywhere
    r = pattern(rng, 4, 32, 48)
    run_case("vid_u8_identical", quant(r, "u8"), quant(r, "u8"), "FHWC", 60, "standard_fhd")


if __name__ == "__main__":
    main()
