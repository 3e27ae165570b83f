"""Worker of tests/test_safe_loads.py: scores the fixed cases with whatever library colorvideovdp_amd._capi loads (the parent sets
CVVDP_DEV_KNOBS=1 CVVDP_LIB=<libcvvdp_hip_safe.so> for the compiler-managed build) and writes Q_per_ch and the level-1 / level-2 planes
of every case to the .npz named on the command line."""
import sys
import numpy as np
import torch

CASES = [
    # name, W, H, frames, fps, display
    ("aligned_4k", 3840, 2160, 3, 60, "standard_4k"),          # the bench clip's frame: 14 strips on k_band4s, 2 on k_band4f<4, 1>
    ("w_mod4_is_2", 1446, 333, 2, 60, "standard_hdr_pq"),      # partial-lane border kernel (k_band4f<4, 2>) beside the split kernel, odd height
    ("small_ragged", 250, 131, 4, 30, "standard_fhd"),         # RAGGED k_band4 on every strip of the unfused route
]
ROUTES = [("fused_split", 1, 0), ("fused_one_wave", 1, 1), ("unfused", 2, 0)]       # name, fuse_mode, band_layout


def clip(W, H, F, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W]
    ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (3.1 * x / W + f / 9.0) + c) * np.cos(2 * np.pi * 2.3 * y / H) for c in range(3)])
                    for f in range(F)], axis=1)[None]
    test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1)
    return np.round(test * 255).astype(np.uint8), np.round(ref * 255).astype(np.uint8)


def run():
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import _capi
    out = {"lib": np.array(_capi.LIB_PATH)}
    for name, W, H, F, fps, disp in CASES:
        t, r = clip(W, H, F, W + H)
        t, r = torch.as_tensor(t).cuda(), torch.as_tensor(r).cuda()
        for route, fuse_mode, layout in ROUTES:
            m = cv.cvvdp(display_name=disp)
            m.fuse_mode, m.band_layout = fuse_mode, layout
            jod, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=fps)
            key = f"{name}.{route}"
            out[key + ".jod"] = np.float64(float(jod))
            out[key + ".q"] = st["Q_per_ch"]
            out[key + ".fused_levels"] = np.int32(m.fused_levels)
            hh, ww = H, W
            for l in (1, 2):
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
                out[f"{key}.g{l}"] = m.debug_buffer(_capi.BUF_GPYR, l)[:8 * F * hh * ww].view(8, F, hh, ww).cpu().numpy().copy()
            del m
    return out


if __name__ == "__main__":
    np.savez(sys.argv[1], **run())
