"""Worker of tests/test_safe_loads.py: scores the fixed cases with whatever library colorvideovdp_amd._capi loads (the parent sets
CVVDP_DEV_KNOBS=1 CVVDP_LIB=<libcvvdp_hip_safe.so> for the compiler-managed build) and writes Q_per_ch, the level-1 / level-2 planes (small
cases: the planes; the full 4K block: two 64-bit checksums of their bits), the heat map of the fused heat kernels and the pooled features of
the fused feature kernels of every case to the .npz named on the command line.

Two occupancies: clips of 2-4 frames (a fraction of one round of workgroups: loads return at their fastest) and a
64-frame 4K block (every CU holds its full complement of workgroups for many rounds, three fused levels: loads return at their slowest --
the condition under which a wait count that is one too small would show)."""
import sys
import numpy as np
import torch

CASES = [
    # name, W, H, frames, fps, display
    ("aligned_4k", 3840, 2160, 3, 60, "standard_4k"),          # the bench clip's frame: 14 strips on k_band4s, 2 on k_band4f<4, 1>
    ("w_mod4_is_2", 1446, 333, 2, 60, "standard_hdr_pq"),      # partial-lane border kernel (k_band4f<4, 2>) beside the split kernel, odd height
    ("small_ragged", 250, 131, 4, 30, "standard_fhd"),         # RAGGED k_band4 on every strip of the unfused route
    ("full_4k64", 3840, 2160, 64, 60, "standard_4k"),          # the bench block: 5 376 + 768 workgroups at level 0, three fused levels
]
ROUTES = [("fused_split", 1, 0), ("fused_one_wave", 1, 1), ("unfused", 2, 0)]       # name, fuse_mode, band_layout
EXTRA_ROUTES = ["heat", "feat"]                                                      # k_band4s_heat / k_band4s_feat (+ their border kernels)
BIG = 16                                                                             # frames from which planes are compared by checksum


def clip(W, H, F, seed):
    if F >= BIG:                            # made on the GPU (the same torch, the same device in both processes: the same bits)
        dev = torch.device("cuda")
        y = torch.arange(H, device=dev, dtype=torch.float32).view(1, H, 1)
        x = torch.arange(W, device=dev, dtype=torch.float32).view(1, 1, W)
        c = torch.arange(3, device=dev, dtype=torch.float32).view(3, 1, 1)
        g = torch.Generator(device=dev)
        t = torch.empty((1, 3, F, H, W), dtype=torch.uint8, device=dev)
        r = torch.empty_like(t)
        for f in range(F):
            ref = 0.45 + 0.3 * torch.sin(2 * np.pi * (3.1 * x / W + f / 9.0) + c) * torch.cos(2 * np.pi * 2.3 * y / H)
            g.manual_seed(seed * 1009 + f)
            test = (ref + 0.05 * torch.randn((3, H, W), generator=g, device=dev)).clamp(0, 1)
            t[0, :, f] = torch.round(test * 255).to(torch.uint8)
            r[0, :, f] = torch.round(ref * 255).to(torch.uint8)
        return t, r
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W]
    ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (3.1 * x / W + f / 9.0) + c) * np.cos(2 * np.pi * 2.3 * y / H) for c in range(3)])
                    for f in range(F)], axis=1)[None]
    test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1)
    return torch.as_tensor(np.round(test * 255).astype(np.uint8)).cuda(), torch.as_tensor(np.round(ref * 255).astype(np.uint8)).cuda()


def bits_checksum(x):
    """Two 64-bit sums over the BITS of a device tensor (plain and position-weighted): equal for equal bits, and a single changed bit
    anywhere changes both."""
    x = x.contiguous()
    b = x.view(torch.int16 if x.element_size() == 2 else torch.int32).flatten()
    s0 = s1 = 0
    step = 1 << 27
    for i in range(0, b.numel(), step):
        part = b[i:i + step].to(torch.int64)
        w = (torch.arange(i, i + part.numel(), device=b.device, dtype=torch.int64) % 65521) + 1
        s0 += int(part.sum())
        s1 += int((part * w).sum())
    return np.array([s0 & 0x7FFFFFFFFFFFFFFF, s1 & 0x7FFFFFFFFFFFFFFF], dtype=np.int64)


class _ChecksumSink:
    """heat-map frames stay on the GPU; per piece a checksum of the fp16 bits"""
    wants_device = True

    def __init__(self, keep):
        self.keep, self.out = keep, []

    def __call__(self, first_frame, frames):
        assert frames.is_cuda
        self.out.append((first_frame, frames.clone() if self.keep else bits_checksum(frames)))


def run():
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import _capi
    from colorvideovdp_amd.video_source import video_source_array
    out = {"lib": np.array(_capi.LIB_PATH)}
    for name, W, H, F, fps, disp in CASES:
        t, r = clip(W, H, F, W + H)
        small = F < BIG
        for route, fuse_mode, layout in ROUTES:
            m = cv.cvvdp(display_name=disp)
            m.fuse_mode, m.band_layout = fuse_mode, layout
            jod, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=fps)
            key = f"{name}.{route}"
            out[key + ".jod"] = np.float64(float(jod))
            out[key + ".q"] = st["Q_per_ch"]
            out[key + ".fused_levels"] = np.int32(m.fused_levels)
            assert m.last_block_frames == F
            hh, ww = H, W
            for l in (1, 2):
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
                g = m.debug_buffer(_capi.BUF_GPYR, l)[:8 * F * hh * ww]
                out[f"{key}.g{l}"] = g.view(8, F, hh, ww).cpu().numpy().copy() if small else bits_checksum(g)
            del m
        if name == "small_ragged":
            continue                         # (not fusable with W % 2 == 1: the HEAT / FEAT instantiations of k_band4 use compiler-managed loads in both builds)
        # the fused heat-map kernels: threshold heat map, frames consumed on the GPU (pieces of a long temporal block)
        m = cv.cvvdp(display_name=disp, heatmap="threshold")
        m.fuse_mode = 1
        sink = _ChecksumSink(keep=small)
        jod, st = m.predict_video_source(video_source_array(t, r, fps, dim_order="BCFHW", display_photometry=m.display_photometry), heatmap_sink=sink)
        assert m.fused_levels >= 1
        out[f"{name}.heat.q"] = st["Q_per_ch"]
        out[f"{name}.heat.firsts"] = np.array([f for f, _ in sink.out], dtype=np.int32)
        out[f"{name}.heat.map"] = (torch.cat([x for _, x in sink.out], 2).cpu().numpy() if small else np.stack([x for _, x in sink.out]))
        del m, sink
        # the fused feature kernels: the ML heads' pooled statistics of every band
        m = cv.cvvdp(display_name=disp)
        m.fuse_mode = 1
        feats, _ = m.extract_features(video_source_array(t, r, fps, dim_order="BCFHW", display_photometry=m.display_photometry))
        assert m.fused_levels >= 1
        for b, f in enumerate(feats):
            out[f"{name}.feat.band{b}"] = f.cpu().numpy() if (small or f.numel() < (1 << 22)) else bits_checksum(f)
        out[f"{name}.feat.n_bands"] = np.int32(len(feats))
        del m, feats, t, r
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    np.savez(sys.argv[1], **run())
