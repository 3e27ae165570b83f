"""CPU oracle for the ColorVideoVDP per-frame hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch torch-CPU restatement of the reference algorithm
(gfxdisp/ColorVideoVDP v0.5.6, `pycvvdp/`), kept op-for-op close to the reference so
that (a) it can be the checker for the HIP kernels and (b) it is a fair "port" CPU
baseline (same conv2d-based structure, block = 1 frame).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it; the product
package `colorvideovdp_amd` never does.

Parity status: PINNED.  `oracle/make_goldens.py` imports the real reference (with the
import shims in `oracle/ref_shims/`, this container only) and stores its inputs/outputs
under `tests/golden/`; `tests/test_oracle_vs_golden.py` holds this file to those vectors.
Caveat: the 13x13 blur comes from un-vendored torchvision; the shim
restates torchvision's published algorithm, so parity is pinned against "reference code +
shim", not against a torchvision binary.

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
import json
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "colorvideovdp_amd", "data", "vvdp_data.json")

# pycvvdp/display_model.py:17-25
_XYZ_TO_LMS2006 = ((0.187596268556126, 0.585168649077728, -0.026384263306304),
                   (-0.133397430663221, 0.405505777260049, 0.034502127690364),
                   (0.000244379021663, -0.000542995890619, 0.019406849066323))
_LMS2006_TO_DKLD65 = ((1.0, 1.0, 0.0),
                      (1.0, -2.311130179947035, 0.0),
                      (-1.0, -1.0, 50.977571328718781))


def load_bundle(path=None):
    with open(path or _DATA) as f:
        return json.load(f)


# --------------------------------------------------------------------------------------
# Display model (pycvvdp/display_model.py)
# --------------------------------------------------------------------------------------
class Display:
    """Photometry + geometry of one entry of display_models.json.

    display_model.py:156-200 (photometry load), :588-626 (geometry load), :503-526 (ppd).
    """

    def __init__(self, name=None, bundle=None, photometry=None, geometry=None):
        """`photometry`/`geometry` dicts mirror the ctor kwargs of vvdp_display_photo_eotf
        (display_model.py:301) and vvdp_display_geometry (:441) for custom displays."""
        bundle = bundle or load_bundle()
        self.name = name
        m = bundle["display_models"][name] if name is not None else None
        if photometry is None:
            self.Y_peak = m["max_luminance"]
            cs = m.get("colorspace", "sRGB")
            if "min_luminance" in m:
                self.contrast = self.Y_peak / m["min_luminance"]
            else:
                self.contrast = m.get("contrast", 500)
            self.E_ambient = m.get("E_ambient", 0)
            self.k_refl = m.get("k_refl", 0.005)
            self.exposure = m.get("exposure", 1)
        else:
            self.Y_peak = photometry["Y_peak"]
            self.contrast = photometry.get("contrast", 1000)
            cs = photometry.get("source_colorspace", "sRGB")
            self.E_ambient = photometry.get("E_ambient", 0)
            self.k_refl = photometry.get("k_refl", 0.005)
            self.exposure = photometry.get("exposure", 1)
        csd = bundle["color_spaces"][cs]
        self.EOTF = csd["EOTF"]
        if photometry is not None and photometry.get("EOTF") is not None:
            self.EOTF = photometry["EOTF"]
        self.rgb2xyz = [csd["RGB2X"], csd["RGB2Y"], csd["RGB2Z"]] if "RGB2X" in csd else None
        if geometry is None:
            W, H = m["resolution"]
            g = dict(ppd=m.get("pixels_per_degree"), fov_diagonal=m.get("fov_diagonal"))
            if "diagonal_size_meters" in m:
                g["diagonal_size_inches"] = m["diagonal_size_meters"] / 0.0254
            else:
                g["diagonal_size_inches"] = m.get("diagonal_size_inches")
            if "viewing_distance_meters" in m:
                g["distance_m"] = m["viewing_distance_meters"]
            elif "viewing_distance_inches" in m:
                g["distance_m"] = m["viewing_distance_inches"] * 0.0254
        else:
            g = dict(geometry)
            W, H = g.pop("resolution")
        self.ppd = self._ppd(W, H, **g)

    @staticmethod
    def _ppd(W, H, ppd=None, distance_m=None, fov_diagonal=None, diagonal_size_inches=None):
        # display_model.py:441-526 (subset reachable from display_models.json)
        if ppd is not None:
            return ppd
        ar = W / H
        size_m = None
        if diagonal_size_inches is not None:
            h_mm = math.sqrt((diagonal_size_inches * 25.4) ** 2 / (1 + ar ** 2))
            size_m = (ar * h_mm / 1000, h_mm / 1000)
        if distance_m is None:
            distance_m = 3
        if fov_diagonal is not None:
            dpx = math.sqrt(W ** 2 + H ** 2) / (2.0 * math.tan(math.radians(fov_diagonal * 0.5)))
            h_deg = math.degrees(math.atan(H / 2 / dpx)) * 2
            h_m = 2 * math.tan(math.radians(h_deg / 2)) * distance_m
            size_m = (h_m * ar, h_m)
        pix_deg = 2 * math.degrees(math.atan(0.5 * size_m[0] / W / distance_m))
        return 1 / pix_deg

    def black_level(self):
        # display_model.py:372-376
        return self.Y_peak / self.contrast, self.E_ambient / math.pi * self.k_refl

    def forward(self, V):
        """Display-encoded [B,C,1,H,W] in [0,1] -> absolute linear.  display_model.py:333-365."""
        if self.EOTF != "linear" and bool(((V > 1).any() or (V < 0).any())):
            V = V.clamp(0.0, 1.0)
        Yb, Yr = self.black_level()
        e = self.EOTF
        if e == "sRGB":
            lin = torch.where(V > 0.04045, ((V + 0.055) / 1.055) ** 2.4, V / 12.92)  # :78-80
            if self.exposure == 1:
                return (self.Y_peak - Yb) * lin + Yb + Yr
            return (self.Y_peak - Yb) * (lin * self.exposure).clip(0.0, 1.0) + Yb + Yr
        if e == "PQ":
            # :58-70
            n, m_, c1, c2, c3 = 0.15930175781250000, 78.843750000000000, 0.83593750000000000, 18.851562500000000, 18.687500000000000
            t = torch.pow(V, 1 / m_)
            lin = 10000 * torch.pow((t - c1).clamp(min=0) / (c2 - c3 * t), 1 / n)
            return (lin * self.exposure).clip(0.005, self.Y_peak) + Yb + Yr
        if e == "linear":
            return (V * self.exposure).clip(max(0.005, Yb), self.Y_peak) + Yr
        if e == "HLG":
            # :89-111, :350-359
            gamma = 1.2
            if self.Y_peak > 1000:
                gamma = 1.2 + 0.42 * math.log10(self.Y_peak / 1000) - 0.07623 * math.log10(self.E_ambient / 5)
            a = 0.17883277
            b = 1 - 4 * a
            c = 0.5 - a * math.log(4 * a)
            s = torch.where(V <= 0.5, torch.pow(V, 2) / 3.0, (torch.exp((V - c) / a) + b) / 12.0)
            Ys = 0.2627 * s[:, 0] + 0.6780 * s[:, 1] + 0.0593 * s[:, 2]
            lin = (Ys ** (gamma - 1)).unsqueeze(1) * s
            if self.exposure == 1:
                return (self.Y_peak - Yb) * lin + Yb + Yr
            return (self.Y_peak - Yb) * (lin * self.exposure).clip(0.0, 1.0) + Yb + Yr
        if e[0].isnumeric():
            g = float(e)
            return (self.Y_peak - Yb) * (torch.pow(V, g) * self.exposure).clip(0.0, 1.0) + Yb + Yr
        raise RuntimeError("Unknown EOTF " + e)

    def dkl_matrix(self):
        """fp32, left-to-right product.  display_model.py:255-256."""
        r = torch.tensor(self.rgb2xyz, dtype=torch.float32)
        return torch.as_tensor(_LMS2006_TO_DKLD65, dtype=torch.float32) @ torch.as_tensor(_XYZ_TO_LMS2006, dtype=torch.float32) @ r

    def to_dkl(self, V):
        """source_2_target_colorspace(V,'DKLd65'): display_model.py:206-239,266-269."""
        L = self.forward(V)
        if V.shape[-4] != 3:
            return L  # luminance-only content, Q6
        M = self.dkl_matrix()
        out = torch.empty_like(L)
        for c in range(3):
            out[:, c:c + 1] = torch.sum(L * M[c, :].view(1, 3, 1, 1, 1), dim=-4, keepdim=True)
        return out


# --------------------------------------------------------------------------------------
# Frame supply from arrays (pycvvdp/video_source.py:120-162, 243-346)
# --------------------------------------------------------------------------------------
def to_bcfhw(a, dim_order):
    if isinstance(a, np.ndarray):
        if a.dtype == np.uint16:
            a = a.astype(np.int16)
        a = torch.tensor(a)
    order = dim_order.upper()
    assert len(order) == a.dim()
    perm = [order.index(ch) for ch in "BCFHW" if ch in order]
    a = a.permute(perm)
    shape = [a.shape[[ch for ch in "BCFHW" if ch in order].index(ch)] if ch in order else 1 for ch in "BCFHW"]
    return a.reshape(shape)


def fetch_frame(arr, f):
    """video_source.py:320-340: one frame as fp32 in [0,1] (or linear)."""
    x = arr[:, :, f:f + 1]
    if x.dtype is torch.float32:
        return x
    if x.dtype is torch.float16:
        return x.to(torch.float32)
    if x.dtype is torch.int16:
        return (x.to(torch.int32) & 0xFFFF).to(torch.float32) / 65535
    if x.dtype is torch.uint8:
        return x.to(torch.float32) / 255
    raise RuntimeError("unsupported dtype")


# --------------------------------------------------------------------------------------
# Pyramid (pycvvdp/lpyr_dec.py)
# --------------------------------------------------------------------------------------
def band_frequencies(W, H, ppd):
    """lpyr_dec.py:18-42.  Returns (height, band_freqs[height+1])."""
    max_levels = int(np.floor(np.log2(min(H, W)))) - 1
    bands = np.concatenate([[1.0], np.power(2.0, -np.arange(0.0, 14.0)) * 0.3228], 0) * ppd / 2.0
    bad = np.nonzero(bands <= 0.2)[0]
    max_band = max_levels if bad.size == 0 else bad[0]
    height = int(np.clip(max_band + 1, 0, max_levels))
    freqs = np.array([1.0] + [0.3228 * 2.0 ** (-f) for f in range(height)]) * ppd / 2.0
    return height, freqs


def _k5(x):
    a = 0.4
    return torch.tensor([0.25 - a / 2.0, 0.25, a, 0.25, 0.25 - a / 2.0], dtype=x.dtype)  # lpyr_dec.py:179


def pyr_reduce(x):
    """lpyr_dec.py:186-211, including the row-parity test for the column edge (Q1, :206)."""
    K = _k5(x)
    H, W = x.shape[-2], x.shape[-1]
    lead = x.shape[:-2]
    ya = F.conv2d(x.reshape(-1, 1, H, W), K.view(1, 1, 5, 1), stride=(2, 1), padding=(2, 0)).view(lead + (-1, W))
    ya[..., 0, :] += x[..., 0, :] * K[1] + x[..., 1, :] * K[0]
    if H % 2 == 1:
        ya[..., -1, :] += x[..., -1, :] * K[3] + x[..., -2, :] * K[4]
    else:
        ya[..., -1, :] += x[..., -1, :] * K[4]
    H2 = ya.shape[-2]
    y = F.conv2d(ya.reshape(-1, 1, H2, W), K.view(1, 1, 1, 5), stride=(1, 2), padding=(0, 2)).view(lead + (H2, -1))
    y[..., :, 0] += ya[..., :, 0] * K[1] + ya[..., :, 1] * K[0]
    if H % 2 == 1:  # sic: rows, not columns
        y[..., :, -1] += ya[..., :, -1] * K[3] + ya[..., :, -2] * K[4]
    else:
        y[..., :, -1] += ya[..., :, -1] * K[4]
    return y


def _stuff(x, size, dim):
    """lpyr_dec.py:129-145: zero-interleave along `dim` to size+4 with the edge copies."""
    shp = list(x.shape)
    shp[dim] = size + 4
    z = torch.zeros(shp, dtype=x.dtype)
    odd = size % 2
    if dim == -2:
        z[..., 2:-2:2, :] = x
        z[..., 0, :] = x[..., 0, :]
        z[..., -2 + odd, :] = x[..., -1, :]
    else:
        z[..., :, 2:-2:2] = x
        z[..., :, 0] = x[..., :, 0]
        z[..., :, -2 + odd] = x[..., :, -1]
    return z


def pyr_expand(x, sz):
    """lpyr_dec.py:223-239."""
    K = _k5(x) * 2
    lead = x.shape[:-2]
    ya = _stuff(x, sz[0], -2)
    H, W = ya.shape[-2], ya.shape[-1]
    ya = F.conv2d(ya.reshape(-1, 1, H, W), K.view(1, 1, 5, 1)).view(lead + (-1, W))
    y = _stuff(ya, sz[1], -1)
    H, W = y.shape[-2], y.shape[-1]
    return F.conv2d(y.reshape(-1, 1, H, W), K.view(1, 1, 1, 5)).view(lead + (H, -1))


def gaussian_pyramid(R, levels):
    g = [R]
    for _ in range(1, levels):
        g.append(pyr_reduce(g[-1]))
    return g


def weber_contrast_pyramid(R, levels):
    """weber_contrast_pyr.decompose with contrast='weber_g1' (lpyr_dec.py:364-414).

    R: [B, 2*nch, F, H, W].  Returns (contrast list, log10 L_bkg list, gaussian pyramid)."""
    g = gaussian_pyramid(R, levels)
    contrast, logL = [], []
    for i in range(levels):
        if i == levels - 1:
            layer = g[i]
            Lb = torch.mean(torch.clamp(g[i][:, 0:2], min=0.01), dim=[-1, -2], keepdim=True)
        else:
            ex = pyr_expand(g[i + 1], [g[i].shape[-2], g[i].shape[-1]])
            layer = g[i] - ex
            Lb = torch.clamp(ex[:, 0:2], min=0.01)
        c = torch.empty_like(layer)
        c[:, 0::2] = torch.clamp(torch.div(layer[:, 0::2], Lb[:, 0:1]), max=1000.0)
        c[:, 1::2] = torch.clamp(torch.div(layer[:, 1::2], Lb[:, 1:2]), max=1000.0)
        contrast.append(c)
        logL.append(torch.log10(Lb))
    return contrast, logL, g


# --------------------------------------------------------------------------------------
# castleCSF (pycvvdp/csf.py, pycvvdp/interp.py)
# --------------------------------------------------------------------------------------
class CSF:
    def __init__(self, bundle):
        lut = bundle["csf_lut_weber_fixed_size"]
        self.log_L = torch.log10(torch.as_tensor(lut["L_bkg"]))  # csf.py:13
        self.log_rho = torch.log10(torch.as_tensor(lut["rho"]))  # csf.py:14
        self.tab = [[torch.as_tensor(lut["o0_c%d" % (c + 1)]) for c in range(3)], [torch.as_tensor(lut["o5_c1"])]]

    def row(self, rho, oo, cc):
        """Interpolate the LUT over rho (csf.py:39-46, interp.py:152-178): 32 log-sensitivities over L_bkg."""
        fp = self.tab[oo][cc].contiguous()
        N = self.log_L.numel()
        x = torch.log10(torch.as_tensor(rho, dtype=torch.float32)).expand(N).contiguous()
        xp = self.log_rho.contiguous()
        idx = torch.clamp(torch.searchsorted(xp, x) - 1, 0, len(xp) - 2)
        x0, x1 = xp[idx], xp[idx + 1]
        y0 = fp[torch.arange(N), idx]
        y1 = fp[torch.arange(N), idx + 1]
        return y0 + (y1 - y0) / (x1 - x0) * (x - x0)

    def sensitivity(self, row, logL):
        """interp1q + 10** (csf.py:49, interp.py:55-60,92-100)."""
        x = self.log_L
        q = logL.flatten()
        ind = ((q - x[0]) / (x[-1] - x[0]) * (x.numel() - 1)).clamp(0, x.shape[0] - 1)
        fr = torch.frac(ind)
        i0 = ind.to(torch.int32)
        i1 = (i0 + 1).clamp(max=x.shape[0] - 1)
        v = row[i0] * (1.0 - fr) + row[i1] * fr
        return 10 ** v.reshape(logL.shape)


# --------------------------------------------------------------------------------------
# Metric (pycvvdp/cvvdp_metric.py)
# --------------------------------------------------------------------------------------
def gauss_kernel1d(ksize=13, sigma=3.0):
    """torchvision GaussianBlur kernel (published algorithm, see oracle/ref_shims)."""
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize, dtype=torch.float32)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def safe_pow(x, p):
    eps = torch.as_tensor(0.00001)  # cvvdp_metric.py:77-84
    return (x + eps) ** p - eps ** p


class Oracle:
    """predict()/predict_video_source() of class cvvdp (cvvdp_metric.py:108-441), CPU, block = 1 frame."""

    def __init__(self, display_name="standard_4k", heatmap=None, temp_padding="replicate", bundle=None, keep=False,
                 photometry=None, geometry=None, features=False):
        self.bundle = bundle or load_bundle()
        self.display = Display(display_name, self.bundle, photometry=photometry, geometry=geometry)
        self.ppd = self.display.ppd
        self.heatmap = heatmap
        self.do_heatmap = heatmap is not None and heatmap != "none"
        self.temp_padding = temp_padding
        self.keep = keep  # keep intermediates of the LAST processed block in self.dbg
        self.dbg = {}
        # features=True: predict() also collects what cvvdp_ml_base.extract_features returns (cvvdp_ml_metric.py:193-277):
        # self.features[band] = list (one entry per frame / per image) of [B, 1, cells_y, cells_x, channels, 6] tensors
        self.features = [] if features else None
        p = self.bundle["cvvdp_parameters"]
        T = torch.as_tensor  # cvvdp_metric.py:154-226: every scalar becomes a tensor of its JSON type
        self.mask_p, self.mask_c = T(p["mask_p"]), T(p["mask_c"])
        self.mask_q = T(p["mask_q"])
        self.beta, self.beta_t, self.beta_tch, self.beta_sch = T(p["beta"]), T(p["beta_t"]), T(p["beta_tch"]), T(p["beta_sch"])
        self.sens_corr = T(p["sensitivity_correction"])
        self.jod_a, self.jod_exp = T(p["jod_a"]), T(p["jod_exp"])
        self.xcm = T(p["xcm_weights"], dtype=torch.float32)
        self.image_int = T(p["image_int"])
        self.ch_chrom_w, self.ch_trans_w = T(p["ch_chrom_w"]), T(p["ch_trans_w"])
        self.sigma_tf, self.beta_tf = T(p["sigma_tf"]), T(p["beta_tf"])
        self.baseband_weight = T(p["baseband_weight"])
        self.d_max = T(p["d_max"])
        self.pu_dilate = p["pu_dilate"]
        self.version = p["version"]
        assert p["masking_model"] == "mult-mutual" and p["contrast"] == "weber_g1" and p["dclamp_type"] == "soft"
        self.csf = CSF(self.bundle)
        k1 = gauss_kernel1d(int(self.pu_dilate * 4) + 1, float(self.pu_dilate))
        self.blur_k2d = torch.mm(k1[:, None], k1[None, :])
        self.pad = int(self.pu_dilate * 2)

    # ---- temporal filters: cvvdp_metric.py:1057-1092 ----
    def temporal_filters(self, fps):
        N = int(math.ceil(0.250 * fps / 2) * 2) + 1
        Nw = int(N / 2) + 1
        w = torch.linspace(0, fps / 2, Nw).view(1, Nw)
        R = torch.empty((4, Nw))
        R[0:3] = torch.exp(-w ** self.beta_tf[0:3].view(3, 1) / self.sigma_tf[0:3].view(3, 1))
        R[3:4] = torch.exp(-(w ** self.beta_tf[3] - torch.as_tensor(5.0) ** self.beta_tf[3]) ** 2 / self.sigma_tf[3])
        return [torch.fft.fftshift(torch.real(torch.fft.irfft(R[k], norm="backward", n=N))) for k in range(4)]

    def ch_weights(self, n):
        w = torch.stack([torch.as_tensor(1.0), self.ch_chrom_w, self.ch_chrom_w, self.ch_trans_w])  # :597-606
        return w[0:n].view(1, -1, 1, 1)

    def lp_norm(self, x, p, dim, normalize=True, keepdim=True):
        # cvvdp_metric.py:1032-1048 (p is always a tensor here)
        N = 1.0
        if normalize:
            for d in (dim if isinstance(dim, tuple) else (dim,)):
                N *= x.shape[d]
        return safe_pow(torch.sum(safe_pow(x, p), dim=dim, keepdim=keepdim) / float(N), 1 / p)

    def met2jod(self, Q):
        # cvvdp_metric.py:646-658
        Qt = 0.1
        a_p = self.jod_a * (Qt ** (self.jod_exp - 1.0))
        out = torch.empty_like(Q)
        out[Q <= Qt] = 10.0 - a_p * Q[Q <= Qt]
        out[Q > Qt] = 10.0 - self.jod_a * (Q[Q > Qt] ** self.jod_exp)
        return out

    def pool(self, Q_per_ch):
        # cvvdp_metric.py:610-643
        nch, nfr, nb = Q_per_ch.shape[1], Q_per_ch.shape[2], Q_per_ch.shape[3]
        wb = torch.ones((1, nch, 1, nb), dtype=torch.float32)
        wb[:, :, 0, -1] = self.baseband_weight[0:nch]
        Qsc = self.lp_norm(Q_per_ch * self.ch_weights(nch) * wb, self.beta_sch, dim=3, normalize=False)
        Qtc = self.lp_norm(Qsc, self.beta_tch, dim=1, normalize=False)
        if nfr == 1:
            Q = Qtc * self.image_int
        else:
            Q = self.lp_norm(Qtc, self.beta_t, dim=2, normalize=True)
        return self.met2jod(Q.squeeze())

    def blur(self, M):
        # phase_uncertainty: cvvdp_metric.py:963-971 (+ torchvision GaussianBlur(13, 3))
        H, W = M.shape[-2], M.shape[-1]
        c10 = 10 ** self.mask_c
        if self.pu_dilate != 0 and H > self.pad and W > self.pad:
            x = F.pad(M.reshape(-1, 1, H, W), [self.pad] * 4, mode="reflect")
            x = F.conv2d(x, self.blur_k2d.view(1, 1, *self.blur_k2d.shape))
            return x.view(M.shape) * c10
        return M * c10

    def masking(self, T, R, S):
        # apply_masking_model 'mult-mutual': cvvdp_metric.py:835-856, mask_pool :753-760, clamp_diffs :948-950
        nch = T.shape[1]
        gain = torch.as_tensor([1, 1.45, 1, 1.0]).view(1, 4, 1, 1, 1)[:, :nch]
        Tp = T * S * gain
        Rp = R * S * gain
        Mmm = self.blur(torch.min(torch.abs(Tp), torch.abs(Rp)))
        q = self.mask_q[0:nch].view(nch, 1, 1, 1)
        C = safe_pow(torch.abs(Mmm), q)
        xw = torch.reshape(2 ** self.xcm, (4, 4))[:nch, :]
        M = torch.empty_like(C)
        for cc in range(nch):
            M[:, cc:cc + 1] = torch.sum(C * xw[:, cc].view(1, -1, 1, 1, 1), dim=-4, keepdim=True)
        Du = safe_pow(torch.abs(Tp - Rp), self.mask_p) / (1 + M)
        mx = 10 ** self.d_max
        return mx * Du / (mx + Du)

    def process_block(self, R, levels, rho_band, is_image):
        """process_block_of_frames: cvvdp_metric.py:660-751.  R: [B, 2*nch, Nb, H, W]."""
        nch = R.shape[1] // 2
        B, Nb = R.shape[0], R.shape[2]
        contrast, logL, g = weber_contrast_pyramid(R, levels)
        Q = torch.empty((B, nch, Nb, levels))
        hm_bands = []
        if self.keep:
            self.dbg.update(R=R, gpyr=g, contrast=contrast, logL=logL, S=[], D=[])
        for bb in range(levels):
            base = bb == levels - 1
            mul = 1.0 if (bb == 0 or base) else 2.0  # lpyr_dec.get_band :60-66
            Bb = contrast[bb] * mul
            Tf, Rf = Bb[:, 0::2], Bb[:, 1::2]
            lL = logL[bb][:, 1:2]
            S = torch.empty((B, nch, Nb) + tuple(lL.shape[-2:]))
            for cc in range(nch):
                oo, ci = (0, cc) if cc < 3 else (1, 0)
                row = self.csf.row(rho_band[bb], oo, ci)
                S[:, cc:cc + 1] = self.csf.sensitivity(row, lL) * 10.0 ** (self.sens_corr / 20.0)
            D = torch.abs(Tf - Rf) * S if base else self.masking(Tf, Rf, S)
            Q[:, :, :, bb] = self.lp_norm(D, self.beta, dim=(-2, -1), normalize=True, keepdim=False)
            if self.features is not None:
                # cvvdp_ml_base.process_block_of_frames, cvvdp_ml_metric.py:351-358: cvvdp_feature_pooling(ceil(pix_per_deg)) of
                # |T_f|*S, |R_f|*S and D (:77-107: AvgPool2d(fs, ceil_mode=True) of x and x**2, variance = E[x^2] - mean^2),
                # laid out [batch, frames, cells_y, cells_x, channels, 6]
                fs = int(math.ceil(self.ppd))
                pool = torch.nn.AvgPool2d((fs, fs), ceil_mode=True)

                def ap(x):                                   # cvvdp_avg_pool (:62-73): batch and channel merged for the 2-D pool
                    v = x.reshape((-1,) + tuple(x.shape[2:]))
                    y = pool(v)
                    return y.reshape(tuple(x.shape[0:2]) + tuple(y.shape[1:]))

                order = [0, 2, 3, 4, 1]
                cols = []
                for x in (torch.abs(Tf) * S, torch.abs(Rf) * S, D):
                    mean = ap(x).permute(order)
                    cols += [mean, ap(x ** 2).permute(order) - mean ** 2]
                self.features[bb].append(torch.stack(cols, dim=5))
            if self.keep:
                self.dbg["S"].append(S)
                self.dbg["D"].append(D)
            if self.do_heatmap:
                t_int = self.image_int if is_image else 1.0
                w = self.ch_weights(nch).view(-1, 1, 1, 1) * t_int
                if base:
                    w = w * self.baseband_weight[0:nch].view(-1, 1, 1, 1)
                Dchr = self.lp_norm(D * w, self.beta_tch, dim=-4, normalize=False)
                hm_bands.append(Dchr / mul)  # lpyr_dec_2.set_lband :308-314
        hm = None
        if self.do_heatmap:
            img = hm_bands[-1]
            for i in reversed(range(levels - 1)):  # lpyr_dec_2.reconstruct :328-335
                img = pyr_expand(img, [hm_bands[i].shape[-2], hm_bands[i].shape[-1]])
                img = img + hm_bands[i]
            hm = 1.0 - self.met2jod(img) / 10.0
            if self.keep:
                self.dbg["hm_bands"] = hm_bands
        return Q, hm

    def _sym_index(self, fi, n):
        # cvvdp_metric.py:445-450
        even = (math.floor((abs(fi) - 1) / (n - 1)) % 2) == 0
        return ((abs(fi) - 1) % (n - 1)) + 1 if even else fi % (n - 1)

    def predict(self, test, ref, dim_order="BCFHW", frames_per_second=0, first_frame=0, n_frames=None, halo_from=None):
        """cvvdp.predict (cvvdp_metric.py:285-441) on arrays, block_N_frames = 1.

        first_frame/n_frames/halo_from are oracle-only conveniences for shard tests: evaluate frames
        [first_frame, first_frame+n_frames) of the clip, with temporal history taken from the real
        preceding frames (padding only before frame 0, exactly as the unsharded run would)."""
        t = to_bcfhw(test, dim_order)
        r = to_bcfhw(ref, dim_order)
        B = max(t.shape[0], r.shape[0])
        C, Ftot, H, W = t.shape[1], t.shape[2], t.shape[3], t.shape[4]
        if frames_per_second == 0 and Ftot > 1:
            raise RuntimeError("When passing video sequences, you must set frames_per_second parameter")
        if C not in (1, 3):
            raise RuntimeError("The content must have either 1 or 3 color channels.")
        is_image = Ftot == 1
        height, freqs = band_frequencies(W, H, self.ppd)
        levels = height + 1
        if self.features is not None:
            self.features = [[] for _ in range(levels)]
        rho_band = freqs.copy()
        rho_band[levels - 1] = 0.1  # Q2, cvvdp_metric.py:685-686
        n_frames = Ftot - first_frame if n_frames is None else n_frames
        def dkl(a, f):
            x = self.display.to_dkl(fetch_frame(a, f))
            return x.expand(-1, 3, -1, -1, -1) if x.shape[1] == 1 else x  # Q6: luminance fills all three planes
        hm_out = None
        if self.do_heatmap:
            hm_out = torch.zeros([1, 1 if self.heatmap == "raw" else 3, n_frames, H, W], dtype=torch.float16)
        if is_image:
            Rb = torch.empty((B, 6, 1, H, W))
            Rb[:, 0::2] = dkl(t, 0)
            Rb[:, 1::2] = dkl(r, 0)
            Qpc, hm = self.process_block(Rb, levels, rho_band, True)
            if self.do_heatmap:
                hm_out[:, :, 0:1] = self._colour(hm, Rb[:, 0])
        else:
            taps = self.temporal_filters(frames_per_second)
            self.taps = taps
            fl = taps[0].numel()
            # history window of fl frames ending at the current one; index <0 -> padding
            def src_index(i):
                if i >= 0:
                    return i
                return 0 if self.temp_padding == "replicate" else self._sym_index(i, Ftot)
            cache = {}
            def get(a, key, i):
                k = (key, i)
                if k not in cache:
                    cache[k] = dkl(a, i)
                return cache[k]
            Qpc = torch.zeros((B, 4, n_frames, levels))
            for fo in range(n_frames):
                ff = first_frame + fo
                idx = [src_index(ff - (fl - 1) + k) for k in range(fl)]
                wt = torch.cat([get(t, 0, i) for i in idx], dim=2)  # [B,3,fl,H,W]
                wr = torch.cat([get(r, 1, i) for i in idx], dim=2)
                for k in list(cache):
                    if k[1] not in idx:
                        del cache[k]
                Rb = torch.zeros((B, 8, 1, H, W))
                for cc in range(4):  # cvvdp_metric.py:554-560
                    cf = taps[cc].flip(0).view(1, 1, fl, 1, 1)
                    sc = 0 if cc == 3 else cc
                    Rb[:, 2 * cc:2 * cc + 1] = (wt[:, sc:sc + 1] * cf).sum(dim=-3, keepdim=True)
                    Rb[:, 2 * cc + 1:2 * cc + 2] = (wr[:, sc:sc + 1] * cf).sum(dim=-3, keepdim=True)
                Qb, hm = self.process_block(Rb, levels, rho_band, False)
                Qpc[:, :, fo:fo + 1, :] = Qb
                if self.do_heatmap:
                    hm_out[:, :, fo:fo + 1] = self._colour(hm, Rb[:, 0])
        stats = {"Q_per_ch": Qpc.numpy(), "rho_band": rho_band, "frames_per_second": frames_per_second,
                 "width": W, "height": H, "N_frames": n_frames}
        if self.features is not None:      # [B, F, cells_y, cells_x, channels, 6] per band, like extract_features (:262-277)
            stats["features"] = [torch.cat(f, dim=1).numpy() for f in self.features]
        if self.do_heatmap:
            stats["heatmap"] = hm_out
        return self.pool(Qpc), stats

    # ---- heatmap colouring: visualize_diff_map.py ----
    def _colour(self, hm, ctx):
        if self.heatmap == "raw":
            return hm.type(torch.float16)  # cvvdp_metric.py:398
        return visualize_diff_map(hm, ctx, self.heatmap).type(torch.float16)


def _interp1(x, v, xq):
    """interp.py:22-31,81-89 (bucketize-based, +1e-6 in the denominator)."""
    shp = xq.shape
    q = xq.flatten()
    imax = torch.bucketize(q, x)
    imax[imax >= x.shape[0]] = x.shape[0] - 1
    imin = (imax - 1).clamp(0, x.shape[0] - 1)
    fr = (q - x[imin]) / (x[imax] - x[imin] + 0.000001)
    fr[imax == imin] = 0.0
    fr[fr < 0.0] = 0.0
    return (v[imin] * (1.0 - fr) + v[imax] * fr).reshape(shp)


def visualize_diff_map(diff_map, context, kind):
    """visualize_diff_map.py:48-106 with vis_tonemap :23-45 and log_luminance :17-20."""
    d = torch.clamp(diff_map, 0.0, 1.0)
    y = context
    if y.shape[1] == 3:  # Q3: a block of exactly 3 frames is read as RGB (visualize_diff_map.py:7)
        y = y[:, 0:1] * 0.212656 + y[:, 1:2] * 0.715158 + y[:, 2:3] * 0.072186
    b = torch.log(torch.clamp(y, min=torch.min(y[y > 0.0])))
    dr = 0.6
    bmin, bmax = torch.min(b), torch.max(b)
    if bmax - bmin < dr:
        tmo = (b - bmin) / (bmax - bmin + 1e-3) * dr + (1 - dr) / 2
    else:
        scale = torch.linspace(bmin, bmax, 1024)
        p = torch.histc(b, 1024, bmin, bmax)
        p = p / torch.sum(p)
        s = torch.sum(torch.pow(p, 1.0 / 3.0))
        dy = torch.pow(p, 1.0 / 3.0) / s
        v = torch.cumsum(dy, 0) * dr + (1.0 - dr) / 2.0
        tmo = _interp1(scale, v, b)
    if kind == "threshold":
        cm = torch.tensor([[0.2, 0.2, 1.0], [0.2, 1.0, 1.0], [0.2, 1.0, 0.2], [1.0, 1.0, 0.2], [1.0, 0.2, 0.2]])
        cin = torch.tensor([0.00, 0.25, 0.50, 0.75, 1.00]) * 0.1
    elif kind == "supra-threshold":
        cm = torch.tensor([[0.2, 1.0, 1.0], [1.0, 1.0, 1.0], [1.0, 1.0, 0.2]])
        cin = torch.tensor([0.0, 0.5, 1.0]) * 0.3
    else:
        raise RuntimeError("unknown colormap")
    n, h, w = d.shape[-3], d.shape[-2], d.shape[-1]
    out = torch.empty([3, n, h, w], dtype=torch.float16)
    cl = cm[:, 0:1] * 0.212656 + cm[:, 1:2] * 0.715158 + cm[:, 2:3] * 0.072186
    cch = cm / (torch.cat([cl] * 3, 1) + 0.0001)
    for c in range(3):
        out[c:c + 1] = _interp1(cin, cch[:, c], d).type(torch.float16)
    return (out * tmo).clip(0.0, 1.0)
