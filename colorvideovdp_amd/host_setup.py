"""Host-side set-up math of the metric (numpy, fp32 where the reference is fp32).

Everything here runs once per metric / per clip and produces the small tables the HIP core consumes:
  band frequencies and pyramid height     lpyr_dec.py:18-52
  temporal filters                         cvvdp_metric.py:1057-1092
  castleCSF rows per band                  csf.py:8-46, interp.py:152-178
  phase-uncertainty blur taps              torchvision GaussianBlur(13, 3) (published algorithm)
  derived masking / pooling constants      cvvdp_metric.py:146-229
"""
import math

import numpy as np

f32 = np.float32


def band_frequencies(W, H, ppd):
    """(height, band_freqs[height+1]) as in lpyr_dec.__init__ (lpyr_dec.py:25-42)."""
    max_levels = int(np.floor(np.log2(min(H, W)))) - 1
    bands = np.concatenate([[1.0], np.power(2.0, -np.arange(0.0, 14.0)) * 0.3228], 0) * ppd / 2.0
    invalid = np.nonzero(bands <= 0.2)[0]
    max_band = max_levels if invalid.size == 0 else invalid[0]
    height = int(np.clip(max_band + 1, 0, max_levels))
    freqs = np.array([1.0] + [0.3228 * 2.0 ** (-f) for f in range(height)]) * ppd / 2.0
    return height, freqs


def temporal_filters(fps, beta_tf, sigma_tf):
    """Four fp32 FIR kernels F[c] (Y-sust, RG, YV, Y-trans), cvvdp_metric.py:1057-1092."""
    N = int(math.ceil(0.250 * fps / 2) * 2) + 1
    Nw = int(N / 2) + 1
    beta = np.asarray(beta_tf, dtype=f32)
    sigma = np.asarray(sigma_tf, dtype=f32)
    w = np.linspace(0.0, fps / 2, Nw).astype(f32)
    R = np.empty((4, Nw), dtype=f32)
    for c in range(3):
        R[c] = np.exp(-np.power(w, beta[c]) / sigma[c])
    R[3] = np.exp(-np.square(np.power(w, beta[3]) - np.power(f32(5.0), beta[3])) / sigma[3])
    F = np.empty((4, N), dtype=f32)
    for c in range(4):
        F[c] = np.fft.fftshift(np.fft.irfft(R[c].astype(np.float64), n=N)).astype(f32)
    return F


class CsfTable:
    """castleCSF look-up table 'weber_fixed_size' (csf.py:8-25)."""

    def __init__(self, lut):
        self.log_L = np.log10(np.asarray(lut["L_bkg"], dtype=f32))
        self.log_rho = np.log10(np.asarray(lut["rho"], dtype=f32))
        self.tab = [np.asarray(lut["o0_c1"], dtype=f32), np.asarray(lut["o0_c2"], dtype=f32),
                    np.asarray(lut["o0_c3"], dtype=f32), np.asarray(lut["o5_c1"], dtype=f32)]  # [L_bkg][rho]

    def rows(self, rho):
        """[4][32] log10 sensitivities over the L_bkg nodes at spatial frequency rho (csf.py:39-46):
        linear interpolation over log10 rho with extrapolation outside the grid (interp.py:152-178)."""
        x = np.log10(f32(rho))
        xp = self.log_rho
        idx = int(np.clip(np.searchsorted(xp, x, side="left") - 1, 0, len(xp) - 2))
        x0, x1 = xp[idx], xp[idx + 1]
        out = np.empty((4, len(self.log_L)), dtype=f32)
        for c, fp in enumerate(self.tab):
            y0, y1 = fp[:, idx], fp[:, idx + 1]
            out[c] = y0 + (y1 - y0) / (x1 - x0) * (x - x0)
        return out


def gaussian_taps(ksize=13, sigma=3.0):
    half = (ksize - 1) * 0.5
    x = np.linspace(-half, half, ksize).astype(f32)
    pdf = np.exp(f32(-0.5) * np.square(x / f32(sigma)))
    return (pdf / pdf.sum(dtype=f32)).astype(f32)


def symmetric_frame_index(frame_ind, frame_count):
    """cvvdp_metric.py:445-450."""
    is_even = (math.floor((abs(frame_ind) - 1) / (frame_count - 1)) % 2) == 0
    if is_even:
        return ((abs(frame_ind) - 1) % (frame_count - 1)) + 1
    return frame_ind % (frame_count - 1)
