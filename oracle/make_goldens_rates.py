#!/usr/bin/env python3
"""Fixtures for frame rates whose temporal filters have 25 taps and more (container only; the REAL reference on CPU, like make_goldens.py):
90 fps (25 taps: k_fir_rot's 32-wide window), 100 fps (27 taps: runs on the 31-tap instantiation with zero taps in front; symmetric padding,
raw heat map) and 144 fps (37 taps: the generic temporal kernel).  The clips are longer than their filters, so blocks, history and padding
are all exercised.

    python oracle/make_goldens_rates.py            # writes tests/golden/vid_*_{90,100,144}_*.npz
"""
import numpy as np

import make_goldens as mg


def main():
    rng = np.random.default_rng(20260929)
    r = mg.pattern(rng, 34, 56, 72)
    t = mg.distort(rng, r)
    mg.run_case("vid_f32_56x72x34_90_fhd", mg.quant(t, "f32"), mg.quant(r, "f32"), "FHWC", 90, "standard_fhd")
    r = mg.pattern(rng, 40, 48, 64)
    t = mg.distort(rng, r, sigma=0.05)
    mg.run_case("vid_u8_48x64x40_100_4k_sym_raw", mg.quant(t, "u8"), mg.quant(r, "u8"), "FHWC", 100, "standard_4k", heatmap="raw", temp_padding="symmetric")
    r = mg.pattern(rng, 44, 40, 56)
    t = mg.distort(rng, r)
    mg.run_case("vid_u16_40x56x44_144_fhd", mg.quant(t, "u16"), mg.quant(r, "u16"), "FHWC", 144, "standard_fhd")


if __name__ == "__main__":
    main()
