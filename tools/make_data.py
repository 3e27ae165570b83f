#!/usr/bin/env python3
"""Build colorvideovdp_amd/data/vvdp_data.json from the reference's calibration data.

Container-only tool.  The hot path consumes calibrated *values* (metric parameters,
display models, colour-space primaries, the castleCSF look-up table); they are data,
not code, and are re-packed here into one bundle in this project's own layout:

    {"cvvdp_parameters": {...}, "display_models": {...}, "color_spaces": {...},
     "csf_lut_weber_fixed_size": {...}}

Sources (reference @ /root/reference/pycvvdp/vvdp_data/): cvvdp_parameters.json,
display_models.json, color_spaces.json, csf_lut_weber_fixed_size.json.
Users can still override any section with reference-format files through
`config_paths` / $CVVDP_PATH (see colorvideovdp_amd/config.py).
"""
import json
import os
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/pycvvdp/vvdp_data"
DST = os.path.join(os.path.dirname(__file__), "..", "colorvideovdp_amd", "data", "vvdp_data.json")

bundle = {"__comment": "Calibration data bundle for colorvideovdp_amd (values from ColorVideoVDP v0.5.6 vvdp_data)"}
for stem in ("cvvdp_parameters", "display_models", "color_spaces", "csf_lut_weber_fixed_size"):
    with open(os.path.join(SRC, stem + ".json")) as f:
        bundle[stem] = json.load(f)

with open(DST, "w") as f:
    json.dump(bundle, f, separators=(",", ":"))
print("wrote", os.path.abspath(DST), os.path.getsize(DST), "bytes")
