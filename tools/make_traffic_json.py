#!/usr/bin/env python3
"""profiles/<tag>_pmc_fetch_size.txt + <tag>_pmc_write_size.txt -> profiles/traffic_band0.json (read by bench.py).
FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 counts the 128-byte requests of wide coalesced reads as
64 B); WRITE_SIZE is used as reported.  Usage: make_traffic_json.py r01"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def level0_row(path, counter):
    best = None
    for line in open(path):
        f = line.split()
        if "k_band4<4" in line and counter in f:
            i = f.index(counter)
            wg, disp, per = int(f[i - 1]), int(f[i + 1]), float(f[i + 3])
            if best is None or wg > best[0]:
                best = (wg, disp, per)
    return best


def main(tag):
    fe = level0_row(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_size.txt"), "FETCH_SIZE")
    wr = level0_row(os.path.join(ROOT, "profiles", f"{tag}_pmc_write_size.txt"), "WRITE_SIZE")
    algo = 3840 * 2160 * 64 * 40.0
    out = {
        "workload": "4k64", "dtype": "f32",
        "kernel": "k_band4<4> level 0 (3840x2160, 64 frames per launch)",
        "FETCH_SIZE_KB_per_launch": fe[2], "WRITE_SIZE_KB_per_launch": wr[2],
        "launches_sampled": [fe[1], wr[1]],
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests of wide coalesced reads as 64 B, "
                      "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
        "hbm_bytes_per_launch": fe[2] * 1024 * 2 + wr[2] * 1024,
        "algorithmic_bytes_per_launch": algo,
        "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py "
                  f"--steps 2 --warmup 1 --no-profile; profiles/{tag}_pmc_*.txt (tools/refresh_profiles.sh)",
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_band0.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
