#!/usr/bin/env python3
"""profiles/<tag>_pmc_fetch_size.txt + <tag>_pmc_write_size.txt -> profiles/traffic.json (read by bench.py): counter bytes
against algorithmic bytes for every dominant kernel of the 4K x 64 fp32 step.

FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 counts the 128-byte requests of wide coalesced reads as 64 B);
WRITE_SIZE is used as reported (calibrated on k_fir_rot, whose reads and writes are exactly its algorithmic bytes).
The file is stamped with the hash of the kernel sources / library the counters were measured on (profiles/<tag>_stamp.json, written
by tools/refresh_profiles.sh on the GPU box in the same call); bench.py drops the counters when its own sources differ.
Usage: make_traffic_json.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIX = 3840 * 2160 * 64
KERNELS = {
    # key: (substring of the kernel name, workgroups of the level-0 launch, algorithmic bytes per launch, what they are)
    "temporal_fir": ("k_fir_rot<3, 17>", None, PIX * (24 + 32.0), "24 B/pixel in (fp32 RGB, test + reference) + 32 B/pixel out (8 level-0 planes)"),
    "pyr_reduce_l0": ("k_reduce2", None, PIX * (32 + 8 + 2.0), "levels 0 -> 1 -> 2 in one pass: 32 B/pixel in, 8 + 2 B/pixel out"),
    "band_level0": ("k_band4<4, false, false, false, false>", None, PIX * 40.0, "g0 (32 B/pixel) + g1 (8 B/pixel) in; partial sums out"),
}


def rows(path, counter):
    out = {}
    for line in open(path):
        f = line.split()
        if counter in f:
            i = f.index(counter)
            name = " ".join(f[:i - 1])
            out.setdefault(name, []).append((int(f[i - 1]), int(f[i + 1]), float(f[i + 3])))
    return out


def biggest(table, sub):
    best = None
    for name, lst in table.items():
        if sub in name:
            for wg, disp, per in lst:
                if best is None or wg * per > best[0] * best[2]:
                    best = (wg, disp, per)
    return best


def main(tag):
    fe = rows(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_size.txt"), "FETCH_SIZE")
    wr = rows(os.path.join(ROOT, "profiles", f"{tag}_pmc_write_size.txt"), "WRITE_SIZE")
    kernels = {}
    for key, (sub, _, algo, what) in KERNELS.items():
        f, w = biggest(fe, sub), biggest(wr, sub)
        if f is None:
            continue
        hbm = f[2] * 1024 * 2 + (w[2] * 1024 if w else 0.0)
        kernels[key] = {"kernel": sub, "workgroups": f[0], "FETCH_SIZE_KB_per_launch": f[2], "WRITE_SIZE_KB_per_launch": w[2] if w else None,
                        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": algo, "measured_over_algorithmic": round(hbm / algo, 4),
                        "algorithmic_bytes": what, "launches_sampled": f[1]}
    # what the counters were measured on: written next to them ON THE GPU BOX by tools/refresh_profiles.sh (bench.code_stamp():
    # SHA-256 of the kernel sources + header, and of the library binary).  bench.py quotes the counters only for the same sources.
    stamp = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_stamp.json")))
    try:
        import subprocess
        stamp["git_head_when_written"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        pass
    out = {"workload": "4k64", "dtype": "f32", "stamp": stamp, "kernels": kernels,
           "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests of wide coalesced reads as 64 B, MI355X_MICROARCH.md HBM section); "
                         "WRITE_SIZE as reported",
           "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
                     f"--no-profile; profiles/{tag}_pmc_*.txt (tools/refresh_profiles.sh)"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
