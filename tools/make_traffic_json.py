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
    # key: (substrings of the kernel names whose largest launches are ADDED, algorithmic bytes per step, what they are)
    "temporal_fir": (["k_fir_rot<3, 17>"], PIX * (24 + 32.0), "24 B/pixel in (fp32 RGB, test + reference) + 32 B/pixel out (8 level-0 planes)"),
    "band_level0": (["k_band4s(", "k_band4s_edge("], PIX * 40.0,
                    "k_band4s + k_band4s_edge: g0 (32 B/pixel) in, g1 (8 B/pixel) out; two launches side by side (strips inside the image; strips at "
                    "its left and right border on the same front / back-wave layout's EDGE body, round 5)"),
    "band_level1": (["k_band4s(#2", "k_band4s_edge(#2"], PIX * 10.0, "the same pair at level 1: g1 in, g2 out"),
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


def ranked(table, sub):
    """launch sizes of the kernel whose name contains `sub`, largest (workgroups x bytes) first: [(workgroups, dispatches, per dispatch)]"""
    out = []
    for name, lst in table.items():
        if sub in name:
            out += lst
    return sorted(out, key=lambda r: -r[0] * r[2])


def pick(table, spec):
    sub, _, nth = spec.partition("#")
    r = ranked(table, sub)
    n = int(nth) - 1 if nth else 0
    return r[n] if len(r) > n else None


def sq_of(tag, spec, expect_wg=None):
    """SQ counters of one launch size of one kernel (profiles/<tag>_pmc_sq_counters.txt, tools/sq_counters.sh): shader clocks, VALU pipe
    utilisation, VALU instructions -- what bounds a kernel that is not HBM-bound."""
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq_counters.txt")
    if not os.path.isfile(path):
        return None
    sub, _, nth = spec.partition("#")
    per = {}
    for line in open(path):
        f = line.split()
        if line.startswith("#") or sub not in line:
            continue
        idx = [i for i, x in enumerate(f) if x.startswith("SQ_")]
        if idx:
            per.setdefault(int(f[idx[0] - 1]), {})[f[idx[0]]] = float(f[idx[0] + 3])
    sizes = sorted(per, reverse=True)
    n = int(nth) - 1 if nth else 0
    if len(sizes) <= n or "SQ_BUSY_CYCLES" not in per[sizes[n]] or (expect_wg is not None and sizes[n] != expect_wg):
        return None
    c = per[sizes[n]]
    clocks = c["SQ_BUSY_CYCLES"] / 32
    out = {"workgroups": sizes[n], "shader_clocks_per_launch": round(clocks), "note": "counter passes run the kernels one at a time"}
    if "SQ_ACTIVE_INST_VALU" in c:
        out["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * clocks), 3)      # 1024 SIMDs; the counter is in quad-cycles
        out["valu_quad_cycles"] = round(c["SQ_ACTIVE_INST_VALU"])
    if "SQ_INSTS_VALU" in c:
        out["valu_instructions_per_launch"] = round(c["SQ_INSTS_VALU"])
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        out["waves_waiting_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    return out


def step_valu(tag):
    """All kernels of one 4K x 64 step on the VALU-issue roofline: sum of SQ_ACTIVE_INST_VALU (quad-cycles: x 4 clocks on one of 1 024
    SIMDs) against the sum of the kernels' shader clocks (SQ_BUSY_CYCLES / 32 SQs), each weighted by its launches per step (= its
    dispatch count relative to the temporal kernel's, which runs once per step)."""
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq_counters.txt")
    if not os.path.isfile(path):
        return None
    per = {}
    for line in open(path):
        f = line.split()
        if line.startswith("#"):
            continue
        idx = [i for i, x in enumerate(f) if x.startswith("SQ_")]
        if idx:
            i = idx[0]
            per.setdefault((" ".join(f[:i - 1]), int(f[i - 1])), {})[f[i]] = (int(f[i + 1]), float(f[i + 3]))
    fir = [v for (name, _), v in per.items() if "k_fir" in name and "SQ_BUSY_CYCLES" in v]
    if not fir:
        return None
    n_steps = max(v["SQ_BUSY_CYCLES"][0] for v in fir)
    valu = clocks = 0.0
    rows_ = []
    for (name, wg), v in sorted(per.items()):
        if "SQ_BUSY_CYCLES" not in v or "SQ_ACTIVE_INST_VALU" not in v:
            continue
        per_step = v["SQ_BUSY_CYCLES"][0] / n_steps
        valu += per_step * v["SQ_ACTIVE_INST_VALU"][1] * 4 / 1024
        clocks += per_step * v["SQ_BUSY_CYCLES"][1] / 32
        rows_.append({"kernel": name.replace("void cvvdp::", ""), "workgroups": wg, "launches_per_step": round(per_step, 3),
                      "valu_busy": round(v["SQ_ACTIVE_INST_VALU"][1] * 4 / 1024 / (v["SQ_BUSY_CYCLES"][1] / 32), 3)})
    if clocks <= 0:
        return None
    return {"valu_clocks_per_simd_per_step": round(valu), "shader_clocks_per_step": round(clocks), "frac_of_issue": round(valu / clocks, 3),
            "kernels": rows_, "note": "kernels measured one at a time by the counter passes; in the timed step the border-strip launches overlap the others",
            "source": f"profiles/{tag}_pmc_sq_counters.txt"}


def main(tag):
    fe = rows(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_size.txt"), "FETCH_SIZE")
    wr = rows(os.path.join(ROOT, "profiles", f"{tag}_pmc_write_size.txt"), "WRITE_SIZE")
    kernels = {}
    for key, (specs, algo, what) in KERNELS.items():
        fs, ws = [pick(fe, sp) for sp in specs], [pick(wr, sp) for sp in specs]
        if any(f is None for f in fs):
            continue
        fetch_kb = sum(f[2] for f in fs)
        write_kb = sum(w[2] for w in ws if w)
        hbm = fetch_kb * 1024 * 2 + write_kb * 1024
        kernels[key] = {"kernel": " + ".join(sp.partition("#")[0].rstrip("(") for sp in specs), "workgroups": [f[0] for f in fs], "FETCH_SIZE_KB_per_launch": fetch_kb,
                        "WRITE_SIZE_KB_per_launch": write_kb, "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": algo,
                        "measured_over_algorithmic": round(hbm / algo, 4), "algorithmic_bytes": what, "launches_sampled": fs[0][1]}
        sq = sq_of(tag, specs[0], fs[0][0])
        if sq:
            kernels[key]["sq"] = sq
            kernels[key]["sq_all"] = [s for s in (sq_of(tag, sp, f[0]) for sp, f in zip(specs, fs)) if s]        # every launch of the group
    # what the counters were measured on: written next to them ON THE GPU BOX by tools/refresh_profiles.sh (bench.code_stamp():
    # SHA-256 of the kernel sources + header, and of the library binary).  bench.py quotes the counters only for the same sources.
    stamp = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_stamp.json")))
    try:
        import subprocess
        stamp["git_head_when_written"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        pass
    out = {"workload": "4k64", "dtype": "f32", "stamp": stamp, "kernels": kernels, "step_valu": step_valu(tag),
           "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests of wide coalesced reads as 64 B, MI355X_MICROARCH.md HBM section); "
                         "WRITE_SIZE as reported",
           "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
                     f"--no-profile; profiles/{tag}_pmc_*.txt (tools/refresh_profiles.sh)"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r05")
