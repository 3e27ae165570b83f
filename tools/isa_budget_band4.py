#!/usr/bin/env python3
"""ISA-derived VALU budget of k_band4<4> (the level-0 band kernel): instruction counts per stage of the steady-state row loop, read
from the assembly of the build (colorvideovdp_amd/csrc/build/band4-hip-amdgcn-amd-amdhsa-gfx950.s, kept by `make`), priced with the per-instruction costs
measured on MI355X (profiles/r01_ubench_valu_rates.txt: ns per wave64 instruction per SIMD; except v_cndmask: that file's 9.9 ns was
the micro-benchmark's own serial VCC dependence -- taking four v_cndmask per row out of k_band4f changed its time by nothing
(profiles/r03_dev_notes.txt 12), so a select is priced like the other one-pass integer / compare operations).

    python tools/isa_budget_band4.py [band4.s] > profiles/r03_band4_isa_budget.txt
    python tools/isa_budget_band4.py --fused [band4f.s] > profiles/r03_band4f_isa_budget.txt     (k_band4f<4, 0>: eight rows per loop trip)
"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN5cvvdp7k_band4ILi4ELb0ELb0ELb0ELb0EEEvNS_8BandArgsE"
COST = {"packed fp32 (v_pk_*)": 2.17, "transcendental (exp/log/rcp)": 3.47, "v_fma / v_fmaak": 1.75, "v_fmac": 1.59,
        "add/sub/mul/mov": 1.2, "min/max/med3/cvt/other": 1.8, "v_cndmask": 1.8}


def cls(op):
    if op.startswith("v_pk_"):
        return "packed fp32 (v_pk_*)"
    if op in ("v_exp_f32_e32", "v_log_f32_e32", "v_rcp_f32_e32"):
        return "transcendental (exp/log/rcp)"
    if op.startswith("v_fmac"):
        return "v_fmac"
    if op.startswith(("v_fma_", "v_fmaak", "v_fmamk")):
        return "v_fma / v_fmaak"
    if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_mov_b32", "v_add_u32", "v_subrev")):
        return "add/sub/mul/mov"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    return "min/max/med3/cvt/other"


def main(path, fused=False):
    global KERNEL
    rows_per_trip, n_bar = 2, 4
    if fused:
        KERNEL, rows_per_trip, n_bar = "_ZN5cvvdp8k_band4fILi4ELi0EEEvNS_8BandArgsE", 8, 16
    L = open(path).read().split("\n")
    s = next(i for i, l in enumerate(L) if l.startswith(KERNEL + ":"))
    e = next(i for i in range(s, len(L)) if ".end_amdhsa_kernel" in L[i])
    K = L[s:e]
    labels = {l.split(":")[0]: i for i, l in enumerate(K) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(K):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    # the steady-state loop (rows inside the image, rolling coarse window): the inner loop with the fewest instructions among the
    # row-pair loops (the reflected-row copy reloads the coarse window: three more loads per row)
    big = [(hi - lo, lo, hi) for lo, hi in loops if sum(1 for x in K[lo:hi] if "s_barrier" in x) == n_bar]
    _, lo, hi = min(big)
    segs, cur = [], []
    for l in K[lo:hi + 1]:
        code = l.split(";")[0].strip()
        if not code or (code.startswith(".") and not code.endswith(":")):
            continue
        if code.endswith(":") or code.startswith(("s_cbranch", "s_barrier", "s_branch")):
            if cur:
                segs.append(cur)
                cur = []
            if code.startswith("s_barrier"):
                segs.append(["BARRIER"])
            continue
        cur.append(code)
    if cur:
        segs.append(cur)

    def what(sg):
        ops = collections.Counter(c.split()[0] for c in sg)
        nv = sum(v for k, v in ops.items() if k.startswith("v_"))
        if nv == 0:
            return None
        if fused and ops["v_fmac_f32_e32"] + ops["v_fma_f32"] >= 24 and ops["v_min_f32_e32"] + ops["v_min_f32_e64"] < 8 and ops["v_rcp_f32_e32"] < 4 and ops["v_pk_fma_f32"] < 20:
            return "reduce of row r+5: horizontal 5-tap of 2 coarse columns x 2 planes, running vertical sums (+ window roll / level l+1 store on even rows)"
        if ops["v_rcp_f32_e32"] >= 4:
            return "pooling stage of row r-7: 1+M (packed), X = d^p - eps^p, D = X/((1+M) + X/dmax), sum D(D+2eps)"
        if ops["v_log_f32_e32"] == 1 and ops["v_exp_f32_e32"] == 4:
            return "luminance stage of row r+1 (one column per thread): L_T, L_R, 1/L, log2 L -> CSF LUT lerp -> 4 x exp2"
        if ops["v_pk_fma_f32"] >= 24 and ops["v_log_f32_e32"] == 4:
            return "vertical 13-tap blur (26 packed FMAs) + Mq = (blur + eps)^q"
        if ops["v_pk_fma_f32"] >= 20:
            return "horizontal 13-tap blur of 4 columns (packed FMAs) + window write"
        if ops["v_min_f32_e32"] + ops["v_min_f32_e64"] >= 8:
            return "contrast stage of row r: horizontal expand x 2 planes, Weber contrast, clamp, min(|T'|,|R'|) * S"
        if ops["v_cndmask_b32_e64"] >= 3:
            return "reflect-padding mirror writes (edge strips only: skipped by execz elsewhere)"
        if nv <= 13 and (ops["v_fmac_f32_e32"] >= 4 or ops["v_mul_f32_e32"] >= 4) and "ds_write_b128" in ops:
            return "vertical expand of row r+1 from the rolling coarse window -> s_ve"
        if ops["v_fma_f32"] >= 4 and nv <= 12:
            return "|T'-R'| * S + eps -> s_d ring"
        if all(k.startswith(("v_mov", "s_")) or not k.startswith("v_") for k in ops):
            return "coarse-window roll (every row pair) and its replicate fix-ups at the strip edges"
        return "other"

    print(f"# ISA-derived VALU budget of {KERNEL}")
    print(f"# steady-state loop of {rows_per_trip} rows per trip, lines {lo}..{hi} of the kernel in {os.path.relpath(path, ROOT)}")
    print("# cost model: ns per wave64 instruction per SIMD, profiles/r01_ubench_valu_rates.txt  " + ", ".join(f"{k} {v}" for k, v in COST.items()))
    print("#")
    print(f"# {'stage':100s} VALU   ns(model)  LDS  SALU  VMEM   classes")
    tot = collections.Counter()
    tot_ns = tot_v = 0
    interior_ns = interior_v = 0
    for sg in segs:
        if sg == ["BARRIER"]:
            print("# ---- s_barrier")
            continue
        w = what(sg)
        c = collections.Counter()
        ns = nl = ng = 0
        for code in sg:
            op = code.split()[0]
            if op.startswith("v_"):
                c[cls(op)] += 1
            elif op.startswith("s_"):
                ns += 1
            elif op.startswith("ds_"):
                nl += 1
            elif op.startswith("global"):
                ng += 1
        nv = sum(c.values())
        t = sum(COST[k] * v for k, v in c.items())
        if w is None:
            if ns or ng:
                print(f"  {'(scalar row arithmetic / stream loads)':100s} {0:4d}  {0.0:8.1f}  {nl:4d} {ns:5d} {ng:5d}")
            continue
        print(f"  {w:100s} {nv:4d}  {t:8.1f}  {nl:4d} {ns:5d} {ng:5d}   " + ", ".join(f"{k.split(' ')[0]} {v}" for k, v in sorted(c.items())))
        tot.update(c)
        tot_ns += t
        tot_v += nv
        if "mirror writes" not in w:
            interior_ns += t
            interior_v += nv
    print("#")
    print(f"# per loop trip ({rows_per_trip} rows): {tot_v} VALU instructions in the loop body, {interior_v} on the path of a strip that touches no image edge "
          f"({interior_v / rows_per_trip:.1f} per wave-row), {interior_ns / rows_per_trip:.0f} ns per wave-row by the cost model")
    print("# classes per loop trip: " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())))
    if fused:
        return
    print("# measured (profiles/r02_pmc_sq_counters.txt): 10.38 M clocks per launch = 1092 clocks = ~590 ns per wave-row and SIMD at 3 waves / SIMD;")
    print("# the model's VALU time is ~72 % of that: the rest is LDS / barrier latency that three waves do not cover, strip and segment halos (10 %).")


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if x != "--fused"]
    fused = "--fused" in sys.argv[1:]
    main(args[0] if args else os.path.join(ROOT, "colorvideovdp_amd", "csrc", "build", ("band4f" if fused else "band4") + "-hip-amdgcn-amd-amdhsa-gfx950.s"), fused)
