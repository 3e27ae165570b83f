"""The band kernel's hand-managed stream loads (csrc/band4.hip, STREAM LOADS) are safe only if no instruction touches a
load's destination registers between the load and the explicit s_waitcnt that covers it -- something the compiler does not
know about.  This test compiles band4.hip to gfx950 assembly (hipcc cross-compiles without a GPU) and runs the checker of
tools/check_band4_isa.py over every instantiation."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
def test_no_instruction_touches_a_register_with_a_load_in_flight(tmp_path):
    asm = tmp_path / "band4.s"
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-x", "hip",
           "--cuda-device-only", "-S", os.path.join(ROOT, "colorvideovdp_amd", "csrc", "band4.hip"), "-o", str(asm)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    spec = importlib.util.spec_from_file_location("check_band4_isa", os.path.join(ROOT, "tools", "check_band4_isa.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    text = asm.read_text().split("\n")
    starts = [i for i, l in enumerate(text) if l.startswith("_ZN5cvvdp7k_band4") and l.rstrip().split(";")[0].rstrip().endswith(":")]
    assert len(starts) == 20                      # NCH x HEAT x RAGGED x DUMP + the four FEAT instantiations (NCH x RAGGED)
    for s in starts:
        e = next(i for i in range(s, len(text)) if ".end_amdhsa_kernel" in text[i] or text[i].startswith("\t.section"))
        bad, n_loads, n_loops = chk.check_kernel(text[s].split(":")[0], text[s:e])
        assert n_loads > 0 and bad == 0, (text[s], bad)
        # and the steady-state loop waits with the exact counts, not with a drain
        body = "\n".join(text[s:e])
        assert "s_waitcnt vmcnt(3)" in body and "s_waitcnt vmcnt(2)" in body


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
def test_fused_kernel_streams_are_safe_too(tmp_path):
    """k_band4f (band4f.hip) keeps six loads per row in flight across two barriers and an 8-row register ring; same check."""
    asm = tmp_path / "band4f.s"
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-x", "hip",
           "--cuda-device-only", "-S", os.path.join(ROOT, "colorvideovdp_amd", "csrc", "band4f.hip"), "-o", str(asm)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    spec = importlib.util.spec_from_file_location("check_band4_isa", os.path.join(ROOT, "tools", "check_band4_isa.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    text = asm.read_text().split("\n")
    starts = [i for i, l in enumerate(text) if l.startswith("_ZN5cvvdp8k_band4f") and l.rstrip().split(";")[0].rstrip().endswith(":")]
    assert len(starts) == 3                                           # strips inside the image / border strips / border strips of a W % 4 == 2 level
    for s in starts:
        e = next(i for i in range(s, len(text)) if ".end_amdhsa_kernel" in text[i] or text[i].startswith("\t.section"))
        bad, n_loads, n_loops = chk.check_kernel(text[s].split(":")[0], text[s:e])
        if "k_band4fILi4ELi1EE" in text[s]:
            # the plain EDGE == 1 instantiation (border strips of the one-wave A/B layout only) has compiler-managed loads since round 6
            # (band4f.hip, F_SAFE): nothing hand-issued is left in it
            body = text[s:e]
            assert sum(1 for i, l in enumerate(body) if i > 0 and "global_load_dword" in l and "ASMSTART" in body[i - 1]) == 0
            continue
        assert n_loads == 48 and bad == 0, (n_loads, bad)          # eight steps x (four neighbour loads + two row loads)
        assert "\n".join(text[s:e]).count("s_waitcnt vmcnt(2)") >= 8


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="hipcc not available")
def test_front_back_wave_kernels_are_safe_and_fit_four_waves_per_simd(tmp_path):
    """k_band4s / k_band4s_heat / k_band4s_feat (band4s.hip): the front waves' 8-row ring with six loads per step and ONE wait per step
    (vmcnt(8); the border-strip kernels k_band4s_edge / _edge_heat: a 6-row ring and vmcnt(6)); 128 VGPRs at most, or the two 8-wave blocks per CU (four waves per SIMD) the layout exists for do not fit; the
    k_band4f_heat / _feat instantiations must hold no hand-issued load at all (compiler-managed: they spill)."""
    import re
    spec = importlib.util.spec_from_file_location("check_band4_isa", os.path.join(ROOT, "tools", "check_band4_isa.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    asm = tmp_path / "band4s.s"
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-x", "hip",
           "--cuda-device-only", "-S", os.path.join(ROOT, "colorvideovdp_amd", "csrc", "band4s.hip"), "-o", str(asm)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    raw = asm.read_text()
    text = raw.split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^_ZN5cvvdp\d+k_band4s(_edge)?(_heat|_feat)?E.*:\s*(;.*)?$", l)]
    assert len(starts) == 5
    for s in starts:
        e = next(i for i in range(s, len(text)) if ".end_amdhsa_kernel" in text[i] or text[i].startswith("\t.section"))
        bad, n_loads, n_loops = chk.check_kernel(text[s].split(":")[0], text[s:e])
        # eight steps x (four neighbour loads + two row loads) in the border-free kernels, one vmcnt(8) per step; the border-strip kernels
        # (round 5: k_band4s_edge / _edge_heat, a ring of six rows): six steps of six loads, one vmcnt(6) each
        kname = text[s].split(":")[0]
        edge = "_edge" in kname
        feat = "_feat" in kname
        assert n_loads == (36 if edge else 48) and bad == 0, (text[s], n_loads, bad)
        body = "\n".join(text[s:e])
        assert body.count("s_waitcnt vmcnt(6)") >= 6 if edge else body.count("s_waitcnt vmcnt(8)") >= 8
        assert "scratch_" not in body or kname not in ("_ZN5cvvdp8k_band4sENS_8BandArgsE", "_ZN5cvvdp13k_band4s_edgeENS_8BandArgsE")   # the plain kernels do not spill
        # front and back waves run their own copies of the row loop and meet at s_barrier, which counts arrivals -- both roles must
        # execute the same number of barriers per row.  On the generated code: every loop that holds barriers holds exactly two per row
        # it advances -- the front's unrolled ring (8 rows: 16; the border body's 6 rows: 12), the back's even / odd row pair (4), and the
        # one-row loops of the reflected rows below the image in either role (2).
        body_lines = text[s:e]
        labels = {l.split(":")[0]: i for i, l in enumerate(body_lines) if re.match(r"^\.LBB\d+_\d+:", l)}
        n_front = 0
        for i, l in enumerate(body_lines):
            mm = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                loop = [x.split(";")[0].strip() for x in body_lines[labels[mm.group(1)]:i + 1]]
                nb = sum(1 for x in loop if x.startswith("s_barrier"))
                if not nb:
                    continue
                assert nb in (2, 4, 12, 16), (text[s].split(":")[0], nb)
                steps = sum(1 for x in loop if re.match(r"s_waitcnt vmcnt\((8|6)\)$", x))      # the front's one wait per row step
                if any(x.startswith("global_load_dwordx4") for x in loop):
                    assert nb == 2 * steps and steps in (6, 8), (text[s].split(":")[0], nb, steps)
                    n_front += 1
        assert n_front == 1
    vg = [int(v) for v in re.findall(r"\.vgpr_count:\s+(\d+)", raw)]
    assert len(vg) == 5 and max(vg) <= 128, vg
    assert chk.check_file(str(asm)) == 0
