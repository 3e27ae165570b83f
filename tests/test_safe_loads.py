"""The band kernels (band4.hip, band4f.hip, band4s.hip) issue their streamed loads from inline assembly and wait for them with
hand-counted `s_waitcnt vmcnt(n)`: the compiler does not know those registers are in flight.  `make` checks the generated assembly
statically (tools/check_band4_isa.py); this is the dynamic check: `make safe` builds the same kernels with ordinary loads the
compiler tracks itself (libcvvdp_hip_safe.so, -DCVVDP_SAFE_LOADS), and the product build has to compute the same BITS -- Q_per_ch and the
level-1 / level-2 planes, the heat map of the fused heat kernels, the pooled features of the fused feature kernels -- on an aligned 4K block,
a W % 4 == 2 frame, a small ragged frame and a FULL 64-frame 4K block (every CU full for many rounds), on both band routes and both wave layouts."""
import os
import subprocess
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "colorvideovdp_amd", "csrc")
SAFE = os.path.join(ROOT, "colorvideovdp_amd", "libcvvdp_hip_safe.so")


def _safe_library():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp"))] + [os.path.join(ROOT, "include", "cvvdp_hip.h")]
    if not os.path.isfile(SAFE) or os.path.getmtime(SAFE) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", CSRC, "-j", "8", "safe"], check=True, capture_output=True, timeout=900)
    return SAFE


@pytest.mark.gpu
def test_product_build_computes_what_the_compiler_scheduled_build_computes(tmp_path):
    sys.path.insert(0, os.path.dirname(__file__))
    import safe_loads_worker as w
    from colorvideovdp_amd import _capi
    assert os.path.basename(_capi.LIB_PATH) == "libcvvdp_hip.so"
    mine = w.run()                                                     # the product library, in this process
    out = tmp_path / "safe.npz"
    env = dict(os.environ, CVVDP_DEV_KNOBS="1", CVVDP_LIB=_safe_library(), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "safe_loads_worker.py"), str(out)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    safe = np.load(out)
    assert os.path.basename(str(safe["lib"])) == "libcvvdp_hip_safe.so"
    keys = [k for k in mine if k != "lib"]
    assert set(keys) == set(safe.files) - {"lib"}
    # every hand-loaded kernel at both occupancies: k_band4 (unfused), k_band4f<4,0> (one wave per channel), k_band4s, k_band4s_heat,
    # k_band4s_feat -- on a 3-frame and on a 64-frame 4K block
    for name in ("aligned_4k", "full_4k64"):
        for route in ("unfused", "fused_one_wave", "fused_split"):
            assert {f"{name}.{route}.{x}" for x in ("jod", "q", "fused_levels", "g1", "g2")} <= set(keys)
        assert f"{name}.heat.map" in keys and f"{name}.heat.q" in keys and int(mine[f"{name}.feat.n_bands"]) >= 8
    assert int(mine["full_4k64.fused_split.fused_levels"]) >= 3      # (fuse_mode 1 fuses every level that can be; the product picks 3 here)
    for k in keys:
        np.testing.assert_array_equal(mine[k], safe[k], err_msg=k)
    # the routes that were asked for did run
    for name, *_ in w.CASES:
        assert int(mine[f"{name}.fused_split.fused_levels"]) >= 1 and int(mine[f"{name}.unfused.fused_levels"]) == 0


def test_makefile_has_the_safe_target():
    text = open(os.path.join(CSRC, "Makefile")).read()
    assert "safe:" in text and "-DCVVDP_SAFE_LOADS" in text
    for f in ("band4.hip", "band4f.hip", "band4s.hip"):
        assert "CVVDP_SAFE_LOADS" in open(os.path.join(CSRC, f)).read(), f
