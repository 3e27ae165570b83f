"""CPU-only checks: the C-ABI library loads and exports everything include/cvvdp_hip.h declares,
host-side set-up math against the reference vectors, frame-source reshaping, shard planning."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from colorvideovdp_amd import _capi
    lib = _capi.lib()
    hdr = open(os.path.join(ROOT, "include", "cvvdp_hip.h")).read()
    declared = set(re.findall(r"\b(cvvdp_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"cvvdp_handle"}
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), name
        assert name in _capi.SYMBOLS, f"{name} not bound in _capi.py"
    assert set(_capi.SYMBOLS) == declared
    assert lib.cvvdp_abi_version() == _capi.ABI_VERSION
    p, c = ctypes.c_int32(), ctypes.c_int32()
    lib.cvvdp_struct_sizes(ctypes.byref(p), ctypes.byref(c))
    assert (p.value, c.value) == (ctypes.sizeof(_capi.Params), ctypes.sizeof(_capi.Clip))


def test_abi_argument_validation_without_gpu():
    from colorvideovdp_amd import _capi
    lib = _capi.lib()
    h = ctypes.c_void_p()
    assert lib.cvvdp_create(None, ctypes.byref(h)) == -1
    P = _capi.Params()
    assert lib.cvvdp_create(ctypes.byref(P), ctypes.byref(h)) == 0
    clip = _capi.Clip()
    assert lib.cvvdp_configure(h, ctypes.byref(clip)) == -1  # empty geometry
    assert b"geometry" in lib.cvvdp_last_error(h)
    assert lib.cvvdp_workspace_bytes(h) == 0
    clip.batch, clip.channels, clip.height, clip.width, clip.is_video, clip.n_frames, clip.n_levels = 1, 3, 1080, 1920, 1, 64, 8
    clip.filter_len, clip.block_frames = 17, 16
    assert lib.cvvdp_configure(h, ctypes.byref(clip)) == 0
    need = lib.cvvdp_workspace_bytes(h)
    P0 = 1080 * 1920
    # DKL tail (6 planes x 16 frames) + one pyramid set of 8 planes x 16 frames x 4/3
    assert need > (6 * 16 + 8 * 16 * 1.33) * P0 * 4 and need < (6 * 16 + 8 * 16 * 1.35) * P0 * 4 + (1 << 22)
    assert lib.cvvdp_process_block(h, None, None, 3, None, None, 0, None, 1, 0, None) == -2  # no workspace bound
    lib.cvvdp_destroy(h)


def test_host_setup_against_reference_vectors(setup_vectors):
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import host_setup as hs
    s = setup_vectors
    m = cv.cvvdp(display_name="standard_4k")
    for fps in (24, 25, 30, 50, 60, 120):
        F = hs.temporal_filters(fps, m.parameters["beta_tf"], m.parameters["sigma_tf"])
        np.testing.assert_allclose(F, s["taps_%d" % fps], atol=2e-7)
    for i, rho in enumerate(s["csf_rhos"]):
        np.testing.assert_allclose(m.csf_table.rows(rho), s["csf_rows"][i], rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(m.display_photometry.rgb2dkl_fp32(), s["dkl_standard_4k"])
    np.testing.assert_allclose(np.array(m.display_photometry.get_black_level()), s["black_standard_4k"], rtol=1e-15)
    hdr = cv.vvdp_display_photometry.load("standard_hdr_pq", [])
    np.testing.assert_array_equal(hdr.rgb2dkl_fp32(), s["dkl_standard_hdr_pq"])
    for row, disp in zip(s["band_sizes"], s["band_displays"]):
        W, H, ppd, nb = int(row[0]), int(row[1]), row[2], int(row[3])
        g = cv.vvdp_display_geometry.load(str(disp))
        assert abs(g.get_ppd() - ppd) < 1e-9
        h, fr = hs.band_frequencies(W, H, g.get_ppd())
        assert h + 1 == nb
        np.testing.assert_allclose(fr, row[4:4 + nb], rtol=1e-12)
    assert m.get_info_string() == '"ColorVideoVDP v0.5.6, 75.4 [pix/deg], Lpeak=200, Lblack=0.2, Lrefl=0.3979 [cd/m^2], (standard_4k)"'
    assert m.short_name() == "cvvdp" and m.quality_unit() == "JOD"
    assert cv.vq_metric_dict["cvvdp"] is cv.cvvdp


def test_symmetric_index_matches_reference_formula():
    from colorvideovdp_amd import host_setup as hs
    # frame -k maps to frame k; short clips ping-pong (cvvdp_metric.py:445-450)
    assert [hs.symmetric_frame_index(-k, 30) for k in range(1, 6)] == [1, 2, 3, 4, 5]
    assert [hs.symmetric_frame_index(-k, 4) for k in range(1, 9)] == [1, 2, 3, 2, 1, 0, 1, 2]


def test_reshuffle_dims_and_array_source():
    import colorvideovdp_amd as cv
    x = torch.arange(2 * 3 * 4 * 5).reshape(4, 5, 3, 2)  # H W C F
    y = cv.reshuffle_dims(x, "HWCF", "BCFHW")
    assert tuple(y.shape) == (1, 3, 2, 4, 5)
    assert y[0, 1, 1, 2, 3] == x[2, 3, 1, 1]
    a = np.zeros((6, 8, 3), dtype=np.uint16)
    a[1, 2, 0] = 65535
    vs = cv.video_source_array(a, a, 0, dim_order="HWC")
    t, r, code = vs.raw_arrays()
    assert code == 1 and t.dtype == torch.int16 and tuple(t.shape) == (1, 3, 1, 6, 8) and t[0, 0, 0, 1, 2] == -1
    assert vs.get_video_size() == (6, 8, 1) and vs.get_batch_size() == 1
    with pytest.raises(RuntimeError):
        cv.video_source_array(np.zeros((2, 6, 8, 3), np.uint8), np.zeros((2, 6, 8, 3), np.uint8), 0, dim_order="FHWC")
    with pytest.raises(RuntimeError):
        cv.video_source_array(np.zeros((6, 8, 3), np.uint8), np.zeros((6, 9, 3), np.uint8), 0, dim_order="HWC")


def test_cpu_device_is_rejected_loudly():
    import colorvideovdp_amd as cv
    with pytest.raises(RuntimeError, match="no CPU path"):
        cv.cvvdp(device="cpu")
    if not torch.cuda.is_available():
        m = cv.cvvdp(display_name="standard_fhd")
        x = np.zeros((16, 16, 3), np.uint8)
        with pytest.raises(RuntimeError, match="no HIP device"):
            m.predict(x, x, dim_order="HWC")


def test_config_paths_override(tmp_path):
    import json
    import colorvideovdp_amd as cv
    models = {"my_display": {"name": "custom", "resolution": [1000, 500], "viewing_distance_meters": 1.0, "diagonal_size_inches": 20,
                             "max_luminance": 321, "contrast": 100, "E_ambient": 0}}
    f = tmp_path / "display_models_custom.json"
    f.write_text(json.dumps(models))
    m = cv.cvvdp(display_name="my_display", config_paths=[str(f)])
    assert m.display_photometry.get_peak_luminance() == 321
    assert "custom-display: my_display" in m.get_info_string()
    with pytest.raises(RuntimeError):
        cv.cvvdp(display_name="my_display")


def test_shard_plan_covers_clip():
    from colorvideovdp_amd.sharding import plan_frame_shard
    for n in (1, 7, 64, 1024, 1023):
        for world in (1, 2, 3, 8):
            ranges = [plan_frame_shard(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n
            for (a, ca), (b, _) in zip(ranges, ranges[1:]):
                assert a + ca == b
            assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1


def test_heatmap_uint8_conversion_is_the_reference_writers_float16_product():
    """np2vid / np2img (run_cvvdp.py:59-63, :76) multiply the FLOAT16 heat map by 255.0, which stays float16: the product is
    rounded to half before it is truncated.  Every fp16 value in [-0.5, 1.5] against that literal expression."""
    import torch
    from colorvideovdp_amd import heatmap_writers as hw
    bits = np.arange(0, 1 << 16, dtype=np.uint16).view(np.float16)
    v = bits[np.isfinite(bits) & (bits >= -0.5) & (bits <= 1.5)]
    n = (v.size // 3) * 3
    frames = torch.from_numpy(v[:n].reshape(1, 3, 1, 1, n // 3).copy())
    got = hw.heatmap_to_uint8(frames)                                           # [1, 1, n/3, 3]
    want = (np.clip(frames[0].permute(1, 2, 3, 0).numpy(), 0.0, 1.0) * 255.0).astype(np.uint8)
    np.testing.assert_array_equal(got, want)
    f32 = (np.clip(frames[0].permute(1, 2, 3, 0).float().numpy(), 0.0, 1.0) * 255.0).astype(np.uint8)
    assert (f32 != want).any()                                                  # the fp32 product is NOT the same thing


def test_product_library_has_no_development_knobs_and_the_binding_loads_only_it(tmp_path):
    """tuning knobs that change launch geometry (and the last bits of Q_per_ch) are compiled out of the product
    build; CVVDP_LIB (load another library) is honoured only together with CVVDP_DEV_KNOBS=1."""
    import subprocess
    import sys
    from colorvideovdp_amd import _capi
    assert _capi.lib().cvvdp_build_flags() & _capi.BUILD_DEV_KNOBS == 0
    src = open(os.path.join(ROOT, "colorvideovdp_amd", "csrc", "core.cpp")).read()
    assert "getenv" not in src                                   # every knob goes through kernels.h::dev_knob
    code = "from colorvideovdp_amd import _capi; print(_capi.LIB_PATH)"
    env = dict(os.environ, CVVDP_LIB=str(tmp_path / "other.so"), PYTHONPATH=ROOT)
    env.pop("CVVDP_DEV_KNOBS", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900).stdout.strip()
    assert out.endswith(os.path.join("colorvideovdp_amd", "libcvvdp_hip.so"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, CVVDP_DEV_KNOBS="1"), capture_output=True, text=True, timeout=900).stdout.strip()
    assert out == str(tmp_path / "other.so")


def test_a_library_with_a_timing_only_switch_identifies_itself_and_is_refused_as_product(tmp_path):
    """band4s.hip carries timing-only switches that compute wrong results (S_DIAG_*).  A library compiled with one
    reports CVVDP_BUILD_DIAG through cvvdp_build_flags() and the binding refuses it unless a development library was asked for; and
    `make` does not take EXTRA flags from the environment."""
    import subprocess
    import sys
    from colorvideovdp_amd import _capi
    assert _capi.lib().cvvdp_build_flags() == 0                              # the in-tree product build: nothing compiled in
    csrc = os.path.join(ROOT, "colorvideovdp_amd", "csrc")
    # (1) an exported EXTRA changes no compile command of the product build
    dry = subprocess.run(["make", "-C", csrc, "-n", "-B"], env=dict(os.environ, EXTRA="-DS_DIAG_NOBAR"), capture_output=True, text=True, timeout=120)
    assert dry.returncode == 0 and "hipcc" in dry.stdout and "S_DIAG_NOBAR" not in dry.stdout
    dry = subprocess.run(["make", "-C", csrc, "-n", "-B", "EXTRA=-DS_DIAG_NOBAR"], capture_output=True, text=True, timeout=120)
    assert "S_DIAG_NOBAR" in dry.stdout                                      # (the command line still can: that is explicit)
    # (2) such a library says what it is, and loads only as a development library
    p = subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), "test_diag_nobar", "band4s.hip", "-DS_DIAG_NOBAR"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    variant = os.path.join(ROOT, "variants", "test_diag_nobar.so")
    code = ("import sys\nfrom colorvideovdp_amd import _capi\n_capi.LIB_PATH = sys.argv[1]\n"
            "try:\n    print('flags', _capi.lib().cvvdp_build_flags())\nexcept ImportError as e:\n    print('refused:', e)\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("CVVDP_DEV_KNOBS", None)
    out = subprocess.run([sys.executable, "-c", code, variant], env=env, capture_output=True, text=True, timeout=900).stdout
    assert out.startswith("refused:") and "not a product build" in out and "= 4" in out
    out = subprocess.run([sys.executable, "-c", code, variant], env=dict(env, CVVDP_DEV_KNOBS="1"), capture_output=True, text=True, timeout=900).stdout
    assert out.strip() == f"flags {_capi.BUILD_DIAG}"
    safe = os.path.join(ROOT, "colorvideovdp_amd", "libcvvdp_hip_safe.so")
    if os.path.isfile(safe) and os.path.getmtime(safe) >= os.path.getmtime(os.path.join(csrc, "kernels.h")):
        out = subprocess.run([sys.executable, "-c", code, safe], env=env, capture_output=True, text=True, timeout=900).stdout
        assert out.startswith("refused:") and "= 2" in out


def test_heatmap_block_policy_keeps_gpu_mem_a_hard_cap(monkeypatch):
    """for heat-map clips resident in HBM the block length was clamped UP to the 16-frame piece even when the user's gpu_mem
    cap (or the free memory) allowed fewer frames.  The long-block rule now applies only when it yields more than a piece."""
    import torch
    import colorvideovdp_amd as cv
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (int(200e9), int(288e9)))
    m = cv.cvvdp(display_name="standard_4k", heatmap="threshold")
    pix8k, pix_fhd = 7680 * 4320, 1920 * 1080
    big = m._pick_block_frames(pix8k, 1, 256, 17, 4, False, True)
    assert 16 < big <= 64                                     # plenty of memory: a long temporal block scored in 16-frame pieces
    m.gpu_mem = 6.0                                           # 6 GB: an 8K frame's planes alone are ~1.4 GB
    capped = m._pick_block_frames(pix8k, 1, 256, 17, 4, False, True)
    per_frame = pix8k * (2 * 4 * 4 * 1.34) + pix8k * 16
    assert capped < 16 and capped * per_frame <= 6.0e9
    assert m._pick_block_frames(pix_fhd, 1, 256, 17, 4, False, True) == 64      # the same cap is no constraint at 1080p
