"""Planar Y'CbCr (.yuv) input: oracle and host mirror against vectors made by the real reference
(oracle/make_goldens_yuv.py), and the HIP path (cvvdp_process_block_yuv through the package's video_source_yuv_file)
against both.  Tolerances as in test_gpu_parity.py: |dJOD| <= 1e-3, Q_per_ch rtol 2e-4 / atol 2e-6."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, resize_cases, yuv_cases

from oracle import yuv_oracle as yo

JOD_TOL = 1e-3


def _props(g):
    return dict(width=int(g["width"]), height=int(g["height"]), bit_depth=int(g["bit_depth"]), chroma_ss=str(g["chroma_ss"]),
                color_space=str(g["color_space"]))


def _write(g, tmp_path):
    ft, fr = os.path.join(tmp_path, str(g["fname_test"])), os.path.join(tmp_path, str(g["fname_ref"]))
    g["test"].tofile(ft)
    g["ref"].tofile(fr)
    return ft, fr


# ------------------------------------------------------------------ CPU: oracle + host mirror
def test_cases_exist():
    assert len(yuv_cases()) >= 4


@pytest.mark.parametrize("name", yuv_cases())
def test_oracle_rgb_matches_reference(name):
    g = load_golden(name)
    p = _props(g)
    F = int(g["frames"])
    for f, want in ((0, g["rgb_first"]), (F - 1, g["rgb_last"])):
        Y, u, v = yo.split_frame(g["test"], f, p["height"], p["width"], p["chroma_ss"])
        got = yo.frame_to_rgb(Y, u, v, p["bit_depth"], p["chroma_ss"], p["color_space"])
        assert got.shape == want.shape and got.dtype == np.float32
        assert np.abs(got - want).max() <= 3e-7


@pytest.mark.parametrize("name", yuv_cases())
def test_oracle_jod_matches_reference(name):
    from oracle.cvvdp_oracle import Oracle
    g = load_golden(name)
    p = _props(g)
    F = int(g["frames"])
    jod, stats = Oracle(display_name=str(g["display"])).predict(yo.clip_to_rgb(g["test"], p, F), yo.clip_to_rgb(g["ref"], p, F),
                                                                 dim_order="BCFHW", frames_per_second=float(g["fps"]))
    assert abs(float(jod) - float(g["jod"])) <= 1e-4
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("factor,n_in", [(2, 7), (2, 1), (1, 5), (2, 24)])
def test_bilinear_restatement_is_torch_interpolate(factor, n_in):
    rng = np.random.default_rng(factor * 100 + n_in)
    x = rng.standard_normal((1, 1, n_in, n_in)).astype(np.float32)
    want = torch.nn.functional.interpolate(torch.tensor(x), scale_factor=factor, mode="bilinear")[0, 0].numpy() if factor > 1 else x[0, 0]
    i0, i1, l = yo.upsample_axis(n_in * factor, n_in, factor)
    rows = x[0, 0][i0] * (1 - l)[:, None] + x[0, 0][i1] * l[:, None]
    got = rows[:, i0] * (1 - l)[None, :] + rows[:, i1] * l[None, :]
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-7)   # one ulp of the N(0,1) samples (fma vs mul+add)


def test_header_parser_matches_reference_rules():
    from colorvideovdp_amd.video_source_yuv import create_yuv_fname, decode_video_props
    for name in yuv_cases():
        g = load_golden(name)
        got = decode_video_props(str(g["fname_test"]))
        assert got == yo.decode_video_props(str(g["fname_test"]))
        assert (got["width"], got["height"], got["bit_depth"], got["chroma_ss"], got["color_space"], got["fps"]) == \
            (int(g["width"]), int(g["height"]), int(g["bit_depth"]), str(g["chroma_ss"]), str(g["color_space"]), float(g["fps"]))
    # defaults and aliases (video_source_yuv.py:9-16, 37-60)
    assert decode_video_props("clip.yuv") == dict(width=1920, height=1080, fps=24, bit_depth=8, color_space="709", chroma_ss="420")
    v = decode_video_props("/x/y/park_3840x2160p60_10bit_444_hdr.yuv")
    assert (v["width"], v["height"], v["fps"], v["bit_depth"], v["chroma_ss"], v["color_space"]) == (3840, 2160, 60, 10, "444", "2020")
    assert decode_video_props("a_640x480_29.97fps_sdr.yuv")["fps"] == 29.97
    assert decode_video_props(create_yuv_fname("a", v)) == v


def test_reader_geometry_and_raw_frames(tmp_path):
    from colorvideovdp_amd.video_source_yuv import YUVReader
    for name in yuv_cases():
        g = load_golden(name)
        ft, _ = _write(g, str(tmp_path))
        rd = YUVReader(ft)
        p = _props(g)
        assert (rd.height, rd.width, rd.get_frame_count()) == (p["height"], p["width"], int(g["frames"]))
        ys, cs = yo.plane_shapes(p["height"], p["width"], p["chroma_ss"])
        assert rd.y_shape == ys and rd.uv_shape == cs
        f = int(g["frames"]) - 1
        for got, want in zip(rd.get_frame_yuv(f), yo.split_frame(g["test"], f, p["height"], p["width"], p["chroma_ss"])):
            np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(rd.raw_frames(1, 3), g["test"][rd.frame_pixels:3 * rd.frame_pixels])
        with pytest.raises(RuntimeError):
            rd.get_frame_yuv(int(g["frames"]))
    with pytest.raises(FileNotFoundError):
        YUVReader(os.path.join(str(tmp_path), "missing_8x8.yuv"))


# ------------------------------------------------------------------ GPU: the HIP path
@pytest.mark.gpu
@pytest.mark.parametrize("name", yuv_cases())
def test_hip_yuv_matches_reference(name, tmp_path):
    import colorvideovdp_amd as cv
    g = load_golden(name)
    ft, fr = _write(g, str(tmp_path))
    vs = cv.video_source_yuv_file(ft, fr, display_photometry=str(g["display"]))
    assert list(vs.get_video_size()) == [int(g["height"]), int(g["width"]), int(g["frames"])]
    met = cv.cvvdp(display_name=str(g["display"]))
    jod, stats = met.predict_video_source(vs)
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(stats["rho_band"], g["rho_band"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", yuv_cases())
def test_hip_yuv_equals_rgb_path_on_unpacked_frames(name, tmp_path):
    """The in-kernel unpack agrees with feeding the oracle's unpacked R'G'B' frames through the array path."""
    import colorvideovdp_amd as cv
    g = load_golden(name)
    p = _props(g)
    F = int(g["frames"])
    ft, fr = _write(g, str(tmp_path))
    met = cv.cvvdp(display_name=str(g["display"]))
    _, s_yuv = met.predict_video_source(cv.video_source_yuv_file(ft, fr, display_photometry=str(g["display"])))
    _, s_rgb = met.predict(yo.clip_to_rgb(g["test"], p, F), yo.clip_to_rgb(g["ref"], p, F), dim_order="BCFHW", frames_per_second=float(g["fps"]))
    np.testing.assert_allclose(s_yuv["Q_per_ch"], s_rgb["Q_per_ch"], rtol=2e-4, atol=2e-6)   # inputs agree to 1e-7: the parity tolerance


@pytest.mark.gpu
def test_hip_yuv_block_size_and_source_display(tmp_path):
    import colorvideovdp_amd as cv
    g = load_golden("yuv420_8b_709_64x48x10_30")
    ft, fr = _write(g, str(tmp_path))
    q = []
    for nb in (None, 3, 1):
        met = cv.cvvdp(display_name="standard_fhd", block_frames=nb)
        q.append(met.predict_video_source(cv.video_source_yuv_file(ft, fr, display_photometry="standard_fhd"))[1]["Q_per_ch"])
    np.testing.assert_array_equal(q[0], q[1])
    np.testing.assert_array_equal(q[0], q[2])
    # the source's own display photometry is the one applied (video_source.py:206-229), whatever the metric was built with
    met = cv.cvvdp(display_name="standard_hdr_pq")
    met.display_geometry = cv.vvdp_display_geometry.load("standard_fhd")
    met.pix_per_deg = met.display_geometry.get_ppd()
    jod, _ = met.predict_video_source(cv.video_source_yuv_file(ft, fr, display_photometry="standard_fhd"))
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    assert met.display_photometry.EOTF == "PQ"          # restored afterwards
    # frames= and offsets (video_source_yuv.py:271-272, 352-366)
    vs = cv.video_source_yuv_file(ft, fr, display_photometry="standard_fhd", frames=6)
    assert vs.get_video_size()[2] == 6
    jod6, s6 = cv.cvvdp(display_name="standard_fhd").predict_video_source(vs)
    np.testing.assert_allclose(s6["Q_per_ch"], g["Q_per_ch"][:, :, :6], rtol=2e-4, atol=2e-6)   # causal filter: a prefix is exact


@pytest.mark.gpu
def test_hip_yuv_errors(tmp_path):
    import ctypes
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import _capi
    g = load_golden("yuv422_8b_709_64x40x5_24")
    ft, fr = _write(g, str(tmp_path))
    with pytest.raises(RuntimeError):
        cv.video_source_yuv_file(ft, fr, full_screen_resize="bilinear")          # no resize_resolution
    vs = cv.video_source_yuv_file(ft, fr, display_photometry="standard_4k")
    with pytest.raises(NotImplementedError):
        vs.get_test_frame(0, torch.device("cuda"))
    met = cv.cvvdp(display_name="standard_4k")
    met.predict_video_source(vs)                        # configures the handle for this clip
    t, r, fmt = vs.get_raw_yuv_block(0, 2, met.device)
    hist = (ctypes.c_int32 * 8)(*([0] * 8))
    lib = _capi.lib()
    for field, bad in (("chroma", 411), ("bit_depth", 7), ("matrix", 601), ("frame_stride_test", 10)):
        f2 = _capi.YuvFormat.from_buffer_copy(fmt)
        setattr(f2, field, bad)
        assert lib.cvvdp_process_block_yuv(met._handle, t.data_ptr(), r.data_ptr(), ctypes.byref(f2), 0, hist, 2, 0, 0) < 0
        assert lib.cvvdp_last_error(met._handle)


# ------------------------------------------------------------------ full_screen_resize (video_source_yuv.py:266-284, 333-336)
def _side_props(g, side):
    return yo.decode_video_props(str(g["fname_" + side]))


@pytest.mark.parametrize("name", resize_cases())
def test_oracle_resize_matches_reference(name):
    """The oracle's restatement of torch.nn.functional.interpolate against the frames the reference handed to its display model,
    and the whole oracle on the resized clips against the reference's JOD."""
    from oracle.cvvdp_oracle import Oracle
    g = load_golden(name)
    H, W, F, mode = int(g["height"]), int(g["width"]), int(g["frames"]), str(g["mode"])
    clips = {}
    for side in ("test", "ref"):
        clips[side] = yo.clip_to_rgb_resized(g[side], _side_props(g, side), F, H, W, mode)
        assert clips[side].shape == (1, 3, F, H, W)
        assert np.abs(clips[side][0, :, 0] - g[f"rgb_{side}_first"]).max() <= 2e-6
    jod, stats = Oracle(display_name=str(g["display"])).predict(clips["test"], clips["ref"], dim_order="BCFHW", frames_per_second=float(g["fps"]))
    assert abs(float(jod) - float(g["jod"])) <= 1e-4
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("mode", ["nearest", "bilinear", "bicubic", "area"])
def test_resize_restatement_is_torch_interpolate(mode):
    rng = np.random.default_rng(len(mode))
    for hs, ws, hd, wd in ((13, 17, 29, 40), (40, 31, 17, 12), (8, 8, 8, 16), (7, 9, 21, 5), (30, 30, 30, 30)):
        x = rng.random((2, hs, ws)).astype(np.float32) * 1.2 - 0.1
        want = torch.nn.functional.interpolate(torch.tensor(x)[None], size=(hd, wd), mode=mode)[0].clip(0.0, 1.0).numpy()
        assert np.abs(yo.resize_planes(x, hd, wd, mode) - want).max() <= 3e-6


def test_resized_source_geometry(tmp_path):
    from colorvideovdp_amd.video_source_yuv import video_source_yuv_file
    g = load_golden("resize_bicubic_down_up_420_8b")                     # test 72x40, reference 36x20
    ft, fr = _write(g, str(tmp_path))
    with pytest.raises(RuntimeError):
        video_source_yuv_file(ft, fr)                                    # different sizes need a resize
    vs = video_source_yuv_file(ft, fr, full_screen_resize="bicubic", resize_resolution=(54, 30))
    assert vs.get_video_size() == [30, 54, int(g["frames"])] and vs.needs_resize()
    vs = video_source_yuv_file(ft, ft, full_screen_resize="bicubic", resize_resolution=(72, 40))
    assert not vs.needs_resize()                                         # already at the target size: the fused path, like the reference (:333)
    vs = video_source_yuv_file(ft, ft, full_screen_resize="bilinear", resize_resolution=(200, 50), retain_aspect_ratio=True)
    assert vs.get_video_size()[:2] == [50, 90]                           # keeps 72:40 inside 200x50 (:273-282)
    vs = video_source_yuv_file(ft, ft, full_screen_resize="bilinear", resize_resolution=(90, 200), retain_aspect_ratio=True)
    assert vs.get_video_size()[:2] == [50, 90]
    with pytest.raises(RuntimeError):
        video_source_yuv_file(ft, ft, full_screen_resize="lanczos", resize_resolution=(72, 40))


@pytest.mark.gpu
@pytest.mark.parametrize("name", resize_cases())
def test_hip_resize_matches_reference(name, tmp_path):
    """cvvdp_unpack_yuv_resized against the reference's resized frames, then the whole metric on the resized source."""
    import ctypes
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import _capi
    g = load_golden(name)
    H, W, F, mode, disp = int(g["height"]), int(g["width"]), int(g["frames"]), str(g["mode"]), str(g["display"])
    ft, fr = _write(g, str(tmp_path))
    vs = cv.video_source_yuv_file(ft, fr, display_photometry=disp, full_screen_resize=mode, resize_resolution=(W, H))
    assert list(vs.get_video_size()) == [H, W, F]
    met = cv.cvvdp(display_name=disp)
    lib = _capi.lib()
    for side in range(2):
        codes, fmt, sw, sh = vs.get_raw_yuv_side(side, 0, 2, met.device)
        tmp = torch.empty(3 * 2 * sh * sw, dtype=torch.float32, device=met.device)
        rgb = torch.full((3, 2, H, W), -1.0, dtype=torch.float32, device=met.device)
        m = _capi.RESIZE_MODES[mode] if (sw, sh) != (W, H) else _capi.RESIZE_MODES["nearest"]
        rc = lib.cvvdp_unpack_yuv_resized(met._handle, codes.data_ptr(), ctypes.byref(fmt), side, sw, sh, 2, W, H, m, tmp.data_ptr(), rgb.data_ptr(),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _capi.check(met._handle, rc, "cvvdp_unpack_yuv_resized")
        want = g["rgb_test_first" if side == 0 else "rgb_ref_first"]
        assert np.abs(rgb[:, 0].cpu().numpy() - want).max() <= 3e-6
        p = _side_props(g, "ref" if side else "test")
        np.testing.assert_allclose(rgb[:, 1].cpu().numpy(), yo.clip_to_rgb_resized(g["ref" if side else "test"], p, 2, H, W, mode)[0, :, 1], atol=3e-6, rtol=0)
    jod, stats = met.predict_video_source(vs)
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    # blocks of one frame: same frames, same result
    _, s1 = cv.cvvdp(display_name=disp, block_frames=1).predict_video_source(vs)
    np.testing.assert_array_equal(s1["Q_per_ch"], stats["Q_per_ch"])


@pytest.mark.gpu
def test_hip_resize_through_the_command_line_and_errors(tmp_path, capsys):
    import ctypes
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import _capi, cli as rc
    g = load_golden("resize_bilinear_up_420_8b")
    ft, fr = _write(g, str(tmp_path))
    E_ARG = -1   # CVVDP_E_ARG, include/cvvdp_hip.h
    # -f resizes to the display's resolution (run_cvvdp.py:308-309): standard_fhd = 1920x1080
    assert rc.main(["-t", ft, "-r", fr, "-d", "standard_fhd", "-f", "bilinear", "--temp-padding", "replicate", "-q"]) == 0
    got = float(capsys.readouterr().out.strip())
    vs = cv.video_source_yuv_file(ft, fr, display_photometry="standard_fhd", full_screen_resize="bilinear", resize_resolution=(1920, 1080))
    want, stats = cv.cvvdp(display_name="standard_fhd").predict_video_source(vs)
    assert stats["width"] == 1920 and stats["height"] == 1080 and f"{got:.4f}" == f"{want.item():.4f}"
    # a source already at the target size goes the fused route and gives what it gives without the option
    vs_same = cv.video_source_yuv_file(ft, fr, display_photometry="standard_fhd", full_screen_resize="bicubic", resize_resolution=(48, 32))
    _, s_a = cv.cvvdp(display_name="standard_fhd").predict_video_source(vs_same)
    _, s_b = cv.cvvdp(display_name="standard_fhd").predict_video_source(cv.video_source_yuv_file(ft, fr, display_photometry="standard_fhd"))
    np.testing.assert_array_equal(s_a["Q_per_ch"], s_b["Q_per_ch"])
    met = cv.cvvdp(display_name="standard_fhd")
    lib = _capi.lib()
    codes, fmt, sw, sh = vs.get_raw_yuv_side(0, 0, 1, met.device)
    buf = torch.empty(3 * 64 * 96, dtype=torch.float32, device=met.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.cvvdp_unpack_yuv_resized(met._handle, codes.data_ptr(), ctypes.byref(fmt), 0, sw, sh, 1, 96, 64, 7, buf.data_ptr(), buf.data_ptr(), st) == E_ARG
    assert lib.cvvdp_unpack_yuv_resized(met._handle, codes.data_ptr(), ctypes.byref(fmt), 0, sw + 1, sh, 1, 96, 64, 1, buf.data_ptr(), buf.data_ptr(), st) == E_ARG
    assert lib.cvvdp_unpack_yuv_resized(met._handle, None, ctypes.byref(fmt), 0, sw, sh, 1, 96, 64, 1, buf.data_ptr(), buf.data_ptr(), st) == E_ARG
