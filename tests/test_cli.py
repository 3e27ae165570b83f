"""The `cvvdp` command line (colorvideovdp_amd/cli.py) against the reference's command-line contract
(pycvvdp/run_cvvdp.py:83-371): options, output lines, CSV / JSON / PNG side outputs, on committed fixtures."""
import json
import os

import numpy as np
import pytest

from conftest import load_golden

JOD_TOL = 1e-3


def _write_yuv(g, d):
    ft, fr = os.path.join(d, str(g["fname_test"])), os.path.join(d, str(g["fname_ref"]))
    g["test"].tofile(ft)
    g["ref"].tofile(fr)
    return ft, fr


def test_arguments_mirror_the_reference():
    from colorvideovdp_amd import cli as rc
    a = rc.parse_args(["--test", "a.png", "--ref", "b.png"])
    # run_cvvdp.py:88-117 defaults
    assert (a.device, a.heatmap, a.distogram, a.features, a.display, a.nframes, a.metric, a.temp_padding, a.quiet, a.interactive) == \
        ("cuda", "none", -1, False, "standard_4k", -1, ["cvvdp"], "symmetric", False, False)
    a = rc.parse_args(["-t", "a.png", "b.png", "-r", "r.png", "-g", "-x", "-o", "out", "--result", "r.csv", "-d", "standard_fhd", "-q", "--heatmap", "threshold"])
    assert a.test == ["a.png", "b.png"] and a.distogram == 10 and a.features and a.output_dir == "out" and a.result == "r.csv" and a.quiet
    assert rc.parse_args(["-g", "7.5"]).distogram == 7.5
    assert rc.expand_wildcards(["x.png"]) == ["x.png"]


def test_refusals_need_no_gpu(tmp_path, capsys):
    from colorvideovdp_amd import cli as rc
    for extra in (["--device", "cpu"], ["--temp-padding", "valid"], ["--temp-resample"], ["--dump-channels", "lpyr"]):
        assert rc.main(["-t", "a.png", "-r", "b.png"] + extra) == 1          # vq_exception -> logged, exit code 1
    assert rc.main([]) == 0                                                   # "Paths to both ... need to be specified", like the reference
    img = (np.arange(64 * 48 * 3).reshape(48, 64, 3) % 251).astype(np.uint8)
    from PIL import Image
    Image.fromarray(img).save(tmp_path / "a.png")
    assert np.array_equal(rc.load_image_as_array(str(tmp_path / "a.png")), img)
    Image.fromarray(img[..., 0]).save(tmp_path / "g.png")
    assert rc.load_image_as_array(str(tmp_path / "g.png")).shape == (48, 64, 1)


@pytest.mark.gpu
def test_yuv_pair_with_all_side_outputs(tmp_path, capsys):
    from colorvideovdp_amd import cli as rc
    g = load_golden("yuv420_8b_709_64x48x10_30")
    ft, fr = _write_yuv(g, str(tmp_path))
    out = tmp_path / "out"
    rc_code = rc.main(["--test", ft, "--ref", fr, "--display", str(g["display"]), "--temp-padding", "replicate", "--heatmap", "supra-threshold",
                       "--distogram", "--features", "--result", str(tmp_path / "res.csv"), "--output-dir", str(out)])
    assert rc_code == 0
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("cvvdp=")]
    assert len(line) == 1 and line[0].endswith(" [JOD]")                       # run_cvvdp.py:326-330
    jod = float(line[0][len("cvvdp="):-len(" [JOD]")])
    assert abs(jod - float(g["jod"])) <= JOD_TOL and len(line[0].split("=")[1].split()[0].split(".")[1]) == 4
    base = os.path.splitext(os.path.basename(ft))[0]
    csv = open(tmp_path / "res.csv").read().splitlines()
    assert csv[0] == "test, reference, cvvdp" and csv[1].startswith(f"{ft}, {fr}, ") and abs(float(csv[1].split(", ")[2]) - float(g["jod"])) <= JOD_TOL
    fmap = json.load(open(out / f"{base}_fmap.json"))
    assert fmap["N_frames"] == int(g["frames"]) and np.asarray(fmap["t0_b0"]).shape == (1, int(g["frames"]))
    np.testing.assert_allclose(np.asarray(fmap["t3_b1"])[0], g["Q_per_ch"][0, 3, :, 1], rtol=2e-4, atol=2e-6)
    assert (out / f"{base}_distogram.png").stat().st_size > 2000
    frames = sorted(f for f in os.listdir(out) if f.startswith(base + "_heatmap_"))
    assert len(frames) == int(g["frames"]) and frames[0].endswith("_00000.png")
    from PIL import Image
    assert Image.open(out / frames[0]).size == (int(g["width"]), int(g["height"]))


@pytest.mark.gpu
def test_image_pairs_quiet_and_interactive(tmp_path, capsys, monkeypatch):
    import io
    from PIL import Image
    from colorvideovdp_amd import cli as rc
    g = load_golden("img_u8_64x96_fhd_thr")             # an image case with a threshold heat map from the reference
    meta = g["meta"]
    assert meta["dim_order"] == "HWC" and meta["heatmap"] == "threshold"
    H, W = g["test"].shape[:2]
    Image.fromarray(g["test"]).save(tmp_path / "t.png")
    Image.fromarray(g["ref"]).save(tmp_path / "r.png")
    args = ["-t", str(tmp_path / "t.png"), "-r", str(tmp_path / "r.png"), "-d", meta["display"], "-q", "--heatmap", "threshold", "-o", str(tmp_path)]
    assert rc.main(args) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and abs(float(out[0]) - float(g["jod"])) <= JOD_TOL      # --quiet: the number only (run_cvvdp.py:327)
    hm = np.asarray(Image.open(tmp_path / "t_heatmap.png"))
    assert hm.shape == (H, W, 3)
    want = (np.clip(g["heatmap"][0, :, 0].astype(np.float32).transpose(1, 2, 0), 0, 1) * 255).astype(np.uint8)   # np2img, run_cvvdp.py:66-76
    assert (np.abs(hm.astype(int) - want.astype(int)) > 1).mean() < 1e-2
    # a .npy pair, two lines through --interactive
    np.save(tmp_path / "t.npy", g["test"]); np.save(tmp_path / "r.npy", g["ref"])
    line = f"-t {tmp_path / 't.npy'} -r {tmp_path / 'r.npy'} -d {meta['display']} -q\n"
    monkeypatch.setattr("sys.stdin", io.StringIO(line + line))
    assert rc.main(["--interactive"]) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 2 and all(abs(float(o) - float(g["jod"])) <= JOD_TOL for o in out)
    assert rc.main(["-t", str(tmp_path / "t.png"), "-r", str(tmp_path / "clip.mp4")]) == 1   # mixed / unsupported kinds are refused


def test_heatmap_video_writer_pipes_rgb24_frames_into_ffmpeg(tmp_path):
    """HeatmapVideoWriter speaks the reference's ffmpeg protocol (run_cvvdp.py:44-66): rawvideo rgb24 on stdin, size and
    frame rate on the command line, mpeg4 / qscale 3 output.  Checked with a stand-in executable that records both."""
    import stat
    import torch
    from colorvideovdp_amd import heatmap_writers as hw
    fake = tmp_path / "ffmpeg"
    fake.write_text("#!/usr/bin/env python3\nimport sys\nout = sys.argv[-1]\nopen(out + '.args', 'w').write('\\n'.join(sys.argv[1:]))\n"
                    "open(out, 'wb').write(sys.stdin.buffer.read())\n")
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    dest = tmp_path / "out" / "clip_heatmap.mp4"
    w = hw.HeatmapVideoWriter(str(dest), 30, ffmpeg=str(fake))
    rng = np.random.default_rng(3)
    frames = torch.tensor(rng.random((1, 3, 5, 6, 8)).astype(np.float16))
    w(0, frames[:, :, :2])
    w(2, frames[:, :, 2:])
    w.close()
    assert w.frames_written == 5
    args = open(str(dest) + ".args").read().split("\n")
    for flag, val in (("-f", "rawvideo"), ("-pix_fmt", "rgb24"), ("-s", "8x6"), ("-r", "30"), ("-c:v", "mpeg4"), ("-qscale:v", "3")):
        assert args[args.index(flag) + 1] == val, (flag, args)
    got = np.frombuffer(open(dest, "rb").read(), np.uint8).reshape(5, 6, 8, 3)
    np.testing.assert_array_equal(got, hw.heatmap_to_uint8(frames))
    # a failing encoder is an error, not a silent truncation
    bad = tmp_path / "ffmpeg_bad"
    bad.write_text("#!/usr/bin/env python3\nimport sys\nsys.stdin.buffer.read()\nsys.exit(3)\n")
    bad.chmod(bad.stat().st_mode | stat.S_IXUSR)
    w = hw.HeatmapVideoWriter(str(tmp_path / "x.mp4"), 24, ffmpeg=str(bad))
    w(0, frames)
    with pytest.raises(RuntimeError):
        w.close()
    if not hw.HeatmapVideoWriter.available():
        with pytest.raises(FileNotFoundError):
            hw.HeatmapVideoWriter(str(tmp_path / "y.mp4"), 24)


def _write_png16(path, a):
    """uint16 [H, W, 3] -> 16-bit RGB PNG (filter type 0 on every row); Pillow cannot write these."""
    import struct
    import zlib
    h, w, _ = a.shape
    raw = b"".join(b"\0" + a[y].astype(">u2").tobytes() for y in range(h))

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_image_frame_sources_host_logic(tmp_path):
    """video_source_image_frames (video_source_file.py:549-612): name patterns, frame counting, the reference's refusals."""
    from PIL import Image
    from colorvideovdp_amd import cli as rc, video_source_image_frames, vq_exception
    conv = video_source_image_frames.convert_c2python_format_str
    assert conv("a/f_%04d.png") == ("a/f_{:04d}.png", True) and conv("f%d.png") == ("f{:d}.png", True) and conv("plain.png") == ("plain.png", False)
    img = (np.arange(24 * 32 * 3).reshape(24, 32, 3) % 251).astype(np.uint8)
    for k in (0, 1, 2, 3, 5):                                        # frame 4 is missing
        Image.fromarray(img + k).save(tmp_path / f"t_{k:03d}.png")
        Image.fromarray(img).save(tmp_path / f"r_{k:03d}.png")
    tp, rp = str(tmp_path / "t_%03d.png"), str(tmp_path / "r_%03d.png")
    vs = video_source_image_frames(tp, rp, fps=30, display_photometry="standard_fhd")
    assert vs.get_video_size() == (24, 32, 4) and vs.get_frames_per_second() == 30 and list(vs.frame_range) == [0, 1, 2, 3]
    t, r, code = vs.get_raw_block(1, 3, "cpu")
    assert code == 0 and t.shape == (1, 3, 2, 24, 32) and np.array_equal(t[0, :, 1].numpy().transpose(1, 2, 0), img + 2) and np.array_equal(r[0, :, 0].numpy().transpose(1, 2, 0), img)
    vs = video_source_image_frames(tp, rp, fps=30, frame_range=rc.parse_frame_range("1:2:9"), display_photometry="standard_fhd")
    assert list(vs.frame_range) == [1, 3, 5]                             # 1:2:9 steps over the missing frame 4; frame 7 ends the run
    assert list(rc.parse_frame_range("2:5")) == [2, 3, 4, 5] and list(rc.parse_frame_range("3:")[:2]) == [3, 4] and rc.parse_frame_range(None) is None
    for args, kw in (((tp, str(tmp_path / "r_000.png")), dict(fps=30)), ((tp, rp), dict(fps=0)), ((str(tmp_path / "t_000.png"), str(tmp_path / "r_000.png")), dict(fps=30)),
                     ((str(tmp_path / "x_%03d.png"), rp), dict(fps=30)), ((tp, rp), dict(fps=30, full_screen_resize="bilinear"))):
        with pytest.raises(vq_exception):
            video_source_image_frames(*args, display_photometry="standard_fhd", **kw)
    a16 = (np.arange(10 * 12 * 3).reshape(10, 12, 3) * 181 % 65536).astype(np.uint16)
    _write_png16(tmp_path / "p16.png", a16)
    assert np.array_equal(rc.load_image_as_array(str(tmp_path / "p16.png")), a16)
    vs = video_source_image_frames(str(tmp_path / "p16.png"), str(tmp_path / "p16.png"), display_photometry="standard_fhd")
    t, r, code = vs.get_raw_block(0, 1, "cpu")
    assert vs.get_video_size() == (10, 12, 1) and code == 1 and t.dtype.is_floating_point is False and np.array_equal(t[0, :, 0].numpy().view(np.uint16).transpose(1, 2, 0), a16)
    with pytest.raises(FileNotFoundError):
        video_source_image_frames(str(tmp_path / "nope.png"), str(tmp_path / "p16.png"), display_photometry="standard_fhd").get_video_size()


@pytest.mark.gpu
def test_numbered_image_frames_are_a_clip(tmp_path, capsys):
    """A clip stored as numbered frames (-t t_%03d.png --fps 30) scores like the same clip handed over as an array: against the
    reference's JOD for 8- and 16-bit frames (symmetric padding), and bit-exactly against predict() for a --frames subset."""
    from PIL import Image
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import cli as rc
    g = load_golden("vid_u16_67x121x20_30_4k_sym")
    for k in range(g["test"].shape[0]):
        _write_png16(tmp_path / f"t16_{k:04d}.png", g["test"][k])
        _write_png16(tmp_path / f"r16_{k:04d}.png", g["ref"][k])
    assert rc.main(["-t", str(tmp_path / "t16_%04d.png"), "-r", str(tmp_path / "r16_%04d.png"), "--fps", "30", "-d", "standard_4k", "-q"]) == 0   # symmetric = CLI default
    assert abs(float(capsys.readouterr().out.strip()) - float(g["jod"])) <= JOD_TOL
    g = load_golden("vid_u8_72x128x12_60_fhd")
    for k in range(g["test"].shape[0]):
        Image.fromarray(g["test"][k]).save(tmp_path / f"t_{k:03d}.png")
        Image.fromarray(g["ref"][k]).save(tmp_path / f"r_{k:03d}.png")
    common = ["-t", str(tmp_path / "t_%03d.png"), "-r", str(tmp_path / "r_%03d.png"), "--fps", "60", "-d", "standard_fhd", "--temp-padding", "replicate", "-q"]
    assert rc.main(common) == 0
    assert abs(float(capsys.readouterr().out.strip()) - float(g["jod"])) <= JOD_TOL
    assert rc.main(common + ["--frames", "1:2:9"]) == 0
    got = float(capsys.readouterr().out.strip())
    m = cv.cvvdp(display_name="standard_fhd", temp_padding="replicate")
    want, _ = m.predict(g["test"][1:10:2], g["ref"][1:10:2], dim_order="FHWC", frames_per_second=60)
    assert f"{got:.4f}" == f"{want.item():.4f}"
    # the class directly, small blocks (frames are decoded block by block, the core keeps the temporal history)
    vs = cv.video_source_file(str(tmp_path / "t_%03d.png"), str(tmp_path / "r_%03d.png"), fps=60, display_photometry="standard_fhd")
    q1, s1 = cv.cvvdp(display_name="standard_fhd", temp_padding="replicate", block_frames=5).predict_video_source(vs)
    q2, s2 = m.predict(g["test"], g["ref"], dim_order="FHWC", frames_per_second=60)
    assert np.array_equal(s1["Q_per_ch"], s2["Q_per_ch"]) and q1.item() == q2.item()
