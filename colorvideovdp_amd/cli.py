"""Command line of the MI355X build: `python -m colorvideovdp_amd` / `cvvdp` (console entry in pyproject.toml).

Mirrors the reference's command line (pycvvdp/run_cvvdp.py:83-118 arguments, :120-371 run_on_args) for the path this build
implements: the `cvvdp` metric on image pairs (PNG / JPEG / anything Pillow reads, 8 or 16 bit), planar .yuv clips (the
file name carries size, frame rate, bit depth and chroma format, video_source_yuv.py:8-62) and .npy arrays.  Same options,
same output lines (`cvvdp=9.1234 [JOD]`, or only the number with --quiet), same side outputs (--result CSV, --features
JSON, --distogram PNG, --heatmap).  Differences, because this image has no ffmpeg and the build is GPU-only:
  * compressed video files (.mp4, .mkv, ...) are refused with a hint to decode them to .yuv first;
  * the heat map of a VIDEO is streamed block by block: into `<base>_heatmap.mp4` through an ffmpeg pipe (the reference's
    file and codec settings) where an `ffmpeg` executable exists, otherwise into a numbered PNG sequence
    `<base>_heatmap_%05d.png` (`ffmpeg -i <base>_heatmap_%05d.png <base>_heatmap.mp4` converts it); an image gives `<base>_heatmap.png`;
  * --device must be a cuda device; --temp-padding 'valid', --temp-resample, --dump-channels and metrics other than cvvdp are
    not available; --full-screen-resize works for .yuv clips (as in the reference it is not implemented for images).
Clips stored as numbered image frames work as in the reference: `-t t_%04d.png -r r_%04d.png --fps 30 [--frames 10:2:50]`.
"""
import argparse
import glob
import logging
import os
import shlex
import sys
import traceback

import numpy as np
import torch

from . import heatmap_writers
from .cvvdp_metric import cvvdp
from .display_model import vvdp_display_geometry, vvdp_display_photometry
from .video_source_file import IMAGE_EXT, VIDEO_EXT, load_image_as_array, video_source_file
from .vq_metric import vq_exception, vq_metric_dict



def expand_wildcards(filestrs):
    """run_cvvdp.py:31-41."""
    if not isinstance(filestrs, list):
        return [filestrs]
    files = []
    for fs in filestrs:
        files += sorted(glob.glob(fs)) if "*" in fs else [fs]
    return files


# The reference's options (run_cvvdp.py:83-118): (flags, argparse keywords).  Kept as data: the names, defaults and arities are
# the contract existing scripts rely on; the help texts say what THIS build does with them.
_NA = "not available in this build"
_OPTIONS = (
    (("-t", "--test"), dict(type=str, nargs="+", help="test images / clips (wildcards allowed)")),
    (("-r", "--ref"), dict(type=str, nargs="+", help="reference images / clips; one reference may serve many tests and vice versa")),
    (("--device",), dict(type=str, default="cuda", help="'cuda' or 'cuda:N' (there is no CPU path)")),
    (("--heatmap",), dict(type=str, default="none", help="difference map: none, raw, threshold or supra-threshold")),
    (("-g", "--distogram"), dict(type=float, default=-1, const=10, nargs="?", help="write a distogram; optional value = JOD at the top of the colour scale")),
    (("-x", "--features"), dict(action="store_true", default=False, help="write the per-band features as JSON")),
    (("-o", "--output-dir"), dict(type=str, default=None, help="where heat maps, distograms and feature files go (default: here)")),
    (("--result",), dict(type=str, default=None, help="CSV file for the predictions")),
    (("-c", "--config-paths"), dict(type=str, nargs="+", default=[], help="extra configuration files / directories (display_models.json, ...)")),
    (("-d", "--display"), dict(type=str, default="standard_4k", help="display model name; ? lists them")),
    (("-n", "--nframes"), dict(type=int, default=-1, help="use only the first N frames")),
    (("--count-frames",), dict(action="store_true", default=False, help="accepted for compatibility (frame counts of .yuv / .npy inputs are exact)")),
    (("-f", "--full-screen-resize"), dict(choices=["bilinear", "bicubic", "nearest", "area"], default=None,
                                          help="resize test and reference to the display's resolution (.yuv clips; on the GPU, torch.nn.functional.interpolate semantics)")),
    (("-m", "--metric"), dict(nargs="+", default=["cvvdp"], help="metric(s); this build registers cvvdp")),
    (("--temp-padding",), dict(choices=["replicate", "symmetric", "valid"], default="symmetric", help="padding before the first frame ('valid': " + _NA + ")")),
    (("--pix-per-deg",), dict(type=float, default=None, help="override the display geometry")),
    (("--fps",), dict(type=float, default=None, help="frame rate: needed for .npy clips and numbered image frames (name_%%04d.png), overrides a .yuv file name")),
    (("--frames",), dict(type=str, default=None, help="frames of an image sequence to use: first:step:last, first:last or first: (both ends included)")),
    (("--gpu-mem",), dict(type=float, default=None, help="GPU memory budget in GB")),
    (("-q", "--quiet"), dict(action="store_true", default=False, help="print the JOD value only")),
    (("-v", "--verbose"), dict(action="store_true", default=False, help="more log output")),
    (("--debug",), dict(action="store_true", default=False, help="stack traces for errors")),
    (("--ffmpeg-cc",), dict(action="store_true", default=False, help="accepted for compatibility, no effect")),
    (("--temp-resample",), dict(type=float, nargs="?", default=-1, const=0, help=_NA)),
    (("-i", "--interactive"), dict(action="store_true", default=False, help="one command line per line of standard input")),
    (("--dump-channels",), dict(nargs="+", choices=["temporal", "lpyr", "difference"], default=None, help=_NA)),
)


def parse_args(arg_list=None):
    p = argparse.ArgumentParser(prog="cvvdp", description="ColorVideoVDP on MI355X: quality of test images / videos against their references")
    for flags, kw in _OPTIONS:
        if flags[-1] == "--metric":
            kw = dict(kw, choices=[mm.replace("_", "-") for mm in vq_metric_dict.keys()])
        p.add_argument(*flags, **kw)
    return p.parse_args(arg_list)


def load_source(test_file, ref_file, display_photometry, config_paths, nframes=-1, fps=None, frame_range=None, full_screen_resize=None,
                resize_resolution=None):
    """The reference's video_source_file dispatch (run_cvvdp.py:296-318) for the formats available here; returns the source
    that does the work."""
    return video_source_file(test_file, ref_file, display_photometry=display_photometry, config_paths=config_paths, frames=nframes, fps=fps,
                             frame_range=frame_range, full_screen_resize=full_screen_resize, resize_resolution=resize_resolution).vs


def parse_frame_range(spec):
    """--frames first:step:last | first:last | first:   (Matlab notation, both ends included; run_cvvdp.py:143-157)."""
    if spec is None:
        return None
    ss = spec.split(":")
    sn = [0, 1, 10000] if len(ss) == 3 else [0, 10000]
    for kk in range(min(len(ss), len(sn))):
        if ss[kk].isnumeric():
            sn[kk] = int(ss[kk])
    return range(sn[0], sn[2] + 1, sn[1]) if len(ss) == 3 else range(sn[0], sn[1] + 1)


def run_on_args(args):
    """run_cvvdp.py:120-371."""
    logging.basicConfig(format="[%(levelname)s] %(message)s", level=logging.ERROR if args.quiet else (logging.DEBUG if args.verbose else logging.INFO), force=True)
    args.metric = [mm.replace("-", "_") for mm in args.metric]
    if args.display == "?":
        vvdp_display_photometry.list_displays(args.config_paths)
        return
    if args.test is None or args.ref is None:
        logging.error("Paths to both test and reference content needs to be specified.")
        return
    frame_range = parse_frame_range(args.frames)
    for opt, what in ((args.dump_channels, "--dump-channels"),):
        if opt is not None:
            raise vq_exception(f"{what} is not available in the MI355X build")
    if args.temp_resample >= 0:
        raise vq_exception("--temp-resample is not available in the MI355X build")
    if args.temp_padding == "valid":
        raise vq_exception("--temp-padding valid is not available in the MI355X build (replicate or symmetric)")
    device = args.device.lower()
    if not device.startswith("cuda"):
        raise vq_exception(f"--device {args.device}: the MI355X build runs on a cuda (HIP) device only")
    if not torch.cuda.is_available():
        raise vq_exception("no HIP device available; this build has no CPU path")
    device = torch.device(device)
    logging.info("Running on device: " + str(device))
    if args.heatmap == "none":
        args.heatmap = None
    if args.heatmap and args.heatmap not in ("raw", "threshold", "supra-threshold"):
        logging.error('The recognized heatmap types are: "none", "raw", "threshold" and "supra-threshold"')
        sys.exit()
    args.test, args.ref = expand_wildcards(args.test), expand_wildcards(args.ref)
    n_test, n_ref = len(args.test), len(args.ref)
    if n_test == 0:
        logging.error("No test images/videos found.")
        sys.exit()
    if n_ref == 0:
        logging.error("No reference images/videos found.")
        sys.exit()
    if n_test != n_ref and n_test != 1 and n_ref != 1:
        logging.error("Pass the same number of reference and test sources, or a single reference (to be used with all test sources), "
                      "or a single test (to be used with all reference sources).")
        sys.exit()
    display_photometry = vvdp_display_photometry.load(args.display, config_paths=args.config_paths)
    if args.pix_per_deg is None:
        display_geometry = vvdp_display_geometry.load(args.display, config_paths=args.config_paths)
    else:
        display_geometry = vvdp_display_geometry([1024, 1024], ppd=args.pix_per_deg)
    out_dir = "." if args.output_dir is None else args.output_dir
    os.makedirs(out_dir, exist_ok=True)

    metrics = []
    for mm in args.metric:
        if mm not in vq_metric_dict:
            raise RuntimeError(f"Unknown metric {mm}")
        fv = vq_metric_dict[mm](display_photometry=display_photometry, display_geometry=display_geometry, device=device, heatmap=args.heatmap,
                                temp_padding=args.temp_padding, config_paths=args.config_paths, gpu_mem=args.gpu_mem, quiet=args.quiet)
        fv.train(False)
        metrics.append(fv)
        info = fv.get_info_string()
        if info is not None:
            logging.info("When reporting metric results, please include the following information:")
            logging.info(info)

    res_fh = None
    if args.result is not None:
        res_fh = open(args.result, "w")
        res_fh.write("test, reference" + "".join(", " + mm.short_name() for mm in metrics) + "\n")
    try:
        for kk in range(max(n_test, n_ref)):
            test_file, ref_file = args.test[min(kk, n_test - 1)], args.ref[min(kk, n_ref - 1)]
            if res_fh is not None:
                res_fh.write(f"{test_file}, {ref_file}")
            logging.info(f"Predicting the quality of '{test_file}' compared to '{ref_file}'")
            for mm in metrics:
                vs = load_source(test_file, ref_file, display_photometry, args.config_paths, nframes=args.nframes, fps=args.fps, frame_range=frame_range,
                                 full_screen_resize=args.full_screen_resize, resize_resolution=display_geometry.resolution)
                base = os.path.splitext(os.path.basename(test_file))[0]
                mm.set_base_fname(os.path.join(out_dir, base))
                is_video = vs.get_video_size()[2] > 1
                sink = None
                if args.heatmap and is_video:          # streamed to disk block by block: bounded host memory at any clip length
                    if heatmap_writers.HeatmapVideoWriter.available():     # the reference's file (run_cvvdp.py:349-354)
                        dest = os.path.join(out_dir, base + "_heatmap.mp4")
                        logging.info(f"Writing heat map '{dest}' ...")
                        sink = heatmap_writers.HeatmapVideoWriter(dest, vs.get_frames_per_second(), verbose=args.verbose)
                    else:
                        pattern = os.path.join(out_dir, base + "_heatmap_%05d.png")
                        logging.info(f"Writing heat map frames '{pattern}' ...")
                        sink = heatmap_writers.HeatmapPngWriter(pattern)
                try:
                    Q_pred, stats = mm.predict_video_source(vs, heatmap_sink=sink) if sink is not None else mm.predict_video_source(vs)
                finally:
                    if sink is not None:
                        sink.close()
                q = Q_pred.item()
                print(f"{q:0.4f}" if args.quiet else f"{mm.short_name()}={q:0.4f} [{mm.quality_unit()}]")
                if res_fh is not None:
                    res_fh.write(f", {q}")
                if args.features and stats is not None:
                    dest = os.path.join(out_dir, base + "_fmap.json")
                    logging.info("Writing feature map '" + dest + "' ...")
                    mm.write_features_to_json(stats, dest)
                if args.heatmap and not is_video and stats is not None:
                    dest = os.path.join(out_dir, base + "_heatmap.png")
                    logging.info("Writing heat map '" + dest + "' ...")
                    from PIL import Image
                    Image.fromarray(heatmap_writers.heatmap_to_uint8(stats["heatmap"])[0]).save(dest)
                if args.distogram != -1:
                    dest = os.path.join(out_dir, base + "_distogram.png")
                    logging.info("Writing distogram '" + dest + "' ...")
                    mm.export_distogram(stats, dest, jod_max=args.distogram)
                del stats
            if res_fh is not None:
                res_fh.write("\n")
    finally:
        if res_fh is not None:
            res_fh.close()


def main(argv=None):
    args = parse_args(argv)
    try:
        if args.interactive:
            while True:
                line = sys.stdin.readline()
                if not line:
                    break
                args = parse_args(shlex.split(line))
                run_on_args(args)
        else:
            run_on_args(args)
    except vq_exception as ex:
        logging.error(str(ex))
        if args.debug:
            traceback.print_exc()
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
