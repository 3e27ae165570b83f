"""File sources, mirroring pycvvdp/video_source_file.py for the formats this build reads without ffmpeg.

  load_image_as_array        video_source_file.py:36-70   8 / 16-bit image file -> [H, W, C] array
  video_source_image_frames  video_source_file.py:549-652 an image pair, or a clip stored as numbered frames ("f_%04d.png" + fps)
  video_source_file          video_source_file.py:755-820 dispatch on the file extension (.yuv, images, .npy; compressed video
                                                          files need ffmpeg and are refused)

The sources hand RAW samples to the metric (`get_raw_block`): unpacking, the display model and the colour transform happen in the
HIP temporal kernel, not here.
"""
import os
import re

import numpy as np
import torch

from .display_model import vvdp_display_photometry
from .video_source import video_source, video_source_array
from .video_source_yuv import video_source_yuv_file
from .vq_metric import vq_exception

IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".ppm", ".pgm", ".gif")
VIDEO_EXT = (".mp4", ".mkv", ".mov", ".avi", ".webm", ".m4v", ".y4m")


def _png_is_16bit_colour(fname):
    with open(fname, "rb") as f:
        head = f.read(26)
    return head[:8] == b"\x89PNG\r\n\x1a\n" and head[12:16] == b"IHDR" and head[24] == 16 and head[25] in (2, 6)


def _read_png16(fname):
    """16-bit RGB(A) PNG -> uint16 [H, W, 3].  Pillow reduces these to 8 bit, which moves the metric by several 1e-3 JOD
    (the reference's own example image, example_media/wavy_facade.png, is such a file), so they are decoded here:
    zlib stream + the five PNG row filters (PNG specification, section 9)."""
    import struct
    import zlib
    data = open(fname, "rb").read()
    pos, idat = 8, b""
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if interlace:
                raise vq_exception(f"'{fname}': interlaced 16-bit PNGs are not supported")
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
    ch = 3 if ctype == 2 else 4
    bpp, stride = 2 * ch, w * 2 * ch
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:                                   # Up
            cur = (line + prev) & 255
        elif ft == 1:                                   # Sub: a running sum per byte lane
            cur = (np.cumsum(line.reshape(w, bpp), axis=0) & 255).reshape(-1)
        else:                                           # Average / Paeth depend on the pixel to the left: sequential
            cur = line.copy()
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = prev[i - bpp] if i >= bpp else 0
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (cur[i] + pred) & 255
        out[y] = cur
        prev = cur
    px = out.reshape(h, w, ch, 2).astype(np.uint16)
    return np.ascontiguousarray(((px[..., 0] << 8) | px[..., 1])[..., :3])


def _palette_is_grey(im):
    """A palette image whose every entry has R == G == B is a greyscale image; any other palette carries colour and is
    decoded to RGB, as the reference's imageio read returns it."""
    pal = np.asarray(im.getpalette() or [], dtype=np.uint8)
    pal = pal[:(pal.size // 3) * 3].reshape(-1, 3)
    return bool(pal.size == 0 or (np.all(pal[:, 0] == pal[:, 1]) and np.all(pal[:, 1] == pal[:, 2])))


def load_image_as_array(fname):
    """8- or 16-bit image file -> uint8 / uint16 array [H, W, C] (the reference reads them with imageio, video_source_file.py)."""
    if not os.path.isfile(fname):
        raise FileNotFoundError(f"File '{fname}' not found")                                     # video_source_file.py:37-40
    if _png_is_16bit_colour(fname):
        return _read_png16(fname)
    from PIL import Image
    with Image.open(fname) as im:
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            a = np.asarray(im).astype(np.uint16)[..., None]
        elif im.mode in ("L", "1") or (im.mode == "P" and _palette_is_grey(im)):
            a = np.asarray(im.convert("L"))[..., None]
        else:
            a = np.asarray(im.convert("RGB"))
            if a.dtype not in (np.uint8, np.uint16):
                a = a.astype(np.uint8)
    return np.ascontiguousarray(a)


class video_source_image_frames(video_source):
    """Test / reference content in image files (video_source_file.py:549-652): one image each (fps == 0), or a clip stored
    as numbered frames -- both names then carry a C-style frame-number field such as "%04d", fps > 0, and the frames of
    `frame_range` (default 0, 1, 2, ...) are used up to the first number for which either file is missing.

    Frames are decoded on demand, a block at a time (`get_raw_block`), and handed to the metric as 8 / 16-bit codes."""

    device_resident = False        # frames cross PCIe block by block: the core keeps the temporal history itself

    def __init__(self, test_fname, reference_fname, fps=0, frame_range=None, display_photometry="sdr_4k_30", config_paths=[],
                 full_screen_resize=None, resize_resolution=None, verbose=False):
        if isinstance(display_photometry, str):
            self.dm_photometry = vvdp_display_photometry.load(display_photometry, config_paths)
        else:
            self.dm_photometry = display_photometry
        self.fps = fps if fps else 0
        self.test_fname, test_has_no = self.convert_c2python_format_str(test_fname)
        self.reference_fname, ref_has_no = self.convert_c2python_format_str(reference_fname)
        if full_screen_resize:
            raise vq_exception("full-screen-resize not implemented for images.")                 # :562-564
        if test_has_no != ref_has_no:
            raise vq_exception("Both test and reference names must contain `%0Nd` string to be replaced with a frame number")
        if (self.fps > 0) != test_has_no:
            raise vq_exception("A valid frames-per-second number (--fps) must be provided when input are video frames, "
                               "or fps should be zero for images.")
        if self.fps == 0:
            self.frame_range = None
            self.N = 1
        else:
            if not frame_range:
                frame_range = range(0, 10000)
            count = 0
            for nn in frame_range:                                                              # :581-590
                if os.path.isfile(self.test_fname.format(nn)) and os.path.isfile(self.reference_fname.format(nn)):
                    count += 1
                else:
                    break
            if count == 0:
                raise vq_exception(f"No frames found for {test_fname} and {reference_fname}")
            self.N = count
            self.frame_range = frame_range[0:count]
        self._size = None
        self._first = None

    _format_re = re.compile(r"%(\d)*d")

    @classmethod
    def convert_c2python_format_str(cls, name):
        """"frame_%04d.png" -> ("frame_{:04d}.png", True)   (video_source_file.py:601-612)."""
        m = cls._format_re.search(name)
        if not m:
            return name, False
        beg, end = m.span()
        return name[:beg] + "{:" + name[beg + 1:end] + "}" + name[end:], True

    def _names(self, frame):
        if self.frame_range is None:
            return self.test_fname, self.reference_fname
        nn = self.frame_range[frame]
        return self.test_fname.format(nn), self.reference_fname.format(nn)

    def _load_pair(self, frame):
        if frame == 0 and self._first is not None:
            pair, self._first = self._first, None            # the pair read for get_video_size
            return pair
        ft, fr = self._names(frame)
        t, r = load_image_as_array(ft), load_image_as_array(fr)
        if t.shape != r.shape or t.dtype != r.dtype:
            raise vq_exception(f"Test and reference images differ in size or bit depth: '{ft}' {t.shape} {t.dtype} vs '{fr}' {r.shape} {r.dtype}")
        return t, r

    def get_frames_per_second(self):
        return self.fps

    def get_video_size(self):
        if self._size is None:
            self._first = self._load_pair(0)
            self._size = (self._first[0].shape[0], self._first[0].shape[1], self.N)
            self._fmt = (self._first[0].dtype, self._first[0].shape[2])       # every frame of the clip must have these
        return self._size

    def get_raw_block(self, first, last, device):
        """Frames [first, last) of test and reference as [1, C, n, H, W] code tensors on `device` + the C ABI's dtype code."""
        h, w, _ = self.get_video_size()
        out, code = [None, None], None
        for k, f in enumerate(range(first, last)):
            pair = self._load_pair(f)
            if pair[0].shape[:2] != (h, w):
                raise vq_exception(f"Frame {f} ('{self._names(f)[0]}') is {pair[0].shape[1]}x{pair[0].shape[0]}, the first frame {w}x{h}")
            if (pair[0].dtype, pair[0].shape[2]) != self._fmt:     # (a 16-bit frame assigned into an 8-bit block would wrap silently)
                raise vq_exception(f"Frame {f} ('{self._names(f)[0]}') is {pair[0].dtype} with {pair[0].shape[2]} channel(s), "
                                   f"the first frame {self._fmt[0]} with {self._fmt[1]}")
            for side in range(2):
                a = pair[side]
                a = a.view(np.int16) if a.dtype == np.uint16 else a        # torch has no uint16 (video_source.py:259-263)
                t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
                if out[side] is None:
                    out[side] = torch.empty((1, t.shape[0], last - first, h, w), dtype=t.dtype, pin_memory=torch.cuda.is_available())
                    code = 1 if t.dtype == torch.int16 else 0
                out[side][0, :, k] = t
        return out[0].to(device, non_blocking=True), out[1].to(device, non_blocking=True), code

    def get_test_frame(self, frame, device, colorspace="Y"):
        raise NotImplementedError("image frames are converted on the GPU by colorvideovdp_amd.cvvdp.predict_video_source")

    get_reference_frame = get_test_frame


class video_source_file(video_source):
    """Source for a pair of files of any supported kind (video_source_file.py:755-820; fps None = taken from the file,
    0 = image, otherwise a clip).  `.vs` is the source that does the work."""

    def __init__(self, test_fname, reference_fname, display_photometry="sdr_4k_30", config_paths=[], frames=-1, frame_range=None, fps=None,
                 full_screen_resize=None, resize_resolution=None, preload=False, ffmpeg_cc=False, verbose=False):
        ext = os.path.splitext(test_fname)[1].lower()
        ext_r = os.path.splitext(reference_fname)[1].lower()
        if (ext in IMAGE_EXT) != (ext_r in IMAGE_EXT):
            raise vq_exception("Test is an image, but reference is a video" if ext in IMAGE_EXT else "Test is a video, but reference is an image")
        if ext in IMAGE_EXT:
            self.vs = video_source_image_frames(test_fname, reference_fname, fps=fps, frame_range=frame_range, display_photometry=display_photometry,
                                                config_paths=config_paths, full_screen_resize=full_screen_resize, resize_resolution=resize_resolution,
                                                verbose=verbose)
            if frames is not None and frames > 0 and self.vs.fps > 0 and frames < self.vs.N:
                self.vs.N, self.vs.frame_range = frames, self.vs.frame_range[0:frames]
        else:
            for f in (test_fname, reference_fname):
                if not os.path.isfile(f):
                    raise vq_exception(f"File not found: '{f}'")
            if ext != ext_r:
                raise vq_exception(f"Test and reference must be files of the same kind ('{test_fname}' vs '{reference_fname}')")
            if ext == ".yuv":
                self.vs = video_source_yuv_file(test_fname, reference_fname, display_photometry=display_photometry, frames=frames,
                                                full_screen_resize=full_screen_resize, resize_resolution=resize_resolution, config_paths=config_paths)
                if fps is not None:
                    self.vs.test_vidr.avg_fps = self.vs.reference_vidr.avg_fps = fps
            elif ext == ".npy":
                if full_screen_resize:
                    raise vq_exception("full-screen-resize is not available for .npy inputs")
                t, r = np.load(test_fname, mmap_mode="r"), np.load(reference_fname, mmap_mode="r")
                if t.ndim == 3:
                    self.vs = video_source_array(np.ascontiguousarray(t), np.ascontiguousarray(r), 0, dim_order="HWC", display_photometry=display_photometry)
                elif t.ndim == 4:
                    if not fps:
                        raise vq_exception("--fps is required for .npy videos")
                    if frames is not None and frames > 0:
                        t, r = t[:frames], r[:frames]
                    self.vs = video_source_array(np.ascontiguousarray(t), np.ascontiguousarray(r), fps, dim_order="FHWC", display_photometry=display_photometry)
                else:
                    raise vq_exception(".npy inputs must be [H, W, C] images or [F, H, W, C] videos")
            elif ext in VIDEO_EXT:
                raise vq_exception(f"'{test_fname}': compressed video files need ffmpeg, which this build does not use. Decode both clips to planar "
                                   f".yuv first, e.g. `ffmpeg -i clip.mp4 -pix_fmt yuv420p clip_1920x1080_30fps_420p8.yuv`")
            else:
                raise vq_exception(f"Unsupported file type '{ext}'")

    def get_video_size(self):
        return self.vs.get_video_size()

    def get_frames_per_second(self):
        return self.vs.get_frames_per_second()

    def get_batch_size(self):
        return self.vs.get_batch_size()

    def get_test_frame(self, frame, device, colorspace="Y"):
        return self.vs.get_test_frame(frame, device, colorspace)

    def get_reference_frame(self, frame, device, colorspace="Y"):
        return self.vs.get_reference_frame(frame, device, colorspace)
