"""Planar Y'CbCr (.yuv) clips, mirroring pycvvdp/video_source_yuv.py (decode_video_props :8-62, YUVReader :67-230,
video_source_yuv_file :262-372).

The reference unpacks every frame with torch ops (fixed point -> float, bilinear chroma up-sampling, Y'CbCr -> R'G'B',
display model, colour transform) before the metric sees it.  Here the raw planes of a block of frames are uploaded as
they are in the file (1.5-6 bytes per pixel instead of 12) and the HIP temporal kernel does all of that while it reads
them (cvvdp_process_block_yuv, include/cvvdp_hip.h), so the class only parses the header, maps the file and hands out
raw frames.
"""
import logging
import os
import re

import numpy as np
import torch

from . import _capi
from .display_model import vvdp_display_photo_eotf, vvdp_display_photometry
from .video_source import video_source


_RES = re.compile(r"(\d+)x(\d+)p?(\d+)?")
_FIELD_ALIASES = {   # file-name field -> (property, value), video_source_yuv.py:37-60
    **{k: ("chroma_ss", k) for k in ("444", "420", "422")},
    **{k: ("bit_depth", 10) for k in ("10", "10b", "10bit")},
    **{k: ("bit_depth", 8) for k in ("8", "8b", "8bit")},
    **{k: ("color_space", "709") for k in ("709", "bt709", "sdr")},
    **{k: ("color_space", "2020") for k in ("2020", "ct2020", "pq2020", "hdr")},
}


def decode_video_props(fname):
    """Header fields encoded in the file name, e.g. clip_1920x1080_10b_420_2020_60fps.yuv (video_source_yuv.py:8-62).
    Unknown fields are ignored; missing ones keep the reference's defaults (1080p, 24 fps, 8 bit, 4:2:0, BT.709)."""
    props = {"width": 1920, "height": 1080, "fps": 24, "bit_depth": 8, "color_space": "709", "chroma_ss": "420"}
    stem = os.path.splitext(os.path.basename(fname))[0]
    for field in stem.split("_"):
        if _RES.match(field):
            nums = [int(n) for n in re.findall(r"\d+", field)]
            if not 2 <= len(nums) <= 3:
                raise ValueError("Cannot decode the resolution")
            props["width"], props["height"] = nums[0], nums[1]
            if len(nums) == 3:
                props["fps"] = nums[2]
        elif field.endswith("fps"):
            props["fps"] = float(field[:-3])
        elif field in _FIELD_ALIASES:
            key, value = _FIELD_ALIASES[field]
            props[key] = value
    return props


def create_yuv_fname(basename, vprops):
    """video_source_yuv.py:65-75."""
    fps = vprops["fps"]
    fps = round(fps, 3) if round(fps) != fps else int(fps)
    return f'{basename}_{vprops["width"]}x{vprops["height"]}_{vprops["bit_depth"]}b_{vprops["chroma_ss"]}_{vprops["color_space"]}_{fps}fps.yuv'


class YUVReader:
    """Geometry and random access to the frames of one .yuv file (video_source_yuv.py:79-145)."""

    def __init__(self, file_name):
        self.file_name = file_name
        if not os.path.isfile(file_name):
            raise FileNotFoundError("File {} not found".format(file_name))
        vprops = decode_video_props(file_name)
        self.width = vprops["width"]
        self.height = vprops["height"]
        self.avg_fps = vprops["fps"]
        self.color_space = vprops["color_space"]
        self.chroma_ss = vprops["chroma_ss"]
        self.bit_depth = vprops["bit_depth"]
        self.y_pixels = int(self.width * self.height)
        self.y_shape = (self.height, self.width)
        if self.chroma_ss == "444":
            self.uv_shape = self.y_shape
        elif self.chroma_ss == "420":
            self.uv_shape = (int(self.height / 2), int(self.width / 2))
        elif self.chroma_ss == "422":
            self.uv_shape = (int(self.height), int(self.width / 2))
        else:
            raise RuntimeError(f"Unsupported chroma subsampling {self.chroma_ss}")
        self.uv_pixels = self.uv_shape[0] * self.uv_shape[1]
        self.frame_pixels = self.y_pixels + 2 * self.uv_pixels          # samples per frame
        self.dtype = np.uint16 if self.bit_depth > 8 else np.uint8
        self.frame_bytes = self.frame_pixels * np.dtype(self.dtype).itemsize
        self.frames = int(os.stat(file_name).st_size / self.frame_bytes)
        self.mm = None

    def get_frame_count(self):
        return int(self.frames)

    def _map(self):
        if self.mm is None:
            self.mm = np.memmap(self.file_name, self.dtype, mode="r")
        return self.mm

    def get_frame_yuv(self, frame_index):
        """(Y, u, v) planes of one frame as numpy views (video_source_yuv.py:131-145)."""
        if frame_index < 0 or frame_index >= self.frames:
            raise RuntimeError("The frame index is outside the range of available frames")
        mm = self._map()
        off = int(frame_index * self.frame_pixels)
        Y = mm[off:off + self.y_pixels]
        u = mm[off + self.y_pixels:off + self.y_pixels + self.uv_pixels]
        v = mm[off + self.y_pixels + self.uv_pixels:off + self.y_pixels + 2 * self.uv_pixels]
        return (np.reshape(Y, self.y_shape, "C"), np.reshape(u, self.uv_shape, "C"), np.reshape(v, self.uv_shape, "C"))

    def raw_frames(self, first, last):
        """Samples of frames [first, last) exactly as stored: one flat array of (last-first)*frame_pixels codes."""
        if first < 0 or last > self.frames or last <= first:
            raise RuntimeError("The frame index is outside the range of available frames")
        return self._map()[first * self.frame_pixels:last * self.frame_pixels]

    def get_frame_rgb_tensor(self, frame_index, device):
        raise NotImplementedError("frames are unpacked by the HIP core while the metric reads them "
                                  "(cvvdp.predict_video_source); there is no host-side Y'CbCr -> RGB path")

    def __enter__(self):
        return self

    def __exit__(self, type, value, tb):
        self.mm = None


class video_source_yuv_file(video_source):
    """Test / reference pair of .yuv files (video_source_yuv.py:262-372)."""

    def __init__(self, test_fname, reference_fname, display_photometry="standard_4k", frames=-1, full_screen_resize=None,
                 resize_resolution=None, retain_aspect_ratio=False, verbose=False, config_paths=[]):
        if full_screen_resize is not None and full_screen_resize not in _capi.RESIZE_MODES:
            raise RuntimeError(f"full_screen_resize must be one of {sorted(_capi.RESIZE_MODES)}")
        self.reference_vidr = YUVReader(reference_fname)
        self.test_vidr = YUVReader(test_fname)
        t, r = self.test_vidr, self.reference_vidr
        # without a resize both files feed one fused kernel launch and must match; with it every side is unpacked and resized
        # on its own, so e.g. a 1080p encode can be measured against its 4K source like in the reference
        if full_screen_resize is None and (t.width, t.height, t.chroma_ss, t.bit_depth, t.color_space) != (r.width, r.height, r.chroma_ss, r.bit_depth, r.color_space):
            raise RuntimeError("Test and reference .yuv files must have the same resolution, chroma subsampling, bit depth and colour space")
        self.full_screen_resize = full_screen_resize
        if full_screen_resize is not None:
            if resize_resolution is None:
                raise RuntimeError("full_screen_resize needs resize_resolution = (width, height)")
            if retain_aspect_ratio:                                   # video_source_yuv.py:273-282
                h, w = t.height, t.width
                if h / resize_resolution[1] * resize_resolution[0] <= w:
                    resize_resolution = (resize_resolution[0], int(resize_resolution[0] / w * h))
                else:
                    resize_resolution = (int(resize_resolution[1] / h * w), resize_resolution[1])
        self.resize_resolution = None if resize_resolution is None else (int(resize_resolution[0]), int(resize_resolution[1]))
        self.total_frames = self.test_vidr.frames
        self.frames = self.total_frames if frames == -1 else min(self.total_frames, frames)
        self.offset = 0
        # video_source_dm.__init__ (video_source.py:206-215)
        if isinstance(display_photometry, str):
            self.dm_photometry = vvdp_display_photometry.load(display_photometry, config_paths)
        elif isinstance(display_photometry, vvdp_display_photo_eotf):
            self.dm_photometry = display_photometry
        else:
            raise RuntimeError("display_photometry must be a display name or a vvdp_display_photo_eotf")
        for vr, what in ((self.test_vidr, "Test"), (self.reference_vidr, "Reference")):
            logging.debug(f"{what} video '{vr.file_name}': [{vr.width}x{vr.height}], colorspace: {vr.color_space}, "
                          f"EOTF: {self.dm_photometry.EOTF}, fps: {vr.avg_fps}, frames: {self.frames}")

    def get_video_size(self):
        if self.full_screen_resize is not None:                       # video_source_yuv.py:308-312
            return [self.resize_resolution[1], self.resize_resolution[0], self.frames]
        return [self.test_vidr.height, self.test_vidr.width, self.frames]

    def needs_resize(self):
        """Does any side have to be interpolated (or unpacked on its own because the two files differ)?"""
        if self.full_screen_resize is None:
            return False
        t, r = self.test_vidr, self.reference_vidr
        same = (t.width, t.height, t.chroma_ss, t.bit_depth, t.color_space) == (r.width, r.height, r.chroma_ss, r.bit_depth, r.color_space)
        return not same or (t.width, t.height) != self.resize_resolution

    def get_frames_per_second(self):
        return self.test_vidr.avg_fps

    def get_batch_size(self):
        return 1

    def set_offset(self, offset):
        self.offset = offset

    def get_total_frames(self):
        return self.total_frames

    def set_num_frames(self, num_frames):
        if self.offset + num_frames > self.total_frames:
            logging.error(f"Cannot set num_frames={num_frames} because offset={self.offset} and total_frames={self.total_frames}. "
                          f"Clipping num_frames to {self.total_frames - self.offset}")
            num_frames = self.total_frames - self.offset
        self.frames = num_frames

    def get_test_frame(self, frame, device, colorspace="Y"):
        raise NotImplementedError("frames of a .yuv source are unpacked on the GPU by colorvideovdp_amd.cvvdp.predict_video_source")

    get_reference_frame = get_test_frame

    # raw access used by the metric (cvvdp_process_block_yuv / cvvdp_unpack_yuv_resized)
    def _upload(self, vr, first, last, device):
        a = vr.raw_frames(self.offset + first, self.offset + last)
        tdt = torch.int16 if a.dtype == np.uint16 else torch.uint8   # torch has no uint16: keep the bit pattern
        try:
            host = torch.empty(a.shape, dtype=tdt, pin_memory=True)   # page-locked staging: the H2D copy is asynchronous
        except RuntimeError:
            host = torch.empty(a.shape, dtype=tdt)
        host.numpy().view(a.dtype)[...] = a                          # one read of the mapped file
        return host.to(device, non_blocking=True)

    def get_raw_yuv_block(self, first, last, device):
        """Frames [first, last) of both files as flat device tensors + the cvvdp_yuv_format describing them."""
        out = [self._upload(vr, first, last, device) for vr in (self.test_vidr, self.reference_vidr)]
        fmt = _capi.YuvFormat()
        fmt.chroma, fmt.bit_depth, fmt.matrix = int(self.test_vidr.chroma_ss), int(self.test_vidr.bit_depth), int(self.test_vidr.color_space)
        fmt.frame_stride_test, fmt.frame_stride_ref = self.test_vidr.frame_pixels, self.reference_vidr.frame_pixels
        return out[0], out[1], fmt

    def get_raw_yuv_side(self, side, first, last, device):
        """Frames [first, last) of the test (side 0) or reference (1) file: (codes, format, width, height) -- for the resize path,
        where the two files may differ in size and format."""
        vr = self.reference_vidr if side else self.test_vidr
        fmt = _capi.YuvFormat()
        fmt.chroma, fmt.bit_depth, fmt.matrix = int(vr.chroma_ss), int(vr.bit_depth), int(vr.color_space)
        fmt.frame_stride_test = fmt.frame_stride_ref = vr.frame_pixels
        return self._upload(vr, first, last, device), fmt, vr.width, vr.height
