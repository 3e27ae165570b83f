"""CPU restatement of the reference's planar Y'CbCr reader -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/pycvvdp/video_source_yuv.py:
  decode_video_props   :8-62     (file-name encoded header)
  YUVReader.__init__   :79-124   (plane geometry, frame count)
  get_frame_rgb_tensor :147-170  (Y'CbCr -> display-encoded R'G'B', BT.709 / BT.2020 matrices, clip to [0,1])
  _fixed2float_upscale :197-223  (limited-range fixed point -> float, bilinear chroma up-sampling)
The bilinear up-sampling is torch.nn.functional.interpolate(mode='bilinear') (align_corners=False), restated here
explicitly (half-pixel centres, source index clamped at 0, neighbour clamped at the last sample) so that the HIP kernel
and this file share one written-down definition; tests/test_yuv.py checks the restatement against torch itself and
against RGB frames produced by the real reference (tests/golden/yuv_*.npz, made by oracle/make_goldens_yuv.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import os
import re

import numpy as np

YCBCR2RGB = {   # video_source_yuv.py:151-160
    "2020": np.array([[1, 0, 1.47460], [1, -0.16455, -0.57135], [1, 1.88140, 0]], dtype=np.float32),
    "709": np.array([[1, 0, 1.402], [1, -0.344136, -0.714136], [1, 1.772, 0]], dtype=np.float32),
}


def decode_video_props(fname):
    """video_source_yuv.py:8-62."""
    v = dict(width=1920, height=1080, fps=24, bit_depth=8, color_space="709", chroma_ss="420")
    for field in os.path.splitext(os.path.basename(fname))[0].split("_"):
        if re.match(r"(\d+)x(\d+)p?(\d+)?", field):
            nums = re.findall(r"\d+", field)
            if len(nums) < 2 or len(nums) > 3:
                raise ValueError("Cannot decode the resolution")
            v["width"], v["height"] = int(nums[0]), int(nums[1])
            if len(nums) == 3:
                v["fps"] = int(nums[2])
        elif field.endswith("fps"):
            v["fps"] = float(field[:-3])
        elif field in ("444", "420", "422"):
            v["chroma_ss"] = field
        elif field in ("10", "10b", "10bit"):
            v["bit_depth"] = 10
        elif field in ("8", "8b", "8bit"):
            v["bit_depth"] = 8
        elif field in ("2020", "709"):
            v["color_space"] = field
        elif field in ("bt709", "sdr"):
            v["color_space"] = "709"
        elif field in ("ct2020", "pq2020", "hdr"):
            v["color_space"] = "2020"
    return v


def plane_shapes(height, width, chroma_ss):
    """video_source_yuv.py:98-112."""
    if chroma_ss == "444":
        return (height, width), (height, width)
    if chroma_ss == "420":
        return (height, width), (height // 2, width // 2)
    if chroma_ss == "422":
        return (height, width), (height, width // 2)
    raise RuntimeError(f"Unsupported chroma subsampling {chroma_ss}")


def split_frame(samples, frame, height, width, chroma_ss):
    """Planes of one frame of a flat sample array (video_source_yuv.py:131-145)."""
    ys, cs = plane_shapes(height, width, chroma_ss)
    ny, nc = ys[0] * ys[1], cs[0] * cs[1]
    off = frame * (ny + 2 * nc)
    return (samples[off:off + ny].reshape(ys), samples[off + ny:off + ny + nc].reshape(cs),
            samples[off + ny + nc:off + ny + 2 * nc].reshape(cs))


def upsample_axis(n_out, n_in, factor):
    """Source taps of bilinear interpolation with half-pixel centres along one axis: i0, i1, weight of i1."""
    src = np.maximum((np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(1.0 / factor) - np.float32(0.5), np.float32(0))
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    return i0, i1, (src - i0.astype(np.float32)).astype(np.float32)


def frame_to_rgb(Y, u, v, bit_depth, chroma_ss, color_space):
    """One frame -> display-encoded R'G'B' float32 [H,W,3] in [0,1] (video_source_yuv.py:147-170, 197-223)."""
    f32 = np.float32
    H, W = Y.shape
    scale = f32(2 ** (bit_depth - 8))
    yf = np.clip(f32(1.0) / (scale * f32(219)) * Y.astype(f32) - f32(16.0 / 219.0), f32(0), f32(1))
    wc, oc = f32(1.0) / (scale * f32(224)), f32(128.0 / 224.0)
    planes = []
    for c in (u, v):
        cf = np.clip(wc * c.astype(f32) - oc, f32(-0.5), f32(0.5))
        fy = 2 if chroma_ss == "420" else 1
        fx = 1 if chroma_ss == "444" else 2
        y0, y1, ly = upsample_axis(H, cf.shape[0], fy)
        x0, x1, lx = upsample_axis(W, cf.shape[1], fx)
        top = cf[y0][:, x0] * (f32(1) - lx)[None, :] + cf[y0][:, x1] * lx[None, :]
        bot = cf[y1][:, x0] * (f32(1) - lx)[None, :] + cf[y1][:, x1] * lx[None, :]
        planes.append((top * (f32(1) - ly)[:, None] + bot * ly[:, None]).astype(f32))
    yuv = np.stack([yf, planes[0], planes[1]], axis=-1).astype(f32)
    return np.clip(yuv @ YCBCR2RGB[color_space].T, f32(0), f32(1)).astype(f32)


def clip_to_rgb(samples, props, n_frames):
    """All frames of a flat sample array -> [1,3,F,H,W] float32 (the layout Oracle.predict takes)."""
    H, W = props["height"], props["width"]
    out = np.empty((1, 3, n_frames, H, W), dtype=np.float32)
    for f in range(n_frames):
        Y, u, v = split_frame(samples, f, H, W, props["chroma_ss"])
        out[0, :, f] = frame_to_rgb(Y, u, v, props["bit_depth"], props["chroma_ss"], props["color_space"]).transpose(2, 0, 1)
    return out


# ---------------------------------------------------------------------------------------------------
# full_screen_resize (video_source_yuv.py:333-336): torch.nn.functional.interpolate(size=..., mode=...) with its defaults
# (align_corners False, no antialiasing), restated from ATen/native/UpSample.h in float32, then clip to [0,1].

def _src_index(n_out, n_in, cubic):
    scale = np.float32(n_in) / np.float32(n_out)                         # area_pixel_compute_scale
    src = scale * (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)
    return src if cubic else np.maximum(src, np.float32(0))              # area_pixel_compute_source_index


def _cubic_weights(t):
    A = np.float32(-0.75)
    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return [c2(t + 1), c1(t), c1(1 - t), c2(2 - t)]


def _resize_axis(x, n_out, mode, axis):
    x = np.moveaxis(x, axis, -1).astype(np.float32)
    n_in = x.shape[-1]
    if mode == "nearest":                                                # nearest_neighbor_compute_source_index
        idx = np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * (np.float32(n_in) / np.float32(n_out))).astype(np.int64), n_in - 1)
        y = x[..., idx]
    elif mode == "bilinear":
        src = _src_index(n_out, n_in, False)
        i0 = src.astype(np.int64)
        i1 = i0 + (i0 < n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        y = (np.float32(1) - l1) * x[..., i0] + l1 * x[..., i1]
    elif mode == "bicubic":
        src = _src_index(n_out, n_in, True)
        fl = np.floor(src)
        w = _cubic_weights((src - fl).astype(np.float32))
        i = fl.astype(np.int64)
        y = sum(x[..., np.clip(i - 1 + k, 0, n_in - 1)] * w[k].astype(np.float32) for k in range(4))
    elif mode == "area":                                                 # adaptive average pooling: start / end index
        o = np.arange(n_out, dtype=np.int64)
        a, b = (o * n_in) // n_out, -((-(o + 1) * n_in) // n_out)
        y = np.stack([x[..., a[k]:b[k]].sum(-1, dtype=np.float32) for k in range(n_out)], -1)
        return np.moveaxis(y.astype(np.float32), -1, axis), (b - a)
    else:
        raise ValueError(mode)
    return np.moveaxis(y.astype(np.float32), -1, axis)


def resize_planes(x, height, width, mode):
    """[..., Hs, Ws] float32 -> [..., height, width], clipped to [0,1] (both separable passes in float32: rows of taps along x first,
    then along y, like ATen's bilinear / bicubic kernels; area divides the window sum by its pixel count)."""
    if mode == "area":
        y, nx = _resize_axis(x, width, mode, -1)
        y, ny = _resize_axis(y, height, mode, -2)
        y = y / (ny[:, None] * nx[None, :]).astype(np.float32)
    else:
        y = _resize_axis(_resize_axis(x, width, mode, -1), height, mode, -2)
    return np.clip(y, 0.0, 1.0).astype(np.float32)


def clip_to_rgb_resized(samples, props, n_frames, height, width, mode):
    """clip_to_rgb followed by the reference's per-frame resize; a clip that already has the target size is left alone (:333)."""
    rgb = clip_to_rgb(samples, props, n_frames)
    if (props["height"], props["width"]) == (height, width):
        return rgb
    return resize_planes(rgb, height, width, mode)
