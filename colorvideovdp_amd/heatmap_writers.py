"""Streaming heat-map sinks for `cvvdp.predict_video_source(vs, heatmap_sink=...)`.

The reference returns the whole heat map as one fp16 CPU tensor (cvvdp_metric.py:344,396-401) and its command line turns
it into an .mp4 through an ffmpeg pipe or into a .png (run_cvvdp.py:44-78,349-363).  At 8K x 256 frames that tensor is
51 GB, so here the metric can hand the frames over block by block instead; these sinks put them on disk with bounded
memory.  ffmpeg is not a dependency: videos become a numbered PNG sequence (`ffmpeg -i base_%05d.png out.mp4` turns it
into the reference's format) or one .npy file with the reference's array layout; where an `ffmpeg` executable is on the
PATH, HeatmapVideoWriter pipes the frames into it and produces the reference's .mp4 directly.

A sink is any callable `sink(first_frame, frames)`; `frames` is a float16 CPU tensor [1, 1|3, n, H, W] with values in
[0, 1] (colour-mapped modes) that is only valid during the call.  Attributes a sink may set: `wants_uint8` (frames arrive as
uint8 [n, H, W, C], converted on the GPU the way the reference's writers convert them) and `wants_device` (frames arrive as
DEVICE tensors on the current stream and never cross PCIe here: statistics, a device-side consumer, one bulk copy later).
"""
import os
import shutil
import subprocess

import numpy as np
import torch


def heatmap_to_uint8(frames):
    """[1, C, n, H, W] fp16 in [0,1] -> [n, H, W, 3] uint8, the conversion of run_cvvdp.py:np2vid / np2img (:62-76).
    Frames that arrive as uint8 [n, H, W, C] (a sink with wants_uint8: the GPU has done exactly this) pass through."""
    if frames.dtype == torch.uint8:
        a = frames.numpy()
        return np.concatenate([a] * 3, -1) if a.shape[-1] == 1 else a
    a = frames[0].permute(1, 2, 3, 0).numpy()                    # [n, H, W, C], float16 like the reference's array
    if a.shape[-1] == 1:
        a = np.concatenate([a] * 3, -1)
    assert a.dtype == np.float16
    return (np.clip(a, 0.0, 1.0) * 255.0).astype(np.uint8)       # float16 product (rounded to half), then truncated: np2vid / np2img


class HeatmapPngWriter:
    """One 8-bit PNG per frame: `pattern % frame_index`, e.g. "out/clip_heatmap_%05d.png" (needs Pillow)."""

    wants_uint8 = True          # frames arrive as uint8 [n, H, W, C], converted on the GPU

    def __init__(self, pattern):
        if "%" not in pattern:
            raise ValueError("the file name pattern needs a frame-number field, e.g. 'heatmap_%05d.png'")
        self.pattern = pattern
        self.frames_written = 0
        d = os.path.dirname(pattern)
        if d:
            os.makedirs(d, exist_ok=True)

    def __call__(self, first_frame, frames):
        from PIL import Image
        rgb = heatmap_to_uint8(frames)
        for i in range(rgb.shape[0]):
            Image.fromarray(rgb[i]).save(self.pattern % (first_frame + i))
            self.frames_written += 1

    def close(self):
        pass


class HeatmapVideoWriter:
    """The reference's heat-map video (run_cvvdp.py:44-66 np2vid): raw rgb24 frames piped into `ffmpeg`, mpeg4 codec at
    qscale 3, the clip's frame rate.  The process is started with the first block (the frame size is known then) and
    closed by close(); `available()` tells whether an ffmpeg executable exists on this machine."""

    wants_uint8 = True          # frames arrive as uint8 [n, H, W, C], converted on the GPU

    def __init__(self, path, fps, verbose=False, ffmpeg=None):
        self.path, self.fps, self.verbose = path, fps, verbose
        self.exe = ffmpeg or shutil.which("ffmpeg")
        if self.exe is None:
            raise FileNotFoundError("no ffmpeg executable on the PATH (use HeatmapPngWriter / HeatmapNpyWriter instead)")
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        self.proc = None
        self.frames_written = 0

    @staticmethod
    def available():
        return shutil.which("ffmpeg") is not None

    def __call__(self, first_frame, frames):
        rgb = heatmap_to_uint8(frames)
        if self.proc is None:
            h, w = rgb.shape[1:3]
            cmd = [self.exe, "-hide_banner", "-loglevel", "info" if self.verbose else "quiet", "-f", "rawvideo", "-pix_fmt", "rgb24",
                   "-s", f"{w}x{h}", "-r", f"{self.fps:g}", "-i", "pipe:", "-f", "mp4", "-c:v", "mpeg4", "-qscale:v", "3", "-y", self.path]
            self.proc = subprocess.Popen(cmd, stdin=subprocess.PIPE)
        self.proc.stdin.write(rgb.tobytes())
        self.frames_written += rgb.shape[0]

    def close(self):
        if self.proc is not None:
            self.proc.stdin.close()
            rc = self.proc.wait()
            self.proc = None
            if rc != 0:
                raise RuntimeError(f"ffmpeg exited with status {rc} while writing '{self.path}'")


class HeatmapNpyWriter:
    """The reference's `stats["heatmap"]` array ([1, C, F, H, W] float16) as a .npy file, written through a memory map."""

    def __init__(self, path, n_frames, height, width, channels=3):
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        self.path = path
        self.mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.float16, shape=(1, channels, n_frames, height, width))
        self.frames_written = 0

    def __call__(self, first_frame, frames):
        n = frames.shape[2]
        self.mm[:, :, first_frame:first_frame + n] = frames.numpy()
        self.frames_written += n

    def close(self):
        self.mm.flush()
        del self.mm


class HeatmapFrameMeans:
    """Keeps only a per-frame mean of every colour plane, taken over every `step`-th pixel in both directions (a cheap sink
    for benchmarks and tests: the host only touches 1/step^2 of the 6 bytes per pixel that crossed PCIe)."""

    def __init__(self, step=32, uint8=False, device=False):
        self.step = step
        self.wants_device = device        # take the frames as device tensors (nothing crosses PCIe; the means are reduced on the GPU and fetched in close())
        self.wants_uint8 = uint8          # take the frames as a file writer would (uint8 [n, H, W, C]); means are then of the 8-bit codes / 255
        self.means = {}
        self.frames_seen = 0

    def __call__(self, first_frame, frames):
        if frames.device.type != "cpu":
            if frames.dtype == torch.uint8:
                m = frames[:, ::self.step, ::self.step].float().mean(dim=(1, 2)).T / 255.0               # [C, n], stays on the device
            else:
                m = frames[0, :, :, ::self.step, ::self.step].float().mean(dim=(2, 3))
            self._pending = getattr(self, "_pending", [])
            self._pending.append((first_frame, m))
            self.frames_seen += m.shape[1]
            return
        if frames.dtype == torch.uint8:
            m = (frames[:, ::self.step, ::self.step].float().mean(dim=(1, 2)) / 255.0).numpy().T      # [C, n]
        else:
            m = frames[0, :, :, ::self.step, ::self.step].float().mean(dim=(2, 3)).numpy()           # [C, n]
        for i in range(m.shape[1]):
            self.means[first_frame + i] = m[:, i].copy()
        self.frames_seen += m.shape[1]

    def close(self):
        for first_frame, m in getattr(self, "_pending", []):
            m = m.cpu().numpy()
            for i in range(m.shape[1]):
                self.means[first_frame + i] = m[:, i].copy()
        self._pending = []

