#!/usr/bin/env python3
"""tests/golden/deep_8k_pq_80f.npz: the REAL reference's scores on the first 80 frames of the 7680x4320 PQ bench clip (configs[4]'s
clip, uint8 codes in the PQ range, no heat map: the reference keeps the whole heat-map tensor in host memory, 51 GB at 256 frames).

VERDICT r4 weak #3: nothing beyond 64 frames at 8K had been held against the reference.  The temporal filter is causal, so the
per-frame Q_per_ch of an 80-frame clip are the first 80 frames' of any longer clip: the GPU test scores the FULL 256-frame clip --
several temporal blocks, the machinery configs[4] runs on -- and compares its first 80 frames with this fixture.
Stored: Q_per_ch [1,4,80,9], rho_band, JOD of the 80-frame clip, checksums of the regenerated inputs.  Container only (imports
/root/reference through oracle/ref_shims); 56 minutes on 8 cores; the frames are made on demand (a streamed video_source).

    python oracle/make_goldens_8k80.py          (80 frames)
    python oracle/make_goldens_8k80.py 256      (round 6: the whole clip of configs[4] -> tests/golden/deep_8k_pq_256f.npz, ~3 hours on 8 cores)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, ".."))

import numpy as np
import torch

import pycvvdp
import bench

OUT = os.path.join(HERE, "..", "tests", "golden")
W, H, FPS, DISP = 7680, 4320, 60, "standard_hdr_pq"
F = int(sys.argv[1]) if len(sys.argv) > 1 else 80


class StreamedClip(pycvvdp.video_source.video_source_dm):
    """The bench clip as a video_source that makes its frames on demand (two 8K uint8 clips of 80 frames as arrays, plus what the
    reference copies of them, do not fit this container's 62 GB).  Same arithmetic as video_source_array._get_frame for uint8
    (video_source.py:339-340: code -> float32 / 255, then the display model)."""

    def __init__(self, display_photometry):
        super().__init__(display_photometry=display_photometry)
        self.last = (-1, None, None)
        self.cs_t, self.cs_r, self.seen = 0, 0, set()

    def get_video_size(self):
        return (H, W, F)

    def get_frames_per_second(self):
        return FPS

    def get_batch_size(self):
        return 1

    def _pair(self, f):
        if self.last[0] != f:
            a, b = bench.synth_frame(f, H, W, "cpu")
            a, b = ((x.float() * 0.65 + 0.10 * 255).round().to(torch.uint8) for x in (a, b))     # bench.ResidentClip(pq_range=True): codes in [0.10, 0.75]
            if f not in self.seen:
                self.seen.add(f)
                self.cs_t += int(a.to(torch.int64).sum())
                self.cs_r += int(b.to(torch.int64).sum())
            self.last = (f, a, b)
        return self.last[1], self.last[2]

    def _frame(self, codes, device, colorspace):
        frame = codes[None, :, None].to(device).to(torch.float32) / 255
        return self.apply_dm_and_color_transform(frame, colorspace)

    def get_test_frame(self, frame, device, colorspace="Y"):
        return self._frame(self._pair(frame)[0], device, colorspace)

    def get_reference_frame(self, frame, device, colorspace="Y"):
        return self._frame(self._pair(frame)[1], device, colorspace)


def main():
    t0 = time.time()
    met = pycvvdp.cvvdp(display_name=DISP, device=torch.device("cpu"), quiet=True, heatmap=None)
    vs = StreamedClip(DISP)
    with torch.no_grad():
        jod, stats = met.predict_video_source(vs)
    assert len(vs.seen) == F
    print(f"reference done {time.time() - t0:.0f} s  jod {float(jod):.5f}", flush=True)
    np.savez_compressed(os.path.join(OUT, f"deep_8k_pq_{F}f.npz"), width=W, height=H, frames=F, fps=FPS, display=DISP, dtype="u8",
                        jod=np.float32(jod.item()), Q_per_ch=stats["Q_per_ch"].copy(), rho_band=stats["rho_band"],
                        checksum_test=np.int64(vs.cs_t), checksum_ref=np.int64(vs.cs_r), torch_version=torch.__version__,
                        reference_seconds=np.float32(time.time() - t0))
    print("saved", stats["Q_per_ch"].shape, f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
