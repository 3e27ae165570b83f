#!/usr/bin/env python3
"""tests/golden/deep_8k_pq_256f.npz, made RESUMABLY: the REAL reference's per-frame scores of the whole 256-frame clip of configs[4]
(7680x4320, standard_hdr_pq, uint8 codes in the PQ range, no heat map), assembled from
  * frames 0..79     tests/golden/deep_8k_pq_80f.npz (oracle/make_goldens_8k80.py, 56 minutes of the reference's CPU path), and
  * frames 80..255   the reference run on the SUB-CLIP of frames s-20 .. 255 (s = the first frame still missing): its temporal filter is
                     causal with 17 taps (cvvdp_metric.py:538-560), so sub-clip frame k >= 16 sees exactly the frames the whole clip's
                     frame s-20+k sees, through the same operations -> the same bits.  The first 16 sub-clip frames carry the replicate
                     padding of the sub-clip's start and are dropped; the next 4 overlap with what is already known and MUST equal it
                     bit for bit (asserted as soon as they exist: the check of the window argument itself).
Why: `make_goldens_8k80.py 256` is one three-hour call of predict_video_source() that leaves nothing behind when the container it runs in
is replaced (it was, twice in round 6).  Here every frame's Q_per_ch row is taken from process_block_of_frames() as it is made (the CPU
path scores one frame per block, cvvdp_metric.py:353-355) and a partial file is rewritten every 8 frames; a later call continues from it.
JOD of the whole clip = the reference's own do_pooling_and_jods() on the assembled Q_per_ch (cvvdp_metric.py:610-644), which is all
predict_video_source() does with it (:398).  Container only (imports /root/reference through oracle/ref_shims).

    THP_MEM_ALLOC_ENABLE=1 python oracle/make_goldens_8k256_resume.py [last_frame_exclusive=256]
    python oracle/make_goldens_8k256_resume.py N      (after stopping a run: writes deep_8k_pq_<N>f.npz from the partial file, N a multiple of 8 it holds)

(THP_MEM_ALLOC_ENABLE=1: torch's CPU allocator asks for transparent huge pages.  Without it this container spends 98 % of the run's CPU time
in the kernel -- eight threads faulting fresh 4 KB pages of 130 MB tensors into one address space -- and a frame takes 95 s instead of 42.
The overlap check at the start of every window would show it if the alignment changed a single bit.)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)

import numpy as np
import torch

import pycvvdp
import make_goldens_8k80 as g80

OUT = os.path.join(HERE, "..", "tests", "golden")
PARTIAL = os.path.join(OUT, "deep_8k_pq_256f.partial.npz")
F_ALL = int(sys.argv[1]) if len(sys.argv) > 1 else 256
HALO, OVERLAP, PREFIX = 16, 4, 80


class Window(g80.StreamedClip):
    """Frames start .. F_ALL-1 of the bench clip as a clip of their own."""

    def __init__(self, display_photometry, start):
        super().__init__(display_photometry)
        self.start = start

    def get_video_size(self):
        return (g80.H, g80.W, F_ALL - self.start)

    def _pair(self, f):
        return super()._pair(f + self.start)


def main():
    t0 = time.time()
    g = np.load(os.path.join(OUT, "deep_8k_pq_80f.npz"))
    known = g["Q_per_ch"].copy()                                     # [1,4,n,9]
    cs = {}                                                          # frame -> (checksum_test, checksum_ref) of the frames made here
    if os.path.isfile(PARTIAL):
        p = np.load(PARTIAL)
        np.testing.assert_array_equal(p["Q_per_ch"][:, :, :known.shape[2]], known)
        known = p["Q_per_ch"].copy()
        cs = {int(f): (int(a), int(b)) for f, a, b in zip(p["cs_frames"], p["cs_t"], p["cs_r"])}
        print(f"resuming: {known.shape[2]} frames known", flush=True)
    if known.shape[2] > F_ALL:                                        # asked for a shorter prefix than what is known: cut (the fixture of a stopped run)
        known = known[:, :, :F_ALL].copy()
        cs = {f: v for f, v in cs.items() if f < F_ALL}
    n0 = known.shape[2]
    if n0 < F_ALL:
        start = n0 - OVERLAP - HALO
        met = pycvvdp.cvvdp(display_name=g80.DISP, device=torch.device("cpu"), quiet=True, heatmap=None)
        vs = Window(g80.DISP, start)
        rows = []
        inner = met.process_block_of_frames

        def save_partial():
            fr = sorted(cs)
            np.savez_compressed(PARTIAL + ".tmp.npz", Q_per_ch=known, cs_frames=np.array(fr, dtype=np.int64),
                                cs_t=np.array([cs[f][0] for f in fr], dtype=np.int64), cs_r=np.array([cs[f][1] for f in fr], dtype=np.int64))
            os.replace(PARTIAL + ".tmp.npz", PARTIAL)

        def hooked(*a, **k):
            nonlocal known
            q, hm = inner(*a, **k)
            assert q.shape[2] == 1                                    # one frame per block on the CPU path
            k_sub = len(rows)
            rows.append(q.detach().cpu().numpy().copy())
            fr = start + k_sub
            if k_sub < HALO:
                print(f"halo frame {fr}  ({time.time() - t0:.0f} s)", flush=True)
            if HALO <= k_sub < HALO + OVERLAP:
                np.testing.assert_array_equal(rows[-1][:, :, 0], known[:, :, fr], err_msg=f"window argument fails at frame {fr}")
                print(f"frame {fr}: equals the known row bit for bit  ({time.time() - t0:.0f} s)", flush=True)
            elif k_sub >= HALO + OVERLAP:
                assert fr == known.shape[2]
                known = np.concatenate([known, rows[-1]], axis=2)
                if known.shape[2] % 8 == 0 or known.shape[2] == F_ALL:
                    for f in sorted(vs.per_frame):
                        if f >= n0 and f < known.shape[2]:
                            cs[f] = vs.per_frame[f]
                    save_partial()
                    print(f"{known.shape[2]} frames  ({time.time() - t0:.0f} s)", flush=True)
            return q, hm

        # per-frame checksums of the frames made here (the whole clip's checksum = the 80-frame fixture's + frames 80..255's)
        vs.per_frame = {}
        pair0 = vs._pair

        def pair(f):
            a, b = pair0(f)
            fa = f + start
            if fa not in vs.per_frame:
                vs.per_frame[fa] = (int(a.to(torch.int64).sum()), int(b.to(torch.int64).sum()))
            return a, b

        vs._pair = pair
        met.process_block_of_frames = hooked
        with torch.no_grad():
            met.predict_video_source(vs)
        for f in sorted(vs.per_frame):
            if n0 <= f < F_ALL:
                cs[f] = vs.per_frame[f]
    assert known.shape[2] == F_ALL and sorted(cs) == list(range(PREFIX, F_ALL)), (known.shape, len(cs))
    met = pycvvdp.cvvdp(display_name=g80.DISP, device=torch.device("cpu"), quiet=True, heatmap=None)
    with torch.no_grad():
        jod = met.do_pooling_and_jods(torch.as_tensor(known))
    jod = float(torch.as_tensor(jod[0] if isinstance(jod, tuple) else jod).reshape(-1)[0])
    np.savez_compressed(os.path.join(OUT, f"deep_8k_pq_{F_ALL}f.npz"), width=g80.W, height=g80.H, frames=F_ALL, fps=g80.FPS, display=g80.DISP, dtype="u8",
                        jod=np.float32(jod), Q_per_ch=known, rho_band=g["rho_band"],
                        checksum_test=np.int64(int(g["checksum_test"]) + sum(v[0] for v in cs.values())),
                        checksum_ref=np.int64(int(g["checksum_ref"]) + sum(v[1] for v in cs.values())), torch_version=torch.__version__,
                        assembled="frames 0-79: deep_8k_pq_80f.npz; frames 80-255: windows of the same clip (oracle/make_goldens_8k256_resume.py)")
    print("saved", known.shape, f"jod {jod:.5f}", f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
