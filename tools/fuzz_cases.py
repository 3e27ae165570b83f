"""The randomised cases of tools/fuzz_shapes.py as a replayable stream: case k of seed s is a pure function of (s, k) -- every draw of
the generator happens here, in a fixed order, whether or not the caller uses it.  Shared by the sweep itself (GPU box), by
oracle/make_goldens_fuzz.py (container: the real reference scores selected cases) and by tests/test_fuzz_goldens.py (GPU box: the
inputs of a pinned case are regenerated from (seed, k) and verified by checksum, so the fixtures hold outputs only).

numpy's Generator(PCG64) stream and its integers / choice / standard_normal / random methods are what the checksums pin."""
import numpy as np


def checksum(a):
    """Order-sensitive 64-bit checksum of an array's bytes (fp16 tensors arrive as numpy views)."""
    b = np.ascontiguousarray(a).view(np.uint8).reshape(-1).astype(np.uint64)
    w = (np.arange(b.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return int((b * w).sum(dtype=np.uint64))


def cases(seed, n, only=None):
    """Yields dicts: k, W, H, F, fps, display, padding, heatmap, dtype ('u8' | 'u16' | 'f16' | 'f32'), B, test, ref (numpy; 'f16' as
    np.float16), block_frames, fuse_mode.  `only`: a set of case indices -- the others are replayed (same draws) but not built or
    yielded."""
    rng = np.random.default_rng(seed)
    for k in range(n):
        W, H = int(rng.integers(16, 700)), int(rng.integers(16, 400))
        F = int(rng.choice([1, 1, 2, 3, 7]))
        fps = 0 if F == 1 else int(rng.choice([24, 30, 50, 60, 120]))
        disp = str(rng.choice(["standard_fhd", "standard_4k", "standard_hdr_pq"]))
        pad = str(rng.choice(["replicate", "symmetric"]))
        heat = str(rng.choice(["none", "none", "raw", "threshold", "supra-threshold"]))
        heat = None if heat == "none" else heat
        build = only is None or k in only
        shape = (1, 3, F, H, W)
        noise = rng.standard_normal(shape)
        dt = str(rng.choice(["u8", "u8", "u16", "f32", "f16"]))
        B = 1 if heat else int(rng.choice([1, 1, 2]))
        noise2 = rng.standard_normal(shape) if B == 2 else None
        lum_only = rng.random() < 0.15
        block_frames = None
        if build:
            y, x = np.mgrid[0:H, 0:W]
            ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (3.1 * x / W + f / 9.0) + c) * np.cos(2 * np.pi * 2.3 * y / H) for c in range(3)])
                            for f in range(F)], axis=1)[None]
            test = np.clip(ref + 0.05 * noise, 0, 1)
            if B == 2:                                        # a batch with a broadcast reference (video_source.py:247-252)
                test = np.concatenate([test, np.clip(test + 0.02 * noise2, 0, 1)], axis=0)
            if lum_only:                                      # luminance-only content
                test, ref = test[:, :1], ref[:, :1]
            if dt == "u8":
                test, ref = np.round(test * 255).astype(np.uint8), np.round(np.clip(ref, 0, 1) * 255).astype(np.uint8)
            elif dt == "u16":
                test, ref = np.round(test * 65535).astype(np.uint16), np.round(np.clip(ref, 0, 1) * 65535).astype(np.uint16)
            elif dt == "f16":
                test, ref = test.astype(np.float16), np.clip(ref, 0, 1).astype(np.float16)
            else:
                test, ref = test.astype(np.float32), np.clip(ref, 0, 1).astype(np.float32)
        block_frames = int(rng.choice([1, 2, 64]))
        fuse_mode = 1 if (F > 1 and not heat and rng.random() < 0.5) else 0
        if build:
            yield dict(k=k, W=W, H=H, F=F, fps=fps, display=disp, padding=pad, heatmap=heat, dtype=dt, B=B, test=test, ref=ref,
                       block_frames=block_frames, fuse_mode=fuse_mode)


def case(seed, k):
    for c in cases(seed, k + 1, only={k}):
        return c


def as_input(a):
    """What the metric classes are handed: fp16 clips as torch tensors (numpy has the dtype, the reference's array source wants torch)."""
    if a.dtype == np.float16:
        import torch
        return torch.tensor(a)
    return a
