import ast
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _share_cpu_generated_frames():
    """GPU session: the frames bench.py's CPU generator makes (the ones the reference fixtures were made from) are kept on the device
    once made -- a dozen tests score prefixes of the same 4K, 8K and 1080p clips (bench.cpu_generated_frames)."""
    import torch
    if torch.cuda.is_available():
        import bench
        bench.CACHE_CPU_FRAMES = True
    yield


@pytest.fixture(autouse=True)
def _return_unused_device_memory():
    """After a test that left tens of gigabytes in torch's caching allocator (8K clips, 100 GB workspaces) the blocks go back to the
    driver: the next big test's workspace is ONE allocation, and a fragmented cache cannot serve it however much of it is unused."""
    yield
    import torch
    if torch.cuda.is_available() and torch.cuda.memory_reserved() - torch.cuda.memory_allocated() > (16 << 30):
        torch.cuda.empty_cache()


def _names(pattern):
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, pattern)))


def golden_cases():
    """Array-input cases made by oracle/make_goldens.py."""
    return [n for n in _names("*.npz") if n != "setup" and not n.startswith(("yuv", "fullsize", "kat_", "bench_", "deep_", "outputs", "features", "resize_", "fuzz_", "checkpoint"))]


def yuv_cases():
    """Planar Y'CbCr file cases made by oracle/make_goldens_yuv.py."""
    return _names("yuv*.npz")


def resize_cases():
    """.yuv pairs with full_screen_resize, made by oracle/make_goldens_resize.py."""
    return _names("resize_*.npz")


def fuzz_golden_cases():
    """The thin class of the randomised sweep, scored by the real reference (oracle/make_goldens_fuzz.py); inputs are replayed."""
    return _names("fuzz_*.npz")


# ---- observed error margins: the generic tolerances of the parity tests are wide enough to hide a drift
# of the coarse-band Laplacian or a one-bin shift of the heat map's tone curve, so the fixtures that matter carry their own bound =
# 1.5 x what this build was OBSERVED to do on them (tests/golden/observed_bounds.json, made by tools/make_observed_bounds.py from the
# values every GPU run of these tests leaves under gpurun_out/observed/).
_BOUNDS = None


def observed_bound(kind, name):
    global _BOUNDS
    if _BOUNDS is None:
        import json
        path = os.path.join(GOLDEN, "observed_bounds.json")
        _BOUNDS = json.load(open(path)) if os.path.isfile(path) else {}
    return (_BOUNDS.get(kind) or {}).get(name)


def record_observed(kind, name, values):
    import json
    try:
        d = os.path.join(ROOT, "gpurun_out", "observed")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"{kind}__{name}.json"), "w") as f:
            json.dump(values, f)
    except OSError:
        pass


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    if "meta" in d:
        d["meta"] = ast.literal_eval(str(d["meta"]))
    return d


@pytest.fixture(scope="session")
def setup_vectors():
    return dict(np.load(os.path.join(GOLDEN, "setup.npz"), allow_pickle=False))


def fullsize_cases():
    """Reference outputs on prefixes of bench.py's synthetic clips (oracle/make_goldens_fullsize.py); inputs are regenerated."""
    return _names("fullsize*.npz")


def bench_cases():
    """Reference outputs on the FULL clips bench.py times (oracle/make_goldens_bench.py); inputs are regenerated."""
    return _names("bench_*.npz")


def fullsize_inputs(g):
    """The clip prefix a fullsize_* fixture was made from, regenerated on the CPU; None if this torch build's CPU
    generator does not reproduce it (checksums differ)."""
    import torch
    import bench
    F, H, W = int(g["frames"]), int(g["height"]), int(g["width"])
    frames = [bench.synth_frame(f, H, W, "cpu") for f in range(F)]
    t = torch.stack([a for a, _ in frames], dim=1)[None]
    r = torch.stack([b for _, b in frames], dim=1)[None]
    if int(t.to(torch.int64).sum()) != int(g["checksum_test"]) or int(r.to(torch.int64).sum()) != int(g["checksum_ref"]):
        return None
    return t.numpy(), r.numpy()


def kat_wavy_facade():
    """The reference's documented known-answer case (examples/ex_simple_image.py:14-17): fixture + the test image made
    with the example's own recipe (scipy gaussian_filter, sigma 2, mode 'nearest', truncate 2.0; ex_utils.py:27-41)."""
    from scipy.ndimage import gaussian_filter
    g = load_golden("kat_wavy_facade")
    ref = g["ref"]
    test = np.zeros_like(ref)
    for c in range(3):
        test[..., c] = gaussian_filter(ref[..., c], 2, mode="nearest", truncate=2.0)
    return g, test, ref


def kat_nancy_church():
    """The reference's documented HDR known-answer case (examples/ex_hdr_images.py:13-45): fixture (Radiance RGBE bytes),
    reference / test images in cd/m^2 made with the example's recipe, and the display photometry arguments."""
    from scipy.ndimage import gaussian_filter
    g = load_golden("kat_nancy_church")
    rgbe = g["rgbe"]
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(np.float32)   # RGBE: mantissa * 2^(e-136)
    img = rgbe[..., :3].astype(np.float32) * scale[..., None]
    ref = (img / img.max() * 4000 * 4).astype(np.float32)                                              # ex_hdr_images.py:29
    test = np.zeros_like(ref)
    for c in range(3):
        test[..., c] = gaussian_filter(ref[..., c], 2, mode="nearest", truncate=2.0)
    photo = dict(Y_peak=4000, contrast=1000000, source_colorspace="BT.709", EOTF="linear", E_ambient=100)
    return g, test, ref, photo
