"""configs[3]'s shape on the GPU: a 4K uint8 clip scored by 2, 4 and 8 frame-range ranks, each
holding only its own frames + the 16 real halo frames before them, through the class's own sharded branch (shard plan, halo
run, all-gather, pooling on every rank).

  * 256 frames made with the fixture's CPU generator: every rank's gathered Q_per_ch must equal the REAL reference's
    (tests/golden/bench_4k256_u8.npz, oracle/make_goldens_bench.py) within the parity tolerance, and be bit-identical between
    the 2-, 4- and 8-rank runs and the unsharded run (the causal FIR makes the split exact, cvvdp_metric.py:554-560);
  * 1024 frames (configs[3]'s length) made on the device: 8 ranks against the unsharded run on the same frames, bit for bit.

The ranks share the one GPU of the box (gloo rendezvous); the RCCL flavour of the same all-gather is what bench.py --gpus N runs."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOD_TOL = 1e-3


def _run(world, src, frames, out):
    os.makedirs(out, exist_ok=True)
    # the ranks share this process's GPU: hand back what this process holds for other tests (the session's cache of CPU-generated frames,
    # torch's unused blocks) before eight of them size their blocks from the memory that is free
    import bench
    bench._cpu_frames.clear()
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, SHARD_SRC=src, SHARD_OUT=str(out), SHARD_FRAMES=str(frames), MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    port = 31300 + (os.getpid() % 1500) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "shard4k_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    own = [l for l in (p.stdout + "\n" + p.stderr).splitlines() if ("Error" in l or "error" in l) and "elastic" not in l and "SIGTERM" not in l]
    assert p.returncode == 0, "\n".join(own[-40:]) + "\n---- tail\n" + p.stderr[-1500:]
    return [dict(np.load(os.path.join(out, f"rank{r}.npz"))) for r in range(world)]


@pytest.fixture(scope="module")
def clip_4k256_files():
    """The 4K x 256 uint8 bench clip (CPU generator, verified by the fixture's checksums) as two .npy files the ranks map."""
    import bench
    g = load_golden("bench_4k256_u8")
    F, H, W = int(g["frames"]), int(g["height"]), int(g["width"])
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 16e9 else None
    d = tempfile.mkdtemp(prefix="cvvdp4k_", dir=base)
    try:
        clip = bench.ResidentClip(F, 0, F, H, W, 60, "u8", torch.device("cuda"), gen="cpu")
        ok = (clip.checksum_test, clip.checksum_ref) == (int(g["checksum_test"]), int(g["checksum_ref"]))
        if ok:
            np.save(os.path.join(d, "test.npy"), clip.test.cpu().numpy())
            np.save(os.path.join(d, "ref.npy"), clip.ref.cpu().numpy())
        yield (d if ok else None), clip, g
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_4k256_shards_against_the_reference(clip_4k256_files, tmp_path):
    import colorvideovdp_amd as cv
    d, clip, g = clip_4k256_files
    assert d is not None, "this torch build's CPU generator does not reproduce the fixture's synthetic frames"
    jod1, st1 = cv.cvvdp(display_name="standard_4k").predict_video_source(clip)        # unsharded, same frames
    np.testing.assert_allclose(st1["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    del clip.test, clip.ref
    torch.cuda.empty_cache()
    for world in (2, 4, 8):
        ranks = _run(world, "files:" + d, 256, tmp_path / f"w{world}")
        assert sum(int(r["count"]) for r in ranks) == 256
        for k, r in enumerate(ranks):
            assert int(r["held"]) == int(r["count"]) + (16 if k else 0)                # own frames + real halo, nothing else
            np.testing.assert_allclose(r["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)   # the real reference
            assert abs(float(r["jod"]) - float(g["jod"])) <= JOD_TOL
            np.testing.assert_array_equal(r["Q_per_ch"], st1["Q_per_ch"])             # and exactly the unsharded features
            assert float(r["jod"]) == float(jod1)


def test_4k1024_eight_shards_equal_the_unsharded_run(tmp_path):
    import bench
    import colorvideovdp_amd as cv
    clip = bench.ResidentClip(1024, 0, 1024, 2160, 3840, 60, "u8", torch.device("cuda"), gen="gpu")
    jod1, st1 = cv.cvvdp(display_name="standard_4k").predict_video_source(clip)
    del clip
    torch.cuda.empty_cache()
    ranks = _run(8, "gpugen", 1024, tmp_path / "w8")
    for k, r in enumerate(ranks):
        assert int(r["count"]) == 128 and int(r["held"]) == 128 + (16 if k else 0)
        np.testing.assert_array_equal(r["Q_per_ch"], st1["Q_per_ch"])
        assert float(r["jod"]) == float(jod1)
