"""Frame-range sharding over torch.distributed, world size 2, gloo backend, CPU only.

The product code under test is the shard plan and the Q_per_ch all-gather (colorvideovdp_amd/sharding.py);
the per-shard compute engine here is the CPU oracle (test infrastructure), because the HIP engine needs a GPU.
The GPU flavour of the same property (bit-exact shards) is tests/test_gpu_parity.py::test_frame_shards_are_exact."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from colorvideovdp_amd.sharding import all_gather_frames, plan_frame_shard
        from oracle import cvvdp_oracle as orc
        g = load_golden(case)
        meta = g["meta"]
        n_total = g["Q_per_ch"].shape[2]
        first, count = plan_frame_shard(n_total, rank, world)
        o = orc.Oracle(display_name=meta["display"], temp_padding=meta["temp_padding"])
        _, stats = o.predict(g["test"], g["ref"], dim_order=meta["dim_order"], frames_per_second=meta["fps"], first_frame=first, n_frames=count)
        q_local = torch.tensor(stats["Q_per_ch"])
        q_all = all_gather_frames(q_local, n_total)
        jod = o.pool(q_all)
        if rank == 0:
            np.save(out, np.concatenate([q_all.numpy().reshape(-1), np.array([float(jod)])]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["vid_u8_36x64x9_60_sym_short", "vid_u8_lum_60x90x5_24"])
def test_two_rank_frame_shards(tmp_path, case):
    out = str(tmp_path / "q.npy")
    mp.spawn(_worker, args=(2, _free_port(), case, out), nprocs=2, join=True)
    got = np.load(out)
    g = load_golden(case)
    # shards reproduce the reference's unsharded features (oracle tolerance) and its JOD
    np.testing.assert_allclose(got[:-1].reshape(g["Q_per_ch"].shape), g["Q_per_ch"], rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(got[-1], g["jod"], atol=2e-5)
