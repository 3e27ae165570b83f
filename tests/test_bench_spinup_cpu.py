"""bench.py's untimed spin-up under rank skew, world size 2 and 3, gloo, CPU only.

The bench step ends in a collective, so every rank must run the same number of spin-up steps whatever its own clock says.
The workers below enter the loop up to 0.6 s apart and take rank-dependent time per step -- the conditions under which a
per-rank timed loop leaves the ranks with different iteration counts and hangs them in mismatched collectives."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        calls = []

        def step():                      # what predict_video_source does at its end: one all-gather per step
            time.sleep(0.01 * (1 + 3 * rank))
            recv = [torch.zeros(4) for _ in range(world)]
            dist.all_gather(recv, torch.full((4,), float(rank)))
            calls.append(1)

        time.sleep(0.3 * rank)           # ranks finish making their clips at different times
        n = bench.lockstep_spinup(step, 0.25, world, torch.device("cpu"))
        assert n == len(calls)
        dist.barrier()                   # the bench's next collective: hangs if a rank is still inside an all_gather
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([n], dtype=torch.int64))
        if rank == 0:
            np.save(out, np.array([int(c) for c in counts]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_spinup_runs_the_same_steps_on_every_rank(tmp_path, world):
    out = str(tmp_path / "n.npy")
    ctx = mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=False)
    t0 = time.time()
    while not ctx.join(timeout=1.0):
        assert time.time() - t0 < 90, "spin-up hung"
    n = np.load(out)
    assert n.min() == n.max() and n[0] >= 1, n


def test_spinup_single_rank_needs_no_process_group():
    import bench
    calls = []
    n = bench.lockstep_spinup(lambda: (calls.append(1), time.sleep(0.01)), 0.05, 1, torch.device("cpu"))
    assert n == len(calls) >= 1
