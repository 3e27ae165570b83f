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


def test_bench_gpus_n_without_a_launcher_launches_itself(monkeypatch):
    """`python bench.py --gpus N` (the command shape of the N=1 line) must start its own ranks instead of exiting.
    No GPU here, so the re-exec is checked at the subprocess boundary: one torch.distributed.run command on the loopback address with
    N processes and this very command line, and the launcher's exit code handed through."""
    import subprocess
    import sys

    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.setenv("CVVDP_BENCH_DEVICE", "0")          # the test hook that pins every rank to one GPU: no device-count refusal
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_gpus_n_on_a_smaller_node_refuses_with_one_line(monkeypatch):
    """... and the fail-fast line for "fewer GPUs than N" stays: no launcher is started (this container shows 0 GPUs)."""
    import subprocess
    import sys

    import bench
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CVVDP_BENCH_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "needs 8 visible GPUs, this node shows 0" in str(e.value.code)
