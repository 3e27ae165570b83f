"""Worker of tests/test_sharding_gpu.py: one rank of a frame-range-sharded cvvdp run (gloo rendezvous, every rank on the
same GPU).  Launched through torch.distributed.run; writes its results to $SHARD_OUT/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import load_golden
    import colorvideovdp_amd as cv
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    g = load_golden(os.environ["SHARD_CASE"])
    meta = g["meta"]
    n_frames = int(os.environ.get("SHARD_FRAMES", "0"))
    heat = os.environ.get("SHARD_HEATMAP") or None
    t, r = g["test"], g["ref"]
    if n_frames:                      # a clip shorter than the number of ranks: some shards are empty
        fdim = meta["dim_order"].index("F")
        t, r = np.take(t, range(n_frames), axis=fdim), np.take(r, range(n_frames), axis=fdim)
    kw = dict(dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    m = cv.cvvdp(display_name=meta["display"], heatmap=heat, temp_padding=meta["temp_padding"], device="cuda:0")
    m.set_frame_sharding("world")
    jod, stats = m.predict(t, r, **kw)
    out = dict(jod=np.float32(float(jod)), Q_per_ch=stats["Q_per_ch"], rank=rank, world=world)
    if heat:
        out["heatmap"] = stats["heatmap"].numpy()
        out["heatmap_frame_range"] = np.asarray(stats.get("heatmap_frame_range", (0, stats["N_frames"])))
    np.savez(os.path.join(os.environ["SHARD_OUT"], f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
