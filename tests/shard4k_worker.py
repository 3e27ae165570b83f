"""Worker of tests/test_sharding_4k_gpu.py: one rank of a frame-range-sharded run over a 4K uint8 clip (gloo rendezvous, every
rank on GPU 0).  The rank holds ONLY its own frames + the fl-1 halo frames before them (as a rank of an 8-GPU job would):
from the .npy files the parent test made with the fixture's CPU generator (SHARD_SRC=files:<dir>) or made on the device
(SHARD_SRC=gpugen).  Writes $SHARD_OUT/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    import colorvideovdp_amd as cv
    from colorvideovdp_amd.sharding import plan_frame_shard
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    W, H, fps = 3840, 2160, 60
    n_total = int(os.environ["SHARD_FRAMES"])
    first, count = plan_frame_shard(n_total, rank, world)
    lo = max(0, first - 16)
    src = os.environ["SHARD_SRC"]
    if src == "gpugen":
        clip = bench.ResidentClip(n_total, lo, first + count, H, W, fps, "u8", dev, gen="gpu")
    else:
        d = src.split(":", 1)[1]
        clip = bench.ResidentClip.__new__(bench.ResidentClip)
        clip.n_total, clip.lo, clip.hi, clip.H, clip.W, clip.fps, clip.code, clip.device_resident = n_total, lo, first + count, H, W, fps, 0, True
        for name in ("test", "ref"):
            mm = np.load(os.path.join(d, name + ".npy"), mmap_mode="r")          # [1,3,F,H,W] uint8
            setattr(clip, name, torch.from_numpy(np.ascontiguousarray(mm[:, :, lo:first + count])).to(dev))
    # (up to eight ranks share the one GPU of the box here and use the product's default block policy: long blocks are sized from an
    # absolute share of the device, and a workspace that cannot be had any more falls back to 64-frame blocks and below)
    m = cv.cvvdp(display_name="standard_4k", device=dev)
    m.set_frame_sharding("world")
    jod, stats = m.predict_video_source(clip)
    np.savez(os.path.join(os.environ["SHARD_OUT"], f"rank{rank}.npz"), jod=np.float32(float(jod)), Q_per_ch=stats["Q_per_ch"],
             first=first, count=count, held=clip.test.shape[2])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
