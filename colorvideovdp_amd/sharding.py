"""Frame-range data parallelism (not present in the reference, which is single-device).

Per-frame features Q_per_ch[b, c, f, band] depend only on frame f and its filter_len-1 predecessors
(causal temporal FIR, cvvdp_metric.py:554-560), so a clip shards exactly by frame range: rank r scores
frames [start_r, start_r + count_r) and reads filter_len-1 real halo frames before start_r (rank 0 pads
instead).  The only exchange is one all-gather of the tiny Q_per_ch shards (<= 150 KB for 1024 frames),
after which every rank pools to the same JOD.  With the "nccl" backend this is RCCL over xGMI; the
message is latency-bound, so no bucketing or overlap is needed.
"""
import torch
import torch.distributed as dist


def plan_frame_shard(n_frames, rank, world):
    """(first, count) of rank's contiguous frame range; ranges differ by at most one frame."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_frames, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def all_gather_frames(q_local, n_frames, group=None):
    """q_local [B, C, count_r, bands] on each rank -> [B, C, n_frames, bands] on every rank."""
    world = dist.get_world_size(group)
    B, C, _, L = q_local.shape
    cap = (n_frames + world - 1) // world
    send = torch.zeros((B, C, cap, L), dtype=q_local.dtype, device=q_local.device)
    send[:, :, :q_local.shape[2]] = q_local
    dev = send.device
    if send.is_cuda and dist.get_backend(group) == "gloo":   # gloo has no all_gather on device tensors: the shards are tiny
        send = send.cpu()
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    recv = [x.to(dev) for x in recv]
    parts = []
    for r in range(world):
        _, cnt = plan_frame_shard(n_frames, r, world)
        parts.append(recv[r][:, :, :cnt])
    return torch.cat(parts, dim=2)
