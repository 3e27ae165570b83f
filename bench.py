#!/usr/bin/env python3
"""Benchmark of the ColorVideoVDP hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--workload 4k64|fhd64|4k256|4k1024|8k256pq] [--dtype f32|u8|yuv420p8|yuv420p10]
                    [--heatmap none|raw|threshold|supra-threshold] [--distogram] [--gen cpu|gpu]
    python bench.py --gpus N ...            # N > 1 without a launcher: starts its own ranks through torch.distributed.run (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full predict() of the workload clip pair (display model -> DKL -> temporal FIR ->
contrast pyramid -> CSF -> masking -> pooling -> JOD) with the test/reference clips already resident in
HBM.  Clips are sharded over the N GPUs by frame range (each rank: its frames + a 16-frame real halo, one RCCL
all-gather of Q_per_ch, pooling on every rank):
  * 4k64 (the BASELINE.json metric clip), fhd64, 4k256: frames PER GPU -- the clip grows with N (weak scaling);
  * 4k1024 (configs[3]) and 8k256pq (configs[4], PQ, supra-threshold heat map streamed off the GPU + distogram): frames in
    TOTAL -- every rank scores 1/N of the clip (strong scaling).

Prints ONE JSON line (rank 0).  value = Mpixel/s of the whole job = W*H*frames_scored_by_all_ranks*K / t.  With one GPU
and a clip the real reference was run on (tests/golden/bench_*.npz, oracle/make_goldens_bench.py) the frames are made with
the CPU generator the fixture was made with, and the line carries jod_delta_vs_reference of the very clip it times.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The CPU frame generator runs many OpenMP teams; idle OpenMP workers that keep spinning afterwards steal the cores the
# HIP runtime's launch path needs (measured: kernels queue late and a 19 ms step takes 40 ms).  Must be set before torch loads.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # multi-process GPU work: the host driver only supports dmabuf IPC (RCCL needs it)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np
import torch

WORKLOADS = {
    # name: (W, H, frames, fps, display, frames are per GPU?, default dtype, default heat map)
    "4k64": (3840, 2160, 64, 60, "standard_4k", True, "f32", "none"),       # BASELINE.json metric: "4K@60fps 64-frame clip"
    "fhd64": (1920, 1080, 64, 60, "standard_fhd", True, "f32", "none"),     # configs[1]
    "4k256": (3840, 2160, 256, 60, "standard_4k", True, "f32", "none"),     # configs[2]
    "4k1024": (3840, 2160, 1024, 60, "standard_4k", False, "u8", "none"),   # configs[3]: 1024 frames over the GPUs of the node
    "8k256pq": (7680, 4320, 256, 60, "standard_hdr_pq", False, "u8", "supra-threshold"),   # configs[4]: + heat map + distogram
}
GOLDEN = {("4k64", "f32"): "bench_4k64_f32", ("fhd64", "f32"): "bench_fhd64_f32", ("4k256", "u8"): "bench_4k256_u8"}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy)
# whole-path algorithmic bytes per pixel of the test clip.  "survey": SURVEY.md 8(d)'s model, which includes a DKL ring
# (24 B written + 24 B read) that this design does not have and reads every pyramid level twice; "build": what this build's
# kernels have to move (build_bytes_per_pixel below)
PATH_BYTES_PER_PIXEL = {"f32": 211.0, "u8": 193.0, "yuv420p8": 190.0, "yuv420p10": 193.0}
INPUT_BYTES_PER_PIXEL = {"f32": 24.0, "u8": 6.0, "yuv420p8": 3.0, "yuv420p10": 6.0}
BAND0_BYTES_PER_PIXEL = 40.0   # level-0 band kernel: g0 in (8 planes x 4 B) + g1 (8 x 4 / 4) in (k_band4) or OUT (k_band4f, which computes it)


def stage_bytes_per_pixel(dtype, fused_levels):
    """Algorithmic bytes per pixel of the test clip, per kernel family.  FIR: samples in + 8 level-0 planes out.  A level l (1/4^l
    of the pixels, 32 B per pixel of the level) whose band kernel is fused (band4f.hip, the first `fused_levels` levels) is read once
    and its next level written once by that kernel (40 / 4^l); from the first unfused level on, every level is read by a reduce pass
    (32/4^l) that writes the next one (8/4^l), and read again with its coarse level by its band kernel (40/4^l)."""
    F = max(int(fused_levels), 0)
    tail = (4.0 / 3.0) / 4 ** F                      # sum over l >= F of 1/4^l
    fused = sum(40.0 / 4 ** l for l in range(F))
    b = {"temporal_fir": INPUT_BYTES_PER_PIXEL[dtype] + 32.0,
         "pyr_reduce": 32.0 * tail + 32.0 * (tail - 1.0 / 4 ** F),
         "band_level0": 40.0,
         "band_rest": fused + 40.0 * tail - 40.0}
    return b


def code_stamp():
    """What ties profiles/traffic.json (counters from separate rocprofv3 --pmc passes) to the code that is being timed: the
    SHA-256 of the kernel sources + C header (the build is reproducible from them) and of the built library itself."""
    import hashlib
    src = hashlib.sha256()
    cs = os.path.join(ROOT, "colorvideovdp_amd", "csrc")
    files = sorted(os.path.join(cs, n) for n in os.listdir(cs) if n.endswith((".hip", ".h", ".cpp")) or n == "Makefile")
    for p in files + [os.path.join(ROOT, "include", "cvvdp_hip.h")]:
        src.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            src.update(f.read())
    lib = os.path.join(ROOT, "colorvideovdp_amd", "libcvvdp_hip.so")
    lib_sha = None
    if os.path.isfile(lib):
        with open(lib, "rb") as f:
            lib_sha = hashlib.sha256(f.read()).hexdigest()
    return {"csrc_sha256": src.hexdigest(), "lib_sha256": lib_sha}


def host_cpu():
    """(model string, physical cores, logical cpus) of the box's host processor(s)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    phys = None
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:
        pass
    return model, phys or os.cpu_count() or 1, os.cpu_count() or 1


def synth_frame(f, H, W, device, seed_ref=1234, seed_noise=5678):
    """One synthetic test/reference frame pair (uint8 [3,H,W]); depends only on the frame index, so a
    rank can generate exactly its own frame range (SURVEY.md 8(d) recipe: moving plaid + smoothed hash
    noise; test = ref + sigma 0.02 noise + 3 % flicker every 8th frame + 3-tap blur on the right half)."""
    g = torch.Generator(device=device)
    y = torch.arange(H, device=device, dtype=torch.float32).view(1, H, 1)
    x = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, W)
    c = torch.arange(3, device=device, dtype=torch.float32).view(3, 1, 1)
    g.manual_seed(seed_ref * 100003 + f)
    u = torch.rand((1, 3, H, W), generator=g, device=device)
    u = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(u, (1, 1, 1, 1), mode="replicate"), 3, stride=1)[0]
    ref = 0.5 + 0.22 * torch.sin(2 * np.pi * (3 * x / W + f / 60.0)) * torch.cos(2 * np.pi * 2 * y / H + 0.7 * c) + 0.18 * (u - 0.5)
    ref = torch.round(ref.clamp(0, 1) * 255) / 255
    g.manual_seed(seed_noise * 100003 + f)
    test = ref + 0.02 * torch.randn((3, H, W), generator=g, device=device)
    if f % 8 == 0:
        test = test * 0.97
    half = W // 2
    blur = (torch.nn.functional.pad(test[None], (1, 1, 0, 0), mode="replicate")[0].unfold(2, 3, 1).mean(-1))
    test = torch.cat([test[:, :, :half], blur[:, :, half:]], dim=2)
    return torch.round(test.clamp(0, 1) * 255).to(torch.uint8), torch.round(ref * 255).to(torch.uint8)


def cpu_frame_pool(frame_ids, H, W):
    """Iterator over synth_frame(f, H, W, "cpu") for f in frame_ids, made concurrently: ~1 s per 4K frame single-threaded, so
    several frames at a time (torch releases the GIL), each with a share of the cores (no nested oversubscription; elementwise
    torch operators on 25 M samples stop scaling at a handful of threads, so: many workers with four threads each).
    Returns (iterator, pool, torch thread count to restore after pool.shutdown())."""
    import concurrent.futures
    ncpu = os.cpu_count() or 1
    workers = max(1, min(32, ncpu // 4))
    threads_before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(threads_before, ncpu // workers)))
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers)
    return pool.map(lambda f: synth_frame(f, H, W, "cpu"), frame_ids), pool, threads_before


# Frames of the CPU generator already made in this process, on the device: (H, W) -> dict(test, ref: uint8 [3, n, H, W]; sum_t, sum_r: per-frame
# checksums).  Off in bench runs (a clip is made once); tests/conftest.py turns it on for the GPU session, where a dozen tests score
# prefixes of the same three clips (4K x 256, 8K x 80, 1080p x 64) against fixtures of the real reference: 900 4K frames at ~0.25 s
# each were a fifth of the suite's time.
CACHE_CPU_FRAMES = False
_cpu_frames = {}


def cpu_generated_frames(H, W, lo, hi, device):
    """Frames lo .. hi-1 of the CPU generator (the one the reference fixtures were made with) as uint8 device tensors [3, hi-lo, H, W]
    (test, ref) and their checksums (sums of all codes; taken on the device: the int64 conversion of 100 M samples on the host cost
    as much as making the frame)."""
    key = (H, W, str(device))
    c = _cpu_frames.get(key) if CACHE_CPU_FRAMES else None
    have = 0 if c is None else c["test"].shape[1]
    start = have if (c is not None and lo <= have) else lo          # extend the cached prefix, or make just this range
    if c is None or hi > have:
        n_new = hi - start
        t = torch.empty((3, n_new, H, W), dtype=torch.uint8, device=device)
        r = torch.empty((3, n_new, H, W), dtype=torch.uint8, device=device)
        made, pool, threads_before = cpu_frame_pool(range(start, hi), H, W)
        for k, (a, b) in enumerate(made):
            t[:, k], r[:, k] = a.to(device), b.to(device)
        pool.shutdown(wait=True)
        torch.set_num_threads(threads_before)
        st, sr = t.sum(dim=(0, 2, 3), dtype=torch.int64).tolist(), r.sum(dim=(0, 2, 3), dtype=torch.int64).tolist()
        if c is not None and start == have:
            c = {"test": torch.cat([c["test"], t], dim=1), "ref": torch.cat([c["ref"], r], dim=1), "sum_t": c["sum_t"] + st, "sum_r": c["sum_r"] + sr, "first": 0}
        else:
            c = {"test": t, "ref": r, "sum_t": st, "sum_r": sr, "first": start}
        if CACHE_CPU_FRAMES and c["first"] == 0:
            _cpu_frames[key] = c
    a, b = lo - c["first"], hi - c["first"]
    return c["test"][:, a:b], c["ref"][:, a:b], sum(c["sum_t"][a:b]), sum(c["sum_r"][a:b])


class ResidentClip:
    """video source whose frames [lo, hi) live in HBM; implements the raw-block fast path.  gen="cpu": the frames are made
    with the CPU generator (the one the reference fixtures were made with) and uploaded; "gpu": made on the device (a
    different random stream, much faster for long clips).  pq_range: map the codes into [0.10, 0.75] (about 0.3 .. 1000
    cd/m^2 on a PQ display).  checksum_test / checksum_ref (gen="cpu"): sums of the generator's codes (before the PQ-range
    map), what the fixtures record."""

    def __init__(self, n_total, lo, hi, H, W, fps, dtype, device, gen="gpu", pq_range=False):
        self.n_total, self.lo, self.hi, self.H, self.W, self.fps = n_total, lo, hi, H, W, fps
        tdt = torch.float32 if dtype == "f32" else torch.uint8
        self.code = 3 if dtype == "f32" else 0
        self.device_resident = True
        self.test = torch.empty((1, 3, hi - lo, H, W), dtype=tdt, device=device)
        self.ref = torch.empty((1, 3, hi - lo, H, W), dtype=tdt, device=device)
        self.checksum_test = self.checksum_ref = 0
        src = None
        if gen == "cpu":
            src_t, src_r, self.checksum_test, self.checksum_ref = cpu_generated_frames(H, W, lo, hi, device)
            src = (src_t, src_r)
        for f in range(lo, hi):
            t, r = (src[0][:, f - lo], src[1][:, f - lo]) if src is not None else synth_frame(f, H, W, device)
            if pq_range:
                t, r = ((x.float() * 0.65 + 0.10 * 255).round().to(torch.uint8) for x in (t, r))
            if dtype == "f32":
                t, r = t.float() / 255, r.float() / 255
            self.test[0, :, f - lo] = t
            self.ref[0, :, f - lo] = r

    def get_video_size(self):
        return (self.H, self.W, self.n_total)

    def get_frames_per_second(self):
        return self.fps

    def get_batch_size(self):
        return 1

    def get_raw_block(self, a, b, device):
        assert self.lo <= a and b <= self.hi, (a, b, self.lo, self.hi)
        return self.test[:, :, a - self.lo:b - self.lo], self.ref[:, :, a - self.lo:b - self.lo], self.code


class ResidentYuvClip:
    """Planar Y'CbCr 4:2:0 clip (BT.709, limited range, 8 or 10 bit) resident in HBM, laid out exactly like a .yuv file
    (frame = Y plane, U plane, V plane); implements the raw-block hook of colorvideovdp_amd.video_source_yuv_file.
    Made from the same synthetic frames as ResidentClip (BT.709 forward matrix, 2x2 box chroma)."""

    def __init__(self, n_total, lo, hi, H, W, fps, bit_depth, device):
        from colorvideovdp_amd import _capi
        self.n_total, self.lo, self.hi, self.H, self.W, self.fps = n_total, lo, hi, H, W, fps
        self.dm_photometry = None
        self.device_resident = True               # no PCIe transfers: the metric may use full-size blocks
        self.frame = H * W + 2 * (H // 2) * (W // 2)
        tdt = torch.uint8 if bit_depth == 8 else torch.int16
        self.bufs = [torch.empty((hi - lo) * self.frame, dtype=tdt, device=device) for _ in range(2)]
        s = float(2 ** (bit_depth - 8))
        for f in range(lo, hi):
            for k, rgb in enumerate(synth_frame(f, H, W, device)):
                rgb = rgb.float() / 255
                Y = 0.2126 * rgb[0] + 0.7152 * rgb[1] + 0.0722 * rgb[2]
                cb, cr = (rgb[2] - Y) / 1.8556, (rgb[0] - Y) / 1.5748
                c2 = torch.nn.functional.avg_pool2d(torch.stack([cb, cr])[None], 2)[0]
                codes = torch.cat([(16 + 219 * Y).flatten(), (128 + 224 * c2[0]).flatten(), (128 + 224 * c2[1]).flatten()])
                self.bufs[k][(f - lo) * self.frame:(f - lo + 1) * self.frame] = torch.round(codes * s).to(tdt)
        self.fmt = _capi.YuvFormat()
        self.fmt.chroma, self.fmt.bit_depth, self.fmt.matrix = 420, bit_depth, 709
        self.fmt.frame_stride_test = self.fmt.frame_stride_ref = self.frame

    def get_video_size(self):
        return (self.H, self.W, self.n_total)

    def get_frames_per_second(self):
        return self.fps

    def get_batch_size(self):
        return 1

    def get_raw_yuv_block(self, a, b, device):
        assert self.lo <= a and b <= self.hi, (a, b, self.lo, self.hi)
        sl = slice((a - self.lo) * self.frame, (b - self.lo) * self.frame)
        return self.bufs[0][sl], self.bufs[1][sl], self.fmt


def cpu_baseline(W, H, fps, display, n_frames, frames=None):
    """SURVEY 8(d) "CPU baseline": the oracle ('port' of the reference's torch-CPU path: same conv2d-based op structure, block = 1
    frame) on the first n_frames of the same synthetic clip, torch on all PHYSICAL cores of the node; CPU model and core count
    are reported.  The figure extrapolates linearly in frames (every frame costs the same: the window is a ring)."""
    from oracle import cvvdp_oracle as orc
    model, phys, logical = host_cpu()
    threads_before = torch.get_num_threads()
    torch.set_num_threads(phys)
    try:
        if frames is None:
            made, pool, tb = cpu_frame_pool(range(n_frames), H, W)
            frames = list(made)
            pool.shutdown(wait=True)
            torch.set_num_threads(phys)
        t = torch.stack([a for a, _ in frames], dim=1).float().div(255)[None]   # [1,3,F,H,W]
        r = torch.stack([b for _, b in frames], dim=1).float().div(255)[None]
        o = orc.Oracle(display)
        t0 = time.time()
        with torch.no_grad():
            jod, _ = o.predict(t, r, dim_order="BCFHW", frames_per_second=fps)
        dt = time.time() - t0
    finally:
        torch.set_num_threads(threads_before)
    return dict(value=W * H * n_frames / dt / 1e6, unit="Mpixel/s", cores=phys, kind="port", cpu_model=model, logical_cpus=logical,
                sample=f"first {n_frames} frames of the {W}x{H}@{fps} workload clip (a prefix: the figure extrapolates linearly in frames), "
                       f"oracle/cvvdp_oracle.py (torch CPU, block=1, {phys} threads = physical cores), {dt:.1f} s",
                context="the REAL reference in the build container (8 threads, BASELINE.md 2): 0.97 Mpixel/s on 1920x1080 x 24 frames, "
                        "0.63 Mpixel/s on 3840x2160 x 4 frames"), float(jod), t, r


def device_identity(rank, dev_index):
    """(rank, device index, uuid / PCI bus id) of the GPU this rank runs on -- gathered over all ranks into `collectives.ranks_seen`."""
    p = torch.cuda.get_device_properties(dev_index)
    uuid = getattr(p, "uuid", None)
    bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    return {"rank": rank, "device_index": dev_index, "uuid": str(uuid) if uuid is not None else None, "pci": bus, "name": p.name,
            "host": os.uname().nodename}


def count_distinct_devices(ranks_seen):
    """How many different GPUs the ranks sit on.  A device is (host, device index, PCI bus id, uuid) TOGETHER: a stack that reports the
    same -- or no -- uuid for every GPU must not turn eight ranks on eight GPUs into "one device" (the RCCL line would be refused), and
    ranks pinned to one GPU (the test hook) agree in all four."""
    return len({(r["host"], r["device_index"], r["pci"], r["uuid"]) for r in ranks_seen})


def measured_copy_ceiling(working_set_mb=None):
    """The rate a float4 device copy (read + write) reaches on a gpurun box AT THE WORKING-SET SIZE of the step: the newest
    profiles/*ubench_hbm_copy_size_shape.txt (tools/ubench/hbm_copy_size_shape.hip, section A: bytes per side -> TB/s for three access
    patterns).  Copies whose two buffers fit the 256 MB Infinity Cache run at 6.3-7.2 TB/s; a pass over gigabytes -- what every kernel
    of this path is -- gets 5.4-6.0.  The row at (or just above) `working_set_mb` per side is taken, the largest row when it is None or
    beyond the sweep.  The second denominator SURVEY 8(d) asks for."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*ubench_hbm_copy_size_shape.txt")))
    if not files:
        return None
    rows = []
    for line in open(files[-1]):
        mt = re.match(r"\s*(\d+) MB per side:(.*)", line)
        if mt:
            rates = [float(x) for x in re.findall(r"([0-9.]+) TB/s", mt.group(2))]
            if rates:
                rows.append((int(mt.group(1)), max(rates)))
    if not rows:
        return None
    rows.sort()
    pick = next((r for r in rows if working_set_mb is not None and r[0] >= working_set_mb), rows[-1])
    return {"GBs": pick[1] * 1e3, "MB_per_side": pick[0], "working_set_MB_per_side": None if working_set_mb is None else round(working_set_mb),
            "source": "profiles/" + os.path.basename(files[-1]) + f" (best of the three float4 copies at {pick[0]} MB per side)"}


def power_probe(step, sync, seconds=1.6):
    """Shader clock and socket power while the step runs back to back (untimed, after the timed region): `rocm-smi --showclocks
    --showpower --json` sampled in a side thread.  The 4K step runs at the socket's power limit (profiles/r04_dev_notes.txt 4): the
    clock it is allowed, not the 2.4 GHz of the data sheet, is what the VALU-issue bound of the band kernels has to be priced at.
    Returns None where rocm-smi is missing or prints something else."""
    import subprocess
    import threading
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
                card = next(iter(json.loads(txt).values()))
                sclk = next(v for k, v in card.items() if k.lower().startswith("sclk clock speed"))
                watts = next(v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower())
                samples.append((time.perf_counter(), float(str(sclk).strip("()MmHhZz")), float(watts)))
            except Exception:      # noqa
                return
            time.sleep(0.03)

    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        step()
        n += 1
    sync()
    t1 = time.perf_counter()
    stop[0] = True
    th.join(timeout=6)
    use = [s for s in samples if t0 + 0.6 <= s[0] <= t1]          # power and clock settle within ~0.4 s of load
    if len(use) < 3:
        return None
    return {"sclk_MHz_under_load": round(sum(s[1] for s in use) / len(use), 1), "socket_W_under_load": round(sum(s[2] for s in use) / len(use), 1),
            "samples": len(use), "steps_run": n, "sclk_MHz_peak": 2400.0,
            "note": "rocm-smi beside an untimed loop of the same step, first 0.6 s dropped; the step draws the socket's power limit and the "
                    "shader clock is what that limit leaves (the guide's VALU peak assumes 2.4 GHz)"}


def lockstep_spinup(step, seconds, world, flag_device, sync=lambda: None, collective=None):
    """Untimed spin-up: run step() for about `seconds`, the SAME number of times on every rank.  Multi-rank, every step() ends in
    a collective (the all-gather of Q_per_ch), so a per-rank clock must not decide when to stop: ranks enter the loop at different
    times and would leave it after different iteration counts -- one rank then sits in barrier() while another is still in
    all_gather, and the job hangs (round 2's bench did exactly that).  The ranks line up at a barrier first, and after every
    step rank 0's clock decides for everybody (one broadcast word).  Returns the number of steps run."""
    import torch.distributed as dist
    collective = world > 1 if collective is None else collective
    if collective:
        dist.barrier()
    t0 = time.perf_counter()
    n = 0
    while True:
        more = torch.tensor([1 if time.perf_counter() - t0 < seconds else 0], dtype=torch.int32, device=flag_device)
        if collective:
            dist.broadcast(more, src=0)
        if int(more.item()) == 0:
            return n
        step()
        sync()
        n += 1


def self_launch(n):
    """Re-run this command line under `python -m torch.distributed.run --nproc-per-node n` (the driver's N=1 command
    shape, `python bench.py --gpus N ...`, must also produce the N>1 lines).  Returns the launcher's exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:       # a free loopback port
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def _capi_build_info():
    from colorvideovdp_amd import _capi
    return _capi.build_info()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="4k64", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, choices=["f32", "u8", "yuv420p8", "yuv420p10"], help="input sample format (default: the workload's)")
    ap.add_argument("--heatmap", default=None, choices=["none", "raw", "threshold", "supra-threshold"], help="default: the workload's")
    ap.add_argument("--heatmap-format", default="u8", choices=["u8", "f16"],
                    help="what the heat-map sink takes: the writers' 8-bit RGB frames made on the GPU (3 B/pixel over PCIe, the default: "
                         "what a heat-map video / PNG sequence needs) or the reference's fp16 planes (6 B/pixel)")
    ap.add_argument("--heatmap-sink", default="host", choices=["host", "device"],
                    help="where the heat-map frames go: page-locked host memory over PCIe (the reference's semantics: 8K x 256 8-bit RGB frames "
                         "are 25.5 GB = 0.49 s of the link, more than the kernels) or a consumer on the GPU (nothing crosses PCIe in the step)")
    ap.add_argument("--distogram", action="store_true", help="also write the distogram of the last step (default for 8k256pq)")
    ap.add_argument("--gen", default=None, choices=["cpu", "gpu"], help="frame generator (default: cpu when a reference fixture exists for the clip)")
    ap.add_argument("--frames", type=int, default=None, help="override the workload's frame count (per GPU or total)")
    ap.add_argument("--block-frames", type=int, default=None)
    ap.add_argument("--score-frames", type=int, default=None, help="heat-map clips resident in HBM: frames per band / heat-map piece of a long temporal block (default: the metric's, 16)")
    ap.add_argument("--fps", type=int, default=None, help="override the workload's frame rate (e.g. 120: 31-tap temporal filters, k_fir_fused)")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the CPU baseline sample: a prefix of the workload clip (SURVEY 8d: 16; 0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the 1.6 s of untimed steps under rocm-smi after the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher (the command shape of the N=1 line): become the launcher.  One rank per GPU
        # through torch.distributed.run on the loopback address, same arguments; rank 0 prints the JSON line, this process only
        # forwards the exit code.  The GPU count is checked HERE, before any rendezvous, unless the test hook pins the ranks.
        if "CVVDP_BENCH_DEVICE" not in os.environ and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this node shows {torch.cuda.device_count()}")
        raise SystemExit(self_launch(args.gpus))
    # test hooks: CVVDP_BENCH_DEVICE pins every rank to one GPU and CVVDP_BENCH_BACKEND=gloo replaces RCCL, so that the
    # multi-rank path (shard plan, halo frames, all-gather, rank-0 JSON) can be exercised on a single-GPU box
    dev_index = int(os.environ.get("CVVDP_BENCH_DEVICE", local_rank))
    backend = os.environ.get("CVVDP_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if "CVVDP_BENCH_DEVICE" not in os.environ and have < max(args.gpus, world):
        # one rank per GPU: fail before any rendezvous, with one line, instead of dying in set_device / inside RCCL on some rank
        raise SystemExit(f"bench.py: --gpus {max(args.gpus, world)} needs {max(args.gpus, world)} visible GPUs, this node shows {have}")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # CVVDP_BENCH_FORCE_DIST=1 (test hook, launched through torch.distributed.run with one process): initialise the process group and run
    # every collective of the multi-rank path -- barrier, broadcast, the all-gather of Q_per_ch, the MAX all-reduce -- with world size 1.
    # With the default backend that is the RCCL code path on the one GPU of a test box (tests/test_bench_multirank.py).
    dist_on = world > 1 or os.environ.get("CVVDP_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend)

    ranks_seen = None
    if dist_on:
        # which device every rank really sits on (the SCALE record shows N distinct GPUs, or says that the ranks share one)
        ranks_seen = [None] * world
        torch.distributed.all_gather_object(ranks_seen, device_identity(rank, dev_index))
        distinct = count_distinct_devices(ranks_seen)
        if backend == "nccl" and distinct != world:
            # a SCALE line whose RCCL ranks share GPUs is not a scaling measurement: every rank refuses, before any clip is made
            raise SystemExit(f"bench.py: {world} RCCL ranks sit on {distinct} distinct GPUs ({[r['pci'] for r in ranks_seen]}): one rank per GPU is the contract")

    import colorvideovdp_amd as cv
    from colorvideovdp_amd.heatmap_writers import HeatmapFrameMeans
    from colorvideovdp_amd.sharding import plan_frame_shard
    W, H, frames, fps, display, per_gpu_frames, wl_dtype, wl_heat = WORKLOADS[args.workload]
    frames = args.frames or frames
    fps = args.fps or fps
    dtype = args.dtype or wl_dtype
    heat = args.heatmap or wl_heat
    heat = None if heat == "none" else heat
    distogram = args.distogram or args.workload == "8k256pq"
    n_total = frames * world if per_gpu_frames else frames
    scaling = "weak" if per_gpu_frames else "strong"
    first, count = plan_frame_shard(n_total, rank, world)
    m = cv.cvvdp(display_name=display, device=device, block_frames=args.block_frames, heatmap=heat)
    if args.score_frames is not None:
        m.score_frames = args.score_frames
    fl = int(np.ceil(0.250 * fps / 2) * 2) + 1   # cvvdp_metric.py:1059
    lo = max(0, first - (fl - 1))
    golden = None
    if world == 1 and args.frames is None and args.fps is None and (args.workload, dtype) in GOLDEN:
        gpath = os.path.join(ROOT, "tests", "golden", GOLDEN[(args.workload, dtype)] + ".npz")
        if os.path.isfile(gpath):
            golden = np.load(gpath, allow_pickle=False)
    gen = args.gen or ("cpu" if golden is not None else "gpu")
    if dtype.startswith("yuv"):
        clip = ResidentYuvClip(n_total, lo, first + count, H, W, fps, 8 if dtype.endswith("p8") else 10, device)
    else:
        clip = ResidentClip(n_total, lo, first + count, H, W, fps, dtype, device, gen=gen, pq_range=args.workload == "8k256pq")
    if dist_on:
        m.set_frame_sharding("world")
    sink = HeatmapFrameMeans(uint8=args.heatmap_format == "u8", device=args.heatmap_sink == "device") if heat is not None else None      # the heat map leaves the GPU block by block (bounded host memory)

    def step():
        return m.predict_video_source(clip, heatmap_sink=sink) if sink is not None else m.predict_video_source(clip)

    # The GPU has been idle while the clip was made (tens of seconds with the CPU generator) and sits in a low-power state: its
    # shader clock takes a few hundred milliseconds of load to come back up, far longer than W warm-up steps of 19 ms.  Measured
    # on a fresh box: the VALU-bound kernels ran 1.6-2x slower through the first ~7 steps.  So the device is first kept busy for
    # about a second (untimed, like the warm-up steps that follow).
    n_spin = lockstep_spinup(step, float(os.environ.get("CVVDP_BENCH_SPINUP_S", "1.0")), world,
                             device if backend == "nccl" else torch.device("cpu"), torch.cuda.synchronize, collective=dist_on)
    if sink is not None:
        sink.frames_seen = 0          # count the streamed heat-map frames of the warm-up + timed steps only
    for _ in range(args.warmup):
        jod, stats = step()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    if not args.no_profile:
        m.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        jod, stats = step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0            # this rank's K steps, before it waits for the others
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    prof = None if args.no_profile else m.profile_read()
    m.profile(False)
    power = power_probe(step, torch.cuda.synchronize) if (world == 1 and not args.no_profile and not args.no_power_probe) else None
    per_rank = None
    if dist_on:
        # what makes the first hardware SCALE line explain itself: every rank's own time for the K steps (the value
        # uses the MAX of the barrier-to-barrier times, as the contract says), and the step's only collective timed on its own
        gdev = device if backend == "nccl" else torch.device("cpu")
        mine = torch.tensor([dt, dt_own], device=gdev, dtype=torch.float64)
        allt = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allt, mine)
        per_rank = [[float(x[0]), float(x[1])] for x in allt]
        dt = max(x[0] for x in per_rank)
        from colorvideovdp_amd.sharding import all_gather_frames
        q_local = torch.zeros((1, 4, count, int(stats["Q_per_ch"].shape[-1])), dtype=torch.float32, device=device)
        n_ag = 20
        all_gather_frames(q_local, n_total)           # (first call: communicator / buffers)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        ta = time.perf_counter()
        for _ in range(n_ag):
            all_gather_frames(q_local, n_total)
        torch.cuda.synchronize()
        allgather_us = (time.perf_counter() - ta) / n_ag * 1e6
    if rank != 0:
        if dist_on:
            torch.distributed.destroy_process_group()
        return
    pixels = W * H * n_total * args.steps
    mpix = pixels / dt / 1e6
    what = f"{W}x{H} {fps}fps, {display}, {dtype} input resident in HBM"
    if world > 1:
        what += (f", {frames}-frame clip pair per GPU ({n_total} frames total)" if per_gpu_frames else f", {n_total}-frame clip pair over {world} GPUs") \
                + f", frame-range shards + {fl - 1}-frame halo, one all-gather of Q_per_ch"
    else:
        what += f", {n_total}-frame clip pair"
    if heat is not None:
        what += f", {heat} heat map " + ("streamed to the host in blocks" if args.heatmap_sink == "host" else "consumed on the GPU in blocks")
    out = {
        "metric": "Mpixels/s", "value": round(mpix, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": cv.COMPUTE_DTYPE, "data": "synthetic",
        "config": {"workload": what, "name": args.workload, "frames_total": n_total, "frames_this_gpu": count, "input_dtype": dtype,
                   "arithmetic": cv.COMPUTE_DTYPE + " in every kernel (8 / 10 / 16-bit and Y'CbCr samples are unpacked to fp32 by the first kernel; "
                                 "block partial sums are finished in f64)",
                   "library_build": "%s (compiled HIP %d, runtime HIP %d)" % _capi_build_info(),
                   "heatmap": heat or "none", "frame_generator": gen, "block_frames": getattr(m, "last_block_frames", None)},
        "jod": round(float(jod), 5), "spinup_steps": n_spin,
    }
    if dist_on:
        distinct = count_distinct_devices(ranks_seen)
        halo = [min(fl - 1, plan_frame_shard(n_total, r, world)[0]) for r in range(world)]
        out["config"]["collectives"] = {"backend": torch.distributed.get_backend(), "world": world,
                                        "per_step": "one all-gather of the Q_per_ch shards", "spin_up": "barrier + one broadcast word per step",
                                        "ranks_seen": ranks_seen, "distinct_devices": distinct,
                                        "allgather_us": round(allgather_us, 1),
                                        "allgather_note": f"the step's only collective ([1,4,{count},bands] fp32 per rank) timed alone, {n_ag} calls back to back "
                                                          "after the timed region (host clock around synchronize)",
                                        "halo_frames_per_rank": halo,
                                        "halo_note": f"real frames before the shard that only pass through the temporal kernel (rank 0 pads instead); "
                                                     f"{fl - 1} of {count} = {(fl - 1) / max(count, 1):.1%} more frames to read and convert on ranks > 0"}
        out["per_rank_ms"] = [round(x[1] / args.steps * 1e3, 3) for x in per_rank]
        out["per_rank_ms_note"] = ("every rank's own K steps / K, before the closing barrier (ms_per_step is the MAX over ranks of the barrier-to-barrier "
                                   "time, as the contract says); the spread says which rank was slow, rank 0 carries no halo frames")
        out["config"]["per_rank_ms"] = out["per_rank_ms"]      # (the driver's record keeps `config`; unknown top-level keys only by name)
    if golden is not None and gen == "cpu":
        if (clip.checksum_test, clip.checksum_ref) == (int(golden["checksum_test"]), int(golden["checksum_ref"])):
            q, qr = stats["Q_per_ch"].astype(np.float64), golden["Q_per_ch"].astype(np.float64)
            out["jod_reference"] = round(float(golden["jod"]), 5)
            out["jod_delta_vs_reference"] = float(abs(float(jod) - float(golden["jod"])))
            out["q_per_ch_max_rel_err_vs_reference"] = float(np.max(np.abs(q - qr) / (np.abs(qr) + 1e-6)))
            # the parity tests' criterion (tests/test_gpu_parity.py: rtol 2e-4, atol 2e-6): 1.0 = at the tolerance
            out["q_per_ch_max_err_over_test_tolerance"] = float(np.max(np.abs(q - qr) / (2e-4 * np.abs(qr) + 2e-6)))
            out["reference_fixture"] = f"tests/golden/{GOLDEN[(args.workload, dtype)]}.npz (the real reference on this very clip, oracle/make_goldens_bench.py)"
            # BASELINE.json's metric is "Mpixels/s + JOD delta vs reference": the second half rides in `config`, which the driver's record keeps
            out["config"]["parity"] = {k: out[k] for k in ("jod_reference", "jod_delta_vs_reference", "q_per_ch_max_rel_err_vs_reference",
                                                           "q_per_ch_max_err_over_test_tolerance", "reference_fixture")}
            out["config"]["parity"]["tolerance"] = "JOD |delta| <= 1e-3 (north_star); Q_per_ch rtol 2e-4 + atol 2e-6 (this build's tests)"
        else:
            out["jod_delta_vs_reference"] = None      # this torch build's CPU generator does not reproduce the fixture's frames
    fused_levels = m.fused_levels
    stage_b = stage_bytes_per_pixel(dtype, fused_levels)
    surv, build = PATH_BYTES_PER_PIXEL[dtype], sum(stage_b.values())
    out["config"]["fused_levels"] = fused_levels    # leading pyramid levels whose band kernel computes the next level itself (no reduce pass)
    out["path_roofline"] = {
        "survey_model": {"bytes_per_pixel": surv, "achieved_GBs": round(surv * pixels / dt / 1e9 / world, 1),
                         "frac_of_8TBs_per_gpu": round(surv * pixels / dt / 1e9 / world / HBM_PEAK_GBS, 4),
                         "note": "SURVEY.md 8(d): includes a 48 B/pixel DKL ring this design does not have and reads every level twice"},
        "build_algorithmic": {"bytes_per_pixel": round(build, 1), "achieved_GBs": round(build * pixels / dt / 1e9 / world, 1),
                              "frac_of_8TBs_per_gpu": round(build * pixels / dt / 1e9 / world / HBM_PEAK_GBS, 4),
                              "note": "fewer algorithmic bytes than round 2 (162.6 for f32): fused band kernels read a level once, so the "
                                      "same pixel rate is a smaller fraction" if fused_levels > 0 else "no fused levels for this clip"},
    }
    ceil = measured_copy_ceiling(build * W * H * count / 2 / 1e6)      # bytes read ~ bytes written per step: one "side" of a copy
    if ceil is not None:
        for k in ("survey_model", "build_algorithmic"):
            out["path_roofline"][k]["frac_of_measured_copy_per_gpu"] = round(out["path_roofline"][k]["achieved_GBs"] / ceil["GBs"], 4)
        out["path_roofline"]["measured_copy_ceiling"] = ceil
    if prof is not None:
        ms, n = prof["band_level0"]
        frames_per_launch = count * args.steps / max(n, 1)
        avg_ms = ms / max(n, 1)
        ach = BAND0_BYTES_PER_PIXEL * W * H * frames_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, ktraffic, traffic_note = None, None, "profiles/traffic.json not found"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            stamp, have = code_stamp(), tj.get("stamp") or {}
            if tj.get("workload") != args.workload or tj.get("dtype") != dtype:
                traffic_note = f"profiles/traffic.json holds counters of workload {tj.get('workload')}/{tj.get('dtype')}, not of this one"
            elif have.get("csrc_sha256") != stamp["csrc_sha256"]:
                # counters are collected in separate rocprofv3 --pmc passes (tools/refresh_profiles.sh), not in this run: they are only
                # quoted when they were measured on exactly these kernel sources
                traffic_note = ("profiles/traffic.json was measured on other kernel sources (csrc hash "
                                f"{str(have.get('csrc_sha256'))[:12]} != {stamp['csrc_sha256'][:12]}): counters dropped, refresh them with "
                                "tools/refresh_profiles.sh + tools/make_traffic_json.py")
            else:
                ktraffic = tj.get("kernels")
                traffic = (ktraffic or {}).get("band_level0", {}).get("hbm_bytes_per_launch")
                traffic_note = ("FETCH_SIZE x 2 + WRITE_SIZE per launch from separate --pmc passes over these kernel sources (stamp matches; "
                                f"library binary {'identical' if have.get('lib_sha256') == stamp['lib_sha256'] else 'rebuilt from the same sources'})")
        sq0 = ((ktraffic or {}).get("band_level0") or {}).get("sq") or {}
        # `bound` is what the counters of the stamped passes say holds the kernel back: VALU issue when its HBM traffic is within 1.25 x of the
        # algorithmic bytes, the pipes are >= 60 % busy and the achieved rate is below half the HBM peak; "hbm" otherwise and whenever
        # there are no counters for these sources.  achieved / peak / frac stay the HBM accounting of the bench contract either way.
        valu_bound = bool(traffic and sq0.get("valu_busy") and traffic <= 1.25 * BAND0_BYTES_PER_PIXEL * W * H * frames_per_launch
                          and sq0["valu_busy"] >= 0.6 and ach < 0.5 * HBM_PEAK_GBS)
        out["roofline"] = {"bound": "valu" if valu_bound else "hbm",
                           "bound_note": ("VALU issue at the shader clock the socket power limit leaves (counters: traffic = %.2f x algorithmic bytes, VALU %.0f %% busy); "
                                          "achieved / peak / frac below are the HBM accounting" % (traffic / (BAND0_BYTES_PER_PIXEL * W * H * frames_per_launch), 100 * sq0["valu_busy"]))
                                         if valu_bound else "HBM accounting (no counters stamped to these sources say otherwise)",
                           "kernel": ("k_band4s level 0 (front waves: ring, 5x5 reduce to level 1, expand, luminance terms; back waves: contrast, CSF, masking, "
                                                      "blurs, pooling) + k_band4s_edge (the same layout's EDGE body) on the border strips as a second launch beside it: the pair is timed") if fused_levels > 0 else
                                                     "k_band4<4> level 0 (fused expand/contrast/CSF/masking/blur/pooling)",
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "traffic_note": traffic_note, "avg_launch_ms": round(avg_ms, 4), "launches": n,
                           "algorithmic_bytes_per_launch": BAND0_BYTES_PER_PIXEL * W * H * frames_per_launch}
        sq = ((ktraffic or {}).get("band_level0") or {}).get("sq")
        if sq:
            # what holds this kernel back when it is not the memory system (same stamped counter passes)
            out["roofline"]["valu"] = {"busy": sq.get("valu_busy"), "instructions_per_launch": sq.get("valu_instructions_per_launch"),
                                       "waves_waiting_frac": sq.get("waves_waiting_frac"), "workgroups": sq.get("workgroups"),
                                       "source": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x SQ_BUSY_CYCLES / 32), profiles/*_pmc_sq_counters.txt (tools/sq_counters.sh)"}
        # the other two dominant kernels on the same footing (algorithmic bytes per step / HIP-event time per step / 8 TB/s)
        px_step = W * H * count
        in_b = INPUT_BYTES_PER_PIXEL[dtype]
        out["kernel_roofline"] = {}
        for key, bpp, what in (("temporal_fir", stage_b["temporal_fir"], f"{in_b:g} B/pixel in + 32 out (8 level-0 planes)"),
                               ("pyr_reduce", stage_b["pyr_reduce"], f"levels {fused_levels}.. read once, levels {fused_levels + 1}.. written once (the fused levels have no reduce pass)"),
                               ("band_level0", stage_b["band_level0"], "g0 32 B/pixel in + g1 8 " + ("out" if fused_levels > 0 else "in")),
                               ("band_rest", stage_b["band_rest"], "levels 1..: (32 + 8)/3 B/pixel")):
            kms = prof[key][0] / args.steps
            if kms > 0:
                gbs = bpp * px_step / (kms * 1e-3) / 1e9
                out["kernel_roofline"][key] = {"ms_per_step": round(kms, 3), "algorithmic_bytes_per_pixel": round(bpp, 2), "what": what,
                                               "achieved_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4)}
        if ktraffic:
            out["kernel_traffic"] = ktraffic     # counter bytes vs algorithmic bytes of every dominant kernel (profiles/traffic.json)
            if tj.get("step_valu"):
                # the other roofline of this path: VALU issue slots (same stamped counter passes, all kernels of one step)
                out["path_roofline"]["valu_issue"] = tj["step_valu"]
                if "valu" in out["roofline"]:
                    out["roofline"]["valu"]["frac_of_issue"] = out["roofline"]["valu"].get("busy")
        if power is not None:
            out["power"] = power
            v = out["roofline"].get("valu")
            if v and v.get("instructions_per_launch") and ktraffic:
                # the band kernels' other bound, at the clock the power limit leaves: VALU quad-cycles of the launch pair / 1024 SIMDs
                quad = sum(s.get("valu_quad_cycles", 0) for s in ((ktraffic.get("band_level0") or {}).get("sq_all") or []))
                if quad > 0:
                    bound_ms = quad * 4 / 1024 / (power["sclk_MHz_under_load"] * 1e3)
                    v["issue_bound_ms_at_load_clock"] = round(bound_ms, 3)
                    v["frac_of_issue_bound_at_load_clock"] = round(bound_ms / avg_ms, 4)
        tot = sum(v[0] for v in prof.values())
        out["kernel_ms_per_step"] = {k: round(v[0] / args.steps, 3) for k, v in prof.items()}
        out["kernel_ms_per_step"]["sum"] = round(tot / args.steps, 3)
    if sink is not None:
        out["heatmap_frames_streamed_per_step"] = sink.frames_seen // (args.steps + args.warmup)
        out["config"]["heatmap_sink"] = ("HeatmapFrameMeans: every frame crosses PCIe into page-locked memory, the host then reads 1/1024 of "
                                         "its pixels (a writer's encode / file cost is NOT in the figure)") if args.heatmap_sink == "host" else \
                                        "HeatmapFrameMeans(device=True): the frames are consumed on the GPU (per-frame means), nothing crosses PCIe in the step"
        if args.heatmap_sink == "host":
            hm_bytes = (1 if heat == "raw" else 3) * (1 if sink.wants_uint8 else 2) * W * H * count
            out["pcie"] = {"heatmap_bytes_per_step": hm_bytes, "GBs_if_the_step_were_only_this_copy": round(hm_bytes / (dt / args.steps) / 1e9, 1),
                           "note": "the heat map's D2H stream bounds this step when the figure approaches what the link sustains (~52 GB/s measured on these boxes)"}
        out["config"]["heatmap_sink_format"] = "uint8 RGB frames (as written to .mp4 / .png), 3 B/pixel D2H" if sink.wants_uint8 else "fp16 planes, 6 B/pixel D2H"
    if distogram:
        try:
            path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"bench_distogram_{args.workload}.png")
            m.export_distogram(stats, path, jod_max=10)
            out["distogram_bytes"] = os.path.getsize(path)
        except Exception as e:   # matplotlib is optional on the box
            out["distogram_bytes"] = None
            out["distogram_error"] = repr(e)
    if args.cpu_frames > 0 and world == 1:
        cb, ojod, t, r = cpu_baseline(W, H, fps, display, args.cpu_frames)
        out["cpu_baseline"] = cb
        hjod, _ = cv.cvvdp(display_name=display, device=device).predict(t, r, dim_order="BCFHW", frames_per_second=fps)
        out["jod_delta_vs_oracle_sample"] = float(abs(float(hjod) - ojod))
    print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
