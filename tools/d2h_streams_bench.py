"""Device -> page-locked host copy rate with 1, 2 and 4 concurrent copy streams (round 6, VERDICT r5 next #4: is the 52 GB/s of the heat-map
stream the link, or one SDMA engine?).    python tools/d2h_streams_bench.py [MB per copy]"""
import sys
import time
import torch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
n = mb * (1 << 20)
src = torch.empty(n, dtype=torch.uint8, device="cuda")
dst = torch.empty(n, dtype=torch.uint8, device="cpu", pin_memory=True)
for k in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(k)]
    part = n // k
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(4):
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    dst[i * part:(i + 1) * part].copy_(src[i * part:(i + 1) * part], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{mb} MB per copy, {k} stream(s): {4 * n / dt / 1e9:.1f} GB/s", flush=True)
# and beside a busy GPU (a compute kernel stream running): does the rate hold?
a = torch.randn(8192, 8192, device="cuda")
s = torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(4):
    with torch.cuda.stream(s):
        dst.copy_(src, non_blocking=True)
    for _ in range(20):
        a = torch.sin(a) * 1.0001
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"1 stream beside elementwise kernels: >= {4 * n / dt / 1e9:.1f} GB/s (copy and kernels overlapped; total {dt * 1e3:.0f} ms)")
