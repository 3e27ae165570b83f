"""Image pairs resident in HBM through cvvdp.predict(): wall time per call against the GPU time of its kernels.
    python tools/image_bench.py [4k|fhd|8k] [n_calls]
Wall time per call = host set-up + ~25 kernel launches; `gpu_ms` (HIP events around the calls, queue kept full) is what the
kernels need when the host keeps ahead."""
import sys
import time
import torch
sys.path.insert(0, ".")
import colorvideovdp_amd as cv

res = {"fhd": (1080, 1920), "4k": (2160, 3840), "8k": (4320, 7680)}[sys.argv[1] if len(sys.argv) > 1 else "4k"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
ref = torch.rand((3, res[0], res[1]), device=dev, generator=g)
test = (ref + 0.03 * torch.randn(ref.shape, device=dev, generator=g)).clamp(0, 1)
m = cv.cvvdp(display_name="standard_4k")
for _ in range(5):
    j, _s = m.predict(test, ref, dim_order="CHW")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(n):
    j, _s = m.predict(test, ref, dim_order="CHW")      # returns stats on the host: one sync per call, like the reference
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
print(f"{res[1]}x{res[0]} image: {wall:.3f} ms per predict() call ({res[0] * res[1] / wall / 1e6:.2f} Gpixel/s), JOD {j.item():.4f}")
