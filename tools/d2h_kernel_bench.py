"""Device -> page-locked host: SDMA copy (tensor.copy_) against a copy KERNEL that stores into the mapped host buffer, idle and beside a busy GPU.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/d2h_kernel.hip -o tools/ubench/libd2h_kernel.so && python tools/d2h_kernel_bench.py"""
import ctypes
import os
import time
import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libd2h_kernel.so"))
lib.d2h_copy_kernel.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
n = 1600 << 20
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
dst = torch.zeros(n, dtype=torch.uint8, device="cpu", pin_memory=True)
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device="cuda")


def busy(k):
    global a
    for _ in range(k):
        a = torch.sin(a) * 1.0001


def run(what, blocks, load):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(4):
        with torch.cuda.stream(side):
            if what == "sdma":
                dst.copy_(src, non_blocking=True)
            else:
                rc = lib.d2h_copy_kernel(src.data_ptr(), dst.data_ptr(), n, blocks, side.cuda_stream)
                assert rc == 0, rc
        if load:
            busy(24)
    side.synchronize()
    t_copy = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return 4 * n / t_copy / 1e9, t_all * 1e3


lib.d2h_copy_kernel(src.data_ptr(), dst.data_ptr(), n, 64, side.cuda_stream)
torch.cuda.synchronize()
assert torch.equal(dst[:1 << 20], src[:1 << 20].cpu()) and torch.equal(dst[-(1 << 20):], src[-(1 << 20):].cpu())
busy(4); torch.cuda.synchronize()
t0 = time.perf_counter(); busy(96); torch.cuda.synchronize(); t_busy = (time.perf_counter() - t0) * 1e3
print(f"96 elementwise kernels alone: {t_busy:.0f} ms")
for load in (False, True):
    r, t = run("sdma", 0, load)
    print(f"SDMA copy {'beside kernels' if load else 'idle GPU      '}: {r:5.1f} GB/s (all done after {t:.0f} ms)")
    for blocks in (16, 32, 64, 128, 256, 1024):
        r, t = run("kernel", blocks, load)
        print(f"copy kernel, {blocks:4d} blocks, {'beside kernels' if load else 'idle GPU      '}: {r:5.1f} GB/s (all done after {t:.0f} ms)")
