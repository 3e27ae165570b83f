// Micro-benchmark: does data written by one kernel come back faster when the next kernel reads it while it still sits in the
// 256 MB Infinity Cache (MALL)?  Kernel A writes X MB (nontemporal or plain stores), kernel B reads the same X MB.
// Build: hipcc --offload-arch=gfx950 -O3 mall_reuse.hip -o mall_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_write(v4f* out, size_t n4, float s) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const v4f v = {s, s + 1, s + 2, (float)i};
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const v4f* in, float* out, size_t n4) {
  v4f acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += NT ? __builtin_nontemporal_load(in + i) : in[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}
int main() {
  const size_t big = (size_t)4 << 30;
  float *buf, *sink, *flush; (void)hipMalloc(&buf, big); (void)hipMalloc(&sink, 4096); (void)hipMalloc(&flush, big);
  (void)hipMemset(buf, 0, big); (void)hipMemset(flush, 0, big);
  hipEvent_t e[4]; for (auto& x : e) (void)hipEventCreate(&x);
  const int blocks = 4096;
  for (int nt = 0; nt < 2; ++nt)
    for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024, 2048}) {
      const size_t n4 = mb * 1024 * 1024 / 16;
      float tw = 0, tr = 0, trc = 0;
      const int reps = 5;
      for (int r = 0; r < reps; ++r) {
        // evict: read 4 GB of something else
        hipLaunchKernelGGL(k_read<true>, dim3(blocks), dim3(256), 0, 0, (const v4f*)flush, sink, big / 16);
        (void)hipEventRecord(e[0]);
        if (nt) hipLaunchKernelGGL(k_write<true>, dim3(blocks), dim3(256), 0, 0, (v4f*)buf, n4, 1.0f);
        else hipLaunchKernelGGL(k_write<false>, dim3(blocks), dim3(256), 0, 0, (v4f*)buf, n4, 1.0f);
        (void)hipEventRecord(e[1]);
        hipLaunchKernelGGL(k_read<false>, dim3(blocks), dim3(256), 0, 0, (const v4f*)buf, sink, n4);     // right after the write
        (void)hipEventRecord(e[2]);
        hipLaunchKernelGGL(k_read<true>, dim3(blocks), dim3(256), 0, 0, (const v4f*)flush, sink, big / 16);   // evict
        (void)hipEventRecord(e[3]);
        hipLaunchKernelGGL(k_read<false>, dim3(blocks), dim3(256), 0, 0, (const v4f*)buf, sink, n4);     // cold
        hipEvent_t e4; (void)hipEventCreate(&e4); (void)hipEventRecord(e4); (void)hipEventSynchronize(e4);
        float a, b, c; (void)hipEventElapsedTime(&a, e[0], e[1]); (void)hipEventElapsedTime(&b, e[1], e[2]); (void)hipEventElapsedTime(&c, e[3], e4);
        if (r > 0) { tw += a; tr += b; trc += c; }
        (void)hipEventDestroy(e4);
      }
      tw /= reps - 1; tr /= reps - 1; trc /= reps - 1;
      const double gb = mb / 1024.0 * 1.073741824;
      printf("%s stores %5zu MB: write %6.2f TB/s   read-after-write %6.2f TB/s   cold read %6.2f TB/s\n", nt ? "nontemporal" : "plain      ", mb,
             gb / tw, gb / tr, gb / trc);
    }
  return 0;
}
