// Does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on gfx950?  (hip_ext.h says the flag is not supported on
// GFX9xx boards for hipExtModuleLaunchKernel.)  Two half-GPU kernels of ~2 ms each in one stream: in-order they take the sum,
// side by side the maximum.   hipcc --offload-arch=gfx950 -O3 anyorder_launch.hip -o anyorder_launch
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, b, 1e-7f);
  if (a == 12345.0f) out[0] = a;
}

int main() {
  float* d; hipMalloc(&d, 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1 << 20;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, d, iters);
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, d, iters);
      else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, iters);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%s: %.3f ms\n", mode == 0 ? "two kernels in order" : mode == 1 ? "second kernel with hipExtAnyOrderLaunch" : "one kernel", ms);
    }
  }
  return 0;
}
