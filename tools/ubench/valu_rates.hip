// Micro-benchmark: issue cost of fp32 VALU flavours on gfx950 (cycles per wave64 instruction per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4096;
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = a[i] * 1.0001f + 0.5f;                       // v_fma_f32
      if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.001f);         // v_mul + v_exp
      if (OP == 2) a[i] = __builtin_amdgcn_logf(a[i] + 2.0f);            // v_add + v_log
      if (OP == 3) a[i] = __builtin_amdgcn_rcpf(a[i] + 2.0f);            // v_add + v_rcp
      if (OP == 4) a[i] = fmaxf(a[i] * 1.0001f, 0.25f);                  // v_mul + v_max
      if (OP == 5) a[i] = a[i] * 1.0001f;                                // v_mul
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int per_iter_instr) {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4;   // 4 blocks/CU = 16 waves/CU = 4 waves/SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD = waves/SIMD * ITER * 8 * per_iter_instr ; cycles = ms * clock
  double instr_per_simd = 4.0 * ITER * 8 * per_iter_instr;
  printf("%-14s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, ms, ms * 1e6 / instr_per_simd);
  (void)hipFree(d);
}
int main() {
  run<0>("fma", 1); run<5>("mul", 1); run<4>("mul+max", 2); run<1>("mul+exp", 2); run<2>("add+log", 2); run<3>("add+rcp", 2);
  return 0;
}
