// Micro-benchmark 2: issue cost of individual gfx950 VALU instructions, pinned with inline asm so that the
// compiler cannot pack or fuse them.  ns per wave64 instruction per SIMD, 4 waves/SIMD, 8 independent chains.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 16384;
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[8]; v2f p[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
  const float m = 1.0001f + seed * 1e-6f, c = 0.5f * seed;
  const v2f mm = {m, m}, cc = {c, c};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
      if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
      if (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(mm), "v"(cc));
      if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(c));
      if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
      if (OP == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 12) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (OP == 13) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(c));
      if (OP == 14) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 15) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 16) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(m));
      if (OP == 17) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
      if (OP == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(mm), "v"(cc));
      if (OP == 19) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (OP == 20) asm volatile("v_cvt_i32_f32_e32 %0, %0" : "+v"(a[i]));
      if (OP == 21) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(c));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name) {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4;
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = 4.0 * ITER * 8;
  printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / instr_per_simd);
  (void)hipFree(d);
}
int main() {
  run<0>("v_fma_f32"); run<9>("v_fma_f32 (sgpr src)"); run<1>("v_pk_fma_f32"); run<8>("v_pk_fma_f32 op_sel splat");
  run<6>("v_mul_f32"); run<7>("v_pk_mul_f32"); run<5>("v_max_f32"); run<10>("v_cndmask_b32"); run<11>("v_mov_b32");
  run<2>("v_exp_f32"); run<3>("v_log_f32"); run<4>("v_rcp_f32");
  run<12>("v_fmac_f32_e32"); run<13>("v_fmac_f32_e32 sgpr"); run<14>("v_add_f32_e32"); run<15>("v_min_f32_e32");
  run<16>("v_med3_f32"); run<17>("v_pk_add_f32"); run<18>("v_pk_fma_f32 sgpr pair"); run<19>("v_mul_f32_e64");
  run<20>("v_cvt_i32_f32"); run<21>("v_lshl_add_u32");
  run<0>("v_fma_f32 (again)"); run<6>("v_mul_f32 (again)"); run<5>("v_max_f32 (again)");
  return 0;
}
