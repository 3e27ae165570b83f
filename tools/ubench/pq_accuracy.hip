// How close can the PQ EOTF (display_model.py:58-70) get to torch's CPU fp32 result?  Writes pq2lin(V) for N inputs computed
//   0: as the kernels do today (hardware log2 / exp2 / rcp),
//   1: with V^(1/m) evaluated in fp64 (frexp + hardware log2 of the mantissa + table + degree-5 polynomial) and rounded to fp32,
//      the rest as today,
//   2: as 1, and the final power ^(1/n) the same way, IEEE division.
// Build: hipcc --offload-arch=gfx950 -O3 pq_accuracy.hip -o pq_accuracy ; ./pq_accuracy in.f32 out0.f32 out1.f32 out2.f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_pow(float x, float p) { return fast_exp2(p * fast_log2(x)); }

// a product that is rounded on its own: hipcc contracts a*b - c into an FMA by default, torch does two operations
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
  const float p = a * b;
  return p;
}

__constant__ double kTab[17];   // 2^(-k/16), k = 0..16

// x^p for x in (0, 1], p > 0, result >= 2^-40 or so: fp64 evaluation of 2^(p*log2 x), rounded to fp32
__device__ __forceinline__ float pow_f64(float x, double p) {
  if (x <= 0.0f) return 0.0f;
  const float mant = __builtin_amdgcn_frexp_mantf(x);          // [0.5, 1)
  const int ex = __builtin_amdgcn_frexp_expf(x);
  const double l = (double)ex + (double)fast_log2(mant);        // log2 x, absolute error ~6e-8
  const double y = l * p;                                        // <= 0
  if (y < -126.0) return 0.0f;
  const double yi = floor(y);                                   // integer part
  const double yf = y - yi;                                     // [0, 1)
  const double k16 = rint(yf * 16.0);                           // 0..16
  const double r = yf - k16 * 0.0625;                           // |r| <= 1/32
  const double z = r * 0.69314718055994530942;
  double pz = 1.0 / 120.0;
  pz = pz * z + 1.0 / 24.0;
  pz = pz * z + 1.0 / 6.0;
  pz = pz * z + 0.5;
  pz = pz * z + 1.0;
  pz = pz * z + 1.0;                                            // e^z
  const double t = pz * (1.0 / kTab[(int)k16]);                 // 2^(k/16) * e^z
  return (float)ldexp(t, (int)yi);
}

template <int MODE>
__global__ void k(const float* in, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = in[i];
  const float nn = 0.15930175781250000f, m = 78.843750000000000f;
  const float c1 = 0.83593750000000000f, c2 = 18.851562500000000f, c3 = 18.687500000000000f;
  float L;
  if (MODE == 0) {
    const float t = fast_pow(v, 1.0f / m);
    L = 10000.0f * fast_pow(fmaxf(t - c1, 0.0f) * fast_rcp(c2 - c3 * t), 1.0f / nn);
  } else if (MODE == 1) {
    const float t = pow_f64(v, (double)(1.0f / m));
    L = 10000.0f * fast_pow(fmaxf(t - c1, 0.0f) * fast_rcp(c2 - c3 * t), 1.0f / nn);
  } else if (MODE == 2) {
    const float t = pow_f64(v, (double)(1.0f / m));
    const float r = fmaxf(t - c1, 0.0f) / (c2 - c3 * t);
    L = 10000.0f * pow_f64(r, (double)(1.0f / nn));
  } else if (MODE == 5) {                        // as today, but c3*t rounded before the subtraction (torch does two operations)
    const float t = fast_pow(v, 1.0f / m);
    const float ct = mul_rounded(c3, t);
    L = 10000.0f * fast_pow(fmaxf(t - c1, 0.0f) * fast_rcp(c2 - ct), 1.0f / nn);
  } else if (MODE == 6) {                        // everything as close to torch as it gets
    const float t = pow_f64(v, (double)(1.0f / m));
    const float ct = mul_rounded(c3, t);
    const float r = fmaxf(t - c1, 0.0f) / (c2 - ct);
    L = 10000.0f * pow_f64(r, (double)(1.0f / nn));
  } else if (MODE == 7) {                        // hardware t, unfused, IEEE division, hardware final pow
    const float t = fast_pow(v, 1.0f / m);
    const float ct = mul_rounded(c3, t);
    const float r = fmaxf(t - c1, 0.0f) / (c2 - ct);
    L = 10000.0f * fast_pow(r, 1.0f / nn);
  } else if (MODE == 10) {                       // fp64 t, unfused, hardware rcp and final pow
    const float t = pow_f64(v, (double)(1.0f / m));
    const float ct = mul_rounded(c3, t);
    L = 10000.0f * fast_pow(fmaxf(t - c1, 0.0f) * fast_rcp(c2 - ct), 1.0f / nn);
  } else if (MODE == 8) {                        // r of the mode-6 path
    const float t = pow_f64(v, (double)(1.0f / m));
    const float ct = mul_rounded(c3, t);
    L = fmaxf(t - c1, 0.0f) / (c2 - ct);
  } else if (MODE == 9) {                        // the final power alone, on v taken as r
    L = pow_f64(v, (double)(1.0f / nn));
  } else if (MODE == 3) {
    L = pow_f64(v, (double)(1.0f / m));          // t itself
  } else {
    L = fast_pow(v, 1.0f / m);                   // t as today
  }
  out[i] = L;
}

int main(int argc, char** argv) {
  if (argc < 13) return 1;
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
  const int n = (int)(bytes / 4);
  std::vector<float> h(n); if (fread(h.data(), 4, n, f) != (size_t)n) return 2; fclose(f);
  double tab[17]; for (int k = 0; k <= 16; ++k) tab[k] = exp2(-k / 16.0);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(kTab), tab, sizeof(tab));
  float *din, *dout; (void)hipMalloc(&din, bytes); (void)hipMalloc(&dout, bytes);
  (void)hipMemcpy(din, h.data(), bytes, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 11; ++mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      if (rep == 1) (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 6) hipLaunchKernelGGL(k<6>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 7) hipLaunchKernelGGL(k<7>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 8) hipLaunchKernelGGL(k<8>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 9) hipLaunchKernelGGL(k<9>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 10) hipLaunchKernelGGL(k<10>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3((n + 255) / 256), dim3(256), 0, 0, din, dout, n);
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), dout, bytes, hipMemcpyDeviceToHost);
    FILE* o = fopen(argv[2 + mode], "wb"); fwrite(h.data(), 4, n, o); fclose(o);
    printf("mode %d: %.3f ms per launch of %d values\n", mode, ms / 2, n);
  }
  return 0;
}
