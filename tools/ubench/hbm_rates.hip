// Micro-benchmark: achievable HBM bandwidth on MI355X for the access shapes the pipeline uses.
// Build: hipcc --offload-arch=gfx950 -O3 hbm_rates.hip -o hbm_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));

// grid-stride read, 16 B per lane
__global__ __launch_bounds__(256) void k_read4(const v4f* in, float* out, size_t n4) {
  v4f acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(in + i);
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}
__global__ __launch_bounds__(256) void k_copy4(const v4f* in, v4f* out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
__global__ __launch_bounds__(256) void k_copy1(const float* in, float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
// FIR-like: a thread owns one pixel and marches over frames; 6 input planes (dword) -> 8 output planes (dword)
__global__ __launch_bounds__(256) void k_march(const float* in, float* out, int P, int frames) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  for (int f = 0; f < frames; ++f) {
    float v[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) v[p] = in[((size_t)p * frames + f) * P + pix];
#pragma unroll
    for (int p = 0; p < 8; ++p) __builtin_nontemporal_store(v[p % 6] + p, &out[((size_t)p * frames + f) * P + pix]);
  }
}
// same traffic, 4 pixels per thread (16-byte accesses)
__global__ __launch_bounds__(256) void k_march4(const v4f* in, v4f* out, int P4, int frames) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= P4) return;
  for (int f = 0; f < frames; ++f) {
    v4f v[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) v[p] = in[((size_t)p * frames + f) * P4 + pix];
#pragma unroll
    for (int p = 0; p < 8; ++p) __builtin_nontemporal_store(v[p % 6] + (float)p, &out[((size_t)p * frames + f) * P4 + pix]);
  }
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipEventRecord(e0); f(); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 2;
}
int main() {
  const int P = 3840 * 2160, frames = 32;
  const size_t n_in = (size_t)6 * frames * P, n_out = (size_t)8 * frames * P;
  float *in, *out; (void)hipMalloc(&in, n_in * 4); (void)hipMalloc(&out, n_out * 4);
  (void)hipMemset(in, 0, n_in * 4); (void)hipMemset(out, 0, n_out * 4);
  for (int blocks : {2048, 8192, 32768}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(256), 0, 0, (const v4f*)in, out, n_in / 4); });
    printf("read  16B/lane  blocks=%6d : %7.3f ms  %6.2f TB/s\n", blocks, ms, n_in * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, 0, (const v4f*)in, (v4f*)out, n_in / 4); });
    printf("copy  16B/lane  blocks=%6d : %7.3f ms  %6.2f TB/s (read+write)\n", blocks, ms, 2 * n_in * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_copy1, dim3(blocks), dim3(256), 0, 0, in, out, n_in); });
    printf("copy   4B/lane  blocks=%6d : %7.3f ms  %6.2f TB/s (read+write)\n", blocks, ms, 2 * n_in * 4 / ms / 1e9);
  }
  float ms = timeit([&] { hipLaunchKernelGGL(k_march, dim3((P + 255) / 256), dim3(256), 0, 0, in, out, P, frames); });
  printf("march  4B/lane 6 in + 8 out planes     : %7.3f ms  %6.2f TB/s\n", ms, (n_in + n_out) * 4 / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(k_march4, dim3((P / 4 + 255) / 256), dim3(256), 0, 0, (const v4f*)in, (v4f*)out, P / 4, frames); });
  printf("march 16B/lane 6 in + 8 out planes     : %7.3f ms  %6.2f TB/s\n", ms, (n_in + n_out) * 4 / ms / 1e9);
  return 0;
}
