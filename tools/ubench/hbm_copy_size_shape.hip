// Micro-benchmark (round 5, VERDICT r4 next #4 ii): why does the best copy on these boxes stop at 5.3 TB/s when MI355X_MICROARCH.md quotes
// 6.29 TB/s for a float4 copy, and can the temporal kernel's stream (3 dword loads, 8 dword stores per lane and frame) have more?
//   A. the float4 copy of hbm_copy_sweep.hip (best shapes) over BUFFER SIZES from 16 MB to 6.4 GB per side: a copy whose two buffers fit the
//      256 MB Infinity Cache is not an HBM measurement;
//   B. the temporal kernel's SHAPE without its arithmetic: thread = pixel walking F frames, 3 input planes read, 8 output planes written per
//      frame, 4 frames prefetched -- with 4-byte accesses per lane (what k_fir_rot does), and with 16-byte accesses (thread = 4 pixels; the
//      window of k_fir_rot would need 4x the registers), plain and nontemporal stores.
// Build: hipcc --offload-arch=gfx950 -O3 hbm_copy_size_shape.hip -o hbm_copy_size_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ void st4(v4f v, v4f* p) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void st1(float v, float* p) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void k_copy_ch(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4) {
  const size_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  size_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) st4<NTS>(v[u] + 1.0f, out + i + u * 256);
  }
  for (; i < hi; i += 256) st4<NTS>(in[i] + 1.0f, out + i);
}
template <bool NTS>
__global__ __launch_bounds__(256) void k_copy_gs(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) st4<NTS>(in[i] + 1.0f, out + i);
}

// B: in [3][F][P] -> out [8][F][P]; thread = V adjacent pixels, walks the frames with 4 frames of loads in flight (static slots)
template <int V, bool NTS>
__global__ __launch_bounds__(256) void k_fir_shape(const float* __restrict__ in, float* __restrict__ out, int P, int F) {
  typedef float vv __attribute__((ext_vector_type(V)));
  const int pix = (blockIdx.x * 256 + threadIdx.x) * V;
  if (pix >= P) return;
  const size_t ip = (size_t)F * P, op = (size_t)F * P;
  vv pf[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 3; ++c) pf[q][c] = *reinterpret_cast<const vv*>(in + c * ip + (size_t)(q < F ? q : F - 1) * P + pix);
  for (int f0 = 0; f0 < F; f0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = f0 + u;
      if (f < F) {
        const vv a = pf[u][0], b = pf[u][1], c = pf[u][2];
        const int fn = f + 4 < F ? f + 4 : F - 1;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) pf[u][cc] = *reinterpret_cast<const vv*>(in + cc * ip + (size_t)fn * P + pix);
#pragma unroll
        for (int pl = 0; pl < 8; ++pl) {
          const vv v = a * (float)(pl + 1) + b * 0.5f + c;
          vv* dst = reinterpret_cast<vv*>(out + pl * op + (size_t)f * P + pix);
          if constexpr (NTS) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
      }
    }
  }
}

__global__ void k_fill(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xFFFFFF) * (1.0f / 16777216.0f);
  }
}
template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); f(); (void)hipEventRecord(e0); for (int k = 0; k < reps; ++k) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return ms / reps;
}
int main() {
  const size_t nmax = (size_t)3840 * 2160 * 8 * 32;      // 8.5 GB: the output side of B (8 planes x 32 frames of 4K)
  float *in, *out; (void)hipMalloc(&in, nmax * 4); (void)hipMalloc(&out, nmax * 4);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, nmax);
  (void)hipMemset(out, 0, nmax * 4);
  (void)hipDeviceSynchronize();
  printf("# A. float4 copy, bytes per side -> TB/s (read + written bytes / time); Infinity Cache = 256 MB\n");
  for (size_t mb : {16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048, 4096, 6400}) {
    const size_t n4 = mb * 1024 * 1024 / 16;
    const int reps = mb <= 512 ? 40 : 8;
    const float a = timeit([&] { hipLaunchKernelGGL((k_copy_ch<true, true, 4>), dim3(4096), dim3(256), 0, 0, (const v4f*)in, (v4f*)out, n4); }, reps);
    const float b = timeit([&] { hipLaunchKernelGGL((k_copy_gs<false>), dim3(1024), dim3(256), 0, 0, (const v4f*)in, (v4f*)out, n4); }, reps);
    const float c = timeit([&] { hipLaunchKernelGGL((k_copy_ch<false, false, 4>), dim3(4096), dim3(256), 0, 0, (const v4f*)in, (v4f*)out, n4); }, reps);
    const double bytes = 2.0 * n4 * 16;
    printf("%6zu MB per side: chunked nt/nt U=4 %6.2f TB/s   grid-stride plain U=1 (1024 blocks) %6.2f TB/s   chunked plain/plain U=4 %6.2f TB/s\n", mb, bytes / a / 1e9, bytes / b / 1e9, bytes / c / 1e9);
    fflush(stdout);
  }
  printf("# B. the temporal kernel's stream shape (3 planes in, 8 planes out per frame, 4 frames prefetched), 4K, F frames: TB/s over 44 B/pixel/frame\n");
  const int P = 3840 * 2160;
  for (int F : {8, 32}) {
    const double bytes = (double)P * F * 44.0;
    const float d1 = timeit([&] { hipLaunchKernelGGL((k_fir_shape<1, true>), dim3((P + 255) / 256), dim3(256), 0, 0, in, out, P, F); }, 6);
    const float d1p = timeit([&] { hipLaunchKernelGGL((k_fir_shape<1, false>), dim3((P + 255) / 256), dim3(256), 0, 0, in, out, P, F); }, 6);
    const float d2 = timeit([&] { hipLaunchKernelGGL((k_fir_shape<2, true>), dim3((P / 2 + 255) / 256), dim3(256), 0, 0, in, out, P, F); }, 6);
    const float d4 = timeit([&] { hipLaunchKernelGGL((k_fir_shape<4, true>), dim3((P / 4 + 255) / 256), dim3(256), 0, 0, in, out, P, F); }, 6);
    const float d4p = timeit([&] { hipLaunchKernelGGL((k_fir_shape<4, false>), dim3((P / 4 + 255) / 256), dim3(256), 0, 0, in, out, P, F); }, 6);
    printf("F=%2d: 4 B/lane nt st %6.2f TB/s (%.3f ms)   4 B/lane plain st %6.2f   8 B/lane nt st %6.2f   16 B/lane nt st %6.2f   16 B/lane plain st %6.2f\n",
           F, bytes / d1 / 1e9, d1, bytes / d1p / 1e9, bytes / d2 / 1e9, bytes / d4 / 1e9, bytes / d4p / 1e9);
    fflush(stdout);
  }
  return 0;
}
