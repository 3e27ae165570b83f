// Micro-benchmark (round 3): what a read+write stream can reach on MI355X, and with which access shape.
// MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; round 1's hbm_rates.hip (nontemporal loads AND stores, one 16-byte
// access in flight per thread) measured 4.7-5.1 TB/s, and k_fir_rot / k_reduce2 were tuned against that figure.
// Sweep: load flavour (plain / nontemporal) x store flavour x independent 16-byte accesses in flight per thread (1, 2, 4, 8)
//        x grid (1024 .. 16384 blocks, grid-stride) x mapping (grid-stride interleaved / one contiguous chunk per block).
// Build: hipcc --offload-arch=gfx950 -O3 hbm_copy_sweep.hip -o hbm_copy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NTL> __device__ __forceinline__ v4f ld(const v4f* p) { if constexpr (NTL) return __builtin_nontemporal_load(p); else return *p; }
template <bool NTS> __device__ __forceinline__ void st(v4f v, v4f* p) { if constexpr (NTS) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride: consecutive blocks touch consecutive 4 KB pieces; U independent accesses in flight per thread, U*stride apart
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void k_copy_gs(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NTL>(in + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NTS>(v[u] + 1.0f, out + i + u * stride);
  }
  for (; i < n4; i += stride) st<NTS>(ld<NTL>(in + i) + 1.0f, out + i);
}
// chunked: block b owns the contiguous range [b*chunk, (b+1)*chunk); U consecutive 4 KB pieces in flight
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void k_copy_ch(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4) {
  const size_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  size_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NTL>(in + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NTS>(v[u] + 1.0f, out + i + u * 256);
  }
  for (; i < hi; i += 256) st<NTS>(ld<NTL>(in + i) + 1.0f, out + i);
}
// read-only and write-only streams with the same shapes (for the mix model)
template <bool NTL, int U>
__global__ __launch_bounds__(256) void k_read_gs(const v4f* __restrict__ in, float* out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  v4f acc = 0.0f;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NTL>(in + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}
template <bool NTS>
__global__ __launch_bounds__(256) void k_write_gs(v4f* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) st<NTS>(v4f{1.f, 2.f, 3.f, (float)i}, out + i);
}
__global__ void k_fill(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xFFFFFF) * (1.0f / 16777216.0f);
  }
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); f(); (void)hipEventRecord(e0); for (int k = 0; k < 5; ++k) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return ms / 5;
}
struct Row { float tbs; char txt[160]; };
int main() {
  const size_t n = (size_t)3840 * 2160 * 6 * 32;         // 6.4 GB in, 6.4 GB out
  float *in, *out; (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, n * 4);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, n);
  (void)hipMemset(out, 0, n * 4);
  (void)hipDeviceSynchronize();
  std::vector<Row> rows;
  auto rec = [&](const char* what, int blocks, float ms, double bytes) {
    Row r; r.tbs = (float)(bytes / ms / 1e9);
    snprintf(r.txt, sizeof r.txt, "%-58s blocks=%6d : %7.3f ms  %6.2f TB/s", what, blocks, ms, r.tbs);
    rows.push_back(r); printf("%s\n", r.txt); fflush(stdout);
  };
  const size_t n4 = n / 4;
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
#define COPY(K, NTL, NTS, U, name) rec(name, blocks, timeit([&] { hipLaunchKernelGGL((K<NTL, NTS, U>), dim3(blocks), dim3(256), 0, 0, (const v4f*)in, (v4f*)out, n4); }), 2.0 * n * 4)
    COPY(k_copy_gs, false, false, 1, "copy grid-stride plain ld / plain st  U=1");
    COPY(k_copy_gs, false, false, 4, "copy grid-stride plain ld / plain st  U=4");
    COPY(k_copy_gs, false, true, 1, "copy grid-stride plain ld / nt st     U=1");
    COPY(k_copy_gs, false, true, 4, "copy grid-stride plain ld / nt st     U=4");
    COPY(k_copy_gs, false, true, 8, "copy grid-stride plain ld / nt st     U=8");
    COPY(k_copy_gs, true, true, 1, "copy grid-stride nt ld    / nt st     U=1");
    COPY(k_copy_gs, true, true, 4, "copy grid-stride nt ld    / nt st     U=4");
    COPY(k_copy_gs, true, false, 4, "copy grid-stride nt ld    / plain st  U=4");
    COPY(k_copy_ch, false, false, 4, "copy chunked     plain ld / plain st  U=4");
    COPY(k_copy_ch, false, true, 4, "copy chunked     plain ld / nt st     U=4");
    COPY(k_copy_ch, true, true, 4, "copy chunked     nt ld    / nt st     U=4");
    COPY(k_copy_ch, false, true, 8, "copy chunked     plain ld / nt st     U=8");
#undef COPY
    rec("read grid-stride plain U=4", blocks, timeit([&] { hipLaunchKernelGGL((k_read_gs<false, 4>), dim3(blocks), dim3(256), 0, 0, (const v4f*)in, out, n4); }), 1.0 * n * 4);
    rec("read grid-stride nt    U=4", blocks, timeit([&] { hipLaunchKernelGGL((k_read_gs<true, 4>), dim3(blocks), dim3(256), 0, 0, (const v4f*)in, out, n4); }), 1.0 * n * 4);
    rec("write grid-stride plain", blocks, timeit([&] { hipLaunchKernelGGL((k_write_gs<false>), dim3(blocks), dim3(256), 0, 0, (v4f*)out, n4); }), 1.0 * n * 4);
    rec("write grid-stride nt", blocks, timeit([&] { hipLaunchKernelGGL((k_write_gs<true>), dim3(blocks), dim3(256), 0, 0, (v4f*)out, n4); }), 1.0 * n * 4);
  }
  rec("hipMemcpyAsync device-to-device", 0, timeit([&] { (void)hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0); }), 2.0 * n * 4);
  std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.tbs > b.tbs; });
  printf("---- best 12\n");
  for (size_t i = 0; i < rows.size() && i < 12; ++i) printf("%s\n", rows[i].txt);
  return 0;
}
