// K9/K10: heat-map finishing.
//   raw:      1 - met2jod(reconstructed difference)/10            cvvdp_metric.py:744, :398
//   coloured: visualize_diff_map (visualize_diff_map.py:48-106): log-luminance context image
//             (:17-20), histogram tone curve (:23-45, 1024 bins, p^(1/3) cumulative), piecewise-linear
//             colour map (interp.py:22-31,81-89), fp16 output.
// Tone-map statistics are per frame (block of 1 = the reference's CPU behaviour).
#include "kernels.h"
#include <hip/hip_fp16.h>

namespace cvvdp {

__device__ __forceinline__ float heat_value(float q, float jod_a, float jod_exp) {
  float jod;
  if (q <= 0.1f) jod = 10.0f - jod_a * powf(0.1f, jod_exp - 1.0f) * q;   // met2jod, cvvdp_metric.py:646-658
  else jod = 10.0f - jod_a * powf(q, jod_exp);
  return 1.0f - jod / 10.0f;
}

// the 8-bit frame the reference's writers make of the fp16 map: (clip(x, 0, 1) * 255.0).astype(uint8)  (run_cvvdp.py:59-63, :76).
// The array is float16 and stays float16 under numpy's scalar promotion, so the product is ROUNDED TO HALF (step 0.125 in
// [128, 256)) before the truncation: 4 % of all fp16 values in [0,1] land on another code than with an fp32 product.
__device__ __forceinline__ uint8_t half_to_u8(__half v) {
  const float p = fminf(fmaxf(__half2float(v), 0.0f), 1.0f) * 255.0f;   // exact in fp32 (11-bit x 8-bit significands)
  return (uint8_t)__half2float(__float2half_rn(p));
}

// level-0 reconstruction of 4 adjacent pixels of row y (x % 4 == 0): band + expand(level-1 reconstruction); k_expand_add4's expressions
__device__ __forceinline__ void heat_recon4(const HeatArgs& a, int item, int y, int x, float (&q)[4]) {
  const float* c = a.coarse + (int64_t)item * a.Hc * a.Wc;
  const int my = y >> 1, mx = x >> 1;
  const int y0 = max(my - 1, 0), y2 = min(my + 1, a.Hc - 1);
  const float e0 = a.kx[0], e1 = a.kx[1], o = a.kx[2];
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cx = min(max(mx - 1 + k, 0), a.Wc - 1);
    if (y & 1) v[k] = expand_odd(c[(int64_t)my * a.Wc + cx], c[(int64_t)y2 * a.Wc + cx], o);
    else v[k] = expand_even(c[(int64_t)y0 * a.Wc + cx], c[(int64_t)my * a.Wc + cx], c[(int64_t)y2 * a.Wc + cx], e0, e1);
  }
  float4 t = *reinterpret_cast<const float4*>(a.recon + (int64_t)item * a.P + (int64_t)y * a.W + x);
  t.x += expand_even(v[0], v[1], v[2], e0, e1);      // (the reference adds the band to the finished expand: lpyr_dec.py:333)
  t.y += expand_odd(v[1], v[2], o);
  t.z += expand_even(v[1], v[2], v[3], e0, e1);
  t.w += expand_odd(v[2], v[3], o);
  q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
}

// raw map with the last reconstruction step fused in (a.coarse != null): 4 pixels per thread
__global__ __launch_bounds__(256) void k_heat_raw4(HeatArgs a) {
  const int item = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= a.P) return;
  const int y = i / a.W, x = i - y * a.W;
  float q[4];
  heat_recon4(a, item, y, x, q);
  const int64_t base = (int64_t)item * a.P + i;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const __half v = __float2half(heat_value(q[k], a.jod_a, a.jod_exp));
    if (a.out_u8) reinterpret_cast<uint8_t*>(a.out)[base + k] = half_to_u8(v);
    else reinterpret_cast<__half*>(a.out)[base + k] = v;
  }
}

__global__ __launch_bounds__(256) void k_heat_raw(HeatArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)a.items * a.P) return;
  const __half v = __float2half(heat_value(a.recon[i], a.jod_a, a.jod_exp));
  if (a.out_u8) reinterpret_cast<uint8_t*>(a.out)[i] = half_to_u8(v);    // [items][P][1]
  else reinterpret_cast<__half*>(a.out)[i] = v;
}

void launch_heat_raw(const HeatArgs& a, hipStream_t s) {
  if (a.coarse) {
    hipLaunchKernelGGL(k_heat_raw4, dim3((a.P / 4 + 255) / 256, a.items), dim3(256), 0, s, a);
    return;
  }
  const int64_t n = (int64_t)a.items * a.P;
  hipLaunchKernelGGL(k_heat_raw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
}

__global__ void k_heat_init(HeatArgs a) {
  const int item = blockIdx.x;
  uint32_t* st = a.stats + (int64_t)item * kHeatStatsWords;
  for (int i = threadIdx.x; i < kHeatStatsWords; i += blockDim.x) st[i] = (i == 0) ? 0x7F7FFFFFu : 0u;
}

// histogram words only (the range words were filled by the level-0 band kernel: HeatArgs::range_done)
__global__ void k_heat_zero_hist(HeatArgs a) {
  uint32_t* st = a.stats + (int64_t)blockIdx.x * kHeatStatsWords;
  for (int i = 4 + threadIdx.x; i < kHeatStatsWords; i += blockDim.x) st[i] = 0u;
}

// min over positive y and max y of the context image (positive floats order like their bit patterns)
__global__ __launch_bounds__(256) void k_heat_range(HeatArgs a) {
  const int item = blockIdx.y;
  const float* y = a.ctx + (int64_t)item * a.P;
  uint32_t mn = 0x7F7FFFFFu, mx = 0u;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.P; i += gridDim.x * 256) {
    const float v = y[i];
    if (v > 0.0f) { mn = min(mn, __float_as_uint(v)); mx = max(mx, __float_as_uint(v)); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, (uint32_t)__shfl_down((int)mn, off, 64));
    mx = max(mx, (uint32_t)__shfl_down((int)mx, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    uint32_t* st = a.stats + (int64_t)item * kHeatStatsWords;
    atomicMin(&st[0], mn);
    atomicMax(&st[1], mx);
  }
}

__device__ __forceinline__ float log_lum(float y, float clampval) { return logf(fmaxf(y, clampval)); }

__global__ __launch_bounds__(256) void k_heat_hist(HeatArgs a) {
  __shared__ uint32_t s_h[1024];
  const int item = blockIdx.y;
  uint32_t* st = a.stats + (int64_t)item * kHeatStatsWords;
  const float clampval = __uint_as_float(st[0]);
  const float bmin = logf(clampval), bmax = logf(fmaxf(__uint_as_float(st[1]), clampval));
  for (int i = threadIdx.x; i < 1024; i += 256) s_h[i] = 0;
  __syncthreads();
  const float* y = a.ctx + (int64_t)item * a.P;
  const float range = bmax - bmin;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.P; i += gridDim.x * 256) {
    const float b = log_lum(y[i], clampval);
    int pos = (int)((b - bmin) / range * 1024.0f);   // torch.histc bin rule
    pos = min(max(pos, 0), 1023);
    atomicAdd(&s_h[pos], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256)
    if (s_h[i]) atomicAdd(&st[4 + i], s_h[i]);
}

// tone curve v = cumsum(p^(1/3) / sum p^(1/3)) * 0.6 + 0.2   (visualize_diff_map.py:33-43)
__global__ __launch_bounds__(256) void k_heat_curve(HeatArgs a) {
  __shared__ float s_p[1024];
  __shared__ float s_tmp[4];
  const int item = blockIdx.x, t = threadIdx.x;
  const uint32_t* st = a.stats + (int64_t)item * kHeatStatsWords;
  float* cv = a.curve + (int64_t)item * kHeatCurveWords;
  const float clampval = __uint_as_float(st[0]);
  const float bmin = logf(clampval), bmax = logf(fmaxf(__uint_as_float(st[1]), clampval));
  float part = 0.0f;
  for (int i = t; i < 1024; i += 256) {
    const float p = (float)st[4 + i] / (float)a.P;
    const float c = powf(p, 1.0f / 3.0f);
    s_p[i] = c;
    part += c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((t & 63) == 0) s_tmp[t >> 6] = part;
  __syncthreads();
  if (t == 0) {
    const float tot = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
    float run = 0.0f;
    for (int i = 0; i < 1024; ++i) {
      run += s_p[i] / tot;
      cv[i] = run * 0.6f + 0.2f;
    }
    cv[1024] = bmin;
    cv[1025] = bmax;
    cv[1026] = (bmax - bmin < 0.6f) ? 0.0f : 1.0f;
  }
}

__device__ __forceinline__ float scale_node(int i, float bmin, float bmax, float step) {
  // torch.linspace(bmin, bmax, 1024): lower half from the start, upper half from the end
  return (i < 512) ? bmin + step * (float)i : bmax - step * (float)(1023 - i);
}

// met2jod for the per-pixel map (cvvdp_metric.py:646-658, :744) with the hardware log2 / exp2 pair instead of powf: the map leaves
// as fp16 (or 8 bit), 1e-6 relative is far below its last bit; the slope of the linear part is a host constant (a.jod_lin)
__device__ __forceinline__ float heat_value_fast(float q, const HeatArgs& a) {
  const float jod = q <= 0.1f ? 10.0f - a.jod_lin * q : 10.0f - a.jod_a * fast_pow(q, a.jod_exp);
  return 1.0f - jod * 0.1f;
}

// One pixel of visualize_diff_map (visualize_diff_map.py:48-106): tone-mapped context luminance x colour-coded difference.
struct HeatPixelCtx {
  float clampval, bmin, bmax, step, inv_step, inv_lin;
  bool lin;
  const float* cv;
  const float* cin;   // colour-map node positions and colours in LDS: a pixel picks its two nodes by its value (indexing the kernel
  const float* cch;   // arguments per lane made every fetch a global load with 64-bit address arithmetic: 25 loads per 4 pixels)
};
__device__ __forceinline__ void heat_pixel(const HeatArgs& a, const HeatPixelCtx& h, float y, float q, __half (&out)[3]) {
  const float b = fast_log2(fmaxf(y, h.clampval)) * 0.6931471805599453f;
  float tmo;
  if (h.lin) {
    tmo = (b - h.bmin) * h.inv_lin * 0.6f + 0.2f;                          // visualize_diff_map.py:28-31
  } else {
    int hi = (int)ceilf((b - h.bmin) * h.inv_step);
    hi = min(max(hi, 0), 1023);
    // bucketize: smallest node >= b.  The estimate is within one node of it (the nodes are bmin + step * i to 1e-7 of a step): one
    // predicated step down, one up -- as two per-pixel while loops this was half of the kernel's instructions (divergent control flow)
    hi = (hi > 0 && scale_node(max(hi - 1, 0), h.bmin, h.bmax, h.step) >= b) ? hi - 1 : hi;
    hi = (hi < 1023 && scale_node(hi, h.bmin, h.bmax, h.step) < b) ? hi + 1 : hi;
    const int lo = max(hi - 1, 0);
    const float xl = scale_node(lo, h.bmin, h.bmax, h.step), xh = scale_node(hi, h.bmin, h.bmax, h.step);
    float fr = (b - xl) * fast_rcp(xh - xl + 0.000001f);
    if (hi == lo || fr < 0.0f) fr = 0.0f;
    tmo = h.cv[lo] * (1.0f - fr) + h.cv[hi] * fr;
  }
  const float d = fminf(fmaxf(heat_value_fast(q, a), 0.0f), 1.0f);
  int hi = a.n_nodes;
#pragma unroll
  for (int k = 4; k >= 0; --k)
    if (k < a.n_nodes && a.cin[k] >= d) hi = k;   // (wave-uniform node positions: scalar operands of a compare + select)
  hi = min(hi, a.n_nodes - 1);
  const int lo = max(hi - 1, 0);
  const float cl = h.cin[lo], chh = h.cin[hi];
  float fr = (d - cl) * fast_rcp(chh - cl + 0.000001f);
  if (hi == lo || fr < 0.0f) fr = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float col = h.cch[lo * 3 + c] * (1.0f - fr) + h.cch[hi * 3 + c] * fr;
    const float c16 = __half2float(__float2half(col));                     // colour map is stored as fp16 first (:96-98)
    out[c] = __float2half(fminf(fmaxf(c16 * tmo, 0.0f), 1.0f));
  }
}

// VEC: P % 4 == 0 -- a thread owns 4 adjacent pixels (16-byte loads, 8-byte fp16 / 12-byte RGB8 stores); otherwise one pixel.
template <bool VEC>
__global__ __launch_bounds__(256) void k_heat_colour(HeatArgs a) {
  const int item = blockIdx.y;
  constexpr int N = VEC ? 4 : 1;
  const int i = (blockIdx.x * 256 + threadIdx.x) * N;
  // the frame's tone curve (1024 nodes + bmin, bmax, flag) in LDS: every pixel reads two neighbouring nodes picked by its luminance
  __shared__ float s_cv[1028];
  __shared__ float s_cm[20];
  {
    const float* cvg = a.curve + (int64_t)item * kHeatCurveWords;
    for (int k = threadIdx.x; k < 1027; k += 256) s_cv[k] = cvg[k];
    if (threadIdx.x < 5) s_cm[threadIdx.x] = a.cin[threadIdx.x];
    else if (threadIdx.x < 20) s_cm[threadIdx.x] = a.cch[threadIdx.x - 5];
  }
  __syncthreads();
  if (i >= a.P) return;
  const uint32_t* st = a.stats + (int64_t)item * kHeatStatsWords;
  HeatPixelCtx h;
  h.cv = s_cv;
  h.cin = s_cm; h.cch = s_cm + 5;
  h.clampval = __uint_as_float(st[0]);
  h.bmin = h.cv[1024]; h.bmax = h.cv[1025];
  h.lin = h.cv[1026] == 0.0f;
  h.step = (h.bmax - h.bmin) / 1023.0f;
  h.inv_step = 1.0f / h.step;
  h.inv_lin = 1.0f / (h.bmax - h.bmin + 1e-3f);
  const int64_t base = (int64_t)item * a.P + i;
  float y[N], q[N];
  if constexpr (VEC) {
    const float4 yv = *reinterpret_cast<const float4*>(a.ctx + base);
    y[0] = yv.x; y[1] = yv.y; y[2] = yv.z; y[3] = yv.w;
    if (a.coarse) {                      // (uniform) the last reconstruction step here instead of a read-modify-write pass over level 0
      const int row = i / a.W;
      heat_recon4(a, item, row, i - row * a.W, q);
    } else {
      const float4 qv = *reinterpret_cast<const float4*>(a.recon + base);
      q[0] = qv.x; q[1] = qv.y; q[2] = qv.z; q[3] = qv.w;
    }
  } else {
    y[0] = a.ctx[base]; q[0] = a.recon[base];
  }
  __half px[N][3];
#pragma unroll
  for (int k = 0; k < N; ++k) heat_pixel(a, h, y[k], q[k], px[k]);
  if (a.out_u8) {                                                          // interleaved RGB frames
    uint8_t* o = reinterpret_cast<uint8_t*>(a.out) + base * 3;
    if constexpr (VEC) {
      uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 12; ++e) w[e >> 2] |= (uint32_t)half_to_u8(px[e / 3][e % 3]) << (8 * (e & 3));
      uint32_t* o32 = reinterpret_cast<uint32_t*>(o);                      // (item*P + i)*3 bytes: a multiple of 12
      o32[0] = w[0]; o32[1] = w[1]; o32[2] = w[2];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = half_to_u8(px[0][c]);
    }
  } else {
    __half* out = reinterpret_cast<__half*>(a.out);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      __half* o = out + ((int64_t)c * a.items + item) * a.P + i;
      if constexpr (VEC) {
        union { __half hv[4]; uint2 u; } pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk.hv[k] = px[k][c];
        *reinterpret_cast<uint2*>(o) = pk.u;
      } else {
        o[0] = px[0][c];
      }
    }
  }
}

void launch_heat_init(uint32_t* stats, int items, hipStream_t s) {
  HeatArgs a{};
  a.stats = stats; a.items = items;
  hipLaunchKernelGGL(k_heat_init, dim3(items), dim3(256), 0, s, a);
}

void launch_heat_colour(const HeatArgs& a, hipStream_t s) {
  const int gx = min((a.P + 255) / 256, 1024);
  if (!a.range_done) {      // (otherwise the level-0 band kernel took the range of the context plane while it streamed it)
    hipLaunchKernelGGL(k_heat_init, dim3(a.items), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_heat_range, dim3(gx, a.items), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_heat_zero_hist, dim3(a.items), dim3(256), 0, s, a);   // (a second fetch of the same frames counts them again)
  }
  hipLaunchKernelGGL(k_heat_hist, dim3(gx, a.items), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_heat_curve, dim3(a.items), dim3(256), 0, s, a);
  if (a.P % 4 == 0) hipLaunchKernelGGL(k_heat_colour<true>, dim3((a.P / 4 + 255) / 256, a.items), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_heat_colour<false>, dim3((a.P + 255) / 256, a.items), dim3(256), 0, s, a);
}

}  // namespace cvvdp
