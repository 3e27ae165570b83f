// K9/K10: heat-map finishing.
//   raw:      1 - met2jod(reconstructed difference)/10            cvvdp_metric.py:744, :398
//   coloured: visualize_diff_map (visualize_diff_map.py:48-106): log-luminance context image
//             (:17-20), histogram tone curve (:23-45, 1024 bins, p^(1/3) cumulative), piecewise-linear
//             colour map (interp.py:22-31,81-89), fp16 output.
// Tone-map statistics are per frame (block of 1 = the reference's CPU behaviour).
#include "kernels.h"
#include <hip/hip_fp16.h>
#include <type_traits>

// NO IMPLICIT MULTIPLY-ADDS IN THIS FILE (round 6b).  hipcc's default (-ffp-contract=fast) lets every a*b+c of the source become a
// fused multiply-add, per use and per kernel as the optimiser sees fit: the two colour kernels below, written from the same
// expressions, disagreed in the last fp16 bit of one value in 30 000 -- k_heat_colour_rows had formed log2(y)*ln2 - node as one
// fma, k_heat_colour had not (found with the debug planes of HEAT_DEBUG_STAGE, tools/runs/r06b_heat_dbg2.py).  The reference's
// visualize_diff_map / interp1 / met2jod are separately rounded tensor operations (visualize_diff_map.py:17-106, interp.py:55-60,
// cvvdp_metric.py:646-658); what IS a fused multiply-add here says so (__builtin_fmaf: the expand's taps, the tone curve's nodes).
#pragma clang fp contract(off)

namespace cvvdp {

// One value of the raw map: 1 - met2jod(q)/10 (cvvdp_metric.py:646-658, :744), with the hardware log2 / exp2 pair instead of powf (the map
// leaves as fp16 or 8 bit: 1e-6 relative is far below its last bit; until round 6b the raw kernels called powf twice per pixel -- once for
// the CONSTANT 0.1^(jod_exp-1) -- and divided by 10: 260 VALU instructions per pixel); the slope of the linear part is a host constant
// (HeatArgs::jod_lin); both branches evaluated and selected (a divergent branch around the power costs more than the power: the empty asm
// makes both values exist before the select)
__device__ __forceinline__ float heat_raw_value(float q, float jod_lin, float jod_a, float jod_exp) {
  float j_lin = jod_lin * q, j_pow = jod_a * fast_pow(q, jod_exp);
  asm("" : "+v"(j_lin), "+v"(j_pow));
  const float jod = 10.0f - (q <= 0.1f ? j_lin : j_pow);
  return 1.0f - jod * 0.1f;
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
    const __half v = __float2half(heat_raw_value(q[k], a.jod_lin, a.jod_a, a.jod_exp));
    if (a.out_u8) reinterpret_cast<uint8_t*>(a.out)[base + k] = half_to_u8(v);
    else reinterpret_cast<__half*>(a.out)[base + k] = v;
  }
}

__global__ __launch_bounds__(256) void k_heat_raw(HeatArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)a.items * a.P) return;
  const __half v = __float2half(heat_raw_value(a.recon[i], a.jod_lin, a.jod_a, a.jod_exp));
  if (a.out_u8) reinterpret_cast<uint8_t*>(a.out)[i] = half_to_u8(v);    // [items][P][1]
  else reinterpret_cast<__half*>(a.out)[i] = v;
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

// Round 6b: the frames this runs on are smooth (a wave's 64 neighbouring pixels fall into a handful of bins), so with ONE copy of the bins in
// LDS the atomics of a wave serialise on a few addresses: 0.72 ms per 16 frames of 8K = 2.9 TB/s for a kernel that only reads 4 B/pixel.
// Eight copies interleaved by bin (copy = lane & 7: the lanes that hit one bin spread over eight banks), 16-byte loads where the frame
// allows them (VEC: P % 4 == 0).  The counts are integers: any order gives the same histogram.
constexpr int kHistCopies = 8;
template <bool VEC>
__global__ __launch_bounds__(256) void k_heat_hist(HeatArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_h[1024 * kHistCopies];
  const int item = blockIdx.y;
  uint32_t* st = a.stats + (int64_t)item * kHeatStatsWords;
  const float clampval = __uint_as_float(st[0]);
  const float bmin = logf(clampval), bmax = logf(fmaxf(__uint_as_float(st[1]), clampval));
  for (int i = threadIdx.x; i < 1024 * kHistCopies; i += 256) s_h[i] = 0;
  __syncthreads();
  const float* y = a.ctx + (int64_t)item * a.P;
  const float range = bmax - bmin;
  const int cp = threadIdx.x & (kHistCopies - 1);
  auto count = [&](float v) {
    const float b = log_lum(v, clampval);
    int pos = (int)((b - bmin) / range * 1024.0f);   // torch.histc bin rule
    pos = min(max(pos, 0), 1023);
    atomicAdd(&s_h[pos * kHistCopies + cp], 1u);
  };
  if constexpr (VEC) {
    const int n4 = a.P >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      const float4 v = reinterpret_cast<const float4*>(y)[i];
      count(v.x); count(v.y); count(v.z); count(v.w);
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.P; i += gridDim.x * 256) count(y[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const uint4 c0 = *reinterpret_cast<const uint4*>(&s_h[i * kHistCopies]), c1 = *reinterpret_cast<const uint4*>(&s_h[i * kHistCopies + 4]);
    const uint32_t n = (c0.x + c0.y + c0.z + c0.w) + (c1.x + c1.y + c1.z + c1.w);
    if (n) atomicAdd(&st[4 + i], n);
  }
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
  // p^(1/3) / sum: element-wise, by every thread; the running sum stays one thread's in-order chain (1024 dependent additions, as the
  // reference's cumsum) -- with the 1024 IEEE divisions inside that chain the kernel took 55 us per launch (round 6b)
  const float tot = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
  for (int i = t; i < 1024; i += 256) s_p[i] = s_p[i] / tot;
  __syncthreads();
  if (t == 0) {
    float run = 0.0f;
    for (int i = 0; i < 1024; ++i) {
      run += s_p[i];
      s_p[i] = run;
    }
  }
  __syncthreads();
  for (int i = t; i < 1024; i += 256) cv[i] = s_p[i] * 0.6f + 0.2f;
  if (t == 0) {
    cv[1024] = bmin;
    cv[1025] = bmax;
    cv[1026] = (bmax - bmin < 0.6f) ? 0.0f : 1.0f;
  }
}

__device__ __forceinline__ float scale_node(int i, float bmin, float bmax, float step) {
  // torch.linspace(bmin, bmax, 1024): lower half from the start, upper half from the end (one fused multiply-add each, as every
  // build of this file has formed them)
  return (i < 512) ? __builtin_fmaf(step, (float)i, bmin) : __builtin_fmaf(-step, (float)(1023 - i), bmax);
}

// Debugging aid (tools/build_variant.sh <name> heatmap.hip -DHEAT_DEBUG_STAGE=n; tools/runs/r06b_heat_dbg2.py reads it back): both colour
// kernels put the fp32 bits of ONE intermediate of every pixel into planes 0 (low half) and 1 (high half) of the fp16 output instead of
// the colours.  n = 1 tone-mapped luminance, 2 map value d, 3 colour-map fraction, 4 reconstructed q, 5 log luminance b, 6 tone-curve
// fraction, 7 b - node.  Product builds compile none of it.
#ifdef HEAT_DEBUG_STAGE
#define HEAT_DBG_DECL float dbg_frt = 0.0f, dbg_num = 0.0f;
#define HEAT_DBG_TONE(fr_, num_) dbg_frt = (fr_); dbg_num = (num_);
#define HEAT_DBG_OUT(out_) { \
    const float dbg = HEAT_DEBUG_STAGE == 1 ? tmo : HEAT_DEBUG_STAGE == 2 ? d : HEAT_DEBUG_STAGE == 3 ? fr : HEAT_DEBUG_STAGE == 4 ? q : \
                      HEAT_DEBUG_STAGE == 5 ? b : HEAT_DEBUG_STAGE == 6 ? dbg_frt : dbg_num; \
    const uint32_t u = __float_as_uint(dbg); \
    out_[0] = __ushort_as_half((unsigned short)(u & 0xffffu)); out_[1] = __ushort_as_half((unsigned short)(u >> 16)); out_[2] = __ushort_as_half((unsigned short)0); }
#else
#define HEAT_DBG_DECL
#define HEAT_DBG_TONE(fr_, num_)
#define HEAT_DBG_OUT(out_)
#endif

// The per-pixel expressions both colour kernels share (separately rounded products and sums: see the top of the file)
__device__ __forceinline__ float lerp_plain(float a, float b, float fr) { return a * (1.0f - fr) + b * fr; }            // interp.py:55-60
__device__ __forceinline__ float tone_linear(float b, float bmin, float inv_lin) { return (b - bmin) * inv_lin * 0.6f + 0.2f; }   // visualize_diff_map.py:28-31
// the map value the colour kernels code: heat_raw_value clamped to [0, 1] (visualize_diff_map.py:76-90)
__device__ __forceinline__ float heat_unit_value(float q, float jod_lin, float jod_a, float jod_exp) {
  return fminf(fmaxf(heat_raw_value(q, jod_lin, jod_a, jod_exp), 0.0f), 1.0f);
}
// colour x tone-mapped luminance -> half (the reference converts a float tensor, visualize_diff_map.py:96-106)
__device__ __forceinline__ __half heat_out_half(float c16, float tmo) { return __float2half(fminf(fmaxf(c16 * tmo, 0.0f), 1.0f)); }

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
  HEAT_DBG_DECL
  if (h.lin) {
    tmo = tone_linear(b, h.bmin, h.inv_lin);
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
    HEAT_DBG_TONE(fr, b - xl)
    tmo = lerp_plain(h.cv[lo], h.cv[hi], fr);
  }
  const float d = heat_unit_value(q, a.jod_lin, a.jod_a, a.jod_exp);
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
    const float col = lerp_plain(h.cch[lo * 3 + c], h.cch[hi * 3 + c], fr);
    const float c16 = __half2float(__float2half(col));                     // colour map is stored as fp16 first (:96-98)
    out[c] = heat_out_half(c16, tmo);
  }
  HEAT_DBG_OUT(out)
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

// Round 6b: k_heat_colour for the frames that fuse the last reconstruction step (a.coarse != null: W % 4 == 0) -- every large frame.
// k_heat_colour<true> spends 709 VALU + 446 SALU instructions on a thread's 4 pixels: every 256-thread block fetches the frame's tone curve
// (4 KB) for 1024 pixels and derives the curve's constants with three IEEE divisions, a thread divides its pixel index by the row length,
// fetches a 3 x 4 coarse patch with twelve clamped 64-bit addresses, evaluates the tone curve's node positions four times per pixel
// (select + multiply-add each) and gathers six colour-map words one by one.  Here a block owns a tile of ROWS rows x up to 1024 columns:
//   * the tone curve goes to LDS once per tile, as {node position, curve value} pairs: a pixel reads its two candidate nodes, then the pair
//     it interpolates between -- no per-pixel node arithmetic (the node values are scale_node's, computed once per tile);
//   * the thread walks DOWN its four columns: the coarse patch is a rolling window of three coarse rows in registers (four loads every
//     second fine row instead of ten per row), row parity is a compile-time constant of the unrolled row pair, no division anywhere;
//   * the colour map sits in LDS as one {r, g, b, position} word per node.
// Same per-pixel operations on the same values as heat_pixel / heat_recon4 (tests/test_gpu_parity.py::
// test_heat_colour_kernels_agree_bit_for_bit holds the two kernels' outputs against each other: HeatArgs::pixel_layout).
constexpr int kHeatTileRows = 16;
// a map value known to lie in [0, 1] (and not NaN) -> the writers' 8-bit code: half_to_u8 without the clamp, the product formed in half
// precision directly (the fp32 product of an 11-bit and an 8-bit significand is exact, so rounding it to half = the half multiply)
__device__ __forceinline__ uint32_t unit_half_to_u8(__half v) { return (uint32_t)__half2ushort_rz(__hmul(v, __float2half(255.0f))); }

template <int NN>      // colour-map nodes: 3 (supra-threshold) or 5 (threshold)
__global__ __launch_bounds__(256) void k_heat_colour_rows(HeatArgs a, int n_chunk, int chunk_cols) {
  __shared__ __attribute__((aligned(16))) float2 s_nc[1026];     // {node, curve}; one pad pair behind node 1023 (read, multiplied by 0)
  __shared__ __attribute__((aligned(16))) float4 s_cm[6];        // colour nodes {r, g, b, position}; one pad node
  __shared__ __attribute__((aligned(16))) float s_node[1024];    // the node positions alone (neighbours in one read)
  const int item = blockIdx.y;
  const int chunk = (int)blockIdx.x % n_chunk, rg = (int)blockIdx.x / n_chunk;
  const float* cvg = a.curve + (int64_t)item * kHeatCurveWords;
  const uint32_t* st = a.stats + (int64_t)item * kHeatStatsWords;
  const float clampval = __uint_as_float(st[0]);
  const float bmin = cvg[1024], bmax = cvg[1025];
  const bool lin = cvg[1026] == 0.0f;
  const float step = (bmax - bmin) / 1023.0f;
  const float inv_step = 1.0f / step;
  const float inv_lin = 1.0f / (bmax - bmin + 1e-3f);
  for (int k = threadIdx.x; k < 1026; k += 256) {
    const int kk = min(k, 1023);
    const float node = scale_node(kk, bmin, bmax, step);
    s_nc[k] = make_float2(node, cvg[kk]);
    if (k < 1024) s_node[k] = node;
  }
  if (threadIdx.x < 6) {
    const int k = min((int)threadIdx.x, NN - 1);
    s_cm[threadIdx.x] = make_float4(a.cch[k * 3], a.cch[k * 3 + 1], a.cch[k * 3 + 2], a.cin[k]);
  }
  __syncthreads();
  const int x = chunk * chunk_cols + 4 * (int)threadIdx.x;
  if (4 * (int)threadIdx.x >= chunk_cols || x >= a.W) return;
  const int W = a.W, Wc = a.Wc, Hc = a.Hc;
  const int y_begin = rg * kHeatTileRows, y_end = min(a.H, y_begin + kHeatTileRows);
  const float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
  const float* coarse = a.coarse + (int64_t)item * Hc * Wc;
  const int mx = x >> 1;
  int cx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cx[k] = min(max(mx - 1 + k, 0), Wc - 1);
  auto load_row = [&](int r, float (&d)[4]) {
    const float* p = coarse + (int64_t)r * Wc;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = p[cx[k]];
  };
  float cA[4], cB[4], cC[4];                                   // coarse rows max(my-1, 0), my, min(my+1, Hc-1) of the current fine row
  {
    const int my = y_begin >> 1;
    load_row(max(my - 1, 0), cA); load_row(my, cB); load_row(min(my + 1, Hc - 1), cC);
  }
  // everything a thread touches of its frame lies within 2^32 bytes of the frame's first sample: uniform bases + 32-bit offsets
  const char* rec_item = reinterpret_cast<const char*>(a.recon + (int64_t)item * a.P);
  const char* ctx_item = reinterpret_cast<const char*>(a.ctx + (int64_t)item * a.P);
  char* out8_item = reinterpret_cast<char*>(a.out) + (int64_t)item * a.P * 3;
  char* out16_item = reinterpret_cast<char*>(a.out) + (int64_t)item * a.P * 2;
  const int64_t plane16 = (int64_t)a.items * a.P * 2;
  float cin[NN];
#pragma unroll
  for (int k = 0; k < NN; ++k) cin[k] = a.cin[k];
  const float jod_lin = a.jod_lin, jod_a = a.jod_a, jod_exp = a.jod_exp;

  auto pixel = [&](float y, float q, __half (&out)[3]) {
    const float b = fast_log2(fmaxf(y, clampval)) * 0.6931471805599453f;
    // both tone curves evaluated, one selected (the flag is per frame; straight-line code lets the four pixels' LDS reads overlap)
    const float tmo_lin = tone_linear(b, bmin, inv_lin);
    float tmo;
    HEAT_DBG_DECL
    {
      // bucketize: smallest node >= b.  heat_pixel: estimate, one predicated step down, one up (after a step down the node is >= b: no
      // step up).  The estimate is taken >= 1 here so that the two candidate nodes are always the neighbours hi-1, hi (one LDS read);
      // heat_pixel's estimate 0 means b <= node 0, which the step down arrives at from 1 as well.
      int hi = (int)ceilf((b - bmin) * inv_step);
      hi = min(max(hi, 1), 1023);
      const float nm1 = s_node[hi - 1], n0 = s_node[hi];
      const bool down = nm1 >= b;
      const bool up = (!down) & (hi < 1023) & (n0 < b);
      hi = hi - (int)down + (int)up;
      const int lo = max(hi - 1, 0);
      const float2 pl = s_nc[lo], ph = s_nc[lo + 1];                     // (hi == lo only for hi == 0: the weight of ph is 0 then)
      float fr = (b - pl.x) * fast_rcp(ph.x - pl.x + 0.000001f);
      if (hi == lo || fr < 0.0f) fr = 0.0f;
      HEAT_DBG_TONE(fr, b - pl.x)
      tmo = lerp_plain(pl.y, ph.y, fr);
    }
    tmo = lin ? tmo_lin : tmo;
    const float d = heat_unit_value(q, jod_lin, jod_a, jod_exp);
    int hi = 0;
#pragma unroll
    for (int k = 0; k < NN; ++k) hi += cin[k] < d ? 1 : 0;     // = the smallest node >= d (sorted positions), NN if none
    hi = min(hi, NN - 1);
    const int lo = max(hi - 1, 0);
    const float4 nl = s_cm[lo], nh = s_cm[lo + 1];
    float fr = (d - nl.w) * fast_rcp(nh.w - nl.w + 0.000001f);
    if (hi == lo || fr < 0.0f) fr = 0.0f;
    const float cl[3] = {nl.x, nl.y, nl.z}, ch[3] = {nh.x, nh.y, nh.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float col = lerp_plain(cl[c], ch[c], fr);
      const float c16 = __half2float(__float2half(col));                   // colour map is stored as fp16 first (:96-98)
      out[c] = heat_out_half(c16, tmo);
    }
    HEAT_DBG_OUT(out)
  };

  auto row = [&](int y, auto odd_) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (decltype(odd_)::value) v[k] = expand_odd(cB[k], cC[k], eo);
      else v[k] = expand_even(cA[k], cB[k], cC[k], e0, e1);
    }
    const uint32_t pix = (uint32_t)(y * W + x);
    float4 t = *reinterpret_cast<const float4*>(rec_item + pix * 4u);
    const float4 yv = *reinterpret_cast<const float4*>(ctx_item + pix * 4u);
    t.x += expand_even(v[0], v[1], v[2], e0, e1);      // (the reference adds the band to the finished expand: lpyr_dec.py:333)
    t.y += expand_odd(v[1], v[2], eo);
    t.z += expand_even(v[1], v[2], v[3], e0, e1);
    t.w += expand_odd(v[2], v[3], eo);
    const float q[4] = {t.x, t.y, t.z, t.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
    __half px[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) pixel(yy[k], q[k], px[k]);
    if (a.out_u8) {
      uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 12; ++e) w[e >> 2] |= unit_half_to_u8(px[e / 3][e % 3]) << (8 * (e & 3));
      uint32_t* o32 = reinterpret_cast<uint32_t*>(out8_item + pix * 3u);
      o32[0] = w[0]; o32[1] = w[1]; o32[2] = w[2];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        union { __half hv[4]; uint2 u; } pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk.hv[k] = px[k][c];
        *reinterpret_cast<uint2*>(out16_item + c * plane16 + pix * 2u) = pk.u;
      }
    }
  };

  for (int y = y_begin; y < y_end; y += 2) {                   // (y_begin is even)
    row(y, std::false_type{});
    if (y + 1 < y_end) row(y + 1, std::true_type{});
    // the next row pair's coarse window: coarse row my+1 becomes my
    const int my = (y >> 1) + 1;
    if (y + 2 < y_end) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { cA[k] = cB[k]; cB[k] = cC[k]; }
      load_row(min(my + 1, Hc - 1), cC);
    }
  }
}

// The raw map on k_heat_colour_rows' walk (a.coarse != null): 16-row tiles, the coarse patch as a rolling window, one 8-byte (fp16) or
// 4-byte (8 bit) store per thread and row.  k_heat_raw4 keeps the frames of the A/B switch; same values (the same expressions on the same samples).
__global__ __launch_bounds__(256) void k_heat_raw_rows(HeatArgs a, int n_chunk, int chunk_cols) {
  const int item = blockIdx.y;
  const int chunk = (int)blockIdx.x % n_chunk, rg = (int)blockIdx.x / n_chunk;
  const int x = chunk * chunk_cols + 4 * (int)threadIdx.x;
  if (4 * (int)threadIdx.x >= chunk_cols || x >= a.W) return;
  const int W = a.W, Wc = a.Wc, Hc = a.Hc;
  const int y_begin = rg * kHeatTileRows, y_end = min(a.H, y_begin + kHeatTileRows);
  const float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
  const float jod_lin = a.jod_lin, jod_a = a.jod_a, jod_exp = a.jod_exp;
  const float* coarse = a.coarse + (int64_t)item * Hc * Wc;
  const char* rec_item = reinterpret_cast<const char*>(a.recon + (int64_t)item * a.P);
  char* out8_item = reinterpret_cast<char*>(a.out) + (int64_t)item * a.P;
  char* out16_item = reinterpret_cast<char*>(a.out) + (int64_t)item * a.P * 2;
  const int mx = x >> 1;
  int cx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cx[k] = min(max(mx - 1 + k, 0), Wc - 1);
  auto load_row = [&](int r, float (&d)[4]) {
    const float* p = coarse + (int64_t)r * Wc;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = p[cx[k]];
  };
  float cA[4], cB[4], cC[4];
  {
    const int my = y_begin >> 1;
    load_row(max(my - 1, 0), cA); load_row(my, cB); load_row(min(my + 1, Hc - 1), cC);
  }
  auto row = [&](int y, auto odd_) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (decltype(odd_)::value) v[k] = expand_odd(cB[k], cC[k], eo);
      else v[k] = expand_even(cA[k], cB[k], cC[k], e0, e1);
    }
    const uint32_t pix = (uint32_t)(y * W + x);
    float4 t = *reinterpret_cast<const float4*>(rec_item + pix * 4u);
    t.x += expand_even(v[0], v[1], v[2], e0, e1);      // (lpyr_dec.py:333)
    t.y += expand_odd(v[1], v[2], eo);
    t.z += expand_even(v[1], v[2], v[3], e0, e1);
    t.w += expand_odd(v[2], v[3], eo);
    const float q[4] = {t.x, t.y, t.z, t.w};
    union { __half hv[4]; uint2 u; } pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) pk.hv[k] = __float2half(heat_raw_value(q[k], jod_lin, jod_a, jod_exp));
    if (a.out_u8) {
      uint32_t w = 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) w |= (uint32_t)half_to_u8(pk.hv[k]) << (8 * k);
      *reinterpret_cast<uint32_t*>(out8_item + pix) = w;
    } else {
      *reinterpret_cast<uint2*>(out16_item + pix * 2u) = pk.u;
    }
  };
  for (int y = y_begin; y < y_end; y += 2) {
    row(y, std::false_type{});
    if (y + 1 < y_end) row(y + 1, std::true_type{});
    if (y + 2 < y_end) {
      const int my = (y >> 1) + 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) { cA[k] = cB[k]; cB[k] = cC[k]; }
      load_row(min(my + 1, Hc - 1), cC);
    }
  }
}

void launch_heat_raw(const HeatArgs& a, hipStream_t s) {
  if (a.coarse && !a.pixel_layout) {           // (W % 4 == 0: core.cpp heat_l0_fused)
    const int n_chunk = (a.W + 1023) / 1024;
    const int chunk_cols = ((a.W / 4 + n_chunk - 1) / n_chunk) * 4;
    const int n_rg = (a.H + kHeatTileRows - 1) / kHeatTileRows;
    hipLaunchKernelGGL(k_heat_raw_rows, dim3(n_chunk * n_rg, a.items), dim3(256), 0, s, a, n_chunk, chunk_cols);
    return;
  }
  if (a.coarse) {
    hipLaunchKernelGGL(k_heat_raw4, dim3((a.P / 4 + 255) / 256, a.items), dim3(256), 0, s, a);
    return;
  }
  const int64_t n = (int64_t)a.items * a.P;
  hipLaunchKernelGGL(k_heat_raw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
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
  {
    // a block zeroes and sums 32 KB of bins: at least 128 pixels per thread, and enough blocks for 5 per CU
    const int by_work = max(1, a.P / (256 * 128)), fill = (256 * 5 * 2 + a.items - 1) / a.items;
    const int gh = max(1, min(min(by_work, 1024), max(fill, 64)));
    if (a.P % 4 == 0) hipLaunchKernelGGL(k_heat_hist<true>, dim3(gh, a.items), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_heat_hist<false>, dim3(gh, a.items), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL(k_heat_curve, dim3(a.items), dim3(256), 0, s, a);
  if (a.coarse && !a.pixel_layout) {           // (W % 4 == 0: core.cpp heat_l0_fused)
    const int n_chunk = (a.W + 1023) / 1024;
    const int chunk_cols = ((a.W / 4 + n_chunk - 1) / n_chunk) * 4;
    const int n_rg = (a.H + kHeatTileRows - 1) / kHeatTileRows;
    if (a.n_nodes == 3) hipLaunchKernelGGL(k_heat_colour_rows<3>, dim3(n_chunk * n_rg, a.items), dim3(256), 0, s, a, n_chunk, chunk_cols);
    else hipLaunchKernelGGL(k_heat_colour_rows<5>, dim3(n_chunk * n_rg, a.items), dim3(256), 0, s, a, n_chunk, chunk_cols);
  } else if (a.P % 4 == 0) {
    hipLaunchKernelGGL(k_heat_colour<true>, dim3((a.P / 4 + 255) / 256, a.items), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_heat_colour<false>, dim3((a.P + 255) / 256, a.items), dim3(256), 0, s, a);
  }
}

}  // namespace cvvdp
