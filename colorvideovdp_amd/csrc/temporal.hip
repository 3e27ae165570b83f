// K1: per-channel temporal FIR over the DKL ring (cvvdp_metric.py:554-560).
//   R[2c+side][fi] = sum_k ring[side][ch(c)][window fi+k] * F[c][fl-1-k],   ch(3) = 0 (Y transient)
// The window -> physical slot indirection replaces torch.roll (cvvdp_metric.py:538-539) and the
// replicate/symmetric padding copies (:506-529).
//
// Layout: one thread owns V consecutive pixels of one (colour plane, side, batch) and walks the block's
// frames in time with a register sliding window of the last FL ring values, so every ring frame is
// read from HBM once per block (not once per output frame).  The next frame's load is issued before
// the current frame's FMAs (software prefetch).  The generic-FL fallback re-reads the window.
#include "kernels.h"

namespace cvvdp {

template <int V> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<4> { using T = float4; };

__device__ __forceinline__ void vfma(float& acc, float a, float b) { acc += a * b; }
__device__ __forceinline__ void vfma(float4& acc, const float4& a, float b) {
  acc.x += a.x * b; acc.y += a.y * b; acc.z += a.z * b; acc.w += a.w * b;
}
__device__ __forceinline__ void vzero(float& a) { a = 0.0f; }
__device__ __forceinline__ void vzero(float4& a) { a = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }

template <int FL, int V>
__global__ __launch_bounds__(256) void k_fir_window(FirArgs a) {
  using T = typename Vec<V>::T;
  const int pv = blockIdx.x * 256 + threadIdx.x;     // index of the V-pixel group
  if (pv * V >= a.P) return;
  const int c = blockIdx.y % 3, b = blockIdx.y / 3, side = blockIdx.z;
  const T* rc = reinterpret_cast<const T*>(a.ring + side * a.r_side + c * a.r_ch + b * a.r_b) + pv;
  const int64_t slot_stride = a.r_slot / V, out_stride = (int64_t)a.batch * a.P / V;
  const float* tp = a.taps + c * CVVDP_MAX_FILTER_LEN;
  const float* tt = a.taps + 3 * CVVDP_MAX_FILTER_LEN;
  T w[FL];
#pragma unroll
  for (int k = 0; k < FL - 1; ++k) w[k + 1] = rc[(int64_t)a.slots[k] * slot_stride];
  T* o_s = reinterpret_cast<T*>(a.out + (int64_t)(2 * c + side) * a.o_plane + (int64_t)b * a.P) + pv;
  T* o_t = reinterpret_cast<T*>(a.out + (int64_t)(6 + side) * a.o_plane + (int64_t)b * a.P) + pv;
  T nxt = rc[(int64_t)a.slots[FL - 1] * slot_stride];
  for (int fi = 0; fi < a.n_frames; ++fi) {
#pragma unroll
    for (int k = 0; k < FL - 1; ++k) w[k] = w[k + 1];
    w[FL - 1] = nxt;
    if (fi + 1 < a.n_frames) nxt = rc[(int64_t)a.slots[fi + FL] * slot_stride];
    T acc;
    vzero(acc);
#pragma unroll
    for (int k = 0; k < FL; ++k) vfma(acc, w[k], tp[k]);
    o_s[(int64_t)fi * out_stride] = acc;
    if (c == 0) {  // block-uniform
      T acct;
      vzero(acct);
#pragma unroll
      for (int k = 0; k < FL; ++k) vfma(acct, w[k], tt[k]);
      o_t[(int64_t)fi * out_stride] = acct;
    }
  }
}

__global__ __launch_bounds__(256) void k_fir_generic(FirArgs a) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.P) return;
  const int item = blockIdx.y, side = blockIdx.z;
  const int fi = item / a.batch, b = item - fi * a.batch;
  const float* rbase = a.ring + side * a.r_side + b * a.r_b + pix;
  for (int c = 0; c < 3; ++c) {
    const float* rc = rbase + c * a.r_ch;
    float acc = 0.0f, acct = 0.0f;
    for (int k = 0; k < a.fl; ++k) {
      const float v = rc[(int64_t)a.slots[fi + k] * a.r_slot];
      acc += v * a.taps[c * CVVDP_MAX_FILTER_LEN + k];
      if (c == 0) acct += v * a.taps[3 * CVVDP_MAX_FILTER_LEN + k];
    }
    a.out[(int64_t)(2 * c + side) * a.o_plane + (int64_t)item * a.P + pix] = acc;
    if (c == 0) a.out[(int64_t)(6 + side) * a.o_plane + (int64_t)item * a.P + pix] = acct;
  }
}

template <int FL>
static void launch_window(const FirArgs& a, hipStream_t s) {
  dim3 block(256);
  if (a.P % 4 == 0) {
    dim3 grid((a.P / 4 + 255) / 256, 3 * a.batch, 2);
    hipLaunchKernelGGL((k_fir_window<FL, 4>), grid, block, 0, s, a);
  } else {
    dim3 grid((a.P + 255) / 256, 3 * a.batch, 2);
    hipLaunchKernelGGL((k_fir_window<FL, 1>), grid, block, 0, s, a);
  }
}

void launch_fir(const FirArgs& a, hipStream_t s) {
  switch (a.fl) {
    case 7: launch_window<7>(a, s); break;    // 24/25 fps... (N = ceil(fps/8)*2+1)
    case 9: launch_window<9>(a, s); break;    // 30 fps
    case 15: launch_window<15>(a, s); break;  // 50 fps
    case 17: launch_window<17>(a, s); break;  // 60 fps
    case 31: launch_window<31>(a, s); break;  // 120 fps
    default: hipLaunchKernelGGL(k_fir_generic, dim3((a.P + 255) / 256, a.n_frames * a.batch, 2), dim3(256), 0, s, a); break;
  }
}

}  // namespace cvvdp
