// K1: per-channel temporal FIR over the DKL ring (cvvdp_metric.py:554-560).
//   R[2c+side][fi] = sum_k ring[side][ch(c)][window fi+k] * F[c][fl-1-k],   ch(3) = 0 (Y transient)
// The window -> physical slot indirection replaces torch.roll (cvvdp_metric.py:538-539) and the
// replicate/symmetric padding copies (:506-529).
//
// Layout: one thread owns one pixel of one (side, batch) and walks the block's frames in time with a
// register sliding window of the last FL ring values per colour plane, so every ring frame is read
// from HBM once per block (not once per output frame).  The generic-FL fallback re-reads the window.
#include "kernels.h"

namespace cvvdp {

template <int FL>
__global__ __launch_bounds__(256) void k_fir_window(FirArgs a) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.P) return;
  const int b = blockIdx.y, side = blockIdx.z;
  const float* rbase = a.ring + side * a.r_side + b * a.r_b + pix;
  // three colour planes; plane 0 (Y) feeds the sustained and the transient filter
  for (int c = 0; c < 3; ++c) {
    const float* rc = rbase + c * a.r_ch;
    const float* tp = a.taps + c * CVVDP_MAX_FILTER_LEN;
    const float* tt = a.taps + 3 * CVVDP_MAX_FILTER_LEN;
    float w[FL];
#pragma unroll
    for (int k = 0; k < FL - 1; ++k) w[k + 1] = rc[(int64_t)a.slots[k] * a.r_slot];
    float* o_s = a.out + (int64_t)(2 * c + side) * a.o_plane + (int64_t)b * a.P + pix;
    float* o_t = a.out + (int64_t)(6 + side) * a.o_plane + (int64_t)b * a.P + pix;
    for (int fi = 0; fi < a.n_frames; ++fi) {
#pragma unroll
      for (int k = 0; k < FL - 1; ++k) w[k] = w[k + 1];
      w[FL - 1] = rc[(int64_t)a.slots[fi + FL - 1] * a.r_slot];
      float acc = 0.0f, acct = 0.0f;
#pragma unroll
      for (int k = 0; k < FL; ++k) acc += w[k] * tp[k];
      o_s[(int64_t)fi * a.batch * a.P] = acc;
      if (c == 0) {
#pragma unroll
        for (int k = 0; k < FL; ++k) acct += w[k] * tt[k];
        o_t[(int64_t)fi * a.batch * a.P] = acct;
      }
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

void launch_fir(const FirArgs& a, hipStream_t s) {
  dim3 block(256);
  const int gx = (a.P + 255) / 256;
  switch (a.fl) {
    case 7: hipLaunchKernelGGL(k_fir_window<7>, dim3(gx, a.batch, 2), block, 0, s, a); break;    // 24/25 fps
    case 9: hipLaunchKernelGGL(k_fir_window<9>, dim3(gx, a.batch, 2), block, 0, s, a); break;    // 30 fps
    case 15: hipLaunchKernelGGL(k_fir_window<15>, dim3(gx, a.batch, 2), block, 0, s, a); break;  // 50 fps
    case 17: hipLaunchKernelGGL(k_fir_window<17>, dim3(gx, a.batch, 2), block, 0, s, a); break;  // 60 fps
    case 31: hipLaunchKernelGGL(k_fir_window<31>, dim3(gx, a.batch, 2), block, 0, s, a); break;  // 120 fps
    default: hipLaunchKernelGGL(k_fir_generic, dim3(gx, a.n_frames * a.batch, 2), block, 0, s, a); break;
  }
}

}  // namespace cvvdp
