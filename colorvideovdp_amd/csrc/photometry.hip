// K0: sample unpack -> EOTF / display model -> DKL-d65, written straight into the temporal ring
// (video) or the level-0 planes (image).
// Reference: video_source.py:320-346 (unpack), display_model.py:333-365 (forward), :266-269 (3x3).
// HBM-bound pointwise kernel: one pixel of one (frame,batch,side) per thread, loads coalesced along W.
#include "kernels.h"
#include <hip/hip_fp16.h>

namespace cvvdp {

template <int DT>
__device__ __forceinline__ float load_sample(const void* base, int64_t off) {
  if constexpr (DT == CVVDP_U8) {
    return (float)reinterpret_cast<const uint8_t*>(base)[off] / 255.0f;
  } else if constexpr (DT == CVVDP_U16) {
    return (float)reinterpret_cast<const uint16_t*>(base)[off] / 65535.0f;
  } else if constexpr (DT == CVVDP_F16) {
    return __half2float(reinterpret_cast<const __half*>(base)[off]);
  } else {
    return reinterpret_cast<const float*>(base)[off];
  }
}

__device__ __forceinline__ float srgb2lin(float p) {
  // display_model.py:78-80
  return p > 0.04045f ? powf((p + 0.055f) / 1.055f, 2.4f) : p / 12.92f;
}

__device__ __forceinline__ float pq2lin(float v) {
  // display_model.py:58-70
  const float n = 0.15930175781250000f, m = 78.843750000000000f;
  const float c1 = 0.83593750000000000f, c2 = 18.851562500000000f, c3 = 18.687500000000000f;
  float t = powf(v, 1.0f / m);
  return 10000.0f * powf(fmaxf(t - c1, 0.0f) / (c2 - c3 * t), 1.0f / n);
}

__device__ __forceinline__ float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

template <int DT>
__global__ __launch_bounds__(256) void k_photometry(PhotoArgs a) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int P = a.H * a.W;
  if (pix >= P) return;
  const int item = blockIdx.y;  // f * batch + b
  const int side = blockIdx.z;
  const int f = item / a.batch, b = item - f * a.batch;
  const int y = pix / a.W, x = pix - y * a.W;
  const int64_t off0 = b * a.sb[side] + f * a.sf[side] + (int64_t)y * a.sh[side] + (int64_t)x * a.sw[side];
  const void* src = a.src[side];
  const int slot = (a.first_slot + f) % a.n_slots;
  float* dst = a.dst + side * a.d_side + slot * a.d_slot + b * a.d_b + pix;

  float v[3];
  if (a.channels == 3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = load_sample<DT>(src, off0 + c * a.sc[side]);
  } else {
    v[0] = v[1] = v[2] = load_sample<DT>(src, off0);
  }
  if constexpr (DT == CVVDP_F32_DKL) {
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c * a.d_ch] = v[c];
    return;
  }
  float L[3];
  const int e = a.eotf;
  if (e != CVVDP_EOTF_LINEAR) {  // display_model.py:335-337 (clamp is a no-op when nothing is out of range)
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = clipf(v[c], 0.0f, 1.0f);
  }
  if (e == CVVDP_EOTF_SRGB) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float lin = srgb2lin(v[c]);
      if (a.exposure != 1.0f) lin = clipf(lin * a.exposure, 0.0f, 1.0f);
      L[c] = a.scale * lin + a.Y_black + a.Y_refl;
    }
  } else if (e == CVVDP_EOTF_PQ) {
#pragma unroll
    for (int c = 0; c < 3; ++c) L[c] = clipf(pq2lin(v[c]) * a.exposure, 0.005f, a.Y_peak) + a.Y_black + a.Y_refl;
  } else if (e == CVVDP_EOTF_LINEAR) {
#pragma unroll
    for (int c = 0; c < 3; ++c) L[c] = clipf(v[c] * a.exposure, a.lin_lo, a.Y_peak) + a.Y_refl;
  } else if (e == CVVDP_EOTF_HLG) {
    // display_model.py:89-111
    const float ha = 0.17883277f, hb = 1.0f - 4.0f * 0.17883277f;
    const float hc = a.hlg_c;  // 0.5 - a*ln(4a), evaluated in double on the host like the reference
    float s[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) s[c] = v[c] <= 0.5f ? v[c] * v[c] / 3.0f : (expf((v[c] - hc) / ha) + hb) / 12.0f;
    const float Ys = 0.2627f * s[0] + 0.6780f * s[1] + 0.0593f * s[2];
    const float gain = powf(Ys, a.gamma - 1.0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float lin = gain * s[c];
      if (a.exposure != 1.0f) lin = clipf(lin * a.exposure, 0.0f, 1.0f);
      L[c] = a.scale * lin + a.Y_black + a.Y_refl;
    }
  } else {  // gamma
#pragma unroll
    for (int c = 0; c < 3; ++c) L[c] = a.scale * clipf(powf(v[c], a.gamma) * a.exposure, 0.0f, 1.0f) + a.Y_black + a.Y_refl;
  }
  if (a.channels == 3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c * a.d_ch] = L[0] * a.m[3 * c] + L[1] * a.m[3 * c + 1] + L[2] * a.m[3 * c + 2];
  } else {  // luminance-only content fills all three planes (cvvdp_metric.py:503 broadcast)
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c * a.d_ch] = L[0];
  }
}

void launch_photometry(const PhotoArgs& a, hipStream_t s) {
  const int P = a.H * a.W;
  dim3 grid((P + 255) / 256, a.n_frames * a.batch, 2);
  switch (a.dtype) {
    case CVVDP_U8: hipLaunchKernelGGL(k_photometry<CVVDP_U8>, grid, dim3(256), 0, s, a); break;
    case CVVDP_U16: hipLaunchKernelGGL(k_photometry<CVVDP_U16>, grid, dim3(256), 0, s, a); break;
    case CVVDP_F16: hipLaunchKernelGGL(k_photometry<CVVDP_F16>, grid, dim3(256), 0, s, a); break;
    case CVVDP_F32: hipLaunchKernelGGL(k_photometry<CVVDP_F32>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(k_photometry<CVVDP_F32_DKL>, grid, dim3(256), 0, s, a); break;
  }
}

}  // namespace cvvdp
