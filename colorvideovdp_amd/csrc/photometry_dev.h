// Device-side display model shared by the image photometry kernel (photometry.hip) and the fused
// photometry + temporal FIR kernel (temporal.hip).
// Reference: video_source.py:320-346 (unpack), display_model.py:333-365 (forward), :266-269 (3x3).
#pragma once
#include "kernels.h"
#include <hip/hip_fp16.h>

namespace cvvdp {

// Constant divisions are multiplications by the rounded reciprocal (an IEEE fp32 divide is a ~10
// instruction sequence on gfx950 and these run once per sample); the result differs from x/255 by at
// most 1 ulp, far inside the parity tolerance.
constexpr float kInv255 = 1.0f / 255.0f;
constexpr float kInv65535 = 1.0f / 65535.0f;

template <int DT>
__device__ __forceinline__ float load_sample(const void* base, int64_t off) {
  if constexpr (DT == CVVDP_U8) {
    return (float)reinterpret_cast<const uint8_t*>(base)[off] * kInv255;
  } else if constexpr (DT == CVVDP_U16) {
    return (float)reinterpret_cast<const uint16_t*>(base)[off] * kInv65535;
  } else if constexpr (DT == CVVDP_F16) {
    return __half2float(reinterpret_cast<const __half*>(base)[off]);
  } else {
    return reinterpret_cast<const float*>(base)[off];
  }
}

__device__ __forceinline__ float srgb2lin(float p) {
  // display_model.py:78-80
  // x^2.4 = x^2 * x^0.4: the SFU part has |0.4*log2 x| <= 1.4, so the result stays within ~2 ulp
  const float x = (p + 0.055f) * (1.0f / 1.055f);
  return p > 0.04045f ? x * x * fast_pow(x, 0.4f) : p * (1.0f / 12.92f);
}

// a product that is rounded on its own (hipcc contracts a*b +- c into an FMA by default)
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
  const float p = a * b;
  return p;
}

__device__ __forceinline__ float pq2lin(float v) {
  // display_model.py:58-70
  const float n = 0.15930175781250000f, m = 78.843750000000000f;
  const float c1 = 0.83593750000000000f, c2 = 18.851562500000000f, c3 = 18.687500000000000f;
  const float t = fast_pow(v, 1.0f / m);
  // c2 - c3*t cancels 18.85 - 18.5.. to ~0.2-0.3 and the quotient is raised to the power 6.28: the rounding of the product is
  // visible in the result (2e-5).  torch rounds the product, then subtracts (two operations); hipcc would contract this into one FMA.
  // With the product rounded on its own the mean distance to torch's CPU result drops from 1.05e-5 to 1.9e-6 (tools/ubench/pq_accuracy.*).
  const float ct = mul_rounded(c3, t);
  return 10000.0f * fast_pow(fmaxf(t - c1, 0.0f) * fast_rcp(c2 - ct), 1.0f / n);
}

__device__ __forceinline__ float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// 8-bit sources: the table of DisplayArgs::lut staged in LDS (256 floats; a per-lane gather from kernel arguments would be a
// global load queued behind the prefetched frames).  Returns whether the table is in use (kernel-uniform).
template <int DT>
__device__ __forceinline__ bool stage_eotf_table(const DisplayArgs& dm, float* s_tab) {
  if constexpr (DT == CVVDP_U8) {
    if (dm.use_lut) {
      for (int i = threadIdx.x; i < 256; i += blockDim.x) s_tab[i] = dm.lut[i];
      __syncthreads();
      return true;
    }
  }
  return false;
}

// display model + colour transform of one pixel; v = display-encoded RGB (or 1 channel replicated); lut: see stage_eotf_table
__device__ __forceinline__ void pixel_to_dkl(const DisplayArgs& a, float (&v)[3], float (&o)[3], const float* lut = nullptr, bool use_lut = false) {
  float L[3];
  const int e = a.eotf;
  if (use_lut) {                 // v = code * (1/255) within an ulp: the code is recovered exactly
#pragma unroll
    for (int c = 0; c < 3; ++c) L[c] = lut[(int)(v[c] * 255.0f + 0.5f)];
  } else {
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
    for (int c = 0; c < 3; ++c) s[c] = v[c] <= 0.5f ? v[c] * v[c] * (1.0f / 3.0f) : (fast_exp2((v[c] - hc) * (1.4426950408889634f / ha)) + hb) * (1.0f / 12.0f);
    const float Ys = 0.2627f * s[0] + 0.6780f * s[1] + 0.0593f * s[2];
    const float gain = fast_pow(Ys, a.gamma - 1.0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float lin = gain * s[c];
      if (a.exposure != 1.0f) lin = clipf(lin * a.exposure, 0.0f, 1.0f);
      L[c] = a.scale * lin + a.Y_black + a.Y_refl;
    }
  } else {  // gamma
#pragma unroll
    for (int c = 0; c < 3; ++c) L[c] = a.scale * clipf(fast_pow(v[c], a.gamma) * a.exposure, 0.0f, 1.0f) + a.Y_black + a.Y_refl;
  }
  }
  if (a.channels == 3) {
#pragma unroll
    // torch.sum(RGB * row, dim=channel) (display_model.py:268-269): three rounded products, summed left to right.  The opponent rows
    // cancel for near-grey pixels, so whether a product is rounded (torch) or kept exact inside an FMA (hipcc's default contraction)
    // shows in RG / YV: same operations as the reference here
    for (int c = 0; c < 3; ++c) o[c] = (mul_rounded(L[0], a.m[3 * c]) + mul_rounded(L[1], a.m[3 * c + 1])) + mul_rounded(L[2], a.m[3 * c + 2]);
  } else {  // luminance-only content fills all three planes (cvvdp_metric.py:503 broadcast)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = L[0];
  }
}

// V consecutive pixels of a row as floats.  V > 1 requires unit W stride and V-sample alignment.
template <int DT, int V>
__device__ __forceinline__ void load_run(const void* base, int64_t off, float (&out)[V]) {
  if constexpr (V == 1) {
    out[0] = load_sample<DT>(base, off);
  } else if constexpr (DT == CVVDP_U8) {
    if constexpr (V == 2) {
      const uchar2 q = *reinterpret_cast<const uchar2*>(reinterpret_cast<const uint8_t*>(base) + off);
      out[0] = (float)q.x * kInv255; out[1] = (float)q.y * kInv255;
    } else {
      const uchar4 q = *reinterpret_cast<const uchar4*>(reinterpret_cast<const uint8_t*>(base) + off);
      out[0] = (float)q.x * kInv255; out[1] = (float)q.y * kInv255; out[2] = (float)q.z * kInv255; out[3] = (float)q.w * kInv255;
    }
  } else if constexpr (DT == CVVDP_U16) {
    if constexpr (V == 2) {
      const ushort2 q = *reinterpret_cast<const ushort2*>(reinterpret_cast<const uint16_t*>(base) + off);
      out[0] = (float)q.x * kInv65535; out[1] = (float)q.y * kInv65535;
    } else {
      const ushort4 q = *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(base) + off);
      out[0] = (float)q.x * kInv65535; out[1] = (float)q.y * kInv65535; out[2] = (float)q.z * kInv65535; out[3] = (float)q.w * kInv65535;
    }
  } else if constexpr (DT == CVVDP_F16) {
    const __half2* p = reinterpret_cast<const __half2*>(reinterpret_cast<const __half*>(base) + off);
    const float2 lo = __half22float2(p[0]);
    out[0] = lo.x; out[1] = lo.y;
    if constexpr (V == 4) {
      const float2 hi = __half22float2(p[1]);
      out[2] = hi.x; out[3] = hi.y;
    }
  } else {
    if constexpr (V == 2) {
      const float2 q = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(base) + off);
      out[0] = q.x; out[1] = q.y;
    } else {
      const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
      out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
    }
  }
}

// can V-wide row loads be used for this source layout?
inline bool can_vectorise(int V, int W, const int64_t* sb, const int64_t* sc, const int64_t* sf, const int64_t* sh, const int64_t* sw,
                          const void* const* src, int elem_bytes) {
  if (W % V) return false;
  for (int k = 0; k < 2; ++k) {
    if (sw[k] != 1 || sh[k] % V || sc[k] % V || sf[k] % V || sb[k] % V) return false;
    if (reinterpret_cast<uintptr_t>(src[k]) % (size_t)(V * elem_bytes)) return false;
  }
  return true;
}

inline int dtype_bytes(int dt) { return (dt == CVVDP_U8 || dt == CVVDP_YUV8) ? 1 : (dt == CVVDP_U16 || dt == CVVDP_F16 || dt == CVVDP_YUV16) ? 2 : 4; }

}  // namespace cvvdp
