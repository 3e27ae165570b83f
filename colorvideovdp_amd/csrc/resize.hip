// Planar Y'CbCr frames at the file's resolution -> display-encoded R'G'B' planes at the display's resolution: the
// reference's full_screen_resize (video_source_yuv.py:333-336: YUVReader.get_frame_rgb_tensor, then
// torch.nn.functional.interpolate(mode = bilinear | bicubic | nearest | area), then clip to [0,1]).
//
//   k_yuv_unpack : one thread per source pixel and frame; same per-pixel arithmetic as the temporal kernel's Y'CbCr path
//                  (yuv_pixel_rgb), result written as three fp32 planes [3][n][Hs][Ws]
//   k_resize     : one thread per destination pixel of one plane; the source rows it gathers from are a few KB apart at most,
//                  i.e. L2 hits -- the kernel is bound by its 4 B/pixel store.  Index and weight formulas follow ATen
//                  (UpSample.h: area_pixel_compute_source_index with align_corners = false, cubic A = -0.75, no antialiasing;
//                  AdaptiveAveragePooling start / end index), in fp32 like its float kernels.
#include "temporal_impl.h"

namespace cvvdp {

template <int DT>
__global__ __launch_bounds__(256) void k_yuv_unpack(YuvUnpackArgs a) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.W * a.H) return;
  const int f = blockIdx.y;
  const int y = pix / a.W, x = pix - y * a.W;
  const YuvCtx cx(a.yuv, pix, y, x);
  const int64_t fb = (int64_t)f * a.frame_stride;
  auto ld = [&](int64_t i) -> uint32_t {
    if constexpr (DT == CVVDP_YUV8) return reinterpret_cast<const uint8_t*>(a.src)[i];
    else return reinterpret_cast<const uint16_t*>(a.src)[i];
  };
  RawYuv in;
  in.y = ld(fb + pix);
#pragma unroll
  for (int k = 0; k < 4; ++k) { in.u[k] = ld(fb + a.yuv.u_off + cx.o[k]); in.w[k] = ld(fb + a.yuv.v_off + cx.o[k]); }
  float v[3];
  yuv_pixel_rgb(a.yuv, cx, in, v);
  const int64_t P = (int64_t)a.W * a.H;
#pragma unroll
  for (int c = 0; c < 3; ++c) a.out[((int64_t)c * a.n_frames + f) * P + pix] = v[c];
}

// ATen UpSample.h: cubic_convolution1 / cubic_convolution2 / get_cubic_upsample_coefficients
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_resize(ResizeArgs a) {
  const int ox = blockIdx.x * 256 + threadIdx.x;
  if (ox >= a.Wd) return;
  const int oy = blockIdx.y;
  const float* in = a.in + (int64_t)blockIdx.z * a.Hs * a.Ws;
  float r;
  if constexpr (MODE == CVVDP_RESIZE_NEAREST) {
    const int sx = min((int)floorf((float)ox * a.sx), a.Ws - 1), sy = min((int)floorf((float)oy * a.sy), a.Hs - 1);
    r = in[(int64_t)sy * a.Ws + sx];
  } else if constexpr (MODE == CVVDP_RESIZE_BILINEAR) {
    const float fx = fmaxf(a.sx * ((float)ox + 0.5f) - 0.5f, 0.0f), fy = fmaxf(a.sy * ((float)oy + 0.5f) - 0.5f, 0.0f);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = x0 + (x0 < a.Ws - 1 ? 1 : 0), y1 = y0 + (y0 < a.Hs - 1 ? 1 : 0);
    const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1, ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
    const float* r0 = in + (int64_t)y0 * a.Ws;
    const float* r1 = in + (int64_t)y1 * a.Ws;
    r = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
  } else if constexpr (MODE == CVVDP_RESIZE_BICUBIC) {
    const float fx = a.sx * ((float)ox + 0.5f) - 0.5f, fy = a.sy * ((float)oy + 0.5f) - 0.5f;
    const float flx = floorf(fx), fly = floorf(fy);
    const int ix = (int)flx, iy = (int)fly;
    float wx[4], wy[4];
    cubic_coeffs(fx - flx, wx);
    cubic_coeffs(fy - fly, wy);
    int xs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xs[k] = min(max(ix - 1 + k, 0), a.Ws - 1);
    r = 0.0f;
    float rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* row = in + (int64_t)min(max(iy - 1 + j, 0), a.Hs - 1) * a.Ws;
      rows[j] = row[xs[0]] * wx[0] + row[xs[1]] * wx[1] + row[xs[2]] * wx[2] + row[xs[3]] * wx[3];
    }
    r = rows[0] * wy[0] + rows[1] * wy[1] + rows[2] * wy[2] + rows[3] * wy[3];
  } else {   // area = adaptive average pooling
    const int x0 = (int)(((int64_t)ox * a.Ws) / a.Wd), x1 = (int)((((int64_t)ox + 1) * a.Ws + a.Wd - 1) / a.Wd);
    const int y0 = (int)(((int64_t)oy * a.Hs) / a.Hd), y1 = (int)((((int64_t)oy + 1) * a.Hs + a.Hd - 1) / a.Hd);
    float sum = 0.0f;
    for (int y = y0; y < y1; ++y) {
      const float* row = in + (int64_t)y * a.Ws;
      for (int x = x0; x < x1; ++x) sum += row[x];
    }
    r = sum / (float)((y1 - y0) * (x1 - x0));
  }
  a.out[((int64_t)blockIdx.z * a.Hd + oy) * a.Wd + ox] = clipf(r, 0.0f, 1.0f);
}

void launch_yuv_unpack(const YuvUnpackArgs& a, hipStream_t s) {
  dim3 grid((a.W * a.H + 255) / 256, a.n_frames);
  if (a.bits16) hipLaunchKernelGGL(k_yuv_unpack<CVVDP_YUV16>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_yuv_unpack<CVVDP_YUV8>, grid, dim3(256), 0, s, a);
}

void launch_resize(const ResizeArgs& a, hipStream_t s) {
  dim3 grid((a.Wd + 255) / 256, a.Hd, a.n_planes);
  switch (a.mode) {
    case CVVDP_RESIZE_NEAREST: hipLaunchKernelGGL(k_resize<CVVDP_RESIZE_NEAREST>, grid, dim3(256), 0, s, a); break;
    case CVVDP_RESIZE_BILINEAR: hipLaunchKernelGGL(k_resize<CVVDP_RESIZE_BILINEAR>, grid, dim3(256), 0, s, a); break;
    case CVVDP_RESIZE_BICUBIC: hipLaunchKernelGGL(k_resize<CVVDP_RESIZE_BICUBIC>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(k_resize<CVVDP_RESIZE_AREA>, grid, dim3(256), 0, s, a); break;
  }
}

}  // namespace cvvdp
