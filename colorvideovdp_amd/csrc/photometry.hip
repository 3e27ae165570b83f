// K0: sample unpack -> EOTF / display model -> DKL-d65, written straight into the temporal ring
// (video) or the level-0 planes (image).
// Reference: video_source.py:320-346 (unpack), display_model.py:333-365 (forward), :266-269 (3x3).
// HBM-bound pointwise kernel: one pixel of one (frame,batch,side) per thread, loads coalesced along W.
#include "photometry_dev.h"

namespace cvvdp {

template <int DT, int V>
__global__ __launch_bounds__(256) void k_photometry(PhotoArgs a) {
  __shared__ float s_tab[DT == CVVDP_U8 ? 256 : 1];
  const bool use_lut = stage_eotf_table<DT>(a.dm, s_tab);
  const int pv = blockIdx.x * 256 + threadIdx.x;
  const int P = a.H * a.W;
  const int pix = pv * V;
  if (pix >= P) return;
  const int item = blockIdx.y;  // f * batch + b
  const int side = blockIdx.z;
  const int f = item / a.batch, b = item - f * a.batch;
  const int y = pix / a.W, x = pix - y * a.W;
  const int64_t off0 = b * a.sb[side] + f * a.sf[side] + (int64_t)y * a.sh[side] + (int64_t)x * a.sw[side];
  const void* src = a.src[side];
  const int slot = (a.first_slot + f) % a.n_slots;
  float* dst = a.dst + side * a.d_side + slot * a.d_slot + b * a.d_b + pix;

  float in[3][V];
  if (a.dm.channels == 3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) load_run<DT, V>(src, off0 + c * a.sc[side], in[c]);
  } else {
    load_run<DT, V>(src, off0, in[0]);
#pragma unroll
    for (int i = 0; i < V; ++i) in[1][i] = in[2][i] = in[0][i];
  }
  float out[3][V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float v[3] = {in[0][i], in[1][i], in[2][i]}, o[3];
    if constexpr (DT == CVVDP_F32_DKL) {
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    } else {
      pixel_to_dkl(a.dm, v, o, s_tab, use_lut);
    }
    out[0][i] = o[0]; out[1][i] = o[1]; out[2][i] = o[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(dst + c * a.d_ch) = make_float4(out[c][0], out[c][1], out[c][2], out[c][3]);
    else dst[c * a.d_ch] = out[c][0];
  }
}

template <int DT>
static void launch_dt(const PhotoArgs& a, hipStream_t s) {
  const int P = a.H * a.W;
  // vector path: planar rows with unit W stride, all strides / W / base pointers 4-sample aligned
  const bool vec = can_vectorise(4, a.W, a.sb, a.sc, a.sf, a.sh, a.sw, a.src, dtype_bytes(a.dtype));
  if (vec) {
    dim3 grid((P / 4 + 255) / 256, a.n_frames * a.batch, 2);
    hipLaunchKernelGGL((k_photometry<DT, 4>), grid, dim3(256), 0, s, a);
  } else {
    dim3 grid((P + 255) / 256, a.n_frames * a.batch, 2);
    hipLaunchKernelGGL((k_photometry<DT, 1>), grid, dim3(256), 0, s, a);
  }
}

void launch_photometry(const PhotoArgs& a, hipStream_t s) {
  switch (a.dtype) {
    case CVVDP_U8: launch_dt<CVVDP_U8>(a, s); break;
    case CVVDP_U16: launch_dt<CVVDP_U16>(a, s); break;
    case CVVDP_F16: launch_dt<CVVDP_F16>(a, s); break;
    case CVVDP_F32: launch_dt<CVVDP_F32>(a, s); break;
    default: launch_dt<CVVDP_F32_DKL>(a, s); break;
  }
}

// Pre-filtered sources (cvvdp_metric.py:470-488): channel c of the test frames is level-0 plane 2c, of the reference 2c+1.
__global__ __launch_bounds__(256) void k_put_planes(PutPlanesArgs a) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int P = a.H * a.W;
  if (pix >= P) return;
  const int item = blockIdx.y, plane = blockIdx.z;      // item = f * batch + b
  const int side = plane & 1, ch = plane >> 1;
  const int f = item / a.batch, b = item - f * a.batch;
  const int y = pix / a.W, x = pix - y * a.W;
  a.dst[plane * a.o_plane + (int64_t)item * P + pix] =
      a.src[side][b * a.sb[side] + ch * a.sc[side] + f * a.sf[side] + (int64_t)y * a.sh[side] + (int64_t)x * a.sw[side]];
}

void launch_put_planes(const PutPlanesArgs& a, hipStream_t s) {
  dim3 grid((a.H * a.W + 255) / 256, a.n_frames * a.batch, 8);
  hipLaunchKernelGGL(k_put_planes, grid, dim3(256), 0, s, a);
}

}  // namespace cvvdp
