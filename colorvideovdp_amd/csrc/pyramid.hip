// K2: Gaussian-pyramid reduce (lpyr_dec.py:186-211) and the stand-alone expand+add used by the
// heat-map reconstruction (lpyr_dec.py:223-239, 328-335).
//
// reduce = separable 5-tap [.05 .25 .4 .25 .05], stride 2, vertical pass first, written in the
// reference as a zero-padded convolution plus explicit edge terms.  The edge terms are reproduced
// literally, including the column edge that tests the ROW parity (lpyr_dec.py:206).
//
// Tiling: a 256-thread block produces a 32x32 output tile.  The (2*32+3) x (2*32+3) input patch is
// staged in LDS with coalesced row loads, the vertical pass writes a 32 x 67 intermediate to LDS, the
// horizontal pass reads it.  Every input pixel is fetched from HBM/L2 once per tile (+ the 3-pixel
// apron).
#include "kernels.h"

namespace cvvdp {

constexpr int RT = 32;            // output tile edge
constexpr int RIN = 2 * RT + 3;   // input patch edge (67)

__global__ __launch_bounds__(256) void k_reduce(ReduceArgs a) {
  __shared__ float s_in[RIN][RIN + 1];
  __shared__ float s_v[RT][RIN + 1];
  const int img = blockIdx.z;                       // plane * n_img + item
  const int plane = img / a.n_img, it = img - plane * a.n_img;
  const float* in = a.in + ((int64_t)plane * a.img_cap + it) * a.H * a.W;
  float* out = a.out + ((int64_t)plane * a.img_cap + it) * a.Ho * a.Wo;
  const int oy0 = blockIdx.y * RT, ox0 = blockIdx.x * RT;
  const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;
  const int t = threadIdx.x;
  // stage input patch; outside the image -> 0 (the reference's zero padding)
  for (int e = t; e < RIN * RIN; e += 256) {
    const int r = e / RIN, c = e - r * RIN;
    const int y = iy0 + r, x = ix0 + c;
    s_in[r][c] = (y >= 0 && y < a.H && x >= 0 && x < a.W) ? in[(int64_t)y * a.W + x] : 0.0f;
  }
  __syncthreads();
  const float k0 = a.k[0], k1 = a.k[1], k2 = a.k[2], k3 = a.k[3], k4 = a.k[4];
  // vertical pass: rows oy0..oy0+RT-1 of y_a for the RIN patch columns
  for (int e = t; e < RT * RIN; e += 256) {
    const int r = e / RIN, c = e - r * RIN;
    const int oy = oy0 + r;
    float v = 0.0f;
    if (oy < a.Ho) {
      const int pr = 2 * r;  // patch row of input row 2*oy-2
      v = s_in[pr][c] * k0 + s_in[pr + 1][c] * k1 + s_in[pr + 2][c] * k2 + s_in[pr + 3][c] * k3 + s_in[pr + 4][c] * k4;
      // edge terms (lpyr_dec.py:195-199); patch row of input row y is y - iy0
      if (oy == 0) v += s_in[0 - iy0][c] * k1 + s_in[1 - iy0][c] * k0;
      if (oy == a.Ho - 1) {
        if (a.H & 1) v += s_in[a.H - 1 - iy0][c] * k3 + s_in[a.H - 2 - iy0][c] * k4;
        else v += s_in[a.H - 1 - iy0][c] * k4;
      }
    }
    s_v[r][c] = v;
  }
  __syncthreads();
  // horizontal pass
  for (int e = t; e < RT * RT; e += 256) {
    const int r = e / RT, c = e - r * RT;
    const int oy = oy0 + r, ox = ox0 + c;
    if (oy >= a.Ho || ox >= a.Wo) continue;
    const int pc = 2 * c;
    float v = s_v[r][pc] * k0 + s_v[r][pc + 1] * k1 + s_v[r][pc + 2] * k2 + s_v[r][pc + 3] * k3 + s_v[r][pc + 4] * k4;
    if (ox == 0) v += s_v[r][0 - ix0] * k1 + s_v[r][1 - ix0] * k0;          // lpyr_dec.py:205
    if (ox == a.Wo - 1) {
      if (a.H & 1) v += s_v[r][a.W - 1 - ix0] * k3 + s_v[r][a.W - 2 - ix0] * k4;  // sic: row parity, :206-207
      else v += s_v[r][a.W - 1 - ix0] * k4;                                        // :209
    }
    out[(int64_t)oy * a.Wo + ox] = v;
  }
}

void launch_reduce(const ReduceArgs& a, hipStream_t s) {
  dim3 grid((a.Wo + RT - 1) / RT, (a.Ho + RT - 1) / RT, a.n_planes * a.n_img);
  hipLaunchKernelGGL(k_reduce, grid, dim3(256), 0, s, a);
}

// fine += expand(coarse).  One thread per fine pixel; coarse reads are L1/L2 hits.
__global__ __launch_bounds__(256) void k_expand_add(ExpandAddArgs a) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.H * a.W) return;
  const int img = blockIdx.y;
  const float* c = a.coarse + (int64_t)img * a.Hc * a.Wc;
  float* f = a.fine + (int64_t)img * a.H * a.W;
  const int y = pix / a.W, x = pix - y * a.W;
  const int my = y >> 1, mx = x >> 1;
  const int y0 = max(my - 1, 0), y2 = min(my + 1, a.Hc - 1);
  const int x0 = max(mx - 1, 0), x2 = min(mx + 1, a.Wc - 1);
  const float e0 = a.kx[0], e1 = a.kx[1], o = a.kx[2];
  // vertical first (lpyr_dec.py:229-232), then horizontal (:234-237)
  auto vert = [&](int cx) -> float {
    if (y & 1) return c[(int64_t)my * a.Wc + cx] * o + c[(int64_t)y2 * a.Wc + cx] * o;
    return c[(int64_t)y0 * a.Wc + cx] * e0 + c[(int64_t)my * a.Wc + cx] * e1 + c[(int64_t)y2 * a.Wc + cx] * e0;
  };
  float v;
  if (x & 1) v = vert(mx) * o + vert(x2) * o;
  else v = vert(x0) * e0 + vert(mx) * e1 + vert(x2) * e0;
  f[pix] += v;
}

void launch_expand_add(const ExpandAddArgs& a, hipStream_t s) {
  dim3 grid((a.H * a.W + 255) / 256, a.n_img);
  hipLaunchKernelGGL(k_expand_add, grid, dim3(256), 0, s, a);
}

}  // namespace cvvdp
