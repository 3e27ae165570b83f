// K8: do_pooling_and_jods + met2jod (cvvdp_metric.py:610-658).  Q_per_ch[B,C,F,bands] -> JOD[B].
// Tiny; one 256-thread block per batch item, threads stride over frames.
#include "kernels.h"

namespace cvvdp {

__device__ __forceinline__ float spow(float x, float p) { return powf(x + kEps, p) - powf(kEps, p); }

__global__ __launch_bounds__(256) void k_pool(PoolArgs a) {
  __shared__ float s_tmp[4];
  const int b = blockIdx.x, t = threadIdx.x;
  float part = 0.0f, single = 0.0f;
  for (int f = t; f < a.F; f += 256) {
    float tc = 0.0f;
    for (int c = 0; c < a.C; ++c) {
      const float* q = a.q + (((int64_t)b * a.C + c) * a.F + f) * a.L;
      float sc = 0.0f;
      for (int l = 0; l < a.L; ++l) {
        const float wb = (l == a.L - 1) ? a.bb_w[c] : 1.0f;
        sc += spow(q[l] * a.ch_w[c] * wb, a.beta_sch);
      }
      const float Qsc = spow(sc, 1.0f / a.beta_sch);   // over bands, not normalised (:625)
      tc += spow(Qsc, a.beta_tch);
    }
    const float Qtc = spow(tc, 1.0f / a.beta_tch);     // over channels (:633)
    single = Qtc;
    part += spow(Qtc, a.beta_t);
  }
  // block sum over frames
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((t & 63) == 0) s_tmp[t >> 6] = part;
  __syncthreads();
  if (t == 0) {
    float Q;
    if (a.F == 1) {
      Q = single * a.image_int;                          // :636
    } else {
      const float s = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
      Q = spow(s / (float)a.F, 1.0f / a.beta_t);         // over frames, normalised (:638)
    }
    const float Qt = 0.1f;                               // met2jod, :646-658
    float jod;
    if (Q <= Qt) jod = 10.0f - a.jod_a * powf(Qt, a.jod_exp - 1.0f) * Q;
    else jod = 10.0f - a.jod_a * powf(Q, a.jod_exp);
    a.jod[b] = jod;
  }
}

void launch_pool(const PoolArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_pool, dim3(a.B), dim3(256), 0, s, a); }

// Features for the ML heads: cvvdp_feature_pooling (cvvdp_ml_metric.py:77-107).  One block per (cell, item, channel):
// mean and E[x^2] - mean^2 of |T'|, |R'| and D over a feature_size x feature_size cell (AvgPool2d with ceil_mode: the last
// cells of a row / column average over the pixels that exist).  Sums are taken in double in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_feature_pool(FeatPoolArgs a) {
  __shared__ double s_sum[6][4];
  const int cx = blockIdx.x, cy = blockIdx.y;
  const int item = blockIdx.z / a.nch, c = blockIdx.z - item * a.nch;
  const int x0 = cx * a.fs, y0 = cy * a.fs;
  const int w = min(a.fs, a.W - x0), h = min(a.fs, a.H - y0);
  const int64_t P = (int64_t)a.H * a.W, ps = (int64_t)a.items_cap * P;
  const float* T = a.tr + (int64_t)c * ps + (int64_t)item * P;
  const float* R = T + 4 * ps;
  const float* D = a.d + (int64_t)c * ps + (int64_t)item * P;
  const float ig = a.inv_gain[c];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int e = threadIdx.x; e < w * h; e += 256) {
    const int yy = e / w, xx = e - yy * w;
    const int64_t o = (int64_t)(y0 + yy) * a.W + x0 + xx;
    const float t = T[o] * ig, r = R[o] * ig, d = D[o];
    acc[0] += t; acc[1] += (double)t * t; acc[2] += r; acc[3] += (double)r * r; acc[4] += d; acc[5] += (double)d * d;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n = (double)w * h;
    float* out = a.out + ((((int64_t)item * a.Hc + cy) * a.Wc + cx) * a.nch + c) * 6;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float mean = (float)((s_sum[2 * q][0] + s_sum[2 * q][1] + s_sum[2 * q][2] + s_sum[2 * q][3]) / n);
      const float msq = (float)((s_sum[2 * q + 1][0] + s_sum[2 * q + 1][1] + s_sum[2 * q + 1][2] + s_sum[2 * q + 1][3]) / n);
      out[2 * q] = mean;
      out[2 * q + 1] = msq - mean * mean;         // fp32, like the reference's avg_pool(x**2) - mean**2
    }
  }
}

// The same statistics from the column sums k_band4 keeps in features mode (band4.hip, FEATURES): one wave per (cell, item,
// channel); lane <-> column of the cell, pieces of the cell row (one per row segment that overlaps it) added in ascending
// order, then the columns across the wave -- all in double, in a fixed order (deterministic).
__global__ __launch_bounds__(64) void k_feature_finish(FeatFinishArgs a) {
  const int cx = blockIdx.x, cy = blockIdx.y;
  const int item = blockIdx.z / a.nch, c = blockIdx.z - item * a.nch;
  const int x0 = cx * a.fs, y0 = cy * a.fs;
  const int w = min(a.fs, a.W - x0), h = min(a.fs, a.H - y0);
  const int seg_lo = y0 / a.seg_h, seg_hi = (y0 + h - 1) / a.seg_h;
  const float* base = a.fsum + ((int64_t)item * a.nch + c) * a.f_pieces * 6 * a.W;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int xx = threadIdx.x; xx < w; xx += 64) {
    for (int sg = seg_lo; sg <= seg_hi; ++sg) {
      const float* p = base + (int64_t)(cy + sg) * 6 * a.W + x0 + xx;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += (double)p[(int64_t)k * a.W];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off, 64);
  }
  if (threadIdx.x == 0) {
    const double n = (double)w * h, ig = (double)a.inv_gain[c];
    const double scale[6] = {ig, ig * ig, ig, ig * ig, 1.0, 1.0};     // |T'|, |R'| carry the channel gain, the reference's features do not
    float* out = a.out + ((((int64_t)item * a.Hc + cy) * a.Wc + cx) * a.nch + c) * 6;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float mean = (float)(acc[2 * q] * scale[2 * q] / n);
      const float msq = (float)(acc[2 * q + 1] * scale[2 * q + 1] / n);
      out[2 * q] = mean;
      out[2 * q + 1] = msq - mean * mean;         // fp32, like the reference's avg_pool(x**2) - mean**2
    }
  }
}

void launch_feature_finish(const FeatFinishArgs& a, hipStream_t s) {
  dim3 grid(a.Wc, a.Hc, a.items * a.nch);
  hipLaunchKernelGGL(k_feature_finish, grid, dim3(64), 0, s, a);
}

void launch_feature_pool(const FeatPoolArgs& a, hipStream_t s) {
  dim3 grid(a.Wc, a.Hc, a.items * a.nch);
  hipLaunchKernelGGL(k_feature_pool, grid, dim3(256), 0, s, a);
}

}  // namespace cvvdp
