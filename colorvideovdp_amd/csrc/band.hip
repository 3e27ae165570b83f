// K3..K7 fused, one launch per pyramid level:
//   expand(g[l+1]) -> Laplacian -> Weber contrast + background luminance   lpyr_dec.py:386-408
//   castleCSF sensitivity (LUT over log10 L_bkg, 10**)                      csf.py:49, interp.py:55-60,92-100
//   T_p, R_p, mutual masking with 13x13 sigma=3 phase-uncertainty blur      cvvdp_metric.py:835-856,963-971
//   cross-channel mask pooling, transducer, soft clamp                      cvvdp_metric.py:753-760,945-950
//   spatial p-norm partial sums (beta = 2)                                  cvvdp_metric.py:722,1032-1048
// Nothing but the two Gaussian levels is read from HBM and nothing but a few partial sums is written
// (plus the optional heat-map band / debug dump).
//
// Geometry ("strip march"): a 256-thread block owns a vertical strip of 256-2R columns (R = blur
// radius 6) and marches down a segment of rows.  Thread t <-> image column x0-R+t.  Per row:
//   1. vertical half of the expand for the strip's coarse columns -> LDS            (sync)
//   2. every thread: horizontal half of the expand, contrast, CSF, T_p/R_p for its column;
//      min(|T_p|,|R_p|) -> LDS row, |T_p-R_p| -> LDS ring of R+1 rows                (sync)
//   3. interior threads: 13-tap horizontal blur from the LDS row into a private 13-row register
//      window, 13-tap vertical blur over that window -> masking, transducer, clamp, accumulate.
// Out-of-image taps of the blur use reflect padding (torchvision GaussianBlur): the halo is evaluated
// at the reflected coordinate, so no separate padding pass exists.
#include "kernels.h"

namespace cvvdp {

constexpr int BT = 256;          // threads per block
constexpr int BR = 6;            // blur radius
constexpr int BW = 2 * BR + 1;   // 13 taps
constexpr int VE_W = BT / 2 + 4; // coarse columns a strip can touch

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

__device__ __forceinline__ float safe_powf(float x, float p, float eps_p) { return powf(x + kEps, p) - eps_p; }

template <int NCH, bool BLUR>
__global__ __launch_bounds__(BT) void k_band(BandArgs a) {
  constexpr int R = BLUR ? BR : 0;
  constexpr int SW = BT - 2 * R;
  constexpr int NP = 2 * NCH;
  __shared__ float s_ve[NP][VE_W];
  __shared__ float s_m[BLUR ? NCH : 1][BT];
  __shared__ float s_d[BLUR ? (R + 1) : 1][BLUR ? NCH : 1][BT];
  __shared__ float s_red[4][BT / 64];
  __shared__ float s_lut[4 * CVVDP_CSF_NODES];  // log2-domain CSF rows: lut*log2(10) + log2(sens_mul)

  const int t = threadIdx.x;
  const int strip = blockIdx.x, seg = blockIdx.y, item = blockIdx.z;
  const int H = a.H, W = a.W, Hc = a.Hc, Wc = a.Wc;
  const int x0 = strip * SW;
  const int xcol = x0 - R + t;
  const bool col_ok = xcol < W + R;                 // xcol >= -R always
  const int xx = col_ok ? reflect_idx(xcol, W) : 0;
  const bool interior = (t >= R) && (t < BT - R) && (xcol < W);
  const int ys = seg * a.seg_h;
  const int ye = min(H, ys + a.seg_h);

  const int64_t P = (int64_t)H * W, Pc = (int64_t)Hc * Wc;
  const float* g = a.g + (int64_t)item * P;
  const float* gc = a.gc + (int64_t)item * Pc;
  const int64_t gps = (int64_t)a.items_cap * P, gcps = (int64_t)a.items_cap_c * Pc;

  // coarse column range this strip can touch (after reflection every xx lies in [xlo, xhi))
  const int xlo = max(x0 - R, 0), xhi = min(x0 - R + BT, W);
  const int cx_lo = max((xlo >> 1) - 1, 0);
  const int cx_hi = min(((xhi - 1) >> 1) + 1, Wc - 1);
  const int n_cx = cx_hi - cx_lo + 1;

  const float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
  const float ind_scale = (float)(CVVDP_CSF_NODES - 1) / (a.logL_last - a.logL_first);
  if (t < 4 * CVVDP_CSF_NODES) s_lut[t] = a.lut[t] * kLog2_10 + fast_log2(a.sens_mul);

  // 13-row vertical-blur window per channel as a register vector: the newest row goes to slot (row mod 13)
  // through an M0-relative register write; the scalar weights are rotated instead of the data (cf. band4.hip)
  typedef float v16f __attribute__((ext_vector_type(16)));
  v16f win[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float wr[BW];
#pragma unroll
  for (int k = 0; k < BW; ++k) wr[k] = a.blur[(k + BW - 1) % BW];
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};

  for (int r = ys - R; r < ye + R; ++r) {
    const int rr = reflect_idx(r, H);
    // ---- 1. vertical half of the expand (lpyr_dec.py:229-232) for the strip's coarse columns
    {
      const int my = rr >> 1;
      const int ya = max(my - 1, 0), yb = min(my + 1, Hc - 1);
      const bool odd = rr & 1;
      // threads 0..127 take planes 0,2,4,.. and threads 128..255 planes 1,3,5,..; coarse column = t & 127
      for (int ci = t & 127; ci < n_cx; ci += 128) {
        for (int p = t >> 7; p < NP; p += 2) {
          const float* cp = gc + p * gcps + (cx_lo + ci);
          float v;
          if (odd) v = expand_odd(cp[(int64_t)my * Wc], cp[(int64_t)yb * Wc], eo);
          else v = expand_even(cp[(int64_t)ya * Wc], cp[(int64_t)my * Wc], cp[(int64_t)yb * Wc], e0, e1);
          s_ve[p][ci] = v;
        }
      }
    }
    __syncthreads();
    // ---- 2. per-column contrast, CSF, T_p / R_p
    float m[4], d[4];
    if (col_ok) {
      const int mx = xx >> 1;
      const int ca = max(mx - 1, 0) - cx_lo, cb = mx - cx_lo, cc = min(mx + 1, Wc - 1) - cx_lo;
      float ex[NP], gv[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        gv[p] = g[p * gps + (int64_t)rr * W + xx];
        if (xx & 1) ex[p] = expand_odd(s_ve[p][cb], s_ve[p][cc], eo);               // lpyr_dec.py:234-237
        else ex[p] = expand_even(s_ve[p][ca], s_ve[p][cb], s_ve[p][cc], e0, e1);
      }
      const float Lt = fmaxf(ex[0], 0.01f), Lr = fmaxf(ex[1], 0.01f);     // lpyr_dec.py:394
      const float rLt = fast_rcp(Lt), rLr = fast_rcp(Lr);
      const float logL = fast_log2(Lr) * kLog10_2;                        // lpyr_dec.py:408, query = reference plane
      float ind = (logL - a.logL_first) * ind_scale;                       // interp.py:93
      ind = fminf(fmaxf(ind, 0.0f), (float)(CVVDP_CSF_NODES - 1));
      const int i0 = (int)ind;
      const float fr = ind - (float)i0;
      const int i1 = min(i0 + 1, CVVDP_CSF_NODES - 1);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float l0 = s_lut[c * CVVDP_CSF_NODES + i0], l1 = s_lut[c * CVVDP_CSF_NODES + i1];
        const float S = fast_exp2(l0 + (l1 - l0) * fr) * a.ch_gain[c];      // csf.py:49, cvvdp_metric.py:709,:836
        const float ct = fminf((gv[2 * c] - ex[2 * c]) * rLt, 1000.0f) * a.band_mul;       // lpyr_dec.py:402, :66
        const float cr = fminf((gv[2 * c + 1] - ex[2 * c + 1]) * rLr, 1000.0f) * a.band_mul;
        const float Tp = ct * S, Rp = cr * S;
        m[c] = fminf(fabsf(Tp), fabsf(Rp));                                 // :845
        d[c] = fabsf(Tp - Rp);
        if (a.fdump && interior && r >= ys && r < ye) {   // features (cvvdp_ml_metric.py:355): every image pixel once (not the halo columns / rows)
          const int64_t o = (int64_t)c * a.items_cap * P + (int64_t)item * P + (int64_t)r * W + xcol;
          a.fdump[o] = fabsf(Tp);
          a.fdump[(int64_t)4 * a.items_cap * P + o] = fabsf(Rp);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) m[c] = d[c] = 0.0f;
    }
    float v[4];
    int yc = r;  // row whose masking is finished in this iteration
    bool have = true;
    if constexpr (BLUR) {
      const int dslot = ((r % (R + 1)) + (R + 1)) % (R + 1);
#pragma unroll
      for (int c = 0; c < NCH; ++c) { s_m[c][t] = m[c]; s_d[dslot][c][t] = d[c]; }
      __syncthreads();
      // ---- 3. separable blur: horizontal from LDS, vertical from the register window
      yc = r - R;
      have = yc >= ys;
      if (interior) {
        float h[4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          float s = 0.0f;
#pragma unroll
          for (int k = 0; k < BW; ++k) s += a.blur[k] * s_m[c][t - R + k];
          h[c] = s;
        }
        const int slot = (r - (ys - R)) % BW;     // block-uniform
#pragma unroll
        for (int c = 0; c < NCH; ++c) win[c][slot] = h[c];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          float acc = 0.0f;
#pragma unroll
          for (int sdx = 0; sdx < BW; ++sdx) acc += wr[sdx] * win[c][sdx];
          v[c] = acc;
        }
        if (have) {
          const int cslot = ((yc % (R + 1)) + (R + 1)) % (R + 1);
#pragma unroll
          for (int c = 0; c < NCH; ++c) d[c] = s_d[cslot][c][t];
        }
      }
      {   // rotate the blur weights for the next row (scalar ALU)
        const float last = wr[BW - 1];
#pragma unroll
        for (int k = BW - 1; k > 0; --k) wr[k] = wr[k - 1];
        wr[0] = last;
      }
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[c] = m[c];
    }
    // ---- masking, transducer, clamp, pooling (cvvdp_metric.py:845-856, 722)
    if (interior && have) {
      float Mq[4];
#pragma unroll
      for (int c = 0; c < NCH; ++c) Mq[c] = fast_pow(fabsf(v[c] * a.mask_c10) + kEps, a.q[c]) - a.eps_q[c];
      float D[4];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float M = 0.0f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) M += Mq[k] * a.xw[k * 4 + c];       // cvvdp_metric.py:758-760
        const float Du = (fast_pow(d[c] + kEps, a.mask_p) - a.eps_p) * fast_rcp(1.0f + M);
        D[c] = a.dmax * Du * fast_rcp(a.dmax + Du);                       // soft clamp, :949-950
        const float de = D[c] + kEps;
        acc[c] += de * de - kEps * kEps;                                  // safe_pow(D, beta=2)
      }
      const int64_t o = (int64_t)yc * W + xcol;
      if (a.ddump) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) a.ddump[(int64_t)c * a.items_cap * P + (int64_t)item * P + o] = D[c];
      }
      if (a.dchr) {  // cvvdp_metric.py:728-734, lpyr_dec.py:308-314
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) s += fast_pow(D[c] * a.hw[c] + kEps, a.beta_tch) - a.eps_btch;
        a.dchr[(int64_t)item * P + o] = (fast_pow(s + kEps, 1.0f / a.beta_tch) - a.eps_inv_btch) / a.band_mul;
      }
    }
    // the next iteration's first __syncthreads orders the reuse of s_m / s_d; s_ve needs its own
    // barrier only when step 3 (and its barrier) is compiled out
    if constexpr (!BLUR) __syncthreads();
  }
  // ---- block reduction of the partial p-norm sums: wave shuffles, then one LDS hop
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float s = acc[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((t & 63) == 0) s_red[c][t >> 6] = s;
  }
  __syncthreads();
  if (t < 4) {
    const float s = s_red[t][0] + s_red[t][1] + s_red[t][2] + s_red[t][3];
    const int nblk = a.n_strip * a.n_seg;
    a.partial[((int64_t)item * nblk + (seg * a.n_strip + strip)) * 4 + t] = s;
  }
}

void launch_band(const BandArgs& a, bool blur, hipStream_t s) {
  dim3 grid(a.n_strip, a.n_seg, a.items);
  if (a.nch == 4) {
    if (blur) hipLaunchKernelGGL((k_band<4, true>), grid, dim3(BT), 0, s, a);
    else hipLaunchKernelGGL((k_band<4, false>), grid, dim3(BT), 0, s, a);
  } else {
    if (blur) hipLaunchKernelGGL((k_band<3, true>), grid, dim3(BT), 0, s, a);
    else hipLaunchKernelGGL((k_band<3, false>), grid, dim3(BT), 0, s, a);
  }
}

// ---------------------------------------------------------------- baseband + finalize
__device__ __forceinline__ float block_sum(float v, float* s_tmp) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int t = threadIdx.x;
  __syncthreads();
  if ((t & 63) == 0) s_tmp[t >> 6] = v;
  __syncthreads();
  return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

// Baseband (lpyr_dec.py:378-384, cvvdp_metric.py:711-712): L_bkg is the per-frame spatial mean of the
// clamped Y planes, D = |T-R|*S without masking.  One block per item; the band is at most a few
// hundred pixels.
__global__ __launch_bounds__(256) void k_baseband(BaseArgs a) {
  __shared__ float s_tmp[4];
  const int item = blockIdx.x, t = threadIdx.x;
  const int P = a.H * a.W;
  const int64_t ps = (int64_t)a.items_cap * P;
  const float* g = a.g + (int64_t)item * P;
  float st = 0.0f, sr = 0.0f;
  for (int i = t; i < P; i += 256) {
    st += fmaxf(g[i], 0.01f);
    sr += fmaxf(g[ps + i], 0.01f);
  }
  const float Lt = block_sum(st, s_tmp) / (float)P;
  const float Lr = block_sum(sr, s_tmp) / (float)P;
  const float logL = log10f(Lr);
  float ind = (logL - a.logL_first) / (a.logL_last - a.logL_first) * (float)(CVVDP_CSF_NODES - 1);
  ind = fminf(fmaxf(ind, 0.0f), (float)(CVVDP_CSF_NODES - 1));
  const int i0 = (int)ind;
  const float fr = ind - (float)i0;
  const int i1 = min(i0 + 1, CVVDP_CSF_NODES - 1);
  float S[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    S[c] = exp10f(a.lut[c * CVVDP_CSF_NODES + i0] * (1.0f - fr) + a.lut[c * CVVDP_CSF_NODES + i1] * fr) * a.sens_mul;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = t; i < P; i += 256) {
    float D[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < a.nch) {
        const float ct = fminf(g[(2 * c) * ps + i] / Lt, 1000.0f);
        const float cr = fminf(g[(2 * c + 1) * ps + i] / Lr, 1000.0f);
        D[c] = fabsf(ct - cr) * S[c];
        if (a.fdump) {
          a.fdump[(int64_t)c * ps + (int64_t)item * P + i] = fabsf(ct) * S[c];
          a.fdump[(int64_t)(4 + c) * ps + (int64_t)item * P + i] = fabsf(cr) * S[c];
        }
        const float de = D[c] + kEps;
        acc[c] += de * de - kEps * kEps;
        if (a.ddump) a.ddump[(int64_t)c * ps + (int64_t)item * P + i] = D[c];
      }
    }
    if (a.dchr) {
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < a.nch) s += powf(D[c] * a.hw[c] + kEps, a.beta_tch) - a.eps_btch;
      a.dchr[(int64_t)item * P + i] = powf(s + kEps, 1.0f / a.beta_tch) - a.eps_inv_btch;
    }
  }
  const int f = item / a.batch, b = item - f * a.batch;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float s = block_sum(acc[c], s_tmp);
    if (t == 0 && c < a.nch) {
      const float q = sqrtf(s / (float)P + kEps) - sqrtf(kEps);  // safe_pow(sum/N, 1/beta), beta = 2
      a.q_out[(((int64_t)b * a.nch + c) * a.q_frames + a.q_frame_offset + f) * a.q_levels + a.level] = q;
    }
  }
}

void launch_baseband(const BaseArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_baseband, dim3(a.items), dim3(256), 0, s, a);
}

// Sum the per-block partials in a fixed order (deterministic, double accumulation) and finish the
// spatial p-norm: Q = (sum/N + eps)^(1/2) - eps^(1/2)   (cvvdp_metric.py:1048, beta = 2).
__global__ __launch_bounds__(64) void k_finalize(FinalizeArgs a) {
  const int item = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
  if (c >= a.nch) return;
  const float* p = a.partial + (int64_t)item * a.nblk * 4 + c;
  double s = 0.0;
  for (int i = lane; i < a.nblk; i += 64) s += (double)p[(int64_t)i * 4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) {
    const int f = item / a.batch, b = item - f * a.batch;
    const float mean = (float)(s / (double)a.P - a.sub_per_term);
    a.q_out[(((int64_t)b * a.nch + c) * a.q_frames + a.q_frame_offset + f) * a.q_levels + a.level] =
        sqrtf(mean + kEps) - sqrtf(kEps);
  }
}

void launch_finalize(const FinalizeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_finalize, dim3(a.items, a.nch), dim3(64), 0, s, a);
}

}  // namespace cvvdp
