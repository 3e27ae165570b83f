// K3..K7 fused, vectorised variant for wide levels (W % 8 == 0): same arithmetic as band.hip, laid out
// for CDNA4 memory instructions.
//
//   wave  <-> colour/temporal channel c (so q_c, the cross-channel weights into c and the CSF row of c
//             are wave-uniform scalars)
//   lane  <-> 4 adjacent columns        (16-byte global loads of g, ds_read_b128/ds_write_b128 for every
//             per-row exchange, 4 independent SFU chains per lane)
//   block <-> strip of 240 interior columns (+8 aligned halo columns each side), marching down a row
//             segment exactly like k_band.
//
// Two barriers per row.  Global loads for row r+1 (coarse rows for the expand, g prefetch) are issued in
// the second phase of row r, behind ~100 FMAs of blur, so HBM latency is off the critical path:
//
//   phase 1:  pooling stage of the row finished last iteration (reads s_q of all channels, s_d)
//             contrast/CSF stage of row r (reads s_ve, s_lum; writes s_m and the s_d ring)
//   barrier
//   phase 2:  13-tap horizontal blur (5 ds_read_b128), 13-row register window, vertical blur,
//             Mq = safe_pow(blur*10^mask_c, q_c) -> s_q ; luminance terms of row r+1 -> s_lum ;
//             vertical expand of row r+1 (r+2 for the luminance planes) -> s_ve ; prefetch g
//   barrier
//
// Image-edge halo columns are produced by the in-image lanes as mirrored LDS writes (reflect padding
// of the blur), halo rows by evaluating the reflected row.
#include "kernels.h"

namespace cvvdp {

constexpr int B4_R = 6;            // blur radius
constexpr int B4_BW = 13;
constexpr int B4_HALO = 8;         // aligned halo columns per side
constexpr int B4_SW = 256 - 2 * B4_HALO;  // 240 interior columns per strip
constexpr int B4_VE = 136;         // s_ve row: element 4+i = coarse column cb+i (i = 0..127), 16-byte aligned chunks

struct f4 { float v[4]; };
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_stream4(const float* p) {   // read-once data: nontemporal, stays out of L2
  const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(q.x, q.y, q.z, q.w);
}

__device__ __forceinline__ f4 lds_read4(const float* p) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  return f4{{q.x, q.y, q.z, q.w}};
}
__device__ __forceinline__ void lds_write4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ int refl(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

template <int NCH, bool HEAT>
__global__ __launch_bounds__(64 * NCH, 3) void k_band4(BandArgs a) {
  constexpr int NP = 2 * NCH;
  // s_ve is a ring of two rows (row parity): the luminance planes (0, 1; wave 0) run TWO rows ahead so that all
  // waves can derive the per-column luminance terms of row r+1 (s_lum) during phase 2 of row r.
  __shared__ __attribute__((aligned(16))) float2 s_ve[2][NP][B4_VE / 2];   // float2 rows: guaranteed 8-byte aligned ds_read_b64
  __shared__ __attribute__((aligned(16))) float s_lum[4][256];             // 1/L_T, 1/L_R, CSF-LUT fraction, LUT byte offset
  __shared__ __attribute__((aligned(16))) float s_m[NCH][256];
  __shared__ __attribute__((aligned(16))) float s_q[NCH][256];
  __shared__ __attribute__((aligned(16))) float s_d[B4_R + 1][NCH][B4_SW];   // lane-private ring of |T'-R'|: interior columns only
  __shared__ __attribute__((aligned(16))) float s_h[HEAT ? NCH : 1][HEAT ? 256 : 4];   // heat-map terms of the pooled row, per channel
  __shared__ float s_lut[NCH][CVVDP_CSF_NODES + 1];                         // [32] = [31]: lerp partner of the last node

  const int t = threadIdx.x;
  const int c = t >> 6, j = t & 63;                 // channel (wave), lane
  const int strip = blockIdx.x, seg = blockIdx.y, item = blockIdx.z;
  const int H = a.H, W = a.W, Hc = a.Hc, Wc = a.Wc;
  const int x0 = strip * B4_SW;
  const int fc0 = x0 - B4_HALO + 4 * j;             // first of this lane's 4 columns
  const bool in_img = fc0 >= 0 && fc0 < W;          // all four in or all four out (W % 4 == 0)
  const bool interior = j >= 2 && j < 62 && fc0 < W;  // columns whose result is pooled
  const int cb = (x0 - B4_HALO) / 2;                // coarse column of s_ve[.][1]
  const int ys = seg * a.seg_h, ye = min(H, ys + a.seg_h);

  const int64_t P = (int64_t)H * W, Pc = (int64_t)Hc * Wc;
  const int64_t gps = (int64_t)a.items_cap * P, gcps = (int64_t)a.items_cap * Pc;
  const float* gT = a.g + (int64_t)item * P + (2 * c) * gps;      // test plane of this channel
  const float* gR = gT + gps;                                      // reference plane
  // stage-1 role of this lane: plane 2c + (j>>5), coarse chunk j&31
  const int vp = 2 * c + (j >> 5);
  const int vch = j & 31;
  const int vcx = cb + 4 * vch;                                    // first coarse column of the chunk
  const float* gcp = a.gc + (int64_t)item * Pc + vp * gcps;

  for (int i = t; i < NCH * (CVVDP_CSF_NODES + 1); i += 64 * NCH) {
    const int cc = i / (CVVDP_CSF_NODES + 1), k = min(i - cc * (CVVDP_CSF_NODES + 1), CVVDP_CSF_NODES - 1);
    // log2-domain CSF row with the constant gains folded in: S*ch_gain*band_mul = 2^(lut*log2(10) + log2(sens_mul*ch_gain*band_mul))
    s_lut[cc][i - cc * (CVVDP_CSF_NODES + 1)] = a.lut[cc * CVVDP_CSF_NODES + k] * kLog2_10 + fast_log2(a.sens_mul * a.ch_gain[cc] * a.band_mul);
  }
  for (int i = t; i < 2 * NP * (B4_VE / 2); i += 64 * NCH) (&s_ve[0][0][0])[i] = make_float2(0.0f, 0.0f);   // unwritten apron elements
  __syncthreads();
  const float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
  const float ind_scale = (float)(CVVDP_CSF_NODES - 1) / (a.logL_last - a.logL_first);
  const float qc = a.q[c], eps_qc = a.eps_q[c];
  const float xw0 = a.xw[0 * 4 + c], xw1 = a.xw[1 * 4 + c], xw2 = a.xw[2 * 4 + c], xw3 = a.xw[3 * 4 + c];

  // vertical half of the expand for fine row rr -> s_ve (lpyr_dec.py:229-232), split in two so that the
  // global loads are issued a whole row-iteration before their values are needed
  const int cx = min(max(vcx, 0), Wc - 4);
  const bool clampL = vcx < 0, clampR = vcx >= Wc;
  const bool edge_block = cb < 0 || cb + 128 > Wc;   // block-uniform: only edge strips pay for the replicate selects
  float4 cA, cB, cC;    // coarse rows my-1, my, my+1 (clamped) of the chunk
  auto stage1_load = [&](int rr) {
    const int my = rr >> 1;
    const int ya = max(my - 1, 0), yb = min(my + 1, Hc - 1);
    cA = *reinterpret_cast<const float4*>(gcp + (int64_t)ya * Wc + cx);
    cB = *reinterpret_cast<const float4*>(gcp + (int64_t)my * Wc + cx);
    cC = *reinterpret_cast<const float4*>(gcp + (int64_t)yb * Wc + cx);
  };
  auto stage1_finish = [&](int rr, int buf) {
    if (edge_block && (clampL || clampR)) {      // replicate column 0 / Wc-1 for chunks left / right of the image
      const float ra = clampL ? cA.x : cA.w, rb = clampL ? cB.x : cB.w, rc = clampL ? cC.x : cC.w;
      cA = make_float4(ra, ra, ra, ra); cB = make_float4(rb, rb, rb, rb); cC = make_float4(rc, rc, rc, rc);
    }
    const float m0[4] = {cA.x, cA.y, cA.z, cA.w}, m1[4] = {cB.x, cB.y, cB.z, cB.w}, m2[4] = {cC.x, cC.y, cC.z, cC.w};
    float o[4];
    if (rr & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = m1[i] * eo + m2[i] * eo;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = m0[i] * e0 + m1[i] * e1 + m2[i] * e0;
    }
    lds_write4(reinterpret_cast<float*>(&s_ve[buf][vp][2 + 2 * vch]), o);
  };

  // horizontal half of the expand for this lane's 4 columns from 4 coarse values (lpyr_dec.py:234-237)
  auto expand4 = [&](const float2* row, float (&ex)[4]) {
    // coarse cb+2j-1 .. cb+2j+2 live at elements 2j+3 .. 2j+6: three aligned ds_read_b64 (conflict-free)
    const float2 p0 = row[j + 1];
    const float2 p1 = row[j + 2];
    const float2 p2 = row[j + 3];
    const float A = p0.y, B = p1.x, C = p1.y, D = p2.x;
    ex[0] = A * e0 + B * e1 + C * e0;
    ex[1] = B * eo + C * eo;
    ex[2] = B * e0 + C * e1 + D * e0;
    ex[3] = C * eo + D * eo;
  };

  // Per-column luminance terms of one row, shared by all channels (lpyr_dec.py:394,:408; interp.py:93): every
  // thread expands the luminance planes for ONE column (blocks of 64*NCH columns) and publishes 1/L_T, 1/L_R,
  // the LUT interpolation fraction and the LUT byte offset, so each channel wave spends 1 SFU op per pixel
  // on the CSF instead of 4.  Element 4+i of an s_ve row is coarse column cb+i; fine column col -> 4+(col>>1).
  const bool lodd = t & 1;                           // 64*NCH is even: a thread's columns keep their parity
  const float lwa = lodd ? 0.0f : e0, lwb = lodd ? eo : e1, lwc = lodd ? eo : e0;
  auto lum_prep = [&](int buf) {
    const float* yT = reinterpret_cast<const float*>(&s_ve[buf][0][0]);
    const float* yR = reinterpret_cast<const float*>(&s_ve[buf][1][0]);
    for (int col = t; col < 256; col += 64 * NCH) {
      const int e = 4 + (col >> 1);
      const float eyT = yT[e - 1] * lwa + yT[e] * lwb + yT[e + 1] * lwc;
      const float eyR = yR[e - 1] * lwa + yR[e] * lwb + yR[e + 1] * lwc;
      const float Lt = fmaxf(eyT, 0.01f), Lr = fmaxf(eyR, 0.01f);              // lpyr_dec.py:394
      float ind = (fast_log2(Lr) * kLog10_2 - a.logL_first) * ind_scale;       // lpyr_dec.py:408, interp.py:93
      ind = fminf(fmaxf(ind, 0.0f), (float)(CVVDP_CSF_NODES - 1));
      const int i0 = (int)ind;
      s_lum[0][col] = fast_rcp(Lt);
      s_lum[1][col] = fast_rcp(Lr);
      s_lum[2][col] = ind - (float)i0;
      s_lum[3][col] = __int_as_float(i0 * 4);
    }
  };

  // Vertical-blur window: slot s of column i holds one horizontally blurred row.  The newest row is written
  // into slot (row index mod 13) with an M0-relative register write (dynamic insertelement on a register
  // vector); the 13 weights -- wave-uniform scalars -- are rotated instead of the data, so the FMAs use
  // static register indices and no 13-way switch / PHI copies are needed.
  typedef float v32f __attribute__((ext_vector_type(32)));
  v32f winA = 0.0f, winB = 0.0f;   // columns (0,1) and (2,3) interleaved: element 2s+i = slot s of column i
  v2f be[6], bo[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) { be[m] = v2f{a.blur[2 * m], a.blur[2 * m + 1]}; bo[m] = v2f{a.blur[2 * m + 1], a.blur[2 * m + 2]}; }
  const float b0 = a.blur[0], b12 = a.blur[12];
  float wr[B4_BW];     // wr[s] = weight of slot s for the NEXT row to be written into slot 0
#pragma unroll
  for (int k = 0; k < B4_BW; ++k) wr[k] = a.blur[(k + B4_BW - 1) % B4_BW];
  float acc = 0.0f;

  // pooling stage of centre row y (cvvdp_metric.py:849-856, 722): needs s_q of all channels and s_d
  auto stage3c = [&](int y) {
    const int ds = ((y % (B4_R + 1)) + (B4_R + 1)) % (B4_R + 1);
    const f4 q0 = lds_read4(&s_q[0][4 * j]), q1 = lds_read4(&s_q[1][4 * j]), q2 = lds_read4(&s_q[2][4 * j]);
    f4 q3 = f4{{0.0f, 0.0f, 0.0f, 0.0f}};
    if constexpr (NCH == 4) q3 = lds_read4(&s_q[3][4 * j]);
    const f4 d = lds_read4(&s_d[ds][c][4 * j - B4_HALO]);
    float D[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float M = q0.v[i] * xw0 + q1.v[i] * xw1 + q2.v[i] * xw2 + q3.v[i] * xw3;     // cvvdp_metric.py:758-760
      // Du = X/(1+M); D = dmax*Du/(dmax+Du) = dmax*X / (dmax*(1+M) + X): one reciprocal (:855-856, :949-950)
      const float X = fast_pow(d.v[i] + kEps, a.mask_p) - a.eps_p;
      D[i] = a.dmax * X * fast_rcp(a.dmax + a.dmax * M + X);
      const float de = D[i] + kEps;
      acc += de * de - kEps * kEps;
    }
    if (a.ddump) *reinterpret_cast<float4*>(a.ddump + (int64_t)c * a.items_cap * P + (int64_t)item * P + (int64_t)y * W + fc0) =
        make_float4(D[0], D[1], D[2], D[3]);
    if constexpr (HEAT) {   // this channel's term of the per-pixel channel norm (cvvdp_metric.py:728-734)
      float ht[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ht[i] = fast_pow(D[i] * a.hw[c] + kEps, a.beta_tch) - a.eps_btch;
      lds_write4(&s_h[c][4 * j], ht);
    }
  };
  // heat-map band of row y from the channel terms published by stage3c (needs a barrier in between): one column
  // per thread, lp_norm over channels, stored /band_mul as lpyr_dec_2.set_lband does (lpyr_dec.py:308-314)
  auto heat_row = [&](int y) {
    if constexpr (HEAT) {
      for (int col = t; col < 256; col += 64 * NCH) {
        const int xs = x0 - B4_HALO + col;
        if (col >= B4_HALO && col < 256 - B4_HALO && xs < W) {
          float sum = s_h[0][col] + s_h[1][col] + s_h[2][col];
          if constexpr (NCH == 4) sum += s_h[3][col];
          a.dchr[(int64_t)item * P + (int64_t)y * W + xs] = (fast_pow(sum + kEps, 1.0f / a.beta_tch) - a.eps_inv_btch) / a.band_mul;
        }
      }
    }
  };

  // ---- prologue: expand rows (luminance: two of them), g prefetch and luminance terms for the first row
  const int ahead = c == 0 ? 2 : 1;                  // wave-uniform
  int r = ys - B4_R;
  int rr = refl(r, H);
  stage1_load(rr);
  float4 pT = make_float4(0, 0, 0, 0), pR = make_float4(0, 0, 0, 0);
  // g rows are streamed from HBM two rows ahead.  Every lane issues every load (out-of-image halo lanes and
  // rows past the segment read a clamped, valid address) so that each iteration has the same five loads in
  // flight and the waits can be exact vmcnt(N) counts instead of a full drain.
  const int fcl = min(max(fc0, 0), W - 4);
  float4 nT, nR;
  {
    pT = ld_stream4(gT + (int64_t)rr * W + fcl);
    pR = ld_stream4(gR + (int64_t)rr * W + fcl);
    const int r1 = refl(r + 1, H);
    nT = ld_stream4(gT + (int64_t)r1 * W + fcl);
    nR = ld_stream4(gR + (int64_t)r1 * W + fcl);
  }
  stage1_finish(rr, r & 1);
  if (c == 0) {
    const int r1 = refl(r + 1, H);
    stage1_load(r1);
    stage1_finish(r1, (r + 1) & 1);
  }
  __syncthreads();
  lum_prep(r & 1);
  __syncthreads();

  for (; r < ye + B4_R; ++r) {
    // ================= phase 1
    const int yprev = r - 1 - B4_R;                  // row whose Mq was published last iteration
    if (interior && yprev >= ys) stage3c(yprev);
    float m[4], d[4];
    if (in_img) {
      float exT[4], exR[4];
      expand4(s_ve[r & 1][2 * c], exT);
      expand4(s_ve[r & 1][2 * c + 1], exR);
      const f4 rLt = lds_read4(&s_lum[0][4 * j]), rLr = lds_read4(&s_lum[1][4 * j]), fr = lds_read4(&s_lum[2][4 * j]);
      const f4 lo = lds_read4(&s_lum[3][4 * j]);
      const float gt[4] = {pT.x, pT.y, pT.z, pT.w}, gr[4] = {pR.x, pR.y, pR.z, pR.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* lp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(&s_lut[c][0]) + __float_as_int(lo.v[i]));
        const float l0 = lp[0], l1 = lp[1];
        const float S = fast_exp2(l0 + (l1 - l0) * fr.v[i]);                     // csf.py:49, cvvdp_metric.py:709,:836
        const float ct = fminf((gt[i] - exT[i]) * rLt.v[i], 1000.0f);            // lpyr_dec.py:402 (band gain :66 is in S)
        const float cr = fminf((gr[i] - exR[i]) * rLr.v[i], 1000.0f);
        const float Tp = ct * S, Rp = cr * S;
        m[i] = fminf(fabsf(Tp), fabsf(Rp));                                      // cvvdp_metric.py:845
        d[i] = fabsf(Tp - Rp);
      }
      lds_write4(&s_m[c][4 * j], m);
      const int ds = ((r % (B4_R + 1)) + (B4_R + 1)) % (B4_R + 1);
      if (interior) lds_write4(&s_d[ds][c][4 * j - B4_HALO], d);
      // reflect padding of the blur at the image's left/right edge: mirror columns 1..6 / W-7..W-2
      if (fc0 < 8) {                                  // strip 0, lanes holding columns 0..7
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int x = fc0 + i;
          if (x >= 1 && x <= B4_R) s_m[c][B4_HALO - x] = m[i];               // column -x
        }
      }
      if (fc0 + 8 >= W) {                             // lanes holding columns W-8..W-1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int x = fc0 + i;
          const int idx = 2 * (W - 1) - x - (x0 - B4_HALO);                  // column 2(W-1)-x
          if (x <= W - 2 && x >= W - 1 - B4_R && idx < 256) s_m[c][idx] = m[i];
        }
      }
    }
    __syncthreads();
    // ================= phase 2
    // issue the global loads right after the barrier: coarse rows are consumed at the end of this phase
    // (-> s_ve), the g rows a whole iteration later
    const bool more = r + 1 < ye + B4_R;
    if (yprev >= ys) heat_row(yprev);                 // terms of row yprev were published in phase 1
    const int rs = min(refl(r + ahead, H), H - 1);
    stage1_load(rs);
    pT = nT; pR = nR;                                 // row r+1, requested a whole iteration ago
    {
      const int r2 = min(refl(r + 2, H), H - 1);
      nT = ld_stream4(gT + (int64_t)r2 * W + fcl);
      nR = ld_stream4(gR + (int64_t)r2 * W + fcl);
    }
    if (more) lum_prep((r + 1) & 1);                  // luminance planes of row r+1 were published an iteration ago
    const int yc = r - B4_R;
    if (interior) {
      // horizontal 13-tap blur of 4 adjacent outputs on packed fp32 FMAs (v_pk_fma_f32: two taps per
      // instruction).  x[2p], x[2p+1] sit in an aligned register pair xp[p]; output i reads taps x[i+2 .. i+14],
      // so even outputs pair the weights as (b0,b1)(b2,b3).. + b12 and odd outputs as b0 + (b1,b2)(b3,b4)..
      const v4f* row = reinterpret_cast<const v4f*>(&s_m[c][4 * j - 8]);
      const v4f a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3], a4 = row[4];
      const v2f xp[10] = {a0.xy, a0.zw, a1.xy, a1.zw, a2.xy, a2.zw, a3.xy, a3.zw, a4.xy, a4.zw};
      float h[4];
      {
        v2f s0 = be[0] * xp[1], s1 = bo[0] * xp[2], s2 = be[0] * xp[2], s3 = bo[0] * xp[3];
#pragma unroll
        for (int m = 1; m < 6; ++m) {
          s0 += be[m] * xp[1 + m]; s1 += bo[m] * xp[2 + m]; s2 += be[m] * xp[2 + m]; s3 += bo[m] * xp[3 + m];
        }
        h[0] = (s0.x + b12 * xp[7].x) + s0.y;
        h[1] = (s1.x + b0 * xp[1].y) + s1.y;
        h[2] = (s2.x + b12 * xp[8].x) + s2.y;
        h[3] = (s3.x + b0 * xp[2].y) + s3.y;
      }
      float v[4];
      const int slot = (r - (ys - B4_R)) % B4_BW;          // wave-uniform
      winA[2 * slot] = h[0]; winA[2 * slot + 1] = h[1]; winB[2 * slot] = h[2]; winB[2 * slot + 1] = h[3];
      // weight of slot s when the newest row sits in `slot`: window position j = (s - slot - 1) mod 13 -> wr[]
      // is kept rotated so that wr[s] is exactly that weight; columns (0,1) and (2,3) share one packed FMA
      {
        v2f va = 0.0f, vb = 0.0f;
#pragma unroll
        for (int sdx = 0; sdx < B4_BW; ++sdx) {
          const v2f wa = {winA[2 * sdx], winA[2 * sdx + 1]}, wb = {winB[2 * sdx], winB[2 * sdx + 1]};
          const v2f ww = {wr[sdx], wr[sdx]};
          va += ww * wa; vb += ww * wb;
        }
        v[0] = va.x; v[1] = va.y; v[2] = vb.x; v[3] = vb.y;
      }
      if (yc >= ys) {
        float Mq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) Mq[i] = fast_pow(fabsf(v[i] * a.mask_c10) + kEps, qc) - eps_qc;   // cvvdp_metric.py:849
        lds_write4(&s_q[c][4 * j], Mq);
      }
    }
    {   // rotate the blur weights for the next row (scalar ALU)
      const float last = wr[B4_BW - 1];
#pragma unroll
      for (int k = B4_BW - 1; k > 0; --k) wr[k] = wr[k - 1];
      wr[0] = last;
    }
    stage1_finish(rs, (r + ahead) & 1);
    __syncthreads();
  }
  // ---- epilogue: pooling stage of the last centre row
  if (interior && (ye - 1) >= ys) stage3c(ye - 1);
  if constexpr (HEAT) {
    __syncthreads();
    if ((ye - 1) >= ys) heat_row(ye - 1);
  }

#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (j == 0) {
    const int nblk = a.n_strip * a.n_seg;
    a.partial[((int64_t)item * nblk + (seg * a.n_strip + strip)) * 4 + c] = acc;
  }
}

void launch_band4(const BandArgs& a, hipStream_t s) {
  dim3 grid(a.n_strip, a.n_seg, a.items);
  if (a.dchr) {
    if (a.nch == 4) hipLaunchKernelGGL((k_band4<4, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_band4<3, true>), grid, dim3(192), 0, s, a);
  } else {
    if (a.nch == 4) hipLaunchKernelGGL((k_band4<4, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_band4<3, false>), grid, dim3(192), 0, s, a);
  }
}

}  // namespace cvvdp
