// K2 + K3..K7 fused for the aligned levels of the plain scoring path: k_band4's row march (band4.hip: same stages, same arithmetic)
// that COMPUTES the coarse level it expands from, instead of loading it, and writes it out as the next pyramid level.
//
// Why: the level-l planes were written once (FIR) and read twice -- by the reduce pass and by the band kernel.  The reduce pass is
// HBM-bound (4.4 ms for 4K x 64 at level 0), the band kernel VALU-bound (5.6 ms): computing the 5x5 reduce from the rows the band
// kernel streams anyway costs it +13 % instructions (measured: 5.6 -> 6.6 ms at two blocks per CU) and removes the pass.
//
//   * Contrast of fine row r needs coarse rows (r>>1)-1 .. (r>>1)+1, i.e. level-l rows up to r+4.  The raw rows therefore wait in
//     an 8-row register RING per plane (row s in slot s mod 8; the row loop is unrolled eight times, so every slot index is
//     static: the hand-issued loads land in their slot and nothing is ever moved).  Step s uses row s, reduces row s+5, and has
//     rows s+6, s+7 in flight.
//   * Horizontal half of the reduce (lpyr_dec.py:186-211; same formulas as pyramid.hip hreduce_row): lane j owns fine columns
//     fc0 .. fc0+3 = coarse columns X0 = fc0/2 (taps fc0-2 .. fc0+2) and X0+1 (taps fc0 .. fc0+4).  The three neighbour samples
//     (two to the left, one to the right) are loaded a second time with their own small loads (L1 hits; issued one row ahead) --
//     every lane of the strip, the halo lanes included, so all 128 coarse columns of the strip are exact.
//   * Vertical half: two running partial sums (the two coarse rows under construction); an odd level-l row adds k3 / k1 to them,
//     an even row completes the older one (k4), continues the younger (k2) and starts a new one (k0).  The reference's border
//     terms (zero padding + the extra taps of the first / last row and column, including the column edge that tests the ROW
//     parity, lpyr_dec.py:206) are folded into the per-row scalar weights and two per-lane edge weights.
//   * A completed coarse row rolls the 3-row expand window (as the loaded row did in k_band4) and is stored as level l+1 by
//     the lanes that own it (interior columns of the strip, rows of the segment).
//   * Rows above / below the image: the 13-row blur window of the reflected rows is filled by symmetry instead of recomputing
//     them -- the top segment starts at row 0 and writes rows 1..6 into their mirror slots too, the bottom segment copies the
//     mirrored slot (reflect padding of the blurred map = the mirrored row of the horizontally blurred map).
//
// Used by launch_band (core.cpp) for even W, no heat map / dump / features; everything else keeps k_band4 + the reduce pass.
// The streamed loads are hand-managed like k_band4's (six per row: four neighbour loads of row s+6, two row loads of row s+7;
// one uniform `s_waitcnt vmcnt(2)` per step); tools/check_band4_isa.py checks this kernel's assembly too.
#include <type_traits>
#include "kernels.h"

namespace cvvdp {

namespace {

constexpr int F_R = 6;             // blur radius
constexpr int F_BW = 13;
constexpr int F_HALO = 8;          // aligned halo columns per side
constexpr int F_SW = 256 - 2 * F_HALO;   // 240 interior columns per strip (= kBand4StripWidth)
constexpr int F_VE = 136;          // s_ve row: element 4+i = coarse column cb+i (i = 0..127)

struct ff4 { float v[4]; };
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ ff4 f_lds_read4(const float* p) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  return ff4{{q.x, q.y, q.z, q.w}};
}
__device__ __forceinline__ void f_lds_write4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace

// EDGE != 0: the launch holds the strips that touch the left or right image border (strip 0 and the last one or two); the other strips
// run the instantiation without any of the border code (zero masks, first / last column taps, replicas, blur mirrors), as a second
// launch beside it.  EDGE == 1: W % 4 == 0 (every lane inside the image holds four columns).  EDGE == 2: W % 4 == 2 -- the lane of
// columns W-2, W-1 is a PARTIAL lane: its 16-byte load is clamped to the row's last four columns and shifted into place (the two
// columns right of the image become the reduce's zero padding), it owns ONE coarse column (the last one: the first / last column
// rule moves from the lane's second coarse column to its first), stores one level-(l+1) sample instead of two, the blur's reflect
// padding is written column by column, and its two columns outside the image are masked out of the pooling.
//
// HEAT: the level's heat-map band as well (band4.hip, HEAT): the channel's term in the pooling stage, the channel norm one column per
// thread a barrier later.  The border instantiations have no registers for it (54 / 16 spilled VGPRs, profiles/r03_dev_notes.txt 17), and
// a spill reload among hand-issued loads breaks their wait counts: the HEAT instantiations therefore use ordinary loads the compiler
// tracks itself (F_SAFE).  They are slower per strip, and they are the two or three border strips of a level only -- the other
// strips run k_band4s<HEAT> (band4s.hip) beside them.
struct __attribute__((packed, aligned(4))) f_u4 { float x, y, z, w; };     // (rows of W % 4 == 2 frames are 8-byte aligned)
struct __attribute__((packed, aligned(4))) f_u2 { float x, y; };

// FEAT (band4.hip, FEATURES): the feature-pooling statistics of the ML heads -- 24 more registers, the same treatment.
template <int NCH, int EDGE, bool HEAT, bool FEAT>
__device__ __forceinline__ void band4f_body(const BandArgs& a) {
  constexpr bool RAG = EDGE == 2;
#ifdef CVVDP_SAFE_LOADS
  constexpr bool F_SAFE = true;       // `make safe`: every instantiation (tests/test_safe_loads.py)
#else
  // EDGE == 1 (round 6): since the border strips of W % 4 == 0 frames run k_band4s_edge, the plain EDGE == 1 instantiation only serves the
  // one-wave A/B layout (cvvdp_clip.band_layout = 1, a test reference).  Its hand-issued loads held only as long as the register
  // allocator happened to keep the ring in place: fixing the expand's operation order (kernels.h expand_even) made it copy ring
  // registers inside the loop, the ISA check failed the build.  A kernel nobody's product path runs is not worth that: ordinary loads.
  constexpr bool F_SAFE = HEAT || FEAT || EDGE == 1;
#endif
  constexpr int NP = 2 * NCH;
  __shared__ __attribute__((aligned(16))) float2 s_ve[2][NP][F_VE / 2];
  __shared__ __attribute__((aligned(16))) float s_lum[2][256];            // 1/L_T, 1/L_R
  __shared__ __attribute__((aligned(16))) float s_S[NCH][256];
  __shared__ __attribute__((aligned(16))) float s_m[NCH][256];
  __shared__ __attribute__((aligned(16))) float s_q[NCH][256];
  __shared__ __attribute__((aligned(16))) float s_d[F_R + 1][NCH][F_SW];  // lane-private ring of |T'-R'| + eps
  __shared__ __attribute__((aligned(8))) float2 s_lut[NCH][CVVDP_CSF_NODES];
  __shared__ __attribute__((aligned(16))) float s_h[HEAT ? NCH : 1][HEAT ? 256 : 4];   // heat-map terms of the pooled row, per channel

  const int t = threadIdx.x;
  const int c = __builtin_amdgcn_readfirstlane(t >> 6);
  const int j = t & 63;
  const int per_xcd = a.per_xcd;
  const int wu = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);      // XCD-aware work-unit order (band4.hip)
  if (wu >= a.n_strip_l * a.n_seg * a.items) return;
  const int sl = wu % a.n_strip_l, seg = (wu / a.n_strip_l) % a.n_seg, item = wu / (a.n_strip_l * a.n_seg);
  // EDGE launch: strip 0 and the last n_strip_l - 1 strips; the other launch: strips strip0 .. strip0 + n_strip_l - 1
  const int strip = EDGE ? (sl == 0 ? 0 : a.n_strip - a.n_strip_l + sl) : a.strip0 + sl;
  const int H = a.H, W = a.W, Hc = a.Hc, Wc = a.Wc;
  const int x0 = strip * F_SW;
  const int fc0 = x0 - F_HALO + 4 * j;
  const bool in_img = fc0 >= 0 && fc0 < W;
  const bool edge_r = x0 + F_SW + F_HALO >= W;     // (>=: a strip whose LAST lane holds columns W-4 .. W-1 computes the last coarse column, band4f_right_edge_strips)
  constexpr bool edge_lr = EDGE != 0;               // (launch_band4f deals the strips accordingly)
  const bool part = RAG && in_img && fc0 + 4 > W;   // the partial lane (fc0 == W - 2)
  const bool interior = j >= 2 && j < 62 && fc0 < W;
  const int cb = (x0 - F_HALO) / 2;
  const int ys = seg * a.seg_h, ye = min(H, ys + a.seg_h);
  const bool top_seg = seg == 0;

  const int64_t P = (int64_t)H * W, Pc = (int64_t)Hc * Wc;
  const int64_t gps = (int64_t)a.items_cap * P, gcps = (int64_t)a.items_cap_c * Pc;
  const float* gT = a.g + (int64_t)item * P + (2 * c) * gps;
  const float* gR = gT + gps;
  float* g1T = a.g1_out + (int64_t)item * Pc + (2 * c) * gcps;     // level l+1 planes of this channel, written here
  float* g1R = g1T + gcps;

  for (int i = t; i < NCH * CVVDP_CSF_NODES; i += 64 * NCH) {
    const int cc = i / CVVDP_CSF_NODES, k = i - cc * CVVDP_CSF_NODES;
    const float l0 = a.lut[cc * CVVDP_CSF_NODES + k] * kLog2_10 + fast_log2(a.sens_mul * a.ch_gain[cc] * a.band_mul);
    const float l1 = a.lut[cc * CVVDP_CSF_NODES + min(k + 1, CVVDP_CSF_NODES - 1)] * kLog2_10 + fast_log2(a.sens_mul * a.ch_gain[cc] * a.band_mul);
    s_lut[cc][k] = make_float2(l0, l1 - l0);
  }
  for (int i = t; i < 2 * NP * (F_VE / 2); i += 64 * NCH) (&s_ve[0][0][0])[i] = make_float2(0.0f, 0.0f);
  __syncthreads();
  float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
  float mask_p = a.mask_p, eps_p = a.eps_p;
#define F_IN_VGPR(x) asm volatile("" : "+v"(x))
  float ind_k1 = a.ind_k1, ind_k0 = a.ind_k0;
  float qc = a.q[c];
  float xw0 = a.xw[0 * 4 + c], xw1 = a.xw[1 * 4 + c], xw2 = a.xw[2 * 4 + c], xw3 = a.xw[3 * 4 + c];
  float m1c = a.m1[c];
  float inv_dmax = a.inv_dmax;
  if constexpr (!RAG) {   // (the EDGE == 2 instantiation is short of VGPRs instead: its constants stay scalar, spilled or not)
    F_IN_VGPR(ind_k0); F_IN_VGPR(xw1); F_IN_VGPR(xw2); F_IN_VGPR(xw3); F_IN_VGPR(m1c); F_IN_VGPR(inv_dmax);
    F_IN_VGPR(e0); F_IN_VGPR(e1); F_IN_VGPR(eo); F_IN_VGPR(mask_p); F_IN_VGPR(eps_p);
    F_IN_VGPR(qc); F_IN_VGPR(ind_k1); F_IN_VGPR(xw0);
  }
#undef F_IN_VGPR

  // ---- the reduce's lane constants.  Samples outside the image are the reference's zero padding: addresses are clamped, the
  // loaded values multiplied by 0 -- only in the strips at the left / right image border (edge_lr), all other lanes see 1.
  const float rk0 = a.rk[0], rk1 = a.rk[1], rk2 = a.rk[2], rk3 = a.rk[3], rk4 = a.rk[4];
  const uint32_t goff = (uint32_t)min(max(fc0, 0), W - 4) * 4u;                       // own four columns
  const uint32_t loff = (uint32_t)min(max(fc0 - 2, 0), W - 2) * 4u;                   // columns fc0-2, fc0-1
  const uint32_t roff = (uint32_t)min(max(fc0 + 4, 0), W - 1) * 4u;                   // column fc0+4
  // (the zero masks and the first / last column's extra taps are made from fc0 inside the edge_lr branches: nothing of them is live in
  // the strips away from the image border)
  const int lane_first = 2;                                       // strip 0: the lane of fine column 0
  const int lane_last = (W - (RAG ? 2 : 4) - (x0 - F_HALO)) >> 2;  // the lane of the last fine columns (edge_r strips: 0 .. 63)

  float4 cA = make_float4(0, 0, 0, 0), cB = cA, cC = cA;          // coarse rows my-1, my, my+1 of this lane's two coarse columns: (T0, T1, R0, R1)
  float4 rP = cA, rQ = cA;                                        // partial sums of the two coarse rows under construction (older, younger)

  // vertical expand of one fine row from the window -> s_ve[buf] (lpyr_dec.py:229-232); roll: a new coarse row enters first
  auto coarse_finish = [&](int buf, auto odd_row, bool roll, float4 emitted) {
    if (roll) { cA = cB; cB = cC; cC = emitted; }
    const float m0[4] = {cA.x, cA.y, cA.z, cA.w}, m1[4] = {cB.x, cB.y, cB.z, cB.w}, m2[4] = {cC.x, cC.y, cC.z, cC.w};
    float o[4];
    if constexpr (decltype(odd_row)::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = expand_odd(m1[i], m2[i], eo);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = expand_even(m0[i], m1[i], m2[i], e0, e1);
    }
    s_ve[buf][2 * c][2 + j] = make_float2(o[0], o[1]);           // coarse columns cb+2j, cb+2j+1 of the test plane ...
    s_ve[buf][2 * c + 1][2 + j] = make_float2(o[2], o[3]);       // ... and of the reference plane
  };

  // One level-l row (four own samples + three neighbours per plane) through the horizontal pass, then into the running sums.
  // a_row: its index (scalar).  Returns true and the completed coarse row (a_row/2 - 1) on even rows.
  auto consume = [&](int a_row, auto odd_a, v4f vT, v4f vR, v2f lT, float rT, v2f lR, float rR, float4& emitted) {
    int fcx = fc0;
    if constexpr (RAG) asm volatile("" : "+v"(fcx));   // EDGE == 2 is out of VGPRs: its lane masks and edge weights are re-made from fc0 row by row
    if constexpr (edge_lr) {
      const float mV = (RAG ? (fcx >= 0 && fcx < W) : in_img) ? 1.0f : 0.0f;
      const float mL = (fcx - 2 >= 0 && fcx - 2 < W) ? 1.0f : 0.0f;
      const float mR = (fcx + 4 >= 0 && fcx + 4 < W) ? 1.0f : 0.0f;
      vT *= mV; vR *= mV; lT *= mL; lR *= mL; rT *= mR; rR *= mR;
    }
    float4 hr;
    hr.x = __builtin_fmaf(vT.z, rk4, __builtin_fmaf(vT.y, rk3, __builtin_fmaf(vT.x, rk2, __builtin_fmaf(lT.y, rk1, lT.x * rk0))));
    hr.y = __builtin_fmaf(rT, rk4, __builtin_fmaf(vT.w, rk3, __builtin_fmaf(vT.z, rk2, __builtin_fmaf(vT.y, rk1, vT.x * rk0))));
    hr.z = __builtin_fmaf(vR.z, rk4, __builtin_fmaf(vR.y, rk3, __builtin_fmaf(vR.x, rk2, __builtin_fmaf(lR.y, rk1, lR.x * rk0))));
    hr.w = __builtin_fmaf(rR, rk4, __builtin_fmaf(vR.w, rk3, __builtin_fmaf(vR.z, rk2, __builtin_fmaf(vR.y, rk1, vR.x * rk0))));
    if constexpr (edge_lr) {
      // first / last output column (lpyr_dec.py:205-209; the last column's extra taps depend on the ROW parity, sic)
      const float wl1 = fcx == 0 ? rk1 : 0.0f, wl0 = fcx == 0 ? rk0 : 0.0f;
      const bool last_lane = fcx == W - (RAG ? 2 : 4);              // the lane of columns W-2, W-1
      const float wr3 = last_lane ? ((H & 1) ? rk3 : rk4) : 0.0f, wr2 = (last_lane && (H & 1)) ? rk4 : 0.0f;
      hr.x = __builtin_fmaf(vT.y, wl0, __builtin_fmaf(vT.x, wl1, hr.x));
      hr.z = __builtin_fmaf(vR.y, wl0, __builtin_fmaf(vR.x, wl1, hr.z));
      if constexpr (RAG) {                                          // columns W-2, W-1 are the partial lane's first two: its coarse column X0
        hr.x = __builtin_fmaf(vT.x, wr2, __builtin_fmaf(vT.y, wr3, hr.x));
        hr.z = __builtin_fmaf(vR.x, wr2, __builtin_fmaf(vR.y, wr3, hr.z));
      } else {
        hr.y = __builtin_fmaf(vT.z, wr2, __builtin_fmaf(vT.w, wr3, hr.y));
        hr.w = __builtin_fmaf(vR.z, wr2, __builtin_fmaf(vR.w, wr3, hr.w));
      }
    }
    // Vertical pass.  Rows outside the image are the zero padding (nothing to add); the first / last coarse row's extra taps
    // (lpyr_dec.py:195-199) are added in uniform branches that only the image's first and last two rows take.
    const bool ok = a_row >= 0 && a_row < H;
    const bool border = a_row <= 1 || a_row >= H - 2;               // (scalar)
    auto axpy = [](float4& y, const float4& x, float w) {
      y.x = __builtin_fmaf(x.x, w, y.x); y.y = __builtin_fmaf(x.y, w, y.y); y.z = __builtin_fmaf(x.z, w, y.z); y.w = __builtin_fmaf(x.w, w, y.w);
    };
    if constexpr (decltype(odd_a)::value) {
      if (ok) {
        axpy(rP, hr, rk3);
        axpy(rQ, hr, rk1);
        if (border) {
          if (a_row == 1) axpy(rP, hr, rk0);                        // coarse row 0: + k0 * row 1
          if (!(H & 1) && a_row == H - 1) axpy(rP, hr, rk4);        // H even, last coarse row: + k4 * row H-1
          if ((H & 1) && a_row == H - 2) axpy(rQ, hr, rk4);         // H odd,  last coarse row: + k4 * row H-2
        }
      }
    } else {
      if (ok) {
        axpy(rP, hr, rk4);
        axpy(rQ, hr, rk2);
        if (border) {
          if (a_row == 0) axpy(rQ, hr, rk1);                        // coarse row 0: + k1 * row 0
          if ((H & 1) && a_row == H - 1) axpy(rQ, hr, rk3);         // H odd, last coarse row: + k3 * row H-1
        }
      }
      emitted = rP;
      rP = rQ;
      rQ = ok ? make_float4(hr.x * rk0, hr.y * rk0, hr.z * rk0, hr.w * rk0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      const int m1 = (a_row >> 1) - 1;                             // the coarse row just completed
      if (m1 > Hc - 1) emitted = cC;                               // below the last coarse row: its replica (the expand clamps)
      if constexpr (edge_lr) {                                     // coarse columns outside the image: replicas of column 0 / Wc-1
        const bool rep_left = fc0 < 0, rep_right = fc0 >= W;
        if (strip == 0) {
          const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, emitted.x), lane_first));
          const float r0_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, emitted.z), lane_first));
          if (rep_left) emitted = make_float4(t0, t0, r0_, r0_);
        }
        if (edge_r) {
          const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, RAG ? emitted.x : emitted.y), lane_last));
          const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, RAG ? emitted.z : emitted.w), lane_last));
          if (rep_right) emitted = make_float4(t1, t1, r1, r1);
          if constexpr (RAG) {
            if (part) { emitted.y = t1; emitted.w = r1; }            // the partial lane's second coarse column is column Wc already
          }
        }
      }
      // level l+1 belongs to the lanes that own its columns (interior of the strip) in the segment that owns its rows
      const int own_end = seg == a.n_seg - 1 ? Hc : (ye >> 1);
      if (interior && m1 >= (ys >> 1) && m1 < own_end) {
        const int64_t o = (int64_t)m1 * Wc + (cb + 2 * j);
        if (RAG && part) {
          __builtin_nontemporal_store(emitted.x, g1T + o);
          __builtin_nontemporal_store(emitted.z, g1R + o);
        } else {
          __builtin_nontemporal_store(v2f{emitted.x, emitted.y}, reinterpret_cast<v2f*>(g1T + o));
          __builtin_nontemporal_store(v2f{emitted.z, emitted.w}, reinterpret_cast<v2f*>(g1R + o));
        }
      }
    }
  };

  // horizontal half of the expand for this lane's 4 columns from 4 coarse values (lpyr_dec.py:234-237)
  auto expand4 = [&](const float2* row, float (&ex)[4]) {
    const float2 p0 = row[j + 1];
    const float2 p1 = row[j + 2];
    const float2 p2 = row[j + 3];
    const float A = p0.y, B = p1.x, C = p1.y, D = p2.x;
    ex[0] = expand_even(A, B, C, e0, e1);
    ex[1] = expand_odd(B, C, eo);
    ex[2] = expand_even(B, C, D, e0, e1);
    ex[3] = expand_odd(C, D, eo);
  };

  // per-column luminance terms of one row, shared by all channels (band4.hip lum_prep)
  const bool lodd = t & 1;
  const float lwa = lodd ? 0.0f : e0, lwb = lodd ? eo : e1, lwc = lodd ? eo : e0;
  auto lum_prep = [&](int buf) {
    const float* yT = reinterpret_cast<const float*>(&s_ve[buf][0][0]);
    const float* yR = reinterpret_cast<const float*>(&s_ve[buf][1][0]);
    for (int col = t; col < 256; col += 64 * NCH) {
      const int e = 4 + (col >> 1);
      const float eyT = __builtin_fmaf(yT[e + 1], lwc, __builtin_fmaf(yT[e], lwb, yT[e - 1] * lwa));   // (kernels.h expand_even / expand_odd: the reference's order)
      const float eyR = __builtin_fmaf(yR[e + 1], lwc, __builtin_fmaf(yR[e], lwb, yR[e - 1] * lwa));
      const float Lt = fmaxf(eyT, 0.01f), Lr = fmaxf(eyR, 0.01f);              // lpyr_dec.py:394
      float ind = fast_log2(Lr) * ind_k1 - ind_k0;
      ind = __builtin_amdgcn_fmed3f(ind, 0.0f, (float)(CVVDP_CSF_NODES - 1));  // clamp (interp.py:93)
      const int i0 = (int)ind;
      const float fr = __builtin_amdgcn_fractf(ind);
      s_lum[0][col] = fast_rcp(Lt);
      s_lum[1][col] = fast_rcp(Lr);
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) {                                       // csf.py:49, cvvdp_metric.py:709,:836
        const float2 ln = s_lut[cc][i0];
        s_S[cc][col] = fast_exp2(ln.x + ln.y * fr);
      }
    }
  };

  // vertical-blur window (band4.hip): slot s of column i = one horizontally blurred row, weights rotate instead of the data
  typedef float v32f __attribute__((ext_vector_type(32)));
  v32f winA = 0.0f, winB = 0.0f;
  const v2f E0 = {a.blur_h[0], a.blur_h[1]}, E1 = {a.blur_h[2], a.blur_h[3]}, E2 = {a.blur_h[4], a.blur_h[5]}, E3 = {a.blur_h[6], a.blur_h[5]};
  const v2f O0 = {a.blur_h[1], a.blur_h[2]}, O1 = {a.blur_h[3], a.blur_h[4]}, O2 = {a.blur_h[5], a.blur_h[6]};
#define F_SWAP(p) __builtin_shufflevector(p, p, 1, 0)
  const v2f be[6] = {E0, E1, E2, E3, F_SWAP(O1), F_SWAP(O0)};
  const v2f bo[6] = {O0, O1, O2, F_SWAP(E2), F_SWAP(E1), F_SWAP(E0)};
#undef F_SWAP
  const float b0 = a.blur_h[0], b12 = a.blur_h[0];
  // The top segment starts at row 0 (its six reflected rows are filled in by symmetry): the window position of its first row is 6
  const int r_start = top_seg ? 0 : ys - F_R;
  const int rend = ye + F_R;
  const int n0 = top_seg ? F_R : 0;
  float wr[F_BW];
#pragma unroll
  for (int k = 0; k < F_BW; ++k) wr[k] = a.blur[(k + 2 * F_BW - 1 - n0) % F_BW];
  float acc = 0.0f;

  // ---- FEATURES (band4.hip): two independent trackers -- |T'|, |R'| belong to row r of the contrast stage, D to the pooled row
  // The 24 column sums of a lane live in LDS (lane-private: no barrier): in registers on top of the ring and the blur window the border
  // instantiation spilled 175 VGPRs, and at levels 1, 2 the border strips' kernel is what the level waits for.
  __shared__ __attribute__((aligned(16))) float s_f[FEAT ? 6 : 1][FEAT ? NCH : 1][FEAT ? 256 : 4];   // |T'|, |T'|^2, |R'|, |R'|^2, D, D^2
  const float zero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  int f_left_tr = 0, f_left_d = 0;                  // rows to the next cell-row boundary (scalar)
  if constexpr (FEAT) {
    f_left_tr = f_left_d = a.fs - ys % a.fs;
#pragma unroll
    for (int q = 0; q < 6; ++q) f_lds_write4(&s_f[q][c][4 * j], zero4);
  }
  auto feat_store = [&](int y_last, int q0, const float (&s0)[4], const float (&s1)[4]) {   // column sums of the piece that ends with row y_last
    if constexpr (FEAT) {
      if (interior) {
        const int piece = y_last / a.fs + seg;
        const int n_valid = part ? 2 : 4;
        float* dst = a.fsum + ((((int64_t)item * NCH + c) * a.f_pieces + piece) * 6 + q0) * W + fc0;
        if (n_valid == 4) {
          *reinterpret_cast<f_u4*>(dst) = f_u4{s0[0], s0[1], s0[2], s0[3]};
          *reinterpret_cast<f_u4*>(dst + W) = f_u4{s1[0], s1[1], s1[2], s1[3]};
        } else {
          for (int i = 0; i < n_valid; ++i) { dst[i] = s0[i]; dst[W + i] = s1[i]; }
        }
      }
    }
  };
  auto feat_flush = [&](int y_last, int q0) {       // sums q0, q0+1 of the piece that ends with row y_last -> global, and back to zero
    if constexpr (FEAT) {
      const ff4 s0 = f_lds_read4(&s_f[q0][c][4 * j]), s1 = f_lds_read4(&s_f[q0 + 1][c][4 * j]);
      feat_store(y_last, q0, s0.v, s1.v);
      f_lds_write4(&s_f[q0][c][4 * j], zero4);
      f_lds_write4(&s_f[q0 + 1][c][4 * j], zero4);
    }
  };
  auto feat_d_row = [&](int yprev) {                // D sums: a cell row ends with row yprev (the segment's last row is the epilogue's)
    if constexpr (FEAT) {
      if (yprev >= ys && --f_left_d == 0) {
        feat_flush(yprev, 4);
        f_left_d = a.fs;
      }
    }
  };
  // pooling stage of centre row y (band4.hip stage3c)
  auto stage3c = [&](int k7) {
    const ff4 q0 = f_lds_read4(&s_q[0][4 * j]), q1 = f_lds_read4(&s_q[1][4 * j]), q2 = f_lds_read4(&s_q[2][4 * j]);
    ff4 q3 = ff4{{0.0f, 0.0f, 0.0f, 0.0f}};
    if constexpr (NCH == 4) q3 = f_lds_read4(&s_q[3][4 * j]);
    const ff4 d = f_lds_read4(&s_d[k7][c][4 * j - F_HALO]);
    float De[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f Q0 = {q0.v[2 * h], q0.v[2 * h + 1]}, Q1 = {q1.v[2 * h], q1.v[2 * h + 1]}, Q2 = {q2.v[2 * h], q2.v[2 * h + 1]}, Q3 = {q3.v[2 * h], q3.v[2 * h + 1]};
      const v2f M1 = Q3 * xw3 + (Q2 * xw2 + (Q1 * xw1 + (Q0 * xw0 + m1c)));
      const v2f X = {fast_pow(d.v[2 * h], mask_p) - eps_p, fast_pow(d.v[2 * h + 1], mask_p) - eps_p};
      const v2f T = X * inv_dmax + M1;
      const float r0 = fast_rcp(T.x), r1 = fast_rcp(T.y);
      De[2 * h] = __builtin_fmaf(X.x, r0, kEps); De[2 * h + 1] = __builtin_fmaf(X.y, r1, kEps);
      if constexpr (HEAT) {   // this channel's term of the per-pixel channel norm (cvvdp_metric.py:728-734; band4.hip stage3c)
        s_h[c][4 * j + 2 * h] = fast_pow((X.x * r0) * a.hw[c] + kEps, a.beta_tch) - a.eps_btch;
        s_h[c][4 * j + 2 * h + 1] = fast_pow((X.y * r1) * a.hw[c] + kEps, a.beta_tch) - a.eps_btch;
      }
      if constexpr (FEAT) {
        const float D0 = X.x * r0, D1 = X.y * r1;
        float2* pd = reinterpret_cast<float2*>(&s_f[4][c][4 * j + 2 * h]);
        float2* pd2 = reinterpret_cast<float2*>(&s_f[5][c][4 * j + 2 * h]);
        float2 fd = *pd, fd2 = *pd2;
        fd.x += D0; fd2.x = __builtin_fmaf(D0, D0, fd2.x);
        fd.y += D1; fd2.y = __builtin_fmaf(D1, D1, fd2.y);
        *pd = fd; *pd2 = fd2;
      }
    }
    if constexpr (RAG) {
      if (part) { De[2] = 0.0f; De[3] = 0.0f; }                        // columns right of the image do not exist: no term
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_fmaf(De[i], De[i], acc);   // sum of (D + eps)^2; k_finalize takes the eps^2 off
  };
  // heat-map band of row y from the channel terms published by stage3c (a barrier in between): one column per thread, lp_norm over
  // the channels, stored / band_mul as lpyr_dec_2.set_lband does (lpyr_dec.py:308-314; band4.hip heat_row)
  auto heat_row = [&](int y) {
    if constexpr (HEAT) {
      const int xs = x0 - F_HALO + t;
      if (y >= ys && t >= F_HALO && t < 256 - F_HALO && xs < W) {
        float sum = s_h[0][t] + s_h[1][t] + s_h[2][t];
        if constexpr (NCH == 4) sum += s_h[3][t];
        a.dchr[(int64_t)item * P + (int64_t)y * W + xs] = (fast_pow(sum + kEps, 1.0f / a.beta_tch) - a.eps_inv_btch) / a.band_mul;
      }
    }
  };

  // ---- STREAM LOADS (see band4.hip): issued from inline assembly, waited for with one exact count per step.
  //   phase 2 of step s:  neighbour samples of row s+6 (4 loads: left pair + right sample, two planes) -> set (s+6)&1,
  //                       rows s+7 of the two planes (2 loads) -> ring slot (s+7)&7
  //   start of step s:    needs ring slot (s+5)&7 and neighbour set (s+5)&1; younger: the 2 row loads of step s-1 -> vmcnt(2)
  v4f ringT[8], ringR[8];
  v2f nbLT[2], nbLR[2];
  float nbRT[2], nbRR[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ringT[i] = 0.0f; ringR[i] = 0.0f; }
#pragma unroll
  for (int i = 0; i < 2; ++i) { nbLT[i] = 0.0f; nbLR[i] = 0.0f; nbRT[i] = 0.0f; nbRR[i] = 0.0f; }
  // F_SAFE (`make safe`, and the HEAT instantiations): ordinary loads the compiler tracks and waits for itself
#define F_ROWPTR(plane, row, off) (reinterpret_cast<const char*>((plane) + (int64_t)(row) * W) + (off))
#define F_LOAD4(dst, off, plane, row) do { if constexpr (F_SAFE) { const f_u4 q_ = *reinterpret_cast<const f_u4*>(F_ROWPTR(plane, row, off)); dst = v4f{q_.x, q_.y, q_.z, q_.w}; } \
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(off), "s"((plane) + (int64_t)(row) * W)); } while (0)
#define F_LOAD2(dst, off, plane, row) do { if constexpr (F_SAFE) { const f_u2 q_ = *reinterpret_cast<const f_u2*>(F_ROWPTR(plane, row, off)); dst = v2f{q_.x, q_.y}; } \
    else asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(dst) : "v"(off), "s"((plane) + (int64_t)(row) * W)); } while (0)
#define F_LOAD1(dst, off, plane, row) do { if constexpr (F_SAFE) dst = *reinterpret_cast<const float*>(F_ROWPTR(plane, row, off)); \
    else asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(off), "s"((plane) + (int64_t)(row) * W)); } while (0)
#define F_WAIT2(a0, a1, a2, a3, a4, a5) do { if constexpr (!F_SAFE) asm volatile("s_waitcnt vmcnt(2)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5)); } while (0)
#define F_DRAIN() do { if constexpr (!F_SAFE) __builtin_amdgcn_s_waitcnt(0x0F70); } while (0)
  auto rowc = [&](int r) { return min(max(r, 0), H - 1); };       // rows outside the image: any valid row (their weight is 0)

  // image-edge mirror roles of the contrast stage (band4.hip): reflect padding of the blur at the left / right image border
  const bool mir_block = EDGE != 0 && (strip == 0 || edge_r);
  int mir_kind = 0, mir_base = 0;
  if (strip == 0 && (fc0 == 0 || fc0 == 4)) { mir_kind = fc0 == 0 ? 1 : 2; mir_base = F_HALO - (fc0 == 0 ? 1 : 4); }
  if (!RAG && (fc0 == W - 8 || fc0 == W - 4)) {
    const int base = 2 * (W - 1) - (fc0 + (fc0 == W - 8 ? 1 : 0)) - (x0 - F_HALO);
    if (base < 256) { mir_kind = fc0 == W - 8 ? 1 : 2; mir_base = base; }
  }

  // HEAT, colour-mapped modes: the range of the context image from the rows the wave of channel 0 streams (band4s.hip range_row)
  uint32_t r_mn = 0x7F7FFFFFu, r_mx = 0u;
  auto range_row = [&](v4f v) {
    if constexpr (HEAT) {
      if (c == 0 && a.hstats) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t u = __float_as_uint(v[i]);
          r_mn = min(r_mn, v[i] > 0.0f ? u : 0x7F7FFFFFu);
          r_mx = max(r_mx, v[i] > 0.0f ? u : 0u);
        }
      }
    }
  };
  // ---- prologue: rows r_start-4 .. r_start+4 prime the reduce (three complete coarse rows in the window, two partial ones),
  // rows r_start .. r_start+4 stay in ring slots 0 .. 4, the rows after them are requested
  {
    auto ld4 = [&](const float* plane, int r) -> v4f {
      return *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(plane + (int64_t)rowc(r) * W) + goff);
    };
    auto placed = [&](v4f q) -> v4f { return part ? v4f{q.z, q.w, 0.0f, 0.0f} : q; };   // (part is false at compile time unless EDGE == 2)
    auto ld2 = [&](const float* plane, int r) -> v2f { return *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(plane + (int64_t)rowc(r) * W) + loff); };
    auto ld1 = [&](const float* plane, int r) -> float { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(plane + (int64_t)rowc(r) * W) + roff); };
    float4 em = cC;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int ar = r_start - 4 + i;                               // r_start is even: i even <-> row even
      const v4f vT = ld4(gT, ar), vR = ld4(gR, ar);
      const v2f lT = ld2(gT, ar), lR = ld2(gR, ar);
      const float rT = ld1(gT, ar), rR = ld1(gR, ar);
      if (i >= 4) { ringT[i - 4] = vT; ringR[i - 4] = vR; }          // (the ring holds the rows as loaded)
      range_row(vT);
      if (i & 1) {
        consume(ar, std::true_type{}, placed(vT), placed(vR), lT, rT, lR, rR, em);
      } else {
        consume(ar, std::false_type{}, placed(vT), placed(vR), lT, rT, lR, rR, em);
        cA = cB; cB = cC; cC = em;
      }
    }
    if (r_start == 0) cA = cB;                                      // coarse row -1 does not exist: the expand clamps to row 0
    F_LOAD4(ringT[5], goff, gT, rowc(r_start + 5));
    F_LOAD4(ringR[5], goff, gR, rowc(r_start + 5));
    F_LOAD2(nbLT[1], loff, gT, rowc(r_start + 5));                  // (r_start + 5) & 1 == 1
    F_LOAD1(nbRT[1], roff, gT, rowc(r_start + 5));
    F_LOAD2(nbLR[1], loff, gR, rowc(r_start + 5));
    F_LOAD1(nbRR[1], roff, gR, rowc(r_start + 5));
    F_LOAD4(ringT[6], goff, gT, rowc(r_start + 6));
    F_LOAD4(ringR[6], goff, gR, rowc(r_start + 6));
    coarse_finish(0, std::false_type{}, false, cC);                 // vertical expand of row r_start (even)
  }
  __syncthreads();
  lum_prep(0);
  __syncthreads();

  int slot = n0, k7 = 0;
  // one real row r (0 <= r < H); U = (r - r_start) mod 8 = its ring slot
  auto step = [&](int r, auto u_) {
    constexpr int U = decltype(u_)::value;
    constexpr bool ODD = (U & 1) != 0;
    (void)&ringT; (void)&ringR; (void)&nbLT; (void)&nbLR; (void)&nbRT; (void)&nbRR; (void)&goff; (void)&loff; (void)&roff; (void)&gT; (void)&gR; (void)&W;
    F_WAIT2(ringT[(U + 5) & 7], ringR[(U + 5) & 7], nbLT[(U + 5) & 1], nbLR[(U + 5) & 1], nbRT[(U + 5) & 1], nbRR[(U + 5) & 1]);
    // (EDGE == 2: the partial lane's rows are shifted into place where they are USED, twice per row -- a ring register that is
    // rewritten stops being pinned, and the compiler then copies ring registers around while their loads are in flight)
    auto placed = [&](v4f q) -> v4f { return part ? v4f{q.z, q.w, 0.0f, 0.0f} : q; };
    const v4f pT = placed(ringT[U]), pR = placed(ringR[U]);
    // ================= phase 1
    const int yprev = r - 1 - F_R;
    if (interior && yprev >= ys) stage3c(k7);
    feat_d_row(yprev);
    const bool feat_row = FEAT && r >= ys && r < ye;  // (scalar) row r belongs to this segment: its |T'|, |R'| are counted
    if (in_img) {
      float exT[4], exR[4];
      expand4(s_ve[ODD][2 * c], exT);
      expand4(s_ve[ODD][2 * c + 1], exR);
      const ff4 rLt = f_lds_read4(&s_lum[0][4 * j]), rLr = f_lds_read4(&s_lum[1][4 * j]);
      const ff4 Sv = f_lds_read4(&s_S[c][4 * j]);
      const float gt[4] = {pT.x, pT.y, pT.z, pT.w}, gr[4] = {pR.x, pR.y, pR.z, pR.w};
      float m[4], d[4];
      ff4 f_t = ff4{{0, 0, 0, 0}}, f_t2 = f_t, f_r = f_t, f_r2 = f_t;
      if constexpr (FEAT) {
        if (feat_row) { f_t = f_lds_read4(&s_f[0][c][4 * j]); f_t2 = f_lds_read4(&s_f[1][c][4 * j]); f_r = f_lds_read4(&s_f[2][c][4 * j]); f_r2 = f_lds_read4(&s_f[3][c][4 * j]); }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float S = Sv.v[i];
        const float ct = fminf((gt[i] - exT[i]) * rLt.v[i], 1000.0f);            // lpyr_dec.py:402 (band gain :66 is in S)
        const float cr = fminf((gr[i] - exR[i]) * rLr.v[i], 1000.0f);
        if constexpr (FEAT) {
          const float at = fabsf(ct) * S, ar = fabsf(cr) * S;                    // |T'|, |R'| (the channel gain inside S is divided out by k_feature_finish)
          m[i] = fminf(at, ar);                                                  // = min(|ct|,|cr|)*S bit for bit (rounding is monotone)
          if (feat_row) { f_t.v[i] += at; f_t2.v[i] = __builtin_fmaf(at, at, f_t2.v[i]); f_r.v[i] += ar; f_r2.v[i] = __builtin_fmaf(ar, ar, f_r2.v[i]); }
        } else {
          m[i] = fminf(fabsf(ct), fabsf(cr)) * S;                                // min(|T'|,|R'|), T' = ct*S (cvvdp_metric.py:845)
        }
        d[i] = fabsf(ct - cr) * S + kEps;                                        // |T'-R'| + eps (:855, safe_pow)
      }
      f_lds_write4(&s_m[c][4 * j], m);
      if (interior) f_lds_write4(&s_d[k7][c][4 * j - F_HALO], d);
      if constexpr (FEAT) {
        if (feat_row) { f_lds_write4(&s_f[0][c][4 * j], f_t.v); f_lds_write4(&s_f[1][c][4 * j], f_t2.v); f_lds_write4(&s_f[2][c][4 * j], f_r.v); f_lds_write4(&s_f[3][c][4 * j], f_r2.v); }
      }
      if (mir_block) {
        if (mir_kind != 0) {
          const float v0 = mir_kind == 1 ? m[1] : m[0], v1 = mir_kind == 1 ? m[2] : m[1], v2 = mir_kind == 1 ? m[3] : m[2];
          float* dst = &s_m[c][mir_base];
          dst[0] = v0; dst[-1] = v1; dst[-2] = v2;
        }
        if constexpr (RAG) {                              // columns W-7 .. W-2, wherever they fall in the lanes, to 2(W-1)-x (band4.hip, mir_any)
          if (edge_r) {
            // (one address register and four immediate offsets: element idx - i of column fc0 + i is p3[3 - i])
            const int idx0 = 2 * (W - 1) - fc0 - (x0 - F_HALO);
            float* p3 = &s_m[c][0] + (idx0 - 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int x = fc0 + i;
              if (x >= W - 1 - F_R && x <= W - 2 && idx0 - i < 256) p3[3 - i] = m[i];
            }
          }
        }
      }
    }
    if constexpr (FEAT) {
      if (feat_row && (--f_left_tr == 0 || r == ye - 1)) {
        feat_flush(r, 0);
        feat_flush(r, 2);
        f_left_tr = a.fs;
      }
    }
    // level-l row r+5 into the reduce; an even row completes a coarse row, which rolls the window for row r+1 (even) below
    float4 emitted = cC;
    consume(r + 5, std::integral_constant<bool, !ODD>{}, placed(ringT[(U + 5) & 7]), placed(ringR[(U + 5) & 7]), nbLT[(U + 5) & 1], nbRT[(U + 5) & 1],
            nbLR[(U + 5) & 1], nbRR[(U + 5) & 1], emitted);
    range_row(ringT[(U + 5) & 7]);
    coarse_finish(ODD ? 0 : 1, std::integral_constant<bool, !ODD>{}, ODD, emitted);   // vertical expand of row r+1
    __syncthreads();
    // ================= phase 2
    heat_row(yprev);
    {
      const int r6 = rowc(r + 6), r7 = rowc(r + 7);
      F_LOAD2(nbLT[U & 1], loff, gT, r6);                           // (r + 6) & 1 == U & 1
      F_LOAD1(nbRT[U & 1], roff, gT, r6);
      F_LOAD2(nbLR[U & 1], loff, gR, r6);
      F_LOAD1(nbRR[U & 1], roff, gR, r6);
      F_LOAD4(ringT[(U + 7) & 7], goff, gT, r7);
      F_LOAD4(ringR[(U + 7) & 7], goff, gR, r7);
    }
    lum_prep(ODD ? 0 : 1);
    const int yc = r - F_R;
    if (interior) {
      const v4f* row = reinterpret_cast<const v4f*>(&s_m[c][4 * j - 8]);
      const v4f a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3], a4 = row[4];
      const v2f xp[10] = {a0.xy, a0.zw, a1.xy, a1.zw, a2.xy, a2.zw, a3.xy, a3.zw, a4.xy, a4.zw};
      float h[4];
      {
        v2f s0 = be[0] * xp[1], s1 = bo[0] * xp[2], s2 = be[0] * xp[2], s3 = bo[0] * xp[3];
#pragma unroll
        for (int mm = 1; mm < 6; ++mm) {
          s0 += be[mm] * xp[1 + mm]; s1 += bo[mm] * xp[2 + mm]; s2 += be[mm] * xp[2 + mm]; s3 += bo[mm] * xp[3 + mm];
        }
        h[0] = (s0.x + b12 * xp[7].x) + s0.y;
        h[1] = (s1.x + b0 * xp[1].y) + s1.y;
        h[2] = (s2.x + b12 * xp[8].x) + s2.y;
        h[3] = (s3.x + b0 * xp[2].y) + s3.y;
      }
      winA[2 * slot] = h[0]; winA[2 * slot + 1] = h[1]; winB[2 * slot] = h[2]; winB[2 * slot + 1] = h[3];
      if (top_seg && r >= 1 && r <= F_R) {              // rows 1..6 of the image are also its reflected rows -1..-6 (window slots 5..0)
        const int ms = F_R - r;
        winA[2 * ms] = h[0]; winA[2 * ms + 1] = h[1]; winB[2 * ms] = h[2]; winB[2 * ms + 1] = h[3];
      }
      if (yc >= ys) {
        v2f va = kEps, vb = kEps;
#pragma unroll
        for (int sdx = 0; sdx < F_BW; ++sdx) {
          const v2f wa = {winA[2 * sdx], winA[2 * sdx + 1]}, wb = {winB[2 * sdx], winB[2 * sdx + 1]};
          const v2f ww = {wr[sdx], wr[sdx]};
          va += ww * wa; vb += ww * wb;
        }
        const float v[4] = {va.x, va.y, vb.x, vb.y};
        float Mq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) Mq[i] = fast_pow(v[i], qc);
        f_lds_write4(&s_q[c][4 * j], Mq);
      }
    }
    {
      const float last = wr[F_BW - 1];
#pragma unroll
      for (int k = F_BW - 1; k > 0; --k) wr[k] = wr[k - 1];
      wr[0] = last;
    }
    slot = slot == F_BW - 1 ? 0 : slot + 1;
    k7 = k7 == F_R ? 0 : k7 + 1;
    __syncthreads();
  };
  // one reflected row below the image (r >= H): its horizontally blurred row is that of row 2(H-1) - r, still in the window
  auto tail_step = [&](int r) {
    const int yprev = r - 1 - F_R;
    if (interior && yprev >= ys) stage3c(k7);
    feat_d_row(yprev);
    __syncthreads();
    heat_row(yprev);
    const int yc = r - F_R;
    if (interior) {
      const int back = 2 * (r - (H - 1));                           // 2, 4, .. 12 rows back
      const int src = slot >= back ? slot - back : slot - back + F_BW;
      const float h0 = winA[2 * src], h1 = winA[2 * src + 1], h2 = winB[2 * src], h3 = winB[2 * src + 1];
      winA[2 * slot] = h0; winA[2 * slot + 1] = h1; winB[2 * slot] = h2; winB[2 * slot + 1] = h3;
      if (yc >= ys) {
        v2f va = kEps, vb = kEps;
#pragma unroll
        for (int sdx = 0; sdx < F_BW; ++sdx) {
          const v2f wa = {winA[2 * sdx], winA[2 * sdx + 1]}, wb = {winB[2 * sdx], winB[2 * sdx + 1]};
          const v2f ww = {wr[sdx], wr[sdx]};
          va += ww * wa; vb += ww * wb;
        }
        const float v[4] = {va.x, va.y, vb.x, vb.y};
        float Mq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) Mq[i] = fast_pow(v[i], qc);
        f_lds_write4(&s_q[c][4 * j], Mq);
      }
    }
    {
      const float last = wr[F_BW - 1];
#pragma unroll
      for (int k = F_BW - 1; k > 0; --k) wr[k] = wr[k - 1];
      wr[0] = last;
    }
    slot = slot == F_BW - 1 ? 0 : slot + 1;
    k7 = k7 == F_R ? 0 : k7 + 1;
    __syncthreads();
  };

  const int rreal = min(rend, H);
  int r = r_start;
  for (; r < rreal; r += 8) {
    step(r, std::integral_constant<int, 0>{});
    if (r + 1 >= rreal) break;
    step(r + 1, std::integral_constant<int, 1>{});
    if (r + 2 >= rreal) break;
    step(r + 2, std::integral_constant<int, 2>{});
    if (r + 3 >= rreal) break;
    step(r + 3, std::integral_constant<int, 3>{});
    if (r + 4 >= rreal) break;
    step(r + 4, std::integral_constant<int, 4>{});
    if (r + 5 >= rreal) break;
    step(r + 5, std::integral_constant<int, 5>{});
    if (r + 6 >= rreal) break;
    step(r + 6, std::integral_constant<int, 6>{});
    if (r + 7 >= rreal) break;
    step(r + 7, std::integral_constant<int, 7>{});
  }
  F_DRAIN();
  for (r = rreal; r < rend; ++r) tail_step(r);
  // ---- epilogue: pooling stage of the last centre row
  if (interior && (ye - 1) >= ys) stage3c(k7);
  if constexpr (FEAT) {
    if ((ye - 1) >= ys) feat_flush(ye - 1, 4);
  }
  if constexpr (HEAT) {
    __syncthreads();
    heat_row(ye - 1);
    if (c == 0 && a.hstats) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        r_mn = min(r_mn, (uint32_t)__shfl_down((int)r_mn, off, 64));
        r_mx = max(r_mx, (uint32_t)__shfl_down((int)r_mx, off, 64));
      }
      if (j == 0) {
        atomicMin(&a.hstats[(int64_t)item * kHeatStatsWords + 0], r_mn);
        atomicMax(&a.hstats[(int64_t)item * kHeatStatsWords + 1], r_mx);
      }
    }
  }

#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (j == 0) {
    const int nblk = a.n_strip * a.n_seg;
    a.partial[((int64_t)item * nblk + (seg * a.n_strip + strip)) * 4 + c] = acc;
  }
}

#undef F_LOAD4
#undef F_LOAD2
#undef F_LOAD1
#undef F_DRAIN
#undef F_WAIT2
#undef F_ROWPTR

template <int NCH, int EDGE>
__global__ __launch_bounds__(64 * NCH, 2) void k_band4f(BandArgs a) { band4f_body<NCH, EDGE, false, false>(a); }
template <int NCH, int EDGE>
__global__ __launch_bounds__(64 * NCH, 2) void k_band4f_heat(BandArgs a) { band4f_body<NCH, EDGE, true, false>(a); }
template <int NCH, int EDGE>
__global__ __launch_bounds__(64 * NCH, 2) void k_band4f_feat(BandArgs a) { band4f_body<NCH, EDGE, false, true>(a); }

bool band4f_supported(int H, int W) { return (W & 1) == 0 && W >= 32 && H >= 32; }

// strips whose 256 columns x0-8 .. x0+247 reach the right image border (trailing; at least the last one).  Reaching column W-1 is
// enough (round 5: >=, it was >): the lane of columns W-4 .. W-1 computes the LAST coarse column, whose extra taps (lpyr_dec.py:205-209)
// and zero right neighbour only the border body applies -- and when that lane is a strip's lane 63 (W = 240 k + 248: 488, 728, 968, ..),
// its coarse column still feeds the expand of columns W-4, W-3, which the blur of the strip's last interior columns reads.
static int band4f_right_edge_strips(int W, int n_strip) {
  int n = 0;
  while (n < n_strip && (n_strip - 1 - n) * F_SW + F_SW + F_HALO >= W) ++n;
  return n;
}

void launch_band4f(const BandArgs& a0, hipStream_t s, hipStream_t s_edge) {
  BandArgs a = a0;
  const int n_edge = std::min(a.n_strip, 1 + band4f_right_edge_strips(a.W, a.n_strip));   // strip 0 + the right-edge strips
  a.strip0 = 0; a.n_strip_l = n_edge;
  a.per_xcd = (a.n_strip_l * a.n_seg * a.items + 7) / 8;
  const bool heat = a.dchr != nullptr, feat = a.fsum != nullptr;
  // the border strips: front / back waves too (band4s.hip EDGE) where that body exists -- W % 4 == 0, not the features clips
  if (!feat && !a.one_wave_layout && (a.W & 3) == 0) launch_band4s_edge(a, s_edge);
  else if (feat) {
    if ((a.W & 3) == 0) hipLaunchKernelGGL((k_band4f_feat<4, 1>), dim3(8 * a.per_xcd), dim3(256), 0, s_edge, a);
    else hipLaunchKernelGGL((k_band4f_feat<4, 2>), dim3(8 * a.per_xcd), dim3(256), 0, s_edge, a);
  } else if (heat) {
    if ((a.W & 3) == 0) hipLaunchKernelGGL((k_band4f_heat<4, 1>), dim3(8 * a.per_xcd), dim3(256), 0, s_edge, a);
    else hipLaunchKernelGGL((k_band4f_heat<4, 2>), dim3(8 * a.per_xcd), dim3(256), 0, s_edge, a);
  } else {
    if ((a.W & 3) == 0) hipLaunchKernelGGL((k_band4f<4, 1>), dim3(8 * a.per_xcd), dim3(256), 0, s_edge, a);     // video only (core.cpp)
    else hipLaunchKernelGGL((k_band4f<4, 2>), dim3(8 * a.per_xcd), dim3(256), 0, s_edge, a);
  }
  if (n_edge < a.n_strip) {
    a.strip0 = 1; a.n_strip_l = a.n_strip - n_edge;
    a.per_xcd = (a.n_strip_l * a.n_seg * a.items + 7) / 8;
    if (a.one_wave_layout && feat) hipLaunchKernelGGL((k_band4f_feat<4, 0>), dim3(8 * a.per_xcd), dim3(256), 0, s, a);
    else if (a.one_wave_layout && heat) hipLaunchKernelGGL((k_band4f_heat<4, 0>), dim3(8 * a.per_xcd), dim3(256), 0, s, a);
    else if (a.one_wave_layout) hipLaunchKernelGGL((k_band4f<4, 0>), dim3(8 * a.per_xcd), dim3(256), 0, s, a);
    else launch_band4s(a, s);
  }
}

int tu_flags_band4f() {
  int f = 0;
#ifdef CVVDP_SAFE_LOADS
  f |= CVVDP_BUILD_SAFE_LOADS;
#endif
  return f;
}

}  // namespace cvvdp
