// K3..K7 fused, vectorised variant for every level of at least 16 x 16 pixels: same arithmetic as band.hip, laid out
// for CDNA4 memory instructions.
//
//   wave  <-> colour/temporal channel c (so q_c, the cross-channel weights into c and the CSF row of c
//             are wave-uniform scalars)
//   lane  <-> 4 adjacent columns        (16-byte global loads of g, ds_read_b128/ds_write_b128 for every
//             per-row exchange, 4 independent SFU chains per lane)
//   block <-> strip of 240 interior columns (+8 aligned halo columns each side), marching down a row
//             segment exactly like k_band; blocks are ordered so that one XCD works on neighbouring strips.
//
// Two barriers per row; every row index is wave-uniform (SALU address arithmetic), the loop is unrolled over
// an even/odd row pair so that the expand's row parity and the two g-row register sets are static:
//
//   phase 1:  pooling stage of the row finished last iteration (reads s_q of all channels, s_d)
//             contrast stage of row r (reads s_ve, s_lum, s_S; writes s_m and the s_d ring)
//             vertical expand of row r+1 from the rolling 3-row coarse window -> s_ve
//   barrier
//   phase 2:  loads: the g rows of row r+2 (into the registers row r just left: two static register sets, no
//             copies) and, on odd rows, ONE coarse row (the window's next row); luminance terms and CSF
//             sensitivities of row r+1 -> s_lum, s_S; 13-tap horizontal blur (5 ds_read_b128), 13-row register
//             window, vertical blur, Mq = (blur*10^mask_c + eps)^q_c -> s_q
//   barrier
//
// The streamed loads are issued from inline assembly and waited for with explicit vmcnt counts a whole row later
// (see STREAM LOADS below); tools/check_band4_isa.py checks the generated code.  Rows reflected at the image's top /
// bottom edge run in separate copies of the loop that reload the coarse window.
//
// Image-edge halo columns are produced by the in-image lanes as mirrored LDS writes (reflect padding
// of the blur), halo rows by evaluating the reflected row.  Instantiations: NCH (3 image / 4 video channels), HEAT
// (per-pixel heat-map band), RAGGED (W % 8 != 0: partial last lane, shifted coarse chunks), DUMP (per-pixel D), FEAT (the
// statistics of the ML heads' feature pooling, accumulated in the row march: see FEATURES below).
#include <type_traits>
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
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };   // 16 bytes at 4-byte alignment (rows of any width)

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

template <int NCH, bool HEAT, bool RAGGED, bool DUMP, bool FEAT>
__global__ __launch_bounds__(64 * NCH, FEAT ? 2 : 3) void k_band4(BandArgs a) {
  constexpr int NP = 2 * NCH;
  // s_ve is a ring of two rows (row parity): row r+1 is written during phase 1 of row r, so that phase 2 can derive
  // the per-column luminance terms of row r+1 (s_lum) from its luminance planes (0, 1).
  __shared__ __attribute__((aligned(16))) float2 s_ve[2][NP][B4_VE / 2];   // float2 rows: guaranteed 8-byte aligned ds_read_b64
  // CSF sensitivity: published per channel and column by the luminance stage (s_S), except in the heat-map variant, whose
  // LDS budget (3 blocks per CU: 53 KB) has no room for it: there the luminance stage publishes the LUT position
  // (fraction, byte offset) and every channel wave does its own lerp + exp2
  constexpr bool S_SHARED = !HEAT;
  __shared__ __attribute__((aligned(16))) float s_lum[S_SHARED ? 2 : 4][256];   // 1/L_T, 1/L_R [, LUT fraction, LUT byte offset]
  __shared__ __attribute__((aligned(16))) float s_S[S_SHARED ? NCH : 1][S_SHARED ? 256 : 4];
  __shared__ __attribute__((aligned(16))) float s_m[NCH][256];
  __shared__ __attribute__((aligned(16))) float s_q[NCH][256];
  __shared__ __attribute__((aligned(16))) float s_d[B4_R + 1][NCH][B4_SW];   // lane-private ring of |T'-R'| + eps: interior columns only
  __shared__ __attribute__((aligned(16))) float s_h[HEAT ? NCH : 1][HEAT ? 256 : 4];   // heat-map terms of the pooled row, per channel
  __shared__ __attribute__((aligned(8))) float2 s_lut[NCH][CVVDP_CSF_NODES];           // (node value, step to the next node), log2 domain

  const int t = threadIdx.x;
  const int c = __builtin_amdgcn_readfirstlane(t >> 6);   // channel = wave: a scalar, so per-channel constants live in SGPRs
  const int j = t & 63;
  // XCD-aware block order.  Workgroups are dealt to the 8 XCDs round-robin by launch index, and each XCD has its own L2.
  // Work unit w = (item, seg, strip) with the strip fastest: launch index b is mapped to w = (b % 8) * per_xcd + b / 8, so the
  // blocks resident on one XCD at any time are CONSECUTIVE work units, i.e. neighbouring strips of the same rows, whose
  // overlapping halo columns and shared 128-byte lines are then fetched from HBM once instead of once per strip.
  const int per_xcd = a.per_xcd;
  const int wu = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (wu >= a.n_strip_l * a.n_seg * a.items) return;    // (block-uniform: the grid is rounded up to 8 * per_xcd)
  const int strip = a.strip0 + wu % a.n_strip_l, seg = (wu / a.n_strip_l) % a.n_seg, item = wu / (a.n_strip_l * a.n_seg);
  const int H = a.H, W = a.W, Hc = a.Hc, Wc = a.Wc;
  const int x0 = strip * B4_SW;
  const int fc0 = x0 - B4_HALO + 4 * j;             // first of this lane's 4 columns
  const bool in_img = fc0 >= 0 && fc0 < W;          // at least the first of the four columns is inside the image
  // Ragged right edge (W % 4 != 0, any parity): one lane of the last strip holds 1..3 valid columns.  Its 16-byte load is
  // clamped to the last four columns of the row and shifted into place; its invalid columns carry finite garbage that the
  // reflect padding overwrites (s_m) or that is masked out (pooling, stores).  Everything ragged sits behind block-uniform
  // branches: full strips and W % 4 == 0 run the same instructions as before.
  // RAGGED is a separate instantiation (launch_band4 picks it when W % 8 != 0): the aligned kernel keeps its register budget.
  // (the strip BEFORE the last one sees the right edge too when the last strip is narrower than the halo: its halo lanes then
  // hold the partial lane and need the mirrored columns)
  const bool edge_r = x0 + B4_SW + B4_HALO > W;      // block-uniform: this strip's columns x0-8 .. x0+247 reach past column W-1
  const bool ragged_blk = RAGGED && (W & 3) != 0 && edge_r;
  const int g_shift = RAGGED && in_img ? max(fc0 - (W - 4), 0) : 0;      // 1..3 in the partial lane: loaded element i+g_shift is column fc0+i
  const int n_valid = RAGGED ? (in_img ? min(W - fc0, 4) : 0) : 4;
  const bool interior = j >= 2 && j < 62 && fc0 < W;  // columns whose result is pooled
  const int cb = (x0 - B4_HALO) / 2;                // coarse column of s_ve[.][1]
  const int ys = seg * a.seg_h, ye = min(H, ys + a.seg_h);   // ys is even (core.cpp keeps seg_h even)

  const int64_t P = (int64_t)H * W, Pc = (int64_t)Hc * Wc;
  const int64_t gps = (int64_t)a.items_cap * P, gcps = (int64_t)a.items_cap_c * Pc;
  const float* gT = a.g + (int64_t)item * P + (2 * c) * gps;      // test plane of this channel (scalar base)
  const float* gR = gT + gps;                                      // reference plane
  // stage-1 role of this lane: plane 2c + (j>>5), coarse chunk j&31
  const int vp = 2 * c + (j >> 5);
  const int vch = j & 31;
  const int vcx = cb + 4 * vch;                                    // first coarse column of the chunk
  const float* gcp = a.gc + (int64_t)item * Pc + (2 * c) * gcps;   // scalar base; the lane's plane / column go into the vector offset

  for (int i = t; i < NCH * CVVDP_CSF_NODES; i += 64 * NCH) {
    const int cc = i / CVVDP_CSF_NODES, k = i - cc * CVVDP_CSF_NODES;
    // log2-domain CSF row with the constant gains folded in: S*ch_gain*band_mul = 2^(lut*log2(10) + log2(sens_mul*ch_gain*band_mul))
    const float l0 = a.lut[cc * CVVDP_CSF_NODES + k] * kLog2_10 + fast_log2(a.sens_mul * a.ch_gain[cc] * a.band_mul);
    const float l1 = a.lut[cc * CVVDP_CSF_NODES + min(k + 1, CVVDP_CSF_NODES - 1)] * kLog2_10 + fast_log2(a.sens_mul * a.ch_gain[cc] * a.band_mul);
    s_lut[cc][k] = make_float2(l0, l1 - l0);
  }
  for (int i = t; i < 2 * NP * (B4_VE / 2); i += 64 * NCH) (&s_ve[0][0][0])[i] = make_float2(0.0f, 0.0f);   // unwritten apron elements
  __syncthreads();
  float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
  float mask_p = a.mask_p, eps_p = a.eps_p;
  // lpyr_dec.py:408, interp.py:93 in the log2 domain: ind = (log10 L - first) * scale = log2 L * ind_k1 - ind_k0 (host constants)
  // Wave-uniform constants that are used as plain VALU operands are parked in VGPRs (the empty asm hides their uniformity):
  // the loop needs ~110 SGPRs (13 + 14 blur taps, row arithmetic, exec masks), and every SGPR spilled to a VGPR lane
  // costs a v_readlane per use and row.  Only the packed-FMA taps must be SGPR pairs.
#define B4_IN_VGPR(x) asm volatile("" : "+v"(x))
  float ind_k1 = a.ind_k1, ind_k0 = a.ind_k0;
  float qc = a.q[c];
  float xw0 = a.xw[0 * 4 + c], xw1 = a.xw[1 * 4 + c], xw2 = a.xw[2 * 4 + c], xw3 = a.xw[3 * 4 + c];
  float m1c = a.m1[c];                             // 1 - sum_k xw[k][c] * eps^q_k: the "1 +" of the mask and the eps terms of safe_pow
  float inv_dmax = a.inv_dmax;
  if constexpr (RAGGED && !HEAT && !DUMP && !FEAT) {   // (the ragged instantiation has about eight VGPRs to spare: SGPR spills 73 -> 53,
    B4_IN_VGPR(xw1); B4_IN_VGPR(xw2); B4_IN_VGPR(xw3); B4_IN_VGPR(m1c); B4_IN_VGPR(inv_dmax); B4_IN_VGPR(ind_k0);   // lane reads per row pair 88 -> 68)
  }
  if constexpr (!RAGGED) {   // (the ragged instantiation is short of VGPRs instead)
    B4_IN_VGPR(ind_k0); B4_IN_VGPR(xw1); B4_IN_VGPR(xw2); B4_IN_VGPR(xw3); B4_IN_VGPR(m1c); B4_IN_VGPR(inv_dmax);
    if constexpr ((!HEAT && !DUMP) || FEAT) {   // (those instantiations have no VGPRs to spare; FEAT runs two blocks per CU: 256)
      B4_IN_VGPR(e0); B4_IN_VGPR(e1); B4_IN_VGPR(eo); B4_IN_VGPR(mask_p); B4_IN_VGPR(eps_p);
      B4_IN_VGPR(qc); B4_IN_VGPR(ind_k1); B4_IN_VGPR(xw0);
    }
  }
#undef B4_IN_VGPR

  // ---- expand, vertical half (lpyr_dec.py:229-232): a rolling window of three coarse rows (my-1, my, my+1, clamped)
  // in registers, my = row >> 1.  While the fine rows ascend inside the image (everywhere but the reflected rows at the
  // image's top and bottom edge) an odd row shares the window of the even row before it and an even row needs ONE new
  // coarse row, requested a row and a half earlier: each coarse row comes from HBM once.  Reflected rows reload the
  // whole window (the loads land in the window registers directly).
  const int cx = min(max(vcx, 0), Wc - 4);
  const bool clampL = vcx < 0;
  const bool edge_block = cb < 0 || cb + 128 > Wc;   // block-uniform: only edge strips pay for the replicate selects
  const float* gcl = gcp + (int64_t)(j >> 5) * gcps + cx;         // the lane's plane and chunk (64-bit: plane strides can pass 4 GB)
  float4 cA, cB, cC;        // coarse rows my-1, my, my+1 (clamped) of the chunk
  v4f cN = 0.0f;            // the next row up, in flight (hand-managed load, see STREAM LOADS)
  bool reloaded = false;    // scalar: the last reload request did load (its rows are still raw)
  auto coarse_load = [&](int row) -> float4 {
    const f4u q = *reinterpret_cast<const f4u*>(gcl + (int64_t)row * Wc);
    return make_float4(q.x, q.y, q.z, q.w);
  };
  // chunk reaching past column Wc-1: element i is loaded element min(i+c_shift, 3); with Wc % 4 == 0 a chunk is all in or all out
  const int c_shift = RAGGED ? min(max(vcx - (Wc - 4), 0), 3) : (vcx >= Wc ? 3 : 0);
  auto replicate = [&](float4 v) -> float4 {     // replicate column 0 / Wc-1 for chunks left of / reaching past the image
    if (edge_block && (clampL || c_shift > 0)) {
      if (clampL) {
        v = make_float4(v.x, v.x, v.x, v.x);
      } else {
        const float e0_ = c_shift == 1 ? v.y : (c_shift == 2 ? v.z : v.w), e1_ = c_shift == 1 ? v.z : v.w;
        v = make_float4(e0_, e1_, v.w, v.w);
      }
    }
    return v;
  };
  // request what the window of fine row q needs.  FAST (rows q-1 and q inside the image, the window holds row q-1): an odd
  // row shares its predecessor's window, an even row needs the next coarse row up.  Otherwise the whole window is reloaded.
  auto coarse_issue = [&](int q, auto odd, auto fast) {
    if constexpr (!decltype(fast)::value) {
      // (an odd row inside the image shares its predecessor's window here too: the rolling loop that may follow finishes
      // such a row without touching the window, so it must not be handed a raw, un-replicated reload)
      reloaded = !(decltype(odd)::value && q >= 1 && q <= H - 1);
      if (reloaded) {
        const int my = min(refl(q, H), H - 1) >> 1;
        cA = coarse_load(max(my - 1, 0));
        cB = coarse_load(my);
        cC = coarse_load(min(my + 1, Hc - 1));
      }
    }
  };
  // move the window to the row requested last (same odd / fast as its coarse_issue) and write its vertically expanded row to s_ve[buf]
  auto coarse_finish = [&](int buf, auto odd, auto fast) {
    if constexpr (decltype(fast)::value) {
      if constexpr (!decltype(odd)::value) { cA = cB; cB = cC; cC = replicate(make_float4(cN.x, cN.y, cN.z, cN.w)); }
    } else if (reloaded) {                          // raw rows: once only (a shifted chunk must not be shifted again)
      cA = replicate(cA); cB = replicate(cB); cC = replicate(cC);
    }
    const float m0[4] = {cA.x, cA.y, cA.z, cA.w}, m1[4] = {cB.x, cB.y, cB.z, cB.w}, m2[4] = {cC.x, cC.y, cC.z, cC.w};
    float o[4];
    if constexpr (decltype(odd)::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = expand_odd(m1[i], m2[i], eo);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = expand_even(m0[i], m1[i], m2[i], e0, e1);
    }
    lds_write4(reinterpret_cast<float*>(&s_ve[buf][vp][2 + 2 * vch]), o);
  };

  // horizontal half of the expand for this lane's 4 columns from 4 coarse values (lpyr_dec.py:234-237)
  auto expand4 = [&](const float2* row, float (&ex)[4]) {
    // coarse cb+2j-1 .. cb+2j+2 live at elements 2j+3 .. 2j+6
    const float2 p0 = row[j + 1];
    const float2 p1 = row[j + 2];
    const float2 p2 = row[j + 3];
    const float A = p0.y, B = p1.x, C = p1.y, D = p2.x;
    ex[0] = expand_even(A, B, C, e0, e1);
    ex[1] = expand_odd(B, C, eo);
    ex[2] = expand_even(B, C, D, e0, e1);
    ex[3] = expand_odd(C, D, eo);
  };

  // Per-column luminance terms of one row, shared by all channels (lpyr_dec.py:394,:408; interp.py:93): every
  // thread expands the luminance planes for ONE column (blocks of 64*NCH columns) and publishes 1/L_T, 1/L_R
  // and the sensitivity of every channel (LUT lerp + exp2), so the log / reciprocal work is done once per pixel.  Element 4+i of an s_ve row is coarse column cb+i; fine column col -> 4+(col>>1).
  const bool lodd = t & 1;                           // 64*NCH is even: a thread's columns keep their parity
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
      const float fr = __builtin_amdgcn_fractf(ind);                           // ind >= 0: ind - floor(ind)
      s_lum[0][col] = fast_rcp(Lt);
      s_lum[1][col] = fast_rcp(Lr);
      if constexpr (S_SHARED) {
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {                                     // csf.py:49, cvvdp_metric.py:709,:836
          const float2 ln = s_lut[cc][i0];
          s_S[cc][col] = fast_exp2(ln.x + ln.y * fr);
        }
      } else {
        s_lum[2][col] = fr;
        s_lum[3][col] = __int_as_float(i0 * 8);                                // byte offset into a float2 LUT row
      }
    }
  };

  // Vertical-blur window: slot s of column i holds one horizontally blurred row.  The newest row is written
  // into slot (row index mod 13) with an M0-relative register write (dynamic insertelement on a register
  // vector); the 13 weights -- wave-uniform scalars -- are rotated instead of the data, so the FMAs use
  // static register indices and no 13-way switch / PHI copies are needed.
  typedef float v32f __attribute__((ext_vector_type(32)));
  v32f winA = 0.0f, winB = 0.0f;   // columns (0,1) and (2,3) interleaved: element 2s+i = slot s of column i
  // horizontal taps carry the mask gain 10^mask_c (the blur is linear; a.blur_h is pre-scaled on the host so that the
  // taps stay in SGPRs): Mq needs no extra multiply
  // The taps are symmetric (b[k] = b[12-k]), so the twelve tap pairs of the two alignments are seven SGPR pairs and
  // their lane swaps (op_sel on the packed FMA): 14 SGPRs instead of 25.
  const v2f E0 = {a.blur_h[0], a.blur_h[1]}, E1 = {a.blur_h[2], a.blur_h[3]}, E2 = {a.blur_h[4], a.blur_h[5]}, E3 = {a.blur_h[6], a.blur_h[5]};
  const v2f O0 = {a.blur_h[1], a.blur_h[2]}, O1 = {a.blur_h[3], a.blur_h[4]}, O2 = {a.blur_h[5], a.blur_h[6]};
#define B4_SWAP(p) __builtin_shufflevector(p, p, 1, 0)
  const v2f be[6] = {E0, E1, E2, E3, B4_SWAP(O1), B4_SWAP(O0)};
  const v2f bo[6] = {O0, O1, O2, B4_SWAP(E2), B4_SWAP(E1), B4_SWAP(E0)};
#undef B4_SWAP
  const float b0 = a.blur_h[0], b12 = a.blur_h[0];
  float wr[B4_BW];     // wr[s] = weight of slot s for the NEXT row to be written into slot 0
#pragma unroll
  for (int k = 0; k < B4_BW; ++k) wr[k] = a.blur[(k + B4_BW - 1) % B4_BW];
  float acc = 0.0f;

  // ---- FEATURES (cvvdp_feature_pooling, cvvdp_ml_metric.py:77-107, called at :351-358): mean and E[x^2] of
  // |T'| = |T_f|*S, |R'| and D over feature_size x feature_size cells.  The march accumulates, per lane and COLUMN, the sums of
  // the six quantities over the rows of the current cell row (24 registers; this instantiation runs two blocks per CU) and
  // stores them as one row of column sums when the march crosses a cell-row boundary or leaves the segment: 24 B per pixel
  // and boundary -- 24/fs B/pixel of traffic instead of the 96 B/pixel of per-pixel planes.  A piece of a cell row is keyed
  // (cell row + segment index), which is unique (segments ascend with the rows); k_feature_finish adds the pieces and the
  // columns of each cell in double, in a fixed order.  Rows: |T'|, |R'| belong to row r of the contrast stage, D to the row
  // of the pooling stage (seven rows behind): two independent trackers.
  float f_t[4] = {0, 0, 0, 0}, f_t2[4] = {0, 0, 0, 0}, f_r[4] = {0, 0, 0, 0}, f_r2[4] = {0, 0, 0, 0}, f_d[4] = {0, 0, 0, 0}, f_d2[4] = {0, 0, 0, 0};
  int f_left_tr = 0, f_left_d = 0;                  // rows to the next cell-row boundary (scalar)
  if constexpr (FEAT) { f_left_tr = f_left_d = a.fs - ys % a.fs; }
  auto feat_store = [&](int y_last, int q0, const float (&s0)[4], const float (&s1)[4]) {   // column sums of the piece that ends with row y_last
    if constexpr (FEAT) {
      if (interior) {
        const int piece = y_last / a.fs + seg;
        float* dst = a.fsum + ((((int64_t)item * NCH + c) * a.f_pieces + piece) * 6 + q0) * W + fc0;
        if (n_valid == 4) {
          *reinterpret_cast<f4u*>(dst) = f4u{s0[0], s0[1], s0[2], s0[3]};
          *reinterpret_cast<f4u*>(dst + W) = f4u{s1[0], s1[1], s1[2], s1[3]};
        } else {
          for (int i = 0; i < n_valid; ++i) { dst[i] = s0[i]; dst[W + i] = s1[i]; }
        }
      }
    }
  };

  // pooling stage of centre row y (cvvdp_metric.py:849-856, 722): needs s_q of all channels and s_d (ring slot k7)
  auto stage3c = [&](int y, int k7) {
    const f4 q0 = lds_read4(&s_q[0][4 * j]), q1 = lds_read4(&s_q[1][4 * j]), q2 = lds_read4(&s_q[2][4 * j]);
    f4 q3 = f4{{0.0f, 0.0f, 0.0f, 0.0f}};
    if constexpr (NCH == 4) q3 = lds_read4(&s_q[3][4 * j]);
    const f4 d = lds_read4(&s_d[k7][c][4 * j - B4_HALO]);
    float D[4], De[4];                 // D (heat map, dump, features) and D + eps (the pooled term)
#pragma unroll
    for (int h = 0; h < 2; ++h) {      // two column pairs: the mask sum and the clamp denominator as packed FMAs (-1.5 %; packing the
                                       // expand or the vertical combine as well costs registers: slower, profiles/r02_dev_notes.txt)
      const v2f Q0 = {q0.v[2 * h], q0.v[2 * h + 1]}, Q1 = {q1.v[2 * h], q1.v[2 * h + 1]}, Q2 = {q2.v[2 * h], q2.v[2 * h + 1]}, Q3 = {q3.v[2 * h], q3.v[2 * h + 1]};
      // 1 + M, M = sum_k xw[k][c] * ((blur_k*10^mask_c + eps)^q_k - eps^q_k)     (cvvdp_metric.py:758-760, :849)
      const v2f M1 = Q3 * xw3 + (Q2 * xw2 + (Q1 * xw1 + (Q0 * xw0 + m1c)));
      // Du = X/(1+M); D = dmax*Du/(dmax+Du) = X / ((1+M) + X/dmax): one reciprocal (:855-856, :949-950); s_d holds |T'-R'| + eps
      const v2f X = {fast_pow(d.v[2 * h], mask_p) - eps_p, fast_pow(d.v[2 * h + 1], mask_p) - eps_p};
      const v2f T = X * inv_dmax + M1;
      const float r0 = fast_rcp(T.x), r1 = fast_rcp(T.y);
      De[2 * h] = __builtin_fmaf(X.x, r0, kEps); De[2 * h + 1] = __builtin_fmaf(X.y, r1, kEps);
      if constexpr (HEAT || DUMP || FEAT) { D[2 * h] = X.x * r0; D[2 * h + 1] = X.y * r1; }
      else { D[2 * h] = 0.0f; D[2 * h + 1] = 0.0f; }
    }
    if (ragged_blk) {                                              // columns right of the image do not exist: no term
#pragma unroll
      for (int i = 1; i < 4; ++i) { D[i] = i < n_valid ? D[i] : 0.0f; De[i] = i < n_valid ? De[i] : 0.0f; }
    }
    // safe_pow(D, 2) = (D + eps)^2 - eps^2 (cvvdp_metric.py:1032-1050, beta = 2): the lanes sum (D + eps)^2 -- one FMA per pixel -- and
    // k_finalize takes the eps^2 of the level's H*W terms off the mean in double (FinalizeArgs::sub_per_term)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_fmaf(De[i], De[i], acc);
    if constexpr (FEAT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { f_d[i] += D[i]; f_d2[i] = __builtin_fmaf(D[i], D[i], f_d2[i]); }
    }
    if constexpr (DUMP) {                                           // per-pixel D for tests / features (its own instantiation)
      float* dd = a.ddump + (int64_t)c * a.items_cap * P + (int64_t)item * P + (int64_t)y * W + fc0;
      if (n_valid == 4) *reinterpret_cast<f4u*>(dd) = f4u{D[0], D[1], D[2], D[3]};
      else for (int i = 0; i < n_valid; ++i) dd[i] = D[i];
    }
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

  // ---- STREAM LOADS.  The g rows (two planes) and the coarse window's next row are requested a
  // whole row (two phases) before they are used and must stay in flight across a barrier and the other row's loads.
  // hipcc's wait-count insertion drains the queue (vmcnt(0)) at the first use after a loop back edge, which halves the
  // bytes in flight and leaves the kernel waiting for HBM (57 % of the wave cycles parked, measured).  So these loads
  // are issued from inline assembly, which that pass does not track, and waited for explicitly:
  //     even step, phase 2:  g rows of row r+2 -> p0T, p0R                      (2 loads)
  //     odd  step, phase 2:  coarse row for the window move of row r+3 -> cN,
  //                          g rows of row r+2 -> p1T, p1R                      (3 loads)
  //     even step, phase 1:  needs p0*: younger loads cN, p1T, p1R  -> s_waitcnt vmcnt(3)
  //     odd  step, phase 1:  needs p1*, cN: younger loads p0T, p0R  -> s_waitcnt vmcnt(2)
  // Loads return in order, so any extra (compiler-issued) memory operation in between only makes these waits stricter;
  // the destination registers are tied ("+v") from the request to the wait, and everything is drained between loops and
  // before the epilogue (a register copy or reuse while a load is in flight would read / clobber stale data:
  // tools/check_band4_isa.py checks the generated code for that).  Out-of-image halo lanes read a clamped, valid address.
  const uint32_t goff = (uint32_t)min(max(fc0, 0), W - 4) * 4u;
  v4f p0T = 0.0f, p0R = 0.0f, p1T = 0.0f, p1R = 0.0f;       // g rows of the even / odd row in flight
// (no nontemporal hint: with the XCD-aware block order the halo columns and edge lines a strip shares with its neighbours
// are L2 hits, 11.2 instead of 12.0 MB-K of FETCH_SIZE per 4K x 64 launch)
#ifdef CVVDP_SAFE_LOADS
// `make safe`: the same kernel with ordinary loads the compiler tracks and waits for itself (tests/test_safe_loads.py)
#define B4_G_LOAD(dst, plane, row) \
  do { const f4u q_ = *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>((plane) + (int64_t)(row) * W) + goff); dst = v4f{q_.x, q_.y, q_.z, q_.w}; } while (0)
#define B4_C_LOAD(dst, row) do { const f4u q_ = *reinterpret_cast<const f4u*>(gcl + (int64_t)(row) * Wc); dst = v4f{q_.x, q_.y, q_.z, q_.w}; } while (0)
#define B4_WAIT_EVEN() do { } while (0)
#define B4_WAIT_ODD() do { } while (0)
#else
#define B4_G_LOAD(dst, plane, row) \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(goff), "s"((plane) + (int64_t)(row) * W))
#define B4_C_LOAD(dst, row) \
  asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(gcl + (int64_t)(row) * Wc))
#define B4_WAIT_EVEN() asm volatile("s_waitcnt vmcnt(3)" : "+v"(p0T), "+v"(p0R))
#define B4_WAIT_ODD() asm volatile("s_waitcnt vmcnt(2)" : "+v"(p1T), "+v"(p1R), "+v"(cN))
#endif
// (the builtin, not asm: the compiler's own wait-count bookkeeping must see that nothing of ITS loads is pending either, or it
// keeps re-waiting inside the next loop; 0x0F70 = vmcnt(0) with the other counters at their maxima on gfx9)
#define B4_DRAIN() do { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" : "+v"(p0T), "+v"(p0R), "+v"(p1T), "+v"(p1R), "+v"(cN)); } while (0)
  auto stream_issue = [&](int r, auto odd_row) {   // phase 2 of row r
    (void)&p0T; (void)&p0R; (void)&p1T; (void)&p1R; (void)&cN; (void)&goff; (void)&gT; (void)&gR; (void)&gcl; (void)&W; (void)&Wc;   // (asm operands alone do not capture in a generic lambda)
    const int r2 = min(refl(r + 2, H), H - 1);
    if constexpr (decltype(odd_row)::value) {
      const int rowc = min(max(((r + 3) >> 1) + 1, 0), Hc - 1);   // what the ascending window move of row r+3 needs
      B4_C_LOAD(cN, rowc);
      B4_G_LOAD(p1T, gT, r2);
      B4_G_LOAD(p1R, gR, r2);
    } else {
      B4_G_LOAD(p0T, gT, r2);
      B4_G_LOAD(p0R, gR, r2);
    }
  };

  // image-edge mirror roles (see the contrast stage): left edge = lanes with fc0 = 0 / 4 (strip 0), right edge = lanes
  // with fc0 = W-8 / W-4 (last strip); W >= 16 keeps them apart
  const bool mir_block = strip == 0 || edge_r;
  const bool mir_any = RAGGED && (W & 3) != 0;       // right edge not on a lane boundary: per-column mirror writes
  int mir_kind = 0, mir_base = 0;
  if (strip == 0 && (fc0 == 0 || fc0 == 4)) { mir_kind = fc0 == 0 ? 1 : 2; mir_base = B4_HALO - (fc0 == 0 ? 1 : 4); }
  if (!mir_any && (fc0 == W - 8 || fc0 == W - 4)) {
    const int base = 2 * (W - 1) - (fc0 + (fc0 == W - 8 ? 1 : 0)) - (x0 - B4_HALO);
    if (base < 256) { mir_kind = fc0 == W - 8 ? 1 : 2; mir_base = base; }   // (the right halo lanes of the strip before the last can hold these columns too: out of its range)
  }

  // ---- prologue: window + expand of the first row, g rows of the first two rows, luminance terms of the first row
  const int r0 = ys - B4_R, rend = ye + B4_R;
  {
    coarse_issue(r0, std::false_type{}, std::false_type{});        // r0 is even
    stream_issue(r0 - 2, std::false_type{});                       // g rows of row r0
    coarse_finish(0, std::false_type{}, std::false_type{});
    coarse_issue(r0 + 1, std::true_type{}, std::false_type{});
    stream_issue(r0 - 1, std::true_type{});                        // g rows of row r0+1, coarse row for the move of row r0+2
  }
  __syncthreads();
  lum_prep(0);
  __syncthreads();

  int slot = 0, k7 = 0;            // (r - r0) mod 13: blur window slot; (r - r0) mod 7: s_d ring slot
  // one row.  Its g values sit in p0* (even row) or p1* (odd row) and are reloaded with row r+2 in phase 2.
  // fin_fast / iss_fast: coarse-window mode of row r+1 (finished here) and of row r+2 (requested here)
  auto step = [&](int r, auto odd_row, auto fin_fast, auto iss_fast) {
    constexpr bool ODD = decltype(odd_row)::value;
    (void)&p0T; (void)&p0R; (void)&p1T; (void)&p1R; (void)&cN;   // (asm operands alone do not capture in a generic lambda)
    if constexpr (ODD) B4_WAIT_ODD(); else B4_WAIT_EVEN();
    v4f pT, pR;
    if constexpr (ODD) { pT = p1T; pR = p1R; } else { pT = p0T; pR = p0R; }
    if (ragged_blk) {                                 // partial lane: its load was clamped to the row's last four columns
      if (g_shift > 0) {
        pT = v4f{g_shift == 1 ? pT.y : (g_shift == 2 ? pT.z : pT.w), g_shift == 1 ? pT.z : pT.w, pT.w, pT.w};
        pR = v4f{g_shift == 1 ? pR.y : (g_shift == 2 ? pR.z : pR.w), g_shift == 1 ? pR.z : pR.w, pR.w, pR.w};
      }
    }
    // ================= phase 1
    const int yprev = r - 1 - B4_R;                  // row whose Mq was published last iteration (ring slot k7, like row r)
    if (interior && yprev >= ys) stage3c(yprev, k7);
    if constexpr (FEAT) {                             // D sums: a cell row ends with row yprev (the segment's last row is the epilogue's)
      if (yprev >= ys && --f_left_d == 0) {
        feat_store(yprev, 4, f_d, f_d2);
#pragma unroll
        for (int i = 0; i < 4; ++i) f_d[i] = f_d2[i] = 0.0f;
        f_left_d = a.fs;
      }
    }
    const bool feat_row = FEAT && r >= ys && r < ye;  // (scalar) row r belongs to this segment: its |T'|, |R'| are counted
    if (in_img) {
      float exT[4], exR[4];
      expand4(s_ve[ODD][2 * c], exT);
      expand4(s_ve[ODD][2 * c + 1], exR);
      const f4 rLt = lds_read4(&s_lum[0][4 * j]), rLr = lds_read4(&s_lum[1][4 * j]);
      f4 Sv;
      if constexpr (S_SHARED) {
        Sv = lds_read4(&s_S[c][4 * j]);
      } else {
        const f4 fr = lds_read4(&s_lum[2][4 * j]), lo = lds_read4(&s_lum[3][4 * j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float2 ln = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(&s_lut[c][0]) + __float_as_int(lo.v[i]));
          Sv.v[i] = fast_exp2(ln.x + ln.y * fr.v[i]);                          // csf.py:49, cvvdp_metric.py:709,:836
        }
      }
      const float gt[4] = {pT.x, pT.y, pT.z, pT.w}, gr[4] = {pR.x, pR.y, pR.z, pR.w};
      float m[4], d[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float S = Sv.v[i];
        const float ct = fminf((gt[i] - exT[i]) * rLt.v[i], 1000.0f);            // lpyr_dec.py:402 (band gain :66 is in S)
        const float cr = fminf((gr[i] - exR[i]) * rLr.v[i], 1000.0f);
        if constexpr (FEAT) {
          const float at = fabsf(ct) * S, ar = fabsf(cr) * S;                    // |T'|, |R'| (the channel gain inside S is divided out by k_feature_finish)
          m[i] = fminf(at, ar);                                                  // = min(|ct|,|cr|)*S bit for bit (rounding is monotone)
          if (feat_row) { f_t[i] += at; f_t2[i] = __builtin_fmaf(at, at, f_t2[i]); f_r[i] += ar; f_r2[i] = __builtin_fmaf(ar, ar, f_r2[i]); }
        } else {
          m[i] = fminf(fabsf(ct), fabsf(cr)) * S;                                // min(|T'|,|R'|), T' = ct*S (cvvdp_metric.py:845)
        }
        d[i] = fabsf(ct - cr) * S + kEps;                                        // |T'-R'| + eps (:855, safe_pow)
      }
      lds_write4(&s_m[c][4 * j], m);
      if (interior) lds_write4(&s_d[k7][c][4 * j - B4_HALO], d);
      // reflect padding of the blur at the image's left/right edge: mirror columns 1..6 / W-7..W-2.  Two lanes per
      // edge hold them: the outer one mirrors its columns 1..3 (kind 1), the inner one its columns 0..2 (kind 2), to
      // three consecutive descending LDS elements
      if (mir_block) {
        if (mir_kind != 0) {
          const float v0 = mir_kind == 1 ? m[1] : m[0], v1 = mir_kind == 1 ? m[2] : m[1], v2 = mir_kind == 1 ? m[3] : m[2];
          float* dst = &s_m[c][mir_base];
          dst[0] = v0; dst[-1] = v1; dst[-2] = v2;
        }
        if (mir_any) {                                // columns W-7 .. W-2, wherever they fall in the lanes, to 2(W-1)-x
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int x = fc0 + i;
            const int idx = 2 * (W - 1) - x - (x0 - B4_HALO);
            if (x >= W - 1 - B4_R && x <= W - 2 && idx < 256) s_m[c][idx] = m[i];
          }
        }
      }
    }
    if constexpr (FEAT) {
      if (feat_row && (--f_left_tr == 0 || r == ye - 1)) {
        feat_store(r, 0, f_t, f_t2);
        feat_store(r, 2, f_r, f_r2);
#pragma unroll
        for (int i = 0; i < 4; ++i) f_t[i] = f_t2[i] = f_r[i] = f_r2[i] = 0.0f;
        f_left_tr = a.fs;
      }
    }
    coarse_finish(ODD ? 0 : 1, std::integral_constant<bool, !ODD>{}, fin_fast);   // vertical expand of row r+1 (its coarse row was requested a phase ago)
    __syncthreads();
    // ================= phase 2
    if (yprev >= ys) heat_row(yprev);                 // terms of row yprev were published in phase 1
    coarse_issue(r + 2, odd_row, iss_fast);           // row r+2 has the parity of row r
    stream_issue(r, odd_row);
    lum_prep(ODD ? 0 : 1);                            // luminance planes of row r+1 were published in phase 1
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
      winA[2 * slot] = h[0]; winA[2 * slot + 1] = h[1]; winB[2 * slot] = h[2]; winB[2 * slot + 1] = h[3];
      // weight of slot s when the newest row sits in `slot`: window position j = (s - slot - 1) mod 13 -> wr[]
      // is kept rotated so that wr[s] is exactly that weight; columns (0,1) and (2,3) share one packed FMA
      if (yc >= ys) {                                     // (the first twelve rows of a segment only fill the window)
        v2f va = kEps, vb = kEps;                         // safe_pow's "+ eps" (cvvdp_metric.py:849) as the accumulators' start value
#pragma unroll
        for (int sdx = 0; sdx < B4_BW; ++sdx) {
          const v2f wa = {winA[2 * sdx], winA[2 * sdx + 1]}, wb = {winB[2 * sdx], winB[2 * sdx + 1]};
          const v2f ww = {wr[sdx], wr[sdx]};
          va += ww * wa; vb += ww * wb;
        }
        const float v[4] = {va.x, va.y, vb.x, vb.y};
        float Mq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) Mq[i] = fast_pow(v[i], qc);          // (blur*10^mask_c + eps)^q_c; "- eps^q" is inside m1c
        lds_write4(&s_q[c][4 * j], Mq);
      }
    }
    {   // rotate the blur weights for the next row (scalar ALU)
      const float last = wr[B4_BW - 1];
#pragma unroll
      for (int k = B4_BW - 1; k > 0; --k) wr[k] = wr[k - 1];
      wr[0] = last;
    }
    slot = slot == B4_BW - 1 ? 0 : slot + 1;
    k7 = k7 == B4_R ? 0 : k7 + 1;
    __syncthreads();
  };

  // Row pairs (even, odd).  Reflected rows (above the image: r < 0; at its bottom: r + 3 > H - 1) reload the coarse window
  // row by row; in between the window rolls.  A row's window mode belongs to the row: requested in one step, finished in
  // the next, so the first step of the rolling loop finishes a reloaded row (for an odd row the two are the same thing).
  int r = r0;
  bool done = false;
  for (int pass = 0; pass < 2 && !done; ++pass) {
    const int lim = pass == 0 ? min(rend, 0) : rend;
    for (; r < lim; r += 2) {
      step(r, std::false_type{}, std::false_type{}, std::false_type{});
      if (r + 1 >= rend) { done = true; break; }
      step(r + 1, std::true_type{}, std::false_type{}, std::false_type{});
    }
    B4_DRAIN();
    if (pass == 0 && !done) {
      for (; r < rend && r + 3 <= H - 1; r += 2) {
        step(r, std::false_type{}, std::true_type{}, std::true_type{});    // finishes odd row r+1 (shares the window)
        if (r + 1 >= rend) { done = true; break; }
        step(r + 1, std::true_type{}, std::true_type{}, std::true_type{});  // finishes even row r+2: the window moves up
      }
      B4_DRAIN();
    }
  }
  // ---- epilogue: pooling stage of the last centre row
  if (interior && (ye - 1) >= ys) stage3c(ye - 1, k7);    // row ye-1 = rend-7 shares the ring slot of row rend
  if constexpr (FEAT) {
    if ((ye - 1) >= ys) feat_store(ye - 1, 4, f_d, f_d2);
  }
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

#undef B4_G_LOAD
#undef B4_C_LOAD
#undef B4_WAIT_EVEN
#undef B4_WAIT_ODD
#undef B4_DRAIN

template <bool RAGGED>
static void launch_band4_w(const BandArgs& a, hipStream_t s) {
  dim3 grid(8 * a.per_xcd);
  if (a.fsum) {          // features for the ML heads (no heat map, no dump: extract_features refuses the one, core.cpp never sets the other)
    if (a.nch == 4) hipLaunchKernelGGL((k_band4<4, false, RAGGED, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_band4<3, false, RAGGED, false, true>), grid, dim3(192), 0, s, a);
  } else if (a.dchr) {   // heat map (with or without the per-pixel dump)
    if (a.ddump) {
      if (a.nch == 4) hipLaunchKernelGGL((k_band4<4, true, RAGGED, true, false>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_band4<3, true, RAGGED, true, false>), grid, dim3(192), 0, s, a);
    } else {
      if (a.nch == 4) hipLaunchKernelGGL((k_band4<4, true, RAGGED, false, false>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_band4<3, true, RAGGED, false, false>), grid, dim3(192), 0, s, a);
    }
  } else if (a.ddump) {
    if (a.nch == 4) hipLaunchKernelGGL((k_band4<4, false, RAGGED, true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_band4<3, false, RAGGED, true, false>), grid, dim3(192), 0, s, a);
  } else {
    if (a.nch == 4) hipLaunchKernelGGL((k_band4<4, false, RAGGED, false, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_band4<3, false, RAGGED, false, false>), grid, dim3(192), 0, s, a);
  }
}

// W % 8 != 0: a lane of the last strip, or a coarse 4-column chunk, straddles the right image edge.  Only the strips that
// can see that edge need the RAGGED instantiation (partial lane, shifted coarse chunks, per-column mirrors: 5 spilled VGPRs and
// ~75 spilled SGPRs in the row loop): a strip whose 256 columns x0-8 .. x0+247 end inside the image (x0 + 248 <= W, which also
// keeps its coarse chunks inside: 2*Wc >= W) runs exactly the instructions of an aligned image -- rows of any pitch are fine,
// the 16-byte loads only need 4-byte alignment.
int band4_edge_strips(int W, int n_strip) {
  if ((W & 7) == 0) return 0;
  int n = 0;
  while (n < n_strip && (n_strip - 1 - n) * B4_SW + B4_SW + B4_HALO > W) ++n;
  return n;
}

void launch_band4(const BandArgs& a0, bool split_edge, hipStream_t s, hipStream_t s_edge) {
  BandArgs a = a0;
  int n_edge = band4_edge_strips(a.W, a.n_strip);
  if (n_edge > 0 && !split_edge) n_edge = a.n_strip;  // one launch: every strip on the RAGGED instantiation
  if (n_edge > 0) {                                  // the few edge strips first: on their own stream they run beside the rest
    a.strip0 = a.n_strip - n_edge; a.n_strip_l = n_edge;
    a.per_xcd = (a.n_strip_l * a.n_seg * a.items + 7) / 8;
    launch_band4_w<true>(a, s_edge);
  }
  if (n_edge < a.n_strip) {
    a.strip0 = 0; a.n_strip_l = a.n_strip - n_edge;
    a.per_xcd = (a.n_strip_l * a.n_seg * a.items + 7) / 8;
    launch_band4_w<false>(a, s);
  }
}

int tu_flags_band4() {
#ifdef CVVDP_SAFE_LOADS
  return CVVDP_BUILD_SAFE_LOADS;
#else
  return 0;
#endif
}

}  // namespace cvvdp
