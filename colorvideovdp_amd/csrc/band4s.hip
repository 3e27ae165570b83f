// k_band4f's row march (band4f.hip: the band kernel that computes the coarse level it expands from) divided between TWO KINDS OF WAVES.
//
// Why (round 3, profiles/r03_dev_notes.txt 2, 14): with one wave per channel carrying everything -- the 8-row ring
// of raw rows (64 VGPRs), the 5x5 reduce, expand, contrast, CSF, both blurs (52-register window), masking and pooling -- k_band4f needs
// 245 VGPRs, runs two waves per SIMD and keeps the VALU pipes 65 % busy: a wave's row is a serial chain of six LDS round trips, two
// block barriers and ~1150 issue cycles, and one other wave cannot cover it.  The two big pieces of per-lane state never meet:
//   * FRONT waves (one per channel, waves 0..3): stream the level-l rows (hand-issued loads, the 8-row register ring), reduce row
//     r+5 (horizontal 5-tap + the running vertical sums), store the completed coarse row as level l+1, roll the 3-row expand window,
//     write the vertically expanded row r+1 into s_ve, hand the RAW row r+1 to the back through LDS (s_g), and -- one column per
//     thread -- the luminance terms of row r+1 (1/L_T, 1/L_R, the CSF sensitivities of all channels).
//   * BACK waves (one per channel, waves 4..7): horizontal expand + Weber contrast + min / difference of row r (from s_ve, s_g,
//     s_lum, s_S), the horizontal 13-tap blur of the same row (s_m is written and read by the same wave: no block barrier), the
//     13-row vertical blur window in registers, masking, soft clamp and pooling seven rows behind.
// Each kind needs <= 128 VGPRs, so a CU holds two 8-wave blocks = four waves per SIMD, and -- waves being dealt to the SIMDs round
// robin -- every SIMD gets the front and the back wave of one channel of each block: the work per SIMD is what it was, but a wave's
// chain is half as long and three other waves stand by.  Same two barriers per row; per-row hand-offs are double-buffered by row
// parity (s_ve, s_g).  LDS: 65.8 KB per block (k_band4f: 50) -- two blocks per CU fit the 160 KB.
//
// The arithmetic is k_band4f<4, 0>'s, operation for operation (same FMA chains, same order): the two kernels' level-(l+1) planes are
// bit-identical and their partial sums agree to the last bit -- they are two separately compiled kernels, and the compiler contracts a
// multiply-add here and not there (tests/test_gpu_parity.py::test_split_band_kernel_matches_the_one_wave_layout).  The strips away from the
// image's left / right border run the EDGE = 0 body (k_band4s / _heat / _feat); since round 5 the border strips of W % 4 == 0 frames run
// the same layout's EDGE = 1 body as kernels of their own (k_band4s_edge / _edge_heat, on the edge stream beside the others); W % 4 == 2
// frames and features clips keep k_band4f<4, 2> / k_band4f_feat<4, 1> for their border strips (launch_band4f).
// BARRIERS IN DIVERGENT ROLES.  Front and back waves take different branches of `if (front)` and each runs its own copy of the
// row loop; they meet at s_barrier, which counts arriving WAVES of the workgroup, not program points -- outside what the HIP model
// promises for __syncthreads, and relied on deliberately.  What keeps it sound: both roles execute exactly two barriers per row step and
// the same number of steps (prologue 2, real rows r_start .. rreal-1, reflected rows rreal .. rend-1, +1 under HEAT), every loop bound is
// wave-uniform and the same expression in both roles, and tests/test_band4_isa.py checks on the generated code that every loop holding
// barriers holds two per row step.  A wave that left early (the `wu >= ...` return) left as a whole workgroup.
// Reference arithmetic: lpyr_dec.py:186-239,386-408, cvvdp_metric.py:835-856,945-950,963-971 (see band4.hip / band4f.hip).
#include <type_traits>
#include <utility>
#include "kernels.h"

namespace cvvdp {

namespace {

constexpr int S_R = 6;             // blur radius
constexpr int S_BW = 13;
constexpr int S_HALO = 8;          // aligned halo columns per side
constexpr int S_SW = 256 - 2 * S_HALO;   // 240 interior columns per strip (= kBand4StripWidth)
constexpr int S_VE = 136;          // s_ve row: element 4+i = coarse column cb+i (i = 0..127)

struct sf4 { float v[4]; };
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ sf4 s_lds_read4(const float* p) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  return sf4{{q.x, q.y, q.z, q.w}};
}
__device__ __forceinline__ void s_lds_write4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

template <class F, int... Us>
__device__ __forceinline__ void s_for_seq(F&& f, std::integer_sequence<int, Us...>) { (f(std::integral_constant<int, Us>{}), ...); }

}  // namespace

#ifndef CVVDP_BAND4S_RING
#define CVVDP_BAND4S_RING 8         // rows of the front waves' register ring (STREAM LOADS)
#endif

// timing-only diagnostics (tools/build_variant.sh; results wrong by construction): S_DIAG_NOBAR drops the block barriers of the row loops
#ifdef S_DIAG_NOBAR
#define S_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define S_SYNC() __syncthreads()
#endif

// HEAT: the level's heat-map band as well (band4.hip, HEAT): the back waves publish their channel's term of the pooled row, and one
// barrier later the 256 back threads take one column each (lp_norm over the channels).
// FEAT: the statistics of the ML heads' feature pooling as well (band4.hip, FEATURES): per lane and column, the sums of |T'|, |T'|^2, |R'|,
// |R'|^2 (contrast stage) and D, D^2 (pooling stage, seven rows behind) over the rows of the current cell row, stored as one row of
// column sums per cell-row piece; k_feature_finish adds the pieces.  24 more registers in the back waves.
struct __attribute__((packed, aligned(4))) s_u4 { float x, y, z, w; };     // 16 bytes at 4-byte alignment (rows of W % 4 == 2 frames)

// The block's LDS (a struct handed to the body: the kernel instantiates the body twice -- strips away from the image border and strips AT
// it -- and function-local __shared__ arrays of two instantiations would be allocated twice)
template <bool HEAT, bool FEAT>
struct S4Lds {
  static constexpr int NCH = 4, NP = 8;
  __attribute__((aligned(16))) float s_h[HEAT ? NCH : 1][HEAT ? 256 : 4];   // heat-map terms of the pooled row, per channel
  // FEAT: the column sums of D and D^2 live in LDS (lane-private: no barrier) -- with all 24 sums in registers the back waves spilled
  // eight constants and reloaded three of them per row from scratch (level 0 of 4K x 64: 10.5 ms against 7.2 of the plain kernel)
  __attribute__((aligned(16))) float s_fd[2][FEAT ? NCH : 1][FEAT ? 256 : 4];
  __attribute__((aligned(16))) float2 s_ve[2][NP][S_VE / 2];
  __attribute__((aligned(16))) float s_g[2][NP][256];             // raw level-l row handed from the front to the back (by row parity)
  __attribute__((aligned(16))) float s_lum[2][256];               // 1/L_T, 1/L_R
  __attribute__((aligned(16))) float s_S[NCH][256];
  __attribute__((aligned(16))) float s_m[NCH][256];
  __attribute__((aligned(16))) float s_q[NCH][256];
  __attribute__((aligned(16))) float s_d[S_R + 1][NCH][S_SW];     // lane-private ring of |T'-R'| + eps
  __attribute__((aligned(16))) float2 s_lut[NCH][CVVDP_CSF_NODES];
};

// EDGE (round 5): the strips that touch the image's left or right border, W % 4 == 0 (band4f.hip EDGE == 1: zero masks of the reduce's
// padding, the first / last column's extra taps, coarse-column replicas by lane reads, the blur's reflect padding written as mirrors,
// lanes outside the image masked out).  The front waves have the registers for it, so these strips keep the hand-managed loads and
// the front / back layout -- until round 4 they ran k_band4f<4, 1> (245 VGPRs, two waves per SIMD) as a second launch beside this
// kernel, and at levels 1 and 2 that launch took as long as all other strips together (profiles/r04_dev_notes.txt 10).
template <bool HEAT, bool FEAT, bool EDGE>
__device__ __forceinline__ void band4s_body(const BandArgs& a, S4Lds<HEAT, FEAT>& L, const int strip, const int seg, const int item) {
  constexpr int NCH = 4, NP = 8;
  auto& s_h = L.s_h; auto& s_fd = L.s_fd; auto& s_ve = L.s_ve; auto& s_g = L.s_g; auto& s_lum = L.s_lum; auto& s_S = L.s_S;
  auto& s_m = L.s_m; auto& s_q = L.s_q; auto& s_d = L.s_d; auto& s_lut = L.s_lut;

  const int t = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool front = wv < NCH;
  const int c = wv & (NCH - 1);
  const int j = t & 63;
  const int H = a.H, W = a.W, Hc = a.Hc, Wc = a.Wc;
  const int x0 = strip * S_SW;
  const int fc0 = x0 - S_HALO + 4 * j;
  const bool in_img = !EDGE || (fc0 >= 0 && fc0 < W);
  const bool edge_r = EDGE && x0 + S_SW + S_HALO >= W;      // (the strip reaches column W-1: band4f.hip band4f_right_edge_strips)
  const bool interior = j >= 2 && j < 62 && (!EDGE || fc0 < W);
  const int cb = (x0 - S_HALO) / 2;
  const int ys = seg * a.seg_h, ye = min(H, ys + a.seg_h);
  const bool top_seg = seg == 0;
  // The top segment starts at row 0 (its six reflected rows are filled in by symmetry): the window position of its first row is 6
  const int r_start = top_seg ? 0 : ys - S_R;
  const int rend = ye + S_R;
  const int rreal = min(rend, H);

  for (int i = t; i < NCH * CVVDP_CSF_NODES; i += 512) {
    const int cc = i / CVVDP_CSF_NODES, k = i - cc * CVVDP_CSF_NODES;
    const float l0 = a.lut[cc * CVVDP_CSF_NODES + k] * kLog2_10 + fast_log2(a.sens_mul * a.ch_gain[cc] * a.band_mul);
    const float l1 = a.lut[cc * CVVDP_CSF_NODES + min(k + 1, CVVDP_CSF_NODES - 1)] * kLog2_10 + fast_log2(a.sens_mul * a.ch_gain[cc] * a.band_mul);
    s_lut[cc][k] = make_float2(l0, l1 - l0);
  }
  for (int i = t; i < 2 * NP * (S_VE / 2); i += 512) (&s_ve[0][0][0])[i] = make_float2(0.0f, 0.0f);
  __syncthreads();

#ifndef S_DIAG_BACK_ONLY
  if (front) {
    // =============================================================================================== FRONT: stream, reduce, expand rows
#ifdef S_PRIO_FRONT
    __builtin_amdgcn_s_setprio(S_PRIO_FRONT);
#endif
    const int64_t P = (int64_t)H * W, Pc = (int64_t)Hc * Wc;
    const int64_t gps = (int64_t)a.items_cap * P, gcps = (int64_t)a.items_cap_c * Pc;
    const float* gT = a.g + (int64_t)item * P + (2 * c) * gps;
    const float* gR = gT + gps;
    float* g1T = a.g1_out + (int64_t)item * Pc + (2 * c) * gcps;     // level l+1 planes of this channel, written here
    float* g1R = g1T + gcps;
    const float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
    const float ind_k1 = a.ind_k1, ind_k0 = a.ind_k0;
    const float rk0 = a.rk[0], rk1 = a.rk[1], rk2 = a.rk[2], rk3 = a.rk[3], rk4 = a.rk[4];
    // own four columns; the neighbour samples (columns fc0-2, fc0-1 and fc0+4) sit at immediate offsets -8 / +16 from them: the strips
    // that run here stay clear of the image's left / right border (x0 >= 240, x0 + 248 <= W), so nothing is clamped or masked.  Column
    // fc0+4 of the last lane may be column W, i.e. the next row's first sample: it only enters coarse columns beyond the blur halo.
    // EDGE: addresses are clamped into the row (samples outside the image are the reference's zero padding: the loaded values are
    // multiplied by 0 in consume), so the neighbour samples need offset registers of their own
    const uint32_t goff = (uint32_t)(EDGE ? min(max(fc0, 0), W - 4) : fc0) * 4u;
    const uint32_t loff = EDGE ? (uint32_t)min(max(fc0 - 2, 0), W - 2) * 4u : goff;       // columns fc0-2, fc0-1 (EDGE == 0: goff - 8 as an immediate)
    const uint32_t roff = EDGE ? (uint32_t)min(max(fc0 + 4, 0), W - 1) * 4u : goff;       // column fc0+4 (EDGE == 0: goff + 16)
    const int lane_first = 2;                                        // strip 0: the lane of fine column 0
    const int lane_last = (W - 4 - (x0 - S_HALO)) >> 2;              // edge_r strips: the lane of the last four fine columns (0 .. 63)

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

    // One level-l row (four own samples + three neighbours per plane) through the horizontal pass, then into the running sums
    // (band4f.hip consume, EDGE == 0).  a_row: its index (scalar).  Even rows complete coarse row a_row/2 - 1 -> emitted.
    auto consume = [&](int a_row, auto odd_a, v4f vT, v4f vR, v2f lT, float rT, v2f lR, float rR, float4& emitted) {
      if constexpr (EDGE) {
        // Zero padding left / right of the image (band4f.hip consume, EDGE == 1: there every sample is multiplied by a lane mask).  Only
        // two lanes hold coarse columns INSIDE the image that see samples outside it -- the lane of column 0 (its left neighbours) and the
        // lane of columns W-4 .. W-1 (its right neighbour); the coarse columns of the lanes outside the image are replaced by replicas
        // below, whatever they were.  So two exec-masked moves instead of masks: no per-lane constants live across the row loop.
        if (strip == 0 && j == lane_first) { lT = v2f{0.0f, 0.0f}; lR = v2f{0.0f, 0.0f}; }
        if (edge_r && j == lane_last) { rT = 0.0f; rR = 0.0f; }
      }
      float4 hr;
      hr.x = __builtin_fmaf(vT.z, rk4, __builtin_fmaf(vT.y, rk3, __builtin_fmaf(vT.x, rk2, __builtin_fmaf(lT.y, rk1, lT.x * rk0))));
      hr.y = __builtin_fmaf(rT, rk4, __builtin_fmaf(vT.w, rk3, __builtin_fmaf(vT.z, rk2, __builtin_fmaf(vT.y, rk1, vT.x * rk0))));
      hr.z = __builtin_fmaf(vR.z, rk4, __builtin_fmaf(vR.y, rk3, __builtin_fmaf(vR.x, rk2, __builtin_fmaf(lR.y, rk1, lR.x * rk0))));
      hr.w = __builtin_fmaf(rR, rk4, __builtin_fmaf(vR.w, rk3, __builtin_fmaf(vR.z, rk2, __builtin_fmaf(vR.y, rk1, vR.x * rk0))));
      if constexpr (EDGE) {
        // first / last output column (lpyr_dec.py:205-209; the last column's extra taps depend on the ROW parity, sic): the same two lanes
        if (strip == 0 && j == lane_first) {
          hr.x = __builtin_fmaf(vT.y, rk0, __builtin_fmaf(vT.x, rk1, hr.x));
          hr.z = __builtin_fmaf(vR.y, rk0, __builtin_fmaf(vR.x, rk1, hr.z));
        }
        if (edge_r && j == lane_last) {                               // columns W-2, W-1: the lane's second coarse column
          const float wr3 = (H & 1) ? rk3 : rk4, wr2 = (H & 1) ? rk4 : 0.0f;
          hr.y = __builtin_fmaf(vT.z, wr2, __builtin_fmaf(vT.w, wr3, hr.y));
          hr.w = __builtin_fmaf(vR.z, wr2, __builtin_fmaf(vR.w, wr3, hr.w));
        }
      }
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
        if constexpr (EDGE) {                                        // coarse columns outside the image: replicas of column 0 / Wc-1
          if (strip == 0) {
            const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, emitted.x), lane_first));
            const float r0_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, emitted.z), lane_first));
            if (fc0 < 0) emitted = make_float4(t0, t0, r0_, r0_);
          }
          if (edge_r) {
            const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, emitted.y), lane_last));
            const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, emitted.w), lane_last));
            if (fc0 >= W) emitted = make_float4(t1, t1, r1, r1);
          }
        }
        // level l+1 belongs to the lanes that own its columns (interior of the strip) in the segment that owns its rows
        const int own_end = seg == a.n_seg - 1 ? Hc : (ye >> 1);
        if (interior && m1 >= (ys >> 1) && m1 < own_end) {
          const int64_t o = (int64_t)m1 * Wc + (cb + 2 * j);
#if defined(S_DIAG_NO_STORE)
          if (emitted.x == 1.2345e-30f) { *reinterpret_cast<v2f*>(g1T + o) = v2f{emitted.x, emitted.y}; }
#elif defined(S_DIAG_PLAIN_STORE)
          *reinterpret_cast<v2f*>(g1T + o) = v2f{emitted.x, emitted.y};
          *reinterpret_cast<v2f*>(g1R + o) = v2f{emitted.z, emitted.w};
#else
          __builtin_nontemporal_store(v2f{emitted.x, emitted.y}, reinterpret_cast<v2f*>(g1T + o));
          __builtin_nontemporal_store(v2f{emitted.z, emitted.w}, reinterpret_cast<v2f*>(g1R + o));
#endif
        }
      }
    };

    // per-column luminance terms of one row, shared by all channels (band4.hip lum_prep): front thread t = column t
    const bool lodd = t & 1;
    const float lwa = lodd ? 0.0f : e0, lwb = lodd ? eo : e1, lwc = lodd ? eo : e0;
    auto lum_prep = [&](int buf) {
      const float* yT = reinterpret_cast<const float*>(&s_ve[buf][0][0]);
      const float* yR = reinterpret_cast<const float*>(&s_ve[buf][1][0]);
      const int col = t;
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
    };

    // ---- STREAM LOADS (band4f.hip): issued from inline assembly, waited for with one exact count per step.
    // The front wave alone at two rows in flight ran at 5.4 ms for level 0 of 4K x 64 (profiles/r04_dev_notes.txt 1): each wave
    // waited for memory it had asked for one phase earlier.  The ring therefore holds S_RING rows (the front has the registers: 77 at
    // six rows): row s+1 leaves for the back in phase 1 of step s, and its slot takes row s+S_RING+1 in phase 2, so rows s+6 ..
    // s+S_RING are in flight during step s; the neighbour samples (columns fc0-2, fc0-1, fc0+4) have two sets by row parity and
    // are requested two steps before their row is consumed.
    //   phase 2 of step s:  neighbour samples of row s+7 (4 loads) -> set (s+7) & 1,  rows s+S_RING+1 of the two planes (2 loads)
    //   start of step s:    needs ring slot (s+5) % S_RING and neighbour set (s+5) & 1; younger than those: the 2 row loads of step
    //                       s-2, the 6 loads of step s-1 -> vmcnt(8)   (stores count too: that only makes the wait stricter)
    // EDGE: a ring of six rows (rows s+6, s+7 in flight during step s, as in k_band4f) -- the border code costs the front ~14 registers
    // and the kernel must stay under 128 for four waves per SIMD.  In general: at the start of step s the youngest loads that must have
    // landed are the row loads of step s-(S_RING-4)/2 ... which leaves exactly S_RING younger ones -> vmcnt(S_RING)
#ifdef S_DIAG_EDGE_RING     // A/B: the border body's ring length
    constexpr int S_RING = EDGE ? S_DIAG_EDGE_RING : CVVDP_BAND4S_RING;
#else
    constexpr int S_RING = EDGE ? 6 : CVVDP_BAND4S_RING;
#endif
    static_assert(S_RING >= 6 && S_RING % 2 == 0, "the row step's wait count assumes the ring slot of row s+5 and the neighbour set of row s+5 are older than S_RING loads");
    v4f ringT[S_RING], ringR[S_RING];
    v2f nbLT[2], nbLR[2];
    float nbRT[2], nbRR[2];
#pragma unroll
    for (int i = 0; i < S_RING; ++i) { ringT[i] = 0.0f; ringR[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { nbLT[i] = 0.0f; nbLR[i] = 0.0f; nbRT[i] = 0.0f; nbRR[i] = 0.0f; }
#ifdef CVVDP_SAFE_LOADS
    // `make safe`: the same kernel with ordinary loads the compiler tracks and waits for itself (the dynamic check of the hand-managed ones)
#define S_ROWPTR(plane, row, off) (reinterpret_cast<const char*>((plane) + (int64_t)(row) * W) + (off))
    struct __attribute__((packed, aligned(4))) u4 { float x, y, z, w; };     // (rows of W % 4 == 2 frames are 8-byte aligned)
    struct __attribute__((packed, aligned(4))) u2 { float x, y; };
#define S_LOAD4(dst, off, plane, row) do { const u4 q_ = *reinterpret_cast<const u4*>(S_ROWPTR(plane, row, off)); dst = v4f{q_.x, q_.y, q_.z, q_.w}; } while (0)
#define S_LOADL(dst, plane, row) do { const u2 q_ = *reinterpret_cast<const u2*>(S_ROWPTR(plane, row, loff) - (EDGE ? 0 : 8)); dst = v2f{q_.x, q_.y}; } while (0)
#define S_LOADR(dst, plane, row) dst = *reinterpret_cast<const float*>(S_ROWPTR(plane, row, roff) + (EDGE ? 0 : 16))
#define S_WAIT8(...) do { } while (0)
#else
#define S_LOAD4(dst, off, plane, row) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(off), "s"((plane) + (int64_t)(row) * W))
#define S_LOADL(dst, plane, row) do { if constexpr (EDGE) asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(dst) : "v"(loff), "s"((plane) + (int64_t)(row) * W)); \
    else asm volatile("global_load_dwordx2 %0, %1, %2 offset:-8" : "+v"(dst) : "v"(goff), "s"((plane) + (int64_t)(row) * W)); } while (0)
#define S_LOADR(dst, plane, row) do { if constexpr (EDGE) asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(roff), "s"((plane) + (int64_t)(row) * W)); \
    else asm volatile("global_load_dword %0, %1, %2 offset:16" : "+v"(dst) : "v"(goff), "s"((plane) + (int64_t)(row) * W)); } while (0)
#ifdef S_DIAG_NO_NB
#define S_WAIT8(a0, a1, a2, a3, a4, a5) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "n"(S_RING - 4))
#else
#define S_WAIT8(a0, a1, a2, a3, a4, a5) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "n"(S_RING))
#endif
#endif
    // (the drain names every register a load may still be heading for: without that use the compiler sees the last rows' loads as dead
    // values and hands their registers to temporaries while the loads are in flight)
    auto drain = [&]() {
#ifndef CVVDP_SAFE_LOADS
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(nbLT[0]), "+v"(nbLT[1]), "+v"(nbLR[0]), "+v"(nbLR[1]), "+v"(nbRT[0]), "+v"(nbRT[1]), "+v"(nbRR[0]), "+v"(nbRR[1]));
#pragma unroll
      for (int i = 0; i < S_RING; ++i) asm volatile("" : "+v"(ringT[i]), "+v"(ringR[i]));
#endif
    };
    auto rowc = [&](int r) { return min(max(r, 0), H - 1); };       // rows outside the image: any valid row (their weight is 0)
    auto load_nb = [&](auto p_, int row) {
      constexpr int PS = decltype(p_)::value;
      (void)&nbLT; (void)&nbLR; (void)&nbRT; (void)&nbRR; (void)&goff; (void)&loff; (void)&roff; (void)&gT; (void)&gR; (void)&W;   // (asm operands alone do not capture)
      S_LOADL(nbLT[PS], gT, row);
      S_LOADR(nbRT[PS], gT, row);
      S_LOADL(nbLR[PS], gR, row);
      S_LOADR(nbRR[PS], gR, row);
    };

    // ---- prologue: rows r_start-4 .. r_start+4 prime the reduce (three complete coarse rows in the window, two partial ones),
    // rows r_start+1 .. r_start+4 stay in ring slots 1 .. 4, the rows after them are requested in the order the steps' count assumes
    {
      auto ld4 = [&](const float* plane, int r) -> v4f {
        return *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(plane + (int64_t)rowc(r) * W) + goff);
      };
      auto ld2 = [&](const float* plane, int r) -> v2f { return *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(plane + (int64_t)rowc(r) * W) + loff - (EDGE ? 0 : 8)); };
      auto ld1 = [&](const float* plane, int r) -> float { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(plane + (int64_t)rowc(r) * W) + roff + (EDGE ? 0 : 16)); };
      float4 em = cC;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int ar = r_start - 4 + i;                               // r_start is even: i even <-> row even
        const v4f vT = ld4(gT, ar), vR = ld4(gR, ar);
        const v2f lT = ld2(gT, ar), lR = ld2(gR, ar);
        const float rT = ld1(gT, ar), rR = ld1(gR, ar);
        if (i == 4) {                                                 // raw row r_start goes straight to the back
          *reinterpret_cast<v4f*>(&s_g[0][2 * c][4 * j]) = vT;
          *reinterpret_cast<v4f*>(&s_g[0][2 * c + 1][4 * j]) = vR;
        }
        if (i > 4) { ringT[i - 4] = vT; ringR[i - 4] = vR; }           // rows r_start+1 .. r_start+4 in slots 1 .. 4 (the ring holds the rows as loaded)
        if (i & 1) {
          consume(ar, std::true_type{}, vT, vR, lT, rT, lR, rR, em);
        } else {
          consume(ar, std::false_type{}, vT, vR, lT, rT, lR, rR, em);
          cA = cB; cB = cC; cC = em;
        }
      }
      if (r_start == 0) cA = cB;                                      // coarse row -1 does not exist: the expand clamps to row 0
#pragma unroll
      for (int k = 5; k <= S_RING - 2; ++k) {
        S_LOAD4(ringT[k], goff, gT, rowc(r_start + k));
        S_LOAD4(ringR[k], goff, gR, rowc(r_start + k));
      }
      load_nb(std::integral_constant<int, 1>{}, rowc(r_start + 5));
      S_LOAD4(ringT[S_RING - 1], goff, gT, rowc(r_start + S_RING - 1));
      S_LOAD4(ringR[S_RING - 1], goff, gR, rowc(r_start + S_RING - 1));
      load_nb(std::integral_constant<int, 0>{}, rowc(r_start + 6));
      S_LOAD4(ringT[0], goff, gT, rowc(r_start + S_RING));
      S_LOAD4(ringR[0], goff, gR, rowc(r_start + S_RING));
      coarse_finish(0, std::false_type{}, false, cC);                 // vertical expand of row r_start (even)
    }
    __syncthreads();
    lum_prep(0);
    __syncthreads();

    // one real row r (0 <= r < H); U = (r - r_start) mod S_RING = its ring slot
    auto step = [&](int r, auto u_) {
      constexpr int U = decltype(u_)::value;
      constexpr bool ODD = (U & 1) != 0;
      constexpr int S5 = (U + 5) % S_RING, S1 = (U + 1) % S_RING, P5 = (U + 5) & 1;
      (void)&ringT; (void)&ringR; (void)&nbLT; (void)&nbLR; (void)&nbRT; (void)&nbRR; (void)&goff; (void)&loff; (void)&roff; (void)&gT; (void)&gR; (void)&W;
      S_WAIT8(ringT[S5], ringR[S5], nbLT[P5], nbLR[P5], nbRT[P5], nbRR[P5]);
      // ================= phase 1: level-l row r+5 into the reduce; an even row completes a coarse row, which rolls the window for row r+1
      float4 emitted = cC;
      consume(r + 5, std::integral_constant<bool, !ODD>{}, ringT[S5], ringR[S5], nbLT[P5], nbRT[P5], nbLR[P5], nbRR[P5], emitted);
      coarse_finish(ODD ? 0 : 1, std::integral_constant<bool, !ODD>{}, ODD, emitted);   // vertical expand of row r+1
#ifndef S_DIAG_NO_SG
      *reinterpret_cast<v4f*>(&s_g[ODD ? 0 : 1][2 * c][4 * j]) = ringT[S1];             // raw row r+1 for the back (arrived long ago)
      *reinterpret_cast<v4f*>(&s_g[ODD ? 0 : 1][2 * c + 1][4 * j]) = ringR[S1];
#endif
      S_SYNC();
      // ================= phase 2
      {
#ifndef S_DIAG_NO_NB
        load_nb(std::integral_constant<int, P5>{}, rowc(r + 7));      // set (r+7) & 1 == (r+5) & 1: consumed in phase 1
#endif
        const int rn = rowc(r + S_RING + 1);
        S_LOAD4(ringT[S1], goff, gT, rn);                             // (r + S_RING + 1) % S_RING == (r + 1) % S_RING: the slot that has just been handed on
        S_LOAD4(ringR[S1], goff, gR, rn);
      }
#ifndef S_DIAG_NO_LUM
      lum_prep(ODD ? 0 : 1);
#endif
      S_SYNC();
    };

    // Whole groups of S_RING rows without an exit inside (a `break` in the unrolled body makes the structuriser route every exit
    // through the loop's latch, and the layout-order ISA check -- tools/check_band4_isa.py -- then sees paths "exit -> latch ->
    // header" that no execution takes); the last 0 .. S_RING-1 rows of the march follow as straight-line code (nested ifs).
    int r = r_start;
    for (; r + S_RING <= rreal; r += S_RING)
      s_for_seq([&](auto u_) { step(r + decltype(u_)::value, u_); }, std::make_integer_sequence<int, S_RING>{});
    auto tail = [&](auto self, int rr, auto u_) -> void {
      constexpr int U = decltype(u_)::value;
      if (rr < rreal) {
        step(rr, u_);
        if constexpr (U + 1 < S_RING - 1) self(self, rr + 1, std::integral_constant<int, U + 1>{});
      }
    };
    tail(tail, r, std::integral_constant<int, 0>{});
    drain();
    for (r = rreal; r < rend; ++r) {          // reflected rows below the image: the back works from its window, the front only keeps step
      S_SYNC();
      S_SYNC();
    }
    if constexpr (HEAT) __syncthreads();      // (the back's epilogue: channel terms of the last pooled row -> its heat-map band)
#undef S_LOAD4
#undef S_LOADL
#undef S_LOADR
#undef S_WAIT8
#ifdef CVVDP_SAFE_LOADS
#undef S_ROWPTR
#endif
  }
#endif
#ifndef S_DIAG_FRONT_ONLY
  if (!front) {
    // =============================================================================================== BACK: contrast, blurs, masking, pooling
#ifdef S_PRIO_BACK
    __builtin_amdgcn_s_setprio(S_PRIO_BACK);
#endif
    const float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
    const float mask_p = a.mask_p, eps_p = a.eps_p;
    const float qc = a.q[c];
    const float xw0 = a.xw[0 * 4 + c], xw1 = a.xw[1 * 4 + c], xw2 = a.xw[2 * 4 + c], xw3 = a.xw[3 * 4 + c];
    const float m1c = a.m1[c];
    const float inv_dmax = a.inv_dmax;
    const float hw_c = a.hw[c], beta_tch = a.beta_tch, eps_btch = a.eps_btch, inv_beta_tch = 1.0f / a.beta_tch, eps_inv_btch = a.eps_inv_btch;
    // ---- FEATURES (band4.hip): two independent trackers -- |T'|, |R'| belong to row r of the contrast stage, D to the pooled row
    v2f f_t[2] = {0.0f, 0.0f}, f_t2[2] = {0.0f, 0.0f}, f_r[2] = {0.0f, 0.0f}, f_r2[2] = {0.0f, 0.0f};   // (column pairs: packed adds / FMAs)
    const float zero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int f_left_tr = 0, f_left_d = 0;                  // rows to the next cell-row boundary (scalar)
    if constexpr (FEAT) {
      f_left_tr = f_left_d = a.fs - ys % a.fs;
      s_lds_write4(&s_fd[0][c][4 * j], zero4);
      s_lds_write4(&s_fd[1][c][4 * j], zero4);
    }
    auto feat_store = [&](int y_last, int q0, const float (&s0)[4], const float (&s1)[4]) {   // column sums of the piece that ends with row y_last
      if constexpr (FEAT) {
        if (interior) {
          const int piece = y_last / a.fs + seg;
          float* dst = a.fsum + ((((int64_t)item * NCH + c) * a.f_pieces + piece) * 6 + q0) * W + (x0 - S_HALO + 4 * j);
          *reinterpret_cast<s_u4*>(dst) = s_u4{s0[0], s0[1], s0[2], s0[3]};
          *reinterpret_cast<s_u4*>(dst + W) = s_u4{s1[0], s1[1], s1[2], s1[3]};
        }
      }
    };
    auto feat_d_row = [&](int yprev) {                // D sums: a cell row ends with row yprev (the segment's last row is the epilogue's)
      if constexpr (FEAT) {
        if (yprev >= ys && --f_left_d == 0) {
          const sf4 fd = s_lds_read4(&s_fd[0][c][4 * j]), fd2 = s_lds_read4(&s_fd[1][c][4 * j]);
          feat_store(yprev, 4, fd.v, fd2.v);
          s_lds_write4(&s_fd[0][c][4 * j], zero4);
          s_lds_write4(&s_fd[1][c][4 * j], zero4);
          f_left_d = a.fs;
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

    // vertical-blur window (band4.hip): slot s of column i = one horizontally blurred row, weights rotate instead of the data
    typedef float v32f __attribute__((ext_vector_type(32)));
    v32f winA = 0.0f, winB = 0.0f;
    const v2f E0 = {a.blur_h[0], a.blur_h[1]}, E1 = {a.blur_h[2], a.blur_h[3]}, E2 = {a.blur_h[4], a.blur_h[5]}, E3 = {a.blur_h[6], a.blur_h[5]};
    const v2f O0 = {a.blur_h[1], a.blur_h[2]}, O1 = {a.blur_h[3], a.blur_h[4]}, O2 = {a.blur_h[5], a.blur_h[6]};
#define S_SWAP(p) __builtin_shufflevector(p, p, 1, 0)
    const v2f be[6] = {E0, E1, E2, E3, S_SWAP(O1), S_SWAP(O0)};
    const v2f bo[6] = {O0, O1, O2, S_SWAP(E2), S_SWAP(E1), S_SWAP(E0)};
#undef S_SWAP
    const float b0 = a.blur_h[0], b12 = a.blur_h[0];
    const int n0 = top_seg ? S_R : 0;
    float wr[S_BW];
#pragma unroll
    for (int k = 0; k < S_BW; ++k) wr[k] = a.blur[(k + 2 * S_BW - 1 - n0) % S_BW];
    float acc = 0.0f;

    // pooling stage of centre row y (band4.hip stage3c)
    auto stage3c = [&](int k7) {
      const sf4 q0 = s_lds_read4(&s_q[0][4 * j]), q1 = s_lds_read4(&s_q[1][4 * j]), q2 = s_lds_read4(&s_q[2][4 * j]);
      const sf4 q3 = s_lds_read4(&s_q[3][4 * j]);
      const sf4 d = s_lds_read4(&s_d[k7][c][4 * j - S_HALO]);
      float De[4], Dh[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const v2f Q0 = {q0.v[2 * h], q0.v[2 * h + 1]}, Q1 = {q1.v[2 * h], q1.v[2 * h + 1]}, Q2 = {q2.v[2 * h], q2.v[2 * h + 1]}, Q3 = {q3.v[2 * h], q3.v[2 * h + 1]};
        const v2f M1 = Q3 * xw3 + (Q2 * xw2 + (Q1 * xw1 + (Q0 * xw0 + m1c)));
        const v2f X = {fast_pow(d.v[2 * h], mask_p) - eps_p, fast_pow(d.v[2 * h + 1], mask_p) - eps_p};
        const v2f T = X * inv_dmax + M1;
        const float r0 = fast_rcp(T.x), r1 = fast_rcp(T.y);
        De[2 * h] = __builtin_fmaf(X.x, r0, kEps); De[2 * h + 1] = __builtin_fmaf(X.y, r1, kEps);
        if constexpr (HEAT || FEAT) { Dh[2 * h] = X.x * r0; Dh[2 * h + 1] = X.y * r1; }
      }
      if constexpr (FEAT) {
        sf4 fd = s_lds_read4(&s_fd[0][c][4 * j]), fd2 = s_lds_read4(&s_fd[1][c][4 * j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { fd.v[i] += Dh[i]; fd2.v[i] = __builtin_fmaf(Dh[i], Dh[i], fd2.v[i]); }
        s_lds_write4(&s_fd[0][c][4 * j], fd.v);
        s_lds_write4(&s_fd[1][c][4 * j], fd2.v);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_fmaf(De[i], De[i], acc);   // sum of (D + eps)^2; k_finalize takes the eps^2 off
      if constexpr (HEAT) {   // this channel's term of the per-pixel channel norm (cvvdp_metric.py:728-734; band4.hip stage3c)
        float ht[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ht[i] = fast_pow(Dh[i] * hw_c + kEps, beta_tch) - eps_btch;
        s_lds_write4(&s_h[c][4 * j], ht);
      }
    };
    // heat-map band of row y from the channel terms published by stage3c (a barrier in between): one column per back thread, lp_norm
    // over the channels, stored / band_mul as lpyr_dec_2.set_lband does (lpyr_dec.py:308-314; band4.hip heat_row)
    auto heat_row = [&](int y) {
      if constexpr (HEAT) {
        const int col = t - 64 * NCH;
        if (y >= ys && col >= S_HALO && col < 256 - S_HALO && (!EDGE || x0 - S_HALO + col < W)) {
          const float sum = (s_h[0][col] + s_h[1][col] + s_h[2][col]) + s_h[3][col];
          a.dchr[(int64_t)item * ((int64_t)H * W) + (int64_t)y * W + (x0 - S_HALO + col)] = (fast_pow(sum + kEps, inv_beta_tch) - eps_inv_btch) / a.band_mul;
        }
      }
    };
    // vertical 13-tap blur of the window -> Mq = (blur + eps)^q -> s_q; then the weights rotate
    auto vblur = [&](int yc) {
      if (interior && yc >= ys) {
#ifdef S_DIAG_VB4      // timing variant: four dependent chains instead of two (results differ in the last bit)
        v2f va = kEps, vb = kEps, va1 = 0.0f, vb1 = 0.0f;
#pragma unroll
        for (int sdx = 0; sdx < S_BW; ++sdx) {
          const v2f wa = {winA[2 * sdx], winA[2 * sdx + 1]}, wb = {winB[2 * sdx], winB[2 * sdx + 1]};
          const v2f ww = {wr[sdx], wr[sdx]};
          if (sdx & 1) { va1 += ww * wa; vb1 += ww * wb; } else { va += ww * wa; vb += ww * wb; }
        }
        va += va1; vb += vb1;
#else
        v2f va = kEps, vb = kEps;
#pragma unroll
        for (int sdx = 0; sdx < S_BW; ++sdx) {
          const v2f wa = {winA[2 * sdx], winA[2 * sdx + 1]}, wb = {winB[2 * sdx], winB[2 * sdx + 1]};
          const v2f ww = {wr[sdx], wr[sdx]};
          va += ww * wa; vb += ww * wb;
        }
#endif
        const float v[4] = {va.x, va.y, vb.x, vb.y};
        float Mq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) Mq[i] = fast_pow(v[i], qc);
        s_lds_write4(&s_q[c][4 * j], Mq);
      }
      const float last = wr[S_BW - 1];
#pragma unroll
      for (int k = S_BW - 1; k > 0; --k) wr[k] = wr[k - 1];
      wr[0] = last;
    };

    __syncthreads();        // (front: prologue -> s_ve[0], s_g[0])
    __syncthreads();        // (front: luminance terms of row r_start)

    int slot = n0, k7 = 0;
    uint32_t r_mn = 0x7F7FFFFFu, r_mx = 0u;
    // one real row r (0 <= r < H)
    auto step = [&](int r, auto odd_) {
      constexpr bool ODD = decltype(odd_)::value;
      // ================= phase 1
      const int yprev = r - 1 - S_R;
      if (interior && yprev >= ys) stage3c(k7);
      feat_d_row(yprev);
      const bool feat_row = FEAT && r >= ys && r < ye;  // (scalar) row r belongs to this segment: its |T'|, |R'| are counted
      if (in_img) {                                     // (EDGE == 0: every lane)
        float exT[4], exR[4];
        expand4(s_ve[ODD][2 * c], exT);
        expand4(s_ve[ODD][2 * c + 1], exR);
        const sf4 rLt = s_lds_read4(&s_lum[0][4 * j]), rLr = s_lds_read4(&s_lum[1][4 * j]);
        const sf4 Sv = s_lds_read4(&s_S[c][4 * j]);
        const sf4 gt = s_lds_read4(&s_g[ODD][2 * c][4 * j]), gr = s_lds_read4(&s_g[ODD][2 * c + 1][4 * j]);
        if constexpr (HEAT) {
          // colour-mapped modes: the range of the context image (test Y-sustained = this level's plane 0, k_heat_range) from the raw rows the
          // back wave of channel 0 is handed anyway -- every sample of the plane passes through some block's wave (the halo columns are
          // samples of the neighbouring strips): minimum over the positive samples and maximum, as bit patterns
          if (c == 0 && a.hstats) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t u = __float_as_uint(gt.v[i]);
              r_mn = min(r_mn, gt.v[i] > 0.0f ? u : 0x7F7FFFFFu);
              r_mx = max(r_mx, gt.v[i] > 0.0f ? u : 0u);
            }
          }
        }
        float m[4], d[4], at4[4], ar4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float S = Sv.v[i];
          const float ct = fminf((gt.v[i] - exT[i]) * rLt.v[i], 1000.0f);            // lpyr_dec.py:402 (band gain :66 is in S)
          const float cr = fminf((gr.v[i] - exR[i]) * rLr.v[i], 1000.0f);
          if constexpr (FEAT) {
            const float at = fabsf(ct) * S, ar = fabsf(cr) * S;                    // |T'|, |R'| (the channel gain inside S is divided out by k_feature_finish)
            m[i] = fminf(at, ar);                                                  // = min(|ct|,|cr|)*S bit for bit (rounding is monotone)
            at4[i] = at; ar4[i] = ar;
          } else {
            m[i] = fminf(fabsf(ct), fabsf(cr)) * S;                                // min(|T'|,|R'|), T' = ct*S (cvvdp_metric.py:845)
          }
          d[i] = fabsf(ct - cr) * S + kEps;                                        // |T'-R'| + eps (:855, safe_pow)
        }
        s_lds_write4(&s_m[c][4 * j], m);
        if (interior) s_lds_write4(&s_d[k7][c][4 * j - S_HALO], d);
        if constexpr (FEAT) {
          if (feat_row) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const v2f at2 = {at4[2 * h], at4[2 * h + 1]}, ar2 = {ar4[2 * h], ar4[2 * h + 1]};
              f_t[h] += at2; f_t2[h] += at2 * at2; f_r[h] += ar2; f_r2[h] += ar2 * ar2;
            }
          }
        }
        if constexpr (EDGE) {
          // the blur's reflect padding at the left / right image border (band4f.hip, mirror roles): the lanes of columns 1..6 / W-7..W-2
          // write their samples a second time at the mirrored positions of this wave's s_m row (column x <-> -x, 2(W-1)-x)
          if (strip == 0) {
            if (j == 2) { float* dst = &s_m[c][S_HALO - 1]; dst[0] = m[1]; dst[-1] = m[2]; dst[-2] = m[3]; }          // columns 1, 2, 3 -> -1, -2, -3
            if (j == 3) { float* dst = &s_m[c][S_HALO - 4]; dst[0] = m[0]; dst[-1] = m[1]; dst[-2] = m[2]; }          // columns 4, 5, 6 -> -4, -5, -6
          }
          if (edge_r) {
            const int ll = (W - 4 - (x0 - S_HALO)) >> 2;              // the lane of columns W-4 .. W-1 (scalar)
            const int e = 2 * (W - 1) - (x0 - S_HALO);                // s_m index of the mirror of column 0 + the column
            if (j == ll - 1 && e - (W - 7) < 256) { float* dst = &s_m[c][e - (W - 7)]; dst[0] = m[1]; dst[-1] = m[2]; dst[-2] = m[3]; }   // W-7, W-6, W-5
            if (j == ll && e - (W - 4) < 256) { float* dst = &s_m[c][e - (W - 4)]; dst[0] = m[0]; dst[-1] = m[1]; dst[-2] = m[2]; }       // W-4, W-3, W-2
          }
        }
      }
      if constexpr (FEAT) {
        if (feat_row && (--f_left_tr == 0 || r == ye - 1)) {
          const float t0[4] = {f_t[0].x, f_t[0].y, f_t[1].x, f_t[1].y}, t1[4] = {f_t2[0].x, f_t2[0].y, f_t2[1].x, f_t2[1].y};
          const float r0[4] = {f_r[0].x, f_r[0].y, f_r[1].x, f_r[1].y}, r1[4] = {f_r2[0].x, f_r2[0].y, f_r2[1].x, f_r2[1].y};
          feat_store(r, 0, t0, t1);
          feat_store(r, 2, r0, r1);
#pragma unroll
          for (int h = 0; h < 2; ++h) { f_t[h] = 0.0f; f_t2[h] = 0.0f; f_r[h] = 0.0f; f_r2[h] = 0.0f; }
          f_left_tr = a.fs;
        }
      }
      // s_m[c] is this wave's own row: its LDS operations execute in order, the horizontal blur needs no block barrier
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
        if (top_seg && r >= 1 && r <= S_R) {              // rows 1..6 of the image are also its reflected rows -1..-6 (window slots 5..0)
          const int ms = S_R - r;
          winA[2 * ms] = h[0]; winA[2 * ms + 1] = h[1]; winB[2 * ms] = h[2]; winB[2 * ms + 1] = h[3];
        }
      }
      S_SYNC();
      // ================= phase 2
      heat_row(yprev);
      vblur(r - S_R);
      slot = slot == S_BW - 1 ? 0 : slot + 1;
      k7 = k7 == S_R ? 0 : k7 + 1;
      S_SYNC();
    };
    // one reflected row below the image (r >= H): its horizontally blurred row is that of row 2(H-1) - r, still in the window
    auto tail_step = [&](int r) {
      const int yprev = r - 1 - S_R;
      if (interior && yprev >= ys) stage3c(k7);
      feat_d_row(yprev);
      S_SYNC();
      heat_row(yprev);
      if (interior) {
        const int back = 2 * (r - (H - 1));                           // 2, 4, .. 12 rows back
        const int src = slot >= back ? slot - back : slot - back + S_BW;
        const float h0 = winA[2 * src], h1 = winA[2 * src + 1], h2 = winB[2 * src], h3 = winB[2 * src + 1];
        winA[2 * slot] = h0; winA[2 * slot + 1] = h1; winB[2 * slot] = h2; winB[2 * slot + 1] = h3;
      }
      vblur(r - S_R);
      slot = slot == S_BW - 1 ? 0 : slot + 1;
      k7 = k7 == S_R ? 0 : k7 + 1;
      S_SYNC();
    };

    int r = r_start;                                       // (even)
    for (; r < rreal; r += 2) {
      step(r, std::false_type{});
      if (r + 1 >= rreal) break;
      step(r + 1, std::true_type{});
    }
    for (r = rreal; r < rend; ++r) tail_step(r);
    // ---- epilogue: pooling stage of the last centre row
    if (interior && (ye - 1) >= ys) stage3c(k7);
    if constexpr (FEAT) {
      if ((ye - 1) >= ys) {
        const sf4 fd = s_lds_read4(&s_fd[0][c][4 * j]), fd2 = s_lds_read4(&s_fd[1][c][4 * j]);
        feat_store(ye - 1, 4, fd.v, fd2.v);
      }
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
#endif
}

// The strips of a fused level: the border-free ones (strips strip0 .. strip0 + n_strip_l - 1, launch_band4f) on k_band4s / _heat / _feat, the
// ones at the image's left / right border (strip 0 and the last n_strip_l - 1) on k_band4s_edge / _edge_heat -- the EDGE body as kernels of
// their own, launched beside the others on the edge stream.  Both bodies in ONE kernel (a block-uniform branch; one launch per level) were
// measured too: in the timed step level 0 of 4K x 64 takes 7.34-7.50 ms instead of 7.13 (1080p: 2.03 against 1.92).  Not the code size
// (52 KB against 25 + 27): the instruction cache hits 99.9997 % of its requests either way, and run alone the two routes take the same
// 460 M SQ busy cycles (profiles/r05_icache_counters.txt) -- two launches side by side pack onto the CUs better than one launch whose
// slower border blocks are dealt among the others (profiles/r05_ab_edge_route.txt).  The work units of a launch are dealt to the launch indices
// so that the blocks resident on one XCD are neighbouring strips of the same rows (band4.hip).
template <bool HEAT, bool FEAT, bool EDGE>
__device__ __forceinline__ void band4s_kernel(const BandArgs& a) {
  __shared__ S4Lds<HEAT, FEAT> lds;
  const int per_xcd = a.per_xcd;
  const int wu = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);      // XCD-aware work-unit order (band4.hip)
  if (wu >= a.n_strip_l * a.n_seg * a.items) return;
  const int sl = wu % a.n_strip_l, seg = (wu / a.n_strip_l) % a.n_seg, item = wu / (a.n_strip_l * a.n_seg);
  const int strip = EDGE ? (sl == 0 ? 0 : a.n_strip - a.n_strip_l + sl) : a.strip0 + sl;
  band4s_body<HEAT, FEAT, EDGE>(a, lds, strip, seg, item);
}
__global__ __launch_bounds__(512, 4) void k_band4s(BandArgs a) { band4s_kernel<false, false, false>(a); }
__global__ __launch_bounds__(512, 4) void k_band4s_heat(BandArgs a) { band4s_kernel<true, false, false>(a); }
__global__ __launch_bounds__(512, 4) void k_band4s_feat(BandArgs a) { band4s_kernel<false, true, false>(a); }
// (no _edge_feat: the back waves' 16 sum registers leave no room for the border code; features clips keep k_band4f_feat<4, 1> for their
// border strips, launch_band4f)
__global__ __launch_bounds__(512, 4) void k_band4s_edge(BandArgs a) { band4s_kernel<false, false, true>(a); }
__global__ __launch_bounds__(512, 4) void k_band4s_edge_heat(BandArgs a) { band4s_kernel<true, false, true>(a); }

void launch_band4s_edge(const BandArgs& a, hipStream_t s) {
  if (a.dchr) hipLaunchKernelGGL(k_band4s_edge_heat, dim3(8 * a.per_xcd), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(k_band4s_edge, dim3(8 * a.per_xcd), dim3(512), 0, s, a);
}

void launch_band4s(const BandArgs& a, hipStream_t s) {
  if (a.fsum) hipLaunchKernelGGL(k_band4s_feat, dim3(8 * a.per_xcd), dim3(512), 0, s, a);
  else if (a.dchr) hipLaunchKernelGGL(k_band4s_heat, dim3(8 * a.per_xcd), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(k_band4s, dim3(8 * a.per_xcd), dim3(512), 0, s, a);
}

int tu_flags_band4s() {
  int f = 0;
#ifdef CVVDP_SAFE_LOADS
  f |= CVVDP_BUILD_SAFE_LOADS;
#endif
#if defined(S_DIAG_NOBAR) || defined(S_DIAG_BACK_ONLY) || defined(S_DIAG_FRONT_ONLY) || defined(S_DIAG_NO_STORE) || defined(S_DIAG_PLAIN_STORE) || \
    defined(S_DIAG_NO_NB) || defined(S_DIAG_NO_SG) || defined(S_DIAG_EDGE_RING) || defined(S_DIAG_NO_LUM) || defined(S_DIAG_VB4) || defined(S_PRIO_FRONT) || defined(S_PRIO_BACK) || CVVDP_BAND4S_RING != 8
  f |= CVVDP_BUILD_DIAG;
#endif
  return f;
}

}  // namespace cvvdp
