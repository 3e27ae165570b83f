// K2: Gaussian-pyramid reduce (lpyr_dec.py:186-211) and the stand-alone expand+add used by the
// heat-map reconstruction (lpyr_dec.py:223-239, 328-335).
//
// reduce = separable 5-tap [.05 .25 .4 .25 .05], stride 2, vertical pass first, written in the
// reference as a zero-padded convolution plus explicit edge terms.  The edge terms are reproduced
// literally, including the column edge that tests the ROW parity (lpyr_dec.py:206).
// Small levels: k_reduce, the reference's arithmetic operation for operation; large levels: the marching kernels
// k_reduce_vec / k_reduce2 (horizontal pass first in registers, no LDS), which round differently.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"

namespace cvvdp {

constexpr int RT = 32;            // output tile edge
constexpr int RIN = 2 * RT + 3;   // input patch edge (67)

// 5-tap dot product with a FIXED operation order -- THE REFERENCE'S: torch's CPU conv2d accumulates the taps in order with fused
// multiply-adds, starting from the first product (checked bit for bit against this torch build for column and row kernels at sizes
// from 7x5 to 2160x3840, tools/torch_conv_order.py), and zero padding contributes exactly nothing to such a chain.  The marching kernels
// below also evaluate the same taps from several inlined copies of their row code (prologue / steady state); left to the compiler,
// each copy may contract a*b + c*d + ... into FMAs differently, and a row's last bit would then depend on where its segment starts.
__device__ __forceinline__ float dot5(float a0, float a1, float a2, float a3, float a4, float k0, float k1, float k2, float k3, float k4) {
  return __builtin_fmaf(a4, k4, __builtin_fmaf(a3, k3, __builtin_fmaf(a2, k2, __builtin_fmaf(a1, k1, a0 * k0))));
}
__device__ __forceinline__ float dot2(float a0, float a1, float k0, float k1) { return __builtin_fmaf(a1, k1, a0 * k0); }
// a product / a sum rounded on its own, never contracted into a fused multiply-add: the reference's edge terms are separate tensor
// operations (`ya[0] += x[0]*K[1] + x[1]*K[0]`, lpyr_dec.py:195-209: two products, their sum, the in-place add: four roundings)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float edge2_rn(float v, float x0, float k0, float x1, float k1) { return add_rn(v, add_rn(mul_rn(x0, k0), mul_rn(x1, k1))); }

// The reference's reduce, OPERATION FOR OPERATION (k_reduce_ref): vertical pass first, each pass a dot5 chain, the edge terms as the
// separately rounded tensor operations they are in lpyr_dec.py:195-209 -- given the same input level, the output level equals torch's
// bit for bit (tests/test_gpu_parity.py::test_small_levels_reduce_like_the_reference_bit_for_bit).  Levels of at most
// kReduceRefPixels samples take this kernel (launch_reduce / run_pyramid_and_bands): the two coarsest Laplacian bands of a clip are a
// few dozen pixels whose Laplacian is a ~1e-4 relative difference of its operands, so one ulp of the Gaussian level is 3 x the
// Q_per_ch tolerance there, and the horizontal-first marching kernels below -- the same linear operator with another rounding order --
// put luminance-only clips 1.0-2.6 x outside it (the emulation of both orders on the CPU:
// profiles/r06_order_experiment.txt).  On large levels the order does not matter (thousands of pixels per band, well-conditioned
// Laplacians) and the marching kernels are 5 x faster.
//
// Tiling: a 256-thread block produces a 32x32 output tile.  The (2*32+3) x (2*32+3) input patch is staged in LDS with coalesced row
// loads, the vertical pass writes a 32 x 67 intermediate to LDS, the horizontal pass reads it.
__global__ __launch_bounds__(256) void k_reduce(ReduceArgs a) {
  __shared__ float s_in[RIN][RIN + 1];
  __shared__ float s_v[RT][RIN + 1];
  const int img = blockIdx.z;                       // plane * n_img + item
  const int plane = img / a.n_img, it = img - plane * a.n_img;
  const float* in = a.in + ((int64_t)plane * a.img_cap + it) * a.H * a.W;
  float* out = a.out + ((int64_t)plane * a.img_cap_out + it) * a.Ho * a.Wo;
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
    if (oy < a.Ho && ix0 + c >= 0 && ix0 + c < a.W) {
      const int pr = 2 * r;  // patch row of input row 2*oy-2
      v = dot5(s_in[pr][c], s_in[pr + 1][c], s_in[pr + 2][c], s_in[pr + 3][c], s_in[pr + 4][c], k0, k1, k2, k3, k4);
      // edge terms (lpyr_dec.py:195-199); patch row of input row y is y - iy0
      if (oy == 0) v = edge2_rn(v, s_in[0 - iy0][c], k1, s_in[1 - iy0][c], k0);
      if (oy == a.Ho - 1) {
        if (a.H & 1) v = edge2_rn(v, s_in[a.H - 1 - iy0][c], k3, s_in[a.H - 2 - iy0][c], k4);
        else v = add_rn(v, mul_rn(s_in[a.H - 1 - iy0][c], k4));
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
    float v = dot5(s_v[r][pc], s_v[r][pc + 1], s_v[r][pc + 2], s_v[r][pc + 3], s_v[r][pc + 4], k0, k1, k2, k3, k4);
    if (ox == 0) v = edge2_rn(v, s_v[r][0 - ix0], k1, s_v[r][1 - ix0], k0);                          // lpyr_dec.py:205
    if (ox == a.Wo - 1) {
      if (a.H & 1) v = edge2_rn(v, s_v[r][a.W - 1 - ix0], k3, s_v[r][a.W - 2 - ix0], k4);            // sic: row parity, :206-207
      else v = add_rn(v, mul_rn(s_v[r][a.W - 1 - ix0], k4));                                         // :209
    }
    out[(int64_t)oy * a.Wo + ox] = v;
  }
}

// Fast path for W % 8 == 0 (all large levels): no LDS, no barriers.  A thread owns 4 adjacent output
// columns and marches down a segment of output rows.  Per input row it loads 16+8 samples as aligned
// float4 (coalesced: a wave covers 1 KiB + apron per load), reduces them horizontally to 4 values and
// keeps the last 5 horizontally-reduced rows in registers; every second input row it emits one output
// row.  Horizontal-then-vertical is the same linear operator as the reference's vertical-then-
// horizontal (the two passes act on different axes); only fp32 rounding order differs (~1e-7).
constexpr int RSEG = 32;  // output rows per thread
typedef float v4f_ __attribute__((ext_vector_type(4)));

// Every call issues the same four 16-byte loads (no control flow around them), so a thread can keep several rows in
// flight and wait with exact counts.  Rows and columns outside the image are the reference's zero padding: the
// address is clamped into the image and the loaded values are multiplied by 0.  EDGE = the block touches the left or
// right image border (block-uniform); interior blocks skip the column clamps and masks.
template <bool EDGE>
__device__ __forceinline__ void hreduce_row(const ReduceArgs& a, const float* img, int y, float live, int ox, bool first, bool last, float (&h)[4]) {
  // input columns 2*ox-4 .. 2*ox+11 (three aligned float4 + one more)
  const int ix = 2 * ox - 4;
  const float* row = img + (int64_t)min(max(y, 0), a.H - 1) * a.W;
  const float my = (y >= 0 && y < a.H) ? live : 0.0f;
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = ix + 4 * q;
    float4 t;
    if constexpr (EDGE) {
      t = *reinterpret_cast<const float4*>(row + min(max(x, 0), a.W - 4));
      const float mx = (x >= 0 && x < a.W) ? 1.0f : 0.0f;
      t.x *= mx; t.y *= mx; t.z *= mx; t.w *= mx;
    } else {
#ifdef R2_NT_LOADS
      const v4f_ q4 = __builtin_nontemporal_load(reinterpret_cast<const v4f_*>(row + x));
      t = make_float4(q4.x, q4.y, q4.z, q4.w);
#else
      t = *reinterpret_cast<const float4*>(row + x);
#endif
    }
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
  const float k0 = a.k[0] * my, k1 = a.k[1] * my, k2 = a.k[2] * my, k3 = a.k[3] * my, k4 = a.k[4] * my;
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // output column ox+j: taps at input 2(ox+j)-2 .. +2 = v[2j+2 .. 2j+6]
    h[j] = dot5(v[2 * j + 2], v[2 * j + 3], v[2 * j + 4], v[2 * j + 5], v[2 * j + 6], k0, k1, k2, k3, k4);
  }
  if constexpr (EDGE) {
    if (first) h[0] += dot2(v[4], v[5], k1, k0);                  // lpyr_dec.py:205 (columns 0 and 1)
    if (last) {                                                     // output column Wo-1 = ox+3, W even here
      const int c1 = a.W - 1 - ix, c2 = a.W - 2 - ix;               // always 11 and 10 for W % 8 == 0
      if (a.H & 1) h[3] += dot2(v[c1], v[c2], k3, k4);              // sic: row parity (lpyr_dec.py:206-207)
      else h[3] = __builtin_fmaf(v[c1], k4, h[3]);                  // :209
    }
  }
}

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };   // 16 bytes at 4-byte alignment (rows of any width)

// Any width (W % 8 != 0, odd or even): the same arithmetic as hreduce_row<true>, with the image-border logic per sample
// instead of per 16-byte group.  Threads whose 16 input columns lie inside the image load them as four (unaligned)
// 16-byte groups; the few threads at the left / right border read sample by sample with clamped addresses and zero masks.
// The last output column Wo-1 can sit in any of the thread's four slots, and its two edge samples (columns W-1, W-2,
// lpyr_dec.py:206-209) are fetched separately, so that no register array is indexed dynamically.
__device__ __forceinline__ void hreduce_row_any(const ReduceArgs& a, const float* img, int y, int ox, float (&h)[4]) {
  const int ix = 2 * ox - 4;
  const float* row = img + (int64_t)min(max(y, 0), a.H - 1) * a.W;
  const float my = (y >= 0 && y < a.H) ? 1.0f : 0.0f;
  // Every lane issues the same four 16-byte loads, a group that leaves the image at a clamped address (ix is a multiple of 4, so
  // a group is left of column 0 as a whole; at the right border it is clamped to the row's last four samples).  Only the lanes
  // at a border then repair their groups in registers: zero padding left of column 0 and from column W on, the clamped group
  // shifted into place.  (Sample-by-sample loads for those lanes cost 16 more load instructions for the whole wave -- and at
  // widths of a few hundred pixels most waves hold a border lane.)
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4u t = *reinterpret_cast<const f4u*>(row + min(max(ix + 4 * q, 0), a.W - 4));
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
  if (ix < 0 || ix + 16 > a.W) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int x = ix + 4 * q;
      const int sh = x - (a.W - 4);                       // > 0: the group starts sh samples right of the clamped load
      const float r1 = v[4 * q + 1], r2 = v[4 * q + 2], r3 = v[4 * q + 3];
      if (x < 0 || sh >= 4) {
        v[4 * q] = v[4 * q + 1] = v[4 * q + 2] = v[4 * q + 3] = 0.0f;
      } else if (sh > 0) {
        v[4 * q] = sh == 1 ? r1 : (sh == 2 ? r2 : r3);
        v[4 * q + 1] = sh == 1 ? r2 : (sh == 2 ? r3 : 0.0f);
        v[4 * q + 2] = sh == 1 ? r3 : 0.0f;
        v[4 * q + 3] = 0.0f;
      }
    }
  }
  const float k0 = a.k[0] * my, k1 = a.k[1] * my, k2 = a.k[2] * my, k3 = a.k[3] * my, k4 = a.k[4] * my;
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = dot5(v[2 * j + 2], v[2 * j + 3], v[2 * j + 4], v[2 * j + 5], v[2 * j + 6], k0, k1, k2, k3, k4);
  if (ox == 0) h[0] += dot2(v[4], v[5], k1, k0);                    // lpyr_dec.py:205 (columns 0 and 1)
  const int jl = a.Wo - 1 - ox;                                     // slot of the last output column, if it is this thread's
  if (jl >= 0 && jl < 4) {
    const float e1 = row[a.W - 1], e2 = row[a.W - 2];
    float add;
    if (a.H & 1) add = dot2(e1, e2, k3, k4);                        // sic: row parity (lpyr_dec.py:206-207)
    else add = e1 * k4;                                             // :209
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j == jl) h[j] = (a.H & 1) ? h[j] + add : __builtin_fmaf(e1, k4, h[j]);
    }
  }
}

// ANYW = false: W % 8 == 0 (aligned 16-byte accesses, the last output column in slot 3); true: any width >= 16.
template <bool ANYW>
__global__ __launch_bounds__(256) void k_reduce_vec(ReduceArgs a) {
  const int img = blockIdx.z;
  const int plane = img / a.n_img, it = img - plane * a.n_img;
  const float* in = a.in + ((int64_t)plane * a.img_cap + it) * a.H * a.W;
  float* out = a.out + ((int64_t)plane * a.img_cap_out + it) * a.Ho * a.Wo;
  const int ox = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (ox >= a.Wo) return;
  const bool first = ox == 0, last = ox + 4 >= a.Wo;
  const int oy0 = blockIdx.y * RSEG, oy1 = min(oy0 + RSEG, a.Ho);
  const float k0 = a.k[0], k1 = a.k[1], k2 = a.k[2], k3 = a.k[3], k4 = a.k[4];
  // window of horizontally reduced input rows y-4 .. y ; rows outside the image are zero (+ edge terms)
  float w[5][4];
  const bool edge = blockIdx.x == 0 || (blockIdx.x + 1) * 1024 >= a.Wo;     // block-uniform
  auto hrow = [&](int y, float (&h)[4]) {
    if constexpr (ANYW) {
      hreduce_row_any(a, in, y, ox, h);      // (interior threads take its 16-byte branch: same loads as the aligned kernel)
    } else {
      if (edge) hreduce_row<true>(a, in, y, 1.0f, ox, first, last, h);
      else hreduce_row<false>(a, in, y, 1.0f, ox, false, false, h);
    }
  };
#pragma unroll
  for (int k = 0; k < 3; ++k) hrow(2 * oy0 - 2 + k, w[k]);         // rows 2oy-2, 2oy-1, 2oy of the first output
#pragma unroll 4
  for (int oy = oy0; oy < oy1; ++oy) {
    if (oy > oy0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { w[0][j] = w[2][j]; w[1][j] = w[3][j]; w[2][j] = w[4][j]; }
    }
    hrow(2 * oy + 1, w[3]);
    hrow(2 * oy + 2, w[4]);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = dot5(w[0][j], w[1][j], w[2][j], w[3][j], w[4][j], k0, k1, k2, k3, k4);
    if (oy == 0) {                                                   // lpyr_dec.py:195: rows 0 and 1 = w[2], w[3]
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += dot2(w[2][j], w[3][j], k1, k0);
    }
    if (oy == a.Ho - 1) {                                            // :196-199
      if (a.H & 1) {                                                 // centre row H-1 = w[2]; row H-2 = w[1]
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += dot2(w[2][j], w[1][j], k3, k4);
      } else {                                                       // centre row H-2 = w[2]; row H-1 = w[3]
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaf(w[3][j], k4, o[j]);
      }
    }
    if constexpr (ANYW) {
      float* dst = out + (int64_t)oy * a.Wo + ox;
      if (ox + 4 <= a.Wo) *reinterpret_cast<f4u*>(dst) = f4u{o[0], o[1], o[2], o[3]};
      else for (int j = 0; j < a.Wo - ox; ++j) dst[j] = o[j];
    } else {
      *reinterpret_cast<float4*>(out + (int64_t)oy * a.Wo + ox) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// Two pyramid levels per pass: level l -> l+1 -> l+2 (W_l % 16 == 0).  The level l+1 rows a thread has just produced
// in registers are reduced again on the spot, so level l+1 is written once and never read back by a reduce pass
// (saves its 32*(1/4) B/pixel re-read: 4.25 GB per 64-frame 4K block).
//   lane  <-> 4 adjacent level-(l+1) columns (as in k_reduce_vec) = 2 level-(l+2) columns; the two neighbouring
//             columns each side come from the adjacent lanes by wave shuffles.  Lanes 0 and 63 of a wave only
//             feed their neighbours: waves overlap by two lanes (3 % duplicated loads), so no LDS, no barriers;
//   march <-> a thread walks down SEG2 level-(l+2) rows = 2*SEG2 level-(l+1) rows (+3 recomputed halo rows)
//             = 4*SEG2 level-l rows, with the 5-row windows of both levels in registers.
// Edge terms of both levels as in k_reduce_vec (lpyr_dec.py:195-209, including the row-parity column edge).
constexpr int R2_LANES = 62;   // lanes of a wave that own level-(l+2) columns
constexpr int R2_SEG = 64;     // most level-(l+2) rows per thread (launch_reduce2 balances the segments)

// ANYW = false: W_l % 16 == 0 (aligned accesses, the last columns of both levels in fixed slots).  ANYW = true: any width >= 32:
// level l goes through hreduce_row_any (per-sample borders), level-(l+1) columns outside the image are the zero padding of the
// second pass, the last level-(l+2) column W2-1 can be either of the lane's two and takes its edge samples W1-1, W1-2 from the
// lane's own quad or its left neighbour's last column, stores are unaligned / partial at the right border.
template <bool ANYW>
__global__ __launch_bounds__(256) void k_reduce2(Reduce2Args a) {
  const int img = blockIdx.z;
  const int plane = img / a.n_img, it = img - plane * a.n_img;
  const int64_t ib = (int64_t)plane * a.img_cap + it;
  const float* in = a.in + ib * a.H * a.W;
  const int64_t ob = (int64_t)plane * a.img_cap_out + it;
  float* out1 = a.out1 + ob * a.H1 * a.W1;
  float* out2 = a.out2 + ob * a.H2 * a.W2;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int v = wave * R2_LANES + lane - 1;                 // quad index: level-(l+1) columns 4v .. 4v+3
  const int nq = ANYW ? (a.W1 + 3) >> 2 : a.W1 >> 2;
  if (wave * R2_LANES >= nq) return;                        // whole wave right of the image (wave-uniform)
  const int c1 = 4 * v;
  const bool in1 = v >= 0 && v < nq;
  const bool own = in1 && lane >= 1 && lane <= R2_LANES;    // writes its level-(l+1) quad and 2 level-(l+2) columns
  const bool first1 = v == 0, last1 = v == nq - 1;
  const float k0 = a.k[0], k1 = a.k[1], k2 = a.k[2], k3 = a.k[3], k4 = a.k[4];
  ReduceArgs ra;                                            // level l geometry for hreduce_row
  ra.H = a.H; ra.W = a.W; ra.Wo = a.W1;
#pragma unroll
  for (int i = 0; i < 5; ++i) ra.k[i] = a.k[i];

  const int r2a = blockIdx.y * a.seg2, r2b = min(r2a + a.seg2, a.H2);
  float w0[5][4];            // horizontally reduced level-l rows 2*y1-2 .. 2*y1+2 of the current level-(l+1) row y1
  bool cold = true;
  // block-uniform: does this block hold the lanes left of column 0 or at / beyond the last quad?
  const int wave0 = (blockIdx.x * 256) >> 6;
  const bool edge = wave0 == 0 || (wave0 + 4) * R2_LANES + 1 >= nq;
  const float live = in1 ? 1.0f : 0.0f;
  // ANYW: validity of the lane's four level-(l+1) columns, and where the last level-(l+2) column and its edge samples sit
  const int nv1 = ANYW ? min(max(a.W1 - c1, 0), 4) * (v >= 0 ? 1 : 0) : 4;       // columns c1 .. c1+nv1-1 exist
  const int slot2 = ANYW ? (a.W2 - 1) - 2 * v : 1;                               // 0 / 1: this lane owns column W2-1 in that slot
  const int e1i = a.W1 - 1 - c1, e2i = a.W1 - 2 - c1;                            // quad index of samples W1-1, W1-2 (e2i = -1: left neighbour's o[3])
  auto hrow0 = [&](int y, float (&h)[4]) {
    if constexpr (ANYW) {
      hreduce_row_any(ra, in, y, c1, h);
    } else {
      if (edge) hreduce_row<true>(ra, in, y, live, c1, first1, last1, h);
      else hreduce_row<false>(ra, in, y, 1.0f, c1, false, false, h);
    }
  };
  // level-(l+1) row y1 -> its two horizontally reduced level-(l+2) samples (zero row outside the image)
  auto l1row = [&](int y1, float (&hr)[2]) {
    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (y1 >= 0 && y1 < a.H1) {                             // block-uniform
      if (cold) {
#pragma unroll
        for (int k = 0; k < 3; ++k) hrow0(2 * y1 - 2 + k, w0[k]);
        cold = false;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { w0[0][j] = w0[2][j]; w0[1][j] = w0[3][j]; w0[2][j] = w0[4][j]; }
      }
      hrow0(2 * y1 + 1, w0[3]);
      hrow0(2 * y1 + 2, w0[4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = dot5(w0[0][j], w0[1][j], w0[2][j], w0[3][j], w0[4][j], k0, k1, k2, k3, k4);
      if (y1 == 0) {                                        // lpyr_dec.py:195
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += dot2(w0[2][j], w0[3][j], k1, k0);
      }
      if (y1 == a.H1 - 1) {                                 // :196-199
        if (a.H & 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += dot2(w0[2][j], w0[1][j], k3, k4);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaf(w0[3][j], k4, o[j]);
        }
      }
      if constexpr (ANYW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = j < nv1 ? o[j] : 0.0f;    // columns outside the image: the zero padding of the second pass
        if (own && y1 >= 2 * r2a && y1 < 2 * r2b) {
          float* dst = out1 + (int64_t)y1 * a.W1 + c1;
          if (nv1 == 4) *reinterpret_cast<f4u*>(dst) = f4u{o[0], o[1], o[2], o[3]};
          else for (int j = 0; j < nv1; ++j) dst[j] = o[j];
        }
      } else {
#ifndef R2_PLAIN_STORES   // level l+1 is next read by a later kernel, long after it left the L2: streaming stores (-1 % on the 4K level-0 pass)
      if (own && y1 >= 2 * r2a && y1 < 2 * r2b) __builtin_nontemporal_store(v4f_{o[0], o[1], o[2], o[3]}, reinterpret_cast<v4f_*>(out1 + (int64_t)y1 * a.W1 + c1));
#else
      if (own && y1 >= 2 * r2a && y1 < 2 * r2b) *reinterpret_cast<float4*>(out1 + (int64_t)y1 * a.W1 + c1) = make_float4(o[0], o[1], o[2], o[3]);
#endif
      }
    } else {
      cold = true;
    }
    // second level, horizontal pass: level-(l+2) column 2v reads level-(l+1) columns 4v-2 .. 4v+2, column 2v+1 reads 4v .. 4v+4
    const float l2 = __shfl_up(o[2], 1, 64), l3 = __shfl_up(o[3], 1, 64), rr0 = __shfl_down(o[0], 1, 64);
    hr[0] = dot5(l2, l3, o[0], o[1], o[2], k0, k1, k2, k3, k4);
    hr[1] = dot5(o[0], o[1], o[2], o[3], rr0, k0, k1, k2, k3, k4);
    if (first1) hr[0] += dot2(o[0], o[1], k1, k0);          // lpyr_dec.py:205 on level l+1
    if constexpr (ANYW) {
      if (slot2 == 0 || slot2 == 1) {                       // this lane owns column W2-1; sic: row parity (:206-209)
        const float s1 = e1i == 0 ? o[0] : (e1i == 1 ? o[1] : (e1i == 2 ? o[2] : o[3]));
        const float s2 = e2i < 0 ? l3 : (e2i == 0 ? o[0] : (e2i == 1 ? o[1] : o[2]));
        const float base = slot2 == 0 ? hr[0] : hr[1];
        const float fin = (a.H1 & 1) ? base + dot2(s1, s2, k3, k4) : __builtin_fmaf(s1, k4, base);
        if (slot2 == 0) hr[0] = fin; else hr[1] = fin;
      }
    } else if (last1) {                                     // column W2-1 = 2v+1 (W1 % 8 == 0); sic: row parity (:206-209)
      if (a.H1 & 1) hr[1] += dot2(o[3], o[2], k3, k4);
      else hr[1] = __builtin_fmaf(o[3], k4, hr[1]);
    }
  };

  float w2[5][2];            // horizontally reduced level-(l+1) rows 2*r2-2 .. 2*r2+2
#pragma unroll
  for (int k = 0; k < 3; ++k) l1row(2 * r2a - 2 + k, w2[k]);
  for (int r2 = r2a; r2 < r2b; ++r2) {
    if (r2 > r2a) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { w2[0][j] = w2[2][j]; w2[1][j] = w2[3][j]; w2[2][j] = w2[4][j]; }
    }
    l1row(2 * r2 + 1, w2[3]);
    l1row(2 * r2 + 2, w2[4]);
    float o2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) o2[j] = dot5(w2[0][j], w2[1][j], w2[2][j], w2[3][j], w2[4][j], k0, k1, k2, k3, k4);
    if (r2 == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) o2[j] += dot2(w2[2][j], w2[3][j], k1, k0);
    }
    if (r2 == a.H2 - 1) {
      if (a.H1 & 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) o2[j] += dot2(w2[2][j], w2[1][j], k3, k4);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) o2[j] = __builtin_fmaf(w2[3][j], k4, o2[j]);
      }
    }
    if constexpr (ANYW) {
      if (own) {
        float* dst = out2 + (int64_t)r2 * a.W2 + 2 * v;
        if (2 * v < a.W2) dst[0] = o2[0];
        if (2 * v + 1 < a.W2) dst[1] = o2[1];
      }
    } else {
      if (own) *reinterpret_cast<float2*>(out2 + (int64_t)r2 * a.W2 + 2 * v) = make_float2(o2[0], o2[1]);
    }
  }
}

// (two levels per pass only while BOTH inputs -- level l and the level l+1 it makes on the way -- are above the size from which the
// reference's operation order is reproduced, kReduceRefPixels)
bool reduce2_supported(int H, int W) {
  return (W % 16 == 0 || W >= 32) && H >= 8 && !reduce_takes_ref_kernel(H, W) && !reduce_takes_ref_kernel((H + 1) / 2, (W + 1) / 2);
}

void launch_reduce2(const Reduce2Args& a0, hipStream_t s) {
  Reduce2Args a = a0;
  const bool anyw = a.W % 16 != 0;
  const int nq = (a.W1 + 3) / 4, waves = (nq + R2_LANES - 1) / R2_LANES, bx = (waves + 3) / 4;
  // equal row segments of at most R2_SEG rows (3 recomputed halo rows each), more of them when the launch would
  // otherwise have fewer than ~4096 blocks (small frames), down to 16 rows
  static const int seg_cap = std::max(4, dev_knob("CVVDP_R2_SEG", R2_SEG));
  int n_seg = (a.H2 + seg_cap - 1) / seg_cap;
  const int64_t per_seg = (int64_t)bx * a.n_planes * a.n_img;
  n_seg = (int)std::max<int64_t>(n_seg, std::min<int64_t>((4096 + per_seg - 1) / per_seg, (a.H2 + 15) / 16));
  a.seg2 = (a.H2 + n_seg - 1) / n_seg;
  dim3 grid(bx, (a.H2 + a.seg2 - 1) / a.seg2, a.n_planes * a.n_img);
  if (anyw) hipLaunchKernelGGL(k_reduce2<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_reduce2<false>, grid, dim3(256), 0, s, a);
}

bool reduce_takes_ref_kernel(int H, int W) { return (int64_t)H * W <= kReduceRefPixels || W < 16 || H < 4; }

void launch_reduce(const ReduceArgs& a, hipStream_t s) {
  if (reduce_takes_ref_kernel(a.H, a.W)) {
    dim3 grid((a.Wo + RT - 1) / RT, (a.Ho + RT - 1) / RT, a.n_planes * a.n_img);
    hipLaunchKernelGGL(k_reduce, grid, dim3(256), 0, s, a);
    return;
  }
  if (a.W % 8 == 0 && a.H >= 4) {
    dim3 grid((a.Wo / 4 + 255) / 256, (a.Ho + RSEG - 1) / RSEG, a.n_planes * a.n_img);
    hipLaunchKernelGGL(k_reduce_vec<false>, grid, dim3(256), 0, s, a);
    return;
  }
  if (a.W >= 16 && a.H >= 4) {             // any other width: the same marching kernel with per-sample border handling
    dim3 grid(((a.Wo + 3) / 4 + 255) / 256, (a.Ho + RSEG - 1) / RSEG, a.n_planes * a.n_img);
    hipLaunchKernelGGL(k_reduce_vec<true>, grid, dim3(256), 0, s, a);
    return;
  }
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
    if (y & 1) return expand_odd(c[(int64_t)my * a.Wc + cx], c[(int64_t)y2 * a.Wc + cx], o);
    return expand_even(c[(int64_t)y0 * a.Wc + cx], c[(int64_t)my * a.Wc + cx], c[(int64_t)y2 * a.Wc + cx], e0, e1);
  };
  float v;
  if (x & 1) v = expand_odd(vert(mx), vert(x2), o);
  else v = expand_even(vert(x0), vert(mx), vert(x2), e0, e1);
  f[pix] += v;
}

// W % 4 == 0: a thread updates 4 adjacent fine pixels (one 16-byte read-modify-write) from a 3x4 coarse patch.
// Same expressions as k_expand_add: vertical first, then horizontal, neighbours clamped per fine pixel.
__global__ __launch_bounds__(256) void k_expand_add4(ExpandAddArgs a) {
  const int wq = a.W >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= a.H * wq) return;
  const int img = blockIdx.y;
  const float* c = a.coarse + (int64_t)img * a.Hc * a.Wc;
  float* f = a.fine + (int64_t)img * a.H * a.W;
  const int y = q / wq, x = (q - y * wq) << 2;
  const int my = y >> 1, mx = x >> 1;
  const int y0 = max(my - 1, 0), y2 = min(my + 1, a.Hc - 1);
  const float e0 = a.kx[0], e1 = a.kx[1], o = a.kx[2];
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cx = min(max(mx - 1 + k, 0), a.Wc - 1);
    if (y & 1) v[k] = expand_odd(c[(int64_t)my * a.Wc + cx], c[(int64_t)y2 * a.Wc + cx], o);
    else v[k] = expand_even(c[(int64_t)y0 * a.Wc + cx], c[(int64_t)my * a.Wc + cx], c[(int64_t)y2 * a.Wc + cx], e0, e1);
  }
  float4* p = reinterpret_cast<float4*>(f + (int64_t)y * a.W + x);
  float4 t = *p;
  t.x += expand_even(v[0], v[1], v[2], e0, e1);
  t.y += expand_odd(v[1], v[2], o);
  t.z += expand_even(v[1], v[2], v[3], e0, e1);
  t.w += expand_odd(v[2], v[3], o);
  *p = t;
}

// Round 6b: k_expand_add4 for the large levels of a heat-map reconstruction (level 1 of an 8K frame: 627 us per 16 frames = 1.9 TB/s
// for a read-modify-write of 8 B/pixel + 1 B/pixel of coarse samples).  A thread of k_expand_add4 divides its index by the row length,
// forms twelve clamped 64-bit addresses and loads a 3 x 4 coarse patch for its 4 pixels; here a block owns a tile of 16 rows x up to
// 1024 columns and a thread walks DOWN its four columns with the three coarse rows it needs as a rolling window in registers (four
// loads every second fine row), 32-bit offsets from uniform bases, row parity a compile-time constant of the unrolled row pair
// (heatmap.hip k_heat_colour_rows is the same walk).  The same expand_even / expand_odd chains on the same values: the same bits.
constexpr int kExpandTileRows = 16;
__global__ __launch_bounds__(256) void k_expand_add_rows(ExpandAddArgs a, int n_chunk, int chunk_cols) {
  const int img = blockIdx.y;
  const int chunk = (int)blockIdx.x % n_chunk, rg = (int)blockIdx.x / n_chunk;
  const int x = chunk * chunk_cols + 4 * (int)threadIdx.x;
  if (4 * (int)threadIdx.x >= chunk_cols || x >= a.W) return;
  const int W = a.W, Wc = a.Wc, Hc = a.Hc;
  const int y_begin = rg * kExpandTileRows, y_end = min(a.H, y_begin + kExpandTileRows);
  const float e0 = a.kx[0], e1 = a.kx[1], eo = a.kx[2];
  const float* coarse = a.coarse + (int64_t)img * Hc * Wc;
  char* fine = reinterpret_cast<char*>(a.fine + (int64_t)img * a.H * W);      // (a level of one image stays below 2^32 bytes)
  const int mx = x >> 1;
  int cx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cx[k] = min(max(mx - 1 + k, 0), Wc - 1);
  auto load_row = [&](int r, float (&d)[4]) {
    const float* p = coarse + (int64_t)r * Wc;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = p[cx[k]];
  };
  float cA[4], cB[4], cC[4];                                   // coarse rows max(my-1, 0), my, min(my+1, Hc-1) of the current fine row
  {
    const int my = y_begin >> 1;
    load_row(max(my - 1, 0), cA); load_row(my, cB); load_row(min(my + 1, Hc - 1), cC);
  }
  auto row = [&](int y, auto odd_) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (decltype(odd_)::value) v[k] = expand_odd(cB[k], cC[k], eo);
      else v[k] = expand_even(cA[k], cB[k], cC[k], e0, e1);
    }
    float4* p = reinterpret_cast<float4*>(fine + (uint32_t)(y * W + x) * 4u);
    float4 t = *p;
    t.x += expand_even(v[0], v[1], v[2], e0, e1);
    t.y += expand_odd(v[1], v[2], eo);
    t.z += expand_even(v[1], v[2], v[3], e0, e1);
    t.w += expand_odd(v[2], v[3], eo);
    *p = t;
  };
  for (int y = y_begin; y < y_end; y += 2) {                   // (y_begin is even)
    row(y, std::false_type{});
    if (y + 1 < y_end) row(y + 1, std::true_type{});
    if (y + 2 < y_end) {                                       // the next row pair's window: coarse row my+1 becomes my
      const int my = (y >> 1) + 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) { cA[k] = cB[k]; cB[k] = cC[k]; }
      load_row(min(my + 1, Hc - 1), cC);
    }
  }
}

void launch_expand_add(const ExpandAddArgs& a, hipStream_t s) {
  if (a.W % 4 == 0 && a.Wc >= 2) {
    if (a.W >= 512 && a.H >= 2 * kExpandTileRows && !a.per_thread_layout) {
      const int n_chunk = (a.W + 1023) / 1024;
      const int chunk_cols = ((a.W / 4 + n_chunk - 1) / n_chunk) * 4;
      const int n_rg = (a.H + kExpandTileRows - 1) / kExpandTileRows;
      hipLaunchKernelGGL(k_expand_add_rows, dim3(n_chunk * n_rg, a.n_img), dim3(256), 0, s, a, n_chunk, chunk_cols);
      return;
    }
    dim3 grid((a.H * (a.W / 4) + 255) / 256, a.n_img);
    hipLaunchKernelGGL(k_expand_add4, grid, dim3(256), 0, s, a);
    return;
  }
  dim3 grid((a.H * a.W + 255) / 256, a.n_img);
  hipLaunchKernelGGL(k_expand_add, grid, dim3(256), 0, s, a);
}

}  // namespace cvvdp
