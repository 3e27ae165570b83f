// K0+K1 fused for video: sample unpack -> display model -> DKL -> per-channel temporal FIR, one pass.
//   R[2c+side][fi] = sum_k dkl[side][ch(c)][window fi+k] * F[c][fl-1-k],   ch(3) = 0 (Y transient)
// Reference: video_source.py:320-346, display_model.py:333-365,266-269, cvvdp_metric.py:453-560.
//
// A thread owns V adjacent pixels of one (side, batch) for ALL three DKL planes and walks the block's
// frames in time, keeping the last FL DKL values of each plane in registers.  Consequences:
//   * every input frame is read from HBM exactly once and converted exactly once; the DKL values of the
//     block's own frames never touch memory (the reference materialises them in a ring and re-reads the
//     ring fl times);
//   * the only DKL state in HBM is the tail of the previous block: the last FL-1 window entries are
//     written to a (FL-1)-slot history buffer at the end of a block and read back at the start of the
//     next one (this replaces torch.roll, cvvdp_metric.py:538-539);
//   * temporal padding (replicate / symmetric, cvvdp_metric.py:506-529) and frame-range shard halos are
//     expressed by the host as "history entry k = raw frame index e" and converted in the prologue.
// The next frame's samples are loaded before the current frame's FMAs (software prefetch).
#pragma once
#include "photometry_dev.h"
#ifndef CVVDP_FIR_PF
#define CVVDP_FIR_PF 3
#endif
// Timing variant (tools/build_variant.sh -DCVVDP_FIR_DIAG_LDS_TAPS_MIN=n): filter lengths >= n of k_fir_rot read their rotated taps from LDS
// instead of scalar loads (see the kernel).  Measured and rejected in round 6 (profiles/r06_ab_fir_lds_taps.txt): 31 taps 8.61-8.66 ms against
// 7.78 with the scalar loads, 17 taps 5.92 against 5.56-5.63; results bit-identical.  Off in the product.
#ifndef CVVDP_FIR_DIAG_LDS_TAPS_MIN
#define CVVDP_FIR_DIAG_LDS_TAPS_MIN 1000
#endif
#define CVVDP_FIR_LDS_TAPS_FOR(FL) ((FL) >= CVVDP_FIR_DIAG_LDS_TAPS_MIN)
#include <cstdlib>

namespace cvvdp {

// What a thread keeps of one not-yet-converted pixel.  RGB / DKL sources: the three samples as floats.  Planar Y'CbCr
// sources (video_source_yuv.py:147-223): the luma code and the four bilinear taps of each chroma plane as integers,
// so that a prefetched frame is not touched (and its loads not waited for) before it is converted.
template <int DT, int V> struct Raw { float v[3][V]; };
struct RawYuv { uint32_t y, u[4], w[4]; };
template <> struct Raw<CVVDP_YUV8, 1> : RawYuv {};
template <> struct Raw<CVVDP_YUV16, 1> : RawYuv {};
constexpr bool is_yuv(int dt) { return dt == CVVDP_YUV8 || dt == CVVDP_YUV16; }

// Per-thread constants of the source addressing.  RGB / DKL: none.  Y'CbCr: the chroma taps of this pixel --
// torch.nn.functional.interpolate(mode='bilinear', align_corners=False) as used at video_source_yuv.py:212-216:
// source coordinate max((i + 0.5) / factor - 0.5, 0), neighbour clamped to the last sample.
template <int DT> struct PixCtx { __device__ PixCtx(const FirArgs&, int, int, int) {} };
struct YuvCtx {
  int32_t pix, o[4];       // luma offset in the frame; chroma tap offsets inside a chroma plane (y0x0, y0x1, y1x0, y1x1)
  float lx, ly;
  __device__ YuvCtx(const FirArgs& a, int pix_, int y, int x) : YuvCtx(a.yuv, pix_, y, x) {}
  __device__ YuvCtx(const YuvArgs& q, int pix_, int y, int x) : pix(pix_) {
    const float sx = fmaxf(((float)x + 0.5f) * q.inv_fx - 0.5f, 0.0f), sy = fmaxf(((float)y + 0.5f) * q.inv_fy - 0.5f, 0.0f);
    const int x0 = min((int)sx, q.Wc - 1), y0 = min((int)sy, q.Hc - 1);
    const int x1 = min(x0 + 1, q.Wc - 1), y1 = min(y0 + 1, q.Hc - 1);
    lx = sx - (float)x0; ly = sy - (float)y0;
    o[0] = y0 * q.Wc + x0; o[1] = y0 * q.Wc + x1; o[2] = y1 * q.Wc + x0; o[3] = y1 * q.Wc + x1;
  }
};
template <> struct PixCtx<CVVDP_YUV8> : YuvCtx { using YuvCtx::YuvCtx; };
template <> struct PixCtx<CVVDP_YUV16> : YuvCtx { using YuvCtx::YuvCtx; };

template <int DT, int V>
__device__ __forceinline__ void load_pixels(const FirArgs& a, const PixCtx<DT>& cx, int side, int64_t off, Raw<DT, V>& in) {
  const void* src = a.src[side];
  if constexpr (is_yuv(DT)) {
    static_assert(V == 1, "planar Y'CbCr sources are read one pixel per thread");
    // strides are (frame, W, 1): off = frame base + luma offset of this pixel
    const int64_t fb = off - cx.pix;
    auto ld = [&](int64_t i) -> uint32_t {
      if constexpr (DT == CVVDP_YUV8) return reinterpret_cast<const uint8_t*>(src)[i];
      else return reinterpret_cast<const uint16_t*>(src)[i];
    };
    in.y = ld(off);
#pragma unroll
    for (int k = 0; k < 4; ++k) { in.u[k] = ld(fb + a.yuv.u_off + cx.o[k]); in.w[k] = ld(fb + a.yuv.v_off + cx.o[k]); }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) load_run<DT, V>(src, off + c * a.sc[side], in.v[c]);   // sc == 0 for 1-channel clips
  }
}

// One Y'CbCr pixel -> display-encoded R'G'B' in [0,1]: limited-range fixed point -> float (video_source_yuv.py:197-210),
// bilinear chroma (:212-216), matrix + clip (:151-170)
__device__ __forceinline__ void yuv_pixel_rgb(const YuvArgs& q, const YuvCtx& cx, const RawYuv& in, float (&v)[3]) {
  const float Y = clipf(q.wy * (float)in.y - q.oy, 0.0f, 1.0f);
  float ch[2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = clipf(q.wc * (float)(pl == 0 ? in.u[k] : in.w[k]) - q.oc, -0.5f, 0.5f);
    const float top = t[0] * (1.0f - cx.lx) + t[1] * cx.lx, bot = t[2] * (1.0f - cx.lx) + t[3] * cx.lx;
    ch[pl] = top * (1.0f - cx.ly) + bot * cx.ly;
  }
  v[0] = clipf(Y + ch[1] * q.rv, 0.0f, 1.0f);
  v[1] = clipf(Y + ch[0] * q.gu + ch[1] * q.gv, 0.0f, 1.0f);
  v[2] = clipf(Y + ch[0] * q.bu, 0.0f, 1.0f);
}

template <int DT, int V>
__device__ __forceinline__ void convert_pixels(const FirArgs& a, const PixCtx<DT>& cx, const Raw<DT, V>& in, float (&dkl)[3][V],
                                               const float* lut = nullptr, bool use_lut = false) {
  if constexpr (is_yuv(DT)) {
    float v[3];
    yuv_pixel_rgb(a.yuv, cx, in, v);
    float o[3];
    pixel_to_dkl(a.dm, v, o);
    dkl[0][0] = o[0]; dkl[1][0] = o[1]; dkl[2][0] = o[2];
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float v[3] = {in.v[0][i], in.v[1][i], in.v[2][i]}, o[3];
      if constexpr (DT == CVVDP_F32_DKL) {
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
      } else {
        pixel_to_dkl(a.dm, v, o, lut, use_lut);
      }
      dkl[0][i] = o[0]; dkl[1][i] = o[1]; dkl[2][i] = o[2];
    }
  }
}

template <int V>
__device__ __forceinline__ void store_run(float* p, const float (&v)[V]) {
  if constexpr (V == 1) *p = v[0];
  else if constexpr (V == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  else *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <int V>
__device__ __forceinline__ void load_f32_run(const float* p, float (&v)[V]) {
  if constexpr (V == 1) v[0] = *p;
  else if constexpr (V == 2) { const float2 q = *reinterpret_cast<const float2*>(p); v[0] = q.x; v[1] = q.y; }
  else { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
}

template <int DT, int FL, int V>
__global__ __launch_bounds__(256) void k_fir_fused(FirArgs a) {
  __shared__ float s_tab[DT == CVVDP_U8 ? 256 : 1];
  const bool use_lut = stage_eotf_table<DT>(a.dm, s_tab);
  const int pix = (blockIdx.x * 256 + threadIdx.x) * V;
  if (pix >= a.P) return;
  const int b = blockIdx.y, side = blockIdx.z;
  const int y = pix / a.W, x = pix - y * a.W;
  const PixCtx<DT> cx(a, pix, y, x);
  const int64_t off0 = b * a.sb[side] + (int64_t)y * a.sh[side] + (int64_t)x * a.sw[side];
  const int64_t sf = a.sf[side];
  float* hist = a.hist + side * a.h_side + b * a.h_b + pix;

  // Window of FL + U - 1 entries: a chunk of U frames is appended at static positions FL-1 .. FL+U-2, the U
  // outputs read statically shifted sub-windows, and the window is shifted by U once per chunk (the
  // per-frame shift of FL*3*V registers was a third of this kernel's VALU work).
  constexpr int U = (V == 1) ? 4 : 1;   // measured on 4K/60: V=1,U=4 8.4 ms; V=2,U=1 9.0 ms; V=2,U=4 12 ms (212 VGPRs)
  constexpr int WL = FL + U - 1;
  float w[3][WL][V];
  // ---- prologue: window positions 0..FL-2 (history / temporal padding)
#pragma unroll
  for (int k = 0; k < FL - 1; ++k) {
    const int e = a.hist_src[k];
    if (e >= 0) {       // raw frame e of the block handed in by the host
      if (k > 0 && e == a.hist_src[k - 1]) {   // replicate padding: the same frame again (uniform test)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < V; ++i) w[p][k][i] = w[p][k - 1][i];
      } else {
        Raw<DT, V> in;
        float d[3][V];
        load_pixels<DT, V>(a, cx, side, off0 + e * sf, in);
        convert_pixels<DT, V>(a, cx, in, d, s_tab, use_lut);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < V; ++i) w[p][k][i] = d[p][i];
      }
    } else {            // slot -1-e of the previous block's tail
#pragma unroll
      for (int p = 0; p < 3; ++p) load_f32_run<V>(hist + p * a.h_plane + (int64_t)(-1 - e) * a.h_slot, w[p][k]);
    }
  }
  float* out = a.out + (int64_t)b * a.P + pix;
  const int64_t o_item = (int64_t)a.batch * a.P;
  // software prefetch PF frames deep: with ~130 VGPRs only 3 waves/SIMD are resident, so the bytes in
  // flight per CU have to come from depth (3 waves x 4 SIMDs x PF frames x 3 loads x 512 B ~ 55 KB)
  constexpr int PF = CVVDP_FIR_PF;
  Raw<DT, V> pf[PF];
#pragma unroll
  for (int q = 0; q < PF; ++q)
    if (q < a.n_frames) load_pixels<DT, V>(a, cx, side, off0 + (int64_t)(a.raw_first + q) * sf, pf[q]);
  for (int f0 = 0; f0 < a.n_frames; f0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int fi = f0 + u;
      if (fi < a.n_frames) {     // uniform
        float d[3][V];
        convert_pixels<DT, V>(a, cx, pf[0], d, s_tab, use_lut);
#pragma unroll
        for (int q = 0; q + 1 < PF; ++q) pf[q] = pf[q + 1];
        if (fi + PF < a.n_frames) load_pixels<DT, V>(a, cx, side, off0 + (int64_t)(a.raw_first + fi + PF) * sf, pf[PF - 1]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < V; ++i) w[p][FL - 1 + u][i] = d[p][i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {     // Y-sust, RG, YV, Y-trans (window of plane 0 again), cvvdp_metric.py:554-560
          const int p = (c == 3) ? 0 : c;
          float acc[V];
#pragma unroll
          for (int i = 0; i < V; ++i) acc[i] = 0.0f;
#pragma unroll
          for (int k = 0; k < FL; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] += w[p][u + k][i] * a.taps[c * CVVDP_MAX_FILTER_LEN + k];
          store_run<V>(out + (int64_t)(2 * c + side) * a.o_plane + (int64_t)fi * o_item, acc);
        }
      }
    }
    if (f0 + U < a.n_frames) {   // more chunks follow: drop the U oldest entries
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int k = 0; k < FL - 1; ++k)
#pragma unroll
          for (int i = 0; i < V; ++i) w[p][k][i] = w[p][k + U][i];
    }
  }
  // ---- epilogue: the last FL-1 frames become the next block's history.  The final chunk held r+1 frames
  // (r = (n-1) % U), so they sit at window positions r+1 .. r+FL-1.
  if (a.write_hist) {
    const int r = (a.n_frames - 1) % U;
#pragma unroll
    for (int rr = 0; rr < U; ++rr) {
      if (r == rr) {
#pragma unroll
        for (int k = 0; k < FL - 1; ++k)
#pragma unroll
          for (int p = 0; p < 3; ++p) store_run<V>(hist + p * a.h_plane + (int64_t)k * a.h_slot, w[p][rr + 1 + k]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Fast path for FL <= 31 (24 .. 120 fps), one pixel per thread: the FL-deep window of each DKL plane is a register
// vector; the newest frame is written to slot (frame index mod (FL-1)) with an M0-relative register write and
// the taps are read rotated instead (scalar loads from a doubled tap table), so nothing is shifted and
// the window costs 51 VGPRs at FL <= 17 (5-6 waves per SIMD hide the HBM latency this kernel is bound by), 99 at 25 / 31 taps
// (90 / 120 fps: k_fir_fused's shifted window made those rates VALU-bound at 0.33 of 8 TB/s).
template <int DT, int FL>
__global__ __launch_bounds__(256) void k_fir_rot(FirArgs a) {
  __shared__ float s_tab[DT == CVVDP_U8 ? 256 : 1];
  // LDS_TAPS (timing variant, off in the product): the rotated taps read from LDS (uniform-address ds_read_b128 = 4 taps per read, in-order
  // lgkmcnt waits the compiler can count) instead of scalar loads from the kernel arguments.  The idea: scalar loads return out of order,
  // so every use sits behind an s_waitcnt lgkmcnt(0), and with 4 x (FL-1) taps per frame against ~100 SGPRs the compiler loads them in
  // chunks into ONE SGPR range and waits for each chunk in full -- eight exposed scalar-cache round trips per frame at 31 taps
  // (profiles/r06_fir_counters.txt: waves parked 62 % of their cycles).  Four copies of the table, shifted by 0..3 taps, make every
  // rotation offset a 16-byte aligned read (copy o & 3 at element o & ~3); same taps, slots and order of the sums: bit-identical results.
  // MEASURED SLOWER (profiles/r06_ab_fir_lds_taps.txt): 159 instead of 135 VGPRs at 31 taps (three waves per SIMD either way), 32 LDS
  // reads per frame on the one LDS port four SIMDs share -- the scalar cache's round trips were not what the waves are parked on.
  constexpr bool LDS_TAPS = CVVDP_FIR_LDS_TAPS_FOR(FL);
  __shared__ __attribute__((aligned(16))) float s_taps[LDS_TAPS ? 4 : 1][LDS_TAPS ? 4 : 1][LDS_TAPS ? CVVDP_ROT_TAPS : 4];
  if constexpr (LDS_TAPS) {
    for (int i = threadIdx.x; i < 4 * 4 * CVVDP_ROT_TAPS; i += 256) {
      const int sh = i / (4 * CVVDP_ROT_TAPS), c = (i / CVVDP_ROT_TAPS) & 3, k = i & (CVVDP_ROT_TAPS - 1);
      s_taps[sh][c][k] = (k + sh < 2 * (FL - 1)) ? a.taps_rot[c * CVVDP_ROT_TAPS + k + sh] : 0.0f;
    }
    __syncthreads();
  }
  const bool use_lut = stage_eotf_table<DT>(a.dm, s_tab);
  static_assert(FL >= 3 && 2 * (FL - 1) <= CVVDP_ROT_NEW, "window = one 16- or 32-wide register vector + the newest frame in a scalar slot");
  typedef float v16f __attribute__((ext_vector_type(FL <= 17 ? 16 : 32)));
  typedef float v2f_ __attribute__((ext_vector_type(2)));
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  static_assert((FL - 1) % 2 == 0, "the older frames are summed in slot pairs");
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.P) return;
  const int b = blockIdx.y, side = blockIdx.z;
  const int y = pix / a.W, x = pix - y * a.W;
  const PixCtx<DT> cx(a, pix, y, x);
  const int64_t off0 = b * a.sb[side] + (int64_t)y * a.sh[side] + (int64_t)x * a.sw[side];
  const int64_t sf = a.sf[side];
  float* hist = a.hist + side * a.h_side + b * a.h_b + pix;

  // Window of frame A = frames A-M..A (M = FL-1).  The newest frame lives in a scalar register per plane (whi), the M
  // older ones in one register vector per plane: frame B sits in slot B mod M and is written there -- by an
  // M0-relative move, the slot is wave-uniform -- one step after it arrived, replacing frame B-M.  The data never
  // moves again; the taps rotate instead: slot s holds window position (s - A) mod M, whose weight is read from a
  // doubled table at offset (M - A mod M) mod M.  Slots are keyed to the CLIP index of a frame, so the order in which a
  // frame's FL products are summed does not depend on how the clip is cut into blocks or shards (bit-identical
  // results for any blocking), and the loop body has no data-dependent control flow.
  constexpr int M = FL - 1;
  v16f wlo[3];
  float whi[3];
  int sA = ((a.abs_first % M) + M) % M;               // A mod M for the frame about to be processed
#ifdef CVVDP_FIR_DIAG_PF   /* timing variant: prefetch depth */
  constexpr int PF = CVVDP_FIR_DIAG_PF;
#else
  constexpr int PF = is_yuv(DT) ? 2 : 4;   // nine integer samples per prefetched Y'CbCr pixel: keep the VGPR count at 5 waves/SIMD
#endif
  // A frame-range shard starts with M real halo frames that sit right before its first frame (hist_src = a run of raw
  // frames): they are pushed through the same pipelined loop as the scored frames, minus the FIR and the stores,
  // instead of one exposed memory round trip per entry.
  const bool warm = a.halo_run && (M % PF == 0);
  // ---- prologue: window positions 0..M-1 of the first frame = frames A0-M .. A0-1
  float d[3][1] = {{0.0f}, {0.0f}, {0.0f}};
  for (int k = 0; k < (warm ? 0 : M); ++k) {
    const int e = a.hist_src[k];
    // replicate padding (cvvdp_metric.py:506-512: the clip's first frame M times): the same raw frame as the entry before -- keep its DKL
    // values instead of loading and converting it again (a uniform test; 15 of the 80 conversions of a 64-frame clip at 60 fps)
    const bool again = k > 0 && e >= 0 && e == a.hist_src[k - 1];
    if (again) {
    } else if (e >= 0) {
      Raw<DT, 1> in;
      load_pixels<DT, 1>(a, cx, side, off0 + e * sf, in);
      convert_pixels<DT, 1>(a, cx, in, d, s_tab, use_lut);
    } else {
      for (int p = 0; p < 3; ++p) d[p][0] = hist[p * a.h_plane + (int64_t)(-1 - e) * a.h_slot];
    }
    if (k < M - 1) {
      const int sl = __builtin_amdgcn_readfirstlane((sA + k) % M);   // frame A0-M+k -> slot (A0+k) mod M
#pragma unroll
      for (int p = 0; p < 3; ++p) wlo[p][sl] = d[p][0];
    } else {
#pragma unroll
      for (int p = 0; p < 3; ++p) whi[p] = d[p][0];    // frame A0-1
    }
  }
  float* out = a.out + (int64_t)b * a.P + pix;
  const int64_t o_item = (int64_t)a.batch * a.P;
  Raw<DT, 1> pf[PF];
  const int f_start = warm ? -M : 0;
#pragma unroll
  for (int q = 0; q < PF; ++q)
    load_pixels<DT, 1>(a, cx, side, off0 + (int64_t)(a.raw_first + min(f_start + q, a.n_frames - 1)) * sf, pf[q]);
  if (warm) {
    whi[0] = whi[1] = whi[2] = 0.0f;                  // lands in the slot that frame A0-1 overwrites before it is read
    for (int fw = -M; fw < 0; fw += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        float d[3][1];
        convert_pixels<DT, 1>(a, cx, pf[u], d, s_tab, use_lut);
        load_pixels<DT, 1>(a, cx, side, off0 + (int64_t)(a.raw_first + min(fw + u + PF, a.n_frames - 1)) * sf, pf[u]);
        const int sw = __builtin_amdgcn_readfirstlane(sA == 0 ? M - 1 : sA - 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) { wlo[p][sw] = whi[p]; whi[p] = d[p][0]; }
        sA = (sA + 1 == M) ? 0 : sA + 1;
      }
    }
  }
  // One frame: convert prefetch slot Q, refill it with frame fi+PF, FIR, store.  The refill is unconditional (the
  // last PF frames re-read the last frame) and the frame loop is unrolled PF times with static prefetch slots, so
  // no register copies touch values still in flight and the compiler can wait with exact vmcnt(N) counts.
#ifdef CVVDP_FIR_PLAIN_STORES
#define CVVDP_FIR_STORE(v, p) (*(p) = (v))
#else
#define CVVDP_FIR_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif
#define CVVDP_FIR_TAP2(s) (LDS_TAPS ? (((s) & 2) ? v2f_{tq[(s) / 4].z, tq[(s) / 4].w} : v2f_{tq[(s) / 4].x, tq[(s) / 4].y}) : v2f_{t[s], t[(s) + 1]})
#ifndef CVVDP_FIR_CHAINS
#define CVVDP_FIR_CHAINS 2            /* independent accumulator pairs per channel (timing variants: tools/build_variant.sh -DCVVDP_FIR_CHAINS=n) */
#endif
#if defined(CVVDP_FIR_DIAG_SCALAR)  /* timing variant: round 5's scalar multiply-adds, one dependent chain of M */
#define CVVDP_FIR_SUM_PAIRS                                                                                      \
      float accs = 0.0f;                                                                                         \
      _Pragma("unroll") for (int s = 0; s < M; ++s) accs += wlo[p][s] * t[s];                                    \
      acc2.x = accs;
#else
/* slot pair j = (2j, 2j+1) goes to chain j mod CHAINS; the chains are added pairwise at the end.  One chain of M/2 dependent packed FMAs   */
/* per channel left the kernel LATENCY-bound (round 6 A/B, profiles/r06_ab_fir_chains.txt: 60 fps 5.94 -> 5.24 ms with two chains at     */
/* unchanged instruction count -- it had been taken for HBM-bound at 5.1 TB/s since round 4).  The order is a function of the slots, i.e. of */
/* the frames' CLIP indices: results still do not depend on how the clip is cut.                                                            */
#define CVVDP_FIR_SUM_PAIRS                                                                                      \
      v2f_ ch_[CVVDP_FIR_CHAINS];                                                                                \
      _Pragma("unroll") for (int q = 0; q < CVVDP_FIR_CHAINS; ++q) ch_[q] = v2f_{0.0f, 0.0f};                    \
      _Pragma("unroll") for (int s = 0; s < M; s += 2) ch_[(s / 2) % CVVDP_FIR_CHAINS] += v2f_{wlo[p][s], wlo[p][s + 1]} * CVVDP_FIR_TAP2(s); \
      _Pragma("unroll") for (int w = 1; w < CVVDP_FIR_CHAINS; w *= 2)                                            \
        _Pragma("unroll") for (int q = 0; q + w < CVVDP_FIR_CHAINS; q += 2 * w) ch_[q] += ch_[q + w];            \
      acc2 = ch_[0];
#endif
#define CVVDP_FIR_FRAME(FI, Q)                                                                                   \
  {                                                                                                              \
    float d[3][1];                                                                                               \
    convert_pixels<DT, 1>(a, cx, pf[Q], d, s_tab, use_lut);                                                                          \
    load_pixels<DT, 1>(a, cx, side, off0 + (int64_t)(a.raw_first + min((FI) + PF, a.n_frames - 1)) * sf, pf[Q]);    \
    const int sw = __builtin_amdgcn_readfirstlane(sA == 0 ? M - 1 : sA - 1);   /* (A-1) mod M, in an SGPR */      \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) { wlo[p][sw] = whi[p]; whi[p] = d[p][0]; }                     \
    const float* tb = a.taps_rot + (sA == 0 ? 0 : M - sA);                                                       \
    const int to_ = (sA == 0 ? 0 : M - sA);                                                                      \
    const float* tl = &s_taps[LDS_TAPS ? (to_ & 3) : 0][0][LDS_TAPS ? (to_ & ~3) : 0];                           \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) { /* Y-sust, RG, YV, Y-trans (plane 0 again), cvvdp_metric.py:554-560 */ \
      const int p = (c == 3) ? 0 : c;                                                                            \
      const float* t = tb + c * CVVDP_ROT_TAPS;                                                                  \
      v4f_ tq[LDS_TAPS ? (M + 3) / 4 : 1];                                                                        \
      if constexpr (LDS_TAPS) {                                                                                  \
        _Pragma("unroll") for (int k4 = 0; k4 < (M + 3) / 4; ++k4) tq[k4] = *reinterpret_cast<const v4f_*>(tl + c * CVVDP_ROT_TAPS + 4 * k4); \
      }                                                                                                          \
      /* Round 6: the M older frames as M/2 PACKED multiply-adds (v_pk_fma_f32: slots 2j, 2j+1 are an aligned register pair of the  */ \
      /* window vector, their taps an SGPR pair) -- 62 + 8 instead of 124 FMAs per frame at 31 taps, where this kernel is VALU-bound. */ \
      /* Two interleaved partial sums (even slots, odd slots), added at the end: the order is a function of the slots, i.e. of the     */ \
      /* frames' CLIP indices, so results still do not depend on how the clip is cut.                                                 */ \
      v2f_ acc2 = {0.0f, 0.0f};                                                                                  \
      CVVDP_FIR_SUM_PAIRS                                                                                        \
      float acc = acc2.x + acc2.y;                                                                               \
      acc += whi[p] * a.taps_rot[c * CVVDP_ROT_TAPS + CVVDP_ROT_NEW];                                            \
      CVVDP_FIR_STORE(acc, &out[(int64_t)(2 * c + side) * a.o_plane + (int64_t)(FI) * o_item]);                \
    }                                                                                                            \
    sA = (sA + 1 == M) ? 0 : sA + 1;                                                                             \
  }
  int fi = 0;
  for (; fi + PF <= a.n_frames; fi += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) CVVDP_FIR_FRAME(fi + u, u)
  }
#pragma unroll
  for (int u = 0; u + 1 < PF; ++u)                    // remainder: slots 0.. hold frames fi.. in order
    if (fi + u < a.n_frames) CVVDP_FIR_FRAME(fi + u, u)
#undef CVVDP_FIR_FRAME
#undef CVVDP_FIR_SUM_PAIRS
#undef CVVDP_FIR_TAP2
  // ---- epilogue: the last M frames in time order = window positions 1..M of the last frame's window: position k < M
  // sits in slot (A_last + k) mod M = (sA - 1 + k) mod M (sA is already A_last + 1), position M is whi
  if (a.write_hist) {
    for (int k = 1; k < M; ++k) {
      const int sl = __builtin_amdgcn_readfirstlane((sA + M - 1 + k) % M);
#pragma unroll
      for (int p = 0; p < 3; ++p) hist[p * a.h_plane + (int64_t)(k - 1) * a.h_slot] = wlo[p][sl];
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) hist[p * a.h_plane + (int64_t)(M - 1) * a.h_slot] = whi[p];
  }
}

// Any filter length (odd frame rates): no register window; every tap re-reads and re-converts its frame.
template <int DT>
__global__ __launch_bounds__(256) void k_fir_generic(FirArgs a) {
  __shared__ float s_tab[DT == CVVDP_U8 ? 256 : 1];
  const bool use_lut = stage_eotf_table<DT>(a.dm, s_tab);
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.P) return;
  const int item = blockIdx.y, side = blockIdx.z;
  const int fi = item / a.batch, b = item - fi * a.batch;
  const int y = pix / a.W, x = pix - y * a.W;
  const PixCtx<DT> cx(a, pix, y, x);
  const int64_t off0 = b * a.sb[side] + (int64_t)y * a.sh[side] + (int64_t)x * a.sw[side];
  const float* hist = a.hist + side * a.h_side + b * a.h_b + pix;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int k = 0; k < a.fl; ++k) {
    const int pos = fi + k;                                    // window position
    float d[3][1];
    const int e = pos < a.fl - 1 ? (int)a.hist_src[pos] : a.raw_first + pos - (a.fl - 1);
    if (e >= 0) {
      Raw<DT, 1> in;
      load_pixels<DT, 1>(a, cx, side, off0 + e * a.sf[side], in);
      convert_pixels<DT, 1>(a, cx, in, d, s_tab, use_lut);
    } else {
      for (int p = 0; p < 3; ++p) d[p][0] = hist[p * a.h_plane + (int64_t)(-1 - e) * a.h_slot];
    }
    for (int c = 0; c < 4; ++c) acc[c] += d[c == 3 ? 0 : c][0] * a.taps[c * CVVDP_MAX_FILTER_LEN + k];
  }
  for (int c = 0; c < 4; ++c) a.out[(int64_t)(2 * c + side) * a.o_plane + (int64_t)item * a.P + pix] = acc[c];
}

// tail of the block -> history, for the generic path (the fused kernel does this itself)
template <int DT>
__global__ __launch_bounds__(256) void k_hist_generic(FirArgs a, float* tmp) {
  __shared__ float s_tab[DT == CVVDP_U8 ? 256 : 1];
  const bool use_lut = stage_eotf_table<DT>(a.dm, s_tab);
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.P) return;
  const int k = blockIdx.y % (a.fl - 1), b = blockIdx.y / (a.fl - 1), side = blockIdx.z;
  const int y = pix / a.W, x = pix - y * a.W;
  const PixCtx<DT> cx(a, pix, y, x);
  const int64_t off0 = b * a.sb[side] + (int64_t)y * a.sh[side] + (int64_t)x * a.sw[side];
  const float* hist = a.hist + side * a.h_side + b * a.h_b + pix;
  const int pos = a.n_frames + k;                              // window position of new slot k
  const int e = pos < a.fl - 1 ? (int)a.hist_src[pos] : a.raw_first + pos - (a.fl - 1);
  float d[3][1];
  if (e >= 0) {
    Raw<DT, 1> in;
    load_pixels<DT, 1>(a, cx, side, off0 + e * a.sf[side], in);
    convert_pixels<DT, 1>(a, cx, in, d, s_tab, use_lut);
  } else {
    for (int p = 0; p < 3; ++p) d[p][0] = hist[p * a.h_plane + (int64_t)(-1 - e) * a.h_slot];
  }
  // write to the shadow copy (tmp) so that slots still to be read are not overwritten
  float* dst = tmp + side * a.h_side + b * a.h_b + pix;
  for (int p = 0; p < 3; ++p) dst[p * a.h_plane + (int64_t)k * a.h_slot] = d[p][0];
}

template <int DT, int FL>
static void launch_fused(const FirArgs& a, hipStream_t s) {
  constexpr int VMAX = FL <= 17 ? 2 : 1;
  const int eb = dtype_bytes(a.dtype);
  static const int vcap = dev_knob("CVVDP_FIR_V", 1);   // 1 = scalar pixels + chunked window (fastest)
  if constexpr (VMAX == 2 && !is_yuv(DT)) {
    if (vcap >= 2 && a.P % 2 == 0 && can_vectorise(2, a.W, a.sb, a.sc, a.sf, a.sh, a.sw, a.src, eb)) {
      dim3 grid((a.P / 2 + 255) / 256, a.batch, 2);
      hipLaunchKernelGGL((k_fir_fused<DT, FL, 2>), grid, dim3(256), 0, s, a);
      return;
    }
  }
  {
    dim3 grid((a.P + 255) / 256, a.batch, 2);
    static const bool rot = dev_knob("CVVDP_FIR_ROT", 1) != 0;
    if constexpr (2 * (FL - 1) <= CVVDP_ROT_NEW) {
      if (rot) { hipLaunchKernelGGL((k_fir_rot<DT, FL>), grid, dim3(256), 0, s, a); return; }
    }
    hipLaunchKernelGGL((k_fir_fused<DT, FL, 1>), grid, dim3(256), 0, s, a);
  }
}

template <int DT>
static bool launch_dt(const FirArgs& a, hipStream_t s) {
  switch (a.fl) {
    case 7: launch_fused<DT, 7>(a, s); return true;     // 24 fps          (N = ceil(fps/8)*2+1)
    case 9: launch_fused<DT, 9>(a, s); return true;     // 25, 30 fps
    case 13: launch_fused<DT, 13>(a, s); return true;   // 48 fps
    case 15: launch_fused<DT, 15>(a, s); return true;   // 50 fps
    case 17: launch_fused<DT, 17>(a, s); return true;   // 60 fps
    case 25: launch_fused<DT, 25>(a, s); return true;   // 90 fps
    case 31: launch_fused<DT, 31>(a, s); return true;   // 120 fps
    default: break;
  }
  hipLaunchKernelGGL(k_fir_generic<DT>, dim3((a.P + 255) / 256, a.n_frames * a.batch, 2), dim3(256), 0, s, a);
  return false;
}

// One sample format per translation unit (temporal_<fmt>.hip), so that the formats compile in parallel.
template <int DT>
static void launch_fir_typed(const FirArgs& a, float* hist_shadow, hipStream_t s) {
  const bool fused = launch_dt<DT>(a, s);
  if (!fused && a.write_hist && a.fl > 1) {      // generic-length path: the tail is produced by its own kernel
    dim3 grid((a.P + 255) / 256, (a.fl - 1) * a.batch, 2);
    hipLaunchKernelGGL(k_hist_generic<DT>, grid, dim3(256), 0, s, a, hist_shadow);
    (void)hipMemcpyAsync(a.hist, hist_shadow, sizeof(float) * (size_t)2 * a.h_side, hipMemcpyDeviceToDevice, s);
  }
}

}  // namespace cvvdp
