// Internal launch interface between core.cpp (planning) and the gfx950 kernels.
// Argument structs are passed by value as kernel arguments (they live in SGPRs / the kernarg
// segment, so per-band constants cost no LDS and no vector loads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cvvdp_hip.h"

namespace cvvdp {

constexpr float kEps = 0.00001f;  // safe_pow epsilon, cvvdp_metric.py:83

// Development knobs (A/B switches and tuning parameters read from the environment) exist only in a build made with
// `make EXTRA=-DCVVDP_DEV_KNOBS`.  In the product build every knob is its compiled-in default: the library reads no
// environment variable, so no stray setting can change the launch geometry (and with it the last bits of Q_per_ch).
#ifdef CVVDP_DEV_KNOBS
inline int dev_knob(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int dev_knob(const char*, int dflt) { return dflt; }
#endif

// Transcendentals on the gfx950 SFU: v_log_f32 / v_exp_f32 / v_rcp_f32 are 1-ulp, quarter-rate
// instructions.  pow(x,p) = exp2(p*log2(x)) has relative error ~ |p*log2 x| * 2^-24, i.e. <= 3e-6 over
// the operand ranges of this metric (x in [1e-5, 1e3], p <= 3.7) -- three orders of magnitude inside the
// JOD tolerance (tests/test_gpu_parity.py) and 10-20x cheaper than the correctly-rounded OCML powf.
#ifdef __HIPCC__
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_pow(float x, float p) { return fast_exp2(p * fast_log2(x)); }

// One output sample of the pyramid's expand along one axis (lpyr_dec.py:223-239: a 5-tap convolution of the zero-stuffed level), in the
// OPERATION ORDER OF THE REFERENCE'S conv2d: torch's CPU convolution accumulates the taps in order with fused multiply-adds starting from
// the first product (verified bit for bit on this torch build, tools/torch_conv_order.py), and the stuffed zeros contribute exactly
// nothing.  Even sample: taps 0, 2, 4 on coarse samples m-1, m, m+1; odd sample: taps 1, 3 on m, m+1.  At the two coarsest Laplacian
// bands (a few dozen pixels whose Laplacian is a 1e-4 relative difference of its operands) any other association of the same
// products moves Q_per_ch by up to 1.5 x the parity tolerance (profiles/r06_order_experiment.txt); left to the compiler, `a*e0 + b*e1 +
// c*e0` is contracted differently from kernel to kernel.
__device__ __forceinline__ float expand_even(float m0, float m1, float m2, float e0, float e1) {
  return __builtin_fmaf(m2, e0, __builtin_fmaf(m1, e1, m0 * e0));
}
__device__ __forceinline__ float expand_odd(float m1, float m2, float eo) { return __builtin_fmaf(m2, eo, m1 * eo); }
#endif
constexpr float kLog2_10 = 3.3219280948873623f;
constexpr float kLog10_2 = 0.30102999566398120f;

// ---------------------------------------------------------------- photometry + DKL (K0)
struct DisplayArgs {         // display model, shared by the image and the fused video kernels
  int32_t eotf, channels;   // CVVDP_EOTF_*; 1 or 3 colour channels in the source
  float Y_peak, Y_black, Y_refl, exposure, gamma, scale;  // scale = fp32(Y_peak - Y_black)
  float lin_lo;             // max(0.005, Y_black) for the linear EOTF
  float hlg_c;              // 0.5 - a*ln(4a)
  float m[9];               // RGB -> DKL
  // 8-bit sources, per-channel EOTFs (sRGB, PQ, gamma, linear): code -> emitted light of one channel, the whole per-channel part of
  // display_model.py:333-365 evaluated by the host once per code (core.cpp eotf_table) with the reference's fp32 operation order
  int32_t use_lut;
  float lut[256];
};
struct PhotoArgs {
  const void* src[2];     // test, ref
  int64_t sb[2], sc[2], sf[2], sh[2], sw[2];  // element strides
  int32_t dtype;
  int32_t H, W, batch, n_frames;
  DisplayArgs dm;
  float* dst;              // destination base
  int64_t d_side, d_ch, d_slot, d_b;  // element strides of the destination
  int32_t first_slot, n_slots;        // slot = (first_slot + f) % n_slots
};
void launch_photometry(const PhotoArgs& a, hipStream_t s);
struct PutPlanesArgs {       // pre-filtered 4-channel frames -> the 8 level-0 planes (cvvdp_metric.py:470-488)
  const float* src[2];      // test, ref: [B, 4, n, H, W] with element strides
  int64_t sb[2], sc[2], sf[2], sh[2], sw[2];
  int32_t H, W, batch, n_frames;
  float* dst;               // level-0 planes [plane][item][P]
  int64_t o_plane;          // items_cap * P
};
void launch_put_planes(const PutPlanesArgs& a, hipStream_t s);

// ---------------------------------------------------------------- temporal FIR (K1)
constexpr int CVVDP_ROT_TAPS = 64;     // k_fir_rot's tap table per channel: 2 (fl-1) rotated weights (fl <= 31), the newest frame's weight at CVVDP_ROT_NEW
constexpr int CVVDP_ROT_NEW = 63;
struct YuvArgs {            // planar Y'CbCr sources (video_source_yuv.py:79-124, 147-223); frame = Y plane, U plane, V plane
  int32_t Wc, Hc;           // chroma plane size
  float inv_fx, inv_fy;     // 1 / up-sampling factor (1 or 0.5) per axis
  int64_t u_off, v_off;     // sample offsets of the chroma planes inside a frame
  float wy, oy, wc, oc;     // limited-range fixed point -> float: Y' = wy*code - oy, C = wc*code - oc
  float rv, gu, gv, bu;     // R' = Y' + rv*Cr, G' = Y' + gu*Cb + gv*Cr, B' = Y' + bu*Cb
};
struct YuvUnpackArgs {      // resize.hip: Y'CbCr frames -> fp32 R'G'B' planes [3][n_frames][H][W]
  const void* src;
  YuvArgs yuv;
  int32_t W, H, n_frames, bits16;
  int64_t frame_stride;     // samples
  float* out;
};
struct ResizeArgs {         // resize.hip: n_planes fp32 planes [Hs][Ws] -> [Hd][Wd], clipped to [0,1]
  const float* in;
  float* out;
  int32_t n_planes, Hs, Ws, Hd, Wd, mode;
  float sy, sx;             // (float)Hs / Hd, (float)Ws / Wd
};
void launch_yuv_unpack(const YuvUnpackArgs& a, hipStream_t s);
void launch_resize(const ResizeArgs& a, hipStream_t s);
struct FirArgs {
  const void* src[2];      // raw test / reference frames handed to this block
  int64_t sb[2], sc[2], sf[2], sh[2], sw[2];  // element strides (B, C, F, H, W)
  int32_t dtype;
  DisplayArgs dm;
  int32_t W, P, batch, n_frames, fl;
  int32_t raw_first;       // raw frame index of the block's first scored frame
  int32_t abs_first;       // clip index of that frame (window rotation phase)
  int32_t write_hist;      // store the last fl-1 DKL frames for the next block
  int32_t halo_run;        // hist_src is the run of raw frames raw_first-(fl-1) .. raw_first-1 (real halo frames of a shard)
  float* hist;             // [side][plane][slot][b][P]: DKL tail of the previous block
  int64_t h_side, h_plane, h_slot, h_b;
  float* out;              // level-0 planes [plane][item][P]
  int64_t o_plane;         // items_cap * P
  float taps[4 * CVVDP_MAX_FILTER_LEN];     // flipped: taps[c][k] multiplies window position k
  float taps_rot[4 * CVVDP_ROT_TAPS];       // k_fir_rot: [c][i < 2(fl-1)] = weight of window position i mod (fl-1), [c][CVVDP_ROT_NEW] = newest
  YuvArgs yuv;                              // used by the CVVDP_YUV* dtypes only
  int16_t hist_src[CVVDP_MAX_FILTER_LEN];   // window position k < fl-1: >= 0 raw frame index, < 0 history slot -1-e
};
static_assert(sizeof(FirArgs) <= 4096, "kernel arguments of the temporal kernels");
void launch_fir(const FirArgs& a, float* hist_shadow, hipStream_t s);
inline bool fir_has_register_window(int fl) { return fl == 7 || fl == 9 || fl == 13 || fl == 15 || fl == 17 || fl == 25 || fl == 31; }
// Filter length the FIR kernels are run with: a filter that has no register-window instantiation of its own runs on the
// next longer one with zero taps on the oldest window positions (exact: the extra products are +-0), instead of on the
// generic kernel, which re-reads and re-converts every tap (100 fps, 27 taps, 4K x 64: 70 ms vs 14 ms).
inline int fir_kernel_len(int fl) {
  const int have[] = {7, 9, 13, 15, 17, 25, 31};
  for (int k : have) if (fl > 1 && fl <= k) return k;
  return fl;
}

// ---------------------------------------------------------------- gaussian pyramid reduce (K2)
struct ReduceArgs {
  const float* in;   // [planes*items][H*W]
  float* out;        // [planes*items][Ho*Wo]
  int32_t H, W, Ho, Wo, n_img;   // n_img = images actually processed per plane
  int32_t img_cap;               // images allocated per plane of the input level (plane stride = img_cap * size)
  int32_t img_cap_out;           // ... and of the output level (the two differ for level 0 of clips scored in pieces, core.cpp)
  int32_t n_planes;
  float k[5];
};
void launch_reduce(const ReduceArgs& a, hipStream_t s);
struct Reduce2Args {         // two levels per pass: l -> l+1 -> l+2
  const float* in;           // level l
  float *out1, *out2;        // levels l+1, l+2
  int32_t H, W, H1, W1, H2, W2, n_img, img_cap, img_cap_out, n_planes;   // img_cap: input level, img_cap_out: both output levels
  int32_t seg2;              // level-(l+2) rows per thread (set by launch_reduce2)
  float k[5];
};
bool reduce2_supported(int H, int W);
// Levels of at most this many samples are reduced by k_reduce, which reproduces the reference's operation order bit for bit (pyramid.hip);
// larger ones by the marching kernels.  128 x 128: from there down a level's band has so few pixels that single roundings of the Gaussian
// level show in Q_per_ch (the thinnest class of the randomised sweeps, tests/test_fuzz_goldens.py).
constexpr int64_t kReduceRefPixels = 16384;
bool reduce_takes_ref_kernel(int H, int W);
void launch_reduce2(const Reduce2Args& a, hipStream_t s);

// ---------------------------------------------------------------- fused band kernel (K3..K7)
struct BandArgs {
  const float* g;    // level l   planes [2*nch][items_cap][H*W]
  const float* gc;   // level l+1 planes [2*nch][items_cap_c][Hc*Wc]
  int32_t H, W, Hc, Wc;
  int32_t items, items_cap, items_cap_c, nch;   // items_cap_c: items allocated per plane of level l+1 (gc, g1_out)
  int32_t seg_h, n_seg, n_strip;
  int32_t per_xcd;   // k_band4: work units per XCD (set by launch_band4)
  int32_t strip0, n_strip_l;   // k_band4: this launch covers strips strip0 .. strip0 + n_strip_l - 1 (set by launch_band4)
  float band_mul;
  float lut[4 * CVVDP_CSF_NODES];
  float logL_first, logL_last;
  float sens_mul;
  float ch_gain[4];
  float mask_c10, mask_p, eps_p;
  float q[4], eps_q[4];
  float xw[16];
  float dmax;
  float m1[4];       // k_band4: 1 - sum_k xw[k][c]*eps^q_k (the "1 +" of the mask with the eps terms of safe_pow folded in)
  float inv_dmax;    // k_band4: 1 / dmax
  float blur_h[13];  // k_band4: blur taps * 10^mask_c (horizontal pass)
  float ind_k1, ind_k0;   // k_band4: CSF-LUT position = log2(L) * ind_k1 - ind_k0
  float blur[13];
  float kx[3];       // expand taps: 2*K[0], 2*K[2] (even), 2*K[1] (odd)
  float* partial;    // [items][n_strip*n_seg][4]
  float* dchr;       // heat band [items_cap][H*W] or null
  uint32_t* hstats;  // fused level 0 of a colour-mapped heat-map clip: per item kHeatStatsWords words; the waves of channel 0 take the minimum
                     // positive and the maximum sample of the test plane they stream anyway (the context image's range, k_heat_range), or null
  float hw[4];       // heat channel weights
  float beta_tch, eps_btch, eps_inv_btch;
  float* ddump;      // debug [4][items_cap][H*W] or null
  float* fdump;      // features: |T'|, |R'| as [2][4][items_cap][H*W] or null (k_band only)
  // k_band4 in features mode: column sums of |T'|, |T'|^2, |R'|, |R'|^2, D, D^2 per piece of a cell row,
  // [items][nch][f_pieces][6][W] (piece = cell row + segment index; f_pieces = ceil(H / fs) + n_seg), or null
  float* fsum;
  int32_t fs, f_pieces;
  // k_band4f (band4f.hip): the level's reduce fused in -- level l+1 planes [2*nch][items_cap][Hc*Wc] written by the band kernel, reduce taps
  float* g1_out;
  float rk[5];
  int32_t one_wave_layout;   // launch_band4f: the border-free strips on k_band4f<4, 0> (one wave per channel) instead of k_band4s (front / back waves)
};
void launch_band(const BandArgs& a, bool blur, hipStream_t s);
// vectorised variant: any W >= 16, blur on, seg_h even.  W % 8 != 0: every strip on the RAGGED instantiation, or (split_edge) the
// strips that stay clear of the right image edge on the aligned instantiation on s and the one or two edge strips on the RAGGED
// one on s_edge (a side stream the caller has made wait for the pyramid and will join before reading the partial sums;
// s_edge == s: one after the other)
void launch_band4(const BandArgs& a, bool split_edge, hipStream_t s, hipStream_t s_edge);
int band4_edge_strips(int W, int n_strip);   // how many trailing strips the RAGGED instantiation takes (0 when W % 8 == 0)
// k_band4 with the level's 5x5 reduce fused in: computes level l+1 from the rows it streams and writes it to a.g1_out instead of
// reading a.gc (which may be the same buffer).  W % 8 == 0, no heat map / dump / features.
bool band4f_supported(int H, int W);
// (the strips at the left / right image border as their own launch on s_edge -- a side stream ordered like launch_band4's, or s)
void launch_band4f(const BandArgs& a, hipStream_t s, hipStream_t s_edge);
// band4s.hip: the border-free strips of a fused level (a.strip0 .. a.strip0 + a.n_strip_l - 1) with the work divided between front and back waves
void launch_band4s(const BandArgs& a, hipStream_t s);
// ... and its border strips (strip 0 and the last a.n_strip_l - 1; W % 4 == 0, plain or heat map) on the same layout's EDGE body
void launch_band4s_edge(const BandArgs& a, hipStream_t s);
constexpr int kBand4StripWidth = 240;
// How each band translation unit was compiled, for cvvdp_build_flags(): CVVDP_BUILD_SAFE_LOADS (-DCVVDP_SAFE_LOADS) and CVVDP_BUILD_DIAG
// (a timing-only switch or a non-default ring / load-hint macro: results wrong or not the product's).  0 for the product build.
int tu_flags_band4();
int tu_flags_band4f();
int tu_flags_band4s();

struct BaseArgs {
  const float* g;    // baseband planes [2*nch][items_cap][P]
  int32_t H, W, items, items_cap, nch;
  float lut[4 * CVVDP_CSF_NODES];
  float logL_first, logL_last, sens_mul;
  float* q_out;      // Q_per_ch base
  int32_t q_frames, q_levels, q_frame_offset, level, batch;
  float* dchr; float hw[4]; float beta_tch, eps_btch, eps_inv_btch;
  float* ddump;
  float* fdump;      // features: |T_f|*S, |R_f|*S as [2][4][items_cap][P] or null
};
void launch_baseband(const BaseArgs& a, hipStream_t s);

struct FinalizeArgs {
  const float* partial;  // [items][nblk][4]
  int32_t items, nblk, nch, P;
  float* q_out; int32_t q_frames, q_levels, q_frame_offset, level, batch;
  double sub_per_term;   // k_band4 sums (D + eps)^2 over the level's P pixels: eps^2 per term comes off the mean; k_band sums (D+eps)^2 - eps^2: 0
};
void launch_finalize(const FinalizeArgs& a, hipStream_t s);

struct FeatFinishArgs {      // cvvdp_feature_pooling from the column sums k_band4 keeps in features mode
  const float* fsum;         // [items][nch][f_pieces][6][W]
  int32_t H, W, items, nch, fs, Hc, Wc, f_pieces, seg_h;
  float inv_gain[4];
  float* out;                // [items][Hc][Wc][nch][6]
};
void launch_feature_finish(const FeatFinishArgs& a, hipStream_t s);
struct FeatPoolArgs {        // cvvdp_feature_pooling, cvvdp_ml_metric.py:77-107
  const float* tr;           // |T'|, |R'|: [2][4][items_cap][P]
  const float* d;            // D: [4][items_cap][P]
  int32_t H, W, items, items_cap, nch, fs, Hc, Wc;
  float inv_gain[4];         // 1 / ch_gain (the kernels' T', R' carry the channel gain, the reference's features do not); 1 for the baseband
  float* out;                // [items][Hc][Wc][nch][6]
};
void launch_feature_pool(const FeatPoolArgs& a, hipStream_t s);

// ---------------------------------------------------------------- pooling + JOD (K8)
struct PoolArgs {
  const float* q; int32_t B, C, F, L;
  float ch_w[4], bb_w[4];
  float beta_sch, beta_tch, beta_t, jod_a, jod_exp, image_int;
  float* jod;
};
void launch_pool(const PoolArgs& a, hipStream_t s);

// ---------------------------------------------------------------- heat map (K9, K10)
struct ExpandAddArgs {
  float* fine; const float* coarse; int32_t H, W, Hc, Wc, n_img; float kx[3];
  int32_t per_thread_layout;   // 1: k_expand_add4 also where the row-tile kernel applies (A/B switch of tests: cvvdp_clip.band_layout == 1)
};
void launch_expand_add(const ExpandAddArgs& a, hipStream_t s);

struct HeatArgs {
  const float* recon;     // [items][P] reconstructed difference pyramid (level-0 heat band) -- or, with `coarse` set, the level-0 BAND alone:
                          // the last step of the reconstruction (lpyr_dec.py:328-335: + expand of the level-1 reconstruction) is then done here,
                          // 4 pixels per thread with k_expand_add4's expressions, and the read-modify-write pass over level 0 does not exist
  const float* coarse;    // [items][Hc*Wc] level-1 reconstruction, or null
  int32_t H, W, Hc, Wc;   // (coarse != null: W % 4 == 0)
  float kx[3];
  const float* ctx;       // context image: test Y-sustained level-0 plane [items][P] (cvvdp_metric.py:400)
  int32_t P, items, mode;
  float jod_a, jod_exp;
  float jod_lin;          // jod_a * 0.1^(jod_exp - 1): slope of met2jod's linear part (cvvdp_metric.py:652-655)
  uint32_t* stats;        // per item kHeatStatsWords words: [0] min positive y (bits), [1] max y (bits), [4..) histogram
  float* curve;           // per item 1024 tone-curve values + [1024]=b_min, [1025]=b_max, [1026]=flag(1: histogram curve)
  void* out;              // fp16 [ch][items][P], or (out_u8) uint8 [items][P][ch]
  int32_t out_u8;
  int32_t range_done;     // stats[0], [1] were initialised before and filled by the level-0 band kernel (BandArgs::hstats): no k_heat_init / k_heat_range
  int32_t n_nodes;        // colour map nodes (5 threshold, 3 supra-threshold)
  float cin[5];           // node positions
  float cch[15];          // node colours / luminance, [node][rgb]  (visualize_diff_map.py:93-94)
  int32_t pixel_layout;   // 1: k_heat_colour<VEC> (a thread = 4 pixels anywhere) also where the row-tile kernel k_heat_colour_rows applies:
                          // the A/B switch of tests (cvvdp_clip.band_layout == 1); the two produce the same bits
};
void launch_heat_raw(const HeatArgs& a, hipStream_t s);
void launch_heat_init(uint32_t* stats, int items, hipStream_t s);     // min / max / histogram words of `items` frames to their start values
void launch_heat_colour(const HeatArgs& a, hipStream_t s);
constexpr int kHeatStatsWords = 4 + 1024;
constexpr int kHeatCurveWords = 1024 + 4;

}  // namespace cvvdp
