// Host-side sanitizer / out-of-bounds harness of the C ABI (VERDICT r2 "next" #9).  Test infrastructure, CPU only.
//
// csrc/core.cpp (planning of the workspace, of the pyramid geometry, of segments and strips; construction of every kernel's
// argument struct: tap tables, history tables, CSF rows, plane offsets) is compiled UNCHANGED with
// -fsanitize=address,undefined and linked with this file instead of the HIP runtime and the kernels:
//   * the hip* calls succeed and do nothing (streams / events are counted so that leaks show up),
//   * every launch_* stub checks that each buffer the kernel would touch lies inside the bound workspace (extents derived from
//     the struct's own geometry), that strips x segments cover the level, and that the tables are within their arrays.
// The driver sweeps a few thousand random clip configurations (sizes 2..4500, every frame rate, block sizes, heat-map modes,
// features mode, batches, shards) through the whole call sequence of the Python mirror.  Exit code 0 = no finding.
//
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
//       colorvideovdp_amd/csrc/core.cpp tests/native/host_sanitize.cpp -o host_sanitize && ./host_sanitize [n_cases] [seed]
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../colorvideovdp_amd/csrc/kernels.h"

// ---------------------------------------------------------------- fake HIP runtime
static int g_live_streams = 0, g_live_events = 0;
extern "C" {
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipRuntimeGetVersion(int* v) { *v = HIP_VERSION; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(new int(0)); ++g_live_events; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<int*>(e); --g_live_events; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { *reinterpret_cast<int*>(e) = 1; return hipSuccess; }   // (touches the event: a stale handle trips ASAN)
hipError_t hipEventSynchronize(hipEvent_t e) { return *reinterpret_cast<int*>(e) >= 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = 0.5f + *reinterpret_cast<int*>(a) + *reinterpret_cast<int*>(b); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(new int(0)); ++g_live_streams; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<int*>(s); --g_live_streams; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return *reinterpret_cast<int*>(e) >= 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t) { return hipSuccess; }
}

// ---------------------------------------------------------------- bounds bookkeeping
static const float* g_ws = nullptr;
static size_t g_ws_floats = 0;
static long g_checks = 0, g_launches = 0;
static char g_case[256];

[[noreturn]] static void die(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  fprintf(stderr, "host_sanitize: FINDING in case {%s}: ", g_case);
  vfprintf(stderr, fmt, ap); fprintf(stderr, "\n");
  va_end(ap);
  abort();
}
static void in_ws(const void* p, size_t n_floats, const char* what) {
  ++g_checks;
  const float* f = static_cast<const float*>(p);
  if (!p) die("%s: null", what);
  if (f < g_ws || f + n_floats > g_ws + g_ws_floats) die("%s: [%td, %td) floats outside the workspace of %zu", what, f - g_ws, f - g_ws + (ptrdiff_t)n_floats, g_ws_floats);
}
#define REQUIRE(c, ...) do { if (!(c)) die(__VA_ARGS__); } while (0)

namespace cvvdp {

static void chk_display(const DisplayArgs& d) { REQUIRE(d.channels == 1 || d.channels == 3, "display: channels %d", d.channels); }

void launch_photometry(const PhotoArgs& a, hipStream_t) {
  ++g_launches; chk_display(a.dm);
  // destination: [side 2][ch 3]... strides given; the farthest element written
  const size_t far = (size_t)a.d_side + 2 * (size_t)a.d_ch + (size_t)(a.batch - 1) * a.d_b + (size_t)a.H * a.W;
  in_ws(a.dst, far, "photometry dst");
}
void launch_put_planes(const PutPlanesArgs& a, hipStream_t) { ++g_launches; in_ws(a.dst, 7 * (size_t)a.o_plane + (size_t)a.n_frames * a.batch * a.H * a.W, "put_planes dst"); }
void launch_yuv_unpack(const YuvUnpackArgs& a, hipStream_t) { ++g_launches; REQUIRE(a.out && a.src, "yuv unpack: null"); }
void launch_resize(const ResizeArgs& a, hipStream_t) { ++g_launches; REQUIRE(a.in && a.out && a.n_planes > 0, "resize: bad args"); }

void launch_fir(const FirArgs& a, float* hist_shadow, hipStream_t) {
  ++g_launches; chk_display(a.dm);
  REQUIRE(a.fl >= 1 && a.fl <= CVVDP_MAX_FILTER_LEN, "fir: fl %d", a.fl);
  REQUIRE(a.n_frames >= 1 && a.fl - 1 + a.n_frames <= CVVDP_MAX_WINDOW, "fir: window %d + %d", a.fl - 1, a.n_frames);
  const int kl = fir_kernel_len(a.fl);
  REQUIRE(kl >= a.fl && kl <= CVVDP_MAX_FILTER_LEN, "fir: kernel length %d for fl %d", kl, a.fl);
  if (kl <= 31) REQUIRE(2 * (kl - 1) <= CVVDP_ROT_NEW, "fir: rotating tap table of %d entries per channel", 2 * (kl - 1));
  in_ws(a.out, 7 * (size_t)a.o_plane + (size_t)a.n_frames * a.batch * a.P, "fir level-0 planes");
  bool uses_hist = a.write_hist != 0;
  for (int k = 0; k < a.fl - 1; ++k) {
    const int e = a.hist_src[k];
    if (e < 0) { uses_hist = true; REQUIRE(-1 - e < kl - 1, "fir: history slot %d of %d", -1 - e, kl - 1); }
  }
  if (uses_hist) {
    REQUIRE(a.hist != nullptr, "fir: history needed but null");
    in_ws(a.hist, (size_t)a.h_side + 2 * (size_t)a.h_plane + (size_t)(kl - 2) * a.h_slot + (size_t)(a.batch - 1) * a.h_b + a.P, "fir DKL tail");
  }
  if (hist_shadow) in_ws(hist_shadow, 1, "fir shadow tail");
}

static void chk_reduce_geom(int H, int W, int Ho, int Wo, const char* what) {
  REQUIRE(Ho == (H + 1) / 2 && Wo == (W + 1) / 2, "%s: %dx%d -> %dx%d", what, W, H, Wo, Ho);
}
void launch_reduce(const ReduceArgs& a, hipStream_t) {
  ++g_launches; chk_reduce_geom(a.H, a.W, a.Ho, a.Wo, "reduce");
  REQUIRE(a.n_img <= a.img_cap && a.n_img <= a.img_cap_out, "reduce: %d images of %d -> %d", a.n_img, a.img_cap, a.img_cap_out);
  in_ws(a.in, ((size_t)(a.n_planes - 1) * a.img_cap + a.n_img) * a.H * a.W, "reduce in");       // (level 0 of a clip scored in pieces: a.in points at the piece)
  in_ws(a.out, (size_t)a.n_planes * a.img_cap_out * a.Ho * a.Wo, "reduce out");
}
bool reduce_takes_ref_kernel(int H, int W) { return (int64_t)H * W <= kReduceRefPixels || W < 16 || H < 4; }   // (mirror of pyramid.hip)
bool reduce2_supported(int H, int W) {
  return (W % 16 == 0 || W >= 32) && H >= 8 && !reduce_takes_ref_kernel(H, W) && !reduce_takes_ref_kernel((H + 1) / 2, (W + 1) / 2);
}
void launch_reduce2(const Reduce2Args& a, hipStream_t) {
  ++g_launches; chk_reduce_geom(a.H, a.W, a.H1, a.W1, "reduce2 l+1"); chk_reduce_geom(a.H1, a.W1, a.H2, a.W2, "reduce2 l+2");
  REQUIRE(reduce2_supported(a.H, a.W), "reduce2 launched on %dx%d", a.W, a.H);
  REQUIRE(a.n_img <= a.img_cap && a.n_img <= a.img_cap_out, "reduce2: %d images of %d -> %d", a.n_img, a.img_cap, a.img_cap_out);
  in_ws(a.in, ((size_t)(a.n_planes - 1) * a.img_cap + a.n_img) * a.H * a.W, "reduce2 in");
  in_ws(a.out1, (size_t)a.n_planes * a.img_cap_out * a.H1 * a.W1, "reduce2 out1");
  in_ws(a.out2, (size_t)a.n_planes * a.img_cap_out * a.H2 * a.W2, "reduce2 out2");
}

static void chk_band(const BandArgs& a, int strip_w, const char* what) {
  ++g_launches;
  const size_t P = (size_t)a.H * a.W, Pc = (size_t)a.Hc * a.Wc;
  REQUIRE(a.nch == 3 || a.nch == 4, "%s: nch %d", what, a.nch);
  REQUIRE(a.items >= 1 && a.items <= a.items_cap && a.items <= a.items_cap_c, "%s: %d items of %d / %d", what, a.items, a.items_cap, a.items_cap_c);
  REQUIRE(a.Hc == (a.H + 1) / 2 && a.Wc == (a.W + 1) / 2, "%s: coarse level %dx%d of %dx%d", what, a.Wc, a.Hc, a.W, a.H);
  REQUIRE((int64_t)a.n_strip * strip_w >= a.W && (int64_t)(a.n_strip - 1) * strip_w < a.W, "%s: %d strips of %d over W %d", what, a.n_strip, strip_w, a.W);
  REQUIRE((int64_t)a.n_seg * a.seg_h >= a.H && (int64_t)(a.n_seg - 1) * a.seg_h < a.H, "%s: %d segments of %d rows over H %d", what, a.n_seg, a.seg_h, a.H);
  in_ws(a.g, ((size_t)(2 * a.nch - 1) * a.items_cap + a.items) * P, "band g");       // (level 0 of a clip scored in pieces: a.g points at the piece)
  in_ws(a.gc, 2 * (size_t)a.nch * a.items_cap_c * Pc, "band gc");
  in_ws(a.partial, (size_t)a.items * a.n_strip * a.n_seg * 4, "band partial sums");
  if (a.dchr) in_ws(a.dchr, (size_t)a.items * P, "band heat band");
  if (a.hstats) in_ws(a.hstats, (size_t)a.items * kHeatStatsWords, "band heat-map range words");
  if (a.ddump) in_ws(a.ddump, 4 * (size_t)a.items_cap * P, "band D dump");
  if (a.fdump) in_ws(a.fdump, 8 * (size_t)a.items_cap * P, "band |T'|,|R'| planes");
  if (a.fsum) {
    REQUIRE(a.fs >= 1 && a.f_pieces >= (a.H + a.fs - 1) / a.fs + a.n_seg, "%s: %d pieces for %d cell rows + %d segments", what, a.f_pieces, (a.H + a.fs - 1) / a.fs, a.n_seg);
    REQUIRE((a.H - 1) / a.fs + (a.n_seg - 1) < a.f_pieces, "%s: last piece index", what);
    in_ws(a.fsum, (size_t)a.items * a.nch * a.f_pieces * 6 * a.W, "band feature column sums");
  }
}
void launch_band(const BandArgs& a, bool blur, hipStream_t) { chk_band(a, blur ? 256 - 12 : 256, "k_band"); }
int band4_edge_strips(int W, int n_strip) {   // (mirror of band4.hip)
  if ((W & 7) == 0) return 0;
  int n = 0;
  while (n < n_strip && (n_strip - 1 - n) * 240 + 240 + 8 > W) ++n;
  return n;
}
void launch_band4(const BandArgs& a, bool split_edge, hipStream_t s, hipStream_t s_edge) {
  chk_band(a, kBand4StripWidth, "k_band4");
  REQUIRE(a.W >= 16 && a.H >= 16 && a.seg_h % 2 == 0, "k_band4 on %dx%d, seg_h %d", a.W, a.H, a.seg_h);
  if (split_edge) { const int n = band4_edge_strips(a.W, a.n_strip); REQUIRE(n > 0 && n < a.n_strip, "k_band4: split with %d edge strips of %d", n, a.n_strip); }
  else REQUIRE(s == s_edge, "k_band4: side stream without a split");
}
bool band4f_supported(int H, int W) { return (W & 1) == 0 && W >= 32 && H >= 32; }   // (mirror of band4f.hip)
int tu_flags_band4() { return 0; }
int tu_flags_band4f() { return 0; }
int tu_flags_band4s() { return 0; }
void launch_band4f(const BandArgs& a, hipStream_t, hipStream_t) {
  chk_band(a, kBand4StripWidth, "k_band4f");
  REQUIRE(band4f_supported(a.H, a.W) && a.nch == 4 && a.seg_h % 2 == 0 && a.seg_h >= 8, "k_band4f on %dx%d, %d channels, seg_h %d", a.W, a.H, a.nch, a.seg_h);
  REQUIRE(!a.ddump && !a.fdump && !(a.dchr && a.fsum), "k_band4f with a dump / per-pixel feature buffer");       // (heat-map band / column sums: the HEAT / FEAT instantiations)
  in_ws(a.g1_out, 2 * (size_t)a.nch * a.items_cap_c * a.Hc * a.Wc, "k_band4f level l+1 planes");
  REQUIRE(a.g1_out == a.gc, "k_band4f writes another buffer than the next level's planes");
}
void launch_baseband(const BaseArgs& a, hipStream_t) {
  ++g_launches;
  const size_t P = (size_t)a.H * a.W;
  in_ws(a.g, ((size_t)(2 * a.nch - 1) * a.items_cap + a.items) * P, "baseband g");
  REQUIRE(a.level == a.q_levels - 1 && a.q_frame_offset + a.items / a.batch <= a.q_frames, "baseband: Q window");
  in_ws(a.q_out, (size_t)a.batch * a.nch * a.q_frames * a.q_levels, "baseband Q_per_ch");
  if (a.dchr) in_ws(a.dchr, (size_t)a.items * P, "baseband heat band");
  if (a.ddump) in_ws(a.ddump, 4 * (size_t)a.items_cap * P, "baseband D dump");
  if (a.fdump) in_ws(a.fdump, 8 * (size_t)a.items_cap * P, "baseband |T'|,|R'| planes");
}
void launch_finalize(const FinalizeArgs& a, hipStream_t) {
  ++g_launches;
  in_ws(a.partial, (size_t)a.items * a.nblk * 4, "finalize partial sums");
  REQUIRE(a.level >= 0 && a.level < a.q_levels - 1 && a.q_frame_offset + a.items / a.batch <= a.q_frames, "finalize: Q window");
  in_ws(a.q_out, (size_t)a.batch * a.nch * a.q_frames * a.q_levels, "finalize Q_per_ch");
}
void launch_feature_finish(const FeatFinishArgs& a, hipStream_t) {
  ++g_launches;
  in_ws(a.fsum, (size_t)a.items * a.nch * a.f_pieces * 6 * a.W, "feature finish column sums");
  REQUIRE(a.out && a.Hc == (a.H + a.fs - 1) / a.fs && a.Wc == (a.W + a.fs - 1) / a.fs, "feature finish: cells");
  REQUIRE((a.H - 1) / a.fs + (a.H - 1) / a.seg_h < a.f_pieces, "feature finish: piece %d of %d", (a.H - 1) / a.fs + (a.H - 1) / a.seg_h, a.f_pieces);
}
void launch_feature_pool(const FeatPoolArgs& a, hipStream_t) {
  ++g_launches;
  const size_t P = (size_t)a.H * a.W;
  in_ws(a.tr, 8 * (size_t)a.items_cap * P, "feature pool |T'|,|R'|");
  in_ws(a.d, 4 * (size_t)a.items_cap * P, "feature pool D");
  REQUIRE(a.out != nullptr, "feature pool: null out");
}
void launch_pool(const PoolArgs& a, hipStream_t) { ++g_launches; REQUIRE(a.q && a.jod && a.B >= 1 && a.C >= 1 && a.F >= 1 && a.L >= 1, "pool: bad args"); }
void launch_expand_add(const ExpandAddArgs& a, hipStream_t) {
  ++g_launches;
  REQUIRE(a.Hc == (a.H + 1) / 2 && a.Wc == (a.W + 1) / 2, "expand_add: geometry");
  in_ws(a.fine, (size_t)a.n_img * a.H * a.W, "expand_add fine"); in_ws(a.coarse, (size_t)a.n_img * a.Hc * a.Wc, "expand_add coarse");
}
static void chk_heat(const HeatArgs& a) {
  ++g_launches;
  in_ws(a.recon, (size_t)a.items * a.P, "heat recon");
  if (a.coarse) { REQUIRE(a.W % 4 == 0 && (size_t)a.H * a.W == (size_t)a.P && a.Hc == (a.H + 1) / 2 && a.Wc == (a.W + 1) / 2, "heat: fused reconstruction geometry"); in_ws(a.coarse, (size_t)a.items * a.Hc * a.Wc, "heat level-1 reconstruction"); }
  if (a.ctx) in_ws(a.ctx, (size_t)a.items * a.P, "heat context");
  if (a.stats) in_ws(a.stats, (size_t)a.items * kHeatStatsWords, "heat stats");
  if (a.curve) in_ws(a.curve, (size_t)a.items * kHeatCurveWords, "heat curve");
  REQUIRE(a.out != nullptr && a.n_nodes <= 5, "heat: out / nodes");
}
void launch_heat_raw(const HeatArgs& a, hipStream_t) { chk_heat(a); }
void launch_heat_init(uint32_t* stats, int items, hipStream_t) { ++g_launches; in_ws(stats, (size_t)items * kHeatStatsWords, "heat stats (init)"); }
void launch_heat_colour(const HeatArgs& a, hipStream_t) { chk_heat(a); }

}  // namespace cvvdp

// ---------------------------------------------------------------- driver
static cvvdp_params make_params(std::mt19937& rng) {
  cvvdp_params p{};
  p.eotf = (int)(rng() % 5);
  p.Y_peak = 200.0f; p.Y_black = 0.2f; p.Y_refl = 0.4f; p.exposure = 1.0f; p.gamma = 2.2f;
  for (int i = 0; i < 9; ++i) p.rgb2dkl[i] = 0.1f * (i + 1);
  p.mask_p = 2.26f; p.mask_c10 = 0.3f;
  for (int i = 0; i < 4; ++i) { p.mask_q[i] = 1.3f + i; p.ch_gain[i] = 1.0f; p.ch_w[i] = 1.0f; p.baseband_weight[i] = 1.0f; }
  for (int i = 0; i < 16; ++i) p.xcm[i] = 0.1f;
  p.d_max10 = 100.0f; p.sens_mul = 1.0f; p.blur_radius = 6;
  for (int i = 0; i < 13; ++i) p.blur_taps[i] = 1.0f / 13;
  p.beta = 2.0f; p.beta_t = 2.0f; p.beta_tch = 2.0f; p.beta_sch = 2.0f; p.jod_a = 0.04f; p.jod_exp = 0.9f; p.image_int = 0.5f;
  p.csf_logL_first = -2.0f; p.csf_logL_last = 4.0f;
  return p;
}

// Which clips take the band kernels that compute the next level themselves (core.cpp, cvvdp_configure): the rule as a table --
// plain video scoring only, >= 16 M pixels of a level in the nominal block, more than 1024 level-0 workgroups, even widths.
static void check_fuse_rule(std::mt19937& rng) {
  struct Row { int w, h, frames, levels, video, heat, fs, dump, mode, want; };
  const Row rows[] = {
      {3840, 2160, 64, 9, 1, 0, 0, 0, 0, 3},   // the bench clip
      {3840, 2160, 1024, 9, 1, 0, 0, 0, 0, 3}, // configs[3]
      {7680, 4320, 256, 10, 1, 0, 0, 0, 0, 4}, // 8K: 8294400 x 64 pixels still at level 3
      {1920, 1080, 64, 8, 1, 0, 0, 0, 0, 2},
      {1360, 768, 64, 8, 1, 0, 0, 0, 0, 0},    // one round of workgroups: independent levels on three streams win
      {3840, 2160, 8, 9, 1, 0, 0, 0, 0, 0},    // short clips
      {3840, 2160, 16, 9, 1, 0, 0, 0, 0, 2},
      {2566, 1444, 40, 9, 1, 0, 0, 0, 0, 1},   // W % 4 == 2; level 1 is 1283 columns wide (odd)
      {3841, 2160, 64, 9, 1, 0, 0, 0, 0, 0},   // odd width
      {3840, 2160, 64, 9, 1, 2, 0, 0, 0, 3},   // heat map: the HEAT instantiations of the fused kernels (k_band4s_heat / k_band4f_heat)
      {7680, 4320, 256, 10, 1, 3, 0, 0, 0, 4}, // configs[4]
      // per-pixel dump: k_band4's instantiations
      {3840, 2160, 64, 9, 1, 0, 38, 0, 0, 3},  // features: the FEAT instantiations
      {3840, 2160, 64, 9, 1, 0, 0, 1, 0, 0},
      {3840, 2160, 1, 9, 0, 0, 0, 0, 0, 0},    // an image
      {3840, 2160, 64, 9, 1, 0, 0, 0, 2, 0},   // test hook: never / wherever possible
      {256, 144, 5, 6, 1, 0, 0, 0, 1, 3},      // 256x144 -> 128x72 -> 64x36 (32x18 is below 32 rows)
  };
  cvvdp_params p = make_params(rng);
  cvvdp_handle* h = nullptr;
  REQUIRE(cvvdp_create(&p, &h) == CVVDP_OK && h, "create failed");
  REQUIRE(cvvdp_fused_levels(h) == -1, "fused levels of a handle that is not configured");
  for (const Row& r : rows) {
    cvvdp_clip c{};
    c.width = r.w; c.height = r.h; c.batch = 1; c.channels = 3; c.is_video = r.video;
    c.filter_len = r.video ? 17 : 1; c.total_frames = c.n_frames = r.frames; c.block_frames = r.video ? std::min(64, r.frames) : 1;
    c.n_levels = r.levels; c.heatmap = r.heat; c.feature_size = r.fs; c.debug_dump = r.dump; c.fuse_mode = r.mode; c.raw_halo = 1;
    for (int i = 0; i < 4 * CVVDP_MAX_FILTER_LEN; ++i) c.taps[i] = 0.01f * (i % 7);
    for (auto& v : c.csf_rows) v = 1.0f;
    snprintf(g_case, sizeof g_case, "fuse rule %dx%dx%d", r.w, r.h, r.frames);
    REQUIRE(cvvdp_configure(h, &c) == CVVDP_OK, "configure: %s", cvvdp_last_error(h));
    REQUIRE(cvvdp_fused_levels(h) == r.want, "fused levels %d, expected %d", cvvdp_fused_levels(h), r.want);
  }
  cvvdp_destroy(h);
}

int main(int argc, char** argv) {
  const int n_cases = argc > 1 ? atoi(argv[1]) : 3000;
  std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 1u);
  check_fuse_rule(rng);
  auto ri = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
  long refused = 0, ran = 0;
  float* const fake_ws = reinterpret_cast<float*>(uintptr_t(1) << 40);     // never dereferenced: all checks are address arithmetic
  const void* const fake_src = reinterpret_cast<const void*>(uintptr_t(1) << 44);
  void* const fake_out = reinterpret_cast<void*>(uintptr_t(3) << 44);
  for (int k = 0; k < n_cases; ++k) {
    cvvdp_params p = make_params(rng);
    cvvdp_handle* h = nullptr;
    REQUIRE(cvvdp_create(&p, &h) == CVVDP_OK && h, "create failed");
    const int reuse = ri(1, 3);                       // a handle is re-configured between clips, like the Python mirror does
    for (int u = 0; u < reuse; ++u) {
      cvvdp_clip c{};
      const int big = ri(0, 9);
      c.width = big == 0 ? ri(2000, 4500) : (big < 4 ? ri(2, 40) : ri(16, 1400));
      c.height = big == 0 ? ri(1000, 2400) : (big < 4 ? ri(2, 40) : ri(16, 900));
      c.batch = ri(0, 4) == 0 ? 2 : 1;
      c.channels = ri(0, 5) == 0 ? 1 : 3;
      c.is_video = ri(0, 3) != 0;
      const int fps_list[] = {24, 25, 30, 50, 60, 90, 100, 120, 144, 240};
      const int fps = fps_list[ri(0, 9)];
      c.filter_len = c.is_video ? (int)std::ceil(0.250 * fps / 2) * 2 + 1 : 1;          // cvvdp_metric.py:1059
      c.total_frames = c.is_video ? ri(1, 300) : 1;
      c.first_frame = c.is_video && ri(0, 3) == 0 ? ri(0, 50) : 0;
      c.n_frames = c.is_video ? ri(1, c.total_frames) : 1;
      c.block_frames = c.is_video ? std::max(1, std::min(ri(1, 64), CVVDP_MAX_WINDOW - c.filter_len + 1)) : 1;
      int lv = 1; { int hh = c.height, ww = c.width; while (lv < CVVDP_MAX_LEVELS && hh >= 4 && ww >= 4 && ri(0, 9) != 0) { hh = (hh + 1) / 2; ww = (ww + 1) / 2; ++lv; } }
      c.n_levels = ri(0, 30) == 0 ? ri(-1, CVVDP_MAX_LEVELS + 2) : lv;                   // sometimes out of range: must be refused
      c.heatmap = ri(0, 2) == 0 ? ri(1, 3) : 0;
      c.debug_dump = ri(0, 9) == 0;
      c.raw_halo = ri(0, 1);
      c.feature_size = (c.heatmap == 0 && ri(0, 3) == 0) ? ri(1, 90) : 0;
      c.fuse_mode = ri(0, 5) == 0 ? ri(0, 3) : 0;                                        // 3: out of range, must be refused
      c.band_layout = ri(0, 7) == 0 ? ri(0, 2) : 0;                                      // 2: out of range
      if (c.is_video && ri(0, 3) == 0) { c.defer_bands = 1; c.score_frames = ri(0, c.block_frames + 1); }   // 0 and block + 1: out of range
      for (int i = 0; i < 4 * CVVDP_MAX_FILTER_LEN; ++i) c.taps[i] = 0.01f * (i % 7);
      for (auto& v : c.csf_rows) v = 1.0f;
      snprintf(g_case, sizeof g_case, "#%d %dx%d B%d C%d video %d fl %d frames %d/%d block %d levels %d heat %d dump %d halo %d fs %d eotf %d", k, c.width,
               c.height, c.batch, c.channels, c.is_video, c.filter_len, c.n_frames, c.total_frames, c.block_frames, c.n_levels, c.heatmap, c.debug_dump,
               c.raw_halo, c.feature_size, p.eotf);
      if (cvvdp_configure(h, &c) != CVVDP_OK) { REQUIRE(cvvdp_last_error(h)[0] != 0, "refusal without a message"); ++refused; continue; }
      const size_t bytes = cvvdp_workspace_bytes(h);
      REQUIRE(bytes > 0 && bytes % sizeof(float) == 0, "workspace bytes %zu", bytes);
      REQUIRE(cvvdp_bind_workspace(h, fake_ws, bytes - 4) != CVVDP_OK, "a short workspace was accepted");
      REQUIRE(cvvdp_bind_workspace(h, fake_ws, bytes) == CVVDP_OK, "bind failed: %s", cvvdp_last_error(h));
      g_ws = fake_ws; g_ws_floats = bytes / sizeof(float);
      if (ri(0, 4) == 0) (void)cvvdp_profile_enable(h, 1);
      const int64_t st[5] = {(int64_t)3 * c.n_frames * c.height * c.width, (int64_t)c.n_frames * c.height * c.width, (int64_t)c.height * c.width, c.width, 1};
      int rc = CVVDP_OK;
      if (!c.is_video) {
        rc = cvvdp_put_image(h, fake_src, fake_src, ri(0, 3), st, st, nullptr);
        if (rc == CVVDP_OK) rc = cvvdp_process_image(h, nullptr);
        if (rc == CVVDP_OK && c.heatmap) rc = cvvdp_get_heatmap(h, 1, fake_out, nullptr);
        if (rc == CVVDP_OK && c.feature_size > 0) for (int b = 0; b < c.n_levels && rc == CVVDP_OK; ++b) rc = cvvdp_get_features(h, b, 1, static_cast<float*>(fake_out), nullptr);
      } else {
        const int M = c.filter_len - 1;
        for (int done = 0; done < c.n_frames && rc == CVVDP_OK;) {
          const int n = std::min(c.block_frames, c.n_frames - done);
          std::vector<int32_t> hs(std::max(M, 1));
          for (int i = 0; i < M; ++i) hs[i] = (done == 0 || c.raw_halo) ? ri(0, n - 1) : -1 - i;     // padding / real halo frames, or the DKL tail
          const int kind = ri(0, 5);
          if (kind == 0 && c.channels == 3 && c.batch == 1 && c.width % 2 == 0 && c.height % 2 == 0) {
            cvvdp_yuv_format f{}; f.chroma = 420; f.bit_depth = ri(0, 1) ? 8 : 10; f.matrix = ri(0, 1) ? 709 : 2020;
            f.frame_stride_test = f.frame_stride_ref = (int64_t)c.width * c.height * 3 / 2;
            rc = cvvdp_process_block_yuv(h, fake_src, fake_src, &f, 0, hs.data(), n, done, nullptr);
          } else if (kind == 1) {
            rc = cvvdp_process_block_filtered(h, fake_src, fake_src, st, st, n, done, nullptr);
          } else {
            rc = cvvdp_process_block(h, fake_src, fake_src, ri(0, 4), st, st, 0, hs.data(), n, done, nullptr);
          }
          if (rc == CVVDP_OK && c.defer_bands) {
            // the block is filtered: bands and heat maps piece by piece, then what must be refused
            for (int p0 = 0; p0 < n && rc == CVVDP_OK; p0 += c.score_frames) {
              const int m = std::min(c.score_frames, n - p0);
              rc = cvvdp_score_frames(h, p0, m, nullptr);
              if (rc == CVVDP_OK && c.heatmap) rc = ri(0, 1) ? cvvdp_get_heatmap(h, m, fake_out, nullptr) : cvvdp_get_heatmap_rgb8(h, m, fake_out, nullptr);
            }
            if (rc == CVVDP_OK) {
              REQUIRE(cvvdp_score_frames(h, n, 1, nullptr) != CVVDP_OK, "a piece beyond the filtered block was accepted");
              REQUIRE(cvvdp_score_frames(h, 0, c.score_frames + 1, nullptr) != CVVDP_OK, "a piece longer than score_frames was accepted");
            }
            done += n;
            continue;
          }
          if (rc == CVVDP_OK) REQUIRE(cvvdp_score_frames(h, 0, 1, nullptr) != CVVDP_OK, "cvvdp_score_frames on a clip without defer_bands");
          if (rc == CVVDP_OK && c.heatmap) rc = ri(0, 1) ? cvvdp_get_heatmap(h, n, fake_out, nullptr) : cvvdp_get_heatmap_rgb8(h, n, fake_out, nullptr);
          if (rc == CVVDP_OK && c.feature_size > 0) for (int b = 0; b < c.n_levels && rc == CVVDP_OK; ++b) rc = cvvdp_get_features(h, b, n, static_cast<float*>(fake_out), nullptr);
          done += n;
        }
      }
      if (rc != CVVDP_OK) { REQUIRE(cvvdp_last_error(h)[0] != 0, "failure %d without a message", rc); ++refused; continue; }
      REQUIRE(cvvdp_get_q_per_ch(h, static_cast<float*>(fake_out), nullptr) == CVVDP_OK, "get_q_per_ch: %s", cvvdp_last_error(h));
      REQUIRE(cvvdp_pool_jod(h, static_cast<const float*>(fake_out), c.batch, c.is_video ? 4 : 3, c.n_frames, c.n_levels, static_cast<float*>(fake_out), nullptr) == CVVDP_OK, "pool");
      double ms[CVVDP_PROF_N]; int32_t cnt[CVVDP_PROF_N];
      (void)cvvdp_profile_read(h, ms, cnt);
      (void)cvvdp_profile_enable(h, 0);
      for (int w = 0; w <= CVVDP_BUF_Q; ++w) { void* ptr = nullptr; size_t nf = 0; if (cvvdp_debug_buffer(h, w, 0, &ptr, &nf) == CVVDP_OK) in_ws(ptr, nf, "debug buffer"); }
      ++ran;
    }
    cvvdp_destroy(h);
    REQUIRE(g_live_streams == 0 && g_live_events == 0, "%d streams and %d events alive after destroy", g_live_streams, g_live_events);
  }
  // null / misuse paths return codes, never crash
  REQUIRE(cvvdp_configure(nullptr, nullptr) != CVVDP_OK && cvvdp_workspace_bytes(nullptr) == 0, "null handle accepted");
  cvvdp_destroy(nullptr);
  printf("host_sanitize: %d handles, %ld clips run, %ld refused, %ld launches checked, %ld buffer extents inside their workspace, 0 findings\n",
         n_cases, ran, refused, g_launches, g_checks);
  return ran > n_cases / 2 ? 0 : 2;
}
