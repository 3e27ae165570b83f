// C ABI of the compute core (include/cvvdp_hip.h): handle, workspace planning, launch sequencing.
// No device memory is allocated here; everything lives in the caller's workspace buffer.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

using namespace cvvdp;

namespace {

struct Level {
  int H = 0, W = 0;
  int64_t P = 0;
  size_t g_off = 0, heat_off = 0, dd_off = 0, fd_off = 0, fs_off = 0;  // float offsets into the workspace
  bool split_edge = false;   // k_band4, W % 8 != 0: aligned strips and edge strips as two launches (decided per clip, not per block)
  bool feat4 = false;   // features mode and the level runs on k_band4: column sums (fs_off) instead of per-pixel planes (dd_off, fd_off)
  int f_pieces = 0;
  size_t partial_off = 0;   // this level's partial sums (levels run concurrently: no sharing)
  int n_strip = 1, n_seg = 1, seg_h = 1;
  bool blur = false;
  bool vec4 = false;  // level is handled by k_band4
};

struct ProfEvent { hipEvent_t a, b; int cat; };

}  // namespace

struct cvvdp_handle {
  cvvdp_params p{};
  cvvdp_clip c{};
  bool configured = false;
  int nch = 4, L = 0;
  // items allocated per plane: of level 0 (the temporal stage's block) and of everything behind it (levels 1.., partial sums, heat
  // bands).  The two differ for clips scored in pieces (cvvdp_clip.defer_bands): a long temporal block, short band / heat-map pieces.
  int items_cap = 0, items_cap_s = 0;
  int filtered_frames = 0, filtered_q0 = 0;   // defer_bands: what the last cvvdp_process_block* left in level 0
  std::vector<Level> lv;
  size_t hist_off = 0, hist_shadow_off = 0, partial_off = 0, q_off = 0, hstats_off = 0, hcurve_off = 0;
  size_t ws_floats = 0;
  float* ws = nullptr;
  int last_items = 0, last_item0 = 0;     // the items the band stage ran on last (count; offset inside level 0)
  bool last_range_done = false;           // ... and whether its level-0 band kernel took the context image's range (heat maps)
  float eotf_tab[256];          // per-code display model of 8-bit sources (eotf_table), made once in cvvdp_create
  bool eotf_tab_ok = false;
  bool prof = false;
  int fuse_levels = 0;          // leading pyramid levels whose band kernel computes the next level itself (k_band4f): no reduce pass for them
  std::vector<ProfEvent> events;
  size_t events_used = 0;
  // two-stage software pipeline over blocks (video, no heat map): FIR + reduce of block k+1 run on the
  // caller's stream while the band kernels of block k run on an internal stream; two pyramid sets.
  bool pipeline = false;
  size_t pyr_set_floats = 0;
  int cur_set = 0;
  hipStream_t band_stream = nullptr;
  // the small pyramid levels (2 and up: each less than a GPU-full of workgroups) run beside level 0 / 1 on two side streams
  hipStream_t aux_stream[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  // frames whose width is not a multiple of 8: the one or two strips at the right image edge (RAGGED instantiation of k_band4) run
  // on their own stream beside the aligned strips of the same level
  // edge streams: the border strips of a level run beside its other strips.  One per stream a level can be enqueued on (0: the caller's,
  // 1 / 2: the side streams of the band stage), each with its own fork / join events -- two levels in flight on different streams never
  // share an event or serialise their border launches on one stream
  hipStream_t edge_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_edge_fork[3] = {nullptr, nullptr, nullptr}, ev_edge_join[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_reduce[2] = {nullptr, nullptr}, ev_band[2] = {nullptr, nullptr};
  bool band_pending[2] = {false, false};
};

namespace {

thread_local std::string t_err;

int fail(cvvdp_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  (void)h;
  t_err = buf;      // per calling thread: a prefetch thread's cvvdp_unpack_yuv_resized and the main thread's calls share a handle
  return code;
}

size_t align_up(size_t n_floats) { return (n_floats + 63) & ~size_t(63); }  // 256-byte granules

int check_launch(cvvdp_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, CVVDP_E_HIP, "%s: %s", what, hipGetErrorString(e));
  return CVVDP_OK;
}

// heat maps: do the finishing kernels add the expanded level-1 reconstruction themselves (4 pixels per thread, heatmap.hip heat_recon4)?
bool heat_l0_fused(const cvvdp_handle* h) { return h->L >= 2 && h->lv[0].W % 4 == 0 && h->lv[1].W >= 2; }
float* gbase(const cvvdp_handle* h, int level, int set) { return h->ws + h->lv[level].g_off + (size_t)set * h->pyr_set_floats; }
int level_cap(const cvvdp_handle* h, int level) { return level == 0 ? h->items_cap : h->items_cap_s; }

int ensure_pipeline_objects(cvvdp_handle* h) {
  if (!h->pipeline || h->band_stream) return CVVDP_OK;
  if (hipStreamCreateWithFlags(&h->band_stream, hipStreamNonBlocking) != hipSuccess) return fail(h, CVVDP_E_HIP, "cannot create the band stream");
  for (int i = 0; i < 2; ++i) {
    if (hipEventCreateWithFlags(&h->ev_reduce[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_band[i], hipEventDisableTiming) != hipSuccess)
      return fail(h, CVVDP_E_HIP, "cannot create pipeline events");
  }
  return CVVDP_OK;
}

// make the caller's stream wait for every band stage still in flight (no host synchronisation)
void join_pipeline(cvvdp_handle* h, hipStream_t s) {
  for (int i = 0; i < 2; ++i) {
    if (h->band_pending[i]) {
      (void)hipStreamWaitEvent(s, h->ev_band[i], 0);
      h->band_pending[i] = false;
    }
  }
}

struct ProfScope {
  cvvdp_handle* h; hipStream_t s; ProfEvent* ev = nullptr;
  ProfScope(cvvdp_handle* h_, int cat, hipStream_t s_) : h(h_), s(s_) {
    if (!h->prof) return;
    if (h->events_used == h->events.size()) {
      ProfEvent e{};
      if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
      h->events.push_back(e);
    }
    ev = &h->events[h->events_used++];
    ev->cat = cat;
    (void)hipEventRecord(ev->a, s);
  }
  ~ProfScope() { if (ev) (void)hipEventRecord(ev->b, s); }
};

// The outer taps of the pyramid's 5-tap kernel (lpyr_dec.py:179: `torch.tensor([0.25 - a/2.0, 0.25, a, 0.25, 0.25 - a/2.0], dtype=float32)`, a = 0.4):
// the reference evaluates 0.25 - 0.4/2.0 in DOUBLE and rounds the result to fp32 = 0.0500000007.  Rounds 1-5 wrote `0.25f - 0.4f / 2.0f`, which
// is 0.0499999970 (the fp32 0.4 is 0.4000000060): two ulps of the tap.  Harmless for every band with structure in it -- and 1-3 x the Q_per_ch
// tolerance at the two coarsest Laplacian bands of smooth clips: the expand's even samples then sum taps to 1 - 3e-8 and its odd samples to 1, and
// the Laplacian there is a 1e-4 relative difference of its operands (the "thin class" of VERDICT r3-r5; found in round 6 by the bit-for-bit test
// of the small-level reduce, tests/test_gpu_parity.py::test_small_levels_reduce_like_the_reference_bit_for_bit).
constexpr float kGaussK0 = (float)(0.25 - 0.4 / 2.0);

void fill_csf(const cvvdp_handle* h, int level, float* lut) {
  std::memcpy(lut, h->c.csf_rows + (size_t)level * 4 * CVVDP_CSF_NODES, sizeof(float) * 4 * CVVDP_CSF_NODES);
}

void heat_weights(const cvvdp_handle* h, bool baseband, float* w) {
  // cvvdp_metric.py:728-731
  const float t_int = h->c.is_video ? 1.0f : h->p.image_int;
  for (int c = 0; c < 4; ++c) {
    w[c] = h->p.ch_w[c] * t_int;
    if (baseband) w[c] = w[c] * h->p.baseband_weight[c];
  }
}

// levels [l_begin, L-1) + baseband + heat-map reconstruction; fused: only levels [l_begin, l_end) on k_band4f (each writes the next level)
// item0: the first item of level 0 the call works on (clips scored in pieces; everything behind level 0 starts at item 0)
int run_bands(cvvdp_handle* h, int n_frames, int q_frame_offset, int set, hipStream_t s, int l_begin = 0, int l_end = -1, bool fused = false, int item0 = 0);

int run_pyramid_and_bands(cvvdp_handle* h, int n_frames, int q_frame_offset, hipStream_t s, int item0 = 0) {
  const int B = h->c.batch, items = n_frames * B, L = h->L, nch = h->nch;
  const float K[5] = {kGaussK0, 0.25f, 0.4f, 0.25f, kGaussK0};  // lpyr_dec.py:179
  const int set = h->pipeline ? h->cur_set : 0;
  static const bool fuse2 = dev_knob("CVVDP_REDUCE2", 1) != 0;
  // The first fuse_levels levels need no reduce pass: their band kernels compute the next level from the rows they stream
  // (band4f.hip) -- first those, in order (level l+1 is level l's by-product), then the reduce chain from the first level that
  // is left, then the remaining bands.
  const int F = h->pipeline ? 0 : h->fuse_levels;
  if (F > 0) {
    if (int e = run_bands(h, n_frames, q_frame_offset, 0, s, 0, F, true, item0)) return e;
  }
  for (int l = F; l + 1 < L; ++l) {
    ProfScope ps(h, CVVDP_PROF_REDUCE, s);
    if (fuse2 && l + 2 < L && reduce2_supported(h->lv[l].H, h->lv[l].W)) {   // two levels per pass
      Reduce2Args r{};
      r.in = gbase(h, l, set) + (l == 0 ? (size_t)item0 * h->lv[0].P : 0); r.out1 = gbase(h, l + 1, set); r.out2 = gbase(h, l + 2, set);
      r.H = h->lv[l].H; r.W = h->lv[l].W; r.H1 = h->lv[l + 1].H; r.W1 = h->lv[l + 1].W; r.H2 = h->lv[l + 2].H; r.W2 = h->lv[l + 2].W;
      r.n_img = items; r.img_cap = level_cap(h, l); r.img_cap_out = h->items_cap_s; r.n_planes = 2 * nch;
      for (int i = 0; i < 5; ++i) r.k[i] = K[i];
      launch_reduce2(r, s);
      ++l;
      continue;
    }
    ReduceArgs r{};
    r.in = gbase(h, l, set) + (l == 0 ? (size_t)item0 * h->lv[0].P : 0);
    r.out = gbase(h, l + 1, set);
    r.H = h->lv[l].H; r.W = h->lv[l].W; r.Ho = h->lv[l + 1].H; r.Wo = h->lv[l + 1].W;
    r.n_img = items; r.img_cap = level_cap(h, l); r.img_cap_out = h->items_cap_s; r.n_planes = 2 * nch;
    for (int i = 0; i < 5; ++i) r.k[i] = K[i];
    launch_reduce(r, s);
  }
  if (int e = check_launch(h, "reduce")) return e;
  if (!h->pipeline) return run_bands(h, n_frames, q_frame_offset, 0, s, F, -1, false, item0);
  // hand the pyramid set to the band stage on the internal stream; the caller's stream is free to start
  // the next block's FIR + reduce into the other set
  if (int e = ensure_pipeline_objects(h)) return e;
  (void)hipEventRecord(h->ev_reduce[set], s);
  (void)hipStreamWaitEvent(h->band_stream, h->ev_reduce[set], 0);
  if (int e = run_bands(h, n_frames, q_frame_offset, set, h->band_stream)) return e;
  (void)hipEventRecord(h->ev_band[set], h->band_stream);
  h->band_pending[set] = true;
  h->cur_set ^= 1;
  return CVVDP_OK;
}

int run_bands(cvvdp_handle* h, int n_frames, int q_frame_offset, int set, hipStream_t s, int l_begin, int l_end, bool fused, int item0) {
  const int B = h->c.batch, items = n_frames * B, L = h->L, nch = h->nch;
  if (l_begin == 0) h->last_range_done = false;
  if (l_end < 0) l_end = L - 1;
  const float K[5] = {kGaussK0, 0.25f, 0.4f, 0.25f, kGaussK0};  // lpyr_dec.py:179
  const bool heat = h->c.heatmap != CVVDP_HEATMAP_NONE;
  // Images and blocks of small frames: not even level 0 is a GPU-full of workgroups (768 resident; a 4K image has 752, then 192,
  // 48, ..) and the chain of per-level launches is latency-bound.  The levels are independent once the pyramid exists, so there the
  // levels from 2 on run on two side streams beside level 0 / 1 (fork: after the reduce passes on s; join: before anything reads Q or
  // the heat bands, and before the next block overwrites the pyramid).  Same kernels, same arguments, same results: 4K image
  // 0.785 -> 0.61 ms, 1080p image 0.64 -> 0.50 ms, 854x480 x 64 frames 1.83 -> 1.69 ms.  Blocks whose level 0 is several GPU-fulls
  // keep the single stream: measured on 4K x 64, the overlap gains 0.1 ms of 19 and only makes the per-kernel timings overlap.
  static const bool fork_env = dev_knob("CVVDP_BAND_STREAMS", 1) != 0;
  static const int fork_max = dev_knob("CVVDP_BAND_STREAMS_MAX", 1024);
  // Round 5: the same for the levels BEHIND the fused ones of a large block (4K x 64: levels 3.., 1080p: levels 2..): each is at most one
  // round of workgroups, they depend on the reduce chain only, and one after the other their launches leave the GPU half empty between
  // them -- so the rule looks at the first level this call runs, not at level 0.
  static const int fork_tail = dev_knob("CVVDP_BAND_STREAMS_TAIL", 1);
  const int l_size = (fork_tail && !fused) ? std::min(l_begin, L - 1) : 0;
  bool fork = !fused && fork_env && L - l_size >= 4 && (int64_t)items * h->lv[l_size].n_strip * h->lv[l_size].n_seg <= fork_max;
  if (fork && !h->aux_stream[0]) {
    bool ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i)
      ok = hipStreamCreateWithFlags(&h->aux_stream[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) return fail(h, CVVDP_E_HIP, "cannot create the side streams of the band stage");
  }
  if (fork) {
    (void)hipEventRecord(h->ev_fork, s);
    for (int i = 0; i < 2; ++i) (void)hipStreamWaitEvent(h->aux_stream[i], h->ev_fork, 0);
  }
  hipStream_t const s_main = s;
  for (int l = l_begin; l < l_end; ++l) {
    const Level& lv = h->lv[l];
    hipStream_t s = (fork && l >= l_size + 2) ? h->aux_stream[l & 1] : s_main;      // (the first two levels of the call stay on the caller's stream)
    ProfScope ps(h, l == 0 ? CVVDP_PROF_BAND0 : CVVDP_PROF_BAND_REST, s);
    BandArgs a{};
    a.g = gbase(h, l, set) + (l == 0 ? (size_t)item0 * h->lv[0].P : 0);
    a.gc = gbase(h, l + 1, set);
    a.H = lv.H; a.W = lv.W; a.Hc = h->lv[l + 1].H; a.Wc = h->lv[l + 1].W;
    a.items = items; a.items_cap = level_cap(h, l); a.items_cap_c = h->items_cap_s; a.nch = nch;
    a.seg_h = lv.seg_h; a.n_seg = lv.n_seg; a.n_strip = lv.n_strip;
    a.band_mul = (l == 0) ? 1.0f : 2.0f;  // lpyr_dec.py:60-66 (baseband handled separately)
    fill_csf(h, l, a.lut);
    a.logL_first = h->p.csf_logL_first; a.logL_last = h->p.csf_logL_last;
    a.sens_mul = h->p.sens_mul;
    for (int c = 0; c < 4; ++c) {
      a.ch_gain[c] = h->p.ch_gain[c];
      a.q[c] = h->p.mask_q[c];
      a.eps_q[c] = std::pow(kEps, h->p.mask_q[c]);
    }
    a.mask_c10 = h->p.mask_c10; a.mask_p = h->p.mask_p; a.eps_p = std::pow(kEps, h->p.mask_p);
    for (int i = 0; i < 16; ++i) a.xw[i] = h->p.xcm[i];
    a.dmax = h->p.d_max10;
    a.inv_dmax = (float)(1.0 / (double)h->p.d_max10);
    for (int c = 0; c < 4; ++c) {
      double m1 = 1.0;
      for (int k = 0; k < nch; ++k) m1 -= (double)h->p.xcm[k * 4 + c] * std::pow((double)kEps, (double)h->p.mask_q[k]);
      a.m1[c] = (float)m1;
    }
    for (int i = 0; i < 13; ++i) { a.blur[i] = h->p.blur_taps[i]; a.blur_h[i] = h->p.blur_taps[i] * h->p.mask_c10; }
    {
      const double ind_scale = (double)(CVVDP_CSF_NODES - 1) / ((double)h->p.csf_logL_last - (double)h->p.csf_logL_first);
      a.ind_k1 = (float)(0.30102999566398120 * ind_scale); a.ind_k0 = (float)((double)h->p.csf_logL_first * ind_scale);
    }
    a.kx[0] = K[0] * 2.0f; a.kx[1] = K[2] * 2.0f; a.kx[2] = K[1] * 2.0f;
    a.partial = h->ws + lv.partial_off;
    a.dchr = heat ? h->ws + lv.heat_off : nullptr;
    heat_weights(h, false, a.hw);
    a.beta_tch = h->p.beta_tch; a.eps_btch = std::pow(kEps, h->p.beta_tch); a.eps_inv_btch = std::pow(kEps, 1.0f / h->p.beta_tch);
    a.ddump = (h->c.debug_dump || (h->c.feature_size > 0 && !lv.feat4)) ? h->ws + lv.dd_off : nullptr;
    a.fdump = (h->c.feature_size > 0 && !lv.feat4) ? h->ws + lv.fd_off : nullptr;
    a.fsum = lv.feat4 ? h->ws + lv.fs_off : nullptr; a.fs = h->c.feature_size; a.f_pieces = lv.f_pieces;
    // Levels whose strips at the image border run as their own launch (k_band4 on ragged frames: split_edge; k_band4f: always): a
    // small launch of a slower instantiation.  In the same stream it would cost a whole extra round of the row march; on its own
    // stream -- ordered after everything that precedes this level on s, joined before the finish kernel reads the partial sums
    // (and before the next level reads what this one wrote) -- it fills the GPU together with the other strips.  The profiling
    // events of the level sit on s before the fork and after the join: they time the pair of launches.
    hipStream_t s_edge = s;
    // colour-mapped heat maps: level 0's fused kernels stream the context plane (test Y-sustained) anyway and take its range on the way
    // (atomic min / max into hstats).  The words are initialised HERE, on s BEFORE the fork event: the border strips' kernel runs on the
    // edge stream and must be ordered after the initialisation (enqueued after the fork it could wipe their contribution).
    if (fused && l == 0 && heat && h->c.heatmap != CVVDP_HEATMAP_RAW) {
      a.hstats = reinterpret_cast<uint32_t*>(h->ws + h->hstats_off);
      launch_heat_init(a.hstats, items, s);
      h->last_range_done = true;
    }
    const int es = s == s_main ? 0 : 1 + (l & 1);      // which edge stream goes with the stream this level is on
    if (fused || (lv.vec4 && lv.split_edge)) {
      if (!h->edge_stream[es]) {
        if (hipStreamCreateWithFlags(&h->edge_stream[es], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_edge_fork[es], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_edge_join[es], hipEventDisableTiming) != hipSuccess)
          return fail(h, CVVDP_E_HIP, "cannot create the edge-strip stream of the band stage");
      }
      s_edge = h->edge_stream[es];
      (void)hipEventRecord(h->ev_edge_fork[es], s);
      (void)hipStreamWaitEvent(s_edge, h->ev_edge_fork[es], 0);
    }
    if (fused) {
      a.g1_out = gbase(h, l + 1, set);
      for (int i = 0; i < 5; ++i) a.rk[i] = K[i];
      a.one_wave_layout = h->c.band_layout == 1;
      launch_band4f(a, s, s_edge);
    } else if (lv.vec4) {
      launch_band4(a, lv.split_edge, s, s_edge);
    } else {
      launch_band(a, lv.blur, s);
    }
    if (s_edge != s) {
      (void)hipEventRecord(h->ev_edge_join[es], s_edge);
      (void)hipStreamWaitEvent(s, h->ev_edge_join[es], 0);
    }
    FinalizeArgs f{};
    f.partial = a.partial; f.items = items; f.nblk = lv.n_strip * lv.n_seg; f.nch = nch; f.P = (int)lv.P;
    f.q_out = h->ws + h->q_off; f.q_frames = h->c.n_frames; f.q_levels = L; f.q_frame_offset = q_frame_offset;
    f.level = l; f.batch = B;
    f.sub_per_term = lv.vec4 ? (double)kEps * (double)kEps : 0.0;
    launch_finalize(f, s);
  }
  if (int e = check_launch(h, "band")) return e;
  if (fused) return CVVDP_OK;       // (the caller goes on with the reduce chain and the remaining levels)
  {
    const Level& lv = h->lv[L - 1];
    hipStream_t s = fork ? h->aux_stream[(L - 1) & 1] : s_main;
    ProfScope ps(h, CVVDP_PROF_BAND_REST, s);
    BaseArgs b{};
    b.g = gbase(h, L - 1, set) + (L == 1 ? (size_t)item0 * h->lv[0].P : 0); b.H = lv.H; b.W = lv.W; b.items = items; b.items_cap = level_cap(h, L - 1); b.nch = nch;
    fill_csf(h, L - 1, b.lut);
    b.logL_first = h->p.csf_logL_first; b.logL_last = h->p.csf_logL_last; b.sens_mul = h->p.sens_mul;
    b.q_out = h->ws + h->q_off; b.q_frames = h->c.n_frames; b.q_levels = L; b.q_frame_offset = q_frame_offset;
    b.level = L - 1; b.batch = B;
    b.dchr = heat ? h->ws + lv.heat_off : nullptr;
    heat_weights(h, true, b.hw);
    b.beta_tch = h->p.beta_tch; b.eps_btch = std::pow(kEps, h->p.beta_tch); b.eps_inv_btch = std::pow(kEps, 1.0f / h->p.beta_tch);
    b.ddump = (h->c.debug_dump || h->c.feature_size > 0) ? h->ws + lv.dd_off : nullptr;
    b.fdump = h->c.feature_size > 0 ? h->ws + lv.fd_off : nullptr;
    launch_baseband(b, s);
  }
  if (int e = check_launch(h, "baseband")) return e;
  if (fork) {
    for (int i = 0; i < 2; ++i) {
      (void)hipEventRecord(h->ev_join[i], h->aux_stream[i]);
      (void)hipStreamWaitEvent(s_main, h->ev_join[i], 0);
    }
  }
  if (heat) {  // lpyr_dec_2.reconstruct, lpyr_dec.py:328-335: coarse to fine, in place
    ProfScope ps(h, CVVDP_PROF_HEATMAP, s);
    // the last step (level 1 -> level 0) is done by the heat-map kernels themselves where they can (get_heatmap_impl): 8 B/pixel less traffic
    for (int l = L - 2; l >= (heat_l0_fused(h) ? 1 : 0); --l) {
      ExpandAddArgs e{};
      e.fine = h->ws + h->lv[l].heat_off; e.coarse = h->ws + h->lv[l + 1].heat_off;
      e.H = h->lv[l].H; e.W = h->lv[l].W; e.Hc = h->lv[l + 1].H; e.Wc = h->lv[l + 1].W; e.n_img = items;
      e.kx[0] = K[0] * 2.0f; e.kx[1] = K[2] * 2.0f; e.kx[2] = K[1] * 2.0f;
      e.per_thread_layout = h->c.band_layout == 1;
      launch_expand_add(e, s);
    }
    if (int e = check_launch(h, "heat reconstruct")) return e;
  }
  h->last_items = items;
  h->last_item0 = item0;
  return CVVDP_OK;
}

}  // namespace

extern "C" {

int cvvdp_abi_version(void) { return CVVDP_ABI_VERSION; }

void cvvdp_struct_sizes(int32_t* params_bytes, int32_t* clip_bytes) {
  if (params_bytes) *params_bytes = (int32_t)sizeof(cvvdp_params);
  if (clip_bytes) *clip_bytes = (int32_t)sizeof(cvvdp_clip);
}

static bool eotf_table(const cvvdp_params& p, float scale, float lin_lo, float (&tab)[256]);
int cvvdp_create(const cvvdp_params* params, cvvdp_handle** out) {
  if (!params || !out) return CVVDP_E_ARG;
  cvvdp_handle* h = new (std::nothrow) cvvdp_handle();
  if (!h) return CVVDP_E_ARG;
  h->p = *params;
  h->eotf_tab_ok = eotf_table(h->p, (float)((double)h->p.Y_peak - (double)h->p.Y_black), std::max(0.005f, h->p.Y_black), h->eotf_tab);
  *out = h;
  return CVVDP_OK;
}

void cvvdp_destroy(cvvdp_handle* h) {
  if (!h) return;
  if (h->band_stream) {
    (void)hipStreamSynchronize(h->band_stream);
    (void)hipStreamDestroy(h->band_stream);
    for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(h->ev_reduce[i]); (void)hipEventDestroy(h->ev_band[i]); }
  }
  for (auto& e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (int i = 0; i < 2; ++i) {
    if (h->aux_stream[i]) { (void)hipStreamSynchronize(h->aux_stream[i]); (void)hipStreamDestroy(h->aux_stream[i]); }
    if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (int i = 0; i < 3; ++i) {
    if (h->edge_stream[i]) { (void)hipStreamSynchronize(h->edge_stream[i]); (void)hipStreamDestroy(h->edge_stream[i]); }
    if (h->ev_edge_fork[i]) (void)hipEventDestroy(h->ev_edge_fork[i]);
    if (h->ev_edge_join[i]) (void)hipEventDestroy(h->ev_edge_join[i]);
  }
  delete h;
}

const char* cvvdp_last_error(const cvvdp_handle* h) { return h ? t_err.c_str() : "null handle"; }
int cvvdp_build_flags(void) {
  int f = tu_flags_band4() | tu_flags_band4f() | tu_flags_band4s();     // each band translation unit reports how IT was compiled
#ifdef CVVDP_DEV_KNOBS
  f |= CVVDP_BUILD_DEV_KNOBS;
#endif
  return f;
}

#define CVVDP_STR2(x) #x
#define CVVDP_STR(x) CVVDP_STR2(x)
const char* cvvdp_build_info(void) {
#ifdef __clang_version__
#define CVVDP_COMPILER "clang " __clang_version__
#else
#define CVVDP_COMPILER "host compiler " __VERSION__      /* (tests/test_host_sanitize.py builds this file with g++) */
#endif
  return CVVDP_COMPILER "; HIP " CVVDP_STR(HIP_VERSION_MAJOR) "." CVVDP_STR(HIP_VERSION_MINOR) "." CVVDP_STR(HIP_VERSION_PATCH) "; gfx950";
}
int cvvdp_compiled_hip_version(void) { return HIP_VERSION; }
int cvvdp_runtime_hip_version(void) {
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) return 0;
  return v;
}

int cvvdp_configure(cvvdp_handle* h, const cvvdp_clip* clip) {
  if (!h || !clip) return CVVDP_E_ARG;
  if (h->band_stream && (h->band_pending[0] || h->band_pending[1])) {   // re-configuring with band work in flight
    (void)hipStreamSynchronize(h->band_stream);
    h->band_pending[0] = h->band_pending[1] = false;
  }
  const cvvdp_clip& c = *clip;
  if (c.batch < 1 || c.height < 2 || c.width < 2 || c.n_frames < 1) return fail(h, CVVDP_E_ARG, "bad clip geometry");
  if (c.channels != 1 && c.channels != 3) return fail(h, CVVDP_E_ARG, "channels must be 1 or 3");
  if (c.n_levels < 1 || c.n_levels > CVVDP_MAX_LEVELS) return fail(h, CVVDP_E_ARG, "n_levels out of range");
  if (c.is_video) {
    if (c.filter_len < 1 || c.filter_len > CVVDP_MAX_FILTER_LEN) return fail(h, CVVDP_E_UNSUPPORTED, "filter_len %d unsupported (max %d)", c.filter_len, CVVDP_MAX_FILTER_LEN);
    if (c.block_frames < 1 || c.filter_len - 1 + c.block_frames > CVVDP_MAX_WINDOW) return fail(h, CVVDP_E_ARG, "block_frames out of range");
  }
  if (c.heatmap != CVVDP_HEATMAP_NONE && c.batch != 1) return fail(h, CVVDP_E_UNSUPPORTED, "heat maps need batch == 1");
  if (c.feature_size < 0) return fail(h, CVVDP_E_ARG, "feature_size must be >= 0");
  // features mode runs the FEAT instantiations of the band kernels, which write neither heat-map bands nor the per-pixel D dump
  if (c.feature_size > 0 && (c.heatmap != CVVDP_HEATMAP_NONE || c.debug_dump))
    return fail(h, CVVDP_E_UNSUPPORTED, "feature_size > 0 cannot be combined with a heat map or debug_dump");
  if (c.fuse_mode < 0 || c.fuse_mode > 2) return fail(h, CVVDP_E_ARG, "fuse_mode must be 0, 1 or 2");
  if (c.band_layout < 0 || c.band_layout > 1) return fail(h, CVVDP_E_ARG, "band_layout must be 0 or 1");
  if (c.defer_bands < 0 || c.defer_bands > 1) return fail(h, CVVDP_E_ARG, "defer_bands must be 0 or 1");
  if (c.defer_bands) {
    if (!c.is_video) return fail(h, CVVDP_E_ARG, "defer_bands is for video clips");
    if (c.score_frames < 1 || c.score_frames > c.block_frames) return fail(h, CVVDP_E_ARG, "score_frames must be in 1 .. block_frames");
    // (the per-pixel dump and feature planes of level 0 are laid out with the level's own item count)
    if (c.debug_dump || c.feature_size > 0) return fail(h, CVVDP_E_UNSUPPORTED, "defer_bands cannot be combined with debug_dump or features");
  }
  h->c = c;
  h->nch = c.is_video ? 4 : 3;
  h->L = c.n_levels;
  h->items_cap = (c.is_video ? c.block_frames : 1) * c.batch;
  h->items_cap_s = c.defer_bands ? c.score_frames * c.batch : h->items_cap;
  h->filtered_frames = 0;
  h->lv.assign(h->L, Level());
  int H = c.height, W = c.width;
  const int pad = h->p.blur_radius;
  for (int l = 0; l < h->L; ++l) {
    Level& lv = h->lv[l];
    lv.H = H; lv.W = W; lv.P = (int64_t)H * W;
    lv.blur = pad > 0 && H > pad && W > pad;
    lv.vec4 = lv.blur && W >= 16 && H >= 16;
    lv.feat4 = c.feature_size > 0 && lv.vec4 && l + 1 < h->L;          // (the baseband and the tiny levels keep per-pixel |T'|, |R'|, D planes)
    const int sw = lv.vec4 ? kBand4StripWidth : (lv.blur ? 256 - 2 * pad : 256);
    lv.n_strip = (W + sw - 1) / sw;
    // Row segments.  Every segment recomputes 12 blur-halo rows, so segments should be long; a block marches ~2.6 us
    // per row, so there must still be enough blocks to fill 256 CUs x 3.  Video: sized for the nominal 64-frame block
    // (the split must not depend on the actual block size, or per-frame sums would round differently per block size):
    // at most 384 rows, and no more segments than needed for ~768 blocks (4K: 360 rows at levels 0 and 1 = 3 % halo,
    // 180 at level 2).  Images have one item per batch entry and launch: as many segments as it takes, down to 16 rows.
    {
      // video: the nominal block is 64 frames, or the whole clip if that is shorter (total_frames is the same for every
      // block and shard of a clip); an image launch holds exactly `batch` items
      const int clip_frames = c.total_frames > 0 ? c.total_frames : 64;
      static const int target_env = dev_knob("CVVDP_SEG_TARGET", 0);
      static const int rows_env = dev_knob("CVVDP_SEG_ROWS", 0);
      // heat-map clips run their band stage on 16 frames at a time (16-frame blocks, or 16-frame pieces of a long temporal block --
      // whatever score_frames says, so that results do not depend on it): sized for 64, their small levels left most of the GPU idle
      // (8K, level 3: 72 workgroups per launch)
      const int nominal = c.is_video ? std::min(c.heatmap != CVVDP_HEATMAP_NONE ? 16 : 64, clip_frames) : 1, target = target_env > 0 ? target_env : 768;
      const int max_rows = rows_env > 0 ? rows_env : (c.is_video ? 384 : 256);
      const int per_seg = lv.n_strip * nominal * c.batch;
      const int want = (target + per_seg - 1) / per_seg;
      const int lo = (H + max_rows - 1) / max_rows, hi = std::max(lo, (H + 15) / 16);
      lv.n_seg = std::min(std::max(want, lo), hi);
      lv.seg_h = (H + lv.n_seg - 1) / lv.n_seg;
      lv.seg_h += lv.seg_h & 1;               // even: k_band4 unrolls its row loop over even/odd row pairs
      lv.n_seg = (H + lv.seg_h - 1) / lv.seg_h;
      // W % 8 != 0: only the one or two strips at the right image edge need the RAGGED instantiation of k_band4 (spills in its row
      // loop).  Splitting them off pays when the aligned rest is at least two GPU-fulls of workgroups on its own; a launch of
      // one round takes one block's march however it is split (measured: 1366x768 x 64, 768 workgroups, 3.23 -> 3.71 ms with
      // the split -- two launches and two event hops per level for nothing).  Decided from the nominal block, like the
      // segments: the same kernels score a frame whatever the block size, so results stay bit-identical.
      const int n_edge = lv.vec4 ? band4_edge_strips(W, lv.n_strip) : 0;
      lv.split_edge = n_edge > 0 && n_edge < lv.n_strip && (int64_t)(lv.n_strip - n_edge) * lv.n_seg * nominal * c.batch >= 1536;
    }
    if (l + 1 < h->L && (H < 2 || W < 2)) return fail(h, CVVDP_E_ARG, "pyramid too deep for %dx%d", c.width, c.height);
    H = (H + 1) / 2; W = (W + 1) / 2;
  }
  if (pad != 0 && pad != 6) return fail(h, CVVDP_E_UNSUPPORTED, "blur radius %d unsupported", pad);
  // Levels whose band kernel computes the next level itself (band4f.hip: the level's planes are then read once, not twice, and
  // the HBM-bound reduce pass of the level disappears: 4K x 64 fp32 18.4 -> 15.7 ms).  The plain scoring path only (no heat map,
  // dump or features), aligned levels, and only clips whose blocks fill the GPU several times over -- images and small frames
  // keep the independent levels that run side by side on three streams.  Decided per clip from the nominal block, like the
  // segments and the edge-strip split: the same kernels score a frame whatever the block size.
  h->fuse_levels = 0;
  {
    const int clip_frames = c.total_frames > 0 ? c.total_frames : 64;
    const int nominal = c.is_video ? std::min(64, clip_frames) : 1;
    static const int fuse_max = dev_knob("CVVDP_FUSE_LEVELS", CVVDP_MAX_LEVELS);
    // (k_band4f / k_band4s are instantiated for 4 channels: plain, with the heat-map band, with the feature statistics; not with the per-pixel dump)
    const bool plain = c.is_video && !c.debug_dump;
    const bool big = (int64_t)nominal * c.batch * h->lv[0].n_strip * h->lv[0].n_seg > 1024;
    // ... and level by level only while the level itself is large (>= 16 M pixels in the nominal block): on small levels the fused
    // kernel's longer prologue and two-block occupancy cost more than the small reduce pass they replace (sweep over the number
    // of fused levels, profiles/r03_dev_notes.txt: best at 3 levels for 4K x 64, 2 for 1080p x 64)
    if (plain && c.fuse_mode != 2 && (big || c.fuse_mode == 1))
      while (h->fuse_levels < fuse_max && h->fuse_levels + 1 < h->L && h->lv[h->fuse_levels].vec4 &&
             band4f_supported(h->lv[h->fuse_levels].H, h->lv[h->fuse_levels].W) &&
             (c.fuse_mode == 1 || (int64_t)h->lv[h->fuse_levels].P * nominal * c.batch >= (int64_t)1 << 24))
        ++h->fuse_levels;
  }
  // Row segments of the fused levels, round 5.  k_band4s runs 512-thread blocks, two per CU: 512 resident workgroups, and a level takes
  // (rounds of 512) x (one block's march).  The 768-block target above is k_band4's (three 256-thread blocks per CU); on a fused level it
  // produced 1.5 rounds where one would do -- 4K level 2 and 1080p level 1: 3 segments of 180 rows = 768 workgroups = two rounds of 201
  // row steps, against 2 segments of 270 rows = 512 workgroups = one round of 291.  So: among the segment counts the row limit allows,
  // the one with the fewest row steps in sequence (21 = blur halo + reduce prologue rows per segment); from the nominal block, like
  // everything else that decides a frame's summation order.
  {
    const int clip_frames = c.total_frames > 0 ? c.total_frames : 64;
    const int nominal = c.is_video ? std::min(c.heatmap != CVVDP_HEATMAP_NONE ? 16 : 64, clip_frames) : 1;
    static const int seg_rule = dev_knob("CVVDP_FUSED_SEG_RULE", 1);
    // (not when the dev knob CVVDP_PIPELINE routes every level to k_band4: the rule prices k_band4s's 512 resident workgroups)
    static const bool pipe_knob = dev_knob("CVVDP_PIPELINE", 0) != 0;
    for (int l = 0; seg_rule && !pipe_knob && c.fuse_mode != 1 && l < h->fuse_levels; ++l) {
      Level& lv = h->lv[l];
      const int64_t per_seg = (int64_t)lv.n_strip * nominal * c.batch;
      static const int fused_rows = dev_knob("CVVDP_FUSED_SEG_ROWS", 384);
      const int lo = (lv.H + fused_rows - 1) / fused_rows, hi = std::max(lo, (lv.H + 47) / 48);      // (segments of at least 48 rows: 21 of them are overhead)
      int64_t best_cost = -1; int best_n = lv.n_seg, best_h = lv.seg_h;
      for (int n = lo; n <= hi; ++n) {
        int sh = (lv.H + n - 1) / n; sh += sh & 1;
        const int nn = (lv.H + sh - 1) / sh;
        const int64_t cost = ((nn * per_seg + 511) / 512) * (int64_t)(sh + 21);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_n = nn; best_h = sh; }
      }
      lv.n_seg = best_n; lv.seg_h = best_h;
    }
  }
  // ---- workspace plan (float offsets)
  size_t off = 0;
  const size_t P0 = (size_t)h->lv[0].P;
  h->hist_off = h->hist_shadow_off = off;
  if (c.is_video && c.filter_len > 1) {
    const size_t hist = align_up((size_t)2 * 3 * (fir_kernel_len(c.filter_len) - 1) * c.batch * P0);
    off += hist;
    if (!fir_has_register_window(fir_kernel_len(c.filter_len))) { h->hist_shadow_off = off; off += hist; }  // generic-FL path double-buffers the tail
  }
  {
    // off by default: since the kernels were tuned, overlapping the band stage of block k with FIR + reduce of block
    // k+1 no longer gains anything (4K x 256: 84-87 ms either way) and costs a second pyramid set of workspace
    static const bool pipe_env = dev_knob("CVVDP_PIPELINE", 0) != 0;
    h->pipeline = pipe_env && c.is_video && c.heatmap == CVVDP_HEATMAP_NONE && !c.debug_dump && c.feature_size <= 0 && c.n_frames > c.block_frames && !c.defer_bands;
    const size_t start = off;
    for (int l = 0; l < h->L; ++l) { Level& lv = h->lv[l]; lv.g_off = off; off += align_up((size_t)2 * h->nch * level_cap(h, l) * lv.P); }
    h->pyr_set_floats = off - start;
    if (h->pipeline) off += h->pyr_set_floats;   // second pyramid set
    h->cur_set = 0;
  }
  h->partial_off = off;
  for (auto& lv : h->lv) { lv.partial_off = off; off += align_up((size_t)h->items_cap_s * lv.n_strip * lv.n_seg * 4); }
  h->q_off = off; off += align_up((size_t)c.batch * h->nch * c.n_frames * h->L);
  if (c.heatmap != CVVDP_HEATMAP_NONE) {
    for (auto& lv : h->lv) { lv.heat_off = off; off += align_up((size_t)h->items_cap_s * lv.P); }
    h->hstats_off = off; off += align_up((size_t)h->items_cap_s * kHeatStatsWords);
    h->hcurve_off = off; off += align_up((size_t)h->items_cap_s * kHeatCurveWords);
  }
  for (auto& lv : h->lv) {
    if (c.debug_dump || (c.feature_size > 0 && !lv.feat4)) { lv.dd_off = off; off += align_up((size_t)4 * h->items_cap * lv.P); }
    if (c.feature_size > 0 && !lv.feat4) { lv.fd_off = off; off += align_up((size_t)8 * h->items_cap * lv.P); }
    if (lv.feat4) {   // column sums per piece of a cell row (band4.hip, FEATURES)
      lv.f_pieces = (lv.H + c.feature_size - 1) / c.feature_size + lv.n_seg;
      lv.fs_off = off; off += align_up((size_t)h->items_cap * h->nch * lv.f_pieces * 6 * lv.W);
    }
  }
  h->ws_floats = off;
  h->ws = nullptr;
  h->configured = true;
  h->last_items = 0; h->last_item0 = 0;
  return CVVDP_OK;
}

size_t cvvdp_workspace_bytes(const cvvdp_handle* h) { return (h && h->configured) ? h->ws_floats * sizeof(float) : 0; }
int cvvdp_fused_levels(const cvvdp_handle* h) { return (h && h->configured) ? (h->pipeline ? 0 : h->fuse_levels) : -1; }

int cvvdp_bind_workspace(cvvdp_handle* h, void* dev, size_t bytes) {
  if (!h || !h->configured) return fail(h, CVVDP_E_STATE, "configure first");
  if (!dev || bytes < h->ws_floats * sizeof(float)) return fail(h, CVVDP_E_ARG, "workspace too small: need %zu bytes", h->ws_floats * sizeof(float));
  if (reinterpret_cast<uintptr_t>(dev) % 256) return fail(h, CVVDP_E_ARG, "workspace must be 256-byte aligned");
  h->ws = static_cast<float*>(dev);
  return CVVDP_OK;
}

// Emitted light of one channel for an 8-bit code: the per-channel part of vvdp_display_photo_eotf.forward
// (display_model.py:333-365; srgb2lin :78-80, pq2lin :58-70) on V = code / 255 (video_source.py:320-346), evaluated once per code
// in fp32 with the reference's operation order (separate roundings: every intermediate is a float variable; libm's powf
// stands where torch.pow stands).  HLG mixes the channels (its OOTF gain depends on the pixel's luminance): no table.
static bool eotf_table(const cvvdp_params& p, float scale, float lin_lo, float (&tab)[256]) {
  static const bool off = dev_knob("CVVDP_NO_EOTF_LUT", 0) != 0;
  if (off) return false;
  if (p.eotf != CVVDP_EOTF_SRGB && p.eotf != CVVDP_EOTF_PQ && p.eotf != CVVDP_EOTF_LINEAR && p.eotf != CVVDP_EOTF_GAMMA) return false;
  auto clip = [](float x, float lo, float hi) { return std::min(std::max(x, lo), hi); };
  for (int code = 0; code < 256; ++code) {
    const volatile float V = (float)code / 255.0f;
    volatile float L;
    if (p.eotf == CVVDP_EOTF_SRGB) {
      volatile float lin;
      if (V > 0.04045f) { const volatile float x = (V + 0.055f) / 1.055f; lin = powf(x, 2.4f); }
      else lin = V / 12.92f;
      if (p.exposure != 1.0f) { const volatile float e = lin * p.exposure; lin = clip(e, 0.0f, 1.0f); }
      const volatile float a = scale * lin;
      const volatile float b = a + p.Y_black;
      L = b + p.Y_refl;
    } else if (p.eotf == CVVDP_EOTF_PQ) {
      const float m = (float)(1.0 / 78.843750000000000), n = (float)(1.0 / 0.15930175781250000);
      const float c1 = 0.83593750000000000f, c2 = 18.851562500000000f, c3 = 18.687500000000000f;
      const volatile float t = powf(V, m);
      const volatile float num = std::max(t - c1, 0.0f);
      const volatile float ct = c3 * t;
      const volatile float den = c2 - ct;
      const volatile float r = num / den;
      const volatile float pw = powf(r, n);
      const volatile float lin = 10000.0f * pw;
      const volatile float e = lin * p.exposure;
      const volatile float c = clip(e, 0.005f, p.Y_peak);
      const volatile float b = c + p.Y_black;
      L = b + p.Y_refl;
    } else if (p.eotf == CVVDP_EOTF_LINEAR) {
      const volatile float e = V * p.exposure;
      const volatile float c = clip(e, lin_lo, p.Y_peak);
      L = c + p.Y_refl;
    } else {
      const volatile float g = powf(V, p.gamma);
      const volatile float e = g * p.exposure;
      const volatile float a = scale * clip(e, 0.0f, 1.0f);
      const volatile float b = a + p.Y_black;
      L = b + p.Y_refl;
    }
    tab[code] = L;
  }
  return true;
}

static void fill_display(const cvvdp_handle* h, DisplayArgs& d) {
  d.eotf = h->p.eotf; d.channels = h->c.channels;
  d.Y_peak = h->p.Y_peak; d.Y_black = h->p.Y_black; d.Y_refl = h->p.Y_refl;
  d.exposure = h->p.exposure; d.gamma = h->p.gamma;
  d.scale = (float)((double)h->p.Y_peak - (double)h->p.Y_black);
  d.lin_lo = std::max(0.005f, h->p.Y_black);
  d.hlg_c = (float)(0.5 - 0.17883277 * std::log(4.0 * 0.17883277));
  for (int i = 0; i < 9; ++i) d.m[i] = h->p.rgb2dkl[i];
  d.use_lut = h->eotf_tab_ok ? 1 : 0;
  if (h->eotf_tab_ok) std::memcpy(d.lut, h->eotf_tab, sizeof(d.lut));
}

int cvvdp_put_image(cvvdp_handle* h, const void* t, const void* r, int32_t dtype, const int64_t st[5], const int64_t sr[5], void* stream) {
  if (!h || !h->ws) return fail(h, CVVDP_E_STATE, "no workspace bound");
  if (!t || !r || !st || !sr) return fail(h, CVVDP_E_ARG, "bad frame arguments");
  if (dtype < CVVDP_U8 || dtype > CVVDP_F32_DKL) return fail(h, CVVDP_E_UNSUPPORTED, "dtype %d unsupported", dtype);
  const cvvdp_clip& c = h->c;
  if (c.is_video) return fail(h, CVVDP_E_STATE, "configured for video: frames go through cvvdp_process_block");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PhotoArgs a{};
  a.src[0] = t; a.src[1] = r;
  const int64_t* S[2] = {st, sr};
  for (int k = 0; k < 2; ++k) { a.sb[k] = S[k][0]; a.sc[k] = S[k][1]; a.sf[k] = S[k][2]; a.sh[k] = S[k][3]; a.sw[k] = S[k][4]; }
  a.dtype = dtype; a.H = c.height; a.W = c.width; a.batch = c.batch; a.n_frames = 1;
  fill_display(h, a.dm);
  const int64_t P0 = h->lv[0].P;
  a.dst = h->ws + h->lv[0].g_off;
  a.d_b = P0; a.d_slot = 0; a.d_side = (int64_t)h->items_cap * P0; a.d_ch = 2 * a.d_side;
  a.first_slot = 0; a.n_slots = 1;
  {
    ProfScope ps(h, CVVDP_PROF_PHOTOMETRY, s);
    launch_photometry(a, s);
  }
  return check_launch(h, "photometry");
}

static int process_block_impl(cvvdp_handle* h, const void* t, const void* r, int32_t dtype, const int64_t st[5], const int64_t sr[5],
                              const cvvdp::YuvArgs* yuv, int32_t raw_first, const int32_t* hist_src, int32_t n_frames,
                              int32_t q_frame_offset, void* stream);

int cvvdp_process_block(cvvdp_handle* h, const void* t, const void* r, int32_t dtype, const int64_t st[5], const int64_t sr[5],
                        int32_t raw_first, const int32_t* hist_src, int32_t n_frames, int32_t q_frame_offset, void* stream) {
  if (h && (dtype < CVVDP_U8 || dtype > CVVDP_F32_DKL)) return fail(h, CVVDP_E_UNSUPPORTED, "dtype %d unsupported", dtype);
  return process_block_impl(h, t, r, dtype, st, sr, nullptr, raw_first, hist_src, n_frames, q_frame_offset, stream);
}

// format description -> the constants of the Y'CbCr unpack for W x H frames (shared by the fused temporal path and the resize path)
static int fill_yuv(cvvdp_handle* h, const cvvdp_yuv_format* fmt, int W, int H, cvvdp::YuvArgs& y) {
  if (!fmt) return fail(h, CVVDP_E_ARG, "format missing");
  if (fmt->bit_depth < 8 || fmt->bit_depth > 16) return fail(h, CVVDP_E_UNSUPPORTED, "bit depth %d unsupported", fmt->bit_depth);
  if (fmt->matrix != 709 && fmt->matrix != 2020) return fail(h, CVVDP_E_UNSUPPORTED, "matrix %d unsupported (709 or 2020)", fmt->matrix);
  y = cvvdp::YuvArgs{};
  if (fmt->chroma == 444) { y.Wc = W; y.Hc = H; y.inv_fx = 1.0f; y.inv_fy = 1.0f; }
  else if (fmt->chroma == 422) { y.Wc = W / 2; y.Hc = H; y.inv_fx = 0.5f; y.inv_fy = 1.0f; }
  else if (fmt->chroma == 420) { y.Wc = W / 2; y.Hc = H / 2; y.inv_fx = 0.5f; y.inv_fy = 0.5f; }
  else return fail(h, CVVDP_E_UNSUPPORTED, "chroma subsampling %d unsupported (420, 422, 444)", fmt->chroma);
  if (fmt->chroma != 444 && (W % 2 || (fmt->chroma == 420 && H % 2)))
    return fail(h, CVVDP_E_ARG, "%dx%d cannot be %d subsampled", W, H, fmt->chroma);
  const int64_t frame = (int64_t)W * H + 2 * (int64_t)y.Wc * y.Hc;
  if (fmt->frame_stride_test < frame || fmt->frame_stride_ref < frame) return fail(h, CVVDP_E_ARG, "frame stride shorter than a frame");
  y.u_off = (int64_t)W * H; y.v_off = y.u_off + (int64_t)y.Wc * y.Hc;
  // video_source_yuv.py:198-206: python-double constants, applied to fp32 tensors
  const double sc = (double)(1 << (fmt->bit_depth - 8));
  y.wy = (float)(1.0 / (sc * 219.0)); y.oy = (float)(16.0 / 219.0);
  y.wc = (float)(1.0 / (sc * 224.0)); y.oc = (float)(128.0 / 224.0);
  if (fmt->matrix == 2020) { y.rv = 1.47460f; y.gu = -0.16455f; y.gv = -0.57135f; y.bu = 1.88140f; }   // :151-154
  else { y.rv = 1.402f; y.gu = -0.344136f; y.gv = -0.714136f; y.bu = 1.772f; }                          // :157-160
  return CVVDP_OK;
}

int cvvdp_process_block_yuv(cvvdp_handle* h, const void* t, const void* r, const cvvdp_yuv_format* fmt, int32_t raw_first,
                            const int32_t* hist_src, int32_t n_frames, int32_t q_frame_offset, void* stream) {
  if (!h) return CVVDP_E_STATE;
  const cvvdp_clip& c = h->c;
  if (c.channels != 3 || c.batch != 1) return fail(h, CVVDP_E_ARG, "Y'CbCr input needs a clip configured with 3 channels and batch 1");
  const int W = c.width;
  cvvdp::YuvArgs y;
  if (int rc = fill_yuv(h, fmt, W, c.height, y)) return rc;
  const int64_t st[5] = {0, 0, fmt->frame_stride_test, W, 1}, sr[5] = {0, 0, fmt->frame_stride_ref, W, 1};
  return process_block_impl(h, t, r, fmt->bit_depth == 8 ? CVVDP_YUV8 : CVVDP_YUV16, st, sr, &y, raw_first, hist_src, n_frames,
                            q_frame_offset, stream);
}

int cvvdp_unpack_yuv_resized(cvvdp_handle* h, const void* codes, const cvvdp_yuv_format* fmt, int32_t is_ref, int32_t src_w, int32_t src_h,
                             int32_t n_frames, int32_t dst_w, int32_t dst_h, int32_t mode, float* tmp, float* rgb, void* stream) {
  if (!h) return CVVDP_E_STATE;
  if (!codes || !tmp || !rgb || n_frames < 1 || src_w < 1 || src_h < 1 || dst_w < 1 || dst_h < 1) return fail(h, CVVDP_E_ARG, "bad resize arguments");
  if (mode < CVVDP_RESIZE_NEAREST || mode > CVVDP_RESIZE_AREA) return fail(h, CVVDP_E_ARG, "resize mode %d unknown", mode);
  cvvdp::YuvUnpackArgs u{};
  if (int rc = fill_yuv(h, fmt, src_w, src_h, u.yuv)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  u.src = codes; u.W = src_w; u.H = src_h; u.n_frames = n_frames; u.bits16 = fmt->bit_depth > 8;
  u.frame_stride = is_ref ? fmt->frame_stride_ref : fmt->frame_stride_test;
  u.out = tmp;
  cvvdp::launch_yuv_unpack(u, s);
  cvvdp::ResizeArgs z{};
  z.in = tmp; z.out = rgb; z.n_planes = 3 * n_frames; z.Hs = src_h; z.Ws = src_w; z.Hd = dst_h; z.Wd = dst_w; z.mode = mode;
  z.sy = (float)src_h / (float)dst_h; z.sx = (float)src_w / (float)dst_w;        // area_pixel_compute_scale, align_corners = false
  cvvdp::launch_resize(z, s);
  return check_launch(h, "yuv resize");
}

static int process_block_impl(cvvdp_handle* h, const void* t, const void* r, int32_t dtype, const int64_t st[5], const int64_t sr[5],
                              const cvvdp::YuvArgs* yuv, int32_t raw_first, const int32_t* hist_src, int32_t n_frames,
                              int32_t q_frame_offset, void* stream) {
  if (!h || !h->ws) return fail(h, CVVDP_E_STATE, "no workspace bound");
  const cvvdp_clip& c = h->c;
  if (!c.is_video) return fail(h, CVVDP_E_STATE, "configured for an image");
  if (!t || !r || !st || !sr || raw_first < 0) return fail(h, CVVDP_E_ARG, "bad frame arguments");
  if (n_frames < 1 || n_frames > c.block_frames) return fail(h, CVVDP_E_ARG, "n_frames out of range");
  if (q_frame_offset < 0 || q_frame_offset + n_frames > c.n_frames) return fail(h, CVVDP_E_ARG, "frame offset out of range");
  const int fl_clip = c.filter_len;
  if (fl_clip > 1 && !hist_src) return fail(h, CVVDP_E_ARG, "hist_src missing");
  // the kernels may run a longer filter with zero taps in front (fir_kernel_len): `pad` extra, weightless window positions
  const int fl = fir_kernel_len(fl_clip), pad = fl - fl_clip;
  hipStream_t s = static_cast<hipStream_t>(stream);
  FirArgs f{};
  const int64_t P0 = h->lv[0].P;
  f.src[0] = t; f.src[1] = r;
  const int64_t* S[2] = {st, sr};
  // 1-channel clips broadcast Y into the three colour planes (cvvdp_metric.py:464-465): a zero channel stride makes
  // the FIR kernels read the same sample three times instead of branching (a fixed number of loads per frame)
  for (int k = 0; k < 2; ++k) { f.sb[k] = S[k][0]; f.sc[k] = c.channels == 3 ? S[k][1] : 0; f.sf[k] = S[k][2]; f.sh[k] = S[k][3]; f.sw[k] = S[k][4]; }
  f.dtype = dtype;
  if (yuv) f.yuv = *yuv;
  fill_display(h, f.dm);
  f.W = c.width; f.P = (int)P0; f.batch = c.batch; f.n_frames = n_frames; f.fl = fl;
  f.raw_first = raw_first;
  f.write_hist = !c.raw_halo && q_frame_offset + n_frames < c.n_frames;   // the DKL tail is only read by the next block of this clip
  f.abs_first = c.first_frame + q_frame_offset;
  f.hist = h->ws + h->hist_off;
  f.h_b = P0; f.h_slot = (int64_t)c.batch * P0; f.h_plane = (int64_t)(fl - 1) * f.h_slot; f.h_side = 3 * f.h_plane;
  const int set = h->pipeline ? h->cur_set : 0;
  if (h->pipeline && h->band_pending[set]) {   // this pyramid set is still being read by the band stage of block k-2
    (void)hipStreamWaitEvent(s, h->ev_band[set], 0);
    h->band_pending[set] = false;
  }
  f.out = gbase(h, 0, set); f.o_plane = (int64_t)h->items_cap * P0;
  for (int ch = 0; ch < 4; ++ch)
    for (int k = 0; k < std::min(2 * fl, CVVDP_MAX_FILTER_LEN); ++k)   // F.flip(0) (:556), written twice back to back so
      f.taps[ch * CVVDP_MAX_FILTER_LEN + k] = c.taps[ch * CVVDP_MAX_FILTER_LEN + (fl - 1 - (k % fl))];   // that a rotated view is contiguous
  if (fl >= 3 && 2 * (fl - 1) <= CVVDP_ROT_NEW) {   // rotating-window kernel (fl <= 31): taps of positions 0..fl-2 twice, newest at [CVVDP_ROT_NEW]
    const int M = fl - 1;
    for (int ch = 0; ch < 4; ++ch) {
      for (int i = 0; i < 2 * M; ++i) f.taps_rot[ch * CVVDP_ROT_TAPS + i] = c.taps[ch * CVVDP_MAX_FILTER_LEN + (fl - 1 - (i % M))];
      f.taps_rot[ch * CVVDP_ROT_TAPS + CVVDP_ROT_NEW] = c.taps[ch * CVVDP_MAX_FILTER_LEN + 0];   // position fl-1 (newest) <- F[0]
    }
  }
  for (int k = 0; k < fl_clip - 1; ++k) {
    const int e = hist_src[k];
    if (e >= 32767 || e < -(fl_clip - 1)) return fail(h, CVVDP_E_ARG, "hist_src[%d] = %d out of range", k, e);
    if (e < 0 && c.raw_halo) return fail(h, CVVDP_E_ARG, "hist_src[%d] refers to the DKL tail, but the clip was configured with raw_halo", k);
    f.hist_src[pad + k] = (int16_t)(e >= 0 ? e : e - pad);     // the kernel's tail has `pad` more (older) slots in front
  }
  for (int k = 0; k < pad; ++k)     // weightless positions: any finite frame will do -- the oldest real entry, or the tail's own slot
    f.hist_src[k] = (fl_clip > 1 && hist_src[0] >= 0) ? (int16_t)hist_src[0] : (int16_t)(-1 - k);
  f.halo_run = fl > 1 && raw_first >= fl - 1;
  for (int k = 0; k < fl - 1 && f.halo_run; ++k) f.halo_run = f.hist_src[k] == raw_first - (fl - 1) + k;
  {
    ProfScope ps(h, CVVDP_PROF_FIR, s);
    launch_fir(f, h->ws + h->hist_shadow_off, s);
  }
  if (int e = check_launch(h, "temporal fir")) return e;
  if (c.defer_bands) { h->filtered_frames = n_frames; h->filtered_q0 = q_frame_offset; return CVVDP_OK; }   // scored in pieces: cvvdp_score_frames
  return run_pyramid_and_bands(h, n_frames, q_frame_offset, s);
}

int cvvdp_process_block_filtered(cvvdp_handle* h, const void* t, const void* r, const int64_t st[5], const int64_t sr[5],
                                 int32_t n_frames, int32_t q_frame_offset, void* stream) {
  if (!h || !h->ws) return fail(h, CVVDP_E_STATE, "no workspace bound");
  const cvvdp_clip& c = h->c;
  if (!c.is_video) return fail(h, CVVDP_E_STATE, "configured for an image");
  if (!t || !r || !st || !sr) return fail(h, CVVDP_E_ARG, "bad frame arguments");
  if (n_frames < 1 || n_frames > c.block_frames) return fail(h, CVVDP_E_ARG, "n_frames out of range");
  if (q_frame_offset < 0 || q_frame_offset + n_frames > c.n_frames) return fail(h, CVVDP_E_ARG, "frame offset out of range");
  if (h->pipeline) return fail(h, CVVDP_E_UNSUPPORTED, "pre-filtered frames are not supported with CVVDP_PIPELINE");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PutPlanesArgs a{};
  a.src[0] = static_cast<const float*>(t); a.src[1] = static_cast<const float*>(r);
  const int64_t* S[2] = {st, sr};
  for (int k = 0; k < 2; ++k) { a.sb[k] = S[k][0]; a.sc[k] = S[k][1]; a.sf[k] = S[k][2]; a.sh[k] = S[k][3]; a.sw[k] = S[k][4]; }
  a.H = c.height; a.W = c.width; a.batch = c.batch; a.n_frames = n_frames;
  a.dst = gbase(h, 0, 0); a.o_plane = (int64_t)h->items_cap * h->lv[0].P;
  {
    ProfScope ps(h, CVVDP_PROF_FIR, s);
    launch_put_planes(a, s);
  }
  if (int e = check_launch(h, "put planes")) return e;
  if (c.defer_bands) { h->filtered_frames = n_frames; h->filtered_q0 = q_frame_offset; return CVVDP_OK; }
  return run_pyramid_and_bands(h, n_frames, q_frame_offset, s);
}

int cvvdp_score_frames(cvvdp_handle* h, int32_t first, int32_t n_frames, void* stream) {
  if (!h || !h->ws) return fail(h, CVVDP_E_STATE, "no workspace bound");
  const cvvdp_clip& c = h->c;
  if (!c.is_video || !c.defer_bands) return fail(h, CVVDP_E_STATE, "the clip was not configured with defer_bands");
  if (first < 0 || n_frames < 1 || n_frames > c.score_frames || first + n_frames > h->filtered_frames)
    return fail(h, CVVDP_E_ARG, "frames %d .. %d are not a piece (at most %d frames) of the %d frames the last block left", first, first + n_frames - 1,
                c.score_frames, h->filtered_frames);
  return run_pyramid_and_bands(h, n_frames, h->filtered_q0 + first, static_cast<hipStream_t>(stream), first * c.batch);
}

int cvvdp_get_features(cvvdp_handle* h, int32_t band, int32_t n_frames, float* dev_out, void* stream) {
  if (!h || !h->ws) return fail(h, CVVDP_E_STATE, "no workspace bound");
  if (h->c.feature_size <= 0) return fail(h, CVVDP_E_STATE, "features not enabled (cvvdp_clip.feature_size)");
  if (band < 0 || band >= h->L) return fail(h, CVVDP_E_ARG, "bad band");
  if (!dev_out || n_frames < 1 || n_frames * h->c.batch != h->last_items) return fail(h, CVVDP_E_ARG, "n_frames does not match the last block");
  const Level& lv = h->lv[band];
  if (lv.feat4) {
    FeatFinishArgs f{};
    f.fsum = h->ws + lv.fs_off;
    f.H = lv.H; f.W = lv.W; f.items = h->last_items; f.nch = h->nch; f.fs = h->c.feature_size;
    f.Hc = (lv.H + f.fs - 1) / f.fs; f.Wc = (lv.W + f.fs - 1) / f.fs; f.f_pieces = lv.f_pieces; f.seg_h = lv.seg_h;
    for (int c = 0; c < 4; ++c) f.inv_gain[c] = 1.0f / h->p.ch_gain[c];
    f.out = dev_out;
    launch_feature_finish(f, static_cast<hipStream_t>(stream));
    return check_launch(h, "feature finish");
  }
  FeatPoolArgs a{};
  a.tr = h->ws + lv.fd_off; a.d = h->ws + lv.dd_off;
  a.H = lv.H; a.W = lv.W; a.items = h->last_items; a.items_cap = h->items_cap_s; a.nch = h->nch; a.fs = h->c.feature_size;
  a.Hc = (lv.H + a.fs - 1) / a.fs; a.Wc = (lv.W + a.fs - 1) / a.fs;
  // the band kernels' T', R' carry the channel gain (cvvdp_metric.py:835); the features are |T_f|*S without it (cvvdp_ml_metric.py:355)
  for (int c = 0; c < 4; ++c) a.inv_gain[c] = band == h->L - 1 ? 1.0f : 1.0f / h->p.ch_gain[c];
  a.out = dev_out;
  launch_feature_pool(a, static_cast<hipStream_t>(stream));
  return check_launch(h, "feature pool");
}

int cvvdp_process_image(cvvdp_handle* h, void* stream) {
  if (!h || !h->ws) return fail(h, CVVDP_E_STATE, "no workspace bound");
  if (h->c.is_video) return fail(h, CVVDP_E_STATE, "configured for video");
  return run_pyramid_and_bands(h, 1, 0, static_cast<hipStream_t>(stream));
}

int cvvdp_get_q_per_ch(cvvdp_handle* h, float* dev_out, void* stream) {
  if (!h || !h->ws) return fail(h, CVVDP_E_STATE, "no workspace bound");
  if (!dev_out) return fail(h, CVVDP_E_ARG, "null output");
  join_pipeline(h, static_cast<hipStream_t>(stream));
  const size_t n = (size_t)h->c.batch * h->nch * h->c.n_frames * h->L;
  hipError_t e = hipMemcpyAsync(dev_out, h->ws + h->q_off, n * sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(h, CVVDP_E_HIP, "copy Q_per_ch: %s", hipGetErrorString(e));
  return CVVDP_OK;
}

int cvvdp_pool_jod(cvvdp_handle* h, const float* q, int32_t B, int32_t C, int32_t F, int32_t bands, float* jod, void* stream) {
  if (!h) return CVVDP_E_ARG;
  if (!q || !jod || B < 1 || C < 1 || C > 4 || F < 1 || bands < 1) return fail(h, CVVDP_E_ARG, "bad pooling arguments");
  PoolArgs a{};
  a.q = q; a.B = B; a.C = C; a.F = F; a.L = bands;
  for (int c = 0; c < 4; ++c) { a.ch_w[c] = h->p.ch_w[c]; a.bb_w[c] = h->p.baseband_weight[c]; }
  a.beta_sch = h->p.beta_sch; a.beta_tch = h->p.beta_tch; a.beta_t = h->p.beta_t;
  a.jod_a = h->p.jod_a; a.jod_exp = h->p.jod_exp; a.image_int = h->p.image_int;
  a.jod = jod;
  launch_pool(a, static_cast<hipStream_t>(stream));
  return check_launch(h, "pool");
}

static int get_heatmap_impl(cvvdp_handle* h, int32_t n_frames, void* dev_out_f16, int out_u8, void* stream);
int cvvdp_get_heatmap(cvvdp_handle* h, int32_t n_frames, void* dev_out_f16, void* stream) {
  return get_heatmap_impl(h, n_frames, dev_out_f16, 0, stream);
}
int cvvdp_get_heatmap_rgb8(cvvdp_handle* h, int32_t n_frames, void* dev_out_u8, void* stream) {
  return get_heatmap_impl(h, n_frames, dev_out_u8, 1, stream);
}
static int get_heatmap_impl(cvvdp_handle* h, int32_t n_frames, void* dev_out_f16, int out_u8, void* stream) {
  if (!h || !h->ws) return fail(h, CVVDP_E_STATE, "no workspace bound");
  if (h->c.heatmap == CVVDP_HEATMAP_NONE) return fail(h, CVVDP_E_STATE, "heat map not enabled");
  if (!dev_out_f16 || n_frames < 1 || n_frames * h->c.batch != h->last_items) return fail(h, CVVDP_E_ARG, "n_frames does not match the last block");
  hipStream_t s = static_cast<hipStream_t>(stream);
  HeatArgs a{};
  a.recon = h->ws + h->lv[0].heat_off;
  if (heat_l0_fused(h)) {
    a.coarse = h->ws + h->lv[1].heat_off;
    a.H = h->lv[0].H; a.W = h->lv[0].W; a.Hc = h->lv[1].H; a.Wc = h->lv[1].W;
    const float K0 = kGaussK0, K1 = 0.25f, K2 = 0.4f;      // lpyr_dec.py:179 (as in run_bands)
    a.kx[0] = K0 * 2.0f; a.kx[1] = K2 * 2.0f; a.kx[2] = K1 * 2.0f;
  }
  a.ctx = h->ws + h->lv[0].g_off + (size_t)h->last_item0 * h->lv[0].P;  // plane 0 = test Y-sustained (cvvdp_metric.py:400)
  a.P = (int)h->lv[0].P; a.items = h->last_items; a.mode = h->c.heatmap;
  a.jod_a = h->p.jod_a; a.jod_exp = h->p.jod_exp;
  a.jod_lin = h->p.jod_a * powf(0.1f, h->p.jod_exp - 1.0f);
  a.stats = reinterpret_cast<uint32_t*>(h->ws + h->hstats_off);
  a.range_done = h->last_range_done ? 1 : 0;
  a.curve = h->ws + h->hcurve_off;
  a.out = dev_out_f16; a.out_u8 = out_u8;
  a.pixel_layout = h->c.band_layout == 1;
  ProfScope ps(h, CVVDP_PROF_HEATMAP, s);
  if (h->c.heatmap == CVVDP_HEATMAP_RAW) {
    launch_heat_raw(a, s);
  } else {
    // colour maps of visualize_diff_map.py:59-94, normalised by their luminance
    static const float thr[5][3] = {{0.2f, 0.2f, 1.0f}, {0.2f, 1.0f, 1.0f}, {0.2f, 1.0f, 0.2f}, {1.0f, 1.0f, 0.2f}, {1.0f, 0.2f, 0.2f}};
    static const float sup[3][3] = {{0.2f, 1.0f, 1.0f}, {1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, 0.2f}};
    const bool is_thr = h->c.heatmap == CVVDP_HEATMAP_THRESHOLD;
    a.n_nodes = is_thr ? 5 : 3;
    for (int k = 0; k < a.n_nodes; ++k) {
      const float* row = is_thr ? thr[k] : sup[k];
      volatile float l = row[0] * 0.212656f;
      volatile float l1 = row[1] * 0.715158f;
      volatile float l2 = row[2] * 0.072186f;
      volatile float lum = l + l1;
      lum = lum + l2;
      for (int ch = 0; ch < 3; ++ch) a.cch[k * 3 + ch] = row[ch] / (lum + 0.0001f);
      a.cin[k] = is_thr ? (0.25f * k) * 0.1f : (0.5f * k) * 0.3f;
    }
    launch_heat_colour(a, s);
  }
  return check_launch(h, "heat map");
}

int cvvdp_debug_buffer(cvvdp_handle* h, int32_t which, int32_t level, void** dev_ptr, size_t* n_floats) {
  if (!h || !h->ws || !dev_ptr || !n_floats) return fail(h, CVVDP_E_STATE, "no workspace bound");
  if (level < 0 || level >= h->L) return fail(h, CVVDP_E_ARG, "bad level");
  const Level& lv = h->lv[level];
  switch (which) {
    case CVVDP_BUF_HIST:
      if (!h->c.is_video) return fail(h, CVVDP_E_STATE, "no temporal history for images");
      *dev_ptr = h->ws + h->hist_off; *n_floats = (size_t)2 * 3 * (fir_kernel_len(h->c.filter_len) - 1) * h->c.batch * h->lv[0].P; break;
    case CVVDP_BUF_GPYR: *dev_ptr = h->ws + lv.g_off; *n_floats = (size_t)2 * h->nch * level_cap(h, level) * lv.P; break;
    case CVVDP_BUF_DDUMP:
      if (!h->c.debug_dump && h->c.feature_size <= 0) return fail(h, CVVDP_E_STATE, "debug_dump not enabled");
      if (lv.feat4 && !h->c.debug_dump) return fail(h, CVVDP_E_STATE, "level %d keeps column sums in features mode, not per-pixel D planes", level);
      *dev_ptr = h->ws + lv.dd_off; *n_floats = (size_t)4 * h->items_cap_s * lv.P; break;
    case CVVDP_BUF_HEAT:
      if (h->c.heatmap == CVVDP_HEATMAP_NONE) return fail(h, CVVDP_E_STATE, "heat map not enabled");
      *dev_ptr = h->ws + lv.heat_off; *n_floats = (size_t)h->items_cap_s * lv.P; break;
    case CVVDP_BUF_Q: *dev_ptr = h->ws + h->q_off; *n_floats = (size_t)h->c.batch * h->nch * h->c.n_frames * h->L; break;
    default: return fail(h, CVVDP_E_ARG, "unknown buffer");
  }
  return CVVDP_OK;
}

int cvvdp_profile_enable(cvvdp_handle* h, int32_t enable) {
  if (!h) return CVVDP_E_ARG;
  h->prof = enable != 0;
  h->events_used = 0;
  return CVVDP_OK;
}

int cvvdp_profile_read(cvvdp_handle* h, double total_ms[CVVDP_PROF_N], int32_t n_launches[CVVDP_PROF_N]) {
  if (!h || !total_ms || !n_launches) return CVVDP_E_ARG;
  for (int i = 0; i < CVVDP_PROF_N; ++i) { total_ms[i] = 0.0; n_launches[i] = 0; }
  for (size_t i = 0; i < h->events_used; ++i) {
    ProfEvent& e = h->events[i];
    if (hipEventSynchronize(e.b) != hipSuccess) return fail(h, CVVDP_E_HIP, "event sync failed");
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) return fail(h, CVVDP_E_HIP, "event elapsed failed");
    total_ms[e.cat] += ms;
    n_launches[e.cat] += 1;
  }
  h->events_used = 0;
  return CVVDP_OK;
}

}  // extern "C"
