/* cvvdp_hip.h -- C ABI of the MI355X (gfx950) ColorVideoVDP compute core.
 *
 * The reference (gfxdisp/ColorVideoVDP, pure Python/PyTorch) has no FFI; its extension point
 * is the Python class `cvvdp` (pycvvdp/cvvdp_metric.py:108).  This header is the boundary a
 * binding for that class talks to: every entry point replaces a span of reference code, cited
 * next to it (paths relative to the reference root).  The Python mirror that consumes it is
 * colorvideovdp_amd/cvvdp_metric.py through ctypes (colorvideovdp_amd/_capi.py); see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; all `dev` pointers are HIP device pointers (e.g. tensor.data_ptr()).
 *   - every function returns 0 on success, a negative CVVDP_E_* code otherwise; nothing throws.
 *     cvvdp_last_error() returns a human-readable message for the last failure of the CALLING THREAD
 *     (thread-local, so a worker thread's failure cannot garble the main thread's message).
 *   - one handle per (process, GPU); a handle is not thread-safe, with one exception: cvvdp_unpack_yuv_resized
 *     touches no handle state and may run on a prefetch thread beside the other calls.
 *   - all device work is enqueued on the caller's hipStream_t (passed as void*).  Host-side waits happen in
 *     exactly two places: cvvdp_profile_read (waits for its own timing events) and cvvdp_destroy (drains the
 *     helper streams the handle owns).  No other entry point synchronises.
 *   - the core allocates no device MEMORY: the caller provides one workspace buffer of
 *     cvvdp_workspace_bytes() bytes (so torch's caching allocator stays the only allocator).  It does own a few
 *     HIP objects, created lazily on first use and destroyed with the handle: up to five non-blocking helper
 *     streams with their fork / join events -- two side streams for the small pyramid levels of images and short blocks (and the levels
 *     behind the fused ones of large blocks), which run beside the caller's stream, and per stream a level can run on (the caller's, the
 *     two side streams) one edge stream on which the border strips of that level run beside its border-free strips; the caller's stream
 *     waits for all of them by event before anything reads the results -- and, while profiling is enabled, timing events.
 *   - scores do not depend on how a clip is cut into blocks or shards (bit for bit), but the band kernels a level
 *     runs on depend on what else is asked for: plain scoring, heat maps and features have their own kernels on the
 *     fused route (k_band4s / _heat / _feat; features keep k_band4f_feat on the border strips) and the debug dump runs
 *     the unfused route (reduce pass + k_band4), so Q_per_ch / JOD of one clip with and without a heat map (or features,
 *     or the dump) agree to rounding (observed <= 5e-5 relative in Q_per_ch), not bit for bit.  cvvdp_clip.fuse_mode = 2
 *     pins the unfused route for callers who need equality.
 *   - the library reads no environment variable (tuning knobs exist only in a -DCVVDP_DEV_KNOBS build,
 *     cvvdp_build_flags()).
 *   - "item" = one (frame-in-block, batch) pair; item index = frame * batch + b.
 */
#ifndef CVVDP_HIP_H
#define CVVDP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVVDP_ABI_VERSION 13
#define CVVDP_MAX_FILTER_LEN 65 /* 0.25 s at up to 256 fps, cvvdp_metric.py:1059 */
#define CVVDP_MAX_LEVELS 16
#define CVVDP_MAX_WINDOW 256    /* filter_len - 1 + frames per block */
#define CVVDP_CSF_NODES 32

enum {
  CVVDP_OK = 0,
  CVVDP_E_ARG = -1,      /* invalid argument */
  CVVDP_E_STATE = -2,    /* call order violated (e.g. no workspace bound) */
  CVVDP_E_HIP = -3,      /* a HIP runtime call or kernel launch failed */
  CVVDP_E_UNSUPPORTED = -4
};

/* input sample formats, video_source.py:320-346 */
enum { CVVDP_U8 = 0, CVVDP_U16 = 1, CVVDP_F16 = 2, CVVDP_F32 = 3, CVVDP_F32_DKL = 4 /* already DKL-d65, fp32 */,
       /* planar Y'CbCr frames, video_source_yuv.py:79-223; only through cvvdp_process_block_yuv */
       CVVDP_YUV8 = 5, CVVDP_YUV16 = 6 };
/* EOTFs, display_model.py:333-365 */
enum { CVVDP_EOTF_SRGB = 0, CVVDP_EOTF_PQ = 1, CVVDP_EOTF_HLG = 2, CVVDP_EOTF_LINEAR = 3, CVVDP_EOTF_GAMMA = 4 };
/* heat-map modes, cvvdp_metric.py:117 */
enum { CVVDP_HEATMAP_NONE = 0, CVVDP_HEATMAP_RAW = 1, CVVDP_HEATMAP_THRESHOLD = 2, CVVDP_HEATMAP_SUPRA = 3 };
/* debug/inspection buffers inside the workspace (tests) */
enum { CVVDP_BUF_HIST = 0, CVVDP_BUF_GPYR = 1, CVVDP_BUF_DDUMP = 2, CVVDP_BUF_HEAT = 3, CVVDP_BUF_Q = 4 };

/* Calibrated parameters + display photometry.  Host-side scalars of cvvdp_parameters.json as
 * loaded by cvvdp.load_config (cvvdp_metric.py:146-229) and of vvdp_display_photo_eotf
 * (display_model.py:301-376).  Derived constants (10^x, 2^x) are computed by the host mirror in
 * fp32 exactly where the reference computes them in fp32. */
typedef struct cvvdp_params {
  /* display photometry, display_model.py:333-376 */
  int32_t eotf;
  float Y_peak, Y_black, Y_refl, exposure;
  float gamma;        /* EOTF exponent for CVVDP_EOTF_GAMMA; system gamma for HLG */
  float rgb2dkl[9];   /* row-major fp32 LMS2006_to_DKLd65 @ XYZ_to_LMS2006 @ rgb2xyz, display_model.py:256 */
  /* masking model "mult-mutual", cvvdp_metric.py:835-856 */
  float mask_p;
  float mask_c10;     /* 10^mask_c */
  float mask_q[4];
  float xcm[16];      /* 2^xcm_weights reshaped [from][to], cvvdp_metric.py:758-760 */
  float ch_gain[4];   /* [1, 1.45, 1, 1], cvvdp_metric.py:835 */
  float d_max10;      /* 10^d_max, cvvdp_metric.py:949 */
  float sens_mul;     /* 10^(sensitivity_correction/20), cvvdp_metric.py:709 */
  int32_t blur_radius;/* int(pu_dilate*2) = 6; 0 disables phase uncertainty blur */
  float blur_taps[13];/* torchvision GaussianBlur(13, 3) 1-D kernel */
  /* pooling + JOD, cvvdp_metric.py:610-658 */
  float beta, beta_t, beta_tch, beta_sch;
  float jod_a, jod_exp, image_int;
  float ch_w[4];           /* [1, ch_chrom_w, ch_chrom_w, ch_trans_w] */
  float baseband_weight[4];
  /* castleCSF luminance axis (uniform in log10), csf.py:13, interp.py:92-100 */
  float csf_logL_first, csf_logL_last;
} cvvdp_params;

/* Per-clip configuration: geometry of the pyramid, temporal filters, CSF rows.
 * Replaces the host-side set-up in cvvdp.predict_video_source (cvvdp_metric.py:304-363),
 * lpyr_dec.__init__ (lpyr_dec.py:18-52), get_temporal_filters (cvvdp_metric.py:1057-1092) and the
 * rho-interpolated CSF cache (csf.py:39-46). */
typedef struct cvvdp_clip {
  int32_t batch, channels;      /* B; C in {1,3} of the input arrays */
  int32_t height, width;
  int32_t is_video;             /* 0: image (3 channels, no temporal filter), 1: video (4 channels) */
  int32_t n_frames;             /* frames this handle scores (capacity of Q_per_ch along F) */
  int32_t first_frame;          /* clip index of the first scored frame (0, or the start of a frame-range shard): keys the
                                   temporal-window rotation so that per-frame sums round identically for any blocking */
  int32_t n_levels;             /* pyramid band count (lpyr.get_band_count()) */
  int32_t filter_len;           /* temporal filter length (video) */
  int32_t block_frames;         /* max frames per process_block call */
  int32_t heatmap;              /* CVVDP_HEATMAP_* */
  int32_t debug_dump;           /* 1: keep per-pixel D of every band in the workspace (tests) */
  int32_t raw_halo;             /* 1: every block is handed its filter_len-1 predecessor frames as raw frames (hist_src >= 0),
                                   so no DKL tail is kept between blocks; 0: later blocks read the tail (hist_src < 0) */
  int32_t total_frames;         /* frames of the whole clip (all shards); 0 = unknown.  Sizes the band kernels' row segments:
                                   the same for every block and shard of a clip, so results stay bit-identical */
  int32_t feature_size;         /* > 0: also keep |T'|, |R'| and D of every band for cvvdp_get_features, pooled over
                                   feature_size x feature_size cells (ceil(pix_per_deg), cvvdp_ml_metric.py:351-355); 0: off */
  int32_t fuse_mode;            /* 0 (normal use): the core decides per clip which pyramid levels run the band kernel that computes the
                                   next level itself (no reduce pass for them; clips whose blocks fill the GPU several times over);
                                   1: every level that supports it, whatever the size of the clip; 2: none.  1 / 2 are test hooks: the
                                   two routes agree to rounding, not bit for bit */
  int32_t band_layout;          /* how a fused level's band kernel divides its work between waves.  0 (normal use): front / back waves
                                   (band4s.hip: 8 waves per block, four per SIMD); 1: one wave per channel (round 3's k_band4f everywhere,
                                   two per SIMD).  Same arithmetic, the same level-(l+1) planes bit for bit, per-frame sums equal to the last bit or so
                                   (two separately compiled kernels): 1 is the A/B switch of tests and benchmarks; a clip is scored by one layout.
                                   1 also selects the heat-map finishing kernel with one thread per 4 pixels where 0 runs the row-tile
                                   kernel (heatmap.hip k_heat_colour / k_heat_colour_rows: the same bits) */
  int32_t defer_bands;          /* 1: cvvdp_process_block* run the temporal stage only and leave the level-0 planes of the block (up to
                                   block_frames frames) in the workspace; the caller scores them in pieces of at most score_frames frames
                                   with cvvdp_score_frames.  For heat-map clips: the frames' heat maps leave the GPU piece by piece
                                   (16 frames), while the temporal stage pays its filter_len-1 halo frames once per LONG block
                                   (cvvdp_metric.py:554-560 has one window per clip; the block structure is this core's).  Everything
                                   behind level 0 (coarser levels, partial sums, heat bands) is sized for score_frames.  Scores and heat
                                   maps do not depend on either length.  Not with debug_dump or features */
  int32_t score_frames;         /* frames per cvvdp_score_frames call at most (read when defer_bands is set) */
  float taps[4 * CVVDP_MAX_FILTER_LEN];                             /* F[c][k], not flipped */
  float csf_rows[CVVDP_MAX_LEVELS * 4 * CVVDP_CSF_NODES];           /* [band][ch][node] log10 S */
} cvvdp_clip;

typedef struct cvvdp_handle cvvdp_handle;

int cvvdp_abi_version(void);
/* Bit set of build properties.  CVVDP_BUILD_DEV_KNOBS: compiled with -DCVVDP_DEV_KNOBS, i.e. development tuning
 * knobs are read from the environment (CVVDP_SEG_TARGET, CVVDP_R2_SEG, ...: they change launch geometry and with it
 * the last bits of Q_per_ch).  0 for the product build. */
#define CVVDP_BUILD_DEV_KNOBS 1
/* CVVDP_BUILD_SAFE_LOADS: the band kernels were compiled with -DCVVDP_SAFE_LOADS (`make safe`: compiler-managed loads, the reference
 * point of tests/test_safe_loads.py).  CVVDP_BUILD_DIAG: a band kernel was compiled with one of its timing-only switches (S_DIAG_*,
 * S_PRIO_*: RESULTS ARE WRONG by construction) or a non-default CVVDP_BAND4S_RING.  A binding must refuse a library whose
 * flags are not 0 unless it was asked for a development library explicitly (colorvideovdp_amd/_capi.py: CVVDP_DEV_KNOBS=1). */
#define CVVDP_BUILD_SAFE_LOADS 2
#define CVVDP_BUILD_DIAG 4
int cvvdp_build_flags(void);
/* The toolchain this library was compiled -- and its hand-scheduled band kernels statically checked (tools/check_band4_isa.py, run by
 * `make` on the assembly of the linked objects) -- with: "clang <version>; HIP <major.minor.patch>; gfx950".  The hand-issued loads of
 * the band kernels are verified against THAT compiler's register allocation; a binding compares the HIP version here with the
 * runtime's (cvvdp_runtime_hip_version) and says so when they differ (colorvideovdp_amd/_capi.py warns; results are guarded either
 * way by tests/test_safe_loads.py).  Static string, never NULL. */
const char* cvvdp_build_info(void);
/* HIP version the library was compiled with (HIP_VERSION = major * 10000000 + minor * 100000 + patch) and the version of the HIP
 * runtime it is running on (hipRuntimeGetVersion; 0 when the call fails, e.g. without a device). */
int cvvdp_compiled_hip_version(void);
int cvvdp_runtime_hip_version(void);
/* sizeof(cvvdp_params), sizeof(cvvdp_clip) as compiled, so a binding can verify its struct layout. */
void cvvdp_struct_sizes(int32_t* params_bytes, int32_t* clip_bytes);

/* cvvdp.__init__/load_config/set_display_model device side (cvvdp_metric.py:109-264). */
int cvvdp_create(const cvvdp_params* params, cvvdp_handle** out);
void cvvdp_destroy(cvvdp_handle* h);
const char* cvvdp_last_error(const cvvdp_handle* h);

/* Start of predict_video_source for one clip or frame-range shard (cvvdp_metric.py:304-372). */
int cvvdp_configure(cvvdp_handle* h, const cvvdp_clip* clip);
size_t cvvdp_workspace_bytes(const cvvdp_handle* h);
/* How many leading pyramid levels of the configured clip run the band kernel that computes the next level itself (no reduce pass
 * for them; cvvdp_clip.fuse_mode).  For reporting (bench.py prices the path's algorithmic bytes with it); -1 if not configured. */
int cvvdp_fused_levels(const cvvdp_handle* h);
int cvvdp_bind_workspace(cvvdp_handle* h, void* dev_workspace, size_t bytes);

/* Image frame supply + display model: video_source_array._get_frame (video_source.py:320-346),
 * vvdp_display_photo_eotf.forward (display_model.py:333-365), linear_2_target_colorspace 'DKLd65'
 * (display_model.py:241-276).  Converts the image pair (device arrays with element strides in
 * B,C,F,H,W order; a broadcast batch has stride 0) into the 6 level-0 planes (cvvdp_metric.py:462-465). */
int cvvdp_put_image(cvvdp_handle* h, const void* dev_test, const void* dev_ref, int32_t dtype,
                    const int64_t strides_test[5], const int64_t strides_ref[5], void* stream);

/* One block of video frames, everything from samples to Q_per_ch: frame supply + display model as above,
 * sliding-window temporal FIR (cvvdp_metric.py:453-560), contrast pyramid (lpyr_dec.py:364-414), CSF
 * (csf.py:28-51), masking + pooling per band (cvvdp_metric.py:691-734).
 *   dev_test/dev_ref  frame f of the handed-in arrays is "raw frame f" (element strides as above)
 *   raw_first         raw index of the first scored frame; frames raw_first .. raw_first+n_frames-1 are scored
 *   hist_src[k]       (host array, filter_len-1 entries) where sliding-window position k of the first scored
 *                     frame comes from, i.e. frame (first - (filter_len-1) + k) after temporal padding:
 *                     e >= 0: raw frame e of the handed-in arrays (replicate/symmetric padding of
 *                     cvvdp_metric.py:506-529, or the real halo frames of a frame-range shard);
 *                     e <  0: entry -1-e of the DKL tail kept from the previous call (replaces the ring +
 *                     torch.roll of cvvdp_metric.py:538-539).
 * The last filter_len-1 DKL frames are kept in the workspace for the next call.
 * Results land in Q_per_ch[:, :, q_frame_offset : q_frame_offset+n_frames, :]. */
int cvvdp_process_block(cvvdp_handle* h, const void* dev_test, const void* dev_ref, int32_t dtype,
                        const int64_t strides_test[5], const int64_t strides_ref[5], int32_t raw_first,
                        const int32_t* hist_src, int32_t n_frames, int32_t q_frame_offset, void* stream);

/* Planar Y'CbCr frames straight from a .yuv file.  Replaces YUVReader.get_frame_rgb_tensor /
 * _fixed2float_upscale (video_source_yuv.py:147-170, 197-223) + video_source_dm.apply_dm_and_color_transform
 * (video_source.py:217-229): limited-range fixed point -> float with clips, bilinear chroma up-sampling
 * (torch interpolate, align_corners=False), BT.709 / BT.2020 Y'CbCr -> R'G'B' matrix, clip to [0,1]; then the
 * display model, DKL and everything cvvdp_process_block does.  A frame is the Y plane followed by the U and the
 * V plane (YUVReader.get_frame_yuv, :131-145); samples are uint8 (bit_depth 8) or uint16 (bit_depth 9..16).
 * The clip must have been configured with channels = 3 and batch = 1. */
typedef struct cvvdp_yuv_format {
  int32_t chroma;         /* 420, 422 or 444 (video_source_yuv.py:98-112) */
  int32_t bit_depth;      /* 8..16 */
  int32_t matrix;         /* 709 or 2020 (video_source_yuv.py:151-160) */
  int32_t reserved;
  int64_t frame_stride_test, frame_stride_ref;   /* samples between consecutive frames of each buffer */
} cvvdp_yuv_format;
int cvvdp_process_block_yuv(cvvdp_handle* h, const void* dev_test, const void* dev_ref, const cvvdp_yuv_format* fmt,
                            int32_t raw_first, const int32_t* hist_src, int32_t n_frames, int32_t q_frame_offset,
                            void* stream);

/* full_screen_resize of .yuv sources (video_source_yuv.py:266-284 constructor, :333-336 in _get_frame): n_frames Y'CbCr frames of
 * src_width x src_height (layout and fmt as for cvvdp_process_block_yuv; is_ref selects fmt->frame_stride_ref) are unpacked to
 * display-encoded R'G'B' (YUVReader.get_frame_rgb_tensor, :147-170), resized with
 * torch.nn.functional.interpolate(size=(dst_height, dst_width), mode=...) semantics (align_corners False, no antialiasing) and
 * clipped to [0,1].  dev_rgb receives fp32 [3][n_frames][dst_height][dst_width] = a [1,3,n,H,W] block for cvvdp_process_block
 * (CVVDP_F32); dev_tmp is scratch for 3*n_frames*src_height*src_width floats.  Needs no configured clip. */
enum { CVVDP_RESIZE_NEAREST = 0, CVVDP_RESIZE_BILINEAR = 1, CVVDP_RESIZE_BICUBIC = 2, CVVDP_RESIZE_AREA = 3 };
int cvvdp_unpack_yuv_resized(cvvdp_handle* h, const void* dev_codes, const cvvdp_yuv_format* fmt, int32_t is_ref, int32_t src_width,
                             int32_t src_height, int32_t n_frames, int32_t dst_width, int32_t dst_height, int32_t mode, float* dev_tmp,
                             float* dev_rgb, void* stream);

/* Sources that deliver temporally pre-filtered channels (vid_source.is_temporally_filtered, cvvdp_metric.py:470-488):
 * frames are fp32 [B, 4, n, H, W] in colour space 'DKLd65_trans' (Y-sustained, RG, YV, Y-transient; element strides in
 * B,C,F,H,W order) and go straight into the 8 level-0 planes (test channel c -> plane 2c, reference -> 2c+1), bypassing
 * the sliding window and the temporal FIR; then the contrast pyramid and everything after it, as in cvvdp_process_block. */
int cvvdp_process_block_filtered(cvvdp_handle* h, const void* dev_test, const void* dev_ref,
                                 const int64_t strides_test[5], const int64_t strides_ref[5], int32_t n_frames,
                                 int32_t q_frame_offset, void* stream);

/* Clips configured with defer_bands: contrast pyramid, bands, pooling and heat-map bands of frames first .. first+n_frames-1 of the block
 * the last cvvdp_process_block* call filtered (n_frames <= score_frames).  Q of those frames lands where cvvdp_process_block would have
 * put it; cvvdp_get_heatmap* afterwards returns the heat maps of exactly these n_frames.  Pieces may be scored in any order, each
 * frame once.  Reference: the per-frame body of predict_video_source, cvvdp_metric.py:596-744. */
int cvvdp_score_frames(cvvdp_handle* h, int32_t first, int32_t n_frames, void* stream);

/* Features for the ML heads: cvvdp_feature_pooling of |T_f|*S, |R_f|*S and D (cvvdp_ml_metric.py:77-107 called
 * at :355-358) for one band of the block processed last: mean and variance (E[x^2] - mean^2) over feature_size x
 * feature_size cells, the last cells of a row / column averaging over the pixels that exist (AvgPool2d, ceil_mode).
 * dev_out: fp32 [n_frames * B][ceil(H_band / fs)][ceil(W_band / fs)][C][6] (mean_T, var_T, mean_R, var_R, mean_D, var_D),
 * item index = frame * B + batch.  The clip must have been configured with feature_size > 0. */
int cvvdp_get_features(cvvdp_handle* h, int32_t band, int32_t n_frames, float* dev_out, void* stream);

/* Image variant: pyramid, CSF, masking, pooling of the planes written by cvvdp_put_image. */
int cvvdp_process_image(cvvdp_handle* h, void* stream);

/* stats['Q_per_ch'] as fp32 [B, C, F, bands] (cvvdp_metric.py:388-392,419). */
int cvvdp_get_q_per_ch(cvvdp_handle* h, float* dev_out, void* stream);
/* do_pooling_and_jods + met2jod (cvvdp_metric.py:610-658) on any [B,C,F,bands] device array. */
int cvvdp_pool_jod(cvvdp_handle* h, const float* dev_q_per_ch, int32_t B, int32_t C, int32_t F,
                   int32_t bands, float* dev_jod, void* stream);

/* Heat map of the frames of the last processed block: lpyr_dec_2.reconstruct + met2jod
 * (cvvdp_metric.py:724-744) and visualize_diff_map (visualize_diff_map.py:48-106, tone-mapped per
 * frame = block of 1, the reference's CPU behaviour).  Output fp16 [channels(1|3), n_frames, H, W]. */
int cvvdp_get_heatmap(cvvdp_handle* h, int32_t n_frames, void* dev_out_f16, void* stream);
/* The same frames as the reference's file writers encode them (np2vid / np2img, run_cvvdp.py:44-78): the fp16 value clipped to
 * [0,1], times 255, truncated.  Output uint8 [n_frames, H, W, channels(1|3)] (interleaved RGB): 3 instead of 6 bytes per pixel
 * to bring to the host, and nothing left to convert there. */
int cvvdp_get_heatmap_rgb8(cvvdp_handle* h, int32_t n_frames, void* dev_out_u8, void* stream);

/* Test/inspection: device pointer + element count of an internal buffer (level where relevant).  CVVDP_BUF_HEAT: level l holds the
 * heat-map reconstruction from the coarsest level up to l (lpyr_dec.py:328-335) -- EXCEPT level 0 of frames with at least two pyramid
 * levels and W % 4 == 0: there the last step (level 1 -> 0) is done inside the finishing kernels (cvvdp_get_heatmap*), so level 0 holds
 * the bare level-0 band and the full reconstruction never exists as a plane. */
int cvvdp_debug_buffer(cvvdp_handle* h, int32_t which, int32_t level, void** dev_ptr, size_t* n_floats);

/* Profiling aid for bench.py: when enabled, the core brackets every kernel launch with hipEvents on
 * the caller's stream, binned by kernel family.  cvvdp_profile_read synchronises on the recorded
 * events, returns summed milliseconds and launch counts per family and resets the accumulators. */
enum {
  CVVDP_PROF_PHOTOMETRY = 0, CVVDP_PROF_FIR = 1, CVVDP_PROF_REDUCE = 2, CVVDP_PROF_BAND0 = 3,
  CVVDP_PROF_BAND_REST = 4, CVVDP_PROF_HEATMAP = 5, CVVDP_PROF_N = 6
};
int cvvdp_profile_enable(cvvdp_handle* h, int32_t enable);
int cvvdp_profile_read(cvvdp_handle* h, double total_ms[CVVDP_PROF_N], int32_t n_launches[CVVDP_PROF_N]);

#ifdef __cplusplus
}
#endif
#endif /* CVVDP_HIP_H */
