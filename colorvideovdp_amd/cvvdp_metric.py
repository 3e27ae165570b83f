"""ColorVideoVDP metric class for MI355X: host-side mirror of pycvvdp/cvvdp_metric.py `class cvvdp`.

Same constructor, methods, argument meaning, outputs and error behaviour as the reference class
(cvvdp_metric.py:108-441, 1094-1218), so `from colorvideovdp_amd import cvvdp` is a drop-in for
`from pycvvdp import cvvdp` on this path.  All per-pixel work is done by the hand-written gfx950
kernels behind the C ABI in include/cvvdp_hip.h; this file only plans blocks of frames, moves
pointers and mirrors the reference's host-side bookkeeping (temporal padding, stats dict, distogram).

Differences that are deliberate:
  * GPU only.  There is no CPU path and no fallback: without the HIP library or a GPU the class raises.
  * inference only.  `loss()` returns 10-JOD without autograd (the reference is differentiable).
  * frames per block are chosen for occupancy / memory, never change results beyond fp32 rounding, and
    heat-map tone-mapping statistics are always per frame (the reference's CPU behaviour; on CUDA the
    reference lets them depend on the block size).
  * optional frame-range sharding over torch.distributed (RCCL): see `set_frame_sharding`.
"""
import ctypes
import json
import math

import numpy as np
import torch

from . import _capi
from . import host_setup as hs
from .config import config_files, json2dict, load_config
from .display_model import vvdp_display_geometry, vvdp_display_photo_eotf, vvdp_display_photometry
from .sharding import all_gather_frames, plan_frame_shard
from .video_source import video_source, video_source_array
from .vq_metric import register_metric, vq_exception, vq_metric

f32 = np.float32


def _storage_is_unshared(t):
    """True if no tensor / array other than `t` itself refers to t's storage (private torch API; False if unavailable)."""
    try:
        return torch._C._storage_Use_Count(t.untyped_storage()._cdata) <= 2   # t + the temporary storage object
    except Exception:
        return False


class _SequentialFrames:
    """Frames of a generic video_source (get_test_frame / get_reference_frame, already in the metric's colour space).  Every
    frame is requested exactly ONCE and in increasing order -- the reference's file sources are strictly sequential
    (video_source_file.py raises 'Random access not implemented' otherwise) -- and kept until no later block can need it:
    the read-ahead frames of symmetric padding (cvvdp_metric.py:513-529, fb.ra_buf) are served from here."""

    def __init__(self, vs, device, colorspace):
        self.vs, self.device, self.colorspace = vs, device, colorspace
        self.frames = {}

    def block(self, a, b, reference_first=False):
        for f in range(a, b):
            if f not in self.frames:
                if reference_first:   # cvvdp_metric.py:476-478: per-frame state (optical flow) is computed from the reference
                    r = self.vs.get_reference_frame(f, device=self.device, colorspace=self.colorspace)
                    t = self.vs.get_test_frame(f, device=self.device, colorspace=self.colorspace)
                else:
                    t = self.vs.get_test_frame(f, device=self.device, colorspace=self.colorspace)
                    r = self.vs.get_reference_frame(f, device=self.device, colorspace=self.colorspace)
                self.frames[f] = (t.to(self.device, torch.float32), r.to(self.device, torch.float32))
        t = torch.cat([self.frames[f][0] for f in range(a, b)], dim=2).contiguous()
        r = torch.cat([self.frames[f][1] for f in range(a, b)], dim=2).contiguous()
        for f in [f for f in self.frames if f < a]:     # blocks move forward: nothing before this block's first frame comes back
            del self.frames[f]
        return t, r, _capi.F32_DKL


class cvvdp(vq_metric):
    def __init__(self, display_name="standard_4k", display_photometry=None, display_geometry=None, config_paths=[],
                 heatmap=None, quiet=False, device=None, temp_padding="replicate", use_checkpoints=False, dump_channels=None,
                 gpu_mem=None, block_frames=None):
        self.quiet = quiet
        self.heatmap = heatmap
        self.temp_padding = temp_padding
        self.use_checkpoints = use_checkpoints
        self.gpu_mem = gpu_mem
        self.block_frames = block_frames
        self.training_mode = False
        assert heatmap in ["threshold", "supra-threshold", "raw", "none", None], "Unknown heatmap type"
        self.do_heatmap = (self.heatmap is not None) and (self.heatmap != "none")
        if dump_channels:
            raise vq_exception("dump_channels is a debugging aid of the reference implementation and is not supported")
        self.dump_channels = None
        _capi.lib()  # fail loudly (ImportError) if the HIP library is missing
        if device is None:
            self.device = torch.device("cuda")
        else:
            self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("colorvideovdp_amd runs on an MI355X only: device must be a CUDA/HIP device; there is no CPU path")
        if self.device.index is None and torch.cuda.is_available():
            # "cuda" and "cuda:0" compare unequal: pin the index once, so that workspace / input tensors are recognised as local
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._cfg_version = 0          # bumped whenever parameters / display change: part of the clip-cache key
        self._handle = ctypes.c_void_p()
        self._ws = None
        self._shard = None
        self.debug_dump = False
        self.score_frames = None      # heat-map clips resident in HBM: frames per band / heat-map piece of a long temporal block (None: 16; >= the block: one piece)
        self.band_layout = 0          # cvvdp_clip.band_layout: 0 = front / back waves on fused levels (normal use); 1 = one wave per channel, and the per-thread heat-map finishing kernels (A/B switch; scores equal to the last bit or so, heat maps bit for bit)
        self.fuse_mode = 0            # cvvdp_clip.fuse_mode: 0 = the core decides (normal use); 1 / 2 = fused band kernels everywhere / nowhere (tests)
        self.set_display_model(display_name, display_photometry=display_photometry, display_geometry=display_geometry,
                               config_paths=config_paths)
        self.load_config(config_paths)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None and self._handle.value:
                _capi.lib().cvvdp_destroy(self._handle)
                self._handle = ctypes.c_void_p()
        except Exception:
            pass

    _INT_PARAMETERS = ("pu_dilate", "filter_len")      # the only numeric parameters that are integers by meaning (the betas are exponents)
    _LIST_LENGTHS = {"mask_q": 4, "xcm_weights": 16, "baseband_weight": 4, "sigma_tf": 4, "beta_tf": 4}

    def _param_device(self):
        return self.device if torch.cuda.is_available() else torch.device("cpu")

    def __getattr__(self, name):
        # only reached when normal lookup fails: the numeric model parameters as the reference's tensor attributes on the metric's
        # device (cvvdp_metric.py:153-219 keeps them as tensors on self.device) -- a view of self.parameters
        p = self.__dict__.get("parameters")
        if p is not None and name in p and not isinstance(p[name], (str, bool)):
            return torch.as_tensor(p[name], dtype=torch.float32 if isinstance(p[name], (float, list)) else None, device=self._param_device())
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        # `metric.mask_c = torch.tensor(..)` is how a parameter is changed on the reference (its parameters ARE attributes): here the value
        # goes into self.parameters and the core's handle is re-made from it, or the assignment raises and nothing changes
        # (`filter_len` is per-call state on the reference too: predict overwrites it with the length of the filters it made, :338)
        p = self.__dict__.get("parameters")
        if p is not None and name in p and name != "filter_len" and not isinstance(p[name], (str, bool)):
            q = dict(p)
            q[name] = self._parameter_value(name, value, p[name])
            self._set_parameters(q)
            return
        object.__setattr__(self, name, value)

    @classmethod
    def _parameter_value(cls, name, value, current):
        """A tensor / array / number handed in for parameter `name` as the plain float, int or list of floats self.parameters holds.
        Scalars stay floats (the core's parameters are all fp32: a checkpoint's beta_tch = 3.7 must not become 3); the two parameters
        that are integers by meaning refuse a fractional value; a list keeps the length the kernels are built for."""
        v = value.detach().to("cpu", torch.float32) if torch.is_tensor(value) else torch.as_tensor(np.asarray(value), dtype=torch.float32)
        if isinstance(current, list):
            if v.dim() != 1 or v.numel() != len(current):
                raise ValueError(f"parameter '{name}' holds {len(current)} values, got shape {tuple(v.shape)}")
            return [float(x) for x in v.tolist()]
        if v.numel() != 1:
            raise ValueError(f"parameter '{name}' is a scalar, got shape {tuple(v.shape)}")
        x = float(v.reshape(()).item())
        if name in cls._INT_PARAMETERS:
            if x != round(x):
                raise ValueError(f"parameter '{name}' is an integer, got {x}")
            return int(round(x))
        return x

    def train(self, do_training=True):
        self.training_mode = do_training

    # ------------------------------------------------------------------ configuration
    def load_config(self, config_paths):
        """cvvdp_metric.py:146-229.  Only the shipped model family is implemented by the kernels."""
        self.parameters_file = config_files.find("cvvdp_parameters.json", config_paths)
        self._config_paths = list(config_paths)
        self._set_parameters(json2dict(self.parameters_file))

    @classmethod
    def _validate_parameters(cls, p):
        """Everything _set_parameters / _make_handle would stumble over, checked BEFORE any state changes."""
        if p["masking_model"] != "mult-mutual" or p["contrast"] != "weber_g1" or p["dclamp_type"] != "soft" \
                or p["csf"] != "weber_fixed_size" or p["xchannel_masking"] != "on" or "block_channels" in p \
                or p.get("temp_filter", "default") != "default" or "ch_chrom_w" not in p or "mask_q" not in p:
            raise RuntimeError("Only the ColorVideoVDP v0.5.x base model is implemented by the HIP kernels "
                               "(masking 'mult-mutual', contrast 'weber_g1', soft clamp, csf 'weber_fixed_size', cross-channel masking on)")
        if p["beta"] != 2:
            raise RuntimeError("The fused band kernel implements the spatial p-norm for beta=2 only")
        if p["pu_dilate"] not in (0, 3):
            raise RuntimeError("pu_dilate must be 0 or 3")
        for name, n in cls._LIST_LENGTHS.items():
            if name in p and (not isinstance(p[name], (list, tuple)) or len(p[name]) != n):
                raise RuntimeError(f"parameter '{name}' must be a list of {n} numbers")
        for name in ("mask_p", "mask_c", "d_max", "sensitivity_correction", "beta_t", "beta_tch", "beta_sch", "jod_a", "jod_exp", "image_int",
                     "ch_chrom_w", "ch_trans_w"):
            if isinstance(p.get(name), (str, bool)) or not np.isfinite(float(p[name])):
                raise RuntimeError(f"parameter '{name}' must be a finite number")

    def _set_parameters(self, p):
        """Second half of the reference's load_config: the parameter dictionary becomes the metric's state and the core's handle is
        re-made from it.  `self.parameters` is what the kernels are configured from; its numeric entries are also tensor attributes of
        the same name (`metric.mask_c`, ...: `__getattr__` / `__setattr__`), the way the reference keeps them (cvvdp_metric.py:153-219).
        All or nothing: a dictionary that does not validate, or from which no handle can be made, leaves the metric as it was."""
        self._validate_parameters(p)
        d = self.__dict__
        keys = ("parameters", "version", "pu_dilate", "csf_table", "_jod_a", "_jod_exp", "_baseband_weight", "_ch_w")
        prev = {k: d[k] for k in keys if k in d}
        try:
            d["parameters"] = p
            d["version"] = p["version"]
            d["pu_dilate"] = p["pu_dilate"]
            d["csf_table"] = hs.CsfTable(load_config("csf_lut_weber_fixed_size.json", self._config_paths))
            d["_jod_a"], d["_jod_exp"] = f32(p["jod_a"]), f32(p["jod_exp"])
            d["_baseband_weight"] = np.asarray(p["baseband_weight"], dtype=f32)
            d["_ch_w"] = np.asarray([1.0, p["ch_chrom_w"], p["ch_chrom_w"], p["ch_trans_w"]], dtype=f32)
            self._make_handle()
        except Exception:
            for k in keys:
                d.pop(k, None)
            d.update(prev)
            if "parameters" in prev:
                self._make_handle()
            raise

    @property
    def ch_w(self):
        """Per-channel weights [1, ch_chrom_w, ch_chrom_w, ch_trans_w] (cvvdp_metric.py:604-607 builds them per call), as a tensor."""
        return torch.as_tensor(self._ch_w, device=self._param_device())

    def _make_handle(self):
        self._cfg_version = getattr(self, "_cfg_version", 0) + 1
        self._clip_cache = None
        if not hasattr(self, "parameters") or not hasattr(self, "display_photometry"):
            return
        p, dm = self.parameters, self.display_photometry
        if not isinstance(dm, vvdp_display_photo_eotf):
            raise RuntimeError("display_photometry must be a vvdp_display_photo_eotf")
        P = _capi.Params()
        P.eotf, P.gamma = dm.eotf_params()
        Yb, Yr = dm.get_black_level()
        P.Y_peak, P.Y_black, P.Y_refl, P.exposure = dm.Y_peak, Yb, Yr, dm.exposure
        P.rgb2dkl[:] = dm.rgb2dkl_fp32().reshape(-1).tolist()
        P.mask_p = p["mask_p"]
        P.mask_c10 = float(np.power(f32(10.0), f32(p["mask_c"])))
        P.mask_q[:] = p["mask_q"]
        P.xcm[:] = np.power(f32(2.0), np.asarray(p["xcm_weights"], dtype=f32)).tolist()
        P.ch_gain[:] = [1.0, 1.45, 1.0, 1.0]
        P.d_max10 = float(np.power(f32(10.0), f32(p["d_max"])))
        P.sens_mul = float(np.power(f32(10.0), f32(p["sensitivity_correction"]) / f32(20.0)))
        P.blur_radius = int(self.pu_dilate * 2)
        P.blur_taps[:] = hs.gaussian_taps(13, 3.0).tolist()
        P.beta, P.beta_t, P.beta_tch, P.beta_sch = p["beta"], p["beta_t"], p["beta_tch"], p["beta_sch"]
        P.jod_a, P.jod_exp, P.image_int = p["jod_a"], p["jod_exp"], p["image_int"]
        P.ch_w[:] = self._ch_w.tolist()
        P.baseband_weight[:] = p["baseband_weight"]
        P.csf_logL_first, P.csf_logL_last = float(self.csf_table.log_L[0]), float(self.csf_table.log_L[-1])
        lib = _capi.lib()
        if self._handle.value:
            lib.cvvdp_destroy(self._handle)
            self._handle = ctypes.c_void_p()
        rc = lib.cvvdp_create(ctypes.byref(P), ctypes.byref(self._handle))
        if rc != 0:
            raise RuntimeError(f"cvvdp_create failed ({rc})")
        self._params = P

    def update_from_checkpoint(self, ckpt):
        """cvvdp_metric.py:231-243: take the calibrated parameters out of a training checkpoint -- every `params.<name>` entry of its
        `state_dict` replaces the parameter <name>; other entries are skipped.  Here the values go into `self.parameters` and the core's handle is re-made, so the next predict() runs with them."""
        import os
        assert os.path.isfile(ckpt), f"Calibrated PyTorch checkpoint not found at: {ckpt}"
        prefix = "params."
        p = dict(self.parameters)
        extra = {}
        for key, value in torch.load(ckpt, map_location=torch.device("cpu"))["state_dict"].items():
            if not key.startswith(prefix):
                continue
            name = key[len(prefix):]
            if isinstance(p.get(name), (str, bool)):
                raise RuntimeError(f"checkpoint entry '{key}' would replace the non-numeric parameter '{name}'")
            if name in p:
                p[name] = self._parameter_value(name, value, p[name])      # floats stay floats; wrong-length lists raise here
            else:
                v = value.detach().to(torch.float32) if torch.is_tensor(value) else torch.as_tensor(value, dtype=torch.float32)
                extra[name] = v.to(self._param_device())   # the reference sets ANY attribute; ones the model does not read stay attributes
        self._set_parameters(p)
        for name, v in extra.items():
            setattr(self, name, v)

    def save_to_config(self, fname, comment):
        """cvvdp_metric.py:1129-1154: write the current parameters in the layout of the parameter file they were loaded from (strings
        and integers as loaded, floats and lists from the metric's state), with `__comment` and today's `calibration_date`."""
        from datetime import date
        assert fname.endswith(".json"), "Please provide a .json file"
        parameters = dict(json2dict(self.parameters_file))
        for key in parameters:
            if isinstance(parameters[key], (str, int)):
                continue
            elif isinstance(parameters[key], float):
                # the reference holds its trained scalars as fp32 tensors (their .item() is written) and `bfilt_duration` as the plain number
                parameters[key] = float(self.parameters[key]) if key == "bfilt_duration" else float(np.float64(f32(self.parameters[key])))
            elif isinstance(parameters[key], list):
                parameters[key] = [float(x) for x in np.asarray(self.parameters[key], dtype=f32).astype(np.float64)]
        parameters["__comment"] = comment
        parameters["calibration_date"] = date.today().strftime("%d/%m/%Y")
        with open(fname, "w") as f:
            json.dump(parameters, f, indent=4)

    def set_display_model(self, display_name="standard_4k", display_photometry=None, display_geometry=None, config_paths=[]):
        """cvvdp_metric.py:246-264."""
        if display_photometry is None:
            self.display_photometry = vvdp_display_photometry.load(display_name, config_paths)
            self.display_name = display_name
        else:
            self.display_photometry = display_photometry
            self.display_name = getattr(display_photometry, "short_name", "unspecified")
        if display_geometry is None:
            self.display_geometry = vvdp_display_geometry.load(display_name, config_paths)
        else:
            self.display_geometry = display_geometry
        self.pix_per_deg = self.display_geometry.get_ppd()
        self._make_handle()

    def set_frame_sharding(self, group="world"):
        """Score only this rank's frame range (plus a temporal halo) and combine Q_per_ch over the
        process group with one RCCL all-gather.  `group=None` turns sharding off.  Frame features depend only
        on the frame and its filter_len-1 predecessors (cvvdp_metric.py:554-560), so the result equals the
        single-GPU one up to fp32 rounding of nothing at all: the per-frame values are bit-identical."""
        self._shard = group

    # ------------------------------------------------------------------ public API
    def predict(self, test_cont, reference_cont, dim_order="BCFHW", frames_per_second=0):
        vs = video_source_array(test_cont, reference_cont, frames_per_second, dim_order=dim_order,
                                display_photometry=self.display_photometry)
        return self.predict_video_source(vs)

    def loss(self, test_cont, reference_cont, dim_order="BCFHW", frames_per_second=0):
        for a in (test_cont, reference_cont):
            if torch.is_tensor(a) and a.requires_grad:
                raise vq_exception("colorvideovdp_amd is inference-only: loss() does not propagate gradients")
        Q_jod, _ = self.predict(test_cont, reference_cont, dim_order=dim_order, frames_per_second=frames_per_second)
        return 10.0 - Q_jod

    def predict_video_source(self, vid_source, heatmap_sink=None):
        """cvvdp_metric.py:304-441.

        heatmap_sink (extension): callable(first_frame, frames) that receives the heat map block by block
        (`frames`: float16 CPU tensor [1, 1|3, n, H, W], valid only during the call) instead of `stats["heatmap"]`
        holding the whole clip -- an 8K x 256-frame colour heat map is 51 GB.  A sink with an attribute `wants_uint8 = True`
        receives the frames as its file format needs them: uint8 [n, H, W, 1|3], converted on the GPU exactly as the reference's
        writers convert the fp16 map (run_cvvdp.py:62-78).  A host sink is called piece by piece IN ORDER from one worker thread of this
        call (never concurrently with itself), so that a slow writer does not hold up the kernels and copies of the pieces behind it;
        predict_video_source() returns after the last call, and an exception of the sink is raised from it.  See
        colorvideovdp_amd.heatmap_writers."""
        inner = getattr(vid_source, "vs", None)             # video_source_file wraps the source that does the work (video_source_file.py:755-820)
        if isinstance(inner, video_source):
            vid_source = inner
        height, width, N_frames = vid_source.get_video_size()
        batch_sz = vid_source.get_batch_size()
        if batch_sz > 1 and self.do_heatmap:
            raise vq_exception("Heatmaps not supported when batches are used")
        if heatmap_sink is not None and not self.do_heatmap:
            raise vq_exception("heatmap_sink given, but the metric was created without a heat map")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device available: colorvideovdp_amd has no CPU path")
        is_image = N_frames == 1
        # a RAW source (arrays, .yuv files) that carries its own display photometry (video_source_dm, video_source.py:206-229)
        # is measured with it; the metric's display geometry still sets the pixels per degree.  Sources that hand out
        # finished DKL frames have applied their display model themselves: theirs is never needed here (and may be an
        # object of another package).
        src_dm = getattr(vid_source, "dm_photometry", None)
        if src_dm is not None and src_dm is not self.display_photometry and self._is_raw_source(vid_source):
            prev = self.display_photometry
            self.display_photometry = src_dm
            self._make_handle()
            try:
                return self._predict_video_source(vid_source, height, width, N_frames, is_image, heatmap_sink)
            finally:
                self.display_photometry = prev
                self._make_handle()
        return self._predict_video_source(vid_source, height, width, N_frames, is_image, heatmap_sink)

    def extract_features(self, vid_source):
        """Features for the ML heads (cvvdp_ml_base.extract_features, cvvdp_ml_metric.py:193-277): per band the
        pooled statistics (mean_T, var_T, mean_R, var_R, mean_D, var_D) of |T_f|*S, |R_f|*S and D over cells of
        ceil(pix_per_deg) pixels.  Returns (features, heatmap): features[band] is a float32 device tensor
        [B, F, H', W', channels, 6]; heatmap is None (the ML metrics produce none, :117-118)."""
        if self.do_heatmap:
            raise vq_exception("Currently cvvdp-ml metrics do not produce heatmaps")
        height, width, N_frames = vid_source.get_video_size()
        self._feature_out = []
        try:
            self._score_range(vid_source, 0, N_frames)
            feats = self._feature_out
        finally:
            self._feature_out = None
        B = vid_source.get_batch_size()
        # [F*B, H', W', C, 6] (item = frame * B + batch) -> [B, F, H', W', C, 6]
        return [f.view((N_frames, B) + tuple(f.shape[1:])).transpose(0, 1).contiguous() for f in feats], None

    @staticmethod
    def _is_raw_source(vs):
        return hasattr(vs, "get_raw_yuv_block") or hasattr(vs, "get_raw_block") or isinstance(vs, video_source_array)

    def _predict_video_source(self, vid_source, height, width, N_frames, is_image, heatmap_sink=None):
        if getattr(self, "_profile_per_call", False):      # (the handle may have been re-made since profile() was called)
            _capi.check(self._handle, _capi.lib().cvvdp_profile_enable(self._handle, 1), "cvvdp_profile_enable")
        first, count = 0, N_frames
        group = None
        sharded = False
        if self._shard is not None and not is_image and torch.distributed.is_available() and torch.distributed.is_initialized():
            group = None if self._shard == "world" else self._shard
            rank, world = torch.distributed.get_rank(group), torch.distributed.get_world_size(group)
            first, count = plan_frame_shard(N_frames, rank, world)
            sharded = True
        if count > 0:
            Q_local, heatmap, rho_band = self._score_range(vid_source, first, count, heatmap_sink)
        else:
            # more ranks than frames: this rank has nothing to score, but still joins the gather with an empty shard
            pyr_height, freqs = hs.band_frequencies(width, height, self.pix_per_deg)
            rho_band = freqs.copy()
            rho_band[pyr_height] = 0.1
            Q_local = torch.zeros((vid_source.get_batch_size(), 4, 0, pyr_height + 1), dtype=torch.float32, device=self.device)
            heatmap = torch.empty((1, 1 if self.heatmap == "raw" else 3, 0, height, width), dtype=torch.float16) if self.do_heatmap else None
        Q_per_ch = all_gather_frames(Q_local, N_frames, group) if sharded else Q_local
        Q_jod = self.do_pooling_and_jods(Q_per_ch)
        stats = {}
        stats["Q_per_ch"] = Q_per_ch.detach().cpu().numpy()
        stats["rho_band"] = rho_band
        stats["frames_per_second"] = vid_source.get_frames_per_second()
        stats["width"] = width
        stats["height"] = height
        stats["N_frames"] = N_frames
        if self.do_heatmap:
            if heatmap_sink is None:
                stats["heatmap"] = heatmap
            if count != N_frames:
                stats["heatmap_frame_range"] = (first, first + count)
        if getattr(self, "_profile_per_call", False):
            stats["kernel_ms"] = {name: ms for name, (ms, _n) in self.profile_read().items()}
        return (Q_jod.squeeze(), stats)

    # ------------------------------------------------------------------ block planning
    def _pick_block_frames(self, pix, batch, n_frames, fl, nch, host_resident=False, device_raw=False):
        """Frames per process_block call.  Results do not depend on it (tested).  Larger blocks amortise the
        (fl-1)-frame DKL tail that is written/re-read between blocks (cf. estimate_block_N,
        cvvdp_metric.py:565-594, for the memory model; CVVDP_PIPELINE=1 software-pipelines the blocks on a
        second stream and needs two pyramid sets, which no longer pays off)."""
        if self.block_frames is not None:
            nb = int(self.block_frames)
        else:
            free, _total = torch.cuda.mem_get_info(self.device)
            # ... plus what torch's caching allocator holds without using it: this process gets those blocks back before the driver is
            # asked (after a large clip the device looks full to mem_get_info while nearly all of it is cached and free)
            try:
                free += max(0, torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device))
            except Exception:
                pass
            budget = free * 0.55
            if self.gpu_mem is not None:
                budget = min(budget, self.gpu_mem * 1e9)
            per_frame = pix * batch * (2 * nch * 4 * 1.34) + (pix * 16 if self.do_heatmap else 0)
            fixed = pix * batch * 24 * (fl - 1)
            nb = int((budget - fixed) // per_frame)
            # heat maps leave the GPU over PCIe (2-6 B/pixel): 16-frame blocks let the copy of a block overlap the
            # kernels of the next one (4K supra-threshold, 64 frames: 111 ms as one block, 81 ms in blocks of 16)
            # Raw clips resident in HBM hand every block its fl-1 predecessor frames again (cvvdp_clip.raw_halo): 16 extra frames of
            # unpack + display model per 64-frame block at 60 fps.  Nothing is staged for them, so their blocks may be as long as the
            # core's window allows (4K x 256 uint8: 61.0 ms in blocks of 64, 56.5 in one block of 240 + one of 16).  Which kernels
            # score the clip does not depend on the block (the core decides from a nominal 64-frame block).
            long_ok = device_raw and not self.do_heatmap and getattr(self, "_feature_out", None) is None
            if long_ok and nb > 64:
                # Blocks beyond 64 frames are worth ~7 % and cost up to 4x the workspace: they are sized from an ABSOLUTE share of the
                # device (30 % of its total memory: 240 frames of 4K on a 288 GB part), not from what happens to be free at call
                # time -- several processes sharing one GPU would each see the same free memory and run out together.  A workspace
                # that cannot be had falls back to 64-frame blocks and below (_score_range).
                nb_abs = int((_total * 0.30 - fixed) // per_frame)
                nb = max(64, min(nb, nb_abs))
            # Heat-map clips resident in HBM: the temporal stage runs over a long block and the band / heat-map stage walks it in
            # 16-frame pieces (cvvdp_clip.defer_bands), so the copy of a piece still overlaps the kernels of the next one while the
            # fl-1 halo frames are unpacked once per long block, not once per 16 frames (8K PQ x 256: temporal stage 94 -> ~60 ms).
            # Level 0 of the long block is what costs memory (8 planes: 1 GB per 8K frame): sized from an absolute share of the device.
            pieces_ok = device_raw and self.do_heatmap and getattr(self, "_feature_out", None) is None
            if pieces_ok:
                piece = self._piece_frames()
                fixed_p = fixed + piece * (pix * batch * (2 * nch * 4 * 0.34) + pix * 16) + piece * 8192     # (+ the range / tone-curve words of a piece)
                nb_abs = int((_total * 0.30 - fixed_p) // (pix * batch * 2 * nch * 4))
                nb_long = min(int((budget - fixed_p) // (pix * batch * 2 * nch * 4)), nb_abs, 64)
                # a long block pays only when it is longer than a piece; otherwise (a small gpu_mem cap, little free memory) the plain
                # 16-frame-block rule applies, so that the budget stays a hard cap (it used to be clamped UP to the piece)
                nb = nb_long if nb_long > piece else min(nb, 16)
            else:
                nb = min(nb, 16 if self.do_heatmap else (_capi.MAX_WINDOW - fl + 1 if long_ok else 64))
            if host_resident and n_frames > 24:
                nb = min(nb, 16)   # the upload of block k+1 (side stream, worker thread) hides behind the kernels of block k
        return max(1, min(nb, n_frames, _capi.MAX_WINDOW - fl + 1))

    def _piece_frames(self):
        """Frames per band / heat-map piece of a long temporal block.  16 lets the copy of a piece to the host overlap the kernels of the
        next one; a sink that consumes the frames on the GPU (`wants_device`) has no copy to hide and takes 32, which fill the GPU better
        at the small pyramid levels (8K x 128: 25.9 -> 26.7 Gpixel/s; heat maps and scores do not depend on the piece length, bit for bit)."""
        if self.score_frames is not None:
            return max(1, int(self.score_frames))
        return 32 if getattr(self, "_sink_on_device", False) else 16

    def _copy_stream(self):
        """A stream for copies across PCIe, on a hardware queue of its own.  HIP maps a process's streams onto a handful of hardware queues
        per priority level (four by default), and streams that share one run in order: the heat map's D2H stream shared its queue with one
        of the core's helper streams, so a piece's border-strip kernels sat behind the PREVIOUS piece's 28-ms copy and the link idled 4 ms
        after every copy (8K x 256: 77 of 550 ms, profiles/r06_d2h_events_timeline.txt).  High-priority streams come from another
        pool of queues; a copy has no use for the priority itself."""
        try:
            return torch.cuda.Stream(self.device, priority=-1)
        except Exception:
            return torch.cuda.Stream(self.device)

    def _alloc_workspace(self, nbytes):
        """The one device allocation of a call (torch's caching allocator; the core allocates nothing itself)."""
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def _raw_block(self, vs, a, b):
        """Frames [a,b) of test and reference as BCFHW tensors on the device + dtype code."""
        if hasattr(vs, "get_raw_block"):
            return vs.get_raw_block(a, b, self.device)
        if isinstance(vs, video_source_array):
            t, r, code = vs.raw_arrays()
            return self._upload_frames(t, a, b), self._upload_frames(r, a, b), code
        # generic video_source: frames arrive one by one, already in DKL (cvvdp_metric.py:503-504)
        return self._seq_frames(vs).block(a, b)

    def _yuv_block_resized(self, vs, a, b, height, width):
        """Frames [a,b) of a .yuv pair with full_screen_resize: [1,3,n,H,W] fp32 R'G'B' blocks at the display's resolution
        (cvvdp_unpack_yuv_resized: unpack + torch.nn.functional.interpolate semantics + clip, video_source_yuv.py:333-336)."""
        lib = _capi.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        out = []
        for side in range(2):
            codes, fmt, sw, sh = vs.get_raw_yuv_side(side, a, b, self.device)
            # a side that already has the target size is not interpolated by the reference: nearest at scale 1 is the identity
            mode = _capi.RESIZE_MODES[vs.full_screen_resize] if (sw, sh) != (width, height) else _capi.RESIZE_MODES["nearest"]
            tmp = torch.empty(3 * (b - a) * sh * sw, dtype=torch.float32, device=self.device)
            rgb = torch.empty((1, 3, b - a, height, width), dtype=torch.float32, device=self.device)
            rc = lib.cvvdp_unpack_yuv_resized(self._handle, codes.data_ptr(), ctypes.byref(fmt), side, sw, sh, b - a, width, height, mode,
                                              tmp.data_ptr(), rgb.data_ptr(), stream)
            _capi.check(self._handle, rc, "cvvdp_unpack_yuv_resized")
            out.append(rgb)
            del tmp, codes            # stream-ordered: the caching allocator may reuse them once the kernels are queued
        return out[0], out[1], _capi.F32

    def _seq_frames(self, vs, colorspace="DKLd65"):
        sf = getattr(self, "_seq", None)
        if sf is None or sf.vs is not vs or sf.colorspace != colorspace:
            sf = self._seq = _SequentialFrames(vs, self.device, colorspace)
        return sf

    def _upload_frames(self, x, a, b):
        """Frames [a,b) of a BCFHW tensor on the device.  Host tensors are copied plane by plane: x[b, c, a:b] is
        contiguous, the [B,C,a:b] slice of a longer clip is not, and a strided host-to-device copy first gathers on
        the CPU (several times slower than the link)."""
        if x.device == self.device:
            return x[:, :, a:b]
        if x.device.type != "cpu" or (a == 0 and b == x.shape[2]) or not x.is_contiguous():
            return x[:, :, a:b].to(self.device, non_blocking=True)
        out = torch.empty((x.shape[0], x.shape[1], b - a) + tuple(x.shape[3:]), dtype=x.dtype, device=self.device)
        for i in range(x.shape[0]):
            for c in range(x.shape[1]):
                out[i, c].copy_(x[i, c, a:b], non_blocking=True)
        return out

    @staticmethod
    def _host_resident(vs):
        """Does the clip have to cross PCIe block by block (host arrays, .yuv files)?"""
        if hasattr(vs, "get_raw_yuv_block"):
            return not getattr(vs, "device_resident", False)
        if isinstance(vs, video_source_array):
            return vs.raw_arrays()[0].device.type == "cpu"
        if hasattr(vs, "get_raw_block"):                   # e.g. image-frame files; resident clips are the default
            return not getattr(vs, "device_resident", True)
        return False

    @staticmethod
    def _strides(t, r):
        B = max(t.shape[0], r.shape[0])
        st, sr = list(t.stride()), list(r.stride())
        if t.shape[0] == 1 and B > 1:
            st[0] = 0  # broadcast batch (video_source.py:247-252)
        if r.shape[0] == 1 and B > 1:
            sr[0] = 0
        return (ctypes.c_int64 * 5)(*st), (ctypes.c_int64 * 5)(*sr)

    def _score_range(self, vs, first, count, heatmap_sink=None):
        """Q_per_ch [B, C, count, bands] (device tensor) of frames [first, first+count)."""
        lib = _capi.lib()
        self._sink_on_device = bool(getattr(heatmap_sink, "wants_device", False))
        height, width, N_total = vs.get_video_size()
        B = vs.get_batch_size()
        is_image = N_total == 1
        nch = 3 if is_image else 4
        pyr_height, freqs = hs.band_frequencies(width, height, self.pix_per_deg)
        L = pyr_height + 1
        rho_band = freqs.copy()
        rho_band[L - 1] = 0.1  # cvvdp_metric.py:685-686
        is_yuv = hasattr(vs, "get_raw_yuv_block")      # planar Y'CbCr file source: unpacked by the temporal kernel
        # full_screen_resize (video_source_yuv.py:333-336): frames are unpacked to R'G'B' and resized on the GPU first, then
        # take the fp32 route; a source whose files already have the target size takes the fused route like the reference
        yuv_resized = is_yuv and bool(getattr(vs, "needs_resize", lambda: False)())
        generic = not self._is_raw_source(vs)           # frames arrive one by one, already in DKL
        # sources with temporally pre-filtered channels bypass the sliding window + FIR (cvvdp_metric.py:470-488)
        prefiltered = bool(getattr(vs, "is_temporally_filtered", False)) and not is_image
        if prefiltered and not generic:
            raise vq_exception("is_temporally_filtered is only meaningful for sources that deliver 'DKLd65_trans' frames")
        self._seq = None
        if is_yuv:
            if is_image:
                raise vq_exception("single-frame .yuv clips are not supported")
            C = 3
        elif generic and not is_image:
            C, code = 3, _capi.F32_DKL                  # no probe: a file source may only be read once, in order
        else:
            probe_t, probe_r, code = self._raw_block(vs, first, first + 1)
            C = probe_t.shape[1]
        # The clip description (temporal taps, CSF rows per band, block size) depends only on the geometry: repeated
        # calls on clips of the same shape reuse it, so the first kernel is not held back by ~0.4 ms of host set-up.
        key = (height, width, N_total, first, count, B, C, is_image, None if is_image else float(vs.get_frames_per_second()), self.heatmap,
               bool(self.debug_dump), int(self.fuse_mode), int(self.band_layout), self.score_frames, self._sink_on_device, self.block_frames, self.gpu_mem, float(self.pix_per_deg), self._cfg_version, prefiltered, getattr(self, "_feature_out", None) is not None,
               self._host_resident(vs))
        cached = getattr(self, "_clip_cache", None)
        if cached is not None and cached[0] == key:
            clip, fl, F = cached[1], cached[2], cached[3]
            if not is_image:
                self.F, self.filter_len, self.last_block_frames = F, fl, clip.block_frames
        else:
            clip = _capi.Clip()
            clip.batch, clip.channels, clip.height, clip.width = B, C, height, width
            clip.is_video, clip.n_frames, clip.n_levels = int(not is_image), count, L
            clip.first_frame = first
            clip.total_frames = N_total
            clip.heatmap = _capi.HEATMAP[self.heatmap]
            clip.debug_dump = int(self.debug_dump)
            clip.fuse_mode = int(self.fuse_mode)
            clip.band_layout = int(self.band_layout)
            clip.feature_size = int(math.ceil(self.pix_per_deg)) if getattr(self, "_feature_out", None) is not None else 0   # cvvdp_ml_metric.py:353
            # device-resident clips: later blocks re-read their fl-1 predecessor frames (like a shard's halo) instead of
            # a DKL tail written by the previous block: 3.2 GB less HBM traffic per 64-frame 4K block
            clip.raw_halo = int(not is_image and not self._host_resident(vs) and (is_yuv or isinstance(vs, video_source_array) or hasattr(vs, "get_raw_block")))
            fl, F = 1, None
            if not is_image:
                F = hs.temporal_filters(vs.get_frames_per_second(), self.parameters["beta_tf"], self.parameters["sigma_tf"])
                self.F = F
                fl = 1 if prefiltered else F.shape[1]     # pre-filtered frames need no window
                if fl > _capi.MAX_FILTER_LEN:
                    raise vq_exception(f"frame rates above {(_capi.MAX_FILTER_LEN - 1) * 4} fps are not supported")
                self.filter_len = fl
                taps = np.zeros((4, _capi.MAX_FILTER_LEN), dtype=f32)
                taps[:, :fl] = F[:, :fl]
                clip.taps[:] = taps.reshape(-1).tolist()
                nb = self._pick_block_frames(height * width, B, count, fl, nch, self._host_resident(vs), bool(clip.raw_halo))
                clip.filter_len, clip.block_frames = fl, nb
                self.last_block_frames = nb
                # heat-map clips resident in HBM are scored in pieces of a long temporal block (see _pick_block_frames)
                piece = self._piece_frames()
                if self.do_heatmap and clip.raw_halo and not prefiltered and not self.debug_dump and nb > piece:
                    clip.defer_bands, clip.score_frames = 1, piece
            rows = np.zeros((_capi.MAX_LEVELS, 4, _capi.CSF_NODES), dtype=f32)
            for bb in range(L):
                rows[bb] = self.csf_table.rows(rho_band[bb])
            clip.csf_rows[:] = rows.reshape(-1).tolist()
            self._clip_cache = (key, clip, fl, F)
        while True:
            _capi.check(self._handle, lib.cvvdp_configure(self._handle, ctypes.byref(clip)), "cvvdp_configure")
            need = lib.cvvdp_workspace_bytes(self._handle)
            if self._ws is not None and self._ws.numel() >= need and self._ws.device == self.device:
                break
            self._ws = None
            try:
                self._ws = self._alloc_workspace(need)
                break
            except torch.cuda.OutOfMemoryError:
                # The block length was sized from the memory that was free a moment ago; somebody else (another rank on this GPU)
                # may have taken it since.  Results do not depend on the block length: retry with shorter blocks (.., 64, 32, 16, ..).
                nb_now = int(clip.block_frames)
                if is_image or self.block_frames is not None or nb_now <= 1:
                    raise
                torch.cuda.empty_cache()
                clip.block_frames = 64 if nb_now > 64 else max(1, nb_now // 2)
                self.last_block_frames = int(clip.block_frames)
                if clip.defer_bands and clip.block_frames <= clip.score_frames:
                    clip.defer_bands, clip.score_frames = 0, 0
        _capi.check(self._handle, lib.cvvdp_bind_workspace(self._handle, self._ws.data_ptr(), self._ws.numel()), "cvvdp_bind_workspace")
        self.last_score_frames = int(clip.score_frames) if clip.defer_bands else 0      # (0: every block scored whole)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        heatmap = None
        hm_ch = 1 if self.heatmap == "raw" else 3
        copy_stream = None
        stage, pending_sink, stage_next = None, [], [0]
        if self.do_heatmap and heatmap_sink is not None:
            # Streaming: a ring of page-locked staging buffers of one piece each; piece k is handed to the sink while
            # the kernels of the pieces after it run.  Host memory is bounded by `heatmap_stage_buffers` pieces whatever the clip length.
            # Four buffers since round 6 (two before): the kernels of an 8K piece take 21 ms and its 8-bit frames 31 ms of the PCIe link,
            # so the link is the bound -- and with two buffers it ran dry at every temporal block boundary, where the next piece is
            # 38 ms away (the block's temporal kernel + the piece's bands) while only one piece was queued: 8K x 256 took 564 ms for 490 ms
            # of copying.  With a backlog of up to three pieces the copy engine always has work.
            # A sink that writes 8-bit frames anyway (PNG, ffmpeg) sets `wants_uint8`: the conversion the reference's writers do on
            # the host is then done by the heat-map kernel, and 3 instead of 6 bytes per pixel cross PCIe
            nb_max = 1 if is_image else (clip.score_frames if clip.defer_bands else clip.block_frames)
            sink_u8 = bool(getattr(heatmap_sink, "wants_uint8", False))
            # A sink that consumes the frames ON THE GPU (statistics, a device-side encoder, a later bulk copy) sets `wants_device`:
            # it is called with the device tensor of each piece, on the current stream, and nothing crosses PCIe here.  (8K x 256
            # frames of 8-bit RGB are 25.5 GB: 0.49 s of a PCIe 5 x16 link at the 52 GB/s it sustains -- more than all the kernels.)
            sink_dev = bool(getattr(heatmap_sink, "wants_device", False))
            stage = getattr(self, "_hm_stage", None)
            if sink_dev:
                stage = "device"
            else:
                n_stage = max(2, int(getattr(self, "heatmap_stage_buffers", 4)))
                need = hm_ch * nb_max * height * width * (1 if sink_u8 else 2)                               # bytes per piece
                if stage is None or stage == "device" or len(stage) != n_stage or stage[0].numel() < need:
                    self._hm_stage = None                                                                   # (unpin the old ring first)
                    stage = self._hm_stage = [torch.empty(need, dtype=torch.uint8, device="cpu", pin_memory=True) for _ in range(n_stage)]
            if not sink_dev:
                copy_stream = getattr(self, "_hm_stream", None) or self._copy_stream()
                self._hm_stream = copy_stream
        elif self.do_heatmap:
            # The reference keeps the whole fp16 heat map on the CPU (cvvdp_metric.py:344).  Page-locked memory
            # + copies on a side stream keep the D2H traffic (6 B/pixel) off the compute stream.
            # Page-locking gigabytes costs more than filling them (~60 ms per GB), so the buffer of the previous call is
            # handed out again when nobody holds a tensor or array on it any more (callers get views: a live view shows
            # up in the storage's use count).
            shape = [1, hm_ch, count, height, width]
            base = getattr(self, "_hm_base", None)
            if base is not None and base.numel() == int(np.prod(shape)) and _storage_is_unshared(base):
                heatmap = base.view(shape)
                copy_stream = self._hm_stream
            else:
                self._hm_base = None
                try:
                    base = torch.empty(int(np.prod(shape)), dtype=torch.float16, device="cpu", pin_memory=True)   # every frame is written below
                    copy_stream = getattr(self, "_hm_stream", None) or self._copy_stream()
                    self._hm_base, self._hm_stream = base, copy_stream
                    heatmap = base.view(shape)
                except RuntimeError:
                    heatmap = torch.empty(shape, dtype=torch.float16, device="cpu")

        # Host sinks are fed from ONE worker thread, piece by piece in order (round 6): the thread waits for the piece's copy and calls the
        # sink, while this thread goes on queueing kernels and copies.  A sink that takes as long as a piece's copy (a PNG / ffmpeg
        # writer; even bench.py's frame means, whose strided reads of 1.6 GB cost 30 ms per 8K piece) used to sit between two
        # pieces' launches: the D2H stream of configs[4] idled 77 of 550 ms (profiles/r06_d2h_events_timeline.txt).  A staging
        # buffer is reused only after the sink has returned from the piece that used it; a sink's exception is re-raised here.
        sink_pool = [None]

        def _feed_sink(ev, frame0, view):
            torch.cuda.set_device(self.device)
            ev.synchronize()
            heatmap_sink(frame0, view)

        def flush_sink(keep=0):
            while len(pending_sink) > keep:
                pending_sink.pop(0)[0].result()            # (the piece has been consumed; raises what the sink raised)

        def fetch_heatmap(ff, n):
            import time as _time
            host_t = [_time.perf_counter()]
            if stage is not None and sink_u8:
                buf = torch.empty((n, height, width, hm_ch), dtype=torch.uint8, device=self.device)
                host_t.append(_time.perf_counter())
                _capi.check(self._handle, lib.cvvdp_get_heatmap_rgb8(self._handle, n, buf.data_ptr(), stream), "cvvdp_get_heatmap_rgb8")
                host_t.append(_time.perf_counter())
            else:
                buf = torch.empty((hm_ch, n, height, width), dtype=torch.float16, device=self.device)
                _capi.check(self._handle, lib.cvvdp_get_heatmap(self._handle, n, buf.data_ptr(), stream), "cvvdp_get_heatmap")
            if stage == "device":
                heatmap_sink(first + ff, buf if sink_u8 else buf.view(1, hm_ch, n, height, width))
                return
            if stage is not None:
                flush_sink(keep=len(stage) - 1)            # the buffer about to be overwritten has been consumed
                host_t.append(_time.perf_counter())
                slot = stage_next[0]
                stage_next[0] = (slot + 1) % len(stage)
                dst = stage[slot]
                if sink_u8:
                    view = dst[:hm_ch * n * height * width].view(n, height, width, hm_ch)
                else:
                    view = dst[:2 * hm_ch * n * height * width].view(torch.float16).view(1, hm_ch, n, height, width)
                trace = getattr(self, "d2h_trace", None)          # tools/d2h_events_timeline.py: HIP-event timeline of the D2H stream (a list to fill)
                if trace is not None:
                    ev_k = torch.cuda.Event(enable_timing=True)
                    ev_k.record(torch.cuda.current_stream(self.device))       # the piece's kernels are done here
                copy_stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(copy_stream):
                    if trace is not None:
                        ev_a = torch.cuda.Event(enable_timing=True)
                        ev_a.record(copy_stream)
                    (view if sink_u8 else view[0]).copy_(buf, non_blocking=True)
                    ev = torch.cuda.Event(enable_timing=trace is not None)
                    ev.record(copy_stream)
                if trace is not None:
                    host_t.append(_time.perf_counter())
                    trace.append((first + ff, n, view.numel() * view.element_size(), ev_k, ev_a, ev, host_t))
                buf.record_stream(copy_stream)
                if sink_pool[0] is None:
                    import concurrent.futures
                    sink_pool[0] = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="cvvdp-heatmap-sink")
                pending_sink.append((sink_pool[0].submit(_feed_sink, ev, first + ff, view), first + ff, view, slot))
                return
            # one copy per colour plane: heatmap[0, ch, ff:ff+n] is contiguous on the host, the [3, n, H, W] slice of a
            # longer clip is not (a strided D2H copy falls off the DMA path: 6x slower end to end)
            if copy_stream is None:
                for ch in range(hm_ch):
                    heatmap[0, ch, ff:ff + n] = buf[ch].cpu()
            else:
                copy_stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(copy_stream):
                    for ch in range(hm_ch):
                        heatmap[0, ch, ff:ff + n].copy_(buf[ch], non_blocking=True)
                buf.record_stream(copy_stream)

        feat_out = getattr(self, "_feature_out", None)
        if feat_out is not None:
            fs = int(clip.feature_size)
            lvl = [(height, width)]
            for _ in range(L - 1):
                lvl.append(((lvl[-1][0] + 1) // 2, (lvl[-1][1] + 1) // 2))
            feat_out[:] = [torch.empty((count * B, (h + fs - 1) // fs, (w + fs - 1) // fs, nch, 6), dtype=torch.float32, device=self.device)
                           for h, w in lvl]

        def fetch_features(ff, n):
            for bb in range(L):
                dst = feat_out[bb][ff * B:(ff + n) * B]
                _capi.check(self._handle, lib.cvvdp_get_features(self._handle, bb, n, dst.data_ptr(), stream), "cvvdp_get_features")

        if is_image:
            st, sr = self._strides(probe_t, probe_r)
            rc = lib.cvvdp_put_image(self._handle, probe_t.data_ptr(), probe_r.data_ptr(), code, st, sr, stream)
            _capi.check(self._handle, rc, "cvvdp_put_image")
            _capi.check(self._handle, lib.cvvdp_process_image(self._handle, stream), "cvvdp_process_image")
            if self.do_heatmap:
                fetch_heatmap(0, 1)
            if feat_out is not None:
                fetch_features(0, 1)
        else:
            nb = clip.block_frames

            def src_index(j):  # temporal padding before frame 0, cvvdp_metric.py:506-529
                if j >= 0:
                    return j
                return 0 if self.temp_padding == "replicate" else hs.symmetric_frame_index(j, N_total)

            if self.temp_padding not in ("replicate", "symmetric"):
                raise RuntimeError(f'Unknown padding method "{self.temp_padding}"')
            if prefiltered:
                seq = self._seq_frames(vs, "DKLd65_trans")
                for ff in range(first, first + count, nb):
                    n = min(nb, first + count - ff)
                    t, r, _ = seq.block(ff, ff + n, reference_first=True)
                    if t.shape[1] != 4:
                        raise vq_exception("a temporally filtered source must deliver 4-channel 'DKLd65_trans' frames")
                    st, sr = self._strides(t, r)
                    rc = lib.cvvdp_process_block_filtered(self._handle, t.data_ptr(), r.data_ptr(), st, sr, n, ff - first, stream)
                    _capi.check(self._handle, rc, "cvvdp_process_block_filtered")
                    if self.do_heatmap:
                        fetch_heatmap(ff - first, n)
                    if feat_out is not None:
                        fetch_features(ff - first, n)
                    del t, r
            blocks = []
            # Heat maps streamed to the host from a resident clip: the link is the bound, and its first byte can only leave after the first
            # temporal block's FIR + the first piece's bands -- so the first temporal block is ONE piece long (8K: the D2H stream starts
            # 13 ms earlier; results do not depend on the cut, bit for bit)
            head = int(clip.score_frames) if (isinstance(stage, list) and clip.defer_bands and clip.raw_halo and 0 < clip.score_frames < nb) else nb
            cuts, ff = [], first
            while not prefiltered and ff < first + count:
                n = min(head if ff == first else nb, first + count - ff)
                cuts.append((ff, n))
                ff += n
            for ff, n in cuts:
                if ff == first or clip.raw_halo:
                    # first block of the clip / shard, and every block of a device-resident clip: the fl-1 window
                    # positions before frame ff are real predecessor / halo frames or temporal padding; all of them
                    # are raw frames of the block handed in
                    hist_frames = [src_index(ff - (fl - 1) + k) for k in range(fl - 1)]
                    lo = min(hist_frames + [ff])
                    hi = max(hist_frames + [ff + n - 1]) + 1
                    hist = [f - lo for f in hist_frames]
                else:
                    # later blocks: the DKL tail of the previous block is still in the workspace
                    lo, hi = ff, ff + n
                    hist = [-1 - k for k in range(fl - 1)]
                blocks.append((ff, n, lo, hi, hist))

            def fetch(blk):
                _, _, lo, hi, _ = blk
                if yuv_resized:
                    return self._yuv_block_resized(vs, lo, hi, height, width)
                if is_yuv:
                    return vs.get_raw_yuv_block(lo, hi, self.device)
                return self._raw_block(vs, lo, hi)

            # Clips that live on the host cross PCIe one block ahead: a worker thread uploads block k+1 on a side stream
            # while this thread queues the kernels of block k (the copy blocks its thread, not the GPU).
            prefetch = len(blocks) > 1 and self._host_resident(vs)
            if prefetch:
                import concurrent.futures
                h2d = self._copy_stream()
                main_stream = torch.cuda.current_stream(self.device)

                dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

                def fetch_async(blk):
                    torch.cuda.set_device(dev_index)   # the current device is per thread
                    with torch.cuda.stream(h2d):
                        out = fetch(blk)
                        ev = torch.cuda.Event()
                        ev.record(h2d)
                    return out, ev

                pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
                pending = pool.submit(fetch_async, blocks[0])
            try:
                for i, blk in enumerate(blocks):
                    ff, n, lo, hi, hist = blk
                    if prefetch:
                        (t, r, third), ev = pending.result()
                        pending = pool.submit(fetch_async, blocks[i + 1]) if i + 1 < len(blocks) else None
                        main_stream.wait_event(ev)
                        t.record_stream(main_stream)
                        r.record_stream(main_stream)
                    else:
                        t, r, third = fetch(blk)
                    hist_c = (ctypes.c_int32 * max(len(hist), 1))(*hist)
                    if is_yuv and not yuv_resized:
                        rc = lib.cvvdp_process_block_yuv(self._handle, t.data_ptr(), r.data_ptr(), ctypes.byref(third), ff - lo, hist_c, n,
                                                         ff - first, stream)
                        _capi.check(self._handle, rc, "cvvdp_process_block_yuv")
                    else:
                        st, sr = self._strides(t, r)
                        rc = lib.cvvdp_process_block(self._handle, t.data_ptr(), r.data_ptr(), third, st, sr, ff - lo, hist_c, n, ff - first, stream)
                        _capi.check(self._handle, rc, "cvvdp_process_block")
                    if clip.defer_bands:
                        # the block is filtered; bands, pooling and heat maps piece by piece (the copy of a piece overlaps the next one's kernels)
                        for p0 in range(0, n, clip.score_frames):
                            m = min(clip.score_frames, n - p0)
                            _capi.check(self._handle, lib.cvvdp_score_frames(self._handle, p0, m, stream), "cvvdp_score_frames")
                            fetch_heatmap(ff - first + p0, m)
                    elif self.do_heatmap:
                        fetch_heatmap(ff - first, n)
                    if feat_out is not None:
                        fetch_features(ff - first, n)
                    del t, r  # stream-ordered: safe to release to the caching allocator once the kernels are queued
            finally:
                if prefetch:
                    pool.shutdown(wait=True)
        Q = torch.empty((B, nch, count, L), dtype=torch.float32, device=self.device)
        _capi.check(self._handle, lib.cvvdp_get_q_per_ch(self._handle, Q.data_ptr(), stream), "cvvdp_get_q_per_ch")
        self._seq = None
        if stage is not None:
            try:
                flush_sink()
            finally:
                if sink_pool[0] is not None:
                    sink_pool[0].shutdown(wait=True)
        elif copy_stream is not None:
            copy_stream.synchronize()   # the heat map is host data: it must be complete when predict() returns
        return Q, heatmap, rho_band

    # ------------------------------------------------------------------ pooling, info, outputs
    def do_pooling_and_jods(self, Q_per_ch):
        """cvvdp_metric.py:610-643 on the GPU.  Q_per_ch[batch, channel, frame, band] -> JOD[batch]."""
        Q = torch.as_tensor(Q_per_ch, dtype=torch.float32, device=self.device).contiguous()
        B, C, F, L = Q.shape
        jod = torch.empty((B,), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = _capi.lib().cvvdp_pool_jod(self._handle, Q.data_ptr(), B, C, F, L, jod.data_ptr(), stream)
        _capi.check(self._handle, rc, "cvvdp_pool_jod")
        return jod

    def get_ch_weights(self, no_channels):
        return self._ch_w[0:no_channels].reshape(1, -1, 1, 1)

    def met2jod(self, Q):
        """cvvdp_metric.py:646-658 (numpy, used by export_distogram)."""
        Q = np.asarray(Q, dtype=f32)
        Q_t = f32(0.1)
        a_p = self._jod_a * np.power(Q_t, self._jod_exp - f32(1.0))
        return np.where(Q <= Q_t, f32(10.0) - a_p * Q, f32(10.0) - self._jod_a * np.power(np.maximum(Q, Q_t), self._jod_exp)).astype(f32)

    def full_name(self):
        return "ColorVideoVDP"

    def short_name(self):
        return "cvvdp"

    def quality_unit(self):
        return "JOD"

    def get_info_string(self):
        if self.display_name.startswith("standard_"):
            standard_str = self.display_name
        else:
            standard_str = f"custom-display: {self.display_name}"
        L_black, L_refl = self.display_photometry.get_black_level()
        return f'"{self.full_name()} v{self.version}, {self.pix_per_deg:.4g} [pix/deg], ' \
               f"Lpeak={self.display_photometry.get_peak_luminance():.5g}, " \
               f'Lblack={L_black:.4g}, Lrefl={L_refl:.4g} [cd/m^2], ({standard_str})"'

    def write_features_to_json(self, stats, dest_fname):
        """cvvdp_metric.py:1112-1127."""
        Q_per_ch = stats["Q_per_ch"]
        fmap = {}
        for key, value in stats.items():
            if key not in ["Q_per_ch", "heatmap"]:
                fmap[key] = value.tolist() if isinstance(value, np.ndarray) else value
        for cc in range(Q_per_ch.shape[1]):
            for bb in range(Q_per_ch.shape[3]):
                fmap[f"t{cc}_b{bb}"] = Q_per_ch[:, cc, :, bb].tolist()
        with open(dest_fname, "w", encoding="utf-8") as f:
            json.dump(fmap, f, ensure_ascii=False, indent=4)

    def distogram_data(self, stats, jod_max=None):
        """The numbers behind a distogram (cvvdp_metric.py:1160-1175): per channel, the JOD loss of every (frame, band)
        cell, scaled by 1/jod_max.  Returns (panels [channels, bands, frames] in [0,1], jod_max).  The band axis is flipped
        like the reference's np.flip(..., axis=0) (:1192): the baseband (the last band) is row 0, the top row imshow draws,
        and band 0 (the finest) the bottom row."""
        q = np.array(stats["Q_per_ch"], dtype=f32)
        if q.shape[0] != 1:
            raise vq_exception("Exporting distograms in batch mode is not supported")
        n_ch = q.shape[1]
        q[..., -1] *= self._baseband_weight[:n_ch].reshape(-1, 1)     # the baseband has its own weight per channel
        q *= self.get_ch_weights(n_ch) * n_ch
        loss = 10.0 - self.met2jod(q)
        if jod_max is None:
            jod_max = math.ceil(loss.max())
        loss = (loss / jod_max).clip(0.0, 1.0)
        return np.ascontiguousarray(loss[0].transpose(0, 2, 1)[:, ::-1, :]), jod_max

    def export_distogram(self, stats, fname, jod_max=None, base_size=6):
        """cvvdp_metric.py:1158-1218: one panel per channel, bands over time, written as an image (needs matplotlib).
        Host-side reporting only: it draws what distogram_data() computes, with the reference's labels and layout."""
        panels, jod_max = self.distogram_data(stats, jod_max)
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            from matplotlib import ticker
            from matplotlib.colors import Normalize
        except ImportError:
            raise RuntimeError("matplotlib is missing. Please install it before exporting distograms.")
        n_ch, n_bands, n_frames = panels.shape
        still = n_frames == 1
        fps = stats["frames_per_second"]
        labels = ["BB"] + [f"{rho:.2f}" for rho in np.flip(stats["rho_band"])[::2][1:]]     # every other band, coarsest first
        cmap = plt.colormaps["plasma"]
        fig, axs = plt.subplots(nrows=n_ch, figsize=(base_size * n_frames / 60 + 1, base_size))
        for ax, panel, name in zip(axs, panels, ("A-sust", "RG", "YV", "A-trans")):
            ax.imshow(panel, cmap=cmap, aspect="auto")
            ax.set_ylabel(name)
            ax.yaxis.set_major_locator(ticker.FixedLocator(range(0, 2 * len(labels), 2)))
            ax.yaxis.set_minor_locator(ticker.MultipleLocator(1.0))
            ax.set_yticklabels(labels)
            ax.set_xticks([])
        if not still:                                                 # a time axis under the last panel only
            last = axs[-1]
            last.xaxis.set_major_locator(ticker.AutoLocator())
            last.xaxis.set_major_formatter(lambda x, pos: str(int(x / fps * 1000)))
            last.xaxis.set_minor_locator(ticker.MultipleLocator(1.0))
            last.set_xlabel("Time [ms]")
        right, bar = (0.5, [0.725, 0.1, 0.125, 0.8]) if still else (0.9, [0.925, 0.1, 0.025, 0.8])
        plt.subplots_adjust(bottom=0.1, right=right, top=0.9)
        plt.colorbar(plt.cm.ScalarMappable(norm=Normalize(0, jod_max), cmap=cmap), cax=plt.axes(bar), cmap=cmap)
        plt.savefig(fname, bbox_inches="tight")
        plt.close(fig)

    # ------------------------------------------------------------------ test / bench hooks
    def debug_buffer(self, which, level=0):
        """float32 view of an internal workspace buffer (tests only)."""
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        rc = _capi.lib().cvvdp_debug_buffer(self._handle, which, level, ctypes.byref(ptr), ctypes.byref(n))
        _capi.check(self._handle, rc, "cvvdp_debug_buffer")
        off = ptr.value - self._ws.data_ptr()
        return self._ws[off:off + n.value * 4].view(torch.float32)

    @property
    def fused_levels(self):
        """Leading pyramid levels of the clip scored last whose band kernel computed the next level itself (band4f.hip)."""
        return int(_capi.lib().cvvdp_fused_levels(self._handle))

    def profile(self, enable=True, per_call=False):
        """HIP-event timing of the kernel families ("hipEvent timings exposed in stats").  per_call=True: every
        predict() / predict_video_source() afterwards carries stats["kernel_ms"] = {family: milliseconds of that call}
        (the events are read -- one host wait -- when the call returns).  per_call=False (bench.py): the times accumulate until
        profile_read()."""
        self._profile_per_call = bool(enable and per_call)
        _capi.check(self._handle, _capi.lib().cvvdp_profile_enable(self._handle, int(enable)), "cvvdp_profile_enable")

    def profile_read(self):
        ms = (ctypes.c_double * _capi.PROF_N)()
        cnt = (ctypes.c_int32 * _capi.PROF_N)()
        _capi.check(self._handle, _capi.lib().cvvdp_profile_read(self._handle, ms, cnt), "cvvdp_profile_read")
        return {name: (ms[i], cnt[i]) for i, name in enumerate(_capi.PROF_NAMES)}


register_metric(cvvdp)
