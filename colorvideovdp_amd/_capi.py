"""ctypes binding of include/cvvdp_hip.h (libcvvdp_hip.so, built by csrc/Makefile).

There is deliberately no fallback: if the shared library is missing or does not load, importing the
metric fails with an ImportError that says how to build it.
"""
import ctypes as C
import os

MAX_FILTER_LEN = 65
MAX_LEVELS = 16
MAX_WINDOW = 256
CSF_NODES = 32
PROF_N = 6
PROF_NAMES = ("photometry", "temporal_fir", "pyr_reduce", "band_level0", "band_rest", "heatmap")
ABI_VERSION = 13
RESIZE_MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2, "area": 3}   # CVVDP_RESIZE_*

U8, U16, F16, F32, F32_DKL, YUV8, YUV16 = range(7)
HEATMAP = {None: 0, "none": 0, "raw": 1, "threshold": 2, "supra-threshold": 3}
BUF_HIST, BUF_GPYR, BUF_DDUMP, BUF_HEAT, BUF_Q = range(5)


class Params(C.Structure):
    _fields_ = [
        ("eotf", C.c_int32),
        ("Y_peak", C.c_float), ("Y_black", C.c_float), ("Y_refl", C.c_float), ("exposure", C.c_float),
        ("gamma", C.c_float),
        ("rgb2dkl", C.c_float * 9),
        ("mask_p", C.c_float),
        ("mask_c10", C.c_float),
        ("mask_q", C.c_float * 4),
        ("xcm", C.c_float * 16),
        ("ch_gain", C.c_float * 4),
        ("d_max10", C.c_float),
        ("sens_mul", C.c_float),
        ("blur_radius", C.c_int32),
        ("blur_taps", C.c_float * 13),
        ("beta", C.c_float), ("beta_t", C.c_float), ("beta_tch", C.c_float), ("beta_sch", C.c_float),
        ("jod_a", C.c_float), ("jod_exp", C.c_float), ("image_int", C.c_float),
        ("ch_w", C.c_float * 4),
        ("baseband_weight", C.c_float * 4),
        ("csf_logL_first", C.c_float), ("csf_logL_last", C.c_float),
    ]


class Clip(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("channels", C.c_int32),
        ("height", C.c_int32), ("width", C.c_int32),
        ("is_video", C.c_int32),
        ("n_frames", C.c_int32),
        ("first_frame", C.c_int32),
        ("n_levels", C.c_int32),
        ("filter_len", C.c_int32),
        ("block_frames", C.c_int32),
        ("heatmap", C.c_int32),
        ("debug_dump", C.c_int32),
        ("raw_halo", C.c_int32), ("total_frames", C.c_int32),
        ("feature_size", C.c_int32), ("fuse_mode", C.c_int32), ("band_layout", C.c_int32),
        ("defer_bands", C.c_int32), ("score_frames", C.c_int32),
        ("taps", C.c_float * (4 * MAX_FILTER_LEN)),
        ("csf_rows", C.c_float * (MAX_LEVELS * 4 * CSF_NODES)),
    ]


class YuvFormat(C.Structure):
    _fields_ = [
        ("chroma", C.c_int32), ("bit_depth", C.c_int32), ("matrix", C.c_int32), ("reserved", C.c_int32),
        ("frame_stride_test", C.c_int64), ("frame_stride_ref", C.c_int64),
    ]


SYMBOLS = {
    "cvvdp_abi_version": (C.c_int, []),
    "cvvdp_build_flags": (C.c_int, []),
    "cvvdp_build_info": (C.c_char_p, []),
    "cvvdp_compiled_hip_version": (C.c_int, []),
    "cvvdp_runtime_hip_version": (C.c_int, []),
    "cvvdp_fused_levels": (C.c_int, [C.c_void_p]),
    "cvvdp_struct_sizes": (None, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cvvdp_create": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "cvvdp_destroy": (None, [C.c_void_p]),
    "cvvdp_last_error": (C.c_char_p, [C.c_void_p]),
    "cvvdp_configure": (C.c_int, [C.c_void_p, C.POINTER(Clip)]),
    "cvvdp_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "cvvdp_bind_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cvvdp_put_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "cvvdp_process_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_void_p]),
    "cvvdp_process_block_yuv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(YuvFormat), C.c_int32, C.POINTER(C.c_int32),
                                          C.c_int32, C.c_int32, C.c_void_p]),
    "cvvdp_unpack_yuv_resized": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(YuvFormat), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvvdp_process_block_filtered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                               C.c_int32, C.c_int32, C.c_void_p]),
    "cvvdp_get_features": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "cvvdp_process_image": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cvvdp_get_q_per_ch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvvdp_pool_jod": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "cvvdp_score_frames": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "cvvdp_get_heatmap": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "cvvdp_get_heatmap_rgb8": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "cvvdp_debug_buffer": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "cvvdp_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "cvvdp_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
}

# The product loads the in-tree library and nothing else.  Development only (tools/ab_bench.sh: kernel variants built side by
# side are benchmarked against each other): CVVDP_LIB=<path> is honoured when CVVDP_DEV_KNOBS=1 is set as well.
_IN_TREE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcvvdp_hip.so")
LIB_PATH = (os.environ.get("CVVDP_LIB") if os.environ.get("CVVDP_DEV_KNOBS") == "1" else None) or _IN_TREE
BUILD_DEV_KNOBS, BUILD_SAFE_LOADS, BUILD_DIAG = 1, 2, 4
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make -C colorvideovdp_amd/csrc` "
                              "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:
            raise ImportError(f"cannot load {LIB_PATH}: {e}. There is no CPU fallback.") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError here = ABI mismatch
            fn.restype = res
            fn.argtypes = args
        v = l.cvvdp_abi_version()
        if v != ABI_VERSION:
            raise ImportError(f"libcvvdp_hip.so has ABI {v}, binding expects {ABI_VERSION}")
        flags = l.cvvdp_build_flags()
        if flags != 0 and os.environ.get("CVVDP_DEV_KNOBS") != "1":
            # environment tuning knobs, compiler-managed loads (`make safe`) or a timing-only switch of the band kernels compiled in:
            # not the product.  Such a library is loaded only when a development library was asked for explicitly.
            raise ImportError(f"{LIB_PATH} is not a product build (cvvdp_build_flags() = {flags}: 1 dev knobs, 2 safe loads, 4 timing-only "
                              "diagnostics); rebuild it with `make -C colorvideovdp_amd/csrc`, or set CVVDP_DEV_KNOBS=1 to load it anyway")
        sp, sc = C.c_int32(), C.c_int32()
        l.cvvdp_struct_sizes(C.byref(sp), C.byref(sc))
        if (sp.value, sc.value) != (C.sizeof(Params), C.sizeof(Clip)):
            raise ImportError(f"struct layout mismatch: library {(sp.value, sc.value)} vs binding {(C.sizeof(Params), C.sizeof(Clip))}")
        # The band kernels' hand-issued loads were checked against the register allocation of the compiler the library was built with
        # (cvvdp_build_info; bench.py prints it in config.library_build).  The HIP runtime in the process is whatever torch's wheel
        # bundles (here 7.0 under a 7.2 toolchain: a minor-version gap is the normal state and says nothing), so only another MAJOR
        # version -- a different code-object / launch ABI generation -- is reported.
        built, running = l.cvvdp_compiled_hip_version(), l.cvvdp_runtime_hip_version()
        if running and built // 10000000 != running // 10000000:
            import warnings
            warnings.warn(f"libcvvdp_hip.so was built and ISA-checked with {l.cvvdp_build_info().decode()}; the HIP runtime in this process is "
                          f"{running // 10000000}.{running // 100000 % 100}.{running % 100000}: rebuild with `make -C colorvideovdp_amd/csrc` "
                          "(the build re-runs the check)", RuntimeWarning)
        _lib = l
    return _lib


def build_info():
    """(toolchain string of the loaded library, compiled HIP version, runtime HIP version)."""
    l = lib()
    return l.cvvdp_build_info().decode(), l.cvvdp_compiled_hip_version(), l.cvvdp_runtime_hip_version()


class CoreError(RuntimeError):
    pass


def check(handle, code, what):
    if code != 0:
        msg = lib().cvvdp_last_error(handle)
        raise CoreError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")
