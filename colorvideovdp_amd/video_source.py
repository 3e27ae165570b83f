"""Frame sources, mirroring the interface of pycvvdp/video_source.py.

`video_source` is the protocol custom sources implement (video_source.py:17-78).
`video_source_array` (video_source.py:243-346) wraps numpy / torch arrays.  Unlike the reference it does
not convert frames itself: the metric hands the raw samples (u8 / u16 / f16 / f32) straight to the HIP
photometry kernel, which does the unpack + display model + DKL transform in one pass.
"""
import numpy as np
import torch


class video_source:
    def get_video_size(self):
        """(height, width, frames)"""
        raise NotImplementedError

    def get_frames_per_second(self):
        raise NotImplementedError

    def get_test_frame(self, frame, device, colorspace):
        raise NotImplementedError

    def get_reference_frame(self, frame, device, colorspace):
        raise NotImplementedError

    def get_frame_count(self):
        return self.get_video_size()[2]

    def get_batch_size(self):
        return 1


def reshuffle_dims(T, in_dims, out_dims):
    """Reorder / add singleton dimensions, e.g. 'HWC' -> 'BCFHW' (video_source.py:120-162)."""
    in_dims, out_dims = in_dims.upper(), out_dims.upper()
    assert len(in_dims) == T.dim(), "The in_dims string must have as many characters as there are dimensions in T"
    keep = [ch for ch in out_dims if ch in in_dims]
    for k in sorted((i for i, ch in enumerate(in_dims) if ch not in keep), reverse=True):
        assert T.shape[k] == 1, "Only the dimensions of size 1 can be skipped in the output"
        T = T.squeeze(dim=k)
    present = [ch for ch in in_dims if ch in keep]
    T = T.permute([present.index(ch) for ch in keep])
    shape = [T.shape[keep.index(ch)] if ch in keep else 1 for ch in out_dims]
    return T.reshape(shape)


_DTYPES = {torch.uint8: 0, torch.int16: 1, torch.float16: 2, torch.float32: 3}


def _as_tensor(a):
    if isinstance(a, np.ndarray):
        if a.dtype == np.uint16:
            a = a.view(np.int16)  # torch has no uint16: keep the bit pattern (video_source.py:259-263)
        if not a.flags.writeable:
            a = a.copy()
        return torch.from_numpy(a)
    return a


class video_source_array(video_source):
    """Test/reference content held in arrays (video_source.py:243-293)."""

    def __init__(self, test_video, reference_video, fps, dim_order="BCFHW", display_photometry=None, config_paths=[]):
        self.display_photometry = display_photometry
        if tuple(test_video.shape) != tuple(reference_video.shape):
            ind = dim_order.find("B")
            if ind >= 0 and (test_video.shape[ind] == 1 or reference_video.shape[ind] == 1):
                pass
            else:
                raise RuntimeError("Test and reference image/video tensors must be exactly the same shape")
        if len(dim_order) != len(test_video.shape):
            raise RuntimeError('Input tensor much have exactly as many dimensions as there are characters in the "dims" parameter')
        test_video = reshuffle_dims(_as_tensor(test_video), dim_order, "BCFHW")
        reference_video = reshuffle_dims(_as_tensor(reference_video), dim_order, "BCFHW")
        B, C, F, H, W = test_video.shape
        if fps == 0 and F > 1:
            raise RuntimeError("When passing video sequences, you must set frames_per_second parameter")
        if C not in (1, 3):
            raise RuntimeError("The content must have either 1 or 3 color channels.")
        for a in (test_video, reference_video):
            if a.dtype not in _DTYPES:
                raise RuntimeError(f"Only uint8, uint16 and float32 is currently supported. {a.dtype} encountered.")
        if test_video.dtype != reference_video.dtype:
            raise RuntimeError("Test and reference must have the same data type")
        self.fps = fps
        self.is_video = fps > 0
        self.is_color = C == 3
        self.test_video = test_video
        self.reference_video = reference_video

    def get_frames_per_second(self):
        return self.fps

    def get_video_size(self):
        sh = self.test_video.shape
        return (sh[3], sh[4], sh[2])

    def get_batch_size(self):
        return max(self.test_video.shape[0], self.reference_video.shape[0])

    # raw access used by the metric's fast path
    def raw_arrays(self):
        return self.test_video, self.reference_video, _DTYPES[self.test_video.dtype]

    def get_test_frame(self, frame, device, colorspace):
        raise NotImplementedError("video_source_array frames are converted on the GPU by colorvideovdp_amd.cvvdp")

    get_reference_frame = get_test_frame
