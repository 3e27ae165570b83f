"""colorvideovdp_amd: MI355X-native compute core for the ColorVideoVDP metric behind the reference's
Python API (`cvvdp.predict`, `cvvdp.predict_video_source`, heat maps, distograms)."""
from .cvvdp_metric import cvvdp
from .display_model import vvdp_display_geometry, vvdp_display_photo_eotf, vvdp_display_photometry
from .video_source import reshuffle_dims, video_source, video_source_array
from .video_source_file import load_image_as_array, video_source_file, video_source_image_frames
from .video_source_yuv import video_source_yuv_file
from .vq_metric import register_metric, vq_exception, vq_metric, vq_metric_dict

__version__ = "0.3.0"
COMPUTE_DTYPE = "f32"      # the arithmetic of every kernel (integer / fp16 / Y'CbCr samples are unpacked to fp32 by the first kernel of the path)
