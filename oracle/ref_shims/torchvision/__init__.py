"""Container-only import shim (test infrastructure, never shipped as product code).

torchvision is not installed in the build image.  The reference imports
`torchvision.transforms.GaussianBlur` (cvvdp_metric.py:11,158,968) and
`torchvision.ops.MLP` (cvvdp_ml_metric.py:21).  This package restates
torchvision's *published* GaussianBlur algorithm (torchvision>=0.9.2,
transforms/_functional_tensor.py: `_get_gaussian_kernel1d/2d`, `gaussian_blur`)
so that /root/reference can be imported here to generate golden vectors.
"""
