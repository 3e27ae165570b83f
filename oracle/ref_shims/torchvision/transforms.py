"""Restatement of torchvision.transforms.GaussianBlur (published algorithm):

    half = (k-1)/2 ; x = linspace(-half, half, k) ; pdf = exp(-0.5 (x/sigma)^2)
    k1d = pdf / pdf.sum() ; k2d = k1d_y[:,None] @ k1d_x[None,:]
    img = pad(img, [k//2]*4, mode='reflect') ; conv2d(img, k2d, groups=C)

sigma is drawn uniformly from (sigma_min, sigma_max); with a scalar sigma both are equal.
"""
import torch
import torch.nn.functional as F


def _kernel1d(ksize, sigma, dtype, device):
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize, dtype=dtype, device=device)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


class GaussianBlur(torch.nn.Module):
    def __init__(self, kernel_size, sigma=(0.1, 2.0)):
        super().__init__()
        if isinstance(kernel_size, (int, float)):
            kernel_size = (int(kernel_size), int(kernel_size))
        self.kernel_size = tuple(kernel_size)
        if isinstance(sigma, (int, float)):
            sigma = (float(sigma), float(sigma))
        self.sigma = tuple(float(s) for s in sigma)

    def forward(self, img):
        s = torch.empty(1).uniform_(self.sigma[0], self.sigma[1]).item()
        kx, ky = self.kernel_size
        dtype = img.dtype if torch.is_floating_point(img) else torch.float32
        k1x = _kernel1d(kx, s, dtype, img.device)
        k1y = _kernel1d(ky, s, dtype, img.device)
        k2d = torch.mm(k1y[:, None], k1x[None, :])
        squeeze = img.dim() == 3
        if squeeze:
            img = img.unsqueeze(0)
        C = img.shape[-3]
        k = k2d.expand(C, 1, ky, kx)
        out = F.pad(img, [kx // 2, kx // 2, ky // 2, ky // 2], mode="reflect")
        out = F.conv2d(out, k, groups=C)
        return out.squeeze(0) if squeeze else out
