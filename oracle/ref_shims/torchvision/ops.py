"""Dummy: only has to exist so that `import pycvvdp` succeeds (cvvdp_ml_metric.py:21)."""
import torch


class MLP(torch.nn.Sequential):
    def __init__(self, *a, **k):
        super().__init__()
