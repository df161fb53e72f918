"""PositionalEncoder: sin/cos view-direction encoding (reference encoder.py:8-33) on the GPU kernel a7."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class PositionalEncoder(nn.Module):
    def __init__(self, d_model, num_freq=10):
        super().__init__()
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.d_model = d_model
        self.max_seq_len = num_freq
        # kept for attribute compatibility (encoder.py:16-17): frequencies k = 0..num_freq-1 as int8
        self.sinus_in = torch.arange(0, num_freq, dtype=torch.int8)[None, None, :]

    def forward(self, x):
        """[..., d_model] -> [..., d_model*2*num_freq]; per coordinate sin(2*x*k) for all k, then cos."""
        return ops.dir_encode(x, self.max_seq_len)
