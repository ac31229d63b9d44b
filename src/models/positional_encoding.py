"""NeRF positional encoding -- drop-in for reference src/models/positional_encoding.py (class
PositionalEncoding, :9-53): same constructor, buffers (`_freqs`, `_phases`: state-dict compatible) and output
layout; forward runs the HIP kernel diner_posenc_f32 (diner_amd/csrc/stage_ops.hip).  Inside the fused renderer
the encoding is computed in registers and this module is not called."""
import numpy as np
import torch

from diner_amd import ops


class PositionalEncoding(torch.nn.Module):
    def __init__(self, num_freqs=6, d_in=3, freq_factor=np.pi, include_input=True):
        super().__init__()
        self.num_freqs = num_freqs
        self.d_in = d_in
        self.freq_factor = float(freq_factor)
        self.freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)
        self.d_out = self.num_freqs * 2 * d_in
        self.include_input = include_input
        if include_input:
            self.d_out += d_in
        # f1 f1 f2 f2 ... / 0 pi/2 0 pi/2 ...   (kept as buffers for checkpoint compatibility)
        self.register_buffer("_freqs", torch.repeat_interleave(self.freqs, 2).view(1, -1, 1))
        _phases = torch.zeros(2 * self.num_freqs)
        _phases[1::2] = np.pi * 0.5
        self.register_buffer("_phases", _phases.view(1, -1, 1))

    def forward(self, x):
        """x (..., d_in) -> (..., d_out): [x, sin(x f0), cos(x f0), sin(x f1), ...] with cos(a) = sin(a + fp32(pi/2))."""
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("diner_amd: the stand-alone HIP positional encoding is not differentiable (the "
                                      "reference never differentiates it with respect to its input either); the training "
                                      "path is PixelNeRF.forward / NeRFRendererDGS.forward (diner_amd/train.py)")
        assert x.shape[-1] == self.d_in
        return ops.posenc(x, self.num_freqs, self.freq_factor, self.include_input)
