"""ResnetFC -- drop-in for reference src/models/resnetfc.py (ResnetBlockFC :18-69, ResnetFC :72-159): same
constructor arguments, parameter names / shapes / initialisation (state-dict compatible with reference
checkpoints); forward runs the fused MFMA kernels of diner_amd/csrc/mlp.hip through diner_mlp_forward_f32.

The packed-weights handle is cached and keyed on every parameter's (data_ptr, _version), so in-place optimiser
updates or load_state_dict() invalidate it."""
import torch
from torch import nn

from diner_amd import ops


class ResnetBlockFC(nn.Module):
    """fc_1(act(fc_0(act(x)))) + shortcut(x); parameters only -- the arithmetic lives in the HIP kernel."""

    def __init__(self, size_in, size_out=None, size_h=None, beta=0.0):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)
        self.activation = nn.Softplus(beta=beta) if beta > 0 else nn.ReLU()
        if size_in == size_out:
            self.shortcut = None
        else:
            self.shortcut = nn.Linear(size_in, size_out, bias=False)
            nn.init.kaiming_normal_(self.shortcut.weight, a=0, mode="fan_in")


class ResnetFC(nn.Module):
    def __init__(self, d_in, d_out=4, n_blocks=5, d_latent=0, d_hidden=128, beta=0.0, combine_layer=1000,
                 combine_type="average"):
        super().__init__()
        if d_in > 0:
            self.lin_in = nn.Linear(d_in, d_hidden)
            nn.init.constant_(self.lin_in.bias, 0.0)
            nn.init.kaiming_normal_(self.lin_in.weight, a=0, mode="fan_in")
        self.lin_out = nn.Linear(d_hidden, d_out)
        nn.init.constant_(self.lin_out.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_out.weight, a=0, mode="fan_in")
        self.n_blocks, self.d_latent, self.d_in, self.d_out, self.d_hidden = n_blocks, d_latent, d_in, d_out, d_hidden
        self.combine_layer, self.combine_type, self.beta = combine_layer, combine_type, beta
        self.blocks = nn.ModuleList([ResnetBlockFC(d_hidden, beta=beta) for _ in range(n_blocks)])
        if d_latent != 0:
            n_lin_z = min(combine_layer, n_blocks)
            self.lin_z = nn.ModuleList([nn.Linear(d_latent, d_hidden) for _ in range(n_lin_z)])
            for i in range(n_lin_z):
                nn.init.constant_(self.lin_z[i].bias, 0.0)
                nn.init.kaiming_normal_(self.lin_z[i].weight, a=0, mode="fan_in")
        self.activation = nn.Softplus(beta=beta) if beta > 0 else nn.ReLU()
        self._hip = None
        self._hip_key = None

    # ---- packed weights -----------------------------------------------------------------------------------
    def _check_supported(self):
        if self.combine_type != "average":      # (the reference's own combine() raises for anything else, resnetfc.py:9-14)
            raise NotImplementedError("diner_amd: ResnetFC implements average view fusion (the only combine_type of the reference)")

    def is_fused_shape(self, nv=4, num_freqs=6, include_input=True):
        """The one configuration the fused field kernels are built for (every shipped DINER config); anything else runs on the
        generic slow path (ops.GenericMlp: exact fp32, one GEMM launch per layer)."""
        return hasattr(self, "lin_in") and ops.fused_shape(self.d_in, self.d_latent, self.d_hidden, self.d_out, self.n_blocks,
                                                           self.combine_layer, nv, num_freqs, include_input, self.beta)

    def hip_mlp(self, num_freqs=6, freq_factor=6.28, include_input=True, nv=4):
        """HipMlp handle (fused kernels) or GenericMlp (any other configuration) for the current parameter values, re-made when any
        parameter changed.  The keyword arguments describe the positional encoding of the inputs (PixelNeRF passes its own;
        irrelevant for forward() on an explicit matrix) and the number of source views."""
        self._check_supported()
        sd = {k: v for k, v in self.state_dict().items()}
        fused = self.is_fused_shape(nv, num_freqs, include_input)
        key = (tuple((k, v.data_ptr(), v._version, str(v.device)) for k, v in sorted(sd.items())),
               int(num_freqs), float(freq_factor), bool(include_input), fused)
        if self._hip is None or key != self._hip_key:
            if fused:
                self._hip = ops.HipMlp(sd, combine_layer=self.combine_layer, d_latent=self.d_latent, num_freqs=num_freqs,
                                       freq_factor=freq_factor, include_input=include_input)
            else:
                self._hip = ops.GenericMlp(sd, combine_layer=self.combine_layer, beta=self.beta, num_freqs=num_freqs,
                                           freq_factor=freq_factor, include_input=include_input, d_latent=self.d_latent)
            self._hip_key = key
        return self._hip

    def forward(self, zx, combine_dim):
        """zx (SB, NV, B, d_latent + d_in) with combine_dim=1 (the reference's only call, pixelnerf.py:131-134),
        or (NV, B, d_latent + d_in) with combine_dim=0  ->  (SB, B, d_out) / (B, d_out); without a combine layer inside the network
        (combine_layer >= n_blocks, the constructor's default) the views stay: (SB, NV, B, d_out) / (NV, B, d_out)  (resnetfc.py:129-159)."""
        assert zx.size(-1) == self.d_latent + self.d_in
        if torch.is_grad_enabled() and (zx.requires_grad or any(p.requires_grad for p in self.parameters())):
            # round 6: differentiable for ANY configuration on the generic exact-fp32 path (diner_mlp_generic_train_forward_f32 /
            # _backward_f32); the fast differentiable path of the shipped configuration is PixelNeRF.forward / NeRFRendererDGS.forward
            self._check_supported()
            from diner_amd import train
            if zx.dim() == 4 and combine_dim in (1, -3):
                return torch.stack([train.generic_mlp_train(self, zx[i]) for i in range(zx.shape[0])])
            if zx.dim() == 3 and combine_dim in (0, -3):
                return train.generic_mlp_train(self, zx)
            raise NotImplementedError(f"diner_amd: ResnetFC.forward supports (SB,NV,B,C)/combine_dim=1 and (NV,B,C)/combine_dim=0, got shape "
                                      f"{tuple(zx.shape)}, combine_dim={combine_dim}")
        if zx.dim() == 4 and combine_dim in (1, -3):
            mlp = self.hip_mlp(nv=zx.shape[1])
            run = mlp.forward if isinstance(mlp, ops.GenericMlp) else (lambda m: ops.mlp_forward(mlp, m))
            return torch.stack([run(zx[i]) for i in range(zx.shape[0])])
        if zx.dim() == 3 and combine_dim in (0, -3):
            mlp = self.hip_mlp(nv=zx.shape[0])
            return mlp.forward(zx) if isinstance(mlp, ops.GenericMlp) else ops.mlp_forward(mlp, zx)
        raise NotImplementedError(f"diner_amd: ResnetFC.forward supports (SB,NV,B,C)/combine_dim=1 and "
                                  f"(NV,B,C)/combine_dim=0, got shape {tuple(zx.shape)}, combine_dim={combine_dim}")
