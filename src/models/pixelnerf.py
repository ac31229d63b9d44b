"""PixelNeRF -- drop-in for reference src/models/pixelnerf.py (:12-145): same constructor (built through the
import_obj plugin seam), attributes (`poses`, `focal`, `c`, `image_shape`, `encoder`, `mlp_fine`, `poscode`,
`depthcode`, `d_in`, `d_latent`, `d_out`) and state-dict keys.

`forward(xyz, viewdirs)` evaluates the radiance field with the fused HIP kernels (projection + positional encoding +
bilinear feature gather + ResnetFC on fp32 MFMA, diner_amd/csrc/mlp.hip) through diner_field_from_points_f32.
`encode` is per-image setup and stays in torch ops, as in the reference.  In grad mode `forward` switches to the training
path of diner_amd/train.py (HIP forward that keeps activations + HIP backward)."""
import torch

from diner_amd import ops
from src.models.positional_encoding import PositionalEncoding
from src.util.depth2normal import depth2normal
from src.util.import_helper import import_obj

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


class _Normalize(torch.nn.Module):
    """torchvision.transforms.Normalize for (..., 3, H, W) tensors (pixelnerf.py:32-33)."""

    def __init__(self, mean, std):
        super().__init__()
        self.mean, self.std = tuple(mean), tuple(std)

    def forward(self, x):
        mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - mean) / std


class PixelNeRF(torch.nn.Module):
    def __init__(self, poscode_conf, encoder_conf, mlp_fine_conf):
        super().__init__()
        self.poscode = PositionalEncoding(**poscode_conf.kwargs, d_in=3)
        self.depthcode = PositionalEncoding(**poscode_conf.kwargs, d_in=1)
        self.encoder = import_obj(encoder_conf.module)(**encoder_conf.kwargs)
        self.d_in = self.poscode.d_out + self.depthcode.d_out + 3
        self.d_latent = self.encoder.latent_size
        self.d_out = 4
        self.mlp_fine = import_obj(mlp_fine_conf.module)(**mlp_fine_conf.kwargs, d_latent=self.d_latent,
                                                         d_in=self.d_in, d_out=self.d_out)
        self.register_buffer("poses", torch.empty(1, 3, 4), persistent=False)
        self.register_buffer("image_shape", torch.empty(2), persistent=False)
        self.register_buffer("focal", torch.empty(1, 2), persistent=False)
        self.register_buffer("c", torch.empty(1, 2), persistent=False)
        self.normalize_rgb = _Normalize(_MEAN, _STD)
        self._scenes = {}

    def encode(self, images, depths, depths_std, extrinsics, intrinsics):
        """images (SB,NV,3,H,W), depths / depths_std (SB,NV,1,H,W), extrinsics (SB,NV,4,4), intrinsics (SB,NV,3,3):
        builds the feature maps and stores the source cameras (:35-53).  Call before forward()."""
        images = self.normalize_rgb(images)
        normals = depth2normal(depths.flatten(end_dim=1), intrinsics.flatten(end_dim=1)).reshape_as(images)
        self.encoder(images, depths, depths_std, normals)
        self.poses = extrinsics
        self.c = intrinsics[:, :, :2, -1]
        self.focal = intrinsics[:, :, torch.tensor([0, 1]), torch.tensor([0, 1])]
        self.image_shape[0] = images.shape[-1]      # width
        self.image_shape[1] = images.shape[-2]      # height
        self._scenes = {}

    # ---- HIP state ------------------------------------------------------------------------------------------------
    def hip_scene(self, sb):
        """HipScene of object `sb` (channels-last latent copy etc.), rebuilt whenever any source tensor changed."""
        enc = self.encoder
        srcs = (enc.latent, enc.depths, enc.depths_std, enc.normals, self.poses, self.focal, self.c, self.image_shape)
        key = tuple((id(t), t.data_ptr(), t._version, tuple(t.shape)) for t in srcs)
        hit = self._scenes.get(sb)
        if hit is None or hit[0] != key:
            scene = ops.HipScene(enc.latent[sb], enc.depths[sb], enc.depths_std[sb], enc.normals[sb], self.poses[sb],
                                 self.focal[sb], self.c[sb], self.image_shape, enc.feature_padding)
            hit = (key, scene, srcs)       # the sources stay alive with the entry: their addresses cannot be recycled
            self._scenes[sb] = hit
        return hit[1]

    def hip_mlp(self):
        """Packed ResnetFC weights (fused kernels) or the generic-path parameter block; either carries the positional encoding that
        feeds the MLP (the fused field kernels evaluate it in registers)."""
        pc = self.poscode
        return self.mlp_fine.hip_mlp(num_freqs=pc.num_freqs, freq_factor=pc.freq_factor, include_input=pc.include_input, nv=self._nv())

    def is_generic(self):
        """A configuration outside the fused field kernels (another d_hidden / n_blocks / combine_layer / positional encoding / number of
        views / latent width): rendered on the generic slow path, exact fp32."""
        pc = self.poscode
        return not self.mlp_fine.is_fused_shape(self._nv(), pc.num_freqs, pc.include_input)

    def _nv(self):
        """Source views of the encoded scene (the shipped 4 before any encode)."""
        nv = getattr(self.encoder, "nviews", None)
        return int(nv) if nv else 4

    def needs_grad(self):
        """True when a call must be differentiable: grad mode and some parameter (or the encoded latent) wants a gradient."""
        return torch.is_grad_enabled() and (self.encoder.latent.requires_grad or
                                            any(p.requires_grad for p in self.mlp_fine.parameters()))

    def _check_poscode(self):
        """The fused kernels (and the fast training path) are built for poscode num_freqs=6, include_input=True (d_in = 55); other
        configurations run on the generic path -- since round 6 in grad mode too (diner_amd.train.field_train_generic).  depthcode shares
        poscode's configuration (pixelnerf.py:15-16)."""
        return None

    def forward(self, xyz, viewdirs):
        """(r, g, b, sigma) at world-space points: xyz (SB,B,3), viewdirs (SB,B,3) -> (SB,B,4) (:55-145)."""
        SB, B, _ = xyz.shape
        assert SB == self.encoder.nobjects
        self._check_poscode()
        if self.needs_grad():
            # training: un-fused HIP forward that keeps the activations + HIP backward
            # (diner_amd/train.py); gradients reach the MLP parameters and, through encoder.latent, the image encoder
            from diner_amd import train
            if self.is_generic():       # any other configuration: exact fp32, one GEMM launch per layer and adjoint
                pc = self.poscode
                return torch.stack([train.field_train_generic(self.hip_scene(sb), self.mlp_fine, xyz[sb], viewdirs[sb], self.encoder.latent[sb],
                                                              pc.num_freqs, pc.include_input, pc.freq_factor) for sb in range(SB)])
            return train.field_train_batch([self.hip_scene(sb) for sb in range(SB)], xyz, viewdirs, self.encoder.latent,
                                           train.mlp_params(self.mlp_fine), self.poscode.freq_factor)
        mlp = self.hip_mlp()
        if isinstance(mlp, ops.GenericMlp):
            return torch.stack([ops.field_generic(self.hip_scene(sb), mlp, xyz=xyz[sb], viewdirs=viewdirs[sb]) for sb in range(SB)])
        return torch.stack([ops.field_from_points(self.hip_scene(sb), mlp, xyz[sb], viewdirs[sb]) for sb in range(SB)])
