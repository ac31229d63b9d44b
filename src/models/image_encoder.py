"""SpatialEncoder -- drop-in for reference src/models/image_encoder.py (:14-303): same constructor, attributes
(`latent`, `depths`, `depths_std`, `normals`, `nviews`, `nobjects`, `latent_size`, `feature_padding`) and
state-dict keys (`model.*` = torchvision ResNet names, `positional_encoding._freqs/_phases`).

* The four pixel-aligned lookups `index`, `index_depth`, `index_depth_std`, `index_normal` (:97-223) run the HIP
  kernels behind diner_index_f32 (diner_amd/csrc/stage_ops.hip).  Inside the fused renderer / sampler the same
  lookups are done in registers and these methods are not called.
* `forward` (the ResNet34 trunk, :225-291) is per-image setup, not part of the hot path: plain torch ops
  (MIOpen convolutions on ROCm).  torchvision is not in the MI355X image, so the trunk is defined here with
  torchvision's module names; torchvision is used instead when it is importable (e.g. for pretrained weights).
"""
import functools

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from diner_amd import ops
from src.models.positional_encoding import PositionalEncoding


# ---- ResNet trunk with torchvision-compatible parameter names --------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), norm_layer(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _ResNetTrunk(nn.Module):
    def __init__(self, layers, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        planes, inpl, stages = [64, 128, 256, 512], 64, []
        for i, n in enumerate(layers):
            blocks = []
            for j in range(n):
                blocks.append(_BasicBlock(inpl, planes[i], (1 if i == 0 else 2) if j == 0 else 1, norm_layer))
                inpl = planes[i]
            stages.append(nn.Sequential(*blocks))
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        self.avgpool = nn.Sequential()
        self.fc = nn.Sequential()


def _make_backbone(backbone, pretrained, norm_layer):
    try:                                                   # pragma: no cover - torchvision absent in the MI355X image
        import torchvision
        return getattr(torchvision.models, backbone)(pretrained=pretrained, norm_layer=norm_layer)
    except ImportError:
        layers = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3]}.get(backbone)
        if layers is None:
            raise NotImplementedError(f"backbone {backbone} needs torchvision")
        if pretrained:
            import warnings
            warnings.warn("torchvision is not installed: ImageNet weights unavailable, the ResNet trunk is randomly "
                          "initialised (load a DINER checkpoint to overwrite it)")
        return _ResNetTrunk(layers, norm_layer)


class SpatialEncoder(nn.Module):
    def __init__(self, backbone="resnet34", pretrained=True, num_layers=4, index_interp="bilinear",
                 index_padding="border", upsample_interp="bilinear", use_first_pool=True, image_padding=0,
                 padding_pe=-1):
        super().__init__()
        self.use_first_pool = use_first_pool
        norm_layer = functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
        self.model = _make_backbone(backbone, pretrained, norm_layer)
        self.model.fc = nn.Sequential()
        self.model.avgpool = nn.Sequential()
        self.latent_size = [0, 64, 128, 256, 512, 1024][num_layers]
        self.image_padding = image_padding
        self.feature_padding = image_padding / self.model.conv1.stride[0]
        assert self.feature_padding % 1 == 0
        self.pad_layer = nn.ReplicationPad2d([self.image_padding] * 4)
        self.padding_pe = padding_pe
        if self.padding_pe < 0 or self.feature_padding == 0:
            self.positional_encoding = None
        else:
            self.positional_encoding = PositionalEncoding(padding_pe, freq_factor=np.pi, d_in=2, include_input=True)
            old = self.model.conv1                       # widen conv1 for the padding-PE channels (:63-86)
            new = nn.Conv2d(old.in_channels + self.positional_encoding.d_out, old.out_channels,
                            kernel_size=old.kernel_size, stride=old.stride, padding=old.padding,
                            bias=old.bias is not None, dilation=old.dilation, padding_mode=old.padding_mode,
                            groups=old.groups)
            nn.init.kaiming_normal_(new.weight, mode="fan_out", nonlinearity="relu")
            with torch.no_grad():
                new.weight[:, :old.weight.shape[1]] = old.weight.detach()
            self.model.conv1 = new
        self.num_layers = num_layers
        self.index_interp, self.index_padding, self.upsample_interp = index_interp, index_padding, upsample_interp
        self.register_buffer("latent", torch.empty(1, 1, 1, 1), persistent=False)
        self.nviews = None
        self.nobjects = None
        self._scene_cache = {}

    # ---- HIP lookups ------------------------------------------------------------------------------------------
    def _lookup_scene(self, sb, which):
        """HipScene holding only the map `which` needs (cameras are irrelevant for a raw uv lookup)."""
        t = {"latent": self.latent, "depth": self.depths, "std": self.depths_std, "normal": self.normals}[which]
        # one entry per (map, object).  The HipScene of the latent holds a channels-last COPY, so the entry is only valid
        # for the tensor object it was built from: the key carries the tensor's identity and version, the entry keeps the
        # tensor alive (its address cannot be recycled for another latent while cached), and forward() drops the cache.
        key = (id(t), t.data_ptr(), t._version, tuple(t.shape))
        hit = self._scene_cache.get((which, sb))
        if hit is None or hit[0] != key:
            nv = t.shape[1]
            eye = torch.eye(4).repeat(nv, 1, 1)
            one = torch.ones(nv, 2)
            # the nearest-neighbour kernel reads all three small maps' geometry from one struct: pass the map it
            # needs and aliases for the others (never dereferenced for this mode)
            d = t[sb] if which != "latent" else None
            scene = ops.HipScene(t[sb] if which == "latent" else None,
                                 d if which == "depth" else (d[:, :1] if d is not None else None),
                                 d if which == "std" else (d[:, :1] if d is not None else None),
                                 d if which == "normal" else (d[:, :1].expand(-1, 3, -1, -1) if d is not None else None),
                                 eye, one, one, torch.ones(2), self.feature_padding)
            hit = (key, scene, t)
            self._scene_cache[(which, sb)] = hit
        return hit[1]

    def _index(self, uv, which, mode):
        if self.index_interp != "bilinear" or self.index_padding != "border":
            raise NotImplementedError("diner_amd: latent lookups implement bilinear / border (all shipped configs)")
        SB = uv.shape[0]
        return torch.stack([ops.index(self._lookup_scene(sb, which), mode, uv[sb]) for sb in range(SB)])

    def index(self, uv):
        """uv (SB,NV,N,2) in [-1,1] (outer pixel edges) -> latent (SB,NV,L,N); bilinear / border after the
        feature-padding correction (:97-146)."""
        assert uv.shape[:2] == self.latent.shape[:2]
        return self._index(uv, "latent", ops.INDEX_LATENT)

    def index_depth(self, uv):
        """-> (SB,NV,1,N), nearest / border (:148-170)."""
        assert uv.shape[:2] == self.depths.shape[:2]
        return self._index(uv, "depth", ops.INDEX_DEPTH)

    def index_depth_std(self, uv):
        """-> (SB,NV,1,N), nearest on the 100 px exponentially padded map, zeros outside (:172-199)."""
        assert uv.shape[:2] == self.depths_std.shape[:2]
        return self._index(uv, "std", ops.INDEX_DEPTH_STD)

    def index_normal(self, uv):
        """-> (SB,NV,3,N), nearest / zeros (:201-223)."""
        assert uv.shape[:2] == self.normals.shape[:2]
        return self._index(uv, "normal", ops.INDEX_NORMAL)

    # ---- per-image setup (not the hot path) ---------------------------------------------------------------------
    def forward(self, imgs, depths, depths_std, normals):
        """imgs (SB,NV,3,H,W) -> stores self.latent (SB,NV,latent_size,Hf,Wf) and the depth / std / normal maps."""
        SB, NV, Cin, H, W = imgs.shape
        self._scene_cache = {}                       # new maps: nothing cached for the previous ones may be served
        self.depths, self.depths_std, self.normals = depths, depths_std, normals
        self.nviews, self.nobjects = NV, SB
        x = self.pad_layer(imgs.view(SB * NV, Cin, H, W))
        if self.padding_pe >= 0 and self.feature_padding > 0:
            p = self.image_padding
            ys, xs = torch.meshgrid(torch.linspace(-1, 1, H + 2 * p, device=x.device),
                                    torch.linspace(-1, 1, W + 2 * p, device=x.device), indexing="ij")
            with torch.no_grad():
                pe = self.positional_encoding(torch.stack((xs, ys), dim=-1))       # (Hp,Wp,18), x first
            pe = pe.clone()
            pe[p:-p, p:-p] = 0                                                      # PE only on the padding ring
            x = torch.cat((x, pe.permute(2, 0, 1).unsqueeze(0).expand(SB * NV, -1, -1, -1)), dim=1)
        m = self.model
        x = m.relu(m.bn1(m.conv1(x)))
        latents = [x]
        if self.num_layers > 1:
            if self.use_first_pool:
                x = m.maxpool(x)
            x = m.layer1(x)
            latents.append(x)
        if self.num_layers > 2:
            x = m.layer2(x)
            latents.append(x)
        if self.num_layers > 3:
            x = m.layer3(x)
            latents.append(x)
        if self.num_layers > 4:
            x = m.layer4(x)
            latents.append(x)
        align = None if self.index_interp == "nearest " else True
        size = latents[0].shape[-2:]
        if x.is_cuda and getattr(self, "latent_channels_last", True):      # (the attribute: a test switch, not a constructor argument of the reference)
            # the pyramid levels in channels-last memory format BEFORE they are upsampled (small maps; level 0 is an eighth of the latent): the
            # interpolation and the concatenation keep the format, so the latent arrives the way the HIP kernels read it (one contiguous 2 KB
            # row per texel) without the NCHW -> channels-last copy of a whole latent per encode, and its gradient returns the same way
            latents = [l.contiguous(memory_format=torch.channels_last) for l in latents]
        latents = [F.interpolate(l, size, mode=self.upsample_interp, align_corners=align) for l in latents]
        lat = torch.cat(latents, dim=1)
        self.latent = lat.view(SB, NV, -1, *lat.shape[-2:])
