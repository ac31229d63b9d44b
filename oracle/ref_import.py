"""DEV-ONLY loader for the upstream reference (test infrastructure, never shipped to the product path).

Imports the reference's hot-path modules from /root/reference with the three
missing third-party modules stubbed.  Only usable in the
build container: /root/reference does not exist on the GPU box, so nothing under
tests/ -m gpu, smoke() or bench.py may import this file.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("DINER_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "src", "models"))


class _DotMap(dict):
    """attribute-dict stand-in for dotmap.DotMap (used at nerf_renderer.py:421-430)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _install_stubs():
    import torch

    if "dotmap" not in sys.modules:
        m = types.ModuleType("dotmap")
        m.DotMap = _DotMap
        sys.modules["dotmap"] = m
    if "imageio" not in sys.modules:
        sys.modules["imageio"] = types.ModuleType("imageio")
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        trf = types.ModuleType("torchvision.transforms.functional")
        mods = types.ModuleType("torchvision.models")
        utils = types.ModuleType("torchvision.utils")

        class Normalize(torch.nn.Module):
            def __init__(self, mean, std):
                super().__init__()
                self.mean, self.std = mean, std

            def forward(self, x):
                mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
                std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
                return (x - mean) / std

        class _DummyResnet(torch.nn.Module):
            def __init__(self, pretrained=False, norm_layer=None):
                super().__init__()
                self.conv1 = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False)

        tr.Normalize = Normalize
        trf.resize = None
        tr.functional = trf
        mods.resnet34 = _DummyResnet
        utils.save_image = None
        tv.transforms, tv.models, tv.utils = tr, mods, utils
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tr,
                            "torchvision.transforms.functional": trf,
                            "torchvision.models": mods, "torchvision.utils": utils})


def import_reference():
    """Returns a namespace with the reference classes / helpers of the hot path."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected in the build container only)")
    sys.dont_write_bytecode = True
    _install_stubs()
    # the repo's own drop-in package is also called `src`; make sure the reference's wins here
    saved = {k: v for k, v in sys.modules.items() if k == "src" or k.startswith("src.")}
    for k in saved:
        del sys.modules[k]
    # the reference's `src` is a namespace package (no __init__.py); a regular package of the same
    # name anywhere on sys.path would shadow it, so hide such entries while importing
    saved_path = list(sys.path)
    sys.path[:] = [REF_ROOT] + [q for q in saved_path
                                if not os.path.isfile(os.path.join(q or ".", "src", "__init__.py"))]
    try:
        import importlib
        importlib.invalidate_caches()
        ns = types.SimpleNamespace()
        ns.nerf_renderer = importlib.import_module("src.models.nerf_renderer")
        ns.pixelnerf = importlib.import_module("src.models.pixelnerf")
        ns.resnetfc = importlib.import_module("src.models.resnetfc")
        ns.positional_encoding = importlib.import_module("src.models.positional_encoding")
        ns.image_encoder = importlib.import_module("src.models.image_encoder")
        ns.torch_helpers = importlib.import_module("src.util.torch_helpers")
        ns.cam_geometry = importlib.import_module("src.util.cam_geometry")
        ns.depth2normal = importlib.import_module("src.util.depth2normal")
        ref_mods = {k: v for k, v in sys.modules.items() if k == "src" or k.startswith("src.")}
    finally:
        sys.path[:] = saved_path
        importlib.invalidate_caches()
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    ns._modules = ref_mods
    return ns


class Conf:
    """tiny stand-in for an OmegaConf node: .module / .kwargs"""

    def __init__(self, module=None, kwargs=None):
        self.module, self.kwargs = module, (kwargs or {})


def build_reference_nerf(ns, fc1_std=0.03, seed=1234):
    """PixelNeRF in the trained DTU config (configs/train_dtu.yaml:31-50) with randomised fc_1."""
    import sys as _s
    import torch
    # import_obj inside the reference resolves dotted `src.*` names through sys.modules
    saved = {k: v for k, v in _s.modules.items() if k == "src" or k.startswith("src.")}
    for k in saved:
        del _s.modules[k]
    _s.modules.update(ns._modules)
    try:
        nerf = ns.pixelnerf.PixelNeRF(
            poscode_conf=Conf(kwargs=dict(num_freqs=6, freq_factor=6.28, include_input=True)),
            encoder_conf=Conf(module="src.models.image_encoder.SpatialEncoder",
                              kwargs=dict(image_padding=64, padding_pe=4, pretrained=False)),
            mlp_fine_conf=Conf(module="src.models.resnetfc.ResnetFC",
                               kwargs=dict(n_blocks=5, d_hidden=512, combine_layer=3, combine_type="average")))
    finally:
        for k in [k for k in _s.modules if k == "src" or k.startswith("src.")]:
            del _s.modules[k]
        _s.modules.update(saved)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for blk in nerf.mlp_fine.blocks:
            blk.fc_1.weight.copy_(torch.randn(blk.fc_1.weight.shape, generator=g) * fc1_std)
            blk.fc_0.bias.copy_(torch.randn(blk.fc_0.bias.shape, generator=g) * 0.05)
            blk.fc_1.bias.copy_(torch.randn(blk.fc_1.bias.shape, generator=g) * 0.05)
        for lz in nerf.mlp_fine.lin_z:
            lz.bias.copy_(torch.randn(lz.bias.shape, generator=g) * 0.05)
        nerf.mlp_fine.lin_in.bias.copy_(torch.randn(nerf.mlp_fine.lin_in.bias.shape, generator=g) * 0.05)
        nerf.mlp_fine.lin_out.bias.copy_(torch.randn(nerf.mlp_fine.lin_out.bias.shape, generator=g) * 0.05)
    return nerf.eval()
