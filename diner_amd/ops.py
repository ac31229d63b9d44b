"""torch-facing wrappers around the C ABI of libdiner_hip.so.

PyTorch is plumbing here: it owns device memory (caching allocator), the stream and (for multi-GPU) the
process group.  Every function takes/returns torch tensors on a HIP device and enqueues kernels on the
current stream.  There is no CPU path and no eager-torch fallback: CPU tensors raise.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib

lib = _lib.load()          # ImportError when the extension is not built -- by design

# points per field launch: bounds the 2 KB/point hand-over workspace (2 GiB at the default) without costing throughput
# (a launch of 2^20 points is 64 tiles per CU)
MAX_POINTS_PER_LAUNCH = int(os.environ.get("DINER_AMD_MAX_POINTS", 1 << 20))


def _require_hip(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("diner_amd: tensors must live on a HIP device (MI355X); there is no CPU fallback "
                               "for the rendering hot path")
        if t.dtype != torch.float32:
            raise TypeError(f"diner_amd: expected float32, got {t.dtype}")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


_const_cache = {}
_const_lock = threading.Lock()


def _t_base(n_cand, device):
    """torch.linspace(0, 1 - 1/n, n): the stratification offsets of sample_coarse (nerf_renderer.py:53-55),
    evaluated by the same torch op the reference uses, then uploaded once."""
    key = ("t_base", n_cand, str(device))
    with _const_lock:
        if key not in _const_cache:
            step = 1.0 / n_cand
            _const_cache[key] = torch.linspace(0, 1 - step, n_cand).to(device)
        return _const_cache[key]


def _std_pad_scale(device):
    """exp(e / 12 * ln 2) for e = 0..99, computed exactly like torch_helpers.py:120."""
    key = ("std_pad", str(device))
    with _const_lock:
        if key not in _const_cache:
            e = torch.arange(100, dtype=torch.float32)
            _const_cache[key] = torch.exp(e / 12 * np.log(2)).to(device)
        return _const_cache[key]


class HipScene:
    """Per-object scene state in the layout the kernels want (DinerScene of include/diner_hip.h).

    latent (NV,C,Hf,Wf) is re-laid-out to channels-last ONCE here (each bilinear tap then is one contiguous
    2 KB read instead of 512 strided 4-byte reads); depth/std/normal maps are used as they are; the three tiny
    camera arrays are kept on the host and travel inside the kernel arguments.
    """

    def __init__(self, latent, depths, depths_std, normals, poses, focal, c, image_shape, feature_padding):
        dev = None
        for t in (latent, depths, depths_std, normals):
            if t is not None:
                _require_hip(t)
                dev = t.device
        if dev is None:
            raise ValueError("HipScene needs at least one map on a HIP device")
        self.device = dev
        self.nv = int(poses.shape[0])
        self.latent_cl = None
        self.C = self.Hf = self.Wf = 0
        if latent is not None:
            assert latent.dim() == 4 and latent.shape[0] == self.nv
            self.latent_cl = _f32c(latent.permute(0, 2, 3, 1))          # (NV,Hf,Wf,C)
            self.C, self.Hf, self.Wf = int(latent.shape[1]), int(latent.shape[2]), int(latent.shape[3])
        self.depth = _f32c(depths).view(self.nv, *depths.shape[-2:]) if depths is not None else None
        self.depth_std = _f32c(depths_std).view(self.nv, *depths_std.shape[-2:]) if depths_std is not None else None
        self.normals = _f32c(normals) if normals is not None else None
        ref = self.depth if self.depth is not None else (self.depth_std if self.depth_std is not None else self.normals)
        self.Hs, self.Ws = (int(ref.shape[-2]), int(ref.shape[-1])) if ref is not None else (0, 0)
        # tiny camera arrays -> host (one sync per scene, at encode time; never inside a render call)
        self.poses_h = _f32c(poses).cpu().contiguous()
        if self.poses_h.shape[-2:] != (4, 4):
            p44 = torch.eye(4).repeat(self.nv, 1, 1)
            p44[:, :self.poses_h.shape[-2], :] = self.poses_h
            self.poses_h = p44.contiguous()
        self.focal_h = _f32c(focal).cpu().contiguous()
        self.c_h = _f32c(c).cpu().contiguous()
        ish = image_shape.detach().cpu().float()
        self.img_w, self.img_h = float(ish[0]), float(ish[1])
        self.feature_padding = float(feature_padding)
        self.std_pad_scale = _std_pad_scale(dev)
        s = _lib.DinerScene()
        s.latent_cl = self.latent_cl.data_ptr() if self.latent_cl is not None else None
        s.depth = self.depth.data_ptr() if self.depth is not None else None
        s.depth_std = self.depth_std.data_ptr() if self.depth_std is not None else None
        s.normals = self.normals.data_ptr() if self.normals is not None else None
        s.poses_host, s.focal_host, s.c_host = self.poses_h.data_ptr(), self.focal_h.data_ptr(), self.c_h.data_ptr()
        s.std_pad_scale = self.std_pad_scale.data_ptr()
        s.img_w, s.img_h, s.feature_padding = self.img_w, self.img_h, self.feature_padding
        s.nv, s.C, s.Hf, s.Wf, s.Hs, s.Ws = self.nv, self.C, self.Hf, self.Wf, self.Hs, self.Ws
        s.latent_proj = None
        s.latent_proj_f16 = None
        s.proj_stamp = 0
        s.proj_stamp_f16 = 0
        self.struct = s
        self.latent_proj = None
        self.latent_proj_f16 = None
        self._prepared_for = None
        self._f16_current = False

    def prepare(self, mlp, force=False, f16=False):
        """Hoist lin_z[0..2] out of the sample loop: project the channels-last latent once (k_hoist_linz).
        Re-run when the MLP handle changes (the handle itself is rebuilt whenever a parameter changes).
        f16: also (re)build the fp16 copy of the projected maps that PRECISION_F16 gathers from (+50 % memory, made only when that mode
        is used; one conversion pass per preparation)."""
        if self.latent_cl is None:
            raise RuntimeError("diner_amd: scene has no latent map")
        fresh = force or self._prepared_for is not mlp or self.latent_proj is None
        if fresh:
            with torch.cuda.device(self.device):
                if self.latent_proj is None:
                    nbytes = lib.diner_scene_proj_bytes(self.ref)
                    self.latent_proj = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
                _lib.check(lib.diner_scene_prepare_f32(self.ref, mlp.handle, _ptr(self.latent_proj), _stream()))
            self.struct.latent_proj = self.latent_proj.data_ptr()
            self.struct.proj_stamp = lib.diner_mlp_stamp(mlp.handle)      # the library refuses these maps with any other handle
            self._prepared_for = mlp
            self._f16_current = False
        if f16 and not self._f16_current:
            with torch.cuda.device(self.device):
                if self.latent_proj_f16 is None:
                    self.latent_proj_f16 = torch.empty(lib.diner_scene_proj_f16_bytes(self.ref) // 2, dtype=torch.float16, device=self.device)
                _lib.check(lib.diner_scene_prepare_f16(self.ref, _ptr(self.latent_proj_f16), _stream()))
            self.struct.latent_proj_f16 = self.latent_proj_f16.data_ptr()
            self.struct.proj_stamp_f16 = self.struct.proj_stamp          # ABI v5: the fp16 copy belongs to the maps of this handle
            self._f16_current = True

    @property
    def ref(self):
        return C.byref(self.struct)


class HipMlp:
    """Packed ResnetFC weights (opaque DinerMlp handle).  Built from a state_dict-like mapping with the
    reference key names (resnetfc.py:72-127); tensors must be on the HIP device."""

    def __init__(self, sd, prefix="", combine_layer=3, d_latent=512, num_freqs=6, freq_factor=6.28, include_input=True):
        self._conf = dict(prefix=prefix, combine_layer=combine_layer, num_freqs=num_freqs, include_input=include_input)
        p = self._params(sd, freq_factor)
        self.device = self._keep["lin_in_w"].device
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            # returns once the packing has completed on the stream (the sources may then be freed or updated in place)
            _lib.check(lib.diner_mlp_create(C.byref(p), _stream(), C.byref(h)))
        self.handle = h
        self._range = None

    def _params(self, sd, freq_factor):
        """DinerMlpParams over the tensors of a state_dict-like mapping (kept alive on self until the next call)."""
        prefix, combine_layer = self._conf["prefix"], self._conf["combine_layer"]
        g = lambda k: _f32c(sd[prefix + k])
        n_blocks = len([k for k in sd if k.startswith(prefix + "blocks.") and k.endswith("fc_0.weight")])
        n_z = len([k for k in sd if k.startswith(prefix + "lin_z.") and k.endswith(".weight")])
        keep = {"lin_in_w": g("lin_in.weight"), "lin_in_b": g("lin_in.bias"),
                "lin_out_w": g("lin_out.weight"), "lin_out_b": g("lin_out.bias")}
        lists = {"fc0_w": [g(f"blocks.{i}.fc_0.weight") for i in range(n_blocks)],
                 "fc0_b": [g(f"blocks.{i}.fc_0.bias") for i in range(n_blocks)],
                 "fc1_w": [g(f"blocks.{i}.fc_1.weight") for i in range(n_blocks)],
                 "fc1_b": [g(f"blocks.{i}.fc_1.bias") for i in range(n_blocks)],
                 "lin_z_w": [g(f"lin_z.{i}.weight") for i in range(n_z)],
                 "lin_z_b": [g(f"lin_z.{i}.bias") for i in range(n_z)]}
        for t in list(keep.values()) + [t for l in lists.values() for t in l]:
            _require_hip(t)
        p = _lib.DinerMlpParams()
        p.d_in = keep["lin_in_w"].shape[1]
        p.d_hidden = keep["lin_in_w"].shape[0]
        p.d_out = keep["lin_out_w"].shape[0]
        p.d_latent = lists["lin_z_w"][0].shape[1] if n_z else 0
        p.n_blocks, p.combine_layer = n_blocks, combine_layer
        # the positional encoding that produces the 55 inputs is evaluated inside the field kernels (pixelnerf.py:15-18)
        p.num_freqs, p.include_input, p.freq_factor = int(self._conf["num_freqs"]), int(bool(self._conf["include_input"])), float(freq_factor)
        p.lin_in_w, p.lin_in_b = keep["lin_in_w"].data_ptr(), keep["lin_in_b"].data_ptr()
        p.lin_out_w, p.lin_out_b = keep["lin_out_w"].data_ptr(), keep["lin_out_b"].data_ptr()
        self._arrays = {}
        for name, ts in lists.items():
            arr = (C.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])
            self._arrays[name] = arr
            setattr(p, name, C.cast(arr, C.POINTER(C.c_void_p)))
        self._keep, self._lists = keep, lists
        return p

    def update(self, sd, freq_factor=6.28, train_only=False):
        """New parameter values into this handle (diner_mlp_update, ABI v6): packed on the current stream, no allocation, no host
        synchronisation.  train_only: only what the fused training forward reads -- the handle then serves diner_amd.train alone."""
        p = self._params(sd, freq_factor)
        with torch.cuda.device(self.device):
            _lib.check(lib.diner_mlp_update(self.handle, C.byref(p), _lib.MLP_UPDATE_TRAIN_ONLY if train_only else 0, _stream()))
        self._range = None

    def _weight_range(self):
        # the fp16-operand modes carry the weights x16 as fp16 hi/lo parts: |w| must stay below 1024.  The range was reduced on the
        # device while packing (read back by diner_mlp_create; after update(): here, one stream wait); the library itself falls back
        # to the exact kernels when it does not fit.
        if self._range is None:
            wmax = C.c_float()
            ok = lib.diner_mlp_weights_fit_f16x3(self.handle, C.byref(wmax))
            if ok < 0:
                _lib.check(ok)
            self._range = (ok == 1, float(wmax.value))
        return self._range

    @property
    def h3_ok(self):
        return self._weight_range()[0]

    @property
    def wmax(self):
        return self._weight_range()[1]

    def fallback_launches(self, reset=False):
        """Field launches with this handle that the fp16-operand kernels could not finish (an activation left the fp16 range or an
        input was not finite) and that the gated exact-fp32 kernels recomputed on the device: correct results, 2-3x the time.
        Synchronises the current stream (one 4-byte read back)."""
        n = C.c_longlong()
        with torch.cuda.device(self.device):
            _lib.check(lib.diner_mlp_fallback_count(self.handle, C.byref(n), int(bool(reset)), _stream()))
        return int(n.value)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib.diner_mlp_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def fused_shape(d_in, d_latent, d_hidden, d_out, n_blocks, combine_layer, nv=4, num_freqs=6, include_input=True, beta=0.0):
    """True for the one ResnetFC / PixelNeRF configuration the fused field kernels are built for (every shipped DINER config,
    configs/train_dtu.yaml:39-50); anything else takes the generic slow path (GenericMlp)."""
    return (d_in, d_latent, d_hidden, d_out, n_blocks, combine_layer, nv, int(num_freqs), bool(include_input)) == \
        (55, 512, 512, 4, 5, 3, 4, 6, True) and not beta > 0


class GenericMlp:
    """ResnetFC parameters for the generic slow path (csrc/generic.hip; ABI v5): ANY configuration the reference's constructor
    accepts (resnetfc.py:72-127) -- d_hidden, n_blocks, combine_layer, d_in / d_latent / d_out, Softplus for beta > 0 -- chained by
    the library on the general exact-fp32 MFMA GEMM, one launch per layer.  No packing: the struct points at the parameter tensors
    (kept alive here)."""

    def __init__(self, sd, prefix="", combine_layer=1000, beta=0.0, num_freqs=6, freq_factor=6.28, include_input=True, d_latent=None):
        g = lambda k: _f32c(sd[prefix + k])
        n_blocks = len([k for k in sd if k.startswith(prefix + "blocks.") and k.endswith("fc_0.weight")])
        n_z = len([k for k in sd if k.startswith(prefix + "lin_z.") and k.endswith(".weight")])
        has_in = prefix + "lin_in.weight" in sd
        self._keep = {"lin_out_w": g("lin_out.weight"), "lin_out_b": g("lin_out.bias")}
        if has_in:
            self._keep.update({"lin_in_w": g("lin_in.weight"), "lin_in_b": g("lin_in.bias")})
        lists = {"fc0_w": [g(f"blocks.{i}.fc_0.weight") for i in range(n_blocks)],
                 "fc0_b": [g(f"blocks.{i}.fc_0.bias") for i in range(n_blocks)],
                 "fc1_w": [g(f"blocks.{i}.fc_1.weight") for i in range(n_blocks)],
                 "fc1_b": [g(f"blocks.{i}.fc_1.bias") for i in range(n_blocks)],
                 "lin_z_w": [g(f"lin_z.{i}.weight") for i in range(n_z)],
                 "lin_z_b": [g(f"lin_z.{i}.bias") for i in range(n_z)]}
        self._lists = lists
        for t in list(self._keep.values()) + [t for l in lists.values() for t in l]:
            _require_hip(t)
        if any(k.startswith(prefix + "blocks.") and ".shortcut." in k for k in sd):
            raise NotImplementedError("diner_amd: ResnetBlockFC shortcut layers (size_in != size_out) do not occur in ResnetFC (resnetfc.py:104)")
        p = _lib.DinerMlpParams()
        p.d_hidden, p.d_out = self._keep["lin_out_w"].shape[1], self._keep["lin_out_w"].shape[0]
        p.d_in = self._keep["lin_in_w"].shape[1] if has_in else 0
        # the latent width is the module's (resnetfc.py:140-142 slices zx by self.d_latent whether or not a lin_z layer exists: combine_layer = 0)
        p.d_latent = int(d_latent) if d_latent is not None else (lists["lin_z_w"][0].shape[1] if n_z else 0)
        p.n_blocks, p.combine_layer = n_blocks, int(min(combine_layer, 2 ** 30))
        p.num_freqs, p.include_input, p.freq_factor = int(num_freqs), int(bool(include_input)), float(freq_factor)
        if has_in:
            p.lin_in_w, p.lin_in_b = self._keep["lin_in_w"].data_ptr(), self._keep["lin_in_b"].data_ptr()
        p.lin_out_w, p.lin_out_b = self._keep["lin_out_w"].data_ptr(), self._keep["lin_out_b"].data_ptr()
        self._arrays = {}
        for name, ts in lists.items():
            arr = (C.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])
            self._arrays[name] = arr
            setattr(p, name, C.cast(arr, C.POINTER(C.c_void_p)))
        self.params = p
        self.beta = float(beta)
        self.device = self._keep["lin_out_w"].device
        self.d_in, self.d_latent, self.d_out, self.d_hidden = p.d_in, p.d_latent, p.d_out, p.d_hidden
        self.combines = 0 <= p.combine_layer < p.n_blocks        # views averaged inside the network
        self.num_freqs, self.include_input, self.freq_factor = int(num_freqs), bool(include_input), float(freq_factor)

    def forward(self, zx):
        """zx (NV, B, d_latent + d_in) -> (B, d_out) when the views are combined inside the network, else (NV, B, d_out)."""
        _require_hip(zx)
        zx = _f32c(zx)
        NV, B, D = zx.shape
        if D != self.d_latent + self.d_in:
            raise ValueError(f"diner_amd: ResnetFC input width {D} != d_latent + d_in = {self.d_latent + self.d_in}")
        out = torch.empty((B, self.d_out) if self.combines else (NV, B, self.d_out), device=zx.device, dtype=torch.float32)
        if B == 0:
            return out
        with torch.cuda.device(zx.device):
            ws = _workspace(lib.diner_mlp_generic_workspace_bytes(C.byref(self.params), NV, B), zx.device)
            _lib.check(lib.diner_mlp_generic_forward_f32(C.byref(self.params), self.beta, _ptr(zx), NV, B, _ptr(out), _ptr(ws), _stream()))
        return out


GENERIC_POINTS_PER_LAUNCH = 1 << 18      # bounds the (NV, P, d_latent + d_in) matrix of the generic path (2.4 GB at NV 4 x 567 floats)


def field_generic(scene: HipScene, mlp: GenericMlp, rays=None, z=None, xyz=None, viewdirs=None):
    """PixelNeRF.forward on the generic slow path: (rays (NR,8), z (NR,K)) or (xyz, viewdirs) (P,3) -> (P,4) [r, g, b, sigma]
    (sigmoid / relu applied, pixelnerf.py:139-143).  The network must combine its views (combine_layer < n_blocks), as PixelNeRF's does."""
    if not mlp.combines:
        raise NotImplementedError("diner_amd: PixelNeRF needs an MLP that combines its views (combine_layer < n_blocks)")
    per = 2 * mlp.num_freqs + (1 if mlp.include_input else 0)
    if mlp.d_in != 4 * per + 3 or mlp.d_latent != scene.C or mlp.d_out != 4:
        raise ValueError(f"diner_amd: MLP d_in {mlp.d_in} / d_latent {mlp.d_latent} / d_out {mlp.d_out} do not match the positional "
                         f"encoding ({4 * per + 3} inputs), the scene's latent width ({scene.C}) and 4 outputs")
    if rays is not None:
        _require_hip(rays, z)
        rays, z = _f32c(rays), _f32c(z)
        NR, K = z.shape
        P, dev = NR * K, rays.device
    else:
        _require_hip(xyz, viewdirs)
        xyz, viewdirs = _f32c(xyz), _f32c(viewdirs)
        P, K, dev = xyz.shape[0], 1, xyz.device
    out = torch.empty(P, 4, device=dev, dtype=torch.float32)
    if P == 0:
        return out
    D = mlp.d_latent + mlp.d_in
    step = GENERIC_POINTS_PER_LAUNCH if rays is None else max(1, GENERIC_POINTS_PER_LAUNCH // K) * K
    with torch.cuda.device(dev):
        for p0 in range(0, P, step):
            p1 = min(P, p0 + step)
            n = p1 - p0
            zx = torch.empty(scene.nv, n, D, device=dev, dtype=torch.float32)
            if rays is not None:
                r0, r1 = p0 // K, p1 // K
                _lib.check(lib.diner_field_inputs_generic_f32(scene.ref, _ptr(rays[r0:r1]), _ptr(z[r0:r1]), K, None, None, n, mlp.num_freqs,
                                                              int(mlp.include_input), mlp.freq_factor, _ptr(zx), _stream()))
            else:
                _lib.check(lib.diner_field_inputs_generic_f32(scene.ref, None, None, 0, _ptr(xyz[p0:p1]), _ptr(viewdirs[p0:p1]), n,
                                                              mlp.num_freqs, int(mlp.include_input), mlp.freq_factor, _ptr(zx), _stream()))
            raw = mlp.forward(zx)
            _lib.check(lib.diner_field_act_f32(_ptr(raw), None, n, 4, _ptr(out[p0:p1]), _stream()))
    return out


# ---------------------------------------------------------------------------------------------------------
def sample_depthguided(scene: HipScene, rays, n_samples, n_candidates, n_gaussian, depth_diff_max=0.05,
                       noise=None, seed=0, want_unfilled=False, ray_index0=0):
    """rays (NR,8) -> ascending z (NR,K) [, unfilled z with zeros].  noise = (coarse, gauss, fill) or None
    (in-kernel Philox keyed by (`seed`, ray_index0 + i): pass the index of rays[0] in the frame's ray list and one seed per
    frame, and the frame does not depend on how its rays are batched or sharded)."""
    _require_hip(rays)
    rays = _f32c(rays)
    NR = rays.shape[0]
    K, G = int(n_samples), int(n_gaussian)
    z = torch.empty(NR, K, device=rays.device, dtype=torch.float32)
    zu = torch.empty(NR, K, device=rays.device, dtype=torch.float32) if want_unfilled else None
    if NR == 0:                     # nothing to do (the C ABI rejects empty launches)
        return (z, zu) if want_unfilled else z
    nc = ng = nf = None
    if noise is not None:
        nc, ng, nf = (_f32c(t) if t is not None else None for t in noise)
        _require_hip(nc, ng, nf)
        assert nc is None or tuple(nc.shape) == (NR, n_candidates)
        assert ng is None or tuple(ng.shape) == (NR, G)
        assert nf is None or tuple(nf.shape) == (NR, K)
    with torch.cuda.device(rays.device):
        _lib.check(lib.diner_sample_depthguided_f32(
            scene.ref, _ptr(rays), NR, int(n_candidates), K, G, float(depth_diff_max),
            _ptr(_t_base(int(n_candidates), rays.device)), _ptr(nc), _ptr(ng), _ptr(nf),
            C.c_uint64(int(seed) & (2 ** 64 - 1)), int(ray_index0), _ptr(z), _ptr(zu), _stream()))
    return (z, zu) if want_unfilled else z


def fill_uniform(z_in, rays, noise_fill=None, seed=0, ray_index0=0):
    _require_hip(z_in, rays, noise_fill)
    z_in, rays = _f32c(z_in), _f32c(rays)
    NR, K = z_in.shape
    out = torch.empty_like(z_in)
    if NR == 0:
        return out
    nf = _f32c(noise_fill) if noise_fill is not None else None
    with torch.cuda.device(rays.device):
        _lib.check(lib.diner_fill_uniform_f32(_ptr(z_in), _ptr(rays), NR, K, _ptr(nf),
                                              C.c_uint64(int(seed) & (2 ** 64 - 1)), int(ray_index0), _ptr(out), _stream()))
    return out


def _workspace(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def field_from_rays(scene: HipScene, mlp: HipMlp, rays, z, precision=None):
    """PixelNeRF.forward at every (ray, sample): (NR,8),(NR,K) -> (NR,K,4) [sigmoid rgb, relu sigma].
    precision: PRECISION_* for this call (default: get_precision())."""
    _require_hip(rays, z)
    rays, z = _f32c(rays), _f32c(z)
    NR, K = z.shape
    prec = _precision_for(scene, precision)
    scene.prepare(mlp, f16=prec == PRECISION_F16)
    out = torch.empty(NR, K, 4, device=rays.device, dtype=torch.float32)
    if NR == 0:
        return out
    rays_per = max(1, MAX_POINTS_PER_LAUNCH // K)
    with torch.cuda.device(rays.device):
        ws = _workspace(lib.diner_field_workspace_bytes(min(NR, rays_per) * K), rays.device)
        for r0 in range(0, NR, rays_per):
            r1 = min(NR, r0 + rays_per)
            _lib.check(lib.diner_field_from_rays_f32(scene.ref, mlp.handle, _ptr(rays[r0:r1]), _ptr(z[r0:r1]),
                                                     r1 - r0, K, prec, _ptr(out[r0:r1]), _ptr(ws), _stream()))
    return out


def field_from_points(scene: HipScene, mlp: HipMlp, xyz, viewdirs, precision=None):
    """PixelNeRF.forward(xyz, viewdirs): (P,3),(P,3) -> (P,4)."""
    _require_hip(xyz, viewdirs)
    xyz, viewdirs = _f32c(xyz), _f32c(viewdirs)
    P = xyz.shape[0]
    prec = _precision_for(scene, precision)
    scene.prepare(mlp, f16=prec == PRECISION_F16)
    out = torch.empty(P, 4, device=xyz.device, dtype=torch.float32)
    if P == 0:
        return out
    step = MAX_POINTS_PER_LAUNCH
    with torch.cuda.device(xyz.device):
        ws = _workspace(lib.diner_field_workspace_bytes(min(P, step)), xyz.device)
        for p0 in range(0, P, step):
            p1 = min(P, p0 + step)
            _lib.check(lib.diner_field_from_points_f32(scene.ref, mlp.handle, _ptr(xyz[p0:p1]), _ptr(viewdirs[p0:p1]),
                                                       p1 - p0, prec, _ptr(out[p0:p1]), _ptr(ws), _stream()))
    return out


def mlp_forward(mlp: HipMlp, zx):
    """ResnetFC.forward on an explicit (NV,B,567) matrix -> raw (B,4)."""
    _require_hip(zx)
    zx = _f32c(zx)
    NV, B, D = zx.shape
    if NV != 4 or D != 567:
        raise ValueError(f"diner_amd: fused ResnetFC is built for (4, B, 567) inputs, got {tuple(zx.shape)}")
    out = torch.empty(B, 4, device=zx.device, dtype=torch.float32)
    if B == 0:
        return out
    with torch.cuda.device(zx.device):
        ws = _workspace(lib.diner_mlp_forward_workspace_bytes(B), zx.device)
        _lib.check(lib.diner_mlp_forward_f32(mlp.handle, _ptr(zx), B, _ptr(out), _ptr(ws), _stream()))
    return out


def composite(field, z, rays, white_bkgd, want_weights=True):
    """(NR,K,4),(NR,K),(NR,8) -> weights (NR,K) | None, rgb (NR,3), depth (NR)."""
    _require_hip(field, z, rays)
    field, z, rays = _f32c(field), _f32c(z), _f32c(rays)
    NR, K = z.shape
    rgb = torch.empty(NR, 3, device=z.device, dtype=torch.float32)
    depth = torch.empty(NR, device=z.device, dtype=torch.float32)
    w = torch.empty(NR, K, device=z.device, dtype=torch.float32) if want_weights else None
    if NR == 0:
        return w, rgb, depth
    with torch.cuda.device(z.device):
        _lib.check(lib.diner_composite_f32(_ptr(field), _ptr(z), _ptr(rays), NR, K, int(bool(white_bkgd)),
                                           _ptr(rgb), _ptr(depth), _ptr(w), _stream()))
    return w, rgb, depth


def render(scene: HipScene, mlp: HipMlp, rays, z, white_bkgd, want_weights=False, precision=None):
    """field + composite (NeRFRendererDGS.composite): -> weights | None, rgb, depth."""
    if isinstance(mlp, GenericMlp):        # a configuration outside the fused kernels: exact fp32, one GEMM launch per layer
        NR, K = z.shape
        field = field_generic(scene, mlp, rays=rays, z=z).view(NR, K, 4)
    else:
        field = field_from_rays(scene, mlp, rays, z, precision=precision)
    return composite(field, z, rays, white_bkgd, want_weights)


def posenc(x, num_freqs, freq_factor, include_input=True):
    _require_hip(x)
    shp = x.shape
    xf = _f32c(x).reshape(-1, shp[-1])
    d_out = shp[-1] * (2 * num_freqs + (1 if include_input else 0))
    out = torch.empty(xf.shape[0], d_out, device=x.device, dtype=torch.float32)
    if xf.shape[0] == 0:
        return out.reshape(*shp[:-1], d_out)
    with torch.cuda.device(x.device):
        _lib.check(lib.diner_posenc_f32(_ptr(xf), xf.shape[0], shp[-1], int(num_freqs), float(freq_factor),
                                        int(bool(include_input)), _ptr(out), _stream()))
    return out.reshape(*shp[:-1], d_out)


INDEX_LATENT, INDEX_DEPTH, INDEX_DEPTH_STD, INDEX_NORMAL = 0, 1, 2, 3


def index(scene: HipScene, mode, uv):
    """uv (NV,N,2) -> (NV,Cout,N)."""
    _require_hip(uv)
    uv = _f32c(uv)
    NV, N, _ = uv.shape
    cout = {0: scene.C, 1: 1, 2: 1, 3: 3}[mode]
    out = torch.empty(NV, cout, N, device=uv.device, dtype=torch.float32)
    if N == 0:
        return out
    with torch.cuda.device(uv.device):
        _lib.check(lib.diner_index_f32(scene.ref, int(mode), _ptr(uv), N, _ptr(out), _stream()))
    return out


def depth2normal(dmap, K):
    """Reference src/util/depth2normal.py:7-87 on the device: dmap (N,1,H,W), K (N,3,3) -> normals (N,3,H,W)."""
    _require_hip(dmap, K)
    dmap, K = _f32c(dmap), _f32c(K)
    N, one, H, W = dmap.shape
    if one != 1 or tuple(K.shape) != (N, 3, 3):
        raise ValueError(f"diner_amd: depth2normal expects dmap (N,1,H,W) and K (N,3,3), got {tuple(dmap.shape)}, {tuple(K.shape)}")
    out = torch.empty(N, 3, H, W, device=dmap.device, dtype=torch.float32)
    with torch.cuda.device(dmap.device):
        _lib.check(lib.diner_depth2normal_f32(_ptr(dmap), _ptr(K), N, H, W, _ptr(out), _stream()))
    return out


def gen_rays(extrinsics, intrinsics, W, H, z_near, z_far, device, ray0=0, n_rays=None):
    """Reference src/util/cam_geometry.py:5-48 on the device: rays [ray0, ray0+n_rays) of each camera's row-major
    (H, W) list -> (B, n_rays, 8).  Camera tensors may live anywhere (they are read on the host: B x 27 floats)."""
    E = extrinsics.detach().to("cpu", torch.float32).contiguous()
    Km = intrinsics.detach().to("cpu", torch.float32).contiguous()
    B = E.shape[0]
    zn = torch.as_tensor(z_near, dtype=torch.float32).detach().to("cpu").reshape(-1).expand(B).contiguous()
    zf = torch.as_tensor(z_far, dtype=torch.float32).detach().to("cpu").reshape(-1).expand(B).contiguous()
    if tuple(E.shape) != (B, 4, 4) or tuple(Km.shape) != (B, 3, 3):
        raise ValueError(f"diner_amd: gen_rays expects (B,4,4) extrinsics and (B,3,3) intrinsics, got {tuple(E.shape)}, {tuple(Km.shape)}")
    n = int(W) * int(H) - int(ray0) if n_rays is None else int(n_rays)
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("diner_amd: gen_rays generates on a HIP device; there is no CPU fallback")
    out = torch.empty(B, n, 8, device=device, dtype=torch.float32)
    with torch.cuda.device(device):
        for b0 in range(0, B, 16):
            b1 = min(B, b0 + 16)
            _lib.check(lib.diner_gen_rays_f32(E[b0:b1].data_ptr(), Km[b0:b1].data_ptr(), zn[b0:b1].data_ptr(),
                                              zf[b0:b1].data_ptr(), b1 - b0, int(W), int(H), int(ray0), n,
                                              _ptr(out[b0:b1]), _stream()))
    return out


# FLOPs of the two field kernels per sample point: NV views x (lin_in + 3 x (lin_z, fc_0, fc_1))
# before the view mean, 2 x (fc_0, fc_1) + lin_out after it.
FLOP_PRE_PER_POINT_REFERENCE = 2 * 4 * (55 * 512 + 9 * 512 * 512)     # as the reference computes it
FLOP_PRE_PER_POINT = 2 * 4 * (55 * 512 + 6 * 512 * 512)               # executed: lin_z hoisted to once per pixel
FLOP_HOIST_PER_PIXEL = 2 * 3 * 512 * 512
FLOP_POST_PER_POINT = 2 * (4 * 512 * 512 + 512 * 4)


def profile_enable(flag=True):
    _lib.check(lib.diner_profile_enable(int(bool(flag))))


def profile_collect():
    """-> dict(pre_ms, post_ms, launches, points): summed HIP-event durations of k_field_pre / k_field_post."""
    a, b, n, p = C.c_double(), C.c_double(), C.c_longlong(), C.c_longlong()
    _lib.check(lib.diner_profile_collect(C.byref(a), C.byref(b), C.byref(n), C.byref(p)))
    return dict(pre_ms=a.value, post_ms=b.value, launches=n.value, points=p.value)


# ---- arithmetic of the MLP GEMMs: a per-call argument of the C ABI (DINER_PRECISION_* of include/diner_hip.h) --------
PRECISION_FP32, PRECISION_F16X3, PRECISION_F16 = 0, 1, 3      # DINER_PRECISION_* (2 is retired and rejected)
PRECISION_NAMES = {"fp32": PRECISION_FP32, "f32": PRECISION_FP32, "exact": PRECISION_FP32,
                   "f16x3": PRECISION_F16X3, "f16x3n": PRECISION_F16X3, "split": PRECISION_F16X3,
                   "f16": PRECISION_F16, "fp16": PRECISION_F16, "half": PRECISION_F16}
_default_precision = [PRECISION_F16X3]
_warned_big_map = False


def set_precision(mode):
    """Default `precision` of the field calls of THIS Python host (the library has no global switch; every C call carries
    its mode).  PRECISION_FP32: exact fp32 MFMA.  PRECISION_F16X3 (default): split products on the fp16 MFMA with fp32
    accumulation, fp32-class accuracy (3e-6 end to end against the reference; every parity test holds it to the fp32
    bar).  PRECISION_F16: plain fp16 operands (BASELINE configs[4], "fp16 MLP on MFMA"): ~1e-3 relative, NOT inside the
    1e-4 parity bar.  Env: DINER_AMD_PRECISION = fp32 | f16x3 | f16."""
    if isinstance(mode, str):
        mode = PRECISION_NAMES[mode.lower()]
    if isinstance(mode, bool) or int(mode) not in (PRECISION_FP32, PRECISION_F16X3, PRECISION_F16):
        raise ValueError(f"diner_amd: unknown precision {mode!r} (use the names 'fp32' / 'f16x3' / 'f16' or the PRECISION_* "
                         f"constants; the integer 2 of ABI v1 is retired)")
    _default_precision[0] = int(mode)


def get_precision():
    return _default_precision[0]


def _precision_for(scene, precision):
    """Mode passed to the library for one call.  A scene whose projected maps exceed the 32-bit addressing of the
    fp16-operand kernels (>= 4 GiB per map: images beyond ~2700 x 2700) is rendered by the exact kernels."""
    prec = get_precision() if precision is None else (PRECISION_NAMES[precision.lower()] if isinstance(precision, str)
                                                      else int(precision))
    if prec not in (PRECISION_FP32, PRECISION_F16X3, PRECISION_F16):
        raise ValueError(f"diner_amd: unknown precision {precision!r}")
    if prec != PRECISION_FP32 and scene.nv * scene.Hf * scene.Wf * 2048 >= (1 << 32):
        global _warned_big_map
        if not _warned_big_map:       # not silent: the exact kernels are ~3x slower than the f16x3 ones
            import warnings
            warnings.warn(f"diner_amd: one projected feature map of this scene is {scene.nv * scene.Hf * scene.Wf * 2048 / 2 ** 30:.1f} GiB; the "
                          f"fp16-operand kernels address it with 32-bit offsets (< 4 GiB), so this scene is rendered by the exact-fp32 "
                          f"kernels (~3x slower)", RuntimeWarning, stacklevel=3)
            _warned_big_map = True
        return PRECISION_FP32
    return prec


_want = os.environ.get("DINER_AMD_PRECISION", "f16x3").lower()
if _want not in PRECISION_NAMES:
    raise ValueError(f"DINER_AMD_PRECISION={_want!r}: expected one of {sorted(PRECISION_NAMES)}")
set_precision(_want)
