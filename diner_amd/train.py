"""Training path of the radiance field and the compositor.

The reference trains by differentiating its torch renderer with autograd (DINER.calc_losses, diner.py:217-290): the loss
touches `fine.rgb` only, gradients flow through the compositor (nerf_renderer.py:341-360), the output activations
(pixelnerf.py:139-143), ResnetFC (resnetfc.py:129-159) and the bilinear latent lookup (image_encoder.py:97-146) into the
MLP parameters and the encoder's feature maps; sample positions carry no gradient (`sample_depthguided` is `@no_grad`).

Here that is two `torch.autograd.Function`s whose forward AND backward are ONE call into libdiner_hip.so each
(diner_field_train_forward_f32 / _backward_f32, csrc/train.hip, sequence the kernels on the C side; torch only owns the buffers):
  * the 13 layer products with a 512 x 512 weight matrix -- forward, data gradient, weight gradient -- on persistent
    one-wave-per-SIMD kernels (csrc/train_lin512.hip, train_wgrad512.hip, entry points in train_512.hip) in the f16x3 arithmetic of the
    inference kernels (two fp16 planes per operand, three MFMA terms, fp32 accumulation); loss gradients are staged times a power of
    two taken from their maximum, operands outside the fp16 split fall to bf16x6 twins (three bf16 planes, six terms) without a host
    synchronisation;
  * lin_in, lin_out, the view mean, the latent gather / scatter and the output activations on small dedicated kernels; a general
    split-bf16 GEMM (`gemm`, `_linear`, `_linear_backward` below drive it from Python; the tests use them) for everything ragged.
The forward (round 5) runs on the inference path's fused f16x3 kernels in their STORING variants (k_train_fwd_pre / k_train_fwd_post,
diner_field_train_forward_fused_f32): activations stay on chip between the layers and every pre-activation the backward needs is written
once; an activation beyond the fp16 range raises a flag and the layer-wise forward enqueued behind the fused kernels (gated on the flag,
no host synchronisation) redoes the object; weights outside the fp16 split go to the layer-wise forward at once
(diner_field_train_forward_f32: one product per launch; DINER_TRAIN_FUSED_FWD=0 makes it the only one).  Sizes: the shipped configs train SB = 4 objects x 4096 rays (a 64 x 64 patch: w_vgg != 0, diner.py:57) x 40 samples x
4 views = 655 k columns per object and step (configs/train_dtu.yaml:16,52-63); the workspace of saved activations is 94 KB per sample
point = 15.4 GiB per object at that size (diner_field_train_workspace_bytes); round 5: 10.8 GiB of it are what the backward reads (kept per
object by autograd, four alive between forward and backward), the other 4.6 GiB are work buffers shared by the objects of a step.
Round 6 (ABI v6): the SB objects of a step are ONE autograd node (FieldBatchFunction -> diner_field_train_forward_batch_f32 / _backward_batch_f32:
object-major rows in one workspace, the backward's layer products once over all objects' rows); the packed-weights handle is persistent and re-packed
on the stream (diner_mlp_update) -- no host synchronisation per step; configurations outside the fused kernels train on the generic exact-fp32 path
(GenericMlpFunction, field_train_generic); release_buffers() frees what the path keeps between steps.
"""
import os

import torch

from . import _lib
from .ops import HipScene, _ptr, _stream, _require_hip, _f32c

lib = _lib.load()

TA, TB, RELU_A, RELU_B, ACCUM, ATOMIC, EXACT = 1, 2, 4, 8, 16, 32, 64


def gemm(A, B, C, M, N, K, lda, ldb, ldc, flags=0, bias=None, mask=None, k_split=1):
    """C (M x N) = op(A) . op(B) on the matrix cores, see include/diner_hip.h::diner_gemm_f32."""
    _lib.check(lib.diner_gemm_f32(_ptr(A), _ptr(B), _ptr(C), int(M), int(N), int(K), int(lda), int(ldb), int(ldc),
                                  int(flags), _ptr(bias), _ptr(mask), int(k_split), _stream()))


def linear512(x, W, out, transpose=False, relu_in=False, accumulate=False, bias=None, resid=None, mask=None, f16x3=False):
    """out (M, 512) (+)= act(x (M, 512)) op(W) (+ bias) (+ resid) [* (mask > 0)] on the feature-sliced training kernel
    (csrc/train_lin512.hip): transpose False = x W^T (nn.Linear forward), True = x W (data gradient).  f16x3: the forward product in the
    inference kernels' arithmetic (two fp16 planes per operand, |x| < 65504; no fall-back at this level)."""
    ws = torch.empty(lib.diner_linear512_pack_bytes(), dtype=torch.uint8, device=x.device)
    _lib.check(lib.diner_linear512_f32(_ptr(x), _ptr(W), _ptr(out), int(x.shape[0]), int(x.stride(0)), int(out.stride(0)),
                                       int(bool(transpose)), (1 if relu_in else 0) | (2 if accumulate else 0) | (4 if f16x3 else 0), _ptr(bias), _ptr(resid),
                                       _ptr(mask), _ptr(ws), _stream()))
    return out


def wgrad512(dy, x, dW, db=None, relu_in=False, scratch=None):
    """dW (512, 512) += dy^T act(x), db (512) += column sums of dy on the persistent training kernel (csrc/train_wgrad512.hip);
    both are accumulated (zero them first).  scratch: None = the row chunks add into dW with atomics; True = allocate, or a uint8 tensor
    of diner_wgrad512_scratch_bytes(): per-chunk partial tiles + one summing pass (what the training step does)."""
    if scratch is True:
        scratch = torch.empty(lib.diner_wgrad512_scratch_bytes(), dtype=torch.uint8, device=dy.device)
    _lib.check(lib.diner_wgrad512_f32(_ptr(dy), _ptr(x), _ptr(dW), _ptr(db), int(dy.shape[0]), int(dy.stride(0)), int(x.stride(0)),
                                      int(bool(relu_in)), _ptr(scratch), _stream()))
    return dW


def _linear(x, W, b, out=None, relu_in=False, accumulate=False):
    """torch.nn.Linear on row-major (M, K) activations: out (M, N) (+)= act(x) W^T + b, K = W.shape[1] <= x row stride."""
    M, N, K = x.shape[0], W.shape[0], W.shape[1]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    gemm(x, W, out, M, N, K, x.stride(0), W.stride(0), out.stride(0),
         TB | (RELU_A if relu_in else 0) | (ACCUM if accumulate else 0), bias=b)
    return out


def _linear_backward(dy, x, W, relu_in, dx_out=None, dx_mask=None, dx_accumulate=False, need_dx=True):
    """Adjoint of out = act(x) W^T + b.  Returns (dW, db, dx): dW (N,K) = dy^T act(x) (split-K, atomics), db = column sums,
    dx (M,K) (+)= (dy W) [* (dx_mask > 0)]."""
    M, N, K = dy.shape[0], W.shape[0], W.shape[1]
    dW = torch.zeros_like(W)
    split = max(1, min(64, M // 640))          # same rule as csrc/train.hip linear_backward
    gemm(dy, x, dW, N, K, M, dy.stride(0), x.stride(0), dW.stride(0), TA | ATOMIC | (RELU_B if relu_in else 0), k_split=split)
    db = torch.zeros(N, device=dy.device, dtype=torch.float32)
    _lib.check(lib.diner_colsum_f32(_ptr(dy), M, N, dy.stride(0), _ptr(db), _stream()))
    dx = None
    if need_dx:
        dx = dx_out if dx_out is not None else torch.empty(M, K, device=dy.device, dtype=torch.float32)
        gemm(dy, W, dx, M, K, N, dy.stride(0), W.stride(0), dx.stride(0), ACCUM if dx_accumulate else 0, mask=dx_mask)
    return dW, db, dx


PARAM_ORDER = (["lin_in.weight", "lin_in.bias"] + [f"lin_z.{b}.{n}" for b in range(3) for n in ("weight", "bias")] +
               [f"blocks.{b}.{l}.{n}" for b in range(5) for l in ("fc_0", "fc_1") for n in ("weight", "bias")] +
               ["lin_out.weight", "lin_out.bias"])


def mlp_params(mlp_module):
    """The 30 parameter tensors of src.models.resnetfc.ResnetFC in PARAM_ORDER."""
    sd = dict(mlp_module.named_parameters())
    return [sd[k] for k in PARAM_ORDER]


def _param_struct(tensors, freq_factor=6.28):
    """DinerMlpParams over 30 device tensors in PARAM_ORDER (parameters or gradient buffers); returns (struct, keep-alive)."""
    import ctypes as C
    t = dict(zip(PARAM_ORDER, tensors))
    p = _lib.DinerMlpParams()
    p.d_in, p.d_hidden, p.d_out = t["lin_in.weight"].shape[1], t["lin_in.weight"].shape[0], t["lin_out.weight"].shape[0]
    p.d_latent, p.n_blocks, p.combine_layer = t["lin_z.0.weight"].shape[1], 5, 3
    p.num_freqs, p.include_input, p.freq_factor = 6, 1, float(freq_factor)
    p.lin_in_w, p.lin_in_b = t["lin_in.weight"].data_ptr(), t["lin_in.bias"].data_ptr()
    p.lin_out_w, p.lin_out_b = t["lin_out.weight"].data_ptr(), t["lin_out.bias"].data_ptr()
    keep = [tensors]
    for field, fmt, n in (("fc0_w", "blocks.{}.fc_0.weight", 5), ("fc0_b", "blocks.{}.fc_0.bias", 5),
                          ("fc1_w", "blocks.{}.fc_1.weight", 5), ("fc1_b", "blocks.{}.fc_1.bias", 5),
                          ("lin_z_w", "lin_z.{}.weight", 3), ("lin_z_b", "lin_z.{}.bias", 3)):
        arr = (C.c_void_p * n)(*[t[fmt.format(i)].data_ptr() for i in range(n)])
        keep.append(arr)
        setattr(p, field, C.cast(arr, C.POINTER(C.c_void_p)))
    return p, keep


def fused_forward_enabled(P=None, scene=None):
    """DINER_TRAIN_FUSED_FWD: 1 = always, 0 = never, unset = by size (from 16384 sample points per object on)."""
    if scene is not None and scene.nv != 4:          # the fused kernels are built for four source views (the layer-wise forward: any)
        return False
    e = os.environ.get("DINER_TRAIN_FUSED_FWD", "")
    if e in ("0", "1"):
        return e == "1"
    # round 6: the projection now costs the TOUCHED texel rows, not the map (csrc/train.hip, k_mark_rows), so the map size is out of the rule:
    # 800 x 600 maps 32.6 ms fused against 38.1 layer-wise per 4096-ray object; what is left is the fixed cost of the fused kernels' launches at
    # tiny batches -- 128 rays x 40 samples (5120 points): 13.0 ms fused against 12.4 layer-wise per four-object step
    # (profiles/r06_train_touched_texel_projection.txt)
    return P is None or P >= 16384


_PROJ = {}


def _shared(pool, dev, nbytes, zero=False):
    """A buffer that lives inside one library call only, shared by the objects of a step (and the steps): one per device AND stream -- calls
    on a stream run one after the other; it only grows (the caching allocator keeps a replaced one alive for the work already enqueued).
    zero: zero-filled when (re)allocated."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = pool.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = pool[key] = (torch.zeros if zero else torch.empty)(int(nbytes), dtype=torch.uint8, device=dev)
    return buf


def _proj_buffer(scene, dev):
    """The latent map projected through lin_z[0..2] with the step's weights: written and read inside the forward call only."""
    # zero-filled once: round 6 projects only the texel rows a batch touches; a texel that another rounding of a tap's last bit would name is read
    # with a weight of ~1e-7 and must hold a finite value (zeros, later older projections), never an unwritten NaN
    return _shared(_PROJ, dev, int(lib.diner_scene_proj_bytes(scene.ref)), zero=True)


_SCRATCH = {}


def workspace_split(P, nv):
    """(saved bytes, scratch bytes) of the training workspace (diner_field_train_workspace_split): the saved part is what autograd keeps per
    object between forward and backward, the scratch part is shared (`_SCRATCH`)."""
    import ctypes as C
    a, b = C.c_size_t(0), C.c_size_t(0)
    _lib.check(lib.diner_field_train_workspace_split(P, nv, C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


_STEP_MLP = {}


def _step_handle(params, freq_factor):
    """The persistent packed-weights handle of the training path: ONE per device and stream, created at the first step (diner_mlp_create: a
    dozen hipMallocs and one stream wait, once) and from then on re-packed in place ON THE STREAM with the parameter values of every call
    (diner_mlp_update: no allocation, no host synchronisation; round 5 re-created a handle per parameter version -- hipMalloc / hipFree and a
    stream wait per optimiser step: 103 of 136 ms of host time).  Nothing is cached per parameter address or version (`_version`
    misses writes through `p.data`, a freed model's addresses can be recycled): the values are packed again for every call, ~40 small
    launches = 0.2 ms.  `release_buffers()` drops the handles."""
    from .ops import HipMlp
    dev = params[0].device
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    h = _STEP_MLP.get(key)
    if h is None:
        _STEP_MLP[key] = h = HipMlp(dict(zip(PARAM_ORDER, params)), freq_factor=float(freq_factor))
    return h


def _step_mlp(params, freq_factor):
    """... re-packed (training subset) with this call's parameter values: what the per-object fused forward takes."""
    h = _step_handle(params, freq_factor)
    h.update(dict(zip(PARAM_ORDER, params)), freq_factor=float(freq_factor), train_only=True)
    return h


def release_buffers():
    """Frees what the training path keeps between steps: the shared work buffers (`_SCRATCH`: 4.6 GiB per stream at one object of 4096 rays
    x 40 samples, 17 GiB for the batched step of four), the projected-map buffers (`_PROJ`) and the persistent packed-weights handles.  Call it
    when leaving training for, e.g., a validation render in the same process; the next step allocates them again."""
    _SCRATCH.clear()
    _PROJ.clear()
    _STEP_MLP.clear()


def _channels_last(latent):
    """True when a (..., C, Hf, Wf) latent already lies channels-last in memory (what PixelNeRF.encode of this repo emits on a HIP device: the
    pyramid is concatenated in that format, image_encoder.py).  HipScene then takes it without a copy, and the backward hands autograd the
    channels-last gradient as a view shaped like the latent instead of transposing it (round 6: 0.7 + 0.8 ms of the shipped step)."""
    return latent.dim() >= 3 and latent.movedim(-3, -1).is_contiguous()


def _latent_grad(d_cl, channels_last, n_img, latent_shape, dev):
    """channels-last gradient buffer (..., Hf, Wf, C) -> the gradient autograd takes for a latent of `latent_shape` (..., C, Hf, Wf)."""
    if channels_last:
        return d_cl.movedim(-1, -3)
    Cc, Hf, Wf = latent_shape[-3:]
    d_lat = torch.empty(*latent_shape, device=dev)
    _lib.check(lib.diner_channels_last_to_nchw_f32(_ptr(d_cl), n_img, Hf * Wf, Cc, _ptr(d_lat), _stream()))
    return d_lat


class FieldFunction(torch.autograd.Function):
    """PixelNeRF.forward (pixelnerf.py:55-145) for one object: (xyz, viewdirs) (P,3) -> (P,4) [sigmoid rgb, relu sigma],
    differentiable with respect to the encoder's latent (NV,512,Hf,Wf) and the MLP parameters.  One library call for the
    forward (keeps every pre-activation in `ws`), one for the backward (diner_field_train_{forward,backward}_f32)."""

    @staticmethod
    def _alloc_outputs(params, latent_shape, dev):
        """Gradient buffers of the 30 parameters + their DinerMlpParams struct + the channels-last latent gradient."""
        grads = [torch.empty_like(p) for p in params]
        nv, Cc, Hf, Wf = latent_shape
        return grads, _param_struct(grads), torch.empty(nv, Hf, Wf, Cc, device=dev)

    @staticmethod
    def forward(ctx, scene: HipScene, xyz, viewdirs, latent, freq_factor, *params):
        import ctypes as C
        _require_hip(xyz, viewdirs, latent)
        xyz, viewdirs = _f32c(xyz.detach()), _f32c(viewdirs.detach())
        params = [_f32c(p.detach()) for p in params]
        P, NV = xyz.shape[0], scene.nv
        dev = xyz.device
        with torch.cuda.device(dev):
            # round 5: autograd keeps the SAVED part of the workspace only (10.8 of 15.4 GiB per object at 4096 rays x 40 samples); the work
            # buffers of the two calls are one shared buffer per device and stream
            saved_bytes, scratch_bytes = workspace_split(P, NV)
            ws = torch.empty(saved_bytes, dtype=torch.uint8, device=dev)
            scratch = _shared(_SCRATCH, dev, scratch_bytes)
            out = torch.empty(P, 4, device=dev)
            ps, keep = _param_struct(params, freq_factor)
            done = False
            if fused_forward_enabled(P, scene):
                # round 5: the forward on the inference kernels' storing variants (activations stay on chip between the layers, the
                # backward's operands are written once); needs this step's packed weights and the latent projected with them
                mlp = _step_mlp(params, freq_factor)
                rc = lib.diner_field_train_forward_fused_f32(scene.ref, mlp.handle, C.byref(ps), _ptr(xyz), _ptr(viewdirs), P,
                                                             _ptr(out), _ptr(ws), _ptr(scratch), _ptr(_proj_buffer(scene, dev)), _stream())
                if rc != _lib.E_UNSUPPORTED:         # (weights outside the fp16 split, maps beyond 4 GiB: the layer-wise forward below)
                    _lib.check(rc)                   # (an activation beyond the fp16 range: the library's gated layer-wise repeat, on the device)
                    done = True
            if not done:
                _lib.check(lib.diner_field_train_forward_s_f32(scene.ref, C.byref(ps), _ptr(xyz), _ptr(viewdirs), P, _ptr(out),
                                                               _ptr(ws), _ptr(scratch), _stream()))
            # the backward's outputs are allocated here, while the device works on the forward: the host is idle now and is the one the
            # device waits for at the start of the backward (64 us of the reference batch's 3.9 ms step)
            ctx.ps = (ps, keep)
            ctx.latent_shape = tuple(latent.shape)
            ctx.latent_cl = _channels_last(latent)
            ctx.prealloc = FieldFunction._alloc_outputs(params, ctx.latent_shape, dev) if any(ctx.needs_input_grad) else None
        ctx.scene, ctx.P = scene, P
        ctx.save_for_backward(ws, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        import ctypes as C
        ws, params = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        with torch.cuda.device(d_out.device):
            # (a second backward through a retained graph gets fresh buffers: the first call's are the parameters' .grad by now)
            pre, ctx.prealloc = ctx.prealloc, None
            grads, (gs, keep_g), d_cl = pre if pre is not None else FieldFunction._alloc_outputs(params, ctx.latent_shape, d_out.device)
            if not ctx.needs_input_grad[3]:
                d_cl = None
            ps, keep = ctx.ps
            d_out = _f32c(d_out)
            scratch = _shared(_SCRATCH, d_out.device, workspace_split(ctx.P, ctx.scene.nv)[1])
            _lib.check(lib.diner_field_train_backward_s_f32(ctx.scene.ref, C.byref(ps), C.byref(gs), ctx.P, _ptr(d_out),
                                                            _ptr(ws), _ptr(scratch), _ptr(d_cl), _stream()))
            d_lat = None
            if d_cl is not None:             # channels-last -> the encoder's (nv, C, Hf, Wf) in the latent's own memory format
                d_lat = _latent_grad(d_cl, ctx.latent_cl, ctx.latent_shape[0], ctx.latent_shape, d_out.device)
        return (None, None, None, d_lat, None) + tuple(grads)


class FieldBatchFunction(torch.autograd.Function):
    """PixelNeRF.forward for the SB objects of a training step in ONE library call pair (ABI v6, diner_field_train_forward_batch_f32 /
    _backward_batch_f32): xyz, viewdirs (SB, P, 3) -> (SB, P, 4).  The forward runs the fused storing kernels object by object into one
    workspace with object-major rows; the backward's layer products run once over SB x P x NV rows (the weight gradients of the step are
    summed inside the kernels, the latent gradients scattered per object).  No host synchronisation: the persistent packed-weights handle is
    re-packed on the stream inside the forward call."""

    @staticmethod
    def forward(ctx, scenes, xyz, viewdirs, latent, freq_factor, *params):
        import ctypes as C
        _require_hip(xyz, viewdirs, latent)
        xyz, viewdirs = _f32c(xyz.detach()), _f32c(viewdirs.detach())
        params = [_f32c(p.detach()) for p in params]
        SB, P, NV = xyz.shape[0], xyz.shape[1], scenes[0].nv
        dev = xyz.device
        with torch.cuda.device(dev):
            a, b = C.c_size_t(0), C.c_size_t(0)
            _lib.check(lib.diner_field_train_batch_workspace_split(P, NV, SB, C.byref(a), C.byref(b)))
            ws = torch.empty(int(a.value), dtype=torch.uint8, device=dev)
            scratch = _shared(_SCRATCH, dev, int(b.value))
            proj = _shared(_PROJ, dev, max(int(lib.diner_scene_proj_bytes(sc.ref)) for sc in scenes), zero=True)
            out = torch.empty(SB, P, 4, device=dev)
            ps, keep = _param_struct(params, freq_factor)
            mlp = _step_handle(params, freq_factor)
            arr = (C.POINTER(_lib.DinerScene) * SB)(*[C.pointer(sc.struct) for sc in scenes])
            rc = lib.diner_field_train_forward_batch_f32(arr, SB, mlp.handle, C.byref(ps), _ptr(xyz), _ptr(viewdirs), P, _ptr(out), _ptr(ws),
                                                         _ptr(scratch), _ptr(proj), _stream())
            if rc == _lib.E_UNSUPPORTED:
                raise _BatchUnsupported()
            _lib.check(rc)
            mlp._range = None
            ctx.ps = (ps, keep)
            ctx.latent_shape = tuple(latent.shape)
            ctx.latent_cl = _channels_last(latent)
            ctx.prealloc = FieldBatchFunction._alloc_outputs(params, ctx.latent_shape, dev) if any(ctx.needs_input_grad) else None
        ctx.scenes, ctx.P, ctx.arr = list(scenes), P, arr
        ctx.save_for_backward(ws, *params)
        return out

    @staticmethod
    def _alloc_outputs(params, latent_shape, dev):
        grads = [torch.empty_like(p) for p in params]
        sb, nv, Cc, Hf, Wf = latent_shape
        return grads, _param_struct(grads), torch.empty(sb, nv, Hf, Wf, Cc, device=dev)

    @staticmethod
    def backward(ctx, d_out):
        import ctypes as C
        ws, params = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        dev = d_out.device
        SB = len(ctx.scenes)
        with torch.cuda.device(dev):
            pre, ctx.prealloc = ctx.prealloc, None
            grads, (gs, keep_g), d_cl = pre if pre is not None else FieldBatchFunction._alloc_outputs(params, ctx.latent_shape, dev)
            want_lat = ctx.needs_input_grad[3]
            ps, keep = ctx.ps
            d_out = _f32c(d_out)
            a, b = C.c_size_t(0), C.c_size_t(0)
            _lib.check(lib.diner_field_train_batch_workspace_split(ctx.P, ctx.scenes[0].nv, SB, C.byref(a), C.byref(b)))
            scratch = _shared(_SCRATCH, dev, int(b.value))
            dl = (C.c_void_p * SB)(*[d_cl[o].data_ptr() if want_lat else None for o in range(SB)])
            # the forward's projection buffer is free in the backward: one map-shaped plane of it carries the lin_z adjoint in map space
            proj = _shared(_PROJ, dev, max(int(lib.diner_scene_proj_bytes(sc.ref)) for sc in ctx.scenes), zero=True)
            _lib.check(lib.diner_field_train_backward_batch_f32(ctx.arr, SB, C.byref(ps), C.byref(gs), ctx.P, _ptr(d_out), _ptr(ws), _ptr(scratch),
                                                                dl, _ptr(proj), _stream()))
            d_lat = None
            if want_lat:                     # channels-last -> the encoder's (SB, nv, C, Hf, Wf) in the latent's own memory format
                d_lat = _latent_grad(d_cl, ctx.latent_cl, ctx.latent_shape[0] * ctx.latent_shape[1], ctx.latent_shape, dev)
        return (None, None, None, d_lat, None) + tuple(grads)


class _BatchUnsupported(Exception):
    pass


def batch_enabled(P, scenes):
    """The batched step (one call pair for the SB objects) runs on the fused forward: same conditions as fused_forward_enabled for every
    object, same number of views and map sizes (the latent is one stacked tensor).  DINER_TRAIN_BATCH=0: one call pair per object (round 5)."""
    if os.environ.get("DINER_TRAIN_BATCH", "") == "0":
        return False
    s0 = scenes[0]
    return all(fused_forward_enabled(P, sc) and (sc.nv, sc.Hf, sc.Wf) == (s0.nv, s0.Hf, s0.Wf) for sc in scenes)


def field_train_batch(scenes, xyz, viewdirs, latent, params, freq_factor=6.28):
    """(SB, P, 3) x 2 -> (SB, P, 4), differentiable with respect to latent (SB, NV, C, Hf, Wf) and the MLP parameters; falls back to one
    call pair per object when the library declines the fused forward (weights outside the fp16 split, maps of 4 GiB or more)."""
    if batch_enabled(xyz.shape[1], scenes):
        try:
            return FieldBatchFunction.apply(list(scenes), xyz, viewdirs, latent, float(freq_factor), *params)
        except _BatchUnsupported:
            pass
    slabs = object_slabs(latent)
    return torch.stack([field_train(scenes[sb], xyz[sb], viewdirs[sb], slabs[sb], params, freq_factor) for sb in range(len(scenes))])


class CompositeFunction(torch.autograd.Function):
    """Compositing arithmetic of NeRFRendererDGS.composite (nerf_renderer.py:299-301, :341-360): field (NR,K,4), z (NR,K),
    rays (NR,8) -> rgb (NR,3), depth (NR); differentiable with respect to the field."""

    @staticmethod
    def forward(ctx, field, z, rays, white_bkgd):
        from . import ops
        field, z, rays = _f32c(field.detach()), _f32c(z.detach()), _f32c(rays.detach())
        _, rgb, depth = ops.composite(field, z, rays, white_bkgd, want_weights=False)
        ctx.save_for_backward(field, z, rays)
        ctx.white = bool(white_bkgd)
        return rgb, depth

    @staticmethod
    def backward(ctx, g_rgb, g_depth):
        field, z, rays = ctx.saved_tensors
        NR, K = z.shape
        d_field = torch.empty_like(field)
        with torch.cuda.device(field.device):
            g_rgb = _f32c(g_rgb) if g_rgb is not None else torch.zeros(NR, 3, device=field.device)
            gd = _f32c(g_depth) if g_depth is not None else None
            _lib.check(lib.diner_composite_bwd_f32(_ptr(field), _ptr(z), _ptr(rays), NR, K, int(ctx.white), _ptr(g_rgb),
                                                   _ptr(gd), _ptr(d_field), _stream()))
        return d_field, None, None, None


class _ObjectSlabs(torch.autograd.Function):
    """latent (SB, NV, C, Hf, Wf) -> its SB per-object slabs as separate autograd inputs of the per-object field nodes.  Plain indexing
    (`latent[sb]`) makes autograd build each object's gradient as a zero-filled tensor of the WHOLE latent with one slab copied in and then
    add SB of those (at SB = 4 and 400 x 300 images: four 1.85 GB zero-fills + three 1.85 GB additions per step, ~5 ms of HBM traffic);
    here the backward stacks the SB slab gradients once."""

    @staticmethod
    def forward(ctx, latent):
        ctx.sb = latent.shape[0]
        return tuple(latent[sb] for sb in range(latent.shape[0]))

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        return torch.stack([g if g is not None else torch.zeros_like(ref) for g in grads])


def object_slabs(latent):
    """Per-object views of encoder.latent for the training path (see _ObjectSlabs); without grad: plain views."""
    if torch.is_grad_enabled() and latent.requires_grad:
        return _ObjectSlabs.apply(latent)
    return tuple(latent[sb] for sb in range(latent.shape[0]))


# ---- training through the generic slow path (ABI v6): any ResnetFC / PixelNeRF configuration the reference's constructors accept
class GenericMlpFunction(torch.autograd.Function):
    """ResnetFC.forward on an explicit (NV, B, d_latent + d_in) matrix (resnetfc.py:129-159), differentiable with respect to the matrix and
    the parameters: forward and backward are one library call each (diner_mlp_generic_train_forward_f32 / diner_mlp_generic_backward_f32:
    the layers and their adjoints chained on the exact-fp32 MFMA GEMM)."""

    @staticmethod
    def forward(ctx, conf, names, zx, *params):
        import ctypes as C
        from .ops import GenericMlp
        _require_hip(zx)
        zx = _f32c(zx.detach())
        params = [_f32c(p.detach()) for p in params]
        gm = GenericMlp(dict(zip(names, params)), **conf)
        NV, B, D = zx.shape
        if D != gm.d_latent + gm.d_in:
            raise ValueError(f"diner_amd: ResnetFC input width {D} != d_latent + d_in = {gm.d_latent + gm.d_in}")
        out = torch.empty((B, gm.d_out) if gm.combines else (NV, B, gm.d_out), device=zx.device)
        with torch.cuda.device(zx.device):
            ws = torch.empty(int(lib.diner_mlp_generic_train_workspace_bytes(C.byref(gm.params), NV, B)), dtype=torch.uint8, device=zx.device)
            _lib.check(lib.diner_mlp_generic_train_forward_f32(C.byref(gm.params), gm.beta, _ptr(zx), NV, B, _ptr(out), _ptr(ws), _stream()))
        ctx.gm, ctx.conf, ctx.names = gm, conf, names
        ctx.save_for_backward(ws, zx, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        import ctypes as C
        from .ops import GenericMlp
        ws, zx, params = ctx.saved_tensors[0], ctx.saved_tensors[1], list(ctx.saved_tensors[2:])
        NV, B, D = zx.shape
        grads = [torch.empty_like(p) for p in params]
        gg = GenericMlp(dict(zip(ctx.names, grads)), **ctx.conf)
        d_zx = torch.empty_like(zx) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(zx.device):
            _lib.check(lib.diner_mlp_generic_backward_f32(C.byref(ctx.gm.params), C.byref(gg.params), ctx.gm.beta, _ptr(zx), NV, B,
                                                          _ptr(_f32c(d_out)), _ptr(ws), _ptr(d_zx), _stream()))
        return (None, None, d_zx) + tuple(grads)


def generic_mlp_train(mlp_module, zx):
    """ResnetFC.forward(zx (NV, B, C)) in grad mode for ANY configuration (the fused shape included: an explicit matrix has no scene to gather
    from, so it runs on the generic exact-fp32 path)."""
    names = [k for k, _ in mlp_module.named_parameters()]
    conf = dict(combine_layer=mlp_module.combine_layer, beta=mlp_module.beta, d_latent=mlp_module.d_latent)
    return GenericMlpFunction.apply(conf, names, zx, *[p for _, p in mlp_module.named_parameters()])


class _GenericInputs(torch.autograd.Function):
    """(xyz, viewdirs) -> the per-view MLP input rows of PixelNeRF.forward for any encoding / latent width (pixelnerf.py:84-128), differentiable
    with respect to the encoder's latent (the bilinear lookup's adjoint, diner_field_inputs_generic_bwd_f32); the geometry carries no gradient."""

    @staticmethod
    def forward(ctx, scene, xyz, viewdirs, latent, num_freqs, include_input, freq_factor, d_row):
        xyz, viewdirs = _f32c(xyz.detach()), _f32c(viewdirs.detach())
        P = xyz.shape[0]
        zx = torch.empty(scene.nv, P, d_row, device=xyz.device)
        with torch.cuda.device(xyz.device):
            _lib.check(lib.diner_field_inputs_generic_f32(scene.ref, None, None, 0, _ptr(xyz), _ptr(viewdirs), P, int(num_freqs), int(include_input),
                                                          float(freq_factor), _ptr(zx), _stream()))
        ctx.scene, ctx.d_row, ctx.latent_shape = scene, d_row, tuple(latent.shape)
        ctx.latent_cl = _channels_last(latent)
        ctx.save_for_backward(xyz, viewdirs)
        return zx

    @staticmethod
    def backward(ctx, d_zx):
        xyz, viewdirs = ctx.saved_tensors
        d_lat = None
        if ctx.needs_input_grad[3]:
            nv, Cc, Hf, Wf = ctx.latent_shape
            with torch.cuda.device(xyz.device):
                d_cl = torch.empty(nv, Hf, Wf, Cc, device=xyz.device)
                _lib.check(lib.diner_field_inputs_generic_bwd_f32(ctx.scene.ref, _ptr(xyz), _ptr(viewdirs), xyz.shape[0], ctx.d_row, _ptr(_f32c(d_zx)),
                                                                  _ptr(d_cl), _stream()))
                d_lat = _latent_grad(d_cl, ctx.latent_cl, nv, ctx.latent_shape, xyz.device)
        return (None, None, None, d_lat, None, None, None, None)


class _FieldAct(torch.autograd.Function):
    """raw (P, 4) -> [sigmoid(rgb), relu(sigma)] (pixelnerf.py:139-143) and its adjoint (diner_field_act_f32)."""

    @staticmethod
    def forward(ctx, raw):
        raw = _f32c(raw.detach())
        out = torch.empty_like(raw)
        with torch.cuda.device(raw.device):
            _lib.check(lib.diner_field_act_f32(_ptr(raw), None, raw.shape[0], 4, _ptr(out), _stream()))
        ctx.save_for_backward(raw)
        return out

    @staticmethod
    def backward(ctx, d_out):
        raw, = ctx.saved_tensors
        d_raw = torch.empty_like(raw)
        with torch.cuda.device(raw.device):
            _lib.check(lib.diner_field_act_f32(_ptr(raw), _ptr(_f32c(d_out)), raw.shape[0], 4, _ptr(d_raw), _stream()))
        return d_raw


def field_train_generic(scene: HipScene, mlp_module, xyz, viewdirs, latent, num_freqs, include_input, freq_factor):
    """PixelNeRF.forward for one object in grad mode on the generic path: (P, 3) x 2 -> (P, 4), differentiable with respect to the latent
    (NV, C, Hf, Wf) and the MLP parameters (any d_hidden / n_blocks / combine_layer < n_blocks / encoding / latent width / NV <= 4 / Softplus)."""
    per = 2 * int(num_freqs) + (1 if include_input else 0)
    d_row = scene.C + 4 * per + 3
    zx = _GenericInputs.apply(scene, xyz, viewdirs, latent, num_freqs, include_input, freq_factor, d_row)
    raw = generic_mlp_train(mlp_module, zx)
    if raw.dim() != 2 or raw.shape[-1] != 4:
        raise NotImplementedError("diner_amd: PixelNeRF needs an MLP that combines its views (combine_layer < n_blocks) and has 4 outputs")
    return _FieldAct.apply(raw)


def field_train(scene: HipScene, xyz, viewdirs, latent, params, freq_factor=6.28):
    return FieldFunction.apply(scene, xyz, viewdirs, latent, float(freq_factor), *params)


def composite_train(field, z, rays, white_bkgd):
    return CompositeFunction.apply(field, z, rays, white_bkgd)
