"""Training path of the radiance field and the compositor.

The reference trains by differentiating its torch renderer with autograd (DINER.calc_losses, diner.py:217-290): the loss
touches `fine.rgb` only, gradients flow through the compositor (nerf_renderer.py:341-360), the output activations
(pixelnerf.py:139-143), ResnetFC (resnetfc.py:129-159) and the bilinear latent lookup (image_encoder.py:97-146) into the
MLP parameters and the encoder's feature maps; sample positions carry no gradient (`sample_depthguided` is `@no_grad`).

Here that is two `torch.autograd.Function`s whose forward AND backward are sequences of calls into libdiner_hip.so
(csrc/train.hip: one fp32 MFMA GEMM with the needed epilogues + small kernels); torch only owns the buffers.  The forward
is the un-fused one that keeps every pre-activation, which is what a backward pass needs; inference keeps using the fused
kernels.  Sizes: 128 rays x 40 samples x 4 views = 20 k columns per object and step (configs/train_dtu.yaml).
"""
import torch

from . import _lib
from .ops import HipScene, _ptr, _stream, _require_hip, _f32c

lib = _lib.load()

TA, TB, RELU_A, RELU_B, ACCUM, ATOMIC = 1, 2, 4, 8, 16, 32


def gemm(A, B, C, M, N, K, lda, ldb, ldc, flags=0, bias=None, mask=None, k_split=1):
    """C (M x N) = op(A) . op(B) on the matrix cores, see include/diner_hip.h::diner_gemm_f32."""
    _lib.check(lib.diner_gemm_f32(_ptr(A), _ptr(B), _ptr(C), int(M), int(N), int(K), int(lda), int(ldb), int(ldc),
                                  int(flags), _ptr(bias), _ptr(mask), int(k_split), _stream()))


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
    split = max(1, min(32, M // 1024))
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


class FieldFunction(torch.autograd.Function):
    """PixelNeRF.forward (pixelnerf.py:55-145) for one object: (xyz, viewdirs) (P,3) -> (P,4) [sigmoid rgb, relu sigma],
    differentiable with respect to the encoder's latent (NV,512,Hf,Wf) and the MLP parameters."""

    @staticmethod
    def forward(ctx, scene: HipScene, xyz, viewdirs, latent, *params):
        _require_hip(xyz, viewdirs, latent)
        xyz, viewdirs = _f32c(xyz.detach()), _f32c(viewdirs.detach())
        params = [_f32c(p.detach()) for p in params]
        P, NV = xyz.shape[0], scene.nv
        cols = NV * P
        dev = xyz.device
        w_in, b_in = params[0], params[1]
        wz = [(params[2 + 2 * b], params[3 + 2 * b]) for b in range(3)]
        blk = [tuple(params[8 + 4 * b: 12 + 4 * b]) for b in range(5)]
        w_out, b_out = params[-2], params[-1]
        with torch.cuda.device(dev):
            feat = torch.empty(cols, 64, device=dev)
            tap_row = torch.empty(cols, 4, device=dev, dtype=torch.int32)
            tap_w = torch.empty(cols, 4, device=dev)
            lat = torch.empty(cols, 512, device=dev)
            _lib.check(lib.diner_train_inputs_f32(scene.ref, _ptr(xyz), _ptr(viewdirs), P, _ptr(feat), _ptr(tap_row),
                                                  _ptr(tap_w), _ptr(lat), _stream()))
            x = _linear(feat, w_in, b_in)                                        # resnetfc.py:143
            X, Hh = [], []
            for b in range(5):
                if b == 3:                                                       # combine_interleaved (:150-152)
                    xm = torch.empty(P, 512, device=dev)
                    _lib.check(lib.diner_view_mean_f32(_ptr(x), NV, P * 512, _ptr(xm), 0, _stream()))
                    x = xm
                if b < 3:
                    _linear(lat, wz[b][0], wz[b][1], out=x, accumulate=True)     # x = x + lin_z[b](z) (:153-155)
                w0, b0, w1, b1 = blk[b]
                h = _linear(x, w0, b0, relu_in=True)                             # fc_0(relu(x))   (resnetfc.py:61-69)
                X.append(x)
                Hh.append(h)
                x = x.clone()
                _linear(h, w1, b1, out=x, relu_in=True, accumulate=True)         # x + fc_1(relu(h))
            raw = _linear(x, w_out, b_out, relu_in=True)                         # lin_out(relu(x)) (:157)
            out = torch.empty(P, 4, device=dev)
            _lib.check(lib.diner_field_act_f32(_ptr(raw), None, P, 4, _ptr(out), _stream()))
        ctx.scene, ctx.P, ctx.NV = scene, P, NV
        ctx.latent_shape = tuple(latent.shape)
        ctx.save_for_backward(feat, tap_row, tap_w, lat, x, raw, *X, *Hh, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        saved = ctx.saved_tensors
        feat, tap_row, tap_w, lat, x_last, raw = saved[:6]
        X, Hh, params = list(saved[6:11]), list(saved[11:16]), list(saved[16:])
        P, NV = ctx.P, ctx.NV
        cols = NV * P
        dev = d_out.device
        w_in = params[0]
        wz = [params[2 + 2 * b] for b in range(3)]
        blk = [tuple(params[8 + 4 * b: 12 + 4 * b]) for b in range(5)]
        w_out = params[-2]
        grads = [None] * len(params)
        with torch.cuda.device(dev):
            d_out = _f32c(d_out)
            d_raw = torch.empty(P, 4, device=dev)
            _lib.check(lib.diner_field_act_f32(_ptr(raw), _ptr(d_out), P, 4, _ptr(d_raw), _stream()))
            grads[-2], grads[-1], dx = _linear_backward(d_raw, x_last, w_out, relu_in=True, dx_mask=x_last)
            d_lat = None
            for b in range(4, -1, -1):
                w0, _, w1, _ = blk[b]
                # x_out = X + fc_1(relu(H)),  H = fc_0(relu(X))
                gw1, gb1, dH = _linear_backward(dx, Hh[b], w1, relu_in=True, dx_mask=Hh[b])
                gw0, gb0, _ = _linear_backward(dH, X[b], w0, relu_in=True, dx_out=dx, dx_mask=X[b], dx_accumulate=True)
                grads[8 + 4 * b: 12 + 4 * b] = [gw0, gb0, gw1, gb1]
                if b < 3:                                                        # X = x_prev + lin_z[b](lat)
                    if d_lat is None:
                        d_lat = torch.empty(cols, 512, device=dev)
                    gz, gbz, _ = _linear_backward(dx, lat, wz[b], relu_in=False, dx_out=d_lat, dx_accumulate=(b < 2))
                    grads[2 + 2 * b], grads[3 + 2 * b] = gz, gbz
                if b == 3:                                                       # adjoint of the view mean
                    dxv = torch.empty(cols, 512, device=dev)
                    _lib.check(lib.diner_view_mean_f32(_ptr(dx), NV, P * 512, _ptr(dxv), 1, _stream()))
                    dx = dxv
            grads[0], grads[1], _ = _linear_backward(dx, feat, w_in, relu_in=False, need_dx=False)
            d_latent = None
            if ctx.needs_input_grad[3]:
                nv, C, Hf, Wf = ctx.latent_shape
                d_cl = torch.zeros(nv, Hf, Wf, C, device=dev)
                _lib.check(lib.diner_scatter_latent_grad_f32(_ptr(d_lat), _ptr(tap_row), _ptr(tap_w), cols, _ptr(d_cl),
                                                             _stream()))
                d_latent = d_cl.permute(0, 3, 1, 2)
        return (None, None, None, d_latent) + tuple(grads)


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


def field_train(scene: HipScene, xyz, viewdirs, latent, params):
    return FieldFunction.apply(scene, xyz, viewdirs, latent, *params)


def composite_train(field, z, rays, white_bkgd):
    return CompositeFunction.apply(field, z, rays, white_bkgd)
