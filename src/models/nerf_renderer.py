"""NeRFRendererDGS -- drop-in for reference src/models/nerf_renderer.py (:12-430): same constructor, mutable
attributes (`n_samples`, `n_gaussian`, `n_depth_candidates`, `eval_batch_size`, `white_bkgd`; read at call time, as
create_prediction_folder.py:44-47 relies on), public methods and return types.

Everything runs in hand-written HIP kernels behind the C ABI of include/diner_hip.h:
  sample_depthguided / fill_up_uniform_samples  -> diner_sample_depthguided_f32 / diner_fill_uniform_f32 (sampler.hip)
  composite                                     -> diner_field_from_rays_f32 (mlp.hip) + diner_composite_f32
  forward                                       -> the three above, per object of the batch
`model` must be the MI355X PixelNeRF of this package (it carries the channels-last scene and the packed weights).

Noise: the reference draws rand/randn from torch's global generator; here the kernels draw the same three noise
fields from an in-kernel Philox generator keyed by a seed taken from torch's global CPU generator, unless explicit
noise is injected with diner_amd.noise.inject (parity tests)."""
import torch

from diner_amd import noise as _noise
from diner_amd import ops
from src.util.general import DotMap


def _seed():
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def _key(sb=0):
    """(seed, ray_index0) of this call's in-kernel noise: the frame key set by the image harness (one seed per frame + the
    position of this ray batch in the frame, diner_amd.noise.keyed), else a fresh seed from torch's global generator."""
    k = _noise.frame_key()
    return (k[0] + 0x9E3779B97F4A7C15 * sb, k[1]) if k is not None else (_seed(), 0)


class NeRFRendererDGS(torch.nn.Module):
    def __init__(self, n_samples=40, n_depth_candidates=1000, n_gaussian=15, eval_batch_size=100000, white_bkgd=True):
        super().__init__()
        self.n_samples = n_samples
        self.n_depth_candidates = n_depth_candidates
        self.n_gaussian = n_gaussian
        self.eval_batch_size = eval_batch_size      # kept for API compatibility: chunking never changes results
        self.white_bkgd = white_bkgd

    @staticmethod
    def _check_model(model):
        if not (hasattr(model, "hip_scene") and hasattr(model, "hip_mlp")):
            raise TypeError("diner_amd: `model` must be src.models.pixelnerf.PixelNeRF of this package")

    def _render_train_batch(self, model, rays, z, want_weights):
        """The SB objects of a training step at once (ABI v6): rays (SB,NR,8), z (SB,NR,K) -> weights | None, rgb (SB,NR,3), depth (SB,NR).
        One field node for all objects (diner_amd.train.field_train_batch: the layer products of the backward run once over SB x NR x K x NV
        rows), one compositor node over the SB x NR rays."""
        from diner_amd import train
        SB, NR, K = z.shape
        r = rays.detach()
        z = z.detach()
        xyz = (r[:, :, None, :3] + z[..., None] * r[:, :, None, 3:6]).reshape(SB, NR * K, 3)
        dirs = r[:, :, None, 3:6].expand(-1, -1, K, -1).reshape(SB, NR * K, 3)
        if model.is_generic():          # a configuration outside the fused kernels: the generic differentiable path, object by object
            field = model.forward(xyz, dirs).view(SB * NR, K, 4)
        else:
            scenes = [model.hip_scene(sb) for sb in range(SB)]
            field = train.field_train_batch(scenes, xyz, dirs, model.encoder.latent, train.mlp_params(model.mlp_fine),
                                            model.poscode.freq_factor).view(SB * NR, K, 4)
        zf, rf = z.reshape(SB * NR, K), r.reshape(SB * NR, 8)
        rgb, depth = train.composite_train(field, zf, rf, self.white_bkgd)
        w = ops.composite(field.detach(), zf, rf, self.white_bkgd, want_weights=True)[0].view(SB, NR, K) if want_weights else None
        return w, rgb.view(SB, NR, 3), depth.view(SB, NR)

    def sample_coarse(self, rays, n_coarse=None):
        """Stratified candidates (:39-63) as a stand-alone helper (torch ops on the rays' device).  The depth-guided
        sampler generates its candidates inside the kernel and does not call this."""
        n_coarse = n_coarse if n_coarse else self.n_depth_candidates
        shp = rays.shape
        r = rays.reshape(-1, 8)
        near, far = r[:, -2:-1], r[:, -1:]
        step = 1.0 / n_coarse
        t = torch.linspace(0, 1 - step, n_coarse, device=rays.device).unsqueeze(0).repeat(r.shape[0], 1)
        t = t + torch.rand_like(t) * step
        return (near * (1 - t) + far * t).view(*shp[:-1], n_coarse)

    @torch.no_grad()
    def sample_depthguided(self, rays, model, n_samples, n_candidates, depth_diff_max=0.05, n_gaussian=None):
        """rays (SB,NR,8) -> z (SB,NR,n_samples): the top (n_samples - n_gaussian) candidates by surface likelihood,
        n_gaussian samples of the likelihood-weighted gaussian, zeros marking empty slots (:65-190).  Slot order
        within a ray is unspecified (the reference's is by descending likelihood); fill_up_uniform_samples sorts."""
        self._check_model(model)
        n_gaussian = n_gaussian if n_gaussian is not None else self.n_gaussian
        assert n_samples >= n_gaussian
        SB = rays.shape[0]
        inj = _noise.current()
        out = []
        for sb in range(SB):
            nz = None if inj is None else tuple(None if t is None else t[sb] for t in inj)
            seed, r0 = _key(sb)
            _, zu = ops.sample_depthguided(model.hip_scene(sb), rays[sb], n_samples, n_candidates, n_gaussian,
                                           depth_diff_max, noise=nz, seed=seed, want_unfilled=True, ray_index0=r0)
            out.append(zu)
        return torch.stack(out)

    def fill_up_uniform_samples(self, z_samples, rays):
        """zeros in z (SB,NR,K) -> stratified samples of [near, far]; returns ascending z (:367-397)."""
        SB = rays.shape[0]
        inj = _noise.current()
        out = []
        for sb in range(SB):
            seed, r0 = _key(sb)
            out.append(ops.fill_uniform(z_samples[sb], rays[sb], None if inj is None or inj[2] is None else inj[2][sb],
                                        seed=seed, ray_index0=r0))
        return torch.stack(out)

    def composite(self, model, rays, z_samp):
        """-> weights (SB,B,K), rgb (SB,B,3), depth (SB,B)   (:286-365)."""
        self._check_model(model)
        model._check_poscode()
        SB = rays.shape[0]
        if model.needs_grad():
            return self._render_train_batch(model, rays, z_samp, True)
        else:
            mlp = model.hip_mlp()
            res = [ops.render(model.hip_scene(sb), mlp, rays[sb], z_samp[sb], self.white_bkgd, want_weights=True)
                   for sb in range(SB)]
        return tuple(torch.stack([r[i] for r in res]) for i in range(3))

    def forward(self, model, rays, want_weights=False):
        """rays (SB,B,8) -> DotMap(fine=DotMap(rgb (SB,B,3), depth (SB,B) [, weights (SB,B,K)]))   (:399-430)."""
        assert len(rays.shape) == 3
        self._check_model(model)
        model._check_poscode()
        assert self.n_samples >= self.n_gaussian
        SB = rays.shape[0]
        training = model.needs_grad()
        mlp = None if training else model.hip_mlp()
        inj = _noise.current()
        rgbs, depths, wts = [], [], []
        if training:
            # the sampler per object (each has its own maps), then field + compositor for the SB objects as ONE autograd node each
            zs = []
            for sb in range(SB):
                nz = None if inj is None else tuple(None if t is None else t[sb] for t in inj)
                seed, r0 = _key(sb)
                zs.append(ops.sample_depthguided(model.hip_scene(sb), rays[sb], self.n_samples, self.n_depth_candidates, self.n_gaussian,
                                                 0.05, noise=nz, seed=seed, ray_index0=r0))
            w, rgb, depth = self._render_train_batch(model, rays, torch.stack(zs), want_weights)
            return DotMap(fine=self._format_outputs(w, rgb, depth, want_weights=want_weights))
        for sb in range(SB):
            scene = model.hip_scene(sb)
            nz = None if inj is None else tuple(None if t is None else t[sb] for t in inj)
            seed, r0 = _key(sb)
            z = ops.sample_depthguided(scene, rays[sb], self.n_samples, self.n_depth_candidates, self.n_gaussian,
                                       0.05, noise=nz, seed=seed, ray_index0=r0)
            w, rgb, depth = ops.render(scene, mlp, rays[sb], z, self.white_bkgd, want_weights=want_weights)
            rgbs.append(rgb)
            depths.append(depth)
            wts.append(w)
        return DotMap(fine=self._format_outputs(torch.stack(wts) if want_weights else None, torch.stack(rgbs),
                                                torch.stack(depths), want_weights=want_weights))

    def _format_outputs(self, weights, rgb, depth, want_weights):
        out = DotMap(rgb=rgb, depth=depth)
        if want_weights:
            out.weights = weights
        return out
