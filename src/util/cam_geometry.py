"""Ray generation (mirrors reference src/util/cam_geometry.py:5-48): pixel-centre rays of a pinhole camera in
the OpenCV convention, `[origin(3), direction(3), near, far]`, row-major over (H, W).

Not on the hot path (once per image).  Cameras on a HIP device generate their rays with the
library's kernel (diner_gen_rays_f32, also used per ray range by the sharded harness); host tensors use the reference's
torch ops.
"""
import torch


def gen_rays(extrinsics, intrinsics, W, H, z_near, z_far):
    """extrinsics (B,4,4) world->cam, intrinsics (B,3,3), z_near/z_far (B) -> rays (B,H,W,8)."""
    B = extrinsics.shape[0]
    dev = extrinsics.device
    if extrinsics.is_cuda:
        from diner_amd import ops
        return ops.gen_rays(extrinsics, intrinsics, W, H, z_near, z_far, dev).view(B, H, W, 8)
    focal = intrinsics[:, [0, 1], [0, 1]]
    c = intrinsics[:, [0, 1], [-1, -1]]
    ys, xs = torch.meshgrid(torch.arange(.5, H, 1, device=dev), torch.arange(.5, W, 1, device=dev), indexing="ij")
    screen = torch.stack((xs, ys), dim=-1).unsqueeze(0).expand(B, -1, -1, -1)            # (B,H,W,2) x,y
    cam = (screen - c.view(B, 1, 1, 2)) / focal.view(B, 1, 1, 2)
    cam = torch.cat((cam, torch.ones_like(cam[..., :1])), dim=-1)
    dirs_cam = cam / cam.pow(2).sum(dim=-1, keepdim=True).sqrt()
    Rc2w = extrinsics[:, :3, :3].permute(0, 2, 1)
    dirs_w = (Rc2w @ dirs_cam.view(B, -1, 3).permute(0, 2, 1)).permute(0, 2, 1).view(B, H, W, 3)
    origins = (-1 * Rc2w @ extrinsics[:, :3, -1:]).view(B, 1, 1, 3).expand(-1, H, W, -1)
    near = z_near.view(B, 1, 1, 1).expand(-1, H, W, -1)
    far = z_far.view(B, 1, 1, 1).expand(-1, H, W, -1)
    return torch.cat((origins, dirs_w, near, far), dim=-1)


# ---- camera-sweep helpers (reference src/util/cam_geometry.py:51-205; used by the datasets' get_cam_sweep_extrinsics and
# ---- by diner_amd.sweep).  Host-side numpy / torch, once per sweep.
def pose_spherical(theta, phi, radius):
    """Camera-to-world pose on a sphere (degrees), NeRF convention (:83-100): translate along +z by radius, rotate by phi about x,
    by theta about y, then swap axes (x -> -x, y <-> z)."""
    import math
    ph, th = phi / 180.0 * math.pi, theta / 180.0 * math.pi
    t = torch.eye(4)
    t[2, 3] = radius
    rp = torch.tensor([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1]],
                      dtype=torch.float32)
    rt = torch.tensor([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1]],
                      dtype=torch.float32)
    flip = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)
    return flip @ (rt @ (rp @ t))


def get_ray_intersections(ray1, ray2):
    """Closest points of two rays [origin(3), direction(3)] to each other (:103-121): least-squares solution of
    o1 + t1 d1 = o2 + t2 d2."""
    A = torch.stack((ray1[3:], -ray2[3:]), dim=-1)
    B = (ray2[:3] - ray1[:3]).unsqueeze(1)
    t = torch.linalg.lstsq(A, B).solution.flatten()
    return ray1[:3] + ray1[3:] * t[0], ray2[:3] + ray2[3:] * t[1]


def to_homogeneous_trafo(trafo):
    """(N,3,4) -> (N,4,4) with the row [0,0,0,1] appended (:124-130)."""
    last = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]]).expand(len(trafo), -1, -1)
    return torch.cat((trafo, last), dim=1)


class TransSlerp:
    """Piece-wise linear interpolation of locations over time, clamped at both ends (:156-205)."""

    def __init__(self, times, locations):
        import numpy as np
        order = np.argsort(times)
        self._times = np.asarray(times)[order]
        self._locations = np.asarray(locations)[order]

    def __call__(self, t_q):
        import numpy as np
        tq = np.clip(np.asarray(t_q, dtype=self._times.dtype), self._times.min(), self._times.max())
        hi = np.clip(np.searchsorted(self._times, tq, side="left"), 0, len(self._times) - 1)     # first fix time >= t
        lo = np.clip(np.searchsorted(self._times, tq, side="right") - 1, 0, len(self._times) - 1)  # last fix time <= t
        dt = np.clip(self._times[hi] - self._times[lo], 1e-4, None)
        w_lo = np.clip((self._times[hi] - np.asarray(t_q)) / dt, 0.0, 1.0)
        return self._locations[lo] * w_lo[:, None] + self._locations[hi] * (1.0 - w_lo)[:, None]


class Slerp:
    """scipy's rotation Slerp plus the location interpolation above (:132-154)."""

    def __init__(self, times, rotations, locations):
        from scipy.spatial.transform import Slerp as _RotSlerp
        self._rot = _RotSlerp(times, rotations)
        self._loc = TransSlerp(times, locations)

    def __call__(self, times):
        return self._rot(times), self._loc(times)
