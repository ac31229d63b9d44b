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
