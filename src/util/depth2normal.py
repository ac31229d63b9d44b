"""Normal maps from depth maps by central differences (mirrors reference src/util/depth2normal.py:7-87).

Encode-side preparation (once per image).  Depth maps on a HIP device go through the
library's kernel (diner_depth2normal_f32); host tensors are processed with the same torch ops the reference uses
(that is the form the oracle-side fixtures were checked in, bit-exact against the reference).  The result is an
*input* of the depth-guided sampler.
"""
import torch
import torch.nn.functional as F


@torch.no_grad()
def depth2normal(dmap, K):
    """dmap (N,1,H,W), K (N,3,3) -> normals (N,3,H,W); zero where depth == 0.

    Steps (reference lines in brackets): back-project pixel centres [:23-31], replicate-pad by 1 [:32],
    cross(down-up, right-left) normalised [:46-55], for pixels with a background (depth 0) neighbour
    copy the normal of the pixel shifted AWAY from the hole [:57-78], zero the background [:79].
    """
    if dmap.is_cuda:
        from diner_amd import ops
        return ops.depth2normal(dmap, K.to(dmap.device))
    N, _, H, W = dmap.shape
    dev = dmap.device
    ys, xs = torch.meshgrid(torch.arange(0.5, H, 1.0, device=dev), torch.arange(0.5, W, 1.0, device=dev),
                            indexing="ij")
    rays = torch.stack((xs, ys), dim=-1).reshape(1, -1, 2).expand(N, -1, -1).clone()
    rays -= K[:, [0, 1], -1].unsqueeze(-2)
    rays /= K[:, [0, 1], [0, 1]].unsqueeze(-2)
    rays = torch.cat((rays, torch.ones_like(rays[..., -1:])), dim=-1)
    pts = (rays.view(N, H, W, 3) * dmap.view(N, H, W, 1)).permute(0, 3, 1, 2)
    pts = F.pad(pts, [1] * 4, mode="replicate")
    down, up = pts[:, :, 2:, 1:-1], pts[:, :, :-2, 1:-1]
    right, left = pts[:, :, 1:-1, 2:], pts[:, :, 1:-1, :-2]
    vdiff = (down - up).permute(0, 2, 3, 1)
    hdiff = (right - left).permute(0, 2, 3, 1)
    normal = torch.linalg.cross(vdiff, hdiff, dim=-1)
    normal = normal / torch.norm(normal, p=2, dim=-1, keepdim=True)

    # hole clean-up: shift = (+1 if the upper neighbour is background) + (-1 if the lower one is), same for columns
    dy = (up[:, 0] == 0).long() - (down[:, 0] == 0).long()
    dx = (left[:, 0] == 0).long() - (right[:, 0] == 0).long()
    moved = (dy != 0) | (dx != 0)
    n_i, y_i, x_i = torch.where(moved)
    src_y = (y_i + dy[moved]).clamp(0, H - 1)
    src_x = (x_i + dx[moved]).clamp(0, W - 1)
    # the reference performs this as one gather-then-scatter (reads see the un-patched map)
    normal[n_i, y_i, x_i] = normal[n_i, src_y, src_x]
    normal[dmap[:, 0] == 0] = 0
    return normal.permute(0, 3, 1, 2)
