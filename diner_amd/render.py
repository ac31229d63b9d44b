"""Image harness: the MI355X counterpart of DINER.predict_imgs_from_batch (reference src/models/diner.py:72-97).

The reference renders one target image by splitting its H*W rays into batches of `ray_batch_size` and calling
`renderer.forward` on each (diner.py:85-92).  Rays are independent, so here the row-major ray list is additionally
sharded into contiguous ranges across the GPUs of a node (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm), each rank renders its range with the HIP kernels, and ONE gather of the packed (rgb, depth) tiles
(16 B/ray) brings the image to rank 0.  There is no other collective on the data path: scene state and MLP weights
are replicated (every rank runs `encode` itself).
"""
import contextlib

import torch

from diner_amd import noise as _noise
from src.util.cam_geometry import gen_rays


def shard_range(n, rank, world):
    """Contiguous, balanced partition of range(n): rank r gets [lo, hi)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def host_staged(group=None):
    """True when the process group cannot move device memory itself: backend "gloo" (ranks that share one GPU -- RCCL
    refuses two ranks on one device --, or a box without RCCL).  The two collectives of the path (tile gather, 8-byte seed
    broadcast) then go through pinned host buffers; with "nccl" (= RCCL) they run on device tensors over xGMI."""
    import torch.distributed as dist
    return "nccl" not in str(dist.get_backend(group))


def gather_tiles(local, n_total, rank, world, group=None, force=False):
    """Gather per-rank (n_r, C) tiles (contiguous ray ranges, see shard_range) to rank 0 -> (n_total, C) or None.

    Uses equal-size padded buffers so that a single gather collective suffices.  `force`: go through the collective even with one rank
    (bench.py --force-dist: exercises RCCL on a 1-GPU box)."""
    if world == 1 and not force:
        return local
    import torch.distributed as dist
    per = (n_total + world - 1) // world
    dev = local.device
    staged = local.is_cuda and host_staged(group)
    if staged:                       # 16 B/ray through pinned host memory (7.7 MB per 800x600 frame over all ranks)
        buf = torch.zeros(per, local.shape[1], dtype=local.dtype).pin_memory()
        buf[:local.shape[0]].copy_(local, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
    else:
        buf = local
        if local.shape[0] != per:
            buf = torch.zeros(per, local.shape[1], device=dev, dtype=local.dtype)
            buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf.contiguous(), out, dst=0, group=group)
    if rank != 0:
        return None
    full = torch.cat(out, dim=0)[:n_total]
    return full.to(dev, non_blocking=False) if staged else full


@torch.no_grad()
def predict_image(nerf, renderer, target_extrinsics, target_intrinsics, W, H, znear, zfar, ray_batch_size=8192,
                  rank=0, world=1, group=None, seed=None):
    """Render the (SB) target views described by target_extrinsics (SB,4,4) / target_intrinsics (SB,3,3) of the
    scene last passed to nerf.encode().  Returns rgb (SB,3,H,W), depth (SB,1,H,W) on rank 0 (None elsewhere).
    Same ray order (row-major pixels, centres at +0.5) and output layout as diner.py:79-92.  Noise injected with
    diner_amd.noise.inject for the whole (SB, H*W, .) ray list is handed to every batch as the matching slice (parity
    tests); without injection the sampler draws in-kernel Philox noise keyed by (`seed`, position of the ray in the frame):
    the image does not depend on ray_batch_size or on the number of ranks.  `seed` None: drawn from torch's global CPU
    generator on rank 0 and, with more than one rank, broadcast (8 bytes) so that all shards belong to the same frame."""
    SB = target_extrinsics.shape[0]
    dev = target_extrinsics.device
    znear = torch.as_tensor(znear, device=dev, dtype=torch.float32).expand(SB)
    zfar = torch.as_tensor(zfar, device=dev, dtype=torch.float32).expand(SB)
    lo, hi = shard_range(H * W, rank, world)
    if dev.type == "cuda":         # this rank's ray range only, generated on the device (diner_gen_rays_f32)
        from diner_amd import ops
        rays = ops.gen_rays(target_extrinsics, target_intrinsics, W, H, znear, zfar, dev, ray0=lo, n_rays=hi - lo)
        base = lo
    else:                          # host tensors (gloo tests): the reference's torch ops
        rays = gen_rays(target_extrinsics, target_intrinsics, W, H, znear, zfar).view(SB, H * W, 8)
        base = 0
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if world > 1:
            import torch.distributed as dist
            on_dev = dev.type == "cuda" and not host_staged(group)
            t = torch.tensor([seed], dtype=torch.int64, device=dev if on_dev else "cpu")
            dist.broadcast(t, src=0, group=group)
            seed = int(t.item())
    tiles = []
    inj = _noise.current()
    for r0 in range(lo, hi, ray_batch_size):
        r1 = min(hi, r0 + ray_batch_size)
        rb = rays[:, r0 - base:r1 - base].contiguous()
        ctx = contextlib.nullcontext() if inj is None else _noise.inject(*(None if t is None else t[:, r0:r1] for t in inj))
        with ctx, _noise.keyed(seed, r0):
            out = renderer.forward(model=nerf, rays=rb)
        tiles.append(torch.cat((out.fine.rgb, out.fine.depth.unsqueeze(-1)), dim=-1))      # (SB, b, 4)
    local = torch.cat(tiles, dim=1) if tiles else torch.zeros(SB, 0, 4, device=dev)
    full = gather_tiles(local.permute(1, 0, 2).reshape(hi - lo, SB * 4), H * W, rank, world, group)
    if full is None:
        return None, None
    full = full.view(H, W, SB, 4).permute(2, 3, 0, 1)                                       # (SB,4,H,W)
    return full[:, :3].contiguous(), full[:, 3:4].contiguous()
