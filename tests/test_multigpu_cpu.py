"""CPU, world_size 2 over gloo: the ray-range sharding + single gather of the image harness assembles exactly the
image a single process renders (the N>1 path of diner_amd.render / bench.py; RCCL replaces gloo on the GPU box)."""
import os
import socket

import torch
import torch.multiprocessing as mp

from src.util.general import DotMap


class _FakeRenderer:
    """Deterministic per-ray stand-in for NeRFRendererDGS.forward (the sharding logic must not care what it is)."""

    def forward(self, model, rays):
        from diner_amd import noise
        seed, r0 = noise.frame_key()          # the harness keys every batch: (frame seed, position of the batch in the frame)
        pos = (r0 + torch.arange(rays.shape[1], dtype=torch.float32)).expand(rays.shape[0], -1)
        rgb = torch.stack((rays[..., 3], rays[..., 4] * 2, rays[..., 5] * 3), dim=-1)
        depth = rays[..., 3] + rays[..., 4] - rays[..., 5] + (seed % 1000) * 1e-3 + pos * 1e-2
        return DotMap(fine=DotMap(rgb=rgb, depth=depth))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, q):
    import torch.distributed as dist
    from diner_amd.render import predict_image
    from diner_amd.synthetic import look_at_extrinsics
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E = torch.stack([look_at_extrinsics((0.1, 0.0, -1.0)), look_at_extrinsics((-0.2, 0.05, -1.0))])
    K = torch.tensor([[1.2 * W, 0, W / 2], [0, 1.2 * W, H / 2], [0, 0, 1.0]]).repeat(2, 1, 1)
    rgb, depth = predict_image(None, _FakeRenderer(), E, K, W, H, 0.5, 1.5, ray_batch_size=37, rank=rank, world=world, seed=4242)
    # without an explicit seed rank 0 draws one and broadcasts it: both shards of the frame carry the same (seed % 1000) offset
    torch.manual_seed(100 + rank)             # different generators on the two ranks
    rgb_b, depth_b = predict_image(None, _FakeRenderer(), E, K, W, H, 0.5, 1.5, ray_batch_size=50, rank=rank, world=world)
    if rank == 0:
        pos = torch.arange(H * W, dtype=torch.float32).view(1, 1, H, W) * 1e-2
        geo = depth[:, :1] - 0.242 - pos                         # what the fake renderer computes from the rays alone
        off = depth_b[:, :1] - pos - geo
        assert float(off.max() - off.min()) < 1e-4, "the two ranks rendered their shards with different frame seeds"
        q.put((rgb, depth))
    else:
        assert rgb is None and depth is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_render_equals_single_process():
    from diner_amd.render import predict_image
    from diner_amd.synthetic import look_at_extrinsics
    W, H = 23, 17                                   # 391 rays: not divisible by 2 -> padded gather path
    E = torch.stack([look_at_extrinsics((0.1, 0.0, -1.0)), look_at_extrinsics((-0.2, 0.05, -1.0))])
    K = torch.tensor([[1.2 * W, 0, W / 2], [0, 1.2 * W, H / 2], [0, 0, 1.0]]).repeat(2, 1, 1)
    rgb1, d1 = predict_image(None, _FakeRenderer(), E, K, W, H, 0.5, 1.5, ray_batch_size=50, seed=4242)
    assert rgb1.shape == (2, 3, H, W) and d1.shape == (2, 1, H, W)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, q)) for r in range(2)]
    for p in procs:
        p.start()
    rgb2, d2 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(rgb1, rgb2) and torch.equal(d1, d2)
