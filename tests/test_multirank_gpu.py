"""GPU, world_size 2 and 8 on ONE MI355X: the N-rank path of the image harness (BASELINE configs[3]; the build's counterpart of the
serial ray-batch loop of the reference's diner.py:85-92) with the REAL renderer -- `src.models.*` drop-in modules, HIP kernels,
device ray generation of each rank's range, frame-keyed in-kernel noise, the seed broadcast and the single gather of the (rgb, depth)
tiles -- executed by N processes that share the one GPU `gpurun` offers.  RCCL refuses two ranks on one device, so the process group is
gloo and the two collectives are staged through pinned host memory (diner_amd.render.host_staged); on a multi-GPU node the same code runs
with backend "nccl" on device tensors.  Statement: the rank-0 image is BIT-EQUAL to the frame a single process renders (a ray's noise is
keyed by (frame seed, index of the ray in the frame), so the frame cannot depend on the number of ranks)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

W, H, K, G, N_CAND, SCENE_SEED, FRAME_SEED = 400, 300, 128, 48, 1000, 0, 20260928


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_MODEL = {}


def _render(rank, world, seed, reseed=None):
    from diner_amd.render import predict_image
    from tests.test_boundary_gpu import setup_model
    if "m" not in _MODEL:                          # one model per process (round 6: the 8-rank case built 16 of them on oversubscribed host
        _MODEL["m"] = setup_model(W, H, SCENE_SEED)    # threads and took 354 of the suite's 761 s)
    sc, nerf, R, _ = _MODEL["m"]
    ren = R(n_samples=K, n_depth_candidates=N_CAND, n_gaussian=G, white_bkgd=False)
    E, Km = sc["target_extrinsics"][None].cuda(), sc["target_intrinsics"][None].cuda()
    if reseed is not None:                         # (building the model consumes the global generator: seed it after that)
        torch.manual_seed(reseed)
    return predict_image(nerf, ren, E, Km, W, H, sc["znear"], sc["zfar"], ray_batch_size=8192 + 5, rank=rank, world=world, seed=seed)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(max(1, (os.cpu_count() or world) // world))      # N processes share the host: no N x all-threads oversubscription
    torch.cuda.set_device(0)                       # every rank on the same device
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rgb, depth = _render(rank, world, FRAME_SEED)
        # second frame without an explicit seed: rank 0 draws it from ITS generator, the broadcast makes it the frame's seed
        rgb2, depth2 = _render(rank, world, None, reseed=1000 + rank)
        if rank == 0:
            torch.manual_seed(1000)               # the seed rank 0 drew, for the parent's single-process frame
            drawn = int(torch.randint(0, 2 ** 62, (1,)).item())
            q.put(tuple(t.cpu().numpy() for t in (rgb, depth, rgb2, depth2)) + (drawn,))     # by value (numpy pickles the bytes)
        else:
            assert rgb is None and depth is None and rgb2 is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_sharing_one_gpu_render_the_single_process_frame(world):
    assert torch.cuda.is_available()
    ref_rgb, ref_d = _render(0, 1, FRAME_SEED)
    assert ref_rgb.shape == (1, 3, H, W) and ref_d.shape == (1, 1, H, W) and torch.isfinite(ref_rgb).all()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        *imgs, drawn = q.get(timeout=900)
        rgb, depth, rgb2, depth2 = (torch.from_numpy(a) for a in imgs)
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert torch.equal(rgb, ref_rgb.cpu()) and torch.equal(depth, ref_d.cpu()), f"{world}-rank frame differs from the single-process frame"
    ref2_rgb, ref2_d = _render(0, 1, drawn)
    assert torch.equal(rgb2, ref2_rgb.cpu()) and torch.equal(depth2, ref2_d.cpu()), "broadcast seed: shards of different frames"
    assert not torch.equal(rgb2, rgb)
