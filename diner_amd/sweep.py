"""Camera sweeps: the MI355X counterpart of DINER.create_cam_sweep (reference
src/models/diner.py:138-215) and of the datasets' get_cam_sweep_extrinsics (src/data/dtu.py:246-318).

A sweep renders `nframes` target views of ONE encoded scene along a path through three cameras (left, centre, right): the
camera centres move on two great-circle arcs about the point the three optical axes (nearly) meet in, the orientations are
slerp-interpolated.  Every frame is an ordinary image render (diner_amd.render.predict_image: ray-range sharding across the
GPUs of a node, HIP sampler / field / compositor), the depth is colour-mapped like torch_cmap and stacked under the colour image,
and the frames are played forth and back.  The reference writes an mp4 through imageio/ffmpeg (absent here); this module
writes an animated PNG with zlib alone and, when asked, the individual frames.
"""
import math
import os
import struct
import zlib

import numpy as np
import torch

from src.util.cam_geometry import get_ray_intersections


def sweep_extrinsics(left_extr, center_extr, right_extr, nframes):
    """World->camera matrices (nframes,4,4) of a sweep left -> centre -> right (dtu.py:256-318).

    Rotation origin = mean of the pairwise closest points of the three optical axes; radius = mean distance of the three
    camera centres from it; centres follow the spherical interpolation sin((1-t) th)/sin(th) x1 + sin(t th)/sin(th) x2 of
    the unit offsets on each half of the path; orientations come from scipy's Slerp over times [0, 0.5, 1]."""
    from scipy.spatial.transform import Rotation, Slerp
    poses = [torch.linalg.inv(e.float()) for e in (left_extr, center_extr, right_extr)]
    rays = [torch.cat((p[:3, -1], p[:3, -2])) for p in poses]          # camera centre + optical axis (third column)
    pts = get_ray_intersections(rays[0], rays[1]) + get_ray_intersections(rays[1], rays[2]) + \
        get_ray_intersections(rays[0], rays[2])
    origin = torch.mean(torch.stack(pts, dim=0), dim=0)
    radius = sum(torch.norm(origin - p[:3, -1], p=2) for p in poses) / 3
    t = torch.linspace(0, 1, nframes)
    x = [p[:3, -1] - origin for p in poses]
    x = [v / torch.norm(v, p=2) for v in x]
    th1 = torch.acos(torch.matmul(x[0], x[1]).clip(min=-1, max=1.0))
    th2 = torch.acos(torch.matmul(x[1], x[2]).clip(min=-1, max=1.0))
    centers = torch.zeros(nframes, 3)
    first = t < 0.5
    t1, t2 = t[first] * 2, t[~first] * 2 - 1
    centers[first] = (torch.sin((1 - t1[:, None]) * th1) / torch.sin(th1) * x[0][None]
                      + torch.sin(t1[:, None] * th1) / torch.sin(th1) * x[1][None])
    centers[~first] = (torch.sin((1 - t2[:, None]) * th2) / torch.sin(th2) * x[1][None]
                       + torch.sin(t2[:, None] * th2) / torch.sin(th2) * x[2][None])
    centers = centers * radius + origin[None]
    rots = Rotation.concatenate([Rotation.from_matrix(p[:3, :3].numpy()) for p in poses])
    target_rots = torch.tensor(Slerp([0.0, 0.5, 1.0], rots)(t.numpy()).as_matrix())
    target_poses = torch.eye(4)[None].repeat(nframes, 1, 1)
    target_poses[:, :3, :3] = target_rots
    target_poses[:, :3, -1] = centers
    return torch.linalg.inv(target_poses)


def write_apng(path, frames, fps=5, level=1):
    """frames (N,H,W,3) uint8 -> animated PNG (PNG 1.2 + the APNG extension: acTL, one fcTL per frame, IDAT for the first
    frame and fdAT for the others), written with zlib alone.  Any PNG viewer shows frame 0; browsers play the loop."""
    a = frames.detach().cpu().numpy() if torch.is_tensor(frames) else np.asarray(frames)
    assert a.dtype == np.uint8 and a.ndim == 4 and a.shape[-1] == 3
    N, H, W, _ = a.shape

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    out = [b"\x89PNG\r\n\x1a\n", chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)),
           chunk(b"acTL", struct.pack(">II", N, 0))]
    seq = 0
    for i in range(N):
        out.append(chunk(b"fcTL", struct.pack(">IIIIIHHBB", seq, W, H, 0, 0, 1, int(fps), 0, 0)))
        seq += 1
        raw = np.concatenate([np.zeros((H, 1), np.uint8), a[i].reshape(H, W * 3)], axis=1).tobytes()
        z = zlib.compress(raw, level)
        if i == 0:
            out.append(chunk(b"IDAT", z))
        else:
            out.append(chunk(b"fdAT", struct.pack(">I", seq) + z))
            seq += 1
    out.append(chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(b"".join(out))


def read_apng_frames(path):
    """-> (N,H,W,3) uint8: the frames write_apng wrote (tests)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, frames, W, H = 8, [], 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) & 0xffffffff == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR":
            W, H = struct.unpack(">II", body[:8])
        elif tag in (b"IDAT", b"fdAT"):
            rows = np.frombuffer(zlib.decompress(body if tag == b"IDAT" else body[4:]), np.uint8).reshape(H, 1 + 3 * W)
            frames.append(rows[:, 1:].reshape(H, W, 3))
        pos += 12 + n
    return np.stack(frames)


@torch.no_grad()
def create_cam_sweep(nerf, renderer, target_extrinsics, target_intrinsics, W, H, znear, zfar, outpath=None, fps=5,
                     ray_batch_size=8192, frames_dir=None, rank=0, world=1, group=None):
    """Renders the sweep of the scene last passed to nerf.encode(): target_extrinsics (N,4,4), one intrinsics matrix (3,3).

    Returns the frame stack (2N-1, 3, 2H, W) float on rank 0 (colour image on top, viridis depth below, played forth and
    back: frames[cat(arange(N), arange(N-1, 0, -1))], diner.py:204-208) and writes it as an animated PNG to `outpath`."""
    from .render import predict_image
    from . import imageio
    N = target_extrinsics.shape[0]
    dev = target_extrinsics.device
    K = target_intrinsics.view(1, 3, 3).to(dev)
    rgbs, depth_u8 = [], []
    for i in range(N):
        rgb, depth = predict_image(nerf, renderer, target_extrinsics[i:i + 1], K, W, H, znear, zfar,
                                   ray_batch_size=ray_batch_size, rank=rank, world=world, group=group)
        if rank != 0:
            continue
        rgbs.append(rgb[0])                                                  # (3,H,W)
        depth_u8.append(imageio.depth_to_uint8(depth[0]))                     # (H,W,3) uint8, per-frame min / max like torch_cmap
    if rank != 0:
        return None
    top = torch.stack([imageio.to_uint8(r) for r in rgbs])                     # (N,H,W,3) uint8
    frames_u8 = torch.cat((top, torch.stack(depth_u8)), dim=1)                  # (N,2H,W,3)
    order = torch.cat((torch.arange(N), torch.arange(N - 1, 0, -1)))
    frames_u8 = frames_u8[order.to(frames_u8.device)]
    if frames_dir is not None:
        os.makedirs(frames_dir, exist_ok=True)
        for j in range(N):
            imageio.write_png(os.path.join(frames_dir, f"frame_{j:03d}.png"), frames_u8[j])
    if outpath is not None:
        os.makedirs(os.path.dirname(os.path.abspath(outpath)), exist_ok=True)
        write_apng(outpath, frames_u8, fps=fps)
    return frames_u8.permute(0, 3, 1, 2).float() / 255.0
