"""CPU: camera-sweep geometry (row f3) against outputs of the reference's own functions (tests/golden/g12_sweep.npz, generated in the
build container by calling the reference's DTUDataSet.get_cam_sweep_extrinsics / TransSlerp / pose_spherical) and the animated
PNG writer."""
import numpy as np
import torch

from tests.helpers import load


def test_sweep_extrinsics_match_reference():
    from diner_amd.sweep import sweep_extrinsics
    g = load("g12_sweep.npz")
    got = sweep_extrinsics(torch.from_numpy(g["left"]), torch.from_numpy(g["center"]), torch.from_numpy(g["right"]), int(g["nframes"]))
    assert got.shape == (9, 4, 4)
    assert np.abs(got.numpy() - g["extrinsics"]).max() < 2e-6
    # the path starts at the left camera's orientation, passes the centre camera's in the middle and ends at the right one's;
    # the camera centres are put on the MEAN radius about the rotation origin (dtu.py:275-277, :303), so they only come close
    for i, k in ((0, "left"), (4, "center"), (8, "right")):
        assert np.abs(got[i, :3, :3].numpy() - g[k][:3, :3]).max() < 2e-5
        assert np.abs(got[i, :3, 3].numpy() - g[k][:3, 3]).max() < 5e-2


def test_sweep_helpers_match_reference():
    import src.util.cam_geometry as cg
    g = load("g12_sweep.npz")
    assert np.abs(cg.TransSlerp(g["ts_times"], g["ts_loc"])(g["ts_q"]) - g["ts_out"]).max() < 1e-12
    assert torch.equal(cg.pose_spherical(30.0, -20.0, 1.3), torch.from_numpy(g["pose_sph"]))
    a, b = cg.get_ray_intersections(torch.tensor([1.0, 0, 0, -1, 0, 0]), torch.tensor([0.0, -1, 0, 0, 1, 0]))
    assert torch.allclose(a, torch.zeros(3), atol=1e-6) and torch.allclose(b, torch.zeros(3), atol=1e-6)
    assert cg.to_homogeneous_trafo(torch.zeros(2, 3, 4)).shape == (2, 4, 4)
    rot, loc = cg.Slerp([0.0, 1.0], __import__("scipy.spatial.transform", fromlist=["Rotation"]).Rotation.from_euler(
        "z", [0, 90], degrees=True), np.array([[0.0, 0, 0], [2.0, 0, 0]]))(np.array([0.5]))
    assert abs(rot.as_euler("zyx", degrees=True)[0][0] - 45) < 1e-9 and np.allclose(loc, [[1.0, 0, 0]])


def test_apng_roundtrip(tmp_path):
    from diner_amd.sweep import write_apng, read_apng_frames
    f = np.random.default_rng(0).integers(0, 256, size=(5, 12, 9, 3), dtype=np.uint8)
    p = str(tmp_path / "s.png")
    write_apng(p, f, fps=5)
    assert np.array_equal(read_apng_frames(p), f)
    from diner_amd import imageio                       # a plain PNG reader sees frame 0 (default image = first frame)
    data = open(p, "rb").read()
    assert data.count(b"fcTL") == 5 and data.count(b"fdAT") == 4 and data.count(b"acTL") == 1
