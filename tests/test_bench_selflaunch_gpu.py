"""GPU: `python bench.py --gpus N` with no launcher in the environment starts its own N ranks (BASELINE configs[3]: the ray-sharded
frame that replaces the serial ray-batch loop of the reference's src/models/diner.py:85-92).  On the one MI355X `gpurun` offers the
ranks share the device (auto => gloo, host-staged collectives, labelled OVERSUBSCRIBED); on a multi-GPU node the same command runs
one rank per GPU over RCCL.  Statements: exit status 0, stdout is exactly ONE JSON line, the line says who took part (`dist`), and the
frame gathered from the N ranks is bit-equal to the frame rank 0 renders alone (`frame_check`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


@pytest.mark.parametrize("n", [2, 8])
def test_bench_launches_its_own_ranks(n):
    rc, out, err = _run(["--gpus", str(n), "--steps", "2", "--warmup", "1"])
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly the JSON line, got {len(lines)} lines:\n{out[:2000]}"
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["steps"] == 2 and line["scaling"] == "strong"
    assert line["config"]["rays_per_step"] == 800 * 600 and line["value"] > 0
    d = line["dist"]
    assert d["world_size"] == n and len(d["ranks"]) == n and sorted(r["rank"] for r in d["ranks"]) == list(range(n))
    assert d["launcher"] == "bench.py self-launch"
    assert len({r["pid"] for r in d["ranks"]}) == n                       # N processes
    # round 5: the line explains itself -- per rank the replicated hoist, the shard and the gather of the timed frames (min / median / max)
    for r in d["ranks"]:
        assert r["frames"] == 2 and r["rays"] == 800 * 600 // n
        for key in ("hoist_ms", "shard_ms", "gather_ms"):
            assert 0 < r[key]["min"] <= r[key]["median"] <= r[key]["max"]
    b = d["breakdown"]
    assert b["frame_ms"] == pytest.approx(line["ms_per_step"], rel=1e-3)
    assert 0 < b["scaling_efficiency_vs_emulated"] <= 1.25 and b["shard_ms_min"] <= b["shard_ms_max"]
    assert b["slowest_rank_hoist_plus_shard_ms"] <= b["frame_ms"] * 1.25       # (medians of 2 frames of ranks that time-slice one GPU)
    assert line["modes"] == {}                                            # off by default for N > 1 (a scaling run needs the headline only)
    fc = line["frame_check"]
    assert fc["bit_equal"] is True and fc["sha256_sharded"] == fc["sha256_single_rank"] and fc["max_abs_diff"] == 0.0
    import torch
    if torch.cuda.device_count() < n:                                     # the gpurun box: ranks share the GPU
        assert line["ranks_share_gpu"] is True and line["backend"] == "gloo" and "OVERSUBSCRIBED" in line["config"]["parallelism"]
    else:                                                                 # a multi-GPU node: RCCL, one device per rank
        assert line["backend"] == "nccl" and d["distinct_devices"] == n and d["rccl_version"]


def test_single_rank_line_names_no_collective():
    """N = 1: no process group, the parallelism string says so; --check-frame emulates the 8-way shard in one process."""
    rc, out, err = _run(["--gpus", "1", "--steps", "1", "--warmup", "1", "--width", "200", "--height", "152", "--no-modes", "--no-configs",
                         "--cpu-rays", "0", "--check-frame", "--no-train", "--no-encode", "--no-power"])
    assert rc == 0, err[-3000:]
    line = json.loads(out.strip().splitlines()[-1])
    assert line["backend"] is None and line["dist"] is None
    assert "no collective" in line["config"]["parallelism"] and "RCCL" not in line["config"]["parallelism"]
    assert line["frame_check"]["bit_equal"] is True


def test_failing_rank_gives_nonzero_status():
    """A rank that dies takes the run down with a non-zero status and no JSON line (an impossible frame: zero width)."""
    rc, out, err = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--width", "0", "--height", "8"], timeout=600)
    assert rc != 0 and not out.strip()
