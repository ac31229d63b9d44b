#!/usr/bin/env python
"""bench.py -- rendered rays/s of the MI355X-native DINER renderer (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 20 --warmup 2          # launches its own 8 ranks (one per GPU, RCCL over xGMI)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W           # the same ranks under an external launcher

Workload (BASELINE.json north_star / configs[2]-shaped, synthetic): 4-source-view scene, 800x600 source and target
images, 128 samples/ray (48 gaussian, 1000 depth candidates), fp32 results, random-init MLP of the trained DINER
configuration, seeded N(0,1) latent.  A "step" = ONE 800x600 target frame = 480,000 rays through the whole hot path:
per-scene projection of the latent (lin_z hoist) -> ray generation -> depth-guided sampling -> projection / gather /
encoding / MLP -> compositing -> assembly of the (rgb, depth) image on rank 0.  Inputs (feature maps, depth / std /
normal maps, cameras, weights) are resident in HBM before the timed region; nothing is cached across steps.

With N GPUs the SAME frame is sharded (BASELINE configs[3]; the build's replacement of the serial ray-batch loop of the
reference's src/models/diner.py:85-92): rank r generates and renders the contiguous ray range
diner_amd.render.shard_range(H*W, r, N) and one RCCL gather per frame brings the 16 B/ray tiles to rank 0 -- "strong"
scaling, value = 480,000 rays x steps / max-over-ranks time.  Scene state is replicated, so every rank repeats the
per-scene hoist (that, the ragged last shard and the gather are inside the timed region).

`--gpus N` without a launcher's WORLD_SIZE in the environment starts the N ranks itself (one child process per rank,
rendezvous on 127.0.0.1 and a free port); the parent's stdout carries exactly rank 0's JSON line and its exit status is
non-zero when any rank failed.  After the timed region an N-rank run renders one more frame with a fixed seed and rank 0
compares the gathered frame BIT FOR BIT with the same frame rendered by itself alone (`frame_check`; a ray's noise is keyed by
(frame seed, index of the ray in the frame), so the frame cannot depend on the number of ranks): the first run on a
multi-GPU node is a parity test of the sharded path as well.  `dist` in the line records what the process group was:
world size, backend, RCCL version, the device of every rank.

The JSON line also carries
  roofline      the dominant kernel (per-view MLP part, ~86 % of the time): executed MFMA FLOP / HIP-event duration
                measured in this run, against the dense MFMA peak of the dtype the products are issued in
  modes         the same frame in each other arithmetic mode (exact fp32; plain fp16 operands): median of 3 timed frames,
                each entry with its own roofline figures
  configs       the other single-GPU BASELINE configurations (400x300; 1024x1024 K=192 in f16 and f16x3) and the 800x600
                frame rendered THROUGH THE DROP-IN MODULES at the reference's call granularity (118 renderer.forward
                calls of 4096 rays, diner.py:85): median of 3 timed frames each, with roofline figures
  cpu_baseline  the CPU oracle (torch restatement of the reference renderer, pinned bit-exact against it) timed on the
                host cores of the same box: 4096 rays of the same frame, one warm-up at size + 3 timed repeats (rank 0,
                N = 1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0      # same guide, dense fp16/bf16 matrix peak (AMD's 5 PF figure is 2:1 sparse)
CHECK_SEED = 0x5EED0F0F            # frame seed of the sharded-vs-single frame check
CHILD_ENV = "DINER_BENCH_CHILD"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--candidates", type=int, default=1000)
    ap.add_argument("--white-bkgd", action="store_true")
    ap.add_argument("--facescape", action="store_true",
                    help="Facescape depth range / sigma law (BASELINE configs[4]): znear/zfar 1.0/2.5, white background")
    ap.add_argument("--weak", action="store_true",
                    help="every rank renders its own full frame (weak scaling) instead of sharding one frame")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays of the CPU baseline sample (0 disables)")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--no-modes", action="store_true", help="skip the extra passes in the other arithmetic modes")
    ap.add_argument("--modes-multi", action="store_true",
                    help="with N > 1 the extra passes in the other arithmetic modes are OFF by default (a scaling run needs the headline only); "
                         "this switches them on")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the extra passes of the other single-GPU BASELINE configs (400x300; 1024x1024 K=192 f16 / f16x3; "
                         "800x600 through the drop-in modules)")
    ap.add_argument("--extra-steps", type=int, default=3, help="timed frames (median) of every modes / configs entry")
    ap.add_argument("--ray-batch", type=int, default=8192, help="rays per launch group (bounds the workspace)")
    ap.add_argument("--via-modules", action="store_true",
                    help="headline measured through the drop-in modules: src.models.* built by import_obj, "
                         "diner_amd.render.predict_image with --module-ray-batch rays per renderer.forward call (diner.py:85)")
    ap.add_argument("--module-ray-batch", type=int, default=4096, help="ray_batch_size of the module path (diner.py:57)")
    ap.add_argument("--emulate-shard", default=None, metavar="R/N",
                    help="single-GPU measurement aid: render only the ray range rank R of an N-way sharded frame would render "
                         "(no process group, no gather); the line then reports that shard's time and the frame rate N such "
                         "GPUs would reach if the slowest shard took this long")
    ap.add_argument("--backend", choices=["auto", "nccl", "gloo"], default="auto",
                    help="process-group backend for --gpus > 1: nccl = RCCL over xGMI (one rank per GPU); gloo = the tile gather "
                         "and the seed broadcast staged through pinned host memory (ranks SHARING a GPU: RCCL refuses two ranks "
                         "on one device); auto = nccl when every local rank has its own device, else gloo")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives of the N-rank path even with ONE rank (a 1-GPU box can "
                         "then exercise RCCL itself: communicator creation, barrier, gather, all-reduce on device tensors)")
    ap.add_argument("--check-frame", dest="check_frame", action="store_true", default=None,
                    help="after the timed region render one frame with a fixed seed sharded and once more on rank 0 alone; compare bit "
                         "for bit (default: on for N > 1; with one rank the shards are emulated as 8 ray ranges rendered in turn)")
    ap.add_argument("--no-check-frame", dest="check_frame", action="store_false")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step entry (row f1: SB 4 x 4096 rays x 40 samples, forward + backward)")
    ap.add_argument("--train-steps", type=int, default=7, help="timed training steps (median; >= 5)")
    ap.add_argument("--no-encode", action="store_true", help="skip the encode_ms entry (PixelNeRF.encode on 4 x 800x600 images, reported beside the metric)")
    ap.add_argument("--no-power", action="store_true", help="do not sample rocm-smi (power / sclk) during the timed region")
    ap.add_argument("--precision", choices=["f16x3", "fp32", "f16"], default=None,
                    help="MLP GEMM arithmetic of the headline number (default: the library default, f16x3)")
    return ap.parse_args(argv)


def kernel_source_digest():
    """sha256 of the field-kernel sources: profiles/pmc_latest.json records the digest its PMC run was taken with."""
    h = hashlib.sha256()
    for f in ("mlp_h3n.hip", "mlp.hip", "field_common.hpp", "common.hpp"):
        with open(os.path.join(ROOT, "diner_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def train_source_digest():
    """sha256 of the training-kernel sources (profiles/pmc_latest.json["train"] records the digest its PMC run was taken with)."""
    h = hashlib.sha256()
    for f in ("train.hip", "train_512.hip", "train_lin512.hip", "train_wgrad512.hip", "mlp_h3n.hip", "field_common.hpp", "common.hpp"):
        with open(os.path.join(ROOT, "diner_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def host_physical_cores():
    """Physical cores of this host (distinct thread-sibling sets), hardware threads -- north_star: 'core count stated'."""
    sibs = set()
    try:
        base = "/sys/devices/system/cpu"
        for d in os.listdir(base):
            if d.startswith("cpu") and d[3:].isdigit():
                fn = os.path.join(base, d, "topology", "thread_siblings_list")
                if os.path.exists(fn):
                    with open(fn) as f:
                        sibs.add(f.read().strip())
    except OSError:
        pass
    return (len(sibs) or None), os.cpu_count()


class PowerSampler:
    """rocm-smi (package power, sclk) sampled in a side process while a timed region runs: the headline runs AT the 1400 W package cap
    (DESIGN.md section 4), so a kernel change shows up as energy per ray before it shows up as time -- the line carries the quantity.
    A sample is one `rocm-smi --showpower --showclocks` call of device `index` (~0.3 s each); nothing is parsed -> the fields stay null."""

    def __init__(self, index=0, enabled=True):
        import shutil
        self.index, self.samples, self.clocks, self._stop, self._thr = index, [], [], None, None
        self.exe = (shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)) if enabled else None

    def _loop(self):
        import re
        while not self._stop.is_set():
            try:
                out = subprocess.run([self.exe, "-d", str(self.index), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                break
            m = re.search(r"Power \(W\): ([\d.]+)", out)
            if m:
                self.samples.append(float(m.group(1)))
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            if m:
                self.clocks.append(float(m.group(1)))
            self._stop.wait(0.1)

    def __enter__(self):
        if self.exe:
            import threading
            self._stop = threading.Event()
            self._thr = threading.Thread(target=self._loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=15)

    def summary(self, elapsed_s, rays):
        med = lambda v: sorted(v)[len(v) // 2] if v else None
        pw, ck = med(self.samples), med(self.clocks)
        return {"power_w": pw, "sclk_mhz": ck, "power_samples": len(self.samples),
                "joule_per_mray": round(pw * elapsed_s / (rays / 1e6), 1) if pw and rays else None,
                "power_source": "median of rocm-smi --showpower --showclocks samples taken by a side process during the timed region" if pw
                                else "no rocm-smi sample parsed"}


# ---- self-launch: `python bench.py --gpus N` starts its own N ranks -------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """One child process per rank (env RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_*: the variables
    torch.distributed.run would set), rank 0's stdout piped back and re-printed, every other rank's stdout on stderr.
    -> exit status: 0 when every rank returned 0, else the first non-zero status (remaining ranks are terminated)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), CHILD_ENV: "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: the only mode the host driver supports
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), stderr=None))
    rc = 0
    alive = set(range(n))
    out0 = b""
    try:
        while alive:
            for r in sorted(alive):
                if r == 0:
                    try:                                   # drain rank 0's pipe while waiting (it carries one line)
                        o, _ = procs[0].communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for q in alive:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    lines = [l for l in out0.decode("utf-8", "replace").splitlines() if l.strip()]
    for l in lines:
        print(l, flush=True)
    if rc == 0 and not any(l.lstrip().startswith("{") for l in lines):
        print("bench.py: rank 0 produced no JSON line", file=sys.stderr)
        rc = 1
    return rc


def encode_entry(torch, ops, dev, msd, W, H, make_scene, build_modules, repeats=3):
    """PixelNeRF.encode (pixelnerf.py:35-53, image_encoder.py:225-291) on 4 source images of the frame size through this repo's own modules:
    normalisation + depth2normal (HIP) + the ResNet34 trunk (torch / MIOpen, random init) + feature pyramid, then what the renderer adds per
    scene: channels-last relayout (HipScene) and the lin_z hoist (diner_scene_prepare_f32).  Not part of `value`: reported beside it."""
    sc = make_scene(W, H, seed=0, latent=False)
    nerf, _ = build_modules(dict(sc, latent=torch.zeros(4, 512, 2, 2)), msd, dev)
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand(1, 4, 3, H, W, generator=g).to(dev)
    depths, stds = sc["depths"][None].to(dev), sc["depths_std"][None].to(dev)
    E, Km = sc["src_extrinsics"][None].to(dev), sc["src_intrinsics"][None].to(dev)
    mlp = nerf.hip_mlp()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    rows = []
    with torch.no_grad():
        for i in range(repeats + 1):
            e = [ev() for _ in range(4)]
            e[0].record()
            nerf.encode(imgs, depths, stds, E, Km)
            e[1].record()
            scene = nerf.hip_scene(0)
            e[2].record()
            scene.prepare(mlp, force=True)
            e[3].record()
            torch.cuda.synchronize()
            if i:
                rows.append([e[k].elapsed_time(e[k + 1]) for k in range(3)])
    med = lambda k: round(sorted(r[k] for r in rows)[len(rows) // 2], 3)
    lat = nerf.encoder.latent
    return {"encode_ms": round(med(0) + med(1) + med(2), 3), "trunk_and_prep_ms": med(0), "relayout_ms": med(1), "hoist_ms": med(2),
            "images": f"4 x {W}x{H} RGB + depth / std maps", "latent": list(lat.shape[1:]),
            "note": "PixelNeRF.encode of src.models (normalise, depth2normal on the device, ResNet34 trunk through torch / MIOpen with random-init "
                    "weights, pyramid upsampling) + channels-last relayout + lin_z hoist; median of %d after one warm-up; excluded from `value`, "
                    "which re-runs only the hoist per frame" % repeats}


def train_entry(torch, ops, dev, msd, make_scene, build_modules, n_steps, SB=4, NR=4096, K=40, size=(400, 300)):
    """One optimiser step's rendering work as the shipped configs run it: SB 4 objects x a 64 x 64 ray patch x 40 samples (15 gaussian, 1000
    candidates) through NeRFRendererDGS.forward in grad mode + MSE on fine.rgb + backward into the MLP parameters and encoder.latent
    (diner.py:217-290, configs/train_dtu.yaml:16,52-63).  The parameters are written in place before every step (as an optimiser does), so
    nothing cached per parameter version is left out.  Steps are enqueued back to back (no synchronisation inside the timed region); HIP events
    on the stream give the per-step period and the forward / backward split."""
    import time as _t
    from diner_amd import train as T
    Wt, Ht = size
    G = int(15 * K / 40)
    scs = [make_scene(Wt, Ht, seed=s) for s in range(SB)]
    nerf, R = build_modules(scs, msd, dev)
    nerf.train()
    from diner_amd.synthetic import as_encoded
    nerf.encoder.latent = as_encoded(nerf.encoder.latent.detach()).requires_grad_(True)      # channels-last strides, as PixelNeRF.encode emits it
    E = torch.stack([s["target_extrinsics"] for s in scs])
    Km = torch.stack([s["target_intrinsics"] for s in scs])
    rays_all = ops.gen_rays(E, Km, Wt, Ht, scs[0]["znear"], scs[0]["zfar"], dev)
    side = int(round(NR ** 0.5))
    ys, xs = torch.meshgrid(torch.arange(side) + (Ht - side) // 2, torch.arange(side) + (Wt - side) // 2, indexing="ij")
    r = rays_all[:, (ys * Wt + xs).reshape(-1).to(dev)].contiguous()
    del rays_all
    gt = torch.rand(SB, NR, 3, device=dev)
    ren = R(n_samples=K, n_depth_candidates=1000, n_gaussian=G, white_bkgd=True)
    params = [p for p in nerf.parameters()]
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def step(e=None):
        for p in params:
            p.grad = None
        nerf.encoder.latent.grad = None
        with torch.no_grad():
            torch._foreach_add_(params, 0.0)
        if e:
            e[0].record()
        out = ren.forward(nerf, r)
        loss = torch.nn.functional.mse_loss(out.fine.rgb, gt)
        if e:
            e[1].record()
        loss.backward()
        if e:
            e[2].record()

    torch.cuda.reset_peak_memory_stats()
    step()
    step()
    torch.cuda.synchronize()
    evs = [[ev() for _ in range(3)] for _ in range(n_steps + 1)]
    host = []
    t0 = _t.perf_counter()
    for i in range(n_steps + 1):
        h0 = _t.perf_counter()
        step(evs[i])
        host.append(_t.perf_counter() - h0)
    enq = _t.perf_counter() - t0
    torch.cuda.synchronize()
    wall = _t.perf_counter() - t0
    med = lambda v: sorted(v)[len(v) // 2]
    period = [evs[i][0].elapsed_time(evs[i + 1][0]) for i in range(n_steps)]
    fwd = [evs[i][0].elapsed_time(evs[i][1]) for i in range(n_steps)]
    bwd = [evs[i][1].elapsed_time(evs[i][2]) for i in range(n_steps)]
    ms = med(period)
    P = NR * K
    f_ref = 2 * P * (4 * (55 * 512 + 9 * 512 * 512) + 4 * 512 * 512 + 4 * 512)          # the reference's forward FLOPs per object
    f_exec_fwd = 2 * P * (4 * (55 * 512 + 6 * 512 * 512) + 4 * 512 * 512 + 4 * 512) + 2 * 3 * 512 * 512 * 4 * scs[0]["latent"].shape[-1] * scs[0]["latent"].shape[-2]
    ref_tf = 3 * SB * f_ref / (ms * 1e-3) / 1e12
    issued = 3 * SB * (f_exec_fwd + 2 * f_ref) / (ms * 1e-3) / 1e12
    hbm = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            ent = json.load(f).get("train")
        if ent and ent.get("source_digest") == train_source_digest():
            hbm = ent.get("hbm_gb_per_object_step")
    except Exception:
        pass
    saved, scratch = T.workspace_split(P, 4)
    return {"ms_per_step": round(ms, 2), "rays_per_s": round(SB * NR / (ms * 1e-3), 1), "steps": n_steps,
            "ms_all": [round(t, 2) for t in period], "forward_ms": round(med(fwd), 2), "backward_ms": round(med(bwd), 2),
            "ms_per_step_wall": round(wall / (n_steps + 1) * 1e3, 2), "host_enqueue_ms": round(med(host) * 1e3, 2),
            "host_enqueue_total_ms": round(enq * 1e3, 1),
            "tflops_fp32_equivalent": round(ref_tf, 1), "mfma_issued_tflops": round(issued, 1), "frac": round(issued / PEAK_F16_MFMA_TFLOPS, 4),
            "peak_memory_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            "saved_gib_per_object": round(saved / 2 ** 30, 2), "hbm_gb_per_object_step": hbm,
            "batched": T.batch_enabled(P, [nerf.hip_scene(sb) for sb in range(SB)]),
            "config": {"workload": f"row f1: SB {SB} objects x {NR} rays (a {side} x {side} patch) x {K} samples ({G} gaussian, 1000 candidates), 4 source views of "
                                   f"{Wt}x{Ht}, NeRFRendererDGS.forward in grad mode + MSE + backward into the MLP parameters and encoder.latent; "
                                   f"parameters written in place before every step; the latent leaf in channels-last strides, the format PixelNeRF.encode of this repo emits on the device", "objects": SB, "rays_per_object": NR, "samples_per_ray": K},
            "note": "ms_per_step = median period between the steps' first HIP events (steps enqueued back to back, nothing synchronised in "
                    "between); tflops_fp32_equivalent = 3 x the reference's forward FLOPs (forward + data gradient + weight gradient) / time; "
                    "mfma_issued = 3 fp16 MFMA products per fp32 product of the EXECUTED work (forward with lin_z hoisted to the maps, backward as the "
                    "reference's), frac against the dense fp16 peak"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not os.environ.get(CHILD_ENV):
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    import torch
    launched_by = "bench.py self-launch" if os.environ.get(CHILD_ENV) else ("external launcher (torch.distributed.run)" if "WORLD_SIZE" in os.environ
                                                                              else "none (one process, --force-dist)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU path to measure")
    if args.emulate_shard and (args.weak or world > 1):
        raise SystemExit("--emulate-shard is a single-process measurement of one shard of the sharded frame: not with --weak / N > 1")
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    shared = local_world > n_dev                    # ranks time-slice a GPU (single-GPU box): not a scaling measurement
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    dist = None
    backend = None
    json_fd = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29577")):
            os.environ.setdefault(k, v)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.backend if args.backend != "auto" else ("gloo" if shared else "nccl")
        # The collective libraries write to the process's stdout themselves (gloo: "[Gloo] Rank ..." at connect; RCCL: a version
        # banner through C stdio, flushed at exit): file descriptor 1 is pointed at stderr for the whole run and the JSON line goes
        # to the saved descriptor, so that rank 0's stdout carries exactly one line.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(backend="gloo")
        dist.barrier()

    from diner_amd import ops
    from diner_amd.render import shard_range, gather_tiles, predict_image
    from diner_amd.synthetic import make_scene, make_mlp_state_dict, look_at_extrinsics, build_modules

    if args.precision:
        ops.set_precision(args.precision)
    head = ops.get_precision()
    names = {ops.PRECISION_FP32: "fp32", ops.PRECISION_F16X3: "f16x3", ops.PRECISION_F16: "f16"}
    msd = make_mlp_state_dict()
    mlp = ops.HipMlp({k: v.to(dev) for k, v in msd.items()})
    n_cand = args.candidates

    def workload(W, H, K, facescape, white, lo_hi=None, via_modules=False):
        """Scene resident in HBM + the per-frame step of one configuration -> dict(step, out, frame, ...)."""
        G = int(15 * K / 40)                           # create_prediction_folder.py:44-47
        scene_kw = dict(scale=1.75, znear=1.0, zfar=2.5, std_law="facescape") if facescape else {}
        sc = make_scene(W, H, seed=0, **scene_kw)
        Kin = sc["src_intrinsics"]
        depths = sc["depths"].to(dev)
        normals = ops.depth2normal(depths, Kin.to(dev))                     # encode-side prep (row f2), not timed
        NRF = W * H
        s_ = scene_kw.get("scale", 1.0)
        if args.weak:       # every rank renders its own target view of the same scene
            tgt = look_at_extrinsics((s_ * (0.03 + 0.04 * rank), -0.02 * s_, -1.0 * s_))
            lo, hi = 0, NRF
        else:               # one frame, contiguous ray range per rank
            tgt = sc["target_extrinsics"]
            lo, hi = lo_hi(NRF) if lo_hi else shard_range(NRF, rank, world)
        tgt_E, tgt_K = tgt[None].contiguous(), sc["target_intrinsics"][None].contiguous()
        frame = [None]
        res = dict(frame=frame, sc=sc, normals=normals, Kin=Kin, NRF=NRF, lo=lo, hi=hi, G=G, scene_kw=scene_kw)

        if via_modules:
            # The product at the API it claims: PixelNeRF / NeRFRendererDGS built through import_obj (diner.py:47-48), the scene
            # injected into the encoder, predict_image = predict_imgs_from_batch's loop (diner.py:79-92): torch.split of the ray list
            # into ray_batch_size = 4096 (diner.py:57) and one renderer.forward per batch -- 118 calls per 800x600 frame.
            nerf, R = build_modules(sc, msd, dev, normals=normals)
            del sc["latent"]
            ren = R(n_samples=40, n_depth_candidates=n_cand, n_gaussian=15, white_bkgd=white)
            ren.n_samples, ren.n_gaussian = K, G                            # create_prediction_folder.py:44-47
            tE, tK = tgt_E.to(dev), tgt_K.to(dev)
            grp = dist.group.WORLD if (multi and not args.weak) else None

            def step(seed, precision, max_rays=None):
                ops.set_precision(precision)
                try:
                    nerf.hip_scene(0).prepare(nerf.hip_mlp(), force=True, f16=precision == ops.PRECISION_F16)   # the hoist belongs to the frame (see below)
                    if max_rays is not None:                                # warm-up on the first rays only
                        with torch.no_grad():
                            rr = ops.gen_rays(tE, tK, W, H, sc["znear"], sc["zfar"], dev, ray0=0, n_rays=min(NRF, max_rays))
                            ren.forward(nerf, rr)
                        return
                    rgb, depth = predict_image(nerf, ren, tE, tK, W, H, sc["znear"], sc["zfar"], ray_batch_size=args.module_ray_batch,
                                               rank=rank if grp is not None else 0, world=world if grp is not None else 1, group=grp, seed=seed)
                    if rgb is not None:
                        frame[0] = torch.cat((rgb[0].permute(1, 2, 0).reshape(NRF, 3), depth[0].reshape(NRF, 1)), dim=1)
                finally:
                    ops.set_precision(head)
            res.update(step=step, out=None, scene=nerf.hip_scene(0), nerf=nerf, mlp=nerf.hip_mlp())
            return res

        scene = ops.HipScene(sc["latent"].to(dev), depths, sc["depths_std"].to(dev), normals,
                             sc["src_extrinsics"], Kin[:, [0, 1], [0, 1]], Kin[:, :2, -1], sc["image_shape"],
                             sc["feature_padding"])
        del sc["latent"]
        out = torch.empty(hi - lo, 4, device=dev)      # packed (rgb, depth) tile of this rank

        def render_range(a, b, seed, precision, dst):
            """rays [a, b) of the frame -> dst (b - a, 4); ray generation, sampling, field, compositing."""
            rays = ops.gen_rays(tgt_E, tgt_K, W, H, sc["znear"], sc["zfar"], dev, ray0=a, n_rays=b - a)[0]
            for r0 in range(0, b - a, args.ray_batch):
                r = rays[r0:r0 + args.ray_batch]
                z = ops.sample_depthguided(scene, r, K, n_cand, G, 0.05, noise=None, seed=seed, ray_index0=a + r0)   # one key per frame
                _, rgb, depth = ops.render(scene, mlp, r, z, white_bkgd=white, want_weights=False, precision=precision)
                dst[r0:r0 + args.ray_batch, :3] = rgb
                dst[r0:r0 + args.ray_batch, 3] = depth

        phases = []             # per timed frame of this rank: (event start, after the hoist, after the shard, after the gather, host s of a staged gather)
        res["phases"] = phases

        def step(seed, precision, max_rays=None):
            # per-scene preparation (projection of the latent through lin_z[0..2]) is redone every
            # frame inside the timed region, so that no cached per-scene output is excluded from the measurement
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if res.get("record_phases") else None
            if ev:
                ev[0].record()
            scene.prepare(mlp, force=True, f16=precision == ops.PRECISION_F16)      # (the fp16 copy of the maps belongs to the f16 mode's frame)
            if ev:
                ev[1].record()
            n = hi - lo if max_rays is None else min(hi - lo, max_rays)
            render_range(lo, lo + n, seed, precision, out)
            if ev:
                ev[2].record()
            if multi and not args.weak:
                host_s = None
                if ev and backend != "nccl":       # the staged gather begins with a stream synchronisation anyway: take it first, then the
                    torch.cuda.current_stream().synchronize()      # host clock brackets the gather alone (pinned copy + gloo + upload on rank 0)
                    t_g = time.perf_counter()
                frame[0] = gather_tiles(out, NRF, rank, world, force=args.force_dist)   # one RCCL gather of the rendered tiles per frame
                if ev:
                    if backend != "nccl":
                        host_s = time.perf_counter() - t_g
                    ev[3].record()                 # (RCCL: the current stream waits for the collective, the event closes behind it)
                    phases.append((ev, host_s))
            elif multi:
                src = out if backend == "nccl" else out.cpu()
                gat = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
                dist.gather(src, gat, dst=0)
            else:
                frame[0] = out
        res.update(step=step, out=out, scene=scene, render_range=render_range, mlp=mlp)
        return res

    W, H, K = args.width, args.height, args.samples
    white = bool(args.white_bkgd or args.facescape)
    lo_hi = None
    if args.emulate_shard:
        er, en = (int(x) for x in args.emulate_shard.split("/"))
        lo_hi = lambda n: shard_range(n, er, en)
    wl_head = workload(W, H, K, args.facescape, white, lo_hi, via_modules=args.via_modules)
    step, out, frame, sc, scene, normals, Kin = (wl_head[k] for k in ("step", "out", "frame", "sc", "scene", "normals", "Kin"))
    NRF, lo, hi, G = (wl_head[k] for k in ("NRF", "lo", "hi", "G"))

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, precision, seed0, profile=False, step=step):
        sync()
        if profile:
            ops.profile_enable(True)
            ops.profile_collect()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(seed0 + i, precision)
        sync()
        el = time.perf_counter() - t0
        prof = None
        if profile:
            prof = ops.profile_collect()
            ops.profile_enable(False)
        if multi:
            t = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, prof

    def mode_facts(m):
        """(kernel name, MFMA products issued per fp32 product, peak TFLOP/s, dtype label) of arithmetic mode m."""
        if m == ops.PRECISION_F16X3:
            return ("k_field_pre_h8x" if os.environ.get("DINER_F16X3_W8") == "1" else "k_field_pre_h3n<true>"), 3, PEAK_F16_MFMA_TFLOPS, "f16 (3 MFMA products per fp32 product, fp32 accumulate)"
        if m == ops.PRECISION_F16:
            # round 5: the plain-fp16 mode runs on the eight-wave kernel (two waves per SIMD) unless DINER_F16_W8=0 selects the four-wave one
            kn = "k_field_pre_h3n<false>" if os.environ.get("DINER_F16_W8") == "0" else "k_field_pre_h8"
            return kn, 1, PEAK_F16_MFMA_TFLOPS, "f16 operands, fp32 accumulate (reduced precision: ~1e-3, outside the parity bar)"
        return "k_field_pre", 1, PEAK_FP32_MFMA_TFLOPS, "f32"

    def traffic_for(kname, dims, prof):
        """HBM bytes per launch of kernel `kname` on workload dims = (W, H, K): PMC bytes/point of tools/profile_round.sh (FETCH_SIZE x2 +
        WRITE_SIZE) x points per launch.  Counters cannot be read inside this process; the committed figure is only quoted when it was
        taken with the kernel sources of this build (digest match) on this workload, otherwise null."""
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
                pmc = json.load(f)
            wl_key = "%dx%dx%d" % dims
            ent = pmc.get(f"{kname} @ {wl_key}") or pmc.get(kname)
            if ent and ent.get("source_digest") == kernel_source_digest() and ent.get("workload") == wl_key:
                return (round(ent["hbm_bytes_per_point"] * prof["points"] / max(prof["launches"], 1)),
                        f"HBM bytes per launch: PMC bytes/point of {ent['source']} x points per launch")
        except Exception:
            pass
        return None, "no PMC figure for this build and workload (profiles/pmc_latest.json missing or stale)"

    def roof(m, prof, wall_s, dims=None):
        """Roofline figures of one measured run: the per-view kernel from its HIP-event time, the whole path from the wall time."""
        kname, mult, peak, label = mode_facts(m)
        tr = traffic_for(kname, dims, prof)[0] if dims else None
        pre_s = prof["pre_ms"] * 1e-3
        ach = prof["points"] * ops.FLOP_PRE_PER_POINT * mult / pre_s / 1e12 if pre_s > 0 else 0.0
        path = prof["points"] * (ops.FLOP_PRE_PER_POINT + ops.FLOP_POST_PER_POINT) * mult / wall_s / 1e12
        return {"bound": "mfma", "kernel": kname, "mfma_dtype": label, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "launches": prof["launches"],
                "avg_launch_ms": round(prof["pre_ms"] / max(prof["launches"], 1), 3),
                "points_per_launch": round(prof["points"] / max(prof["launches"], 1)), "traffic": tr,
                "whole_path_achieved": round(path, 2), "whole_path_frac": round(path / peak, 4)}

    def measure(stepf, m, rays, n, dims=None, handle=None):
        """median-of-n timed frames of `stepf` in mode m (each frame bracketed by barrier + synchronize) + roofline figures;
        fallback_launches: field launches of the timed frames that the gated exact-fp32 pass recomputed (must be 0: else the number is that pass's)."""
        sync()
        handle = handle if handle is not None else mlp
        handle.fallback_launches(reset=True)
        ops.profile_enable(True)
        ops.profile_collect()
        ts = []
        for i in range(n):
            el, _ = timed(1, m, 7 + i, step=stepf)
            ts.append(el)
        prof = ops.profile_collect()
        ops.profile_enable(False)
        med = sorted(ts)[len(ts) // 2]
        return {"rays_per_s": round(rays / med, 1), "ms_per_step": round(med * 1e3, 2), "steps": n,
                "ms_all": [round(t * 1e3, 2) for t in ts], "mode": names[m], "fallback_launches": handle.fallback_launches(reset=True),
                "roofline": roof(m, prof, sum(ts), dims)}

    for i in range(args.warmup):
        step(i, head)
    wl_head["record_phases"] = bool(multi and not args.weak and not args.via_modules)
    wl_head["mlp"].fallback_launches(reset=True)
    with PowerSampler(dev.index or 0, enabled=not args.no_power and rank == 0) as psamp:
        elapsed, prof = timed(args.steps, head, args.warmup, profile=True)
    head_fallbacks = wl_head["mlp"].fallback_launches(reset=True)
    wl_head["record_phases"] = False
    if not os.environ.get("DINER_AMD_LIB"):        # (timing experiments with ablated libraries produce garbage)
        if out is not None:
            assert torch.isfinite(out).all(), "non-finite render output"
        if rank == 0 and not args.weak and not args.emulate_shard:
            assert frame[0].shape == (NRF, 4) and torch.isfinite(frame[0]).all()

    rays_per_step = NRF * (world if args.weak else 1)
    if args.emulate_shard:
        rays_per_step = hi - lo
    rays_per_s = rays_per_step * args.steps / elapsed

    # ---- sharded frame == single-rank frame (bit for bit) -----------------------------------------------------------------------
    frame_check = None
    do_check = args.check_frame if args.check_frame is not None else (world > 1)
    if do_check and not args.weak and not args.emulate_shard and not args.via_modules:
        rr = wl_head["render_range"]
        if world > 1 or args.force_dist:
            step(CHECK_SEED, head)                      # every rank its range + the gather
            sync()
            how = f"{world} ranks ({backend}) + gather"
            sharded = frame[0].clone() if rank == 0 else None
        else:
            n_em = 8                                    # one rank: the 8 ray ranges of an 8-way shard rendered in turn
            sharded = torch.empty(NRF, 4, device=dev)
            scene.prepare(mlp, force=True, f16=head == ops.PRECISION_F16)
            for r in range(n_em):
                a, b = shard_range(NRF, r, n_em)
                rr(a, b, CHECK_SEED, head, sharded[a:b])
            how = f"{n_em} ray ranges rendered in turn by one rank (emulation)"
        if rank == 0:
            single = torch.empty(NRF, 4, device=dev)
            scene.prepare(mlp, force=True, f16=head == ops.PRECISION_F16)
            rr(0, NRF, CHECK_SEED, head, single)
            torch.cuda.synchronize()
            sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
            frame_check = {"bit_equal": bool(torch.equal(sharded, single)), "sharded": how, "seed": CHECK_SEED,
                           "sha256_sharded": sha(sharded), "sha256_single_rank": sha(single),
                           "max_abs_diff": float((sharded - single).abs().max().item())}
            del single, sharded
        sync()

    # ---- who took part (answerable from the line alone: did RCCL see N ranks, which devices) ----------------------------------
    dist_info = None
    if multi:
        pr = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "local_rank": local_rank, "device": dev.index, "name": pr.name,
              "pci_bus_id": getattr(pr, "pci_bus_id", None), "uuid": str(getattr(pr, "uuid", "")) or None, "pid": os.getpid()}
        ph = wl_head.get("phases") or []
        if ph:                                  # the timed frames of THIS rank: hoist / shard / gather, HIP events on the launch stream
            torch.cuda.synchronize()
            stat = lambda v: {"min": round(min(v), 3), "median": round(sorted(v)[len(v) // 2], 3), "max": round(max(v), 3)}
            hoist = [e[0].elapsed_time(e[1]) for e, _ in ph]
            shard = [e[1].elapsed_time(e[2]) for e, _ in ph]
            gath = [(e[2].elapsed_time(e[3]) if hs is None else hs * 1e3) for e, hs in ph]
            me.update({"rays": hi - lo, "frames": len(ph), "hoist_ms": stat(hoist), "shard_ms": stat(shard), "gather_ms": stat(gath)})
        infos = [None] * world
        dist.all_gather_object(infos, me)
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        dist_info = {"world_size": dist.get_world_size(), "backend": str(dist.get_backend()), "rccl_version": rccl,
                     "devices_visible_per_rank": n_dev, "distinct_devices": len({(i["device"], i["pci_bus_id"]) for i in infos}),
                     "ranks": infos, "launcher": launched_by}
        if all("shard_ms" in i for i in infos):
            # what a sub-linear SCALE line is made of: the slowest rank's replicated hoist + its shard against the measured frame time
            # (the rest = the gather, the barrier and launch gaps); shard raggedness = slowest / fastest shard
            busy = [i["hoist_ms"]["median"] + i["shard_ms"]["median"] for i in infos]
            frame_ms = elapsed / args.steps * 1e3
            dist_info["breakdown"] = {
                "frame_ms": round(frame_ms, 3), "slowest_rank_hoist_plus_shard_ms": round(max(busy), 3),
                "fastest_rank_hoist_plus_shard_ms": round(min(busy), 3),
                "hoist_ms_max": max(i["hoist_ms"]["median"] for i in infos), "shard_ms_max": max(i["shard_ms"]["median"] for i in infos),
                "shard_ms_min": min(i["shard_ms"]["median"] for i in infos), "gather_ms_rank0": infos[0]["gather_ms"]["median"],
                "gather_ms_max": max(i["gather_ms"]["median"] for i in infos),
                "scaling_efficiency_vs_emulated": round(max(busy) / frame_ms, 4),
                "note": "per rank and timed frame, HIP events on the launch stream: hoist = per-frame scene preparation (replicated), shard = "
                        "ray generation + sampler + field + compositor of the rank's ray range, gather = the tile gather (RCCL: event "
                        "behind the collective; gloo: host clock around the staged gather, behind a stream synchronisation); "
                        "scaling_efficiency_vs_emulated = slowest rank's (hoist + shard) / measured frame time: 1.0 = the frame costs what "
                        "its slowest shard costs, the remainder is gather + barrier + launch gaps"}

    # ---- the other arithmetic modes ------------------------------------------------------------------------------------------
    modes = {}
    if not args.no_modes and (world == 1 or args.modes_multi):
        for m in (ops.PRECISION_FP32, ops.PRECISION_F16X3, ops.PRECISION_F16):
            if m == head:
                continue
            step(0, m)                              # warm-up (first launch of that kernel family)
            e = measure(step, m, rays_per_step, max(1, args.extra_steps), (W, H, K))
            e["parity"] = ("outside the 1e-4 bar (~1e-3), BASELINE configs[4] only" if m == ops.PRECISION_F16
                           else "1e-4 bar (same tests as the headline mode)")
            modes[names[m]] = e

    # ---- roofline of the dominant kernel (this rank's launches, HIP events on the launch stream) ----------
    h3 = head == ops.PRECISION_F16X3
    f16 = head == ops.PRECISION_F16
    pre_kernel, mfma_per_product, peak, _ = mode_facts(head)
    rf = roof(head, prof, elapsed)
    pre_s = prof["pre_ms"] * 1e-3
    fp32_equiv = prof["points"] * ops.FLOP_PRE_PER_POINT / pre_s / 1e12 if pre_s > 0 else 0.0
    ref_equiv = prof["points"] * ops.FLOP_PRE_PER_POINT_REFERENCE / pre_s / 1e12 if pre_s > 0 else 0.0
    traffic, traffic_note = traffic_for(pre_kernel, (W, H, K), prof)
    roofline = {"bound": "mfma", "kernel": pre_kernel, "mfma_dtype": rf["mfma_dtype"],
                "achieved": rf["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": rf["frac"],
                "traffic": traffic, "traffic_unit": traffic_note,
                "launches": prof["launches"],
                "flop_per_point_executed_fp32_products": ops.FLOP_PRE_PER_POINT,
                "flop_per_point_reference": ops.FLOP_PRE_PER_POINT_REFERENCE,
                "achieved_fp32_equivalent": round(fp32_equiv, 2),
                "achieved_reference_flops": round(ref_equiv, 2),
                "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS,
                "avg_launch_ms": rf["avg_launch_ms"], "points_per_launch": rf["points_per_launch"],
                "post_kernel_ms_total": round(prof["post_ms"], 2), "pre_kernel_ms_total": round(prof["pre_ms"], 2)}
    # whole path: every MFMA FLOP of the two field kernels over the WALL time of the timed frames (sampler, hoist, compositor, ray
    # generation, launch gaps and -- for N > 1 -- the gather included), against the same peak
    roofline["whole_path"] = {"achieved": rf["whole_path_achieved"], "frac": rf["whole_path_frac"],
                              "unit": "TFLOP/s", "note": "MFMA FLOP of the per-view and post kernels of this rank / wall time of the "
                              "timed steps (sampler, per-frame hoist, compositor, ray generation and launch gaps included)",
                              "field_kernels_share_of_wall": round((prof["pre_ms"] + prof["post_ms"]) * 1e-3 / elapsed, 4)}

    # ---- CPU baseline: the oracle on the host cores of this box (rank 0, N = 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_rays != 0:
        from oracle import diner_oracle as O
        lat = scene.latent_cl.permute(0, 3, 1, 2).contiguous().cpu()         # the same feature maps, reference layout
        oscene = O.Scene(latent=lat, depths=sc["depths"], depths_std=sc["depths_std"], normals=normals.cpu(),
                         poses=sc["src_extrinsics"], focal=Kin[:, [0, 1], [0, 1]], c=Kin[:, :2, -1],
                         image_shape=sc["image_shape"], feature_padding=sc["feature_padding"])
        ow = O.MLPWeights.from_state_dict(msd)
        g = torch.Generator().manual_seed(0)
        rays_all = O.gen_rays(sc["target_extrinsics"], sc["target_intrinsics"], W, H, sc["znear"], sc["zfar"])

        def sample(n):
            idx = torch.linspace(0, NRF - 1, n).long()            # spread over the frame
            return (rays_all[idx].contiguous(), torch.rand(n, n_cand, generator=g), torch.randn(n, G, generator=g),
                    torch.rand(n, K, generator=g))

        def cpu_once(smp):
            t = time.perf_counter()
            with torch.no_grad():      # same chunking as the reference: one renderer.forward call of <= 4096 rays,
                O.render(oscene, ow, smp[0], K, n_cand, G, white, smp[1], smp[2], smp[3])   # 100,000-point MLP chunks
            return time.perf_counter() - t

        # thread sweep on a small sample (torch / MKL does not scale to every SMT thread), warm-up at each setting
        hw = os.cpu_count() or 1
        probe = sample(256)
        sweep = {}
        for nt in sorted({min(hw, c) for c in (16, 32, 64, 128, hw)}):
            torch.set_num_threads(nt)
            cpu_once(probe)
            sweep[nt] = round(256 / cpu_once(probe), 1)
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        n_cpu = min(args.cpu_rays, NRF)
        smp = sample(n_cpu)
        cpu_once(sample(min(512, n_cpu)))                         # warm-up with the chosen thread count (the sweep above warmed every code path)
        times = sorted(cpu_once(smp) for _ in range(max(1, args.cpu_repeats)))
        med = times[len(times) // 2]
        cpu = {"value": round(n_cpu / med, 2), "unit": "rays/s", "cores": best, "kind": "port",
               "sample": f"{n_cpu} rays spread over the same {W}x{H} frame, {K} samples/ray, one call of the torch CPU oracle "
                         f"(restatement of the reference renderer, pinned bit-exact; 100,000-point MLP chunks) per repeat; "
                         f"1 warm-up (512 rays, after the thread sweep) + {len(times)} timed repeats, median {med:.1f} s (min {times[0]:.1f}, max "
                         f"{times[-1]:.1f}); {best} of {hw} hardware threads = fastest of the sweep {sweep} (rays/s on 256 rays)",
               "repeats_s": [round(t, 2) for t in times], "thread_sweep_rays_per_s": sweep, "host_threads": hw,
               "host_cores": host_physical_cores()[0], "torch_threads_used": best,
               "oracle_over_reference": 1.415,
               "oracle_over_reference_note": "throughput of this restatement / throughput of the IMPORTED reference on the same 1024 rays in the build "
                                             "container (8 threads, 5 alternating repeats, outputs bit-equal: medians 74.2 / 52.4 rays/s; oracle/"
                                             "time_vs_reference.py, profiles/r06_cpu_oracle_vs_reference_timing.txt): the reference's own CPU path is "
                                             "~1.4x SLOWER than `value`, i.e. the baseline flatters the CPU",
               "cores_note": "`cores` = torch intra-op threads of the fastest setting of the sweep (what was actually used); `host_cores` = physical "
                             "cores of the box (distinct thread-sibling sets), `host_threads` = hardware threads"}

    # ---- the other single-GPU configurations of BASELINE.json + the module-level API (N = 1 only) ---------------------
    configs = {}
    if world == 1 and not args.no_configs and not args.emulate_shard and not args.via_modules \
            and (W, H, K, bool(args.facescape)) == (800, 600, 128, False):
        nx = max(1, args.extra_steps)
        plan = {
            "800x600 K=128 through src.models (import_obj), 118 renderer.forward calls of 4096 rays per frame (diner.py:85)":
                (800, 600, 128, False, (head,), True),
            "configs[1] 400x300 K=128": (400, 300, 128, False, (ops.PRECISION_F16X3,), False),
            "configs[4] 1024x1024 K=192 Facescape range, white background":
                (1024, 1024, 192, True, (ops.PRECISION_F16, ops.PRECISION_F16X3), False)}
        for key, (cw, ch, ck, cfs, cmodes, via) in plan.items():
            wl = workload(cw, ch, ck, cfs, cfs, via_modules=via)
            for m in cmodes:
                wl["step"](0, m, max_rays=2 * args.ray_batch)          # warm-up on the first two ray batches
                e = measure(wl["step"], m, cw * ch, nx, (cw, ch, ck), handle=wl["mlp"])
                if wl["out"] is not None:
                    assert torch.isfinite(wl["out"]).all()
                else:
                    assert torch.isfinite(wl["frame"][0]).all()
                e.update({"ms_per_frame": e["ms_per_step"], "rays_per_frame": cw * ch, "samples_per_ray": ck,
                          "parity": "outside the 1e-4 bar (~1e-3; 85.8 dB against the reference image, tests/test_hip_parity.py::test_cfg5_fp16_mlp_psnr)"
                          if m == ops.PRECISION_F16 else "1e-4 bar (tests/test_hip_parity.py::test_render_at_metric_sample_counts)"})
                if via:
                    e["vs_ops_level_headline"] = round(e["rays_per_s"] / rays_per_s, 4)
                configs[f"{key} [{names[m]}]"] = e
            del wl
            torch.cuda.empty_cache()

    # ---- encode, reported beside the metric ("encode excluded, reported separately"; pixelnerf.py:35-53) ------------------------
    encode = None
    if world == 1 and not args.no_encode and not args.emulate_shard:
        try:
            encode = encode_entry(torch, ops, dev, msd, W, H, make_scene, build_modules)
        except Exception as e:      # (a trunk that does not run on this box must not cost the headline)
            encode = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    # ---- the training step, row f1 (N = 1 only; DINER.calc_losses, diner.py:217-290 at configs/train_dtu.yaml:16,52-63) -----------------
    train_line = None
    if world == 1 and not args.no_train and not args.emulate_shard:
        del wl_head, step, out, frame, scene
        torch.cuda.empty_cache()
        train_line = train_entry(torch, ops, dev, msd, make_scene, build_modules, max(5, args.train_steps))
        torch.cuda.empty_cache()

    if rank == 0:
        frame_name = f"{W}x{H}"
        if (W, H, K) == (800, 600, 128) and not args.facescape:
            wl = "BASELINE north_star / configs[2]-shaped"
        elif (W, H, K) == (400, 300, 128) and not args.facescape:
            wl = "BASELINE configs[1]"
        elif args.facescape:
            wl = "BASELINE configs[4]-shaped (Facescape range, white background)"
        else:
            wl = "variant of BASELINE configs[2]"
        if args.weak:
            par = f"{world} independent frames (weak), tiles gathered to rank 0 ({backend})"
        elif world == 1:
            par = "one frame on one GPU (no collective)" + (
                f"; --force-dist: the gather / barrier / all-reduce of the N-rank path run through {backend} with one rank" if multi else "")
        else:
            par = f"one frame ray-sharded x{world} (BASELINE configs[3]), " + (
                "one RCCL gather of (rgb,depth) tiles to rank 0 per frame" if backend == "nccl" else
                "one gloo gather of (rgb,depth) tiles to rank 0 per frame staged through pinned host memory")
            if shared:
                par += f"; OVERSUBSCRIBED: {local_world} ranks time-slice {n_dev} GPU(s) -- exercises the N-rank code path, NOT a scaling number"
        if args.via_modules:
            par += f"; through src.models.* + diner_amd.render.predict_image, {args.module_ray_batch} rays per renderer.forward call"
        if args.emulate_shard:
            par = (f"EMULATION on one GPU: the ray range of rank {er} of {en} only (rays {lo}..{hi}), scene preparation included, no "
                   f"process group and no gather; {en} GPUs whose slowest shard takes this long render {NRF * args.steps / elapsed:.0f} rays/s")
        line = {
            "metric": f"rendered rays/sec ({K} samples/ray, 4 src views)", "value": round(rays_per_s, 1),
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None,
            "dtype": ("f32 via f16x3 split MFMA products, fp32 accumulate" if h3 else
                      "f16 operands / f32 accumulate (REDUCED PRECISION, not the headline configuration)" if f16 else "f32"),
            "data": "synthetic",
            "config": {"workload": f"{wl}: synthetic 4-view scene, {frame_name} target = {NRF} rays per frame, "
                                   f"{rays_per_step} rays per step, {K} samples/ray ({G} gaussian, {n_cand} candidates), MLP "
                                   f"d_hidden 512 / 5 blocks ({'f16x3 split-product' if h3 else 'plain fp16-operand' if f16 else 'exact fp32'} "
                                   f"MFMA GEMMs), random-init weights, in-kernel Philox noise, "
                                   f"{'white' if white else 'black'} background",
                       "rays_per_step": rays_per_step, "rays_per_gpu_per_step": hi - lo, "samples_per_ray": K, "src_views": 4,
                       "frame": frame_name, "parallelism": par},
            "backend": backend, "ranks_share_gpu": bool(shared and world > 1),
            "fallback_launches": head_fallbacks,
            # (N > 1: rank 0 samples ITS device only -- power_w / sclk_mhz are that device's, the energy per ray of the job is not computed)
            "energy": psamp.summary(elapsed, rays_per_step * args.steps if world == 1 else 0),
            "encode": encode,
            "train": train_line,
            "dist": dist_info, "frame_check": frame_check,
            "roofline": roofline,
            "modes": modes,
            "configs": configs,
            "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu"] = round(rays_per_s / cpu["value"], 1)
        if json_fd is not None:
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()
    if frame_check is not None and not frame_check["bit_equal"]:
        raise SystemExit("bench.py: the sharded frame differs from the single-rank frame (frame_check.bit_equal = false)")


if __name__ == "__main__":
    main()
