#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the 2D-Gaussian hot path.

One STEP = one pass of the hot path over one image's gaussians, everything resident in HBM:
    project_gaussians_2d (fwd) -> tile binning -> rasterize_sum forward
    -> L2-loss gradient of the rendered image -> rasterize_sum backward -> project_gaussians_2d (bwd)
i.e. the work one training iteration of models/gaussianimage_cholesky.py:302-317 hands to the `gsplat`
operator surface, with the gradient image derived from the step's own render exactly as loss.backward() does
(clamp + MSE against a fixed synthetic target).  tanh / +bound / the optimizer are not part of the metric
(BASELINE.json: "training iters/sec (fwd+bwd rasterize)"); they are in the separate `train_step` figure.
In the timed loop a step is two launches, as in the training loop: the tile pass, then one kernel that finishes the
step (gradient reduce + project backward) and projects + bins the gaussians for the next one (HotPath.step); every
timed step contains exactly one of each operation.

N GPUs (`--gpus N`): one process per GPU, one independent image per rank (images shard embarrassingly,
no data-path collective) -> weak scaling; value = ranks * K / max-over-ranks time.  Launched by the driver under
torch.distributed.run the ranks come from RANK/LOCAL_RANK/WORLD_SIZE; launched bare (`python bench.py --gpus N`,
WORLD_SIZE unset) this file starts the N ranks itself as child processes -- the parent never touches the GPU -- and
relays rank 0's JSON line.

The second BASELINE metric, "Kodak images/sec at 1/2/4/8 GPU", is the `images_per_s` block of the same line: the
per-image loop of train.py:294-340 (covariance model, prune / grow schedule, best model on the device) over 24
Kodak-shaped synthetic images sharded image i -> rank i mod N, one all-reduce for the "Average:" figures.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
EVENT_STRIDE = 8


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--num-points", type=int, default=50000)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=768)
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--images", type=int, default=24,
                   help="images of the images/sec leg (Kodak-24 shaped: 768x512 synthetic pictures); 0 skips the leg")
    p.add_argument("--image-iterations", type=int, default=10000,
                   help="training iterations per image in the images/sec leg (the reference's default is 50000)")
    p.add_argument("--images-per-gpu", type=int, default=4,
                   help="images fitted concurrently on each GPU in the images/sec leg (one HIP stream + host thread each)")
    p.add_argument("--images-per-gpu-probe", action="store_true",
                   help="also report the aggregate step rate of 2, 3 and 4 independent images stepped concurrently on "
                        "separate HIP streams of this GPU (extra information, not `value`)")
    p.add_argument("--train-step", action="store_true",
                   help="also time the whole training iteration (gi2d_train_step) after the timed region")
    return p.parse_args(argv)


# ------------------------------------------------------------------------------------------ N ranks from one command
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N child ranks (fresh interpreters, so nothing that has
    initialised the GPU is ever re-exec'ed; this parent imports neither torch nor the HIP library), relay rank 0's
    stdout -- the one JSON line -- and return the worst exit code."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
    return worst


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a mislabelled run")
    run_rank(args)


def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py rank {rank}: no GPU, no number (this file measures the HIP path only)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI on a real node; GI2D_BENCH_BACKEND=gloo only to rehearse N ranks on a box with fewer GPUs
        dist.init_process_group(os.environ.get("GI2D_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    dev_index = local_rank % torch.cuda.device_count()  # one GPU per rank; ranks share only in a 1-GPU rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from helpers import synth_cholesky, synth_gt
    from gaussianimage_plus_amd.hotpath import HotPath

    n, h, w = args.num_points, args.height, args.width
    xyz, L, col, op = synth_cholesky(n, h, w, 3047 + rank)  # reference default seed (train.py:225) + rank
    hp = HotPath(n, h, w, device=dev)
    hp.set_inputs(xyz, L, col, op)
    # every step renders, forms the L2 gradient against a seeded smooth target and back-propagates it
    gt_np = synth_gt(h, w, 1 + rank)
    gt = torch.from_numpy(gt_np).to(dev)
    hp.set_target(gt)
    hp.forward()
    m = hp.num_intersects()

    def barrier():
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[dev_index])  # this rank's own GPU, stated rather than guessed
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        hp.step()
    barrier()
    # HIP start/stop events attached to the rasterizer tile-pass dispatch (gi2d_timer_*: the kernel's own begin/end
    # timestamps on the launch stream), on every EVENT_STRIDE-th step of the timed region
    n_timed = max(1, args.steps // EVENT_STRIDE)
    ev = hp.kernel_timers(n_timed)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % EVENT_STRIDE == 0 and i // EVENT_STRIDE < n_timed:
            hp.step(timer=ev, index=i // EVENT_STRIDE)
        else:
            hp.step()
    barrier()
    elapsed = time.perf_counter() - t0
    hp.check_status()

    red_dev = dev if (world == 1 or dist.get_backend() == "nccl") else "cpu"
    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    ms = torch.tensor([float(m)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ms, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    value = world * args.steps / elapsed

    images = images_per_s(args, rank, world, dev, red_dev, barrier) if args.images > 0 else None

    if rank == 0:
        dom = hp.dominant_kernel_stats(ev)  # name, avg_us, algorithmic bytes per launch
        achieved = dom["bytes"] / (dom["avg_us"] * 1e-6) / 1e9
        pair_bytes = 80 * m + 36 * h * w + 36 * n  # SURVEY 8d north-star figure (fwd + bwd rasterize)
        traffic, traffic_src = pmc_traffic(dom["name"], n, h, w)
        line = {
            "metric": f"training iters/sec (fwd+bwd rasterize) at N Gaussians, {w}x{h}",
            "value": value,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"Cholesky model, N={n} Gaussians, {w}x{h}, one image per GPU: project fwd + tile binning + "
                            f"rasterize_sum fwd + L2 gradient of the render + rasterize_sum bwd + project bwd per step",
                "num_points": n, "height": h, "width": w, "num_intersects_rank0": m,
                "num_intersects_mean": float(ms.item()) / world, "seed": 3047,
                "host_path": hp.describe(),
            },
            "roofline": {
                "bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": dom["bytes"], "avg_kernel_us": dom["avg_us"],
                "note": "VALU-bound by construction (each staged gaussian is reused by up to 256 pixels); see DESIGN.md",
            },
            "rasterize_pair": hp.pair_stats(ev, pair_bytes),
        }
        if images is not None:
            line["images_per_s"] = images
        if args.train_step:
            line["train_step"] = train_step_rate(gt, n, dev)
            line["quantized_train_step"] = quantized_train_step_rate(gt, dev)
        if args.images_per_gpu_probe:
            line["concurrent_images"] = concurrent_images_rate(n, h, w, dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(xyz, L, col, op, gt_np, h, w, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


def images_per_s(args, rank, world, dev, red_dev, barrier):
    """BASELINE.json's second metric: the per-image fitting loop of train.py:294-340 over a Kodak-24-shaped batch,
    image i on rank i mod N (launch.run_sharded), whole-job images / wall second (max over ranks by the closing
    barrier).  Model and schedule are train.py's defaults scaled to a stated iteration count: covariance model, Adam
    lr 0.018, 5000 -> 50000 gaussians (BASELINE config 3: densification on), prune every 100 iterations, growth every
    iterations/10 with the whole remaining budget released at the last growth step (train.py:91-99)."""
    import torch
    import torch.distributed as dist
    from gaussianimage_plus_amd import launch

    iters = int(args.image_iterations)
    h, w = 512, 768  # Kodak: 18 landscape + 6 portrait pictures of 768x512 pixels; the synthetic batch is all landscape
    num_points, max_points = 5000, 50000
    grow_iter, prune_iter = max(iters // 10, 1), 100
    pics = [launch.synthetic_image(h, w, 100 + i) for i in range(args.images)]
    kw = dict(lr=0.018, seed=3047, kind="covariance", max_points=max_points, prune_iter=prune_iter, grow_iter=grow_iter,
              eps=1e-15, optimizer="adam", eval_renders=1)

    def fit_one(i, img):
        return launch.fit_image_native(img.to(dev), num_points, iters, **kw)

    def fit_group(idx, imgs):
        return launch.fit_images_native([im.to(dev) for im in imgs], num_points, iters, threaded=True, **kw)

    # untimed warm-up, the counterpart of --warmup for the step metric: one small image through the same schedule, so that
    # every kernel of the loop (tile pass, update, prune / growth, render) has its code object loaded -- tens of ms each on
    # first use, a fixed cost per process that would otherwise be charged to the 3 images a rank fits at N = 8
    launch.fit_image_native(launch.synthetic_image(96, 144, 99).to(dev), 500, 300, lr=0.018, seed=3047,
                            kind="covariance", max_points=1500, prune_iter=100, grow_iter=100, eps=1e-15,
                            optimizer="adam", eval_renders=1)
    torch.cuda.synchronize(dev)
    barrier()
    t0 = time.perf_counter()
    out = launch.run_sharded(pics, fit_one, rank, world, device=red_dev, group=max(1, args.images_per_gpu),
                             fit_group=fit_group)
    barrier()
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    wall = float(wall.item())
    return {"value": out["images"] / wall, "unit": "images/s", "images": out["images"], "wall_s": wall,
            "iterations_per_image": iters, "avg_psnr": out["avg_psnr"], "avg_num_gaussians": out["avg_num_gaussians"],
            "images_concurrent_per_gpu": max(1, args.images_per_gpu),
            "warmup": "one 144x96 image, 300 iterations of the same schedule, untimed (code objects loaded)",
            "workload": f"{args.images} synthetic 768x512 images (Kodak-24 shape), covariance model {num_points}->"
                        f"{max_points} gaussians, {iters} iterations/image (reference default 50000), prune every "
                        f"{prune_iter}, grow every {grow_iter}; image i -> rank i mod {world}"}


def pmc_traffic(kernel, n, h, w):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/traffic.json, built by
    tools/make_profiles.py: 2*FETCH_SIZE + WRITE_SIZE per the gfx950 correction of MI355X_MICROARCH.md) and where the
    figure comes from.  Hardware counters cannot be read from inside the timed process, so this is a STORED value of
    the same command under rocprofv3, labelled as such; null when no counters were collected for this workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        c = t["config"]
        if (c["num_points"], c["height"], c["width"]) != (n, h, w):
            return None, f"none: profiles/traffic.json holds counters for {c['num_points']} gaussians at " \
                         f"{c['width']}x{c['height']}, not for this workload"
        for k, v in t["kernels"].items():
            if k.replace(" ", "").endswith(kernel.replace(" ", "")):
                return v["hbm_bytes_per_launch"], \
                    f"stored: profiles/traffic.json ({t.get('source', 'rocprofv3 --pmc passes of bench.py')}), " \
                    f"not measured in this run"
        return None, f"none: profiles/traffic.json has no entry for {kernel}"
    except (OSError, KeyError, ValueError) as e:
        return None, f"none: profiles/traffic.json unreadable ({type(e).__name__})"


def concurrent_images_rate(n, h, w, dev, rounds=300):
    """Extra information, not `value`: K independent images (own buffers, own HIP stream) stepped round-robin from
    this process.  One image leaves most CUs idle between its dependent phases, so the aggregate rate rises with K
    until the host's launch rate (3 C-ABI calls per step) becomes the limit."""
    import torch
    from helpers import synth_cholesky, synth_gt
    from gaussianimage_plus_amd.hotpath import HotPath
    out = []
    for k in (2, 3, 4):
        hps, streams = [], []
        for i in range(k):
            hp = HotPath(n, h, w, device=dev)
            hp.set_inputs(*synth_cholesky(n, h, w, 4000 + i))
            hp.set_target(torch.from_numpy(synth_gt(h, w, 10 + i)).to(dev))
            hps.append(hp)
            streams.append(torch.cuda.Stream(device=dev))

        def run(r):
            for _ in range(r):
                for hp, st in zip(hps, streams):
                    with torch.cuda.stream(st):
                        hp.step()
        run(20)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(rounds)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        for hp in hps:
            hp.check_status()
        out.append({"images": k, "steps_per_s": k * rounds / dt, "us_per_round": dt / rounds * 1e6})
    return out


def train_step_rate(gt, n, dev, iters=400):
    """Extra information, not `value`: the whole training iteration (hot path + L2 loss gradient + Adam update,
    gi2d_train_step = 3 launches, no host sync) on the same image size / gaussian count, measured after the
    timed region."""
    import torch
    from gaussianimage_plus_amd.trainer import NativeFitter
    fit = NativeFitter(gt.contiguous(), n, kind="cholesky", lr=1e-3, seed=3047)
    fit.train(40)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    fit.train(iters)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    fit.check_status()
    return {"iters_per_s": iters / dt, "us_per_iter": dt / iters * 1e6, "num_intersects": int(fit.nth.sum().item()),
            "what": "full training iteration incl. activations, L2 loss gradient and Adam (gi2d_train_step)"}


def quantized_train_step_rate(gt, dev, n=30000, iters=400):
    """Extra information, not `value`: BASELINE config 5 -- N = 30 000, quantisation-aware iteration (train_quantize.py
    after its warm-up; 4 launches, no host sync) for the rotation-scale model the config names (LSQ 12-bit positions,
    6-bit scaling, signed 6-bit rotation, 6-bit colour: models/gaussianimage_rs.py:131-163) and for the covariance
    model train_quantize.py actually wires (12 / 10 / 6 bits), each with its plain iteration beside it."""
    import torch
    from gaussianimage_plus_amd.trainer import NativeFitter
    out = {}
    for kind, lr, bits in (("scale_rot", 1e-3, (12, 6, 6)), ("covariance", 0.018, (12, 10, 6))):
        fit = NativeFitter(gt.contiguous(), n, kind=kind, lr=lr, eps=1e-15, seed=3047, track_best=True)
        def stretch():
            # median of four quarter stretches: one host hiccup on a shared box (tens of ms, seen once in a dozen
            # runs) would otherwise multiply a 7 ms measurement
            times = []
            for _ in range(4):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                fit.train(iters // 4)
                torch.cuda.synchronize(dev)
                times.append(time.perf_counter() - t0)
            return sorted(times)[1:3][0] * 0.5 * 4 + sorted(times)[1:3][1] * 0.5 * 4

        fit.train(200)
        fit.prune_non_definite()
        plain = stretch()
        fit.load_best()
        fit.enable_quantize(*bits)
        fit.train(40)
        dt = stretch()
        fit.check_status()
        out[kind] = {"iters_per_s": iters / dt, "us_per_iter": dt / iters * 1e6,
                     "plain_us_per_iter": plain / iters * 1e6, "num_points": fit.n, "bits": list(bits)}
    out["what"] = "quantisation-aware iteration (gi2d_train_steps with gi2d_train_quant), N = 30 000, 768x512"
    return out


def cpu_baseline(xyz, L, col, op, gt, h, w, budget_s):
    """The CPU oracle (oracle/gi2d_oracle.c, OpenMP) on the host cores of this box, bounded to ~budget_s seconds in
    all.  kind "port": the reference has no CPU implementation of this path.  Three figures, as
    SURVEY 8d asks: `value` = full steps/s of the bench workload on all cores of this box's CPU share, the same on one
    thread, and BASELINE config 1 (768x512, N = 2500, Cholesky model: whole training iterations/s -- activations,
    hot path on the oracle, L2 loss, Adam in numpy)."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    # this box's CPU share for one GPU is 16 cores (more threads only add reduction overhead)
    cores = max(1, min(O.num_threads(), os.cpu_count() or 1, 16))
    n = xyz.shape[0]
    tb = O.tile_bounds(h, w)
    scale = 2.0 / (3 * h * w)

    def one(xyz_, L_, col_, op_):
        nn = xyz_.shape[0]
        xys, depths, radii, conics, nth = O.project_gaussians_2d_forward(nn, 3.0, xyz_, L_, h, w, tb, 0.01, 1.0)
        m, cum = O.compute_cumulative_intersects(nth)
        _, _, so, go, bins = O.bin_and_sort_gaussians(nn, m, xys, depths, radii, cum, tb, 1.0)
        out, fT, fidx = O.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, xys, conics, col_, op_)
        # the L2 gradient of the step's own render against the same target the GPU leg uses
        v_out = np.where((out >= 0) & (out <= 1), scale * (np.clip(out, 0, 1) - gt), 0).astype(np.float32)
        v_xy, v_conic, v_rgb, v_op = O.rasterize_sum_backward(h, w, 16, 16, go, bins, xys, conics, col_, op_, None, fT,
                                                               fidx, v_out)
        g = O.project_gaussians_2d_backward(nn, xyz_, L_, h, w, radii, conics, v_xy, None, v_conic)
        return g, v_rgb

    def rate(fn, budget, cap):
        fn()
        t0 = time.perf_counter()
        k = 0
        while k < 1 or (time.perf_counter() - t0 < budget and k < cap):
            fn()
            k += 1
        return k, time.perf_counter() - t0

    O.set_num_threads(cores)
    k, dt = rate(lambda: one(xyz, L, col, op), 0.5 * budget_s, 200)
    O.set_num_threads(1)
    k1, dt1 = rate(lambda: one(xyz, L, col, op), 0.2 * budget_s, 20)

    # config 1: the Cholesky model's train_iter (models/gaussianimage_cholesky.py:302-317) at N = 2500
    import math
    O.set_num_threads(cores)
    n1 = 2500
    rng = np.random.default_rng(3047)
    p_xyz = np.arctanh(np.clip(2 * (rng.random((n1, 2)) - 0.5), -0.999999, 0.999999)).astype(np.float32)
    p_chol = rng.random((n1, 3)).astype(np.float32)
    p_feat = np.zeros((n1, 3), np.float32)
    lp = min(h * w / (9 * math.pi * n1), 300)
    bound = np.array([lp, 0, lp], np.float32)
    op1 = np.ones((n1, 1), np.float32)
    state = {"t": 0, "m": [np.zeros_like(a) for a in (p_xyz, p_chol, p_feat)],
             "v": [np.zeros_like(a) for a in (p_xyz, p_chol, p_feat)]}

    def train_iter():
        mean = np.tanh(p_xyz)
        (v_cov2d, v_mean, v_L), v_rgb = one(mean, p_chol + bound, p_feat, op1)
        grads = [v_mean * (1 - mean * mean), v_L, v_rgb]
        state["t"] += 1
        t, lr, b1, b2, eps = state["t"], 1e-3, 0.9, 0.999, 1e-8
        for p, g_, m_, v_ in zip((p_xyz, p_chol, p_feat), grads, state["m"], state["v"]):
            m_ += (g_ - m_) * (1 - b1)
            v_ *= b2
            v_ += (1 - b2) * g_ * g_
            p -= (lr / (1 - b1 ** t)) * m_ / (np.sqrt(v_) / math.sqrt(1 - b2 ** t) + eps)

    kc, dtc = rate(train_iter, 0.3 * budget_s, 2000)
    return {"value": k / dt, "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"{k} full steps (project+bin+rasterize fwd+bwd, N={n}, {w}x{h}, same target image) of the "
                      f"OpenMP oracle in {dt:.1f} s",
            "single_thread": {"value": k1 / dt1, "unit": "iters/s", "cores": 1,
                              "sample": f"{k1} of the same steps on one thread in {dt1:.1f} s"},
            "config1_train_loop": {"value": kc / dtc, "unit": "iters/s", "cores": cores,
                                   "sample": f"{kc} training iterations (tanh/+bound, oracle hot path, L2, numpy Adam) "
                                             f"of the Cholesky model, N={n1}, {w}x{h}, in {dtc:.1f} s"}}


if __name__ == "__main__":
    main()
