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

N GPUs: one process per GPU, one independent image per rank (images shard embarrassingly,
no data-path collective) -> weak scaling; value = ranks * K / max-over-ranks time.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
EVENT_STRIDE = 8


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--num-points", type=int, default=50000)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=768)
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--images-per-gpu-probe", action="store_true",
                   help="also report the aggregate step rate of 2, 3 and 4 independent images stepped concurrently on "
                        "separate HIP streams of this GPU (extra information, not `value`)")
    p.add_argument("--train-step", action="store_true",
                   help="also time the whole training iteration (gi2d_train_step) after the timed region")
    return p.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI on a real node; GI2D_BENCH_BACKEND=gloo only to rehearse N ranks on a box with fewer GPUs
        dist.init_process_group(os.environ.get("GI2D_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; no GPU, no number"
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
    gt = torch.from_numpy(synth_gt(h, w, 1 + rank)).to(dev)
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

    if rank == 0:
        dom = hp.dominant_kernel_stats(ev)  # name, avg_us, algorithmic bytes per launch
        achieved = dom["bytes"] / (dom["avg_us"] * 1e-6) / 1e9
        pair_bytes = 80 * m + 36 * h * w + 36 * n  # SURVEY 8d north-star figure (fwd + bwd rasterize)
        line = {
            "metric": "training iters/sec (fwd+bwd rasterize) at N Gaussians, 768x512",
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
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom["name"], n, h, w),
                "algorithmic_bytes_per_launch": dom["bytes"], "avg_kernel_us": dom["avg_us"],
                "note": "VALU-bound by construction (each staged gaussian is reused by up to 256 pixels); see DESIGN.md",
            },
            "rasterize_pair": hp.pair_stats(ev, pair_bytes),
        }
        if args.train_step:
            line["train_step"] = train_step_rate(gt, n, dev)
            line["quantized_train_step"] = quantized_train_step_rate(gt, dev)
        if args.images_per_gpu_probe:
            line["concurrent_images"] = concurrent_images_rate(n, h, w, dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(xyz, L, col, op, h, w, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel, n, h, w):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/traffic.json, built by
    tools/make_profiles.py: 2*FETCH_SIZE + WRITE_SIZE per the gfx950 correction of MI355X_MICROARCH.md); null when no
    counters were collected for this workload -- counters cannot be read from inside the timed process."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        c = t["config"]
        if (c["num_points"], c["height"], c["width"]) != (n, h, w):
            return None
        for k, v in t["kernels"].items():
            if k.replace(" ", "").endswith(kernel.replace(" ", "")):
                return v["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def concurrent_images_rate(n, h, w, dev, rounds=300):
    """Extra information, not `value`: K independent images (own buffers, own HIP stream) stepped round-robin from
    this process.  One image leaves most CUs idle between its dependent phases, so the aggregate rate rises with K
    until the host's launch rate (3 C-ABI calls per step) becomes the limit."""
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
    """Extra information, not `value`: BASELINE config 5 -- covariance model, N = 30 000, quantisation-aware iteration
    (train_quantize.py after its warm-up: LSQ / log quantisers at 12 / 10 / 6 bits in front of the projection, their
    own Adam optimizers; 4 launches, no host sync), with the plain iteration of the same model beside it."""
    from gaussianimage_plus_amd.trainer import NativeFitter
    fit = NativeFitter(gt.contiguous(), n, kind="covariance", lr=0.018, eps=1e-15, seed=3047, track_best=True)
    fit.train(200)
    fit.prune_non_definite()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    fit.train(iters)
    torch.cuda.synchronize(dev)
    plain = time.perf_counter() - t0
    fit.load_best()
    fit.enable_quantize(12, 10, 6)
    fit.train(40)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    fit.train(iters)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    fit.check_status()
    return {"iters_per_s": iters / dt, "us_per_iter": dt / iters * 1e6, "plain_us_per_iter": plain / iters * 1e6,
            "num_points": fit.n, "bits": [12, 10, 6],
            "what": "covariance model, quantisation-aware iteration (gi2d_train_steps with gi2d_train_quant)"}


def cpu_baseline(xyz, L, col, op, h, w, budget_s):
    """The CPU oracle (oracle/gi2d_oracle.c, OpenMP) on the host cores of this box, same workload,
    bounded to ~budget_s seconds.  kind "port": the reference has no CPU implementation of this path."""
    from oracle import oracle as O
    O.build()
    # this box's CPU share for one GPU is 16 cores (more threads only add reduction overhead)
    cores = max(1, min(O.num_threads(), os.cpu_count() or 1, 16))
    O.set_num_threads(cores)
    n = xyz.shape[0]
    tb = O.tile_bounds(h, w)

    def one():
        xys, depths, radii, conics, nth = O.project_gaussians_2d_forward(n, 3.0, xyz, L, h, w, tb, 0.01, 1.0)
        m, cum = O.compute_cumulative_intersects(nth)
        _, _, so, go, bins = O.bin_and_sort_gaussians(n, m, xys, depths, radii, cum, tb, 1.0)
        out, fT, fidx = O.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, xys, conics, col, op)
        v_out = (2 * (np.clip(out, 0, 1) - 0.5) / (3 * h * w)).astype(np.float32)
        v_xy, v_conic, v_rgb, v_op = O.rasterize_sum_backward(h, w, 16, 16, go, bins, xys, conics, col, op, None, fT,
                                                               fidx, v_out)
        O.project_gaussians_2d_backward(n, xyz, L, h, w, radii, conics, v_xy, None, v_conic)

    one()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < budget_s and k < 200:
        one()
        k += 1
    dt = time.perf_counter() - t0
    return {"value": k / dt, "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"{k} full steps (project+bin+rasterize fwd+bwd, same N/size) of the OpenMP oracle in {dt:.1f} s"}


if __name__ == "__main__":
    main()
