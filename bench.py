#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the 2D-Gaussian hot path.

One STEP = one training iteration of models/gaussianimage_cholesky.py:302-317 for one image, everything resident in HBM
and every parameter MOVING (the optimizer is on):
    tanh / +bound -> project_gaussians_2d (fwd) -> tile binning -> rasterize_sum forward
    -> L2-loss gradient of the render -> rasterize_sum backward -> project_gaussians_2d (bwd) -> tanh' -> Adam
as gi2d_train_steps issues it: per step two launches -- the tile pass (rasterize forward + loss gradient + backward), then
one kernel that finishes the step (gradient reduce, projection backward, Adam) and projects + bins the updated gaussians
for the next one.  The timed region is ONE call of exactly K iterations.  (Rounds 1-2 timed a loop over FROZEN
parameters, where the incremental binning has nothing to append; that figure is still reported, as
`static_scene_step`.)

N GPUs (`--gpus N`): one process per GPU, one independent image per rank (images shard embarrassingly,
no data-path collective) -> weak scaling; value = ranks * K / max-over-ranks time.  Launched by the driver under
torch.distributed.run the ranks come from RANK/LOCAL_RANK/WORLD_SIZE; launched bare (`python bench.py --gpus N`,
WORLD_SIZE unset) this file starts the N ranks itself as child processes -- the parent never touches the GPU -- and
relays rank 0's JSON line.

`batched`: the same iteration for K = 4 / 8 / 24 images in lockstep, every kernel launched once for all of them
(gi2d_train_steps_batched) -- how BASELINE config 3 (a 24-image batch) is fitted; roofline on K x algorithmic bytes.

The second BASELINE metric, "Kodak images/sec at 1/2/4/8 GPU", is the `images_per_s` block of the same line: the
per-image loop of train.py:294-340 (covariance model, prune / grow schedule, best model on the device) over the 24
Kodak pictures (tests/golden/kodak24.npz: pixels of datasets/kodak/kodim01..24.png), 50 000 iterations each
(train.py:204), sharded image i -> rank i mod N, each rank's images fitted as one batch, one all-reduce for the
"Average:" figures.

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
EVENT_STRIDE = 2      # every 2nd tile pass of the event-carrying regions has a HIP event pair attached
REPEATS = 5           # timed regions of `steps` iterations each; the median is reported
SIMDS, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; shader clock under this load (tools/clock_probe.sh: 2.40 GHz)


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
                   help="images of the images/sec leg (the first so many of Kodak-24); 0 skips the leg")
    p.add_argument("--image-iterations", type=int, default=50000,
                   help="training iterations per image in the images/sec leg (train.py:204: 50000)")
    p.add_argument("--images-per-gpu", type=int, default=24,
                   help="images of a rank fitted concurrently in the images/sec leg (gi2d_train_steps_batched)")
    p.add_argument("--batch-groups", type=int, default=3,
                   help="... as this many batches (image i of the rank in batch i mod G), each on its own HIP stream and "
                        "host thread: one batch's update kernel overlaps another's tile pass (1 = one batch)")
    p.add_argument("--synthetic-images", action="store_true",
                   help="images/sec leg on Kodak-shaped synthetic pictures instead of the Kodak fixture")
    p.add_argument("--no-batched", action="store_true", help="skip the `batched` block")
    p.add_argument("--no-static", action="store_true", help="skip the `static_scene_step` block")
    p.add_argument("--no-dropin", action="store_true", help="skip the `dropin_autograd_step` block")
    p.add_argument("--images-per-gpu-probe", action="store_true",
                   help="also report the aggregate step rate of 2, 3 and 4 independent images stepped concurrently on "
                        "separate HIP streams of this GPU (extra information, not `value`)")
    p.add_argument("--train-step", action="store_true",
                   help="also time the quantisation-aware iterations of BASELINE config 5 after the timed region")
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

    import math
    from helpers import synth_cholesky, synth_gt
    from gaussianimage_plus_amd import _lib
    from gaussianimage_plus_amd.trainer import NativeFitter

    n, h, w = args.num_points, args.height, args.width
    # positions atanh(U), Cholesky rows U[0,1) + the low-pass bound, colours U[0,1) (the reference starts
    # them at zero; random ones exercise every branch from the first step), opacity 1; reference default seed + rank
    xyz, L, col, op = synth_cholesky(n, h, w, 3047 + rank)
    gt_np = synth_gt(h, w, 1 + rank)  # seeded smooth target image
    gt = torch.from_numpy(gt_np).to(dev)
    fit = make_fitter(gt, xyz, L, col, n, h, w)
    fit.max_call = 1 << 30  # the timed region is one C-ABI call

    def barrier():
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[dev_index])  # this rank's own GPU, stated rather than guessed
            else:
                dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        fit.train(args.warmup)
    barrier()
    red_dev = dev if (world == 1 or dist.get_backend() == "nccl") else "cpu"
    # The timed region -- exactly `steps` iterations as ONE call, a barrier + synchronize on both sides, max over ranks
    # -- is taken REPEATS times and the median reported: the shared boxes show queue stalls of tens of ms once in a few
    # hundred ms of runtime (measurement hazard), which would multiply a single 0.7 ms region.
    # Between two of them the SAME region is taken once more with HIP start/stop events attached to every
    # EVENT_STRIDE-th tile-pass dispatch (gi2d_timer_*: the kernel's own begin/end timestamps on the launch stream) --
    # the dominant kernel's duration for `roofline`.  The two are kept apart because an event pair is not free: the
    # queue spends about 5 us on every dispatch that carries one (round 6, measured: 28.3 us per iteration with a pair on
    # every 2nd tile pass, 26.7 on every 5th, 25.7 without), which rounds 1-5 charged to `value`.  Both kinds are
    # reported (`ms_per_step`, `ms_per_step_event_regions`).
    def timed_region():
        barrier()
        t0 = time.perf_counter()
        fit.train(args.steps)
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    per_region = max(1, (args.steps + EVENT_STRIDE - 1) // EVENT_STRIDE)
    regions, event_regions, kernel_us, first_of_call = [], [], [], []
    for _ in range(REPEATS):
        regions.append(timed_region())
        timers = _lib.TilePassTimers(per_region)
        timers.arm(EVENT_STRIDE)
        event_regions.append(timed_region())
        timers.cancel()
        kernel_us += timers.us()
        timers.close()
        # sample j sits on tile-pass dispatch EVENT_STRIDE * j of its region.  The FIRST tile pass of every C-ABI call
        # (a region is one call, or several of fit.max_call iterations) is another instantiation of the kernel -- it
        # follows the projection kernel that opens a call, not an update kernel, so it is built without the inbox code
        # (fast_fwdbwd_kernel<1, 0, false>) -- and is reported beside the dominant kernel, not averaged into it.
        first_of_call += [(EVENT_STRIDE * j) % fit.max_call == 0 for j in range(per_region)]
    fit.check_status()
    first_pass_us = [u for u, f in zip(kernel_us, first_of_call) if f]
    if fit.tx * fit.ty <= 1536 and len(first_pass_us) < len(kernel_us):
        kernel_us = [u for u, f in zip(kernel_us, first_of_call) if not f]
    m = int(fit.nth[:n].sum().item())  # tile intersections of the last projection (drifts as the gaussians move)

    ms = torch.tensor([float(m)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(ms, op=dist.ReduceOp.SUM)
    elapsed = sorted(regions)[len(regions) // 2]
    value = world * args.steps / elapsed

    images = images_per_s(args, rank, world, dev, red_dev, barrier) if args.images > 0 else None

    if rank == 0:
        avg_us = float(np.mean(kernel_us))
        pair_bytes = 80 * m + 36 * h * w + 36 * n  # SURVEY 8d north-star figure (fwd + bwd rasterize)
        achieved = pair_bytes / (avg_us * 1e-6) / 1e9
        # one launch of the general form up to one residency round of the chip (1536 tiles); larger images run the tile
        # pass in two launches -- the small form, then the general one on the tiles it passed over (csrc/gi2d_fast.hip):
        # the events bracket both, the stored counters are summed over both
        # (which of the two the timed calls took: gi2d_batch_tile_pass_form on the fitter's workspace -- two launches
        # while at most one row in sixteen is fuller than the small form's 128 candidates)
        two = bool(_lib.load().gi2d_batch_tile_pass_form(fit.ws.data_ptr()))
        # (third template argument: built with the code that takes entrants out of the tiles' inboxes -- every tile
        # pass of a single-image call but its first; csrc/gi2d_fast_internal.h::Inbox)
        kernel = "gi2d::fast_fwdbwd_kernel<1, 1, false> + gi2d::fast_fwdbwd_kernel<1, 2, false>" if two else \
            ("gi2d::fast_fwdbwd_kernel<1, 0, true>" if fit.tx * fit.ty <= 1536 else "gi2d::fast_fwdbwd_kernel<1, 0, false>")
        traffic, traffic_src = pmc_traffic(kernel, n, h, w)
        # the counters were collected at the intersection count of THAT run (the scene drifts while it trains, and a
        # 20-step run sits at another M than a 200-step one): the wasted-traffic ratio is formed like for like
        traffic_m = stored_num_intersects(n, h, w)
        traffic_ratio = (traffic / (80 * traffic_m + 36 * h * w + 36 * n)) if (traffic and traffic_m) else None
        line = {
            "metric": f"training iters/sec (fwd+bwd rasterize) at N Gaussians, {w}x{h}",
            "value": value,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "repeats": REPEATS,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_each_region": [r / args.steps * 1e3 for r in regions],
            # the same region with a HIP event pair on every EVENT_STRIDE-th tile pass (where `roofline` gets its kernel
            # time from): what rounds 1-5 reported as ms_per_step
            "ms_per_step_event_regions": sorted(event_regions)[len(event_regions) // 2] / args.steps * 1e3,
            "event_stride": EVENT_STRIDE,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if (args.images <= 0 or args.synthetic_images) else
                    "synthetic (step metric, batched); kodak (images_per_s)",
            "config": {
                "workload": f"Cholesky model, N={n} Gaussians, {w}x{h}, one image per GPU, whole training iterations with "
                            f"the optimizer ON (moving gaussians): tanh/+bound + project fwd + incremental tile binning + "
                            f"rasterize_sum fwd + L2 gradient of the render + rasterize_sum bwd + project bwd + Adam per "
                            f"step",
                "num_points": n, "height": h, "width": w, "num_intersects_rank0": m,
                "num_intersects_mean": float(ms.item()) / world, "seed": 3047,
                "host_path": "gi2d_train_steps: ONE C-ABI call for the timed region, 2 launches per iteration (tile pass; "
                             "gradient reduce + project bwd + Adam of this iteration fused with activations + project + "
                             "binning of the next) on persistent HBM buffers, eager launches on the current HIP stream",
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_measured_at_num_intersects": traffic_m, "traffic_over_algorithmic_at_that_count": traffic_ratio,
                "algorithmic_bytes_per_launch": pair_bytes, "avg_kernel_us": avg_us,
                "min_kernel_us": float(np.min(kernel_us)), "kernel_samples": len(kernel_us),
                "first_pass_of_a_call": {"kernel": "gi2d::fast_fwdbwd_kernel<1, 0, false>", "samples": len(first_pass_us),
                                         "avg_kernel_us": float(np.mean(first_pass_us)) if first_pass_us else None},
                "note": "not HBM-bound: each staged gaussian is reused by up to 256 pixels, so the tile pass is bound by "
                        "instruction issue and dependent latency (roofline_valu; DESIGN.md 3.4)",
            },
            "roofline_valu": valu_roofline(kernel, n, h, w, avg_us),
            "rasterize_pair": {
                "fwdbwd_kernel_us": avg_us, "algorithmic_bytes": pair_bytes, "achieved_GBps": achieved,
                # every staged (tile, gaussian) entry against every pixel of its tile, forward + backward: the NOMINAL
                # pair count of the reference's loops (forward.cu:650, backward.cu:1258), not evaluated work
                "nominal_pairs_per_s": 2 * 256.0 * m / (avg_us * 1e-6),
                "note": "HIP start/stop events of the tile-pass kernel inside timed calls of the same loop "
                        "(ms_per_step_event_regions)"},
        }
        if not args.no_static:
            line["static_scene_step"] = static_scene_rate(xyz, L, col, op, gt, n, h, w, dev)
        if not args.no_dropin:
            line["dropin_autograd_step"] = dropin_autograd_rate(gt, n)
        if not args.no_batched:
            line["batched"] = batched_rate(n, h, w, dev)
        if images is not None:
            line["images_per_s"] = images
        if args.train_step:
            line["quantized_train_step"] = quantized_train_step_rate(gt, dev)
        if args.images_per_gpu_probe:
            line["concurrent_images"] = concurrent_images_rate(n, h, w, dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(xyz, L, col, op, gt_np, h, w, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


def make_fitter(gt, xyz, L, col, n, h, w, **kw):
    """The Cholesky model on the synthetic inputs of SURVEY 8d: raw positions atanh(xyz), raw Cholesky rows L - bound
    (the fitter adds the low-pass bound back), colours as given; torch.optim.Adam, lr 1e-3."""
    import math
    import numpy as np
    import torch
    from gaussianimage_plus_amd.trainer import NativeFitter
    lp = min(h * w / (9 * math.pi * n), 300)
    init = {"xyz": torch.from_numpy(np.arctanh(xyz.astype(np.float64)).astype(np.float32)),
            "chol": torch.from_numpy(L - np.array([lp, 0, lp], np.float32)), "feat": torch.from_numpy(col)}
    return NativeFitter(gt.contiguous(), n, kind="cholesky", lr=1e-3, seed=3047, init=init, **kw)


def median_stretch_us(train, iters, reps=5):
    """Microseconds per iteration, median over `reps` stretches of `iters` iterations: the shared boxes show a host
    stall of tens of ms once in a few hundred ms of runtime, which would multiply a 10 ms measurement."""
    import torch
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train(iters)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2] / iters * 1e6


def static_scene_rate(xyz, L, col, op, gt, n, h, w, dev, steps=200):
    """What rounds 1-2 reported as `value`: HotPath.step() in a loop over FROZEN parameters (same inputs every step, so
    the incremental binning appends nothing and no parameter is written): project + bin + rasterize fwd + L2 gradient +
    rasterize bwd + project bwd, no activations, no optimizer."""
    from gaussianimage_plus_amd.hotpath import HotPath
    hp = HotPath(n, h, w, device=dev)
    hp.set_inputs(xyz, L, col, op)
    hp.set_target(gt)
    hp.forward()
    for _ in range(20):
        hp.step()

    def run(k):
        for _ in range(k):
            hp.step()
    us = median_stretch_us(run, steps)
    hp.check_status()
    return {"steps_per_s": 1e6 / us, "us_per_step": us, "num_intersects": hp.num_intersects(),
            "what": "frozen parameters: no appends to the tile lists, no optimizer (the round-1/2 headline loop)"}


def dropin_autograd_rate(gt, n, iters=200, reps=5):
    """The path a user of the reference runs: the loop of models/gaussianimage_cholesky.py:302-317 (tanh / +bound,
    project_gaussians_2d, rasterize, L2 loss, loss.backward(), torch.optim.Adam, StepLR) through the drop-in `gsplat`
    autograd wrappers, as launch.fit_image issues it -- PyTorch's own dispatcher, autograd engine and optimizer included.
    Median of `reps` fits of `iters` iterations after one warm-up fit."""
    from gaussianimage_plus_amd import launch
    from gaussianimage_plus_amd.gsplat import cuda as table, _raster_common
    launch.fit_image(gt, n, 300, eval_renders=1)  # warm-up: code objects, the caching allocator, the workspace pool
    runs = sorted(launch.fit_image(gt, n, iters, eval_renders=1)["train_s"] / iters * 1e6 for _ in range(reps))
    us = runs[len(runs) // 2]
    # the same loop with one whole iteration captured in a HIP graph and replayed (launch.fit_image(graph=True)): what is
    # left when PyTorch's per-op host work is out of the way -- longer fits, so that the three eager iterations and the
    # capture in front weigh little
    g_iters = 5 * iters
    launch.fit_image(gt, n, 300, eval_renders=1, graph=True)
    g_runs = sorted(launch.fit_image(gt, n, g_iters, eval_renders=1, graph=True)["train_s"] / g_iters * 1e6 for _ in range(3))
    g_us = g_runs[len(g_runs) // 2]
    return {"us_per_iteration": us, "iters_per_s": 1e6 / us, "us_per_iteration_min_max": [runs[0], runs[-1]], "num_points": n, "iterations": iters, "repeats": reps,
            "hip_graph": {"us_per_iteration": g_us, "iters_per_s": 1e6 / g_us, "us_per_iteration_min_max": [g_runs[0], g_runs[-1]],
                          "iterations": g_iters, "repeats": 3,
                          "what": "launch.fit_image(graph=True): three eager iterations, then render + mse_loss + backward + "
                                  "Adam step captured once with torch.cuda.graph and replayed (Adam capturable + fused: "
                                  "step count and learning rate on the device, same update rule); the wrappers record "
                                  "their passes without the host-side status protocol, tile-row overflow is checked "
                                  "between replays (every 256; gsplat/_raster_common.py::check_captured)"},
            "binding": table.BINDING, "status_check": "every forward" if _raster_common.SYNC_EVERY_FORWARD else
            "one call late (synchronous on a workspace's first use and above half the tile-row capacity)",
            "loop": "launch.fit_image: tanh, +bound, project_gaussians_2d, rasterize_gaussians_plus, clamp, mse_loss, "
                    "loss.backward(), torch.optim.Adam.step(), zero_grad, StepLR.step() -- 4 native calls per iteration "
                    "(project fwd, bin + rasterize fwd, rasterize bwd tiles + reduce, project bwd), the rest is PyTorch: "
                    "the loop is HOST-bound (profiles/round4_dropin_profile_after.txt: autograd engine, torch.optim.Adam's "
                    "foreach kernels, mse_loss / tanh / clamp dispatch = three quarters of an iteration)",
            "first_measurement_round4": {"us_per_iteration": 473.2, "binding": "ctypes", "status_check": "every forward",
                                         "source": "profiles/round4_dropin_profile_before.txt"}}


def batched_rate(n, h, w, dev, ks=(4, 8, 24), iters=60):
    """K images per launch (gi2d_train_steps_batched): the headline's training iteration for K independent images in
    lockstep.  Per K: image-iterations/s, the batched tile-pass kernel's own time (HIP events on its dispatch) and the
    roofline fraction on K x algorithmic bytes."""
    import numpy as np
    import torch
    from helpers import synth_cholesky, synth_gt
    from gaussianimage_plus_amd import _lib
    from gaussianimage_plus_amd.trainer import BatchFitter
    out = []
    for k in ks:
        fits = []
        for i in range(k):
            xyz, L, col, _ = synth_cholesky(n, h, w, 5000 + i)
            fits.append(make_fitter(torch.from_numpy(synth_gt(h, w, 50 + i)).to(dev), xyz, L, col, n, h, w))
        b = BatchFitter(fits)
        b.max_call = 1 << 30
        b.train(20)
        us = median_stretch_us(b.train, iters)
        timers = _lib.TilePassTimers(8)
        timers.arm(4)
        b.train(32)
        torch.cuda.synchronize(dev)
        kus = timers.us()
        timers.close()
        # which form those passes took (include/gi2d.h: gi2d_batch_tile_pass_form; decided from the previous call's report)
        two = bool(_lib.load().gi2d_batch_tile_pass_form(b.table.data_ptr()))
        kname = ("gi2d::fast_fwdbwd_batched_kernel<1, 1> + gi2d::fast_fwdbwd_batched_kernel<1, 2>" if two
                 else "gi2d::fast_fwdbwd_batched_kernel<1, 0>")
        for f in fits:
            f.check_status()
        m = [int(f.nth[:n].sum().item()) for f in fits]
        nbytes = sum(80 * mi + 36 * h * w + 36 * n for mi in m)
        avg = float(np.median(kus))
        out.append({"images_per_launch": k, "image_iters_per_s": k * 1e6 / us, "us_per_batch_iteration": us,
                    "us_per_image_iteration": us / k, "tile_pass_kernel_us": avg,
                    "tile_pass_us_per_image": avg / k, "algorithmic_bytes_per_launch": nbytes,
                    "tile_pass_form": "two launches (small form, then general form on fuller tiles)" if two
                    else "one launch (general form)",
                    "roofline": {"bound": "hbm", "kernel": kname,
                                 "achieved": nbytes / (avg * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": nbytes / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": pmc_traffic(kname, n, h, w, k)[0],
                                 "traffic_source": pmc_traffic(kname, n, h, w, k)[1]},
                    "roofline_valu": valu_roofline(kname, n, h, w, avg, k),
                    "num_intersects_mean": float(np.mean(m))})
        del b, fits
        torch.cuda.empty_cache()
    return {"workload": f"the headline's training iteration (Cholesky model, N={n}, {w}x{h}, Adam) for K images in "
                        f"lockstep, every kernel launched once for all of them; median of 5 stretches of {iters} "
                        f"iterations", "per_k": out}


def load_kodak(count):
    """The Kodak pictures as float [H, W, 3] tensors in [0, 1] (utils.py:21-26: PIL -> ToTensor) from the data fixture
    tests/golden/kodak24.npz (pixels of datasets/kodak/kodim01..24.png; tests/golden/make_kodak_fixture.py)."""
    import numpy as np
    import torch
    z = np.load(os.path.join(ROOT, "tests", "golden", "kodak24.npz"))
    names = sorted(z.files)[:count]
    return names, [torch.from_numpy(z[k].astype(np.float32) / 255.0) for k in names]


def images_per_s(args, rank, world, dev, red_dev, barrier):
    """BASELINE.json's second metric: the per-image fitting loop of train.py:294-340 over Kodak-24, image i on rank
    i mod N (launch.run_sharded), whole-job images / wall second (max over ranks by the closing barrier).  Model and
    schedule are train.py's defaults: covariance model, Adam lr 0.018 / eps 1e-15, 5000 -> 50000 gaussians (BASELINE
    config 3: densification on), 50 000 iterations, prune every 100, growth every 5000 with the whole remaining budget
    released at the last growth step (train.py:91-99,204-215).  A rank fits its images as one batch
    (gi2d_train_steps_batched: every kernel of an iteration launched once for all of them)."""
    import torch
    import torch.distributed as dist
    from gaussianimage_plus_amd import launch

    iters = int(args.image_iterations)
    num_points, max_points = 5000, 50000
    grow_iter, prune_iter = (5000 if iters >= 20000 else max(iters // 10, 1)), 100
    if args.synthetic_images:
        names = [f"synthetic{i:02d}" for i in range(args.images)]
        pics = [launch.synthetic_image(512, 768, 100 + i) for i in range(args.images)]
    else:
        names, pics = load_kodak(args.images)
    kw = dict(lr=0.018, seed=3047, kind="covariance", max_points=max_points, prune_iter=prune_iter, grow_iter=grow_iter,
              eps=1e-15, optimizer="adam", eval_renders=1)
    rows = {}

    def fit_one(i, img):
        rows[i] = launch.fit_image_native(img.to(dev), num_points, iters, **kw)
        return rows[i]

    def fit_group(idx, imgs):
        res = launch.fit_images_native([im.to(dev) for im in imgs], num_points, iters, batched=max(int(args.batch_groups), 1), **kw)
        rows.update(zip(idx, res))
        return res

    # untimed warm-up, the counterpart of --warmup for the step metric: two small images through the same schedule as a
    # batch, so that every kernel of the loop (batched tile pass and update, prune / growth, render) has its code object
    # loaded -- tens of ms each on first use, a fixed cost per process that would otherwise be charged to the 3 images a
    # rank fits at N = 8
    warm = [launch.synthetic_image(96, 144, 98 + i).to(dev) for i in range(2)]
    launch.fit_images_native(warm, 500, 300, lr=0.018, seed=3047, kind="covariance", max_points=1500, prune_iter=100,
                             grow_iter=100, eps=1e-15, optimizer="adam", eval_renders=1, batched=True)
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
    mine = sorted(rows)
    portrait = sum(1 for p in pics if p.shape[0] > p.shape[1])
    # what every rank actually fitted, so that a scaling run reads without guessing: image i -> rank i mod N, a rank's
    # images split into min(batch_groups, images) batches (3 images per rank at N = 8 = three one-image batches)
    share = {"rank": rank, "images": len(mine), "batches": min(max(1, args.batch_groups), max(1, len(mine))) if mine else 0,
             "images_per_batch": [len(mine[g::max(1, min(max(1, args.batch_groups), len(mine)))])
                                  for g in range(min(max(1, args.batch_groups), len(mine)))] if mine else []}
    per_rank = [share]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, share)
    return {"value": out["images"] / wall, "unit": "images/s", "images": out["images"], "wall_s": wall,
            "data": "synthetic" if args.synthetic_images else "kodak",
            "images_landscape_768x512": len(pics) - portrait, "images_portrait_512x768": portrait,
            "iterations_per_image": iters, "avg_psnr": out["avg_psnr"],
            "avg_num_gaussians": out["avg_num_gaussians"],
            "avg_num_gaussians_note": "gaussians of the evaluated (best-PSNR) model, as train.py:157-160 reports them",
            "rank0_images": [{"image": names[i], "psnr": round(rows[i]["psnr"], 3),
                              "best_model_gaussians": int(rows[i]["num_gaussians"]),
                              "final_model_gaussians": int(rows[i].get("final_num_gaussians", rows[i]["num_gaussians"]))}
                             for i in mine],
            "images_concurrent_per_gpu": min(max(1, args.images_per_gpu), max(1, len(mine))),
            "batches_per_gpu": min(max(1, args.batch_groups), max(1, len(mine))),
            "per_rank": per_rank,
            "warmup": "two 144x96 images, 300 iterations of the same schedule as one batch, untimed (code objects loaded)",
            "workload": f"{len(pics)} {'synthetic 768x512' if args.synthetic_images else 'Kodak'} images, covariance "
                        f"model {num_points}->{max_points} gaussians, {iters} iterations/image (train.py:204: 50000), "
                        f"prune every {prune_iter}, grow every {grow_iter}; image i -> rank i mod {world}, a rank's images "
                        f"fitted concurrently as {max(1, args.batch_groups)} batches (every kernel of an iteration "
                        f"launched once per batch) on as many HIP streams"}


def pmc_traffic(kernel, n, h, w, images_per_launch=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/traffic.json, built by
    tools/make_profiles_rounds.py: 2*FETCH_SIZE + WRITE_SIZE per the gfx950 correction of MI355X_MICROARCH.md; one section per
    profiled workload) and where the figure comes from.  Hardware counters cannot be read from inside the timed
    process, so this is a STORED value of the same command under rocprofv3, labelled as such; null when no counters were
    collected for this workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        for wl in t["workloads"]:
            c = wl.get("config") or {}
            if (c.get("num_points"), c.get("height"), c.get("width")) != (n, h, w) or \
                    c.get("images_per_launch") != images_per_launch:
                continue
            total = 0
            for part in kernel.split(" + "):
                hit = [v for v in wl["kernels"].values()
                       if v["kernel"].replace(" ", "").endswith(part.replace(" ", "")) and "hbm_bytes_per_launch" in v]
                if not hit:
                    return None, f"none: profiles/traffic.json, workload {wl['workload']}, has no entry for {part}"
                total += hit[0]["hbm_bytes_per_launch"]
            return total, f"stored: profiles/traffic.json, workload {wl['workload']} " \
                          f"({t.get('source', 'rocprofv3 --pmc')}), not measured in this run"
        return None, f"none: profiles/traffic.json holds no counters for {n} gaussians at {w}x{h}" + \
                     (f", {images_per_launch} images per launch" if images_per_launch else "")
    except (OSError, KeyError, ValueError, TypeError) as e:
        return None, f"none: profiles/traffic.json unreadable ({type(e).__name__})"


def stored_num_intersects(n, h, w, images_per_launch=None):
    """The intersection count M of the run whose counters profiles/traffic.json holds for this workload (None: unknown)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        for wl in t["workloads"]:
            c = wl.get("config") or {}
            if (c.get("num_points"), c.get("height"), c.get("width")) == (n, h, w) and \
                    c.get("images_per_launch") == images_per_launch:
                return c.get("num_intersects_rank0")
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None


def _stored_kernel(kernel, n, h, w, images_per_launch=None):
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for wl in t["workloads"]:
        c = wl.get("config") or {}
        if (c.get("num_points"), c.get("height"), c.get("width")) != (n, h, w) or \
                c.get("images_per_launch") != images_per_launch:
            continue
        out = None
        for part in kernel.split(" + "):  # a tile pass in two launches: counters summed
            hit = [v for v in wl["kernels"].values() if v["kernel"].replace(" ", "").endswith(part.replace(" ", ""))]
            if not hit or "valu" not in hit[0]:
                return t, wl, None
            if out is None:
                out = {"valu": dict(hit[0]["valu"])}
            else:
                for c, x in hit[0]["valu"].items():
                    out["valu"][c] = out["valu"].get(c, 0) + x
        return t, wl, out
    return t, None, None


def valu_roofline(kernel, n, h, w, kernel_us, images_per_launch=None):
    """The VALU side of the tile pass (what DESIGN.md 3.4 argues it is bound by, next to the mandated HBM figure):
    wave-instructions per image and the time the SIMDs spend issuing them -- SQ_ACTIVE_INST_VALU quad-cycles x 4 /
    1024 SIMDs / shader clock -- against the kernel's duration measured live in this run.  The counters are STORED
    values of the same command under `rocprofv3 --pmc` (profiles/traffic.json, tools/profile_rounds.sh); the useful-lane
    fractions come from tools/lane_model.py (the kernels' own scheduling rules replayed on the scene in numpy)."""
    try:
        t, wl, v = _stored_kernel(kernel, n, h, w, images_per_launch)
        valu = (v or {}).get("valu")
        if not valu:
            return {"source": "none: profiles/traffic.json holds no VALU counters for this workload"}
        k = images_per_launch or 1
        issue_us = valu["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS / (CLOCK_GHZ * 1e3)
        out = {"bound": "valu issue + dependent latency", "insts_per_image": valu["SQ_INSTS_VALU"] / k,
               "issue_us": issue_us, "kernel_us": kernel_us, "util": issue_us / kernel_us,
               "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
               "source": f"stored: profiles/traffic.json, workload {wl['workload']} ({t.get('source', 'rocprofv3 --pmc')}); "
                         f"kernel_us measured in this run"}
        for key in ("useful_lane_frac_fwd", "useful_lane_frac_bwd", "lane_model"):
            if key in (wl.get("lane_model") or {}):
                out[key] = wl["lane_model"][key]
        return out
    except (OSError, KeyError, ValueError, TypeError) as e:
        return {"source": f"none: profiles/traffic.json unreadable ({type(e).__name__})"}


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
