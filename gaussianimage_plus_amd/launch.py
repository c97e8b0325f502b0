"""One-image-per-GPU fitting launcher.

The reference fits images one after another on `cuda:0` (train.py:294-308) -- there is no parallelism to
port.  Images are independent optimisation problems, so the MI355X-native scale-out is the trivial one:
one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m gaussianimage_plus_amd.launch ...`),
image i goes to rank i mod world_size, every rank runs the unchanged per-image loop on its own device, and a
single RCCL all-reduce of five floats at the end reproduces the reference's "Average:" line
(train.py:327-340).  No gradient or parameter ever crosses xGMI.

The per-image loop here is the Cholesky training step of models/gaussianimage_cholesky.py:302-317 written
against the drop-in `gsplat` surface (tanh / +bound / project / rasterize / clamp / L2 / Adam); it exists so
that images/sec can be measured on a box that does not have the reference checked out.  With the reference
on PYTHONPATH, its own train.py runs unmodified on top of the same operators.
"""
from __future__ import annotations

import argparse
import math
import os
import time
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


# per-image results of the quantised loop that ride along in the same five-float-style all-reduce
CODEC_KEYS = ("bpp", "position_bpp", "cholesky_bpp", "feature_dc_bpp", "bpp_wc", "psnr_decoded")


def partition(num_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin shard of image indices: Kodak-24 -> 24/12/6/3 images per rank at 1/2/4/8 GPUs."""
    return list(range(rank, num_items, world_size))


def reduce_metrics(local: Dict[str, float], device="cpu") -> Dict[str, float]:
    """Sum the per-rank totals (psnr, train seconds, eval seconds, gaussians, images) across ranks with one
    all-reduce and return the averages the reference logs."""
    keys = ["psnr", "train_s", "eval_s", "num_gaussians", "count"] + list(CODEC_KEYS)
    t = torch.tensor([float(local.get(k, 0.0)) for k in keys], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    tot = dict(zip(keys, t.tolist()))
    n = max(tot["count"], 1.0)
    out = {"images": int(tot["count"]), "avg_psnr": tot["psnr"] / n, "avg_train_s": tot["train_s"] / n,
           "avg_eval_s": tot["eval_s"] / n, "avg_num_gaussians": tot["num_gaussians"] / n,
           "sum_train_s": tot["train_s"]}
    for k in CODEC_KEYS:  # train_quantize.py:411-420 averages the sizes over the images too
        out["avg_" + k] = tot[k] / n
    return out


def run_sharded(items: Sequence, fit_one: Callable[[int, object], Dict[str, float]], rank: int, world_size: int,
                device="cpu", group: int = 1, fit_group: Optional[Callable] = None) -> Dict[str, float]:
    """Fit this rank's shard of `items` and reduce the metrics: one at a time with `fit_one(index, item)`, or
    `group` at a time with `fit_group(indices, items) -> list of rows` (several images concurrently on one GPU;
    their shared wall time is counted once)."""
    local = {"psnr": 0.0, "train_s": 0.0, "eval_s": 0.0, "num_gaussians": 0.0, "count": 0.0}
    rows = []
    mine = partition(len(items), rank, world_size)
    for g0 in range(0, len(mine), max(group, 1)):
        idx = mine[g0:g0 + max(group, 1)]
        res = fit_group(idx, [items[i] for i in idx]) if (fit_group and group > 1) else [fit_one(i, items[i]) for i in idx]
        shared = fit_group is not None and group > 1
        for k, (i, r) in enumerate(zip(idx, res)):
            rows.append((i, r))
            for key in ("psnr", "eval_s", "num_gaussians") + CODEC_KEYS:
                local[key] = local.get(key, 0.0) + float(r.get(key, 0.0))
            if not shared or k == 0:
                local["train_s"] += float(r.get("train_s", 0.0))
            local["count"] += 1
    out = reduce_metrics(local, device)
    out["rows"] = rows
    return out


# ----------------------------------------------------------------------------------- per-image loop
def fit_images_native(gts: Sequence[torch.Tensor], num_points: int, iterations: int, lr: float = 1e-3,
                      seed: int = 3047, eval_renders: int = 10, kind: str = "cholesky", max_points: int = 0,
                      prune_iter: int = 100, grow_iter: int = 5000, eps: float = 1e-8,
                      chunk: int = 16, optimizer: str = "adam", quantize: bool = False, warmup_iter: int = 6000,
                      bits=(12, 10, 6), threaded: bool = False, batched=False) -> List[Dict[str, float]]:
    """Fit the images of `gts` CONCURRENTLY on one GPU on the fused training iteration (trainer.NativeFitter ->
    gi2d_train_step: one C-ABI call, three kernel launches, no host synchronisation per iteration).  With `batched`
    the images run in lockstep and every kernel of an iteration is launched ONCE
    for all of them (trainer.BatchFitter -> gi2d_train_steps_batched); `batched` = G > 1 makes G such batches (image i in
    batch i mod G), each on its own HIP stream and host thread: a batch runs its two kernels strictly one after the
    other, and the second, one lane per gaussian, is dependent-load latency -- another batch's tile pass fills that time.
    Otherwise one HIP stream per image:  One image's kernels leave most of the chip idle between their dependent phases, so
    two to four independent images per GPU raise the aggregate iteration rate by 1.4-3.6x (the smaller the model,
    the more); `chunk` iterations of one image are enqueued before the host turns to the next, or, with `threaded`,
    every image has its own host thread (the C-ABI calls release the GIL).  With `max_points` > `num_points` (covariance model)
    each image runs the adaptive loop of train.py:120-160: prune every `prune_iter`, grow every `grow_iter`, keep
    the best model on the device and evaluate that one.  `train_s` is the wall time of the whole group.
    With `quantize` (covariance model) the loop is train_quantize.py's: plain fitting up to `warmup_iter`, then
    quantisation-aware iterations with `bits` = (xy, covariance, colour) bit depths; the result rows then also carry
    the size of the encoding in bits per pixel (analysis_wo_ec) and the PSNR of the decoded image."""
    from .trainer import BatchFitter, NativeFitter

    dev = gts[0].device
    adaptive = kind == "covariance" and max_points > num_points
    if quantize and kind == "cholesky":
        raise ValueError("quantisation-aware fitting is wired for the covariance model (train_quantize.py) and the "
                         "rotation-scale model (models/gaussianimage_rs.py:131-163)")
    fitters = [NativeFitter(gt, num_points, kind=kind, lr=lr, seed=seed, eps=eps, optimizer=optimizer,
                            max_points=max_points if adaptive else None, track_best=adaptive or quantize,
                            device_resident=adaptive)
               for gt in gts]
    # batched: True = one batch, an int G > 1 = G batches of every G-th image, each on its own HIP stream and host thread
    groups = (1 if batched else 0) if isinstance(batched, bool) else max(int(batched), 0)
    if len(fitters) < 2:
        groups = 0
    groups = min(groups, len(fitters))
    batched = groups >= 1
    if groups > 1:
        group_streams = [torch.cuda.Stream(device=dev) for _ in range(groups)]
        streams = [group_streams[i % groups] for i in range(len(fitters))]
    else:
        streams = ([torch.cuda.Stream(device=dev) for _ in fitters] if len(fitters) > 1 and not batched
                   else [torch.cuda.current_stream(dev)] * len(fitters))
    torch.cuda.synchronize(dev)
    t0 = time.time()
    sched_kw = dict(prune_iter=prune_iter, grow_iter=grow_iter, adaptive_add=adaptive,
                    max_points=max_points if adaptive else None,
                    chunk=chunk if (len(fitters) > 1 and not threaded) else None)
    if batched:
        sched_kw["chunk"] = None
        runs = []
        def fit_batch(part):
            # (a batch of ONE image -- the three-image shard of an 8-GPU Kodak run is three of them -- takes the
            # single-image calls: same results bit for bit, no table lookups)
            runner = BatchFitter(part) if len(part) > 1 else part[0]
            if quantize:  # train_quantize.py's loop for the whole batch
                for _ in runner.fit_quantize_schedule(iterations, warmup_iter, bits=bits, **sched_kw):
                    pass
            else:
                runner.fit(iterations, **sched_kw)

        if groups == 1:
            with torch.cuda.device(dev):
                fit_batch(fitters)
        else:
            # one batch alone runs its two kernels strictly one after the other, and the per-gaussian update kernel is
            # dependent-load latency (on trained scenes with a tail: the waves that hold the large gaussians); a second
            # batch on another stream fills that time with its tile pass
            import threading

            batch_errors: List[BaseException] = []

            def drive_group(g):
                try:
                    with torch.cuda.device(dev), torch.cuda.stream(group_streams[g]):
                        fit_batch(fitters[g::groups])
                except BaseException as e:  # surfaced on the main thread below
                    batch_errors.append(e)

            group_workers = [threading.Thread(target=drive_group, args=(g,)) for g in range(groups)]
            for t in group_workers:
                t.start()
            for t in group_workers:
                t.join()
            if batch_errors:
                raise batch_errors[0]
    elif quantize:
        runs = [f.fit_quantize_schedule(iterations, warmup_iter, bits=bits, **sched_kw) for f in fitters]
    else:
        runs = [f.fit_schedule(iterations, **sched_kw) for f in fitters]
    if batched:
        pass
    elif threaded and len(fitters) > 1:
        # one host thread per image: the C-ABI calls release the GIL, so the launches of different images are issued
        # in parallel instead of round-robin from one thread (which becomes the limit beyond ~4 small images)
        import threading

        errors: List[BaseException] = []

        def drive(i):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(streams[i]):
                    for _ in runs[i]:
                        pass
            except BaseException as e:  # surfaced on the main thread below
                errors.append(e)

        workers = [threading.Thread(target=drive, args=(i,)) for i in range(len(fitters))]
        for t in workers:
            t.start()
        for t in workers:
            t.join()
        if errors:
            raise errors[0]
    else:
        live = list(range(len(fitters)))
        while live:
            for i in list(live):
                with torch.cuda.stream(streams[i]):
                    if next(runs[i], None) is None:
                        live.remove(i)
    torch.cuda.synchronize(dev)
    train_s = time.time() - t0
    out = []
    for f, st in zip(fitters, streams):
        with torch.cuda.stream(st):
            f.check_status()
            final_n = f.n  # population of the last iteration's model (the evaluated one below is the best-PSNR snapshot)
            if adaptive or quantize:
                f.load_best()
            # the render-only kernels have not run yet (training uses the single-pass tile kernel): their code object
            # loads on first use, tens of ms that are no part of a frame time (the reference's forward kernel is the one
            # its training loop has been running, train.py:150-155)
            f.render()
            st.synchronize()
            t0 = time.time()
            for _ in range(max(int(eval_renders), 1)):  # at least one render: the PSNR below needs it
                img = f.render()
            st.synchronize()
            eval_s = (time.time() - t0) / max(int(eval_renders), 1)
            mse = torch.nn.functional.mse_loss(img, f.gt).item()
        row = {"psnr": 10 * math.log10(1.0 / max(mse, 1e-12)), "train_s": train_s, "eval_s": eval_s,
               "num_gaussians": f.n, "final_num_gaussians": final_n, "mse": mse}
        if quantize:  # train_quantize.py:239-270 encode(): codes, decoded render, size
            with torch.cuda.stream(st):
                enc = f.compress_wo_ec()
                dec = f.decompress_wo_ec(enc)
                row.update(f.analysis_wo_ec(enc, entropy_estimate=True))
                row["psnr_decoded"] = 10 * math.log10(1.0 / max(torch.nn.functional.mse_loss(dec, f.gt).item(), 1e-12))
                row["num_gaussians"] = f.n
        out.append(row)
    return out


def fit_image_native(gt_hwc: torch.Tensor, num_points: int, iterations: int, **kw) -> Dict[str, float]:
    """One image on the native loop (fit_images_native with a group of one)."""
    return fit_images_native([gt_hwc], num_points, iterations, **kw)[0]


def fit_image(gt_hwc: torch.Tensor, num_points: int, iterations: int, lr: float = 1e-3, seed: int = 3047,
              eval_renders: int = 10, graph: bool = False) -> Dict[str, float]:
    """Cholesky model, L2 loss, Adam (models/gaussianimage_cholesky.py:57-58,80-82,302-317 with
    opt_type="adam") through the autograd wrappers.  gt_hwc: float32 [H, W, 3] in [0, 1] on the target GPU.

    `graph`: the same loop, but one whole iteration -- render, loss, backward, optimizer step -- is captured in a HIP
    graph (torch.cuda.graph) after three eager iterations and REPLAYED for the rest.  The eager loop is bound by the
    host (PyTorch's dispatcher, autograd engine and optimizer issue ~35 small kernels per iteration); a replay is one
    launch.  What changes: Adam runs with `capturable=True` (step count and learning rate live on the device; same
    update rule), the wrappers record their passes without the host-side status protocol, and the tile-row overflow
    check happens between replays (every 256, _raster_common.check_captured), where it can only raise -- the eager
    loop's exact fallback cannot run inside a graph."""
    from .gsplat.project_gaussians_2d import project_gaussians_2d
    from .gsplat.rasterize_sum_plus import rasterize_gaussians_plus

    dev = gt_hwc.device
    h, w = int(gt_hwc.shape[0]), int(gt_hwc.shape[1])
    tile_bounds = ((w + 15) // 16, (h + 15) // 16, 1)
    g = torch.Generator(device="cpu").manual_seed(seed)
    xyz = torch.atanh(2 * (torch.rand(num_points, 2, generator=g) - 0.5)).to(dev).requires_grad_(True)
    chol = torch.rand(num_points, 3, generator=g).to(dev).requires_grad_(True)
    feat = torch.zeros(num_points, 3, device=dev, requires_grad=True)
    opacity = torch.ones(num_points, 1, device=dev)
    low_pass = min(h * w / (9 * math.pi * num_points), 300)
    bound = torch.tensor([low_pass, 0.0, low_pass], device=dev).view(1, 3)
    background = torch.ones(3, device=dev)
    if graph:
        opt = torch.optim.Adam([xyz, chol, feat], lr=torch.tensor(float(lr), device=dev), capturable=True,
                               fused=os.environ.get("GI2D_GRAPH_FUSED_ADAM", "1") == "1")
    else:
        opt = torch.optim.Adam([xyz, chol, feat], lr=lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=20000, gamma=0.5)

    def render():
        xys, depths, radii, conics, nth = project_gaussians_2d(torch.tanh(xyz), chol + bound, h, w, tile_bounds)
        img = rasterize_gaussians_plus(xys, depths, radii, conics, nth, feat, opacity, h, w, 16, 16,
                                       background=background)
        return torch.clamp(img, 0, 1)

    def iteration():
        loss = torch.nn.functional.mse_loss(render(), gt_hwc)
        loss.backward()
        opt.step()

    torch.cuda.synchronize(dev)
    t0 = time.time()
    if graph and iterations > 3:
        from .gsplat import _raster_common
        side = torch.cuda.Stream(device=dev)  # (torch.cuda.graph wants the warm-up off the default stream)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                iteration()
                opt.zero_grad(set_to_none=True)
                sched.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        captured = torch.cuda.CUDAGraph()
        with torch.cuda.graph(captured):  # gradients are allocated here, in the graph's pool, and rewritten by every replay
            iteration()
        for it in range(3, iterations):
            captured.replay()
            sched.step()
            if (it & 255) == 0:
                _raster_common.check_captured()
        _raster_common.check_captured()
    else:
        for _ in range(iterations):
            iteration()
            opt.zero_grad(set_to_none=True)
            sched.step()
    torch.cuda.synchronize(dev)
    train_s = time.time() - t0
    with torch.no_grad():
        t0 = time.time()
        for _ in range(eval_renders):
            img = render()
        torch.cuda.synchronize(dev)
        eval_s = (time.time() - t0) / max(eval_renders, 1)
        mse = torch.nn.functional.mse_loss(img, gt_hwc).item()
    from .gsplat import _raster_common as _rc
    _rc.settle_all()  # the last render's status words (read one call late otherwise: no call follows)
    psnr = 10 * math.log10(1.0 / max(mse, 1e-12))
    return {"psnr": psnr, "train_s": train_s, "eval_s": eval_s, "num_gaussians": num_points, "mse": mse}


def synthetic_image(h: int, w: int, seed: int) -> torch.Tensor:
    """Seeded smooth test picture in [0,1], [H,W,3] (stands in for a Kodak image when no dataset is present)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    img = torch.zeros(h, w, 3)
    for c in range(3):
        for _ in range(6):
            fx, fy = (torch.rand(2, generator=g) * 5.5 + 0.5).tolist()
            p0, p1 = (torch.rand(2, generator=g) * 2 * math.pi).tolist()
            img[..., c] += torch.sin(2 * math.pi * fx * xx / w + p0) * torch.cos(2 * math.pi * fy * yy / h + p1)
    img = (img - img.min()) / (img.max() - img.min())
    return img.contiguous()


def load_images(path: str | None, count: int, h: int, w: int) -> List[torch.Tensor]:
    if path and path.endswith(".npz"):  # a pixel fixture: one uint8 [H, W, 3] array per image (tests/golden/kodak24.npz)
        import numpy as np
        z = np.load(path)
        return [torch.from_numpy(z[k].astype("float32") / 255.0) for k in sorted(z.files)[:count or None]]
    if path:
        import numpy as np
        from PIL import Image
        files = sorted(f for f in os.listdir(path) if f.lower().endswith((".png", ".jpg", ".jpeg")))
        return [torch.from_numpy(np.asarray(Image.open(os.path.join(path, f)).convert("RGB"), dtype="float32") / 255.0)
                for f in files]
    return [synthetic_image(h, w, 100 + i) for i in range(count)]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--dataset", type=str, default=None,
                    help="directory of images (e.g. datasets/kodak) or an .npz of uint8 [H,W,3] arrays (tests/golden/kodak24.npz)")
    ap.add_argument("--synthetic", type=int, default=24, help="number of synthetic 768x512 images if no dataset")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--num_points", type=int, default=5000)
    ap.add_argument("--iterations", type=int, default=2000)
    ap.add_argument("--lr", type=float, default=None,
                    help="default: train.py's choice for the model (covariance 0.018, otherwise 0.001)")
    ap.add_argument("--opt_type", choices=["adam", "adan"], default=None,
                    help="default: train.py's choice for the model (covariance: adam, otherwise adan; main():251-257)")
    ap.add_argument("--seed", type=int, default=3047)
    ap.add_argument("--model", choices=["cholesky", "covariance", "scale_rot"], default="cholesky",
                    help="covariance = train.py's default model (pixel coordinates, lr 0.018, Adam eps 1e-15); "
                         "scale_rot = the rotation-scale parameterisation of models/gaussianimage_rs.py")
    ap.add_argument("--max_num_points", type=int, default=0,
                    help="> num_points: adaptive growth/pruning as in train.py (covariance model, native loop)")
    ap.add_argument("--prune_iter", type=int, default=100)
    ap.add_argument("--grow_iter", type=int, default=5000)
    ap.add_argument("--images_per_gpu", type=int, default=1,
                    help="images fitted concurrently on each GPU (native loop): in lockstep with one launch per kernel for "
                         "all of them (plain AND --quantize fits, as --batch_groups batches), or, with --streams, one HIP "
                         "stream and one host thread each")
    ap.add_argument("--single_host_thread", action="store_true",
                    help="issue the concurrent images' launches round-robin from one host thread instead")
    ap.add_argument("--batch_groups", type=int, default=3,
                    help="plain fitting: the concurrent images of a GPU run as this many batches (every kernel of an "
                         "iteration launched once per batch), each batch on its own HIP stream and host thread, so that one "
                         "batch's per-gaussian update kernel overlaps another's tile pass (measured on Kodak: 3 batches "
                         "beat 1 by 5 % at 24 images per GPU and by 37 % at 3)")
    ap.add_argument("--streams", action="store_true",
                    help="fit the --images_per_gpu concurrent images on one HIP stream each instead of in batched launches "
                         "(gi2d_train_steps_batched: one launch per kernel for all of them, the default for plain fitting)")
    ap.add_argument("--quantize", action="store_true",
                    help="train_quantize.py's loop (covariance model, or scale_rot with that model's quantiser set): "
                         "plain fitting up to --warmup_iter, then quantisation-aware iterations; reports bits per pixel "
                         "and the PSNR of the decoded image")
    ap.add_argument("--warmup_iter", type=int, default=6000)
    ap.add_argument("--xy_bit", type=int, default=12)
    ap.add_argument("--cov_bit", type=int, default=None,
                    help="covariance model: HybirdQuant bits (default 10); scale_rot: bits of the scaling quantiser "
                         "(default 6, models/gaussianimage_rs.py:142)")
    ap.add_argument("--color_bit", type=int, default=6)
    ap.add_argument("--loop", choices=["native", "autograd"], default="native",
                    help="native: fused training iteration (gi2d_train_step); autograd: gsplat wrappers + torch Adam")
    ap.add_argument("--graph", action="store_true",
                    help="--loop autograd only: capture one whole iteration (render, loss, backward, Adam step) in a HIP "
                         "graph after three eager iterations and replay it (fit_image(graph=True))")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if use_gpu else "gloo", rank=rank, world_size=world)
    if not use_gpu:
        raise SystemExit("gaussianimage_plus_amd.launch fits on the GPU; no device found")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    images = load_images(args.dataset, args.synthetic, args.height, args.width)

    cov = args.model == "covariance"
    if args.lr is None:
        args.lr = 0.018 if cov else 0.001
    if args.opt_type is None:
        args.opt_type = "adam" if cov else "adan"
    if args.cov_bit is None:
        args.cov_bit = 10 if cov else 6
    if args.quantize and args.model != "cholesky" and args.opt_type != "adam":
        args.opt_type = "adam"  # training_setup(quantize=True) rebuilds the optimizer as Adam; run the warm-up on it too
    # eps as the reference passes it: 1e-15 only for the covariance model's grouped Adam
    # (models/gaussianimage_covariance.py:98); Adan(self.parameters(), lr) and the plain Adam keep the 1e-8 default
    # (optimizer.py:69, models/gaussianimage_rs.py:111, models/gaussianimage_cholesky.py:113,129)
    eps = 1e-15 if (cov and args.opt_type == "adam") else 1e-8
    native_kw = dict(lr=args.lr, seed=args.seed, kind=args.model, max_points=args.max_num_points,
                     prune_iter=args.prune_iter, grow_iter=args.grow_iter, eps=eps, optimizer=args.opt_type,
                     quantize=args.quantize, warmup_iter=args.warmup_iter,
                     bits=(args.xy_bit, args.cov_bit, args.color_bit))
    if args.quantize and (args.model == "cholesky" or args.loop != "native"):
        raise SystemExit("--quantize needs --model covariance or scale_rot and the native loop")

    def report(i, img, r):
        print(f"[rank {rank}] image {i}: {img.shape[0]}x{img.shape[1]}, PSNR:{r['psnr']:.4f}, "
              f"Training:{r['train_s']:.4f}s, Eval:{r['eval_s']:.8f}s, FPS:{1.0 / r['eval_s']:.4f}, "
              f"gaussians:{int(r['num_gaussians'])}" +
              (f", bpp:{r['bpp']:.4f} (position {r['position_bpp']:.4f}, cholesky {r['cholesky_bpp']:.4f}, "
               f"feature_dc {r['feature_dc_bpp']:.4f})" +
               (f", entropy-coded estimate bpp_wc:{r['bpp_wc']:.4f}" if "bpp_wc" in r else "") +
               f", decoded PSNR:{r['psnr_decoded']:.4f}" if "bpp" in r else ""),
              flush=True)

    def fit_one(i, img):
        if args.loop == "native":
            r = fit_image_native(img.to(dev), args.num_points, args.iterations, **native_kw)
        else:
            r = fit_image(img.to(dev), args.num_points, args.iterations, lr=args.lr, seed=args.seed, graph=args.graph)
        report(i, img, r)
        return r

    def fit_group(idx, imgs):
        res = fit_images_native([im.to(dev) for im in imgs], args.num_points, args.iterations,
                                threaded=not args.single_host_thread,
                                batched=False if args.streams else max(int(args.batch_groups), 1), **native_kw)
        for i, im, r in zip(idx, imgs, res):
            report(i, im, r)
        return res

    t0 = time.time()
    group = args.images_per_gpu if args.loop == "native" else 1
    out = run_sharded(images, fit_one, rank, world, device=dev, group=group, fit_group=fit_group)
    if world > 1:
        dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()
    wall = time.time() - t0
    if rank == 0:
        sizes = sorted({(int(im.shape[1]), int(im.shape[0])) for im in images})
        size = f"{sizes[0][0]}x{sizes[0][1]}" if len(sizes) == 1 else f"{len(sizes)} image sizes"
        print(f"Average: {size}, PSNR:{out['avg_psnr']:.4f}, Training:{out['avg_train_s']:.4f}s, "
              f"Eval:{out['avg_eval_s']:.8f}s, FPS:{1.0 / max(out['avg_eval_s'], 1e-12):.4f}, "
              f"images:{out['images']}, gpus:{world}, wall:{wall:.2f}s, images/sec:{out['images'] / wall:.4f}" +
              (f", bpp:{out['avg_bpp']:.4f}" +
               # the rotation-scale model has no entropy-coded size (models/gaussianimage_rs.py has no analysis_wc)
               (f", bpp_wc (estimate):{out['avg_bpp_wc']:.4f}" if out["avg_bpp_wc"] > 0 else "") +
               f", decoded PSNR:{out['avg_psnr_decoded']:.4f}" if args.quantize else ""), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
