"""CPU-only, world_size 2 over gloo: the one-image-per-rank sharding and the single metric all-reduce of
gaussianimage_plus_amd.launch (the N>1 path of SURVEY.md section 8e; no data-path collective)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussianimage_plus_amd.launch import partition, reduce_metrics, run_sharded


def test_partition_is_a_round_robin_cover():
    for n, world in [(24, 1), (24, 2), (24, 4), (24, 8), (5, 8), (0, 2)]:
        shards = [partition(n, r, world) for r in range(world)]
        assert sorted(i for s in shards for i in s) == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert [len(partition(24, r, 8)) for r in range(8)] == [3] * 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = list(range(7))  # 7 "images": rank 0 gets 0,2,4,6 and rank 1 gets 1,3,5
    seen = []

    def fit_one(i, item):
        seen.append(i)
        return {"psnr": 30.0 + i, "train_s": 1.0 + 0.5 * i, "eval_s": 0.001, "num_gaussians": 100 * (i + 1)}

    out = run_sharded(items, fit_one, rank, world, device="cpu")
    q.put((rank, seen, {k: v for k, v in out.items() if k != "rows"}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_images_and_agree_on_the_average():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    a, b = res[0][2], res[1][2]
    assert a == b  # every rank holds the same reduced metrics
    assert a["images"] == 7
    assert abs(a["avg_psnr"] - (30.0 + 3.0)) < 1e-9
    assert abs(a["sum_train_s"] - sum(1.0 + 0.5 * i for i in range(7))) < 1e-9
    assert abs(a["avg_num_gaussians"] - 400.0) < 1e-9


def test_single_process_needs_no_process_group():
    out = reduce_metrics({"psnr": 60.0, "train_s": 4.0, "eval_s": 0.2, "num_gaussians": 10, "count": 2})
    assert out["images"] == 2 and out["avg_psnr"] == 30.0 and out["avg_train_s"] == 2.0


def test_grouped_images_share_their_wall_time():
    """--images_per_gpu: a rank fits its shard `group` images at a time; the group's wall time counts once."""
    items = list(range(5))
    calls = []

    def fit_group(idx, its):
        calls.append(list(idx))
        return [{"psnr": 30.0 + i, "train_s": 2.0, "eval_s": 0.001, "num_gaussians": 10} for i in idx]

    out = run_sharded(items, lambda i, it: (_ for _ in ()).throw(AssertionError("ungrouped")), 0, 1, group=2,
                      fit_group=fit_group)
    assert calls == [[0, 1], [2, 3], [4]]
    assert out["images"] == 5 and abs(out["sum_train_s"] - 6.0) < 1e-12  # three groups of 2.0 s
    assert abs(out["avg_psnr"] - 32.0) < 1e-12
    assert [i for i, _ in out["rows"]] == [0, 1, 2, 3, 4]


def test_codec_metrics_are_averaged_with_the_rest():
    """Rows of the quantised loop carry sizes and the decoded PSNR; they ride in the same reduction."""
    items = list(range(4))

    def fit_one(i, item):
        return {"psnr": 30.0 + i, "train_s": 1.0, "eval_s": 0.01, "num_gaussians": 100, "bpp": 0.5 + 0.1 * i,
                "bpp_wc": 0.4, "psnr_decoded": 29.0 + i, "position_bpp": 0.2, "cholesky_bpp": 0.2, "feature_dc_bpp": 0.1}

    out = run_sharded(items, fit_one, 0, 1)
    assert abs(out["avg_bpp"] - 0.65) < 1e-12 and abs(out["avg_psnr_decoded"] - 30.5) < 1e-12
    assert abs(out["avg_bpp_wc"] - 0.4) < 1e-12
    plain = run_sharded(items, lambda i, it: {"psnr": 30.0, "train_s": 1.0, "eval_s": 0.01, "num_gaussians": 1}, 0, 1)
    assert plain["avg_bpp"] == 0.0  # rows without a codec report nothing


def _worker8(rank, world, port, q, counts):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for count in counts:  # several "datasets" through the same process group, one after the other
        seen = []

        def fit_one(i, item):
            seen.append(i)
            return {"psnr": 20.0 + i, "train_s": 0.25 * (i + 1), "eval_s": 0.002, "num_gaussians": 1000 + i}

        out = run_sharded(list(range(count)), fit_one, rank, world, device="cpu")
        res.append((seen, {k: v for k, v in out.items() if k != "rows"}))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_rehearsal_even_uneven_and_idle_ranks():
    """The N = 8 layout of SURVEY 8e rehearsed on CPU (gloo): Kodak-24 gives every rank 3 images; 21 images give
    3,3,3,3,3,2,2,2; 5 images leave three ranks with NOTHING to fit -- they still join the one all-reduce (no deadlock)
    and contribute zeros, so the average is over the images, not over the ranks."""
    world, counts = 8, (24, 21, 5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, counts)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for ci, count in enumerate(counts):
        shards = [res[r][ci][0] for r in range(world)]
        assert shards == [list(range(r, count, world)) for r in range(world)]  # image i -> rank i mod 8
        want_sizes = {24: [3] * 8, 21: [3, 3, 3, 3, 3, 2, 2, 2], 5: [1, 1, 1, 1, 1, 0, 0, 0]}[count]
        assert [len(s) for s in shards] == want_sizes
        reduced = [res[r][ci][1] for r in range(world)]
        assert all(m == reduced[0] for m in reduced)  # every rank, the idle ones too, holds the same figures
        m = reduced[0]
        assert m["images"] == count
        assert abs(m["avg_psnr"] - (20.0 + (count - 1) / 2)) < 1e-9
        assert abs(m["sum_train_s"] - sum(0.25 * (i + 1) for i in range(count))) < 1e-9
        assert abs(m["avg_num_gaussians"] - (1000 + (count - 1) / 2)) < 1e-9
        assert abs(m["avg_eval_s"] - 0.002) < 1e-12


def test_launcher_reads_the_kodak_pixel_fixture(golden_dir):
    """`--dataset tests/golden/kodak24.npz`: the Kodak pictures without the reference checkout (18 landscape, 6
    portrait, float [H, W, 3] in [0, 1] as utils.py:21-26 image_path_to_tensor hands them to the trainer)."""
    from gaussianimage_plus_amd.launch import load_images
    imgs = load_images(os.path.join(golden_dir, "kodak24.npz"), 24, 0, 0)
    assert len(imgs) == 24
    shapes = [tuple(im.shape) for im in imgs]
    assert shapes.count((512, 768, 3)) == 18 and shapes.count((768, 512, 3)) == 6
    assert all(im.dtype == torch.float32 and 0.0 <= float(im.min()) and float(im.max()) <= 1.0 for im in imgs)
    assert len(load_images(os.path.join(golden_dir, "kodak24.npz"), 3, 0, 0)) == 3
