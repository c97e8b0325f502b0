"""GPU: bench.py prints ONE JSON line with the fields the driver and the judge read (small workload, short CPU leg)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "24", "--warmup", "4",
           "--num-points", "3000", "--height", "96", "--width", "144", "--cpu-seconds", "1.5", "--images", "2",
           "--image-iterations", "200", "--images-per-gpu", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["warmup"] == 4 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["data"].startswith("synthetic") and "kodak" in d["data"]
    assert "optimizer ON" in d["config"]["workload"]  # the timed loop trains: parameters move every step
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["traffic"] is None  # the committed counters are for the default workload, not this one
    assert r["traffic_source"].startswith("none:")
    assert d["metric"].endswith("144x96")
    # five timed regions of 24 steps each for `value`, none of whose dispatches carries an event pair; between them the
    # same region five times more with a pair on every 2nd tile pass (the kernel time of `roofline`)
    assert d["repeats"] == 5 and len(d["ms_per_step_each_region"]) == 5
    assert sorted(d["ms_per_step_each_region"])[2] == d["ms_per_step"]  # the median region
    assert d["event_stride"] == 2 and d["ms_per_step_event_regions"] > 0
    # (a call's first tile pass is another instantiation of the kernel and is reported beside the dominant one)
    fp = r["first_pass_of_a_call"]
    assert r["kernel_samples"] == 55 and fp["samples"] == 5 and fp["avg_kernel_us"] > 0 and fp["kernel"].endswith("<1, 0, false>")
    assert r["kernel"].endswith("<1, 0, true>") and r["avg_kernel_us"] >= r["min_kernel_us"] > 0
    assert d["roofline_valu"]["source"].startswith("none:")  # counters are stored for the default workload only
    dr = d["dropin_autograd_step"]
    assert dr["us_per_iteration"] > 0 and dr["binding"] in ("compiled", "ctypes") and dr["num_points"] == 3000
    assert 0 < dr["hip_graph"]["us_per_iteration"] < dr["us_per_iteration"]  # one launch per iteration beats ~35
    assert dr["hip_graph"]["iterations"] == 5 * dr["iterations"]
    im = d["images_per_s"]
    assert im["unit"] == "images/s" and im["images"] == 2 and im["iterations_per_image"] == 200 and im["value"] > 0
    assert abs(im["value"] - im["images"] / im["wall_s"]) <= 1e-9 * im["value"]
    assert im["data"] == "kodak" and im["images_landscape_768x512"] == 2 and im["images_portrait_512x768"] == 0
    assert [row["image"] for row in im["rank0_images"]] == ["kodim01", "kodim02"]
    assert all(10 < row["psnr"] < 60 and row["best_model_gaussians"] > 0 for row in im["rank0_images"])
    assert im["images_concurrent_per_gpu"] == 2 and im["batches_per_gpu"] == 2  # three asked for, two images to share
    assert im["per_rank"] == [{"rank": 0, "images": 2, "batches": 2, "images_per_batch": [1, 1]}]
    st = d["static_scene_step"]
    assert st["steps_per_s"] > 0 and st["num_intersects"] > 0
    ks = d["batched"]["per_k"]
    assert [b["images_per_launch"] for b in ks] == [4, 8, 24]
    for b in ks:
        assert b["image_iters_per_s"] > 0 and b["tile_pass_kernel_us"] > 0
        assert abs(b["roofline"]["frac"] - b["roofline"]["achieved"] / 8000.0) < 1e-12
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert c["single_thread"]["cores"] == 1 and c["single_thread"]["value"] > 0
    assert c["config1_train_loop"]["value"] > 0 and "N=2500" in c["config1_train_loop"]["sample"]


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks (here both on the one GPU of
    the box, rendezvous over gloo) and relays rank 0's line, which must say n_gpus = 2 and carry the whole-job rate."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "4",
           "--num-points", "3000", "--height", "96", "--width", "144", "--images", "2", "--image-iterations", "100",
           "--images-per-gpu", "1", "--no-batched"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GI2D_BENCH_BACKEND"] = "gloo"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    assert d["images_per_s"]["images"] == 2 and "rank i mod 2" in d["images_per_s"]["workload"]
    assert [(r["rank"], r["images"], r["batches"]) for r in d["images_per_s"]["per_rank"]] == [(0, 1, 1), (1, 1, 1)]


def test_launcher_cli_runs_both_loops_and_reports_the_average_line():
    base = [sys.executable, "-m", "gaussianimage_plus_amd.launch", "--synthetic", "2", "--height", "96", "--width",
            "144", "--num_points", "1500", "--iterations", "60"]
    for extra in (["--model", "cholesky"], ["--model", "covariance", "--max_num_points", "3000", "--grow_iter", "20",
                                            "--prune_iter", "10", "--images_per_gpu", "2"]):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        rows = [l for l in out.stdout.splitlines() if l.startswith("[rank 0] image")]
        avg = [l for l in out.stdout.splitlines() if l.startswith("Average:")]
        assert len(rows) == 2 and len(avg) == 1
        assert "images:2" in avg[0] and "gpus:1" in avg[0] and "PSNR:" in avg[0]
