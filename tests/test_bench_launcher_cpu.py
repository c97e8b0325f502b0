"""CPU: the process-launching half of bench.py (`--gpus N` without a launcher around it).  No GPU here, so the ranks
themselves stop at "no GPU, no number" -- what is checked is that N of them are started with the right environment,
that their failure is the parent's exit code, and that a mislabelled run is refused."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(kw)
    return env


def test_world_size_must_match_gpus():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=120,
                         env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and not out.stdout.strip()


def test_parent_starts_n_ranks_and_reports_their_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("with a GPU the ranks run for real: tests/test_bench_contract_gpu.py")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True,
                         text=True, timeout=300, env=_env(GI2D_BENCH_BACKEND="gloo"))
    assert out.returncode != 0
    assert out.stderr.count("no GPU, no number") == 2, out.stderr[-1500:]  # both ranks got as far as the device check
    assert "rank exit codes" in out.stderr and not out.stdout.strip()


def test_launch_ranks_sets_the_rendezvous_environment(tmp_path, monkeypatch):
    """launch_ranks() with a stand-in for the interpreter: every rank must see its own RANK / LOCAL_RANK, the common
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT, and the original argument list."""
    sys.path.insert(0, ROOT)
    import bench
    probe = tmp_path / "probe.sh"
    probe.write_text("#!/bin/sh\necho \"$RANK $LOCAL_RANK $WORLD_SIZE $MASTER_ADDR $MASTER_PORT $HSA_ENABLE_IPC_MODE_LEGACY $*\" "
                     f">> {tmp_path}/seen.txt\n[ \"$RANK\" = 0 ] && echo '{{\"n_gpus\": 3}}'\nexit 0\n")
    probe.chmod(0o755)
    monkeypatch.setattr(bench.sys, "executable", str(probe))
    monkeypatch.delenv("MASTER_PORT", raising=False)
    rc = bench.launch_ranks(3, ["--gpus", "3", "--steps", "5"])
    assert rc == 0
    rows = sorted(l.split() for l in (tmp_path / "seen.txt").read_text().splitlines())
    assert [r[0] for r in rows] == ["0", "1", "2"] and [r[1] for r in rows] == ["0", "1", "2"]
    assert all(r[2] == "3" and r[3] == "127.0.0.1" and r[5] == "0" for r in rows)
    assert len({r[4] for r in rows}) == 1 and int(rows[0][4]) > 0
    assert all(r[-4:] == ["--gpus", "3", "--steps", "5"] for r in rows)


def test_stored_counters_exist_for_the_kernels_bench_names():
    """bench.py copies `roofline.traffic` (and the VALU counters) from profiles/traffic.json by KERNEL NAME; a kernel that
    gains a template argument silently turns the figure into null.  The names the default run, BASELINE config 4 and the
    24-image block use must have an entry with HBM bytes -- regenerate the profiles (tools/profile_rounds.sh,
    tools/make_profiles_rounds.py) when this fails."""
    sys.path.insert(0, ROOT)
    import bench
    cases = [("gi2d::fast_fwdbwd_kernel<1, 0, true>", 50000, 512, 768, None),
             ("gi2d::fast_fwdbwd_kernel<1, 1, false> + gi2d::fast_fwdbwd_kernel<1, 2, false>", 50000, 1356, 2040, None),
             ("gi2d::fast_fwdbwd_batched_kernel<1, 1> + gi2d::fast_fwdbwd_batched_kernel<1, 2>", 50000, 512, 768, 24)]
    for kernel, n, h, w, k in cases:
        traffic, source = bench.pmc_traffic(kernel, n, h, w, k)
        assert traffic and traffic > 1e6, (kernel, source)
        assert source.startswith("stored:"), source
    valu = bench.valu_roofline(cases[0][0], 50000, 512, 768, 18.0)
    assert valu.get("insts_per_image", 0) > 1e6, valu
