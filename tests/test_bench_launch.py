"""bench.py's own multi-GPU launch (missing #1): `python bench.py --gpus N` must work without torchrun.
CPU: the launch plan (environment of every rank, refusal when GPUs are missing).  GPU: the spawn path end to end at
N = 2 -- over RCCL when the box has two GPUs, as the labelled shared-GPU dry run (gloo) on a 1-GPU box."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_plan():
    assert bench.launch_plan(1, 0, False) == ([], None)                 # N = 1 runs in-process
    plan, err = bench.launch_plan(4, 8, False)
    assert err is None and [e["RANK"] for e in plan] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["LOCAL_RANK"] == e["RANK"] for e in plan)
    plan, err = bench.launch_plan(2, 1, False)
    assert plan is None and "only 1 GPU" in err                         # refuses instead of oversubscribing silently
    plan, err = bench.launch_plan(2, 1, True)                           # FN2_BENCH_SHARE_GPU=1: labelled dry run
    assert err is None and len(plan) == 2
    assert bench.launch_plan(0, 8, False)[0] is None


def test_self_launch_reports_missing_gpus_without_hanging():
    """No GPU here: --gpus 2 must fail fast with a message (either the GPU assert or the launch plan), never hang."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and ("GPU" in r.stderr)


@pytest.mark.gpu
def test_bench_gpus_2_self_launch():
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    share = torch.cuda.device_count() < 2
    if share:
        env["FN2_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--model", "off"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                            # ONE JSON line, rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert len(line["per_rank"]) == 2 and all(p["finite"] for p in line["per_rank"])
    assert line.get("dry_run_shared_gpu", False) == share
    if not share:
        assert sorted(p["device"] for p in line["per_rank"]) == [0, 1]


@pytest.mark.gpu
def test_bench_gpus_8_line_explains_itself():
    """the 8-rank line before an 8-GPU node shows up -- over RCCL where the box has eight GPUs, else the labelled
    shared-GPU dry run.  Eight `per_rank` entries, each with its own step time, pairs/s, the graded kernel's warm time and a copy
    rate of ITS GPU, and the fastest-over-slowest ratio that caps the run's scaling efficiency."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    share = torch.cuda.device_count() < 8
    if share:
        env["FN2_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--model", "off"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    assert len(line["per_rank"]) == 8
    for p in line["per_rank"]:
        assert p["finite"] and p["ms_per_step"] > 0 and p["image_pairs_per_s"] > 0 and p["corr_fwd_us_warm"] > 0 and p["copy_GBps_torch"] > 0
    slowest = max(p["ms_per_step"] for p in line["per_rank"])
    assert abs(line["ms_per_step"] - slowest) <= 0.25 * slowest + 0.05       # the step time is the slowest rank's (barriers aside)
    assert 0 < line["rank_balance_fastest_over_slowest"] <= 1.0
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_gpus8_dry_run.json"), "w") as f:
        f.write(lines[0] + "\n")
