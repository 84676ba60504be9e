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


def test_rank_cpu_cores_are_disjoint_and_cover_the_node():
    """every rank of an N-GPU run pins itself to its own host cores (near its GPU where the box says which)."""
    avail = set(range(256))
    flat = [bench.rank_cpu_cores(r, 8, avail) for r in range(8)]
    assert all(len(c) == 32 for c in flat) and len(set().union(*flat)) == 256
    assert all(flat[i].isdisjoint(flat[j]) for i in range(8) for j in range(i))
    # two NUMA nodes, four GPUs each: a rank's cores come from ITS GPU's node, split four ways
    near = [frozenset(range(0, 128))] * 4 + [frozenset(range(128, 256))] * 4
    numa = [bench.rank_cpu_cores(r, 8, avail, near) for r in range(8)]
    assert all(len(c) == 32 for c in numa) and all(numa[i].isdisjoint(numa[j]) for i in range(8) for j in range(i))
    assert numa[1] <= set(range(0, 128)) and numa[5] <= set(range(128, 256)) and numa[4] == set(range(128, 160))
    # a restricted affinity mask (container with 8 cores), more ranks than cores, no topology for some GPUs, one rank: never empty
    assert bench.rank_cpu_cores(3, 8, set(range(8))) == {3}
    assert bench.rank_cpu_cores(3, 16, set(range(8))) == set(range(8))
    assert bench.rank_cpu_cores(1, 2, set(range(8)), [None, None]) == {4, 5, 6, 7}
    assert bench.rank_cpu_cores(0, 1, {2, 3}) == {2, 3}
    assert bench.rank_cpu_cores(1, 2, set(range(8)), [frozenset({100}), frozenset({100})]) == {4, 5, 6, 7}   # "near" cores not available
    assert bench._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    info = bench.gpu_identity(0)                                        # no GPU here: every field present, nothing raises
    assert {"pci_bus_id", "numa_node", "local_cpulist", "xgmi_hive_id", "hip_visible_devices"} <= set(info)


@pytest.mark.gpu
def test_bench_gpus_2_model_on():
    """`--gpus N --model on` staggers MIOpen's kernel search (rank 0 first) -- that warm-up must issue no
    collective, or rank 0's gradient all-reduce meets the other ranks' barrier.  Two ranks (RCCL on a 2-GPU box, else the shared-GPU
    gloo dry run, where a mismatched collective is a hard error rather than a hang): the whole FlowNet2C pass must complete."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    share = torch.cuda.device_count() < 2
    if share:
        env["FN2_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--model", "on", "--model-steps", "2", "--model-warmup", "1", "--model-timeout", "900"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    m = line["flownet2c"]
    assert "error" not in m, m
    assert m["train_step_ms"] > 0 and m["fwd_bwd_ms"] > 0 and m["inference_ms"] > 0 and m["grad_buckets"] >= 2
    assert line["launch"].startswith("hipGraph")                        # N > 1 replays a graph of the step by default
    assert all("pci_bus_id" in p and "cpu_cores" in p for p in line["per_rank"])


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
        # round 6: where the rank ran -- the GPU's slot and the host cores it pinned itself to
        assert {"pci_bus_id", "numa_node", "xgmi_hive_id", "hip_visible_devices", "cpu_cores", "pid"} <= set(p)
    assert len({p["cpu_cores"] for p in line["per_rank"]}) == 8                 # eight disjoint core sets
    assert line["launch"].startswith("hipGraph") and "placement_autotune" in line
    slowest = max(p["ms_per_step"] for p in line["per_rank"])
    assert abs(line["ms_per_step"] - slowest) <= 0.25 * slowest + 0.05       # the step time is the slowest rank's (barriers aside)
    assert 0 < line["rank_balance_fastest_over_slowest"] <= 1.0
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_gpus8_dry_run.json"), "w") as f:
        f.write(lines[0] + "\n")
