"""A process group of ONE rank sends every collective of the multi-GPU path through the backend: on CPU over gloo (runs
here), on the GPU box over `nccl` = RCCL -- the only way the 1-GPU boxes of this pool can execute the RCCL call sites
(dist_utils.init_from_env with device_id, barrier(device_ids=...), broadcast_state on device tensors, the MAX all-reduce of
the timing, harness/ddp.py's bucketed all-reduce on its side stream).  The reference scales with nn.DataParallel
(main.py:187-201); what replaces it is DESIGN.md 6.  Each case runs in its own interpreter: a default process group is
process-global state."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

SCRIPT = r"""
import os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import torch, torch.distributed as dist
import dist_utils
backend = %(backend)r
dev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
rank, world, local = dist_utils.init_from_env(backend=backend, force=True)
assert (rank, world, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == backend
# timing collectives of bench.py
assert dist_utils.max_over_ranks(1.25, device=dev) == 1.25
assert dist_utils.whole_job_throughput(8.0, 2.0, device=dev) == 4.0
if backend == "nccl":
    dist.barrier(device_ids=[local])
else:
    dist.barrier()
# one-time weight broadcast (flat per-dtype buffers) on the device
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 3)).to(dev)
before = [p.detach().clone() for p in model.parameters()]
dist_utils.broadcast_state(model, src=0)
assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
# bucketed gradient all-reduce, launched from the hooks on the side stream, joined by finish()
from harness.ddp import BucketedGradAllReduce
red = BucketedGradAllReduce(model, bucket_bytes=1 << 10)      # several buckets
assert red.collective and len(red.buckets) > 1
x = torch.randn(2, 3, 16, 16, device=dev)
ref = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 3)).to(dev)
ref.load_state_dict(model.state_dict())
red.zero_grad(); red.reset()
model(x).square().mean().backward()
launched = len(red._handles)
red.finish()
ref(x).square().mean().backward()
for p, q in zip(model.parameters(), ref.parameters()):
    assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7)
assert launched == len(red.buckets), (launched, len(red.buckets))
# a collective on a large flat buffer (the size class of a real bucket: 48 MB)
big = torch.ones(12 << 20, device=dev)
dist.all_reduce(big); dist.broadcast(big, src=0)
out = [torch.empty_like(big[:1024])]
dist.all_gather(out, big[:1024])
if backend == "nccl":
    torch.cuda.synchronize()
assert float(big.sum().item()) == float(12 << 20) and torch.equal(out[0], big[:1024])
maps = open("/proc/self/maps").read()
print("librccl mapped:", "librccl" in maps)
dist.destroy_process_group()
print("ONE-RANK-GROUP-OK", backend)
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(backend, timeout):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "pkg": PKG, "backend": backend}], env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "ONE-RANK-GROUP-OK " + backend in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_one_rank_group_gloo():
    _run("gloo", 300)


@pytest.mark.gpu
def test_one_rank_group_rccl():
    """RCCL executes: librccl is loaded next to the one HIP runtime and every nccl-only call site of the multi-GPU path runs."""
    out = _run("nccl", 600)
    assert "librccl mapped: True" in out, out
