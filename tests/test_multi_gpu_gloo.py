"""The N>1 path on CPU: world_size-2 gloo processes exercise the sharding / broadcast / timing
helpers bench.py uses under RCCL (the kernels themselves need no cross-GPU communication)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import dist_utils
    r, w, lr = dist_utils.init_from_env(backend="gloo")
    assert (r, w, lr) == (rank, world, rank)
    # batch sharding: 8 pairs over the ranks, contiguous, complete, disjoint
    g = torch.Generator().manual_seed(0)
    batch = torch.randn(8, 3, 4, 4, generator=g)
    flow = torch.randn(8, 2, 4, 4, generator=g)
    mine_img, mine_flow = dist_utils.shard_batch([batch, flow], w, r)
    lo, hi = dist_utils.shard_bounds(8, w, r)
    assert torch.equal(mine_img, batch[lo:hi]) and mine_flow.shape[0] == hi - lo
    gathered = [torch.zeros(1, dtype=torch.int64) for _ in range(w)]
    dist.all_gather(gathered, torch.tensor([hi - lo]))
    assert sum(int(t) for t in gathered) == 8
    # weight broadcast from rank 0 (mixed dtypes)
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    model[1].running_mean.add_(rank + 1.0)
    dist_utils.broadcast_state(model, src=0)
    sig = torch.cat([p.detach().reshape(-1).double() for p in model.parameters()] +
                    [b.reshape(-1).double() for b in model.buffers()])
    sigs = [torch.zeros_like(sig) for _ in range(w)]
    dist.all_gather(sigs, sig)
    assert all(torch.equal(sigs[0], s) for s in sigs)
    # timing: max over ranks, throughput = all items / slowest time
    assert dist_utils.max_over_ranks(1.0 + rank) == float(w)
    assert abs(dist_utils.whole_job_throughput(8.0, 1.0 + rank) - 8.0 * w / w) < 1e-12
    dist.barrier()
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]


def test_shard_bounds_cover_everything():
    import dist_utils
    for total in (0, 1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [dist_utils.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dist_utils.shard_bounds(8, 2, 2)
