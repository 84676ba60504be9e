"""FlowNet2C training / inference steps on synthetic data (reference main.py:246-340 with
MultiScale L1 and Adam lr 1e-4, README.md:81-84).  One process per GPU; ``python -m torch.distributed.run
--nproc-per-node N bench.py --model`` drives it through bench.py."""
import time

import torch

import dist_utils
from harness.ddp import BucketedGradAllReduce
from harness.flownet2c import FlowNet2C
from losses_fused import MultiScaleL1


def synthetic_batch(batch, height, width, device, seed=0, rgb_max=255.0):
    """inputs B x 3 x 2 x H x W in [0, rgb_max), target flow B x 2 x H x W ~ N(0, 5^2)."""
    g = torch.Generator().manual_seed(seed)
    inputs = rgb_max * torch.rand(batch, 3, 2, height, width, generator=g)
    target = 5.0 * torch.randn(batch, 2, height, width, generator=g)
    return inputs.to(device), target.to(device)


class Trainer:
    def __init__(self, device, lr=1e-4, seed=1, bucket_bytes=48 << 20):
        torch.manual_seed(seed)
        self.device = device
        self.model = FlowNet2C().to(device)
        dist_utils.broadcast_state(self.model, src=0)          # once, not every step (reference: DataParallel)
        self.reducer = BucketedGradAllReduce(self.model, bucket_bytes=bucket_bytes)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr)
        self.criterion = MultiScaleL1()

    def train_step(self, inputs, target):
        self.model.train()
        self.reducer.zero_grad()
        self.reducer.reset()
        loss, epe = self.criterion(self.model(inputs), target)
        loss.backward()                                         # bucket all-reduces start inside
        self.reducer.finish()
        self.opt.step()
        return loss.detach(), epe.detach()

    @torch.no_grad()
    def infer(self, inputs):
        self.model.eval()
        return self.model(inputs)


def time_steps(fn, steps, warmup, device):
    for _ in range(warmup):
        fn()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / steps
