"""Whole-network probe: FlowNet2C fwd+bwd (bs 8 @ 384x512, fp32) with MIOpen find mode / channels_last switches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
from harness.train import Trainer, synthetic_batch, time_steps
dev = torch.device("cuda:0")
for bench_mode, cl in ((False, False), (True, False), (True, True)):
    torch.backends.cudnn.benchmark = bench_mode
    tr = Trainer(dev)
    inputs, target = synthetic_batch(8, 384, 512, dev)
    if cl:
        tr.model = tr.model.to(memory_format=torch.channels_last)
    try:
        t = time_steps(lambda: tr.train_step(inputs, target), 10, 4, dev)
        ti = time_steps(lambda: tr.infer(inputs), 10, 4, dev)
        print(f"cudnn.benchmark={bench_mode} channels_last={cl}: train step {t*1e3:.2f} ms, inference {ti*1e3:.2f} ms", flush=True)
    except Exception as e:
        print("failed", bench_mode, cl, repr(e)[:200])
    del tr
