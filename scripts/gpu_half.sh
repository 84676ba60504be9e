#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "half or other_dtypes or fused" 2>&1 | tail -25
python - <<'PY'
import sys, os
sys.path.insert(0, "flownet2-pytorch_amd")
import torch, fn2_capi
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
a = torch.randn(8, 256, 48, 64, generator=g).to(dev); b = torch.randn(8, 256, 48, 64, generator=g).to(dev)
ah, bh = a.half(), b.half()
for name, x1, x2 in (("fp32", a, b), ("fp16", ah, bh)):
    out = torch.empty(8, 441, 48, 64, dtype=x1.dtype, device=dev)
    for _ in range(5): fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(40): fn2_capi.correlation_forward(x1, x2, 20, 1, 20, 1, 2, out=out)
    e.record(); torch.cuda.synchronize()
    print(name, "correlation forward 8x256x48x64: %.1f us back-to-back" % (s.elapsed_time(e) * 1e3 / 40))
PY
