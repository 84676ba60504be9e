import sys
sys.path.insert(0, "flownet2-pytorch_amd"); sys.path.insert(0, ".")
import numpy as np, torch, fn2_capi
dev = torch.device("cuda:0")
B, C, H, W = 1, 128, 46, 64
a = torch.ones(B, C, H, W).half().to(dev)
b = torch.ones(B, C, H, W).half().to(dev)
out = torch.full((B, 441, H, W), float("nan"), dtype=torch.float16, device=dev)
fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, out=out)
ref = fn2_capi.correlation_forward(a, b, 20, 1, 20, 1, 2, algo=fn2_capi.FN2_CORR_DIRECT)
got, r = out.float().cpu().numpy(), ref.float().cpu().numpy()
bad = np.argwhere(got != r)
print("bad", len(bad))
from collections import Counter
print("by (ti):", sorted(Counter(int(i[1]) % 21 for i in bad).items()))
print("by x:", sorted(Counter(int(i[3]) for i in bad).items()))
print("by y%8:", sorted(Counter(int(i[2]) % 8 for i in bad).items()))
print("by tj:", sorted(Counter(int(i[1]) // 21 for i in bad).items()))
print("values:", Counter(float(got[tuple(i)]) for i in bad).most_common(8))
