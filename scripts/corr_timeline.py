"""Per-chunk s_memtime timeline of the instrumented DMA forward kernel (profiling only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch, fn2_capi
algo = int(sys.argv[1]) if len(sys.argv) > 1 else 1008
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
in1 = torch.randn(8, 256, 48, 64, generator=g).to(dev); in2 = torch.randn(8, 256, 48, 64, generator=g).to(dev)
out = torch.empty(8, 441, 48, 64, device=dev)
dbg = torch.zeros(2048, dtype=torch.int64, device=dev)
lib = fn2_capi.lib()
lib.fn2_debug_set_buffer.restype = None
for _ in range(3):
    fn2_capi.correlation_forward(in1, in2, 20, 1, 20, 1, 2, algo=algo, out=out)
lib.fn2_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
fn2_capi.correlation_forward(in1, in2, 20, 1, 20, 1, 2, algo=algo, out=out)
torch.cuda.synchronize()
lib.fn2_debug_set_buffer(ctypes.c_void_p(0))
d = dbg.cpu().numpy()
for blk in (0, 1):
    t = d[blk * 1024: blk * 1024 + 32 * 8].reshape(32, 8)
    if t[0, 0] == 0: print("block", blk, "no stamps"); continue
    t0 = t[0, 0]
    print("block", "0" if blk == 0 else "300", ": chunk  start  wait  barrier  dma_issue  mma   (cycles, s_memtime ticks)")
    for c in range(32):
        if t[c, 0] == 0: break
        r = t[c]
        print("  %2d  %7d  %5d  %5d  %5d  %5d   total %5d" % (c, r[0] - t0, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], (t[c + 1, 0] if c + 1 < 32 and t[c + 1, 0] else r[4]) - r[0]))
