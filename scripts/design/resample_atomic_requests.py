"""Cost model of the Resample2d backward's global atomics (scripts/ubench/atomic_rate.hip): fp32 atomics cost per REQUEST
= per aligned 64-byte segment one instruction touches, 20.5 G requests/s chip-wide.  For the bench's flow (image 0 x 8) and a tile shape
(TH, TW, R): pixels whose corners leave the window (12 single-lane atomics each today, half with the two corners of a row in one request)
and the 16-float segments of the window rows that receive a non-zero contribution (x 3 channels).  tests/test_tiling_model.py ties
the 32 x 64 +- 16 row to the numbers the document quotes."""
import numpy as np, torch
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 3, 384, 512
img = (torch.rand(B, C, H, W, generator=g) - 0.5)
flow = torch.randn(B, 2, H, W, generator=g) * 4.0
idx = torch.randint(0, flow.numel(), (flow.numel() // 100,), generator=g)
flow.view(-1)[idx] *= 20.0
flow = flow.numpy()
def count(TH, TW, R, b=0):
    fx, fy = flow[b,0], flow[b,1]
    ys, xs = np.mgrid[0:H, 0:W]
    xf = xs + fx; yf = ys + fy
    xL = np.clip(np.floor(xf), 0, W-1).astype(int); xR = np.clip(np.floor(xf)+1, 0, W-1).astype(int)
    yT = np.clip(np.floor(yf), 0, H-1).astype(int); yB = np.clip(np.floor(yf)+1, 0, H-1).astype(int)
    ty, tx = ys // TH, xs // TW
    wx0 = tx*TW - R; wy0 = ty*TH - R
    inw = (xL - wx0 >= 0) & (xR - wx0 < TW+2*R) & (yT - wy0 >= 0) & (yB - wy0 < TH+2*R)
    nout = (~inw).sum()
    # segments touched: (tile, row, seg) for in-window pixels, corners (yT,xL),(yT,xR),(yB,xL),(yB,xR)
    tile = (ty * (W//TW) + tx)[inw]
    segs = set()
    keys = []
    for yy, xx in ((yT,xL),(yT,xR),(yB,xL),(yB,xR)):
        k = (tile.astype(np.int64) * H + yy[inw]) * (W//16) + xx[inw]//16
        keys.append(k)
    nseg = np.unique(np.concatenate(keys)).size
    # cells
    keysc = []
    for yy, xx in ((yT,xL),(yT,xR),(yB,xL),(yB,xR)):
        keysc.append((tile.astype(np.int64) * H + yy[inw]) * W + xx[inw])
    ncell = np.unique(np.concatenate(keysc)).size
    # far pixels: distinct (row, seg) pairs per pixel: rows 1-2, x pair in same seg or not
    far_req = 0
    o = ~inw
    rows = 1 + (yB[o] != yT[o]); cols = 1 + ((xR[o]//16) != (xL[o]//16)); cols_now = 1 + (xR[o] != xL[o])
    return dict(far_px=int(nout), far_req_now=int((rows*cols_now).sum())*3, far_req_merged=int((rows*cols).sum())*3, flush_cells=int(ncell)*3, flush_req=int(nseg)*3)
for shape in ((32,64,16),(48,64,16),(64,64,16),(32,128,16),(64,128,16),(32,64,12),(32,64,24),(64,64,24),(64,128,24), (64,128,32)):
    r = count(*shape)
    tot_now = (r['far_req_now'] + r['flush_req'])*8; tot_m = (r['far_req_merged'] + r['flush_req'])*8
    print(shape, {k: v*8 for k, v in r.items()}, "total req now %.2f M = %.1f us; merged %.2f M = %.1f us" % (tot_now/1e6, tot_now/20.5e3, tot_m/1e6, tot_m/20.5e3))
