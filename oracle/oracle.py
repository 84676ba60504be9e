"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Two libraries, same numpy-in / numpy-out calling convention:

* ``Oracle()``  -> oracle/libfn2_oracle.so, the plain-C restatement of the reference's
  CUDA kernels (oracle/fn2_oracle.c; each function cites the reference file:line).
* ``Oracle(ref=True)`` -> oracle/_ref/libfn2_ref.so, the reference's own ``__global__``
  kernels compiled from /root/reference against the CPU SIMT shim (oracle/simt/).  Only
  buildable in the dev container; the built .so travels to the GPU box.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (flownet2-pytorch_amd/) never does.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libfn2_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libfn2_ref.so")

_I = ctypes.c_int
_P = ctypes.c_void_p


def build(ref=False):
    """(Re)build the oracle library (and, when /root/reference exists, oracle/_ref)."""
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    if ref:
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)


def have_ref():
    return os.path.exists(REF_SO)


def corr_shapes(H, W, pad, k, md, s1, s2):
    """Shape math of correlation_forward_cuda (reference correlation_cuda.cc:19-34)."""
    kr = (k - 1) // 2
    br = kr + md
    pH, pW = H + 2 * pad, W + 2 * pad
    D = (md // s2) * 2 + 1
    oH = int(math.ceil(np.float32(pH - 2 * br) / np.float32(s1)))
    oW = int(math.ceil(np.float32(pW - 2 * br) / np.float32(s1)))
    return pH, pW, D * D, oH, oW


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a, a.ctypes.data_as(_P)


class Oracle:
    def __init__(self, ref=False):
        path = REF_SO if ref else ORACLE_SO
        if not os.path.exists(path):
            if ref:
                raise FileNotFoundError(path + " (run `make -C oracle ref` in the dev container)")
            build()
        self.lib = ctypes.CDLL(path)
        self.p = "fn2ref_" if ref else "fn2o_"
        self.ref = ref

    def _fn(self, name, dt):
        suf = {np.float32: "f32", np.float64: "f64"}[dt]
        f = getattr(self.lib, f"{self.p}{name}_{suf}")
        f.restype = _I
        return f

    # ---------------------------------------------------------------- correlation
    def corr_fwd(self, in1, in2, pad, k, md, s1, s2):
        dt = np.float64 if in1.dtype == np.float64 else np.float32
        B, C, H, W = in1.shape
        _, _, nOut, oH, oW = corr_shapes(H, W, pad, k, md, s1, s2)
        a, pa = _c(in1, dt)
        b, pb = _c(in2, dt)
        out = np.zeros((B, nOut, oH, oW), dt)
        rc = self._fn("corr_fwd", dt)(pa, pb, out.ctypes.data_as(_P), _I(B), _I(C), _I(H), _I(W),
                                      _I(pad), _I(k), _I(md), _I(s1), _I(s2))
        assert rc == 0, rc
        return out

    def corr_bwd(self, in1, in2, gout, pad, k, md, s1, s2):
        dt = np.float64 if in1.dtype == np.float64 else np.float32
        B, C, H, W = in1.shape
        a, pa = _c(in1, dt)
        b, pb = _c(in2, dt)
        g, pg = _c(gout, dt)
        g1 = np.zeros((B, C, H, W), dt)
        g2 = np.zeros((B, C, H, W), dt)
        rc = self._fn("corr_bwd", dt)(pa, pb, pg, g1.ctypes.data_as(_P), g2.ctypes.data_as(_P),
                                      _I(B), _I(C), _I(H), _I(W), _I(pad), _I(k), _I(md), _I(s1), _I(s2))
        assert rc == 0, rc
        return g1, g2

    # ---------------------------------------------------------------- resample2d (float only)
    def resample_fwd(self, img, flow, kernel_size=1, bilinear=True):
        B, C, Hi, Wi = img.shape
        Bf, two, H, W = flow.shape
        assert two == 2 and Bf == B
        a, pa = _c(img, np.float32)
        f, pf = _c(flow, np.float32)
        out = np.zeros((B, C, H, W), np.float32)
        rc = self._fn("resample_fwd", np.float32)(pa, pf, out.ctypes.data_as(_P), _I(B), _I(C), _I(Hi), _I(Wi),
                                                  _I(H), _I(W), _I(kernel_size), _I(int(bilinear)))
        assert rc == 0, rc
        return out

    def resample_bwd(self, img, flow, gout, kernel_size=1, bilinear=True):
        B, C, Hi, Wi = img.shape
        _, _, H, W = flow.shape
        a, pa = _c(img, np.float32)
        f, pf = _c(flow, np.float32)
        g, pg = _c(gout, np.float32)
        gimg = np.zeros((B, C, Hi, Wi), np.float32)
        gflow = np.zeros((B, 2, H, W), np.float32)
        rc = self._fn("resample_bwd", np.float32)(pa, pf, pg, gimg.ctypes.data_as(_P), gflow.ctypes.data_as(_P),
                                                  _I(B), _I(C), _I(Hi), _I(Wi), _I(H), _I(W),
                                                  _I(kernel_size), _I(int(bilinear)))
        assert rc == 0, rc
        return gimg, gflow

    # ---------------------------------------------------------------- channelnorm
    def chnorm_fwd(self, x):
        dt = np.float64 if x.dtype == np.float64 else np.float32
        B, C, H, W = x.shape
        a, pa = _c(x, dt)
        out = np.zeros((B, 1, H, W), dt)
        rc = self._fn("chnorm_fwd", dt)(pa, out.ctypes.data_as(_P), _I(B), _I(C), _I(H), _I(W))
        assert rc == 0, rc
        return out

    def chnorm_bwd(self, x, out, gout):
        """gout: (B,1,H,W) values.  The reference indexes gradOutput as if contiguous
        (channelnorm_kernel.cu:92); pass the values you want element (b,y,x) to see."""
        dt = np.float64 if x.dtype == np.float64 else np.float32
        B, C, H, W = x.shape
        a, pa = _c(x, dt)
        o, po = _c(out, dt)
        g, pg = _c(gout, dt)
        gin = np.zeros((B, C, H, W), dt)
        f = self._fn("chnorm_bwd", dt)
        if self.ref:
            rc = f(pa, po, pg, gin.ctypes.data_as(_P), _I(B), _I(C), _I(H), _I(W))
        else:
            rc = f(pa, po, pg, ctypes.c_long(H * W), gin.ctypes.data_as(_P), _I(B), _I(C), _I(H), _I(W))
        assert rc == 0, rc
        return gin


def multiscale_l1_epe_sums(outputs, target, start_scale=4, div_flow=0.05):
    """numpy restatement of the per-scale sums of the reference's MultiScale loss (losses.py:74-78) and EPE (:12):
    t = div_flow * target (:74); t_i = AvgPool2d(k_i, k_i)(t) with k_i = start_scale << i (:69,:76), floor semantics;
    returns ([sum |out_i - t_i|], [sum_{b,y,x} ||t_i - out_i||_2]) in float64 from float32 inputs (test infrastructure)."""
    t = (np.float32(div_flow) * target.astype(np.float32)).astype(np.float32)
    l1, epe = [], []
    for i, o in enumerate(outputs):
        k = start_scale << i
        B, _, H, W = t.shape
        Hi, Wi = H // k, W // k
        ti = t[:, :, :Hi * k, :Wi * k].reshape(B, 2, Hi, k, Wi, k).astype(np.float64).mean(axis=(3, 5))
        d = o.astype(np.float64) - ti
        l1.append(np.abs(d).sum())
        epe.append(np.sqrt((d * d).sum(axis=1)).sum())
    return np.array(l1), np.array(epe)


def multiscale_grads(outputs, target, weights, norm=1, start_scale=4, div_flow=0.05):
    """d(sum_i w_i loss_i)/d out_i of the reference's MultiScale (losses.py:74-78) in float64 (test infrastructure):
    norm 1: w_i / N_i * sign(out_i - t_i) (L1, :17); norm 2: w_i / (N_i / 2) * (out_i - t_i) / ||out_i - t_i||_2 (L2, :25),
    0 where the norm is 0.  Also returns |out_i - t_i| so that a test can leave sign flips at rounding level aside."""
    t = (np.float32(div_flow) * target.astype(np.float32)).astype(np.float32)
    grads, absd = [], []
    for i, o in enumerate(outputs):
        k = start_scale << i
        B, _, H, W = t.shape
        Hi, Wi = H // k, W // k
        ti = t[:, :, :Hi * k, :Wi * k].reshape(B, 2, Hi, k, Wi, k).astype(np.float64).mean(axis=(3, 5))
        d = o.astype(np.float64) - ti
        if norm == 1:
            g = weights[i] / max(d.size, 1) * np.sign(d)
        else:
            r = np.sqrt((d * d).sum(axis=1, keepdims=True))
            g = weights[i] / max(d.size // 2, 1) * np.where(r > 0, d / np.where(r > 0, r, 1.0), 0.0)
        grads.append(g)
        absd.append(np.abs(d))
    return grads, absd
