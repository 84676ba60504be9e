"""ctypes view of the C ABI (include/flownet2_hip.h) for callers that hold raw device pointers --
used by the parity tests and bench.py to reach entry points the pybind modules do not expose
(e.g. the ``*_ex`` algorithm selectors).  Raises if libflownet2_hip.so has not been built: there
is no fallback implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libflownet2_hip.so")
DEBUG_LIB_PATH = os.path.join(_HERE, "lib", "libflownet2_hip_debug.so")   # same kernels + fn2_debug_* (csrc/fn2_debug.h)

FN2_F32, FN2_F16, FN2_F64 = 0, 1, 2
FN2_CORR_AUTO, FN2_CORR_DIRECT, FN2_CORR_MFMA_F32, FN2_CORR_MFMA_BF16X3, FN2_CORR_MFMA_F16X2 = 0, 1, 2, 3, 4

EXPORTS = [
    "fn2_strerror", "fn2_abi_version", "fn2_correlation_output_shape",
    "fn2_correlation_forward", "fn2_correlation_forward_ex", "fn2_correlation_forward_fused",
    "fn2_correlation_backward", "fn2_correlation_backward_ex",
    "fn2_correlation_backward_fused_workspace_bytes", "fn2_correlation_backward_fused",
    "fn2_resample2d_forward", "fn2_resample2d_backward", "fn2_warp_diff_norm_cat",
    "fn2_warp_diff_norm_cat_backward", "fn2_warp_diff_norm", "fn2_warp_diff_norm_backward",
    "fn2_channelnorm_forward", "fn2_channelnorm_backward",
    "fn2_multiscale_workspace_bytes", "fn2_multiscale_l1_epe", "fn2_multiscale_loss",
    "fn2_multiscale_loss_fused", "fn2_multiscale_scale_grads",
]

# profiling / ablation entry points (csrc/fn2_debug.h): not in include/flownet2_hip.h, results wrong by design; exported by
# libflownet2_hip_debug.so only
DEBUG_EXPORTS = ["fn2_debug_correlation_forward", "fn2_debug_correlation_backward", "fn2_debug_set_buffer",
                 "fn2_debug_resample2d_forward", "fn2_debug_resample2d_backward", "fn2_debug_stream_copy", "fn2_debug_mfma_probe",
                 "fn2_debug_xcc_census"]

_lib = None
_dbg = None


def lib():
    """Loads libflownet2_hip.so.  ``import torch`` first so that the HIP runtime torch already
    mapped (same soname) is the one the kernels register with."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run `python flownet2-pytorch_amd/build.py` "
                               "(the HIP kernels are the only implementation)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.fn2_strerror.restype = ctypes.c_char_p
        _lib.fn2_strerror.argtypes = [ctypes.c_int]
        for name in EXPORTS[1:]:
            getattr(_lib, name).restype = ctypes.c_int
        _lib.fn2_multiscale_workspace_bytes.restype = ctypes.c_size_t
        _lib.fn2_correlation_backward_fused_workspace_bytes.restype = ctypes.c_size_t
    return _lib


def debug_lib():
    """libflownet2_hip_debug.so: the product kernels plus the profiling instantiations (scripts/, ablations)."""
    global _dbg
    if _dbg is None:
        import torch  # noqa: F401
        if not os.path.exists(DEBUG_LIB_PATH):
            raise RuntimeError(f"{DEBUG_LIB_PATH} not found: run `python flownet2-pytorch_amd/build.py`")
        _dbg = ctypes.CDLL(DEBUG_LIB_PATH)
        _dbg.fn2_strerror.restype = ctypes.c_char_p
        for name in EXPORTS[1:] + DEBUG_EXPORTS:
            if hasattr(_dbg, name):      # (A/B runs load older builds that lack the newer entry points)
                getattr(_dbg, name).restype = ctypes.c_int
        _dbg.fn2_debug_set_buffer.restype = None
    return _dbg


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().fn2_strerror(rc).decode()} (code {rc})")


def _dtype_code(t):
    import torch
    return {torch.float32: FN2_F32, torch.float16: FN2_F16, torch.float64: FN2_F64}[t.dtype]


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def correlation_output_shape(H, W, pad, k, md, s1, s2):
    n, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(lib().fn2_correlation_output_shape(H, W, pad, k, md, s1, s2, ctypes.byref(n), ctypes.byref(oh),
                                             ctypes.byref(ow)), "fn2_correlation_output_shape")
    return n.value, oh.value, ow.value


def correlation_forward(in1, in2, pad, k, md, s1, s2, algo=FN2_CORR_AUTO, out=None):
    import torch
    B, C, H, W = in1.shape
    nOut, oH, oW = correlation_output_shape(H, W, pad, k, md, s1, s2)
    if out is None:
        out = torch.empty((B, nOut, oH, oW), dtype=in1.dtype, device=in1.device)
    # algo >= 100: profiling instantiations, only reachable through the debug entry point
    fn, what = ((debug_lib().fn2_debug_correlation_forward, "fn2_debug_correlation_forward") if algo >= 100 else
                (lib().fn2_correlation_forward_ex, "fn2_correlation_forward_ex"))
    with torch.cuda.device_of(in1):
        check(fn(_p(in1), _p(in2), _p(out), _dtype_code(in1), B, C, H, W, pad, k, md, s1, s2, algo, _stream(in1)), what)
    return out


def correlation_forward_fused(in1, in2, buffer, channel_offset, negative_slope, pad, k, md, s1, s2, algo=FN2_CORR_AUTO):
    """LeakyReLU(correlation) written into channels [channel_offset, +nOut) of the contiguous N x Ctot x oH x oW `buffer`."""
    import torch
    B, C, H, W = in1.shape
    nOut, oH, oW = correlation_output_shape(H, W, pad, k, md, s1, s2)
    assert buffer.is_contiguous() and buffer.shape[0] == B and tuple(buffer.shape[2:]) == (oH, oW)
    assert 0 <= channel_offset and channel_offset + nOut <= buffer.shape[1]
    dst = ctypes.c_void_p(buffer.data_ptr() + channel_offset * oH * oW * buffer.element_size())
    with torch.cuda.device_of(in1):
        check(lib().fn2_correlation_forward_fused(_p(in1), _p(in2), dst, ctypes.c_int64(buffer.shape[1] * oH * oW),
                                                  ctypes.c_float(negative_slope), _dtype_code(in1), B, C, H, W, pad, k,
                                                  md, s1, s2, algo, _stream(in1)), "fn2_correlation_forward_fused")
    return buffer


def correlation_backward(in1, in2, gout, pad, k, md, s1, s2, algo=FN2_CORR_AUTO, out=None):
    import torch
    B, C, H, W = in1.shape
    g1, g2 = out if out is not None else (torch.empty_like(in1), torch.empty_like(in2))
    fn, what = ((debug_lib().fn2_debug_correlation_backward, "fn2_debug_correlation_backward") if algo >= 100 else
                (lib().fn2_correlation_backward_ex, "fn2_correlation_backward_ex"))
    with torch.cuda.device_of(in1):
        check(fn(_p(in1), _p(in2), _p(gout), _p(g1), _p(g2), _dtype_code(in1), B, C, H, W, pad, k, md, s1, s2, algo,
                 _stream(in1)), what)
    return g1, g2


def correlation_backward_fused(in1, in2, buffer, grad_buffer, channel_offset, negative_slope, pad, k, md, s1, s2, algo=FN2_CORR_AUTO):
    """Gradients of the correlation branch of cat((redir, leaky_relu(corr(in1, in2)))): `buffer` is the forward's concat buffer
    (correlation_forward_fused), `grad_buffer` the gradient wrt it, both contiguous N x Ctot x oH x oW."""
    import torch
    B, C, H, W = in1.shape
    nOut, oH, oW = correlation_output_shape(H, W, pad, k, md, s1, s2)
    assert buffer.is_contiguous() and grad_buffer.is_contiguous() and buffer.shape == grad_buffer.shape
    es = buffer.element_size()
    off = channel_offset * oH * oW * es
    wsb = lib().fn2_correlation_backward_fused_workspace_bytes(_dtype_code(in1), B, H, W, pad, k, md, s1, s2)
    ws = torch.empty(max(wsb // es, 1), dtype=in1.dtype, device=in1.device)
    g1, g2 = torch.empty_like(in1), torch.empty_like(in2)
    bs = ctypes.c_int64(buffer.shape[1] * oH * oW)
    with torch.cuda.device_of(in1):
        check(lib().fn2_correlation_backward_fused(_p(in1), _p(in2), ctypes.c_void_p(buffer.data_ptr() + off), bs,
                                                   ctypes.c_void_p(grad_buffer.data_ptr() + off), bs, ctypes.c_float(negative_slope),
                                                   _p(ws), ctypes.c_size_t(wsb), _p(g1), _p(g2), _dtype_code(in1), B, C, H, W,
                                                   pad, k, md, s1, s2, algo, _stream(in1)), "fn2_correlation_backward_fused")
    return g1, g2


def warp_diff_norm_cat(pair, flow, div_flow=20.0, bilinear=True):
    """cat(pair, warp(pair[:, C:], flow), flow / div_flow, ||pair[:, :C] - warped||_2) (models.py:133-138) in one pass."""
    import torch
    B, C2, H, W = pair.shape
    C = C2 // 2
    assert pair.is_contiguous() and flow.is_contiguous() and pair.dtype == torch.float32 and C2 == 2 * C
    out = torch.empty((B, 3 * C + 3, H, W), dtype=pair.dtype, device=pair.device)
    with torch.cuda.device_of(pair):
        check(lib().fn2_warp_diff_norm_cat(_p(pair), _p(flow), _p(out), ctypes.c_float(div_flow), B, C, H, W,
                                           1 if bilinear else 0, _stream(pair)), "fn2_warp_diff_norm_cat")
    return out


def warp_diff_norm_cat_backward(pair, flow, out_cat, grad_cat, div_flow=20.0, bilinear=True, want_grad_pair=True):
    """fn2_warp_diff_norm_cat_backward through ctypes: returns (grad_pair or None, grad_flow)."""
    import torch
    B, C2, H, W = pair.shape
    C = C2 // 2
    assert pair.is_contiguous() and flow.is_contiguous() and out_cat.is_contiguous() and grad_cat.is_contiguous()
    gpair = torch.full_like(pair, float("nan")) if want_grad_pair else None
    gflow = torch.full_like(flow, float("nan"))
    with torch.cuda.device_of(pair):
        check(lib().fn2_warp_diff_norm_cat_backward(_p(pair), _p(flow), _p(out_cat), _p(grad_cat), _p(gpair) if want_grad_pair else None,
                                                    _p(gflow), ctypes.c_float(div_flow), B, C, H, W, 1 if bilinear else 0,
                                                    _stream(pair)), "fn2_warp_diff_norm_cat_backward")
    return gpair, gflow


def warp_diff_norm(pair, flow, bilinear=True):
    """||pair[:, :C] - warp(pair[:, C:], flow)||_2 (models.py:157-161): B x 1 x H x W."""
    import torch
    B, C2, H, W = pair.shape
    assert pair.is_contiguous() and flow.is_contiguous() and pair.dtype == torch.float32
    out = torch.full((B, 1, H, W), float("nan"), dtype=pair.dtype, device=pair.device)
    with torch.cuda.device_of(pair):
        check(lib().fn2_warp_diff_norm(_p(pair), _p(flow), _p(out), B, C2 // 2, H, W, 1 if bilinear else 0, _stream(pair)), "fn2_warp_diff_norm")
    return out


def warp_diff_norm_backward(pair, flow, norm, grad_norm, bilinear=True):
    import torch
    B, C2, H, W = pair.shape
    assert pair.is_contiguous() and flow.is_contiguous() and norm.is_contiguous() and grad_norm.is_contiguous()
    gflow = torch.full_like(flow, float("nan"))
    with torch.cuda.device_of(pair):
        check(lib().fn2_warp_diff_norm_backward(_p(pair), _p(flow), _p(norm), _p(grad_norm), _p(gflow), B, C2 // 2, H, W,
                                                1 if bilinear else 0, _stream(pair)), "fn2_warp_diff_norm_backward")
    return gflow


def multiscale_l1_epe(outputs, target, weights, start_scale=4, div_flow=0.05, want_grads=False, grad_scale=1.0, norm=1):
    """SURVEY.md 8f N3 (losses.py:52-86): returns (sums, grads).  sums is a device tensor of 2*n floats --
    sums[i] = sum |out_i - AvgPool_{k_i}(div_flow * target)|, sums[n+i] = sum of the per-pixel channel 2-norms; grads
    (if requested) are grad_scale * weights[i] / numel(out_i) * sign(out_i - t_i) for norm = 1 (L1) and
    grad_scale * weights[i] / (numel(out_i) / 2) * (out_i - t_i) / ||out_i - t_i||_2 for norm = 2 (L2, losses.py:64-67)."""
    import torch
    n = len(outputs)
    B, two, H, W = target.shape
    assert two == 2 and target.is_contiguous() and target.dtype == torch.float32
    for i, o in enumerate(outputs):
        k = start_scale << i
        assert o.is_contiguous() and o.dtype == torch.float32 and tuple(o.shape) == (B, 2, H // k, W // k), (i, o.shape)
    sums = torch.empty(2 * n, dtype=torch.float32, device=target.device)
    grads = [torch.empty_like(o) for o in outputs] if want_grads else None
    wsb = lib().fn2_multiscale_workspace_bytes(B, H, W, start_scale, n)
    ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=target.device)
    outs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outputs])
    gptr = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads]) if want_grads else None
    wts = (ctypes.c_float * n)(*[float(w) for w in weights])
    with torch.cuda.device_of(target):
        check(lib().fn2_multiscale_loss(outs, _p(target), _p(sums), gptr, wts, ctypes.c_float(grad_scale), int(norm), B, H, W,
                                        start_scale, n, ctypes.c_float(div_flow), _p(ws), ctypes.c_size_t(wsb),
                                        _stream(target)), "fn2_multiscale_loss")
    return sums, grads
