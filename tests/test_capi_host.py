"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/flownet2_hip.h declares, rejects bad calls before touching the GPU, and the pybind
modules / Python wrappers keep the reference's names and fail loudly without a GPU."""
import ctypes
import inspect
import os
import re

import pytest
import torch

from conftest import ROOT

import fn2_capi


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "flownet2_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fn2_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported():
    lib = fn2_capi.lib()
    names = declared_symbols()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/flownet2_hip.h but not exported"
    assert sorted(fn2_capi.EXPORTS) == names
    assert lib.fn2_abi_version() == 3


def test_product_library_exports_only_the_public_abi():
    """The profiling entry points (csrc/fn2_debug.h) -- among them the one that sets a process-global timeline buffer -- live in
    libflownet2_hip_debug.so; the product library exports exactly the fn2_* symbols include/flownet2_hip.h declares
    ('no global mutable state')."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", fn2_capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3})
    # literally: no mangled fn2::... helpers, no kernel handles (-fvisibility=hidden + csrc/exports.map)
    assert exported == declared_symbols(), set(exported) ^ set(declared_symbols())
    dbg = fn2_capi.debug_lib()
    for n in fn2_capi.DEBUG_EXPORTS + fn2_capi.EXPORTS:
        assert hasattr(dbg, n), n


def test_output_shape_matches_reference_formula():
    # correlation_cuda.cc:19-34
    assert fn2_capi.correlation_output_shape(48, 64, 20, 1, 20, 1, 2) == (441, 48, 64)
    assert fn2_capi.correlation_output_shape(48, 64, 3, 3, 20, 1, 2) == (441, 12, 28)
    assert fn2_capi.correlation_output_shape(16, 16, 4, 1, 4, 2, 2) == (25, 8, 8)
    with pytest.raises(RuntimeError):
        fn2_capi.correlation_output_shape(4, 4, 0, 1, 20, 1, 2)   # empty output
    from oracle.oracle import corr_shapes
    for args in [(48, 64, 20, 1, 20, 1, 2), (10, 12, 2, 1, 4, 1, 2), (9, 9, 3, 3, 2, 2, 1), (17, 5, 4, 1, 4, 3, 2)]:
        assert fn2_capi.correlation_output_shape(*args) == corr_shapes(*args)[2:]


def test_rejected_calls_return_codes_without_gpu():
    lib = fn2_capi.lib()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.fn2_channelnorm_forward(p, p, 7, 1, 1, 2, 2, null) == -2            # FN2_EDTYPE
    assert lib.fn2_channelnorm_forward(p, p, 0, 1, 0, 2, 2, null) == -1            # C < 1
    assert lib.fn2_channelnorm_forward(null, p, 0, 1, 1, 2, 2, null) == -1         # null pointer
    assert lib.fn2_channelnorm_forward(p, p, 0, 0, 3, 2, 2, null) == 0             # empty batch: nothing to do
    assert lib.fn2_resample2d_forward(p, null, p, p, 1, 1, 2, 2, 2, 2, 0, 1, null) == -1   # kernel_size < 1: FN2_EINVAL
    assert lib.fn2_correlation_forward(p, p, p, 0, 1, 4, 4, 4, 0, 1, 20, 1, 2, null) == -1  # empty output
    assert lib.fn2_correlation_backward(p, p, p, p, p, 0, 1, 4, 8, 8, 4, 1, 4, 2, 2, null) == -4  # stride1 != 1
    assert lib.fn2_correlation_forward_ex(p, p, p, 0, 1, 4, 8, 8, 4, 3, 4, 1, 2, 2, null) == -4   # MFMA path needs k=1
    i64, f32 = ctypes.c_int64, ctypes.c_float
    assert lib.fn2_correlation_forward_fused(p, p, p, i64(10), f32(0.1), 0, 1, 4, 8, 8, 4, 1, 4, 1, 2, 0, null) == -1  # stride < 25*8*8
    assert lib.fn2_correlation_forward_fused(p, p, p, i64(3200), f32(0.1), 0, 0, 4, 8, 8, 4, 1, 4, 1, 2, 0, null) == 0   # empty batch
    mis = ctypes.c_void_p(ctypes.addressof(buf) + 2)
    assert lib.fn2_channelnorm_forward(mis, p, 0, 1, 1, 2, 2, null) == -3          # FN2_EALIGN
    assert b"dtype" in lib.fn2_strerror(-2) and lib.fn2_strerror(0) == b"ok"
    # ABI v2 entry points (round 5): fn2_warp_diff_norm_cat_backward, fn2_multiscale_loss
    wb = lib.fn2_warp_diff_norm_cat_backward
    assert wb(p, p, p, p, null, p, f32(20.0), 1, 0, 8, 8, 1, null) == -1           # C < 1
    assert wb(p, p, p, p, null, p, f32(0.0), 1, 3, 8, 8, 1, null) == -1            # div_flow == 0
    assert wb(p, p, p, p, null, p, f32(float("nan")), 1, 3, 8, 8, 1, null) == -1   # div_flow nan
    assert wb(p, p, p, null, null, p, f32(20.0), 1, 3, 8, 8, 1, null) == -1        # no concat gradient
    assert wb(p, p, p, p, null, null, f32(20.0), 1, 3, 8, 8, 1, null) == -1        # no flow gradient to write
    assert wb(p, p, p, p, null, p, f32(20.0), 0, 3, 8, 8, 1, null) == 0            # empty batch
    assert wb(mis, p, p, p, null, p, f32(20.0), 1, 3, 8, 8, 1, null) == -3         # FN2_EALIGN
    assert lib.fn2_warp_diff_norm(p, p, null, 1, 3, 8, 8, 1, null) == -1                    # no output
    assert lib.fn2_warp_diff_norm(p, p, p, 1, 0, 8, 8, 1, null) == -1                       # C < 1
    assert lib.fn2_warp_diff_norm(p, p, p, 0, 3, 8, 8, 1, null) == 0                        # empty batch
    assert lib.fn2_warp_diff_norm_backward(p, p, p, null, p, 1, 3, 8, 8, 1, null) == -1     # no gradient of the norm
    assert lib.fn2_warp_diff_norm_backward(p, p, mis, p, p, 1, 3, 8, 8, 1, null) == -3      # FN2_EALIGN
    outs = (ctypes.c_void_p * 1)(p.value)
    w1 = (ctypes.c_float * 1)(0.32)
    ml = lib.fn2_multiscale_loss
    assert ml(outs, p, p, None, w1, f32(1.0), 3, 1, 8, 8, 4, 1, f32(0.05), p, ctypes.c_size_t(1 << 20), null) == -1    # norm must be 1 or 2
    assert ml(outs, p, p, None, w1, f32(1.0), 2, 1, 8, 8, 3, 1, f32(0.05), p, ctypes.c_size_t(1 << 20), null) == -1    # start_scale not a power of two
    assert ml(outs, p, p, None, w1, f32(1.0), 2, 1, 8, 8, 4, 1, f32(0.05), p, ctypes.c_size_t(0), null) == -1          # workspace too small


def test_modules_keep_reference_names_and_signatures():
    import channelnorm_cuda
    import correlation_cuda
    import resample2d_cuda
    for m in (correlation_cuda, resample2d_cuda, channelnorm_cuda):
        assert callable(m.forward) and callable(m.backward)
        assert ".so" in m.__file__ and os.path.dirname(m.__file__).endswith("flownet2-pytorch_amd")
    from networks.channelnorm_package.channelnorm import ChannelNorm, ChannelNormFunction
    from networks.correlation_package.correlation import Correlation, CorrelationFunction
    from networks.resample2d_package.resample2d import Resample2d, Resample2dFunction
    assert list(inspect.signature(Correlation.__init__).parameters)[1:] == [
        "pad_size", "kernel_size", "max_displacement", "stride1", "stride2", "corr_multiply"]
    assert [p.default for p in list(inspect.signature(Correlation.__init__).parameters.values())[1:]] == [0, 0, 0, 1, 2, 1]
    assert [p.default for p in list(inspect.signature(CorrelationFunction.forward).parameters.values())[3:]] == [3, 3, 20, 1, 2, 1]
    assert [p.default for p in list(inspect.signature(Resample2d.__init__).parameters.values())[1:]] == [1, True]
    assert [p.default for p in list(inspect.signature(ChannelNorm.__init__).parameters.values())[1:]] == [2]
    assert issubclass(Resample2dFunction, torch.autograd.Function) and issubclass(ChannelNormFunction, torch.autograd.Function)


def test_cpu_tensors_fail_loudly():
    """There is no CPU path behind the boundary: a CPU tensor must raise, not silently compute."""
    import channelnorm_cuda
    import correlation_cuda
    import resample2d_cuda
    a = torch.zeros(1, 16, 8, 8)
    e = torch.zeros(0)
    with pytest.raises(RuntimeError, match="GPU"):
        correlation_cuda.forward(a, a, e, e, e, 4, 1, 4, 1, 2, 1)
    with pytest.raises(RuntimeError, match="GPU"):
        correlation_cuda.backward(a, a, e, e, a, e, e, 4, 1, 4, 1, 2, 1)
    with pytest.raises(RuntimeError, match="GPU"):
        resample2d_cuda.forward(a, torch.zeros(1, 2, 8, 8), a.clone(), 1, True)
    with pytest.raises(RuntimeError, match="GPU"):
        channelnorm_cuda.forward(a, torch.zeros(1, 1, 8, 8), 2)
    with pytest.raises(TypeError):
        correlation_cuda.forward(a, a)   # positional signature of the reference binding


def test_product_path_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under flownet2-pytorch_amd/ may reference it."""
    pkg = os.path.join(ROOT, "flownet2-pytorch_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                for needle in ("import oracle", "from oracle", "libfn2_oracle", "libfn2_ref", "fn2o_", "fn2ref_",
                               "oracle/", "oracle.py"):
                    assert needle not in txt, (os.path.join(d, f), needle)
