// binding_common.h -- shared helpers of the three pybind modules (host C++, no kernels).
// The modules keep the reference's module names and positional signatures
// (correlation_cuda.cc:169-172, resample2d_cuda.cc:28-31, channelnorm_cuda.cc:27-30) and
// forward to the C ABI in include/flownet2_hip.h on the caller's current HIP stream.
#pragma once
#include <torch/extension.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>

#include "flownet2_hip.h"

namespace fn2b {

inline int dtype_of(const at::Tensor &t, const char *op)
{
    switch (t.scalar_type()) {
    case at::kFloat: return FN2_F32;
    case at::kHalf: return FN2_F16;
    case at::kDouble: return FN2_F64;
    default: TORCH_CHECK(false, op, ": unsupported dtype ", t.scalar_type(), " (float, half or double expected)");
    }
    return -1;
}

// The HIP kernels are the only implementation: there is no CPU path to fall back to.
inline void check_gpu(const at::Tensor &t, const char *op, const char *name)
{
    TORCH_CHECK(t.defined(), op, ": ", name, " is undefined");
    TORCH_CHECK(t.is_cuda(), op, ": ", name, " must be a GPU (HIP) tensor; this extension has no CPU implementation");
}

inline void check_same(const at::Tensor &a, const at::Tensor &b, const char *op, const char *nb)
{
    check_gpu(b, op, nb);
    TORCH_CHECK(a.device() == b.device(), op, ": ", nb, " is on ", b.device(), ", expected ", a.device());
    TORCH_CHECK(a.scalar_type() == b.scalar_type(), op, ": ", nb, " has dtype ", b.scalar_type(), ", expected ",
                a.scalar_type());
}

inline void check_rc(int rc, const char *op)
{
    // reference: launcher returns 0 -> AT_ERROR("CUDA call failed") (correlation_cuda.cc:81-83)
    TORCH_CHECK(rc == FN2_OK, op, ": HIP call failed: ", fn2_strerror(rc), " (code ", rc, ")");
}

// PyTorch-ROCm presents HIP devices as device type "cuda"; the masquerading accessor is the one
// that matches the tensors' device type (the plain c10::hip guard/stream classes expect type "hip").
inline void *current_stream(const at::Tensor &t)
{
    return static_cast<void *>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream());
}

} // namespace fn2b
