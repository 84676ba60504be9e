// channelnorm_cuda.cpp -- pybind module `channelnorm_cuda` (drop-in for the reference's module,
// channelnorm_cuda.cc:6-30).  norm_deg is accepted and ignored, as by the reference kernels.
#include "binding_common.h"

using namespace fn2b;

// channelnorm_cuda_forward (channelnorm_cuda.cc:6-14)
int channelnorm_forward_hip(at::Tensor &input1, at::Tensor &output, int norm_deg)
{
    (void)norm_deg;
    const char *op = "channelnorm_cuda.forward";
    check_gpu(input1, op, "input1");
    check_same(input1, output, op, "output");
    TORCH_CHECK(input1.dim() == 4 && output.dim() == 4, op, ": tensors must be 4-D");
    const int dt = dtype_of(input1, op);
    const int B = input1.size(0), C = input1.size(1), H = input1.size(2), W = input1.size(3);
    TORCH_CHECK(output.size(0) == B && output.size(1) == 1 && output.size(2) == H && output.size(3) == W, op,
                ": output has shape ", output.sizes(), ", expected [", B, ", 1, ", H, ", ", W, "]");
    TORCH_CHECK(output.is_contiguous(), op, ": output must be contiguous");
    c10::DeviceGuard guard(input1.device());
    at::Tensor a = input1.contiguous();
    check_rc(fn2_channelnorm_forward(a.data_ptr(), output.data_ptr(), dt, B, C, H, W, current_stream(input1)), op);
    return 1;
}

// channelnorm_cuda_backward (channelnorm_cuda.cc:16-25).  gradOutput's strides are honoured (the
// reference indexes it as if contiguous, channelnorm_kernel.cu:92 -- SURVEY.md 5, last bullet).
int channelnorm_backward_hip(at::Tensor &input1, at::Tensor &output, at::Tensor &gradOutput, at::Tensor &gradInput1,
                             int norm_deg)
{
    (void)norm_deg;
    const char *op = "channelnorm_cuda.backward";
    check_gpu(input1, op, "input1");
    check_same(input1, output, op, "output");
    check_same(input1, gradOutput, op, "gradOutput");
    check_same(input1, gradInput1, op, "gradInput1");
    TORCH_CHECK(input1.dim() == 4 && output.dim() == 4 && gradOutput.dim() == 4, op, ": tensors must be 4-D");
    const int dt = dtype_of(input1, op);
    const int B = input1.size(0), C = input1.size(1), H = input1.size(2), W = input1.size(3);
    TORCH_CHECK(output.size(0) == B && output.size(1) == 1 && output.size(2) == H && output.size(3) == W, op,
                ": output has shape ", output.sizes());
    TORCH_CHECK(gradOutput.sizes() == output.sizes(), op, ": gradOutput ", gradOutput.sizes(), " must match output ",
                output.sizes());
    TORCH_CHECK(gradInput1.sizes() == input1.sizes() && gradInput1.is_contiguous(), op,
                ": gradInput1 must be contiguous and shaped like input1");
    c10::DeviceGuard guard(input1.device());
    at::Tensor a = input1.contiguous(), o = output.contiguous();
    int64_t gs[4];
    for (int i = 0; i < 4; ++i) gs[i] = gradOutput.stride(i);
    check_rc(fn2_channelnorm_backward(a.data_ptr(), o.data_ptr(), gradOutput.data_ptr(), gs, gradInput1.data_ptr(), dt,
                                      B, C, H, W, current_stream(input1)), op);
    return 1;
}

// forward / backward with their outputs allocated here (the wrappers of this repository; the reference's signatures stay above)
at::Tensor channelnorm_forward_alloc(at::Tensor &input1, int norm_deg)
{
    check_gpu(input1, "channelnorm_cuda.forward_alloc", "input1");
    TORCH_CHECK(input1.dim() == 4, "channelnorm_cuda.forward_alloc: input1 must be 4-D");
    c10::DeviceGuard guard(input1.device());
    at::Tensor output = at::empty({input1.size(0), 1, input1.size(2), input1.size(3)}, input1.options());
    channelnorm_forward_hip(input1, output, norm_deg);
    return output;
}

at::Tensor channelnorm_backward_alloc(at::Tensor &input1, at::Tensor &output, at::Tensor &gradOutput, int norm_deg)
{
    check_gpu(input1, "channelnorm_cuda.backward_alloc", "input1");
    c10::DeviceGuard guard(input1.device());
    at::Tensor g = at::empty(input1.sizes(), input1.options());
    channelnorm_backward_hip(input1, output, gradOutput, g, norm_deg);
    return g;
}

// ---- autograd Function on the C++ side; semantics of networks/channelnorm_package/channelnorm.py
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct ChannelNormOp : public torch::autograd::Function<ChannelNormOp> {
    static at::Tensor forward(AutogradContext *ctx, const at::Tensor &input1, int64_t norm_deg)
    {
        TORCH_CHECK(input1.is_contiguous(), "ChannelNormFunction: input1 must be contiguous (reference channelnorm.py:9)");
        at::Tensor a = input1;
        at::Tensor output = channelnorm_forward_alloc(a, (int)norm_deg);
        ctx->save_for_backward({input1, output});
        ctx->saved_data["n"] = norm_deg;
        return output;
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_outputs)
    {
        TORCH_CHECK(!(grad_outputs[0].requires_grad() && at::GradMode::is_enabled()),
                    "ChannelNormFunction.backward: a HIP kernel, not differentiable a second time (create_graph=True)");
        auto saved = ctx->get_saved_variables();
        at::Tensor a = saved[0], o = saved[1], go = grad_outputs[0];
        return {channelnorm_backward_alloc(a, o, go, (int)ctx->saved_data["n"].toInt()), at::Tensor()};
    }
};

at::Tensor channelnorm_apply(const at::Tensor &input1, int64_t norm_deg)
{
    return ChannelNormOp::apply(input1, norm_deg);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("apply", &channelnorm_apply, "ChannelNormFunction.apply: differentiable, autograd node on the C++ side", py::arg("input1"), py::arg("norm_deg") = 2);
    m.def("forward_alloc", &channelnorm_forward_alloc, "forward returning a freshly allocated output");
    m.def("backward_alloc", &channelnorm_backward_alloc, "backward returning a freshly allocated gradient");
    m.doc() = "FlowNet2 ChannelNorm layer, gfx950 HIP kernels (drop-in for the reference channelnorm_cuda)";
    m.def("forward", &channelnorm_forward_hip, "Channel norm forward (HIP, gfx950)");
    m.def("backward", &channelnorm_backward_hip, "Channel norm backward (HIP, gfx950)");
}
