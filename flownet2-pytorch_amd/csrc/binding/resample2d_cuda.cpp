// resample2d_cuda.cpp -- pybind module `resample2d_cuda` (drop-in for the reference's module,
// resample2d_cuda.cc:6-31).  float32 only, as in the reference (resample2d_kernel.cu:221-230).
#include "binding_common.h"

using namespace fn2b;

static void strides4(const at::Tensor &t, int64_t s[4])
{
    for (int i = 0; i < 4; ++i) s[i] = t.stride(i);
}

// resample2d_cuda_forward (resample2d_cuda.cc:6-13).  `output` is allocated (zero-filled) by the
// Python wrapper with shape (B_flow, C_img, H_flow, W_flow) (resample2d.py:16-18).
int resample2d_forward_hip(at::Tensor &input1, at::Tensor &input2, at::Tensor &output, int kernel_size, bool bilinear)
{
    const char *op = "resample2d_cuda.forward";
    check_gpu(input1, op, "input1");
    check_same(input1, input2, op, "input2");
    check_same(input1, output, op, "output");
    TORCH_CHECK(input1.scalar_type() == at::kFloat, op, ": float32 tensors expected, got ", input1.scalar_type());
    TORCH_CHECK(input1.dim() == 4 && input2.dim() == 4 && output.dim() == 4, op, ": tensors must be 4-D");
    TORCH_CHECK(input2.size(1) == 2, op, ": input2 (flow) must have 2 channels, got ", input2.size(1));
    const int B = output.size(0), C = output.size(1), H = output.size(2), W = output.size(3);
    TORCH_CHECK(input2.size(0) == B && input2.size(2) == H && input2.size(3) == W, op, ": flow ", input2.sizes(),
                " does not match output ", output.sizes());
    TORCH_CHECK(input1.size(0) == B && input1.size(1) == C, op, ": input1 ", input1.sizes(), " does not match output ",
                output.sizes());
    TORCH_CHECK(output.is_contiguous(), op, ": output must be contiguous");
    c10::DeviceGuard guard(input1.device());
    at::Tensor flow = input2.contiguous();
    int64_t is[4];
    strides4(input1, is); // honoured by the kernel, like the reference's DIM3_INDEX
    check_rc(fn2_resample2d_forward(input1.data_ptr<float>(), is, flow.data_ptr<float>(), output.data_ptr<float>(), B,
                                    C, (int)input1.size(2), (int)input1.size(3), H, W, kernel_size, bilinear ? 1 : 0,
                                    current_stream(input1)), op);
    return 1;
}

// resample2d_cuda_backward (resample2d_cuda.cc:15-24).  gradInput1 arrives zero-filled and is
// accumulated into; gradInput2 is overwritten (resample2d.py:31-36).
int resample2d_backward_hip(at::Tensor &input1, at::Tensor &input2, at::Tensor &gradOutput, at::Tensor &gradInput1,
                            at::Tensor &gradInput2, int kernel_size, bool bilinear)
{
    const char *op = "resample2d_cuda.backward";
    check_gpu(input1, op, "input1");
    check_same(input1, input2, op, "input2");
    check_same(input1, gradOutput, op, "gradOutput");
    check_same(input1, gradInput1, op, "gradInput1");
    check_same(input1, gradInput2, op, "gradInput2");
    TORCH_CHECK(input1.scalar_type() == at::kFloat, op, ": float32 tensors expected, got ", input1.scalar_type());
    TORCH_CHECK(input1.dim() == 4 && input2.dim() == 4 && gradOutput.dim() == 4, op, ": tensors must be 4-D");
    const int B = gradOutput.size(0), C = gradOutput.size(1), H = gradOutput.size(2), W = gradOutput.size(3);
    TORCH_CHECK(input2.size(0) == B && input2.size(1) == 2 && input2.size(2) == H && input2.size(3) == W, op,
                ": flow ", input2.sizes(), " does not match gradOutput ", gradOutput.sizes());
    TORCH_CHECK(input1.size(0) == B && input1.size(1) == C, op, ": input1 ", input1.sizes(),
                " does not match gradOutput ", gradOutput.sizes());
    TORCH_CHECK(gradInput1.sizes() == input1.sizes() && gradInput1.is_contiguous(), op,
                ": gradInput1 must be contiguous and shaped like input1");
    TORCH_CHECK(gradInput2.sizes() == input2.sizes() && gradInput2.is_contiguous(), op,
                ": gradInput2 must be contiguous and shaped like input2");
    c10::DeviceGuard guard(input1.device());
    at::Tensor flow = input2.contiguous(), go = gradOutput.contiguous();
    int64_t is[4];
    strides4(input1, is);
    check_rc(fn2_resample2d_backward(input1.data_ptr<float>(), is, flow.data_ptr<float>(), go.data_ptr<float>(),
                                     gradInput1.data_ptr<float>(), gradInput2.data_ptr<float>(), B, C,
                                     (int)input1.size(2), (int)input1.size(3), H, W, kernel_size, bilinear ? 1 : 0,
                                     current_stream(input1)), op);
    return 1;
}

// N2, not in the reference module: cat(pair, warp(pair[:, C:], flow), flow / div_flow,
// channelnorm(pair[:, :C] - warped)) in one pass (models.py:133-138).
int warp_diff_norm_cat_hip(at::Tensor &pair, at::Tensor &flow, at::Tensor &output, double div_flow, bool bilinear)
{
    const char *op = "resample2d_cuda.warp_diff_norm_cat";
    check_gpu(pair, op, "pair");
    check_same(pair, flow, op, "flow");
    check_same(pair, output, op, "output");
    TORCH_CHECK(pair.scalar_type() == at::kFloat, op, ": float32 tensors expected, got ", pair.scalar_type());
    TORCH_CHECK(pair.dim() == 4 && flow.dim() == 4 && output.dim() == 4, op, ": tensors must be 4-D");
    TORCH_CHECK(pair.size(1) % 2 == 0 && pair.size(1) > 0, op, ": pair must hold two images (even channel count), got ",
                pair.sizes());
    const int B = pair.size(0), C = pair.size(1) / 2, H = pair.size(2), W = pair.size(3);
    TORCH_CHECK(flow.size(0) == B && flow.size(1) == 2 && flow.size(2) == H && flow.size(3) == W, op, ": flow ",
                flow.sizes(), " does not match pair ", pair.sizes());
    TORCH_CHECK(output.size(0) == B && output.size(1) == 3 * C + 3 && output.size(2) == H && output.size(3) == W &&
                    output.is_contiguous(),
                op, ": output must be contiguous [", B, ", ", 3 * C + 3, ", ", H, ", ", W, "], got ", output.sizes());
    c10::DeviceGuard guard(pair.device());
    at::Tensor p = pair.contiguous(), f = flow.contiguous();
    check_rc(fn2_warp_diff_norm_cat(p.data_ptr<float>(), f.data_ptr<float>(), output.data_ptr<float>(), (float)div_flow,
                                    B, C, H, W, bilinear ? 1 : 0, current_stream(pair)), op);
    return 1;
}

// Backward of warp_diff_norm_cat (fn2_warp_diff_norm_cat_backward).  gradPair: an empty tensor (numel 0) = not wanted.
int warp_diff_norm_cat_backward_hip(at::Tensor &pair, at::Tensor &flow, at::Tensor &output, at::Tensor &gradOutput,
                                    at::Tensor &gradPair, at::Tensor &gradFlow, double div_flow, bool bilinear)
{
    const char *op = "resample2d_cuda.warp_diff_norm_cat_backward";
    check_gpu(pair, op, "pair");
    check_same(pair, flow, op, "flow");
    check_same(pair, output, op, "output");
    check_same(pair, gradOutput, op, "gradOutput");
    check_same(pair, gradFlow, op, "gradFlow");
    TORCH_CHECK(pair.scalar_type() == at::kFloat, op, ": float32 tensors expected, got ", pair.scalar_type());
    TORCH_CHECK(pair.dim() == 4 && flow.dim() == 4 && output.dim() == 4 && gradOutput.dim() == 4, op, ": tensors must be 4-D");
    TORCH_CHECK(pair.size(1) % 2 == 0 && pair.size(1) > 0, op, ": pair must hold two images, got ", pair.sizes());
    const int B = pair.size(0), C = pair.size(1) / 2, H = pair.size(2), W = pair.size(3);
    TORCH_CHECK(flow.size(0) == B && flow.size(1) == 2 && flow.size(2) == H && flow.size(3) == W, op, ": flow ", flow.sizes(),
                " does not match pair ", pair.sizes());
    TORCH_CHECK(output.sizes() == gradOutput.sizes() && output.size(0) == B && output.size(1) == 3 * C + 3 && output.size(2) == H &&
                    output.size(3) == W, op, ": output / gradOutput must be [", B, ", ", 3 * C + 3, ", ", H, ", ", W, "]");
    TORCH_CHECK(pair.is_contiguous() && flow.is_contiguous() && output.is_contiguous(), op, ": pair, flow and output must be contiguous");
    TORCH_CHECK(gradFlow.sizes() == flow.sizes() && gradFlow.is_contiguous(), op, ": gradFlow must be contiguous and shaped like flow");
    const bool want_pair = gradPair.defined() && gradPair.numel() > 0;
    if (want_pair) {
        check_same(pair, gradPair, op, "gradPair");
        TORCH_CHECK(gradPair.sizes() == pair.sizes() && gradPair.is_contiguous(), op, ": gradPair must be contiguous and shaped like pair");
    }
    c10::DeviceGuard guard(pair.device());
    at::Tensor go = gradOutput.contiguous();
    check_rc(fn2_warp_diff_norm_cat_backward(pair.data_ptr<float>(), flow.data_ptr<float>(), output.data_ptr<float>(), go.data_ptr<float>(),
                                             want_pair ? gradPair.data_ptr<float>() : nullptr, gradFlow.data_ptr<float>(), (float)div_flow,
                                             B, C, H, W, bilinear ? 1 : 0, current_stream(pair)), op);
    return 1;
}

// models.py:157-161 / :170-174: ||pair[:, :C] - warp(pair[:, C:], flow)||_2 (fn2_warp_diff_norm) and its flow gradient
at::Tensor warp_diff_norm_hip(at::Tensor &pair, at::Tensor &flow, bool bilinear)
{
    const char *op = "resample2d_cuda.warp_diff_norm";
    check_gpu(pair, op, "pair");
    check_same(pair, flow, op, "flow");
    TORCH_CHECK(pair.scalar_type() == at::kFloat, op, ": float32 tensors expected, got ", pair.scalar_type());
    TORCH_CHECK(pair.dim() == 4 && flow.dim() == 4 && pair.size(1) % 2 == 0 && pair.size(1) > 0, op, ": pair must be 4-D with two images");
    const int B = pair.size(0), C = pair.size(1) / 2, H = pair.size(2), W = pair.size(3);
    TORCH_CHECK(flow.size(0) == B && flow.size(1) == 2 && flow.size(2) == H && flow.size(3) == W, op, ": flow ", flow.sizes(),
                " does not match pair ", pair.sizes());
    c10::DeviceGuard guard(pair.device());
    at::Tensor p = pair.contiguous(), f = flow.contiguous();
    at::Tensor out = at::empty({B, 1, H, W}, pair.options());
    check_rc(fn2_warp_diff_norm(p.data_ptr<float>(), f.data_ptr<float>(), out.data_ptr<float>(), B, C, H, W, bilinear ? 1 : 0,
                                current_stream(pair)), op);
    return out;
}

at::Tensor warp_diff_norm_backward_hip(at::Tensor &pair, at::Tensor &flow, at::Tensor &norm, at::Tensor &gradNorm, bool bilinear)
{
    const char *op = "resample2d_cuda.warp_diff_norm_backward";
    check_gpu(pair, op, "pair");
    check_same(pair, flow, op, "flow");
    check_same(pair, norm, op, "norm");
    check_same(pair, gradNorm, op, "gradNorm");
    TORCH_CHECK(pair.scalar_type() == at::kFloat, op, ": float32 tensors expected, got ", pair.scalar_type());
    TORCH_CHECK(pair.dim() == 4 && flow.dim() == 4 && pair.size(1) % 2 == 0 && pair.size(1) > 0, op, ": pair must be 4-D with two images");
    const int B = pair.size(0), C = pair.size(1) / 2, H = pair.size(2), W = pair.size(3);
    TORCH_CHECK(flow.size(0) == B && flow.size(1) == 2 && flow.size(2) == H && flow.size(3) == W, op, ": flow ", flow.sizes(),
                " does not match pair ", pair.sizes());
    TORCH_CHECK(norm.sizes() == gradNorm.sizes() && norm.dim() == 4 && norm.size(0) == B && norm.size(1) == 1 && norm.size(2) == H &&
                    norm.size(3) == W, op, ": norm / gradNorm must be [", B, ", 1, ", H, ", ", W, "]");
    TORCH_CHECK(pair.is_contiguous() && flow.is_contiguous() && norm.is_contiguous(), op, ": pair, flow and norm must be contiguous");
    c10::DeviceGuard guard(pair.device());
    at::Tensor gn = gradNorm.contiguous();
    at::Tensor gflow = at::empty(flow.sizes(), flow.options());
    check_rc(fn2_warp_diff_norm_backward(pair.data_ptr<float>(), flow.data_ptr<float>(), norm.data_ptr<float>(), gn.data_ptr<float>(),
                                         gflow.data_ptr<float>(), B, C, H, W, bilinear ? 1 : 0, current_stream(pair)), op);
    return gflow;
}

// forward / backward with their outputs allocated here (the wrappers of this repository; the reference's signatures stay above)
at::Tensor resample2d_forward_alloc(at::Tensor &input1, at::Tensor &input2, int kernel_size, bool bilinear)
{
    const char *op = "resample2d_cuda.forward_alloc";
    check_gpu(input1, op, "input1");
    check_same(input1, input2, op, "input2");
    TORCH_CHECK(input1.dim() == 4 && input2.dim() == 4, op, ": tensors must be 4-D");
    c10::DeviceGuard guard(input1.device());
    at::Tensor output = at::empty({input2.size(0), input1.size(1), input2.size(2), input2.size(3)}, input1.options());   // (resample2d.py:16-18)
    resample2d_forward_hip(input1, input2, output, kernel_size, bilinear);
    return output;
}

std::vector<at::Tensor> resample2d_backward_alloc(at::Tensor &input1, at::Tensor &input2, at::Tensor &gradOutput, int kernel_size, bool bilinear)
{
    const char *op = "resample2d_cuda.backward_alloc";
    check_gpu(input1, op, "input1");
    check_same(input1, input2, op, "input2");
    c10::DeviceGuard guard(input1.device());
    at::Tensor g1 = at::zeros(input1.sizes(), input1.options());        // accumulated into: starts at zero (resample2d.py:31)
    at::Tensor g2 = at::empty(input2.sizes(), input2.options());        // fully written
    resample2d_backward_hip(input1, input2, gradOutput, g1, g2, kernel_size, bilinear);
    return {g1, g2};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("forward_alloc", &resample2d_forward_alloc, "forward returning a freshly allocated output");
    m.def("backward_alloc", &resample2d_backward_alloc, "backward returning freshly allocated gradients");
    m.doc() = "FlowNet2 Resample2d layer, gfx950 HIP kernels (drop-in for the reference resample2d_cuda)";
    m.def("forward", &resample2d_forward_hip, "Resample2D forward (HIP, gfx950)");
    m.def("backward", &resample2d_backward_hip, "Resample2D backward (HIP, gfx950)");
    m.def("warp_diff_norm_cat", &warp_diff_norm_cat_hip,
          "cat(pair, warp(second image, flow), flow / div_flow, ||first image - warped||) in one pass");
    m.def("warp_diff_norm_cat_backward", &warp_diff_norm_cat_backward_hip, "backward of warp_diff_norm_cat in one pass");
    m.def("warp_diff_norm", &warp_diff_norm_hip, "||first image - warp(second image, flow)|| in one pass (models.py:157-161)");
    m.def("warp_diff_norm_backward", &warp_diff_norm_backward_hip, "flow gradient of warp_diff_norm in one pass");
}
