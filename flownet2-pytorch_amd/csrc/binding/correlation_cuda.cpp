// correlation_cuda.cpp -- pybind module `correlation_cuda` (drop-in for the reference's module of
// the same name, correlation_cuda.cc:10-172): forward / backward with the reference's positional
// signature, tensors owned by Python, outputs resized in place.
#include "binding_common.h"

using namespace fn2b;

// correlation_forward_cuda (correlation_cuda.cc:10-87).  rInput1 / rInput2 are the reference's
// padded-NHWC scratch tensors; this implementation needs no scratch and leaves them untouched.
int correlation_forward_hip(at::Tensor &input1, at::Tensor &input2, at::Tensor &rInput1, at::Tensor &rInput2,
                            at::Tensor &output, int pad_size, int kernel_size, int max_displacement, int stride1,
                            int stride2, int corr_type_multiply)
{
    (void)rInput1; (void)rInput2; (void)corr_type_multiply; // accepted, unused (as in the reference kernels)
    const char *op = "correlation_cuda.forward";
    check_gpu(input1, op, "input1");
    check_same(input1, input2, op, "input2");
    check_same(input1, output, op, "output");
    TORCH_CHECK(input1.dim() == 4 && input2.dim() == 4, op, ": inputs must be 4-D (N, C, H, W)");
    TORCH_CHECK(input1.sizes() == input2.sizes(), op, ": input1 ", input1.sizes(), " and input2 ", input2.sizes(),
                " must have the same shape");
    const int dt = dtype_of(input1, op);
    const int B = input1.size(0), C = input1.size(1), H = input1.size(2), W = input1.size(3);
    int nOut = 0, oH = 0, oW = 0;
    check_rc(fn2_correlation_output_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &nOut, &oH,
                                          &oW), op);
    c10::DeviceGuard guard(input1.device());
    at::Tensor a = input1.contiguous(), b = input2.contiguous();
    output.resize_({B, nOut, oH, oW}); // correlation_cuda.cc:38; fully written by the kernel, no fill_(0)
    TORCH_CHECK(output.is_contiguous(), op, ": output must be contiguous");
    check_rc(fn2_correlation_forward(a.data_ptr(), b.data_ptr(), output.data_ptr(), dt, B, C, H, W, pad_size,
                                     kernel_size, max_displacement, stride1, stride2, current_stream(input1)), op);
    return 1;
}

// N1, not in the reference module: correlation forward + LeakyReLU(negative_slope) written straight into
// channels [channel_offset, channel_offset + nOut) of `buffer` (N x Ctot x oH x oW, contiguous) -- the
// torch.cat((conv_redir, corr), 1) input of conv3_1 (FlowNetC.py:87,92) without the activation and concat passes.
int correlation_forward_fused_hip(at::Tensor &input1, at::Tensor &input2, at::Tensor &buffer, int channel_offset,
                                  double negative_slope, int pad_size, int kernel_size, int max_displacement, int stride1,
                                  int stride2)
{
    const char *op = "correlation_cuda.forward_fused";
    check_gpu(input1, op, "input1");
    check_same(input1, input2, op, "input2");
    check_same(input1, buffer, op, "buffer");
    TORCH_CHECK(input1.dim() == 4 && input1.sizes() == input2.sizes(), op, ": inputs must be 4-D and equally shaped");
    const int dt = dtype_of(input1, op);
    const int B = input1.size(0), C = input1.size(1), H = input1.size(2), W = input1.size(3);
    int nOut = 0, oH = 0, oW = 0;
    check_rc(fn2_correlation_output_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &nOut, &oH,
                                          &oW), op);
    TORCH_CHECK(buffer.dim() == 4 && buffer.is_contiguous() && buffer.size(0) == B && buffer.size(2) == oH &&
                    buffer.size(3) == oW && channel_offset >= 0 && channel_offset + nOut <= buffer.size(1),
                op, ": buffer ", buffer.sizes(), " cannot hold ", nOut, " channels of ", oH, "x", oW, " at channel ",
                channel_offset);
    c10::DeviceGuard guard(input1.device());
    at::Tensor a = input1.contiguous(), b = input2.contiguous();
    char *dst = static_cast<char *>(buffer.data_ptr()) + (int64_t)channel_offset * oH * oW * buffer.element_size();
    check_rc(fn2_correlation_forward_fused(a.data_ptr(), b.data_ptr(), dst, buffer.size(1) * (int64_t)oH * oW,
                                           (float)negative_slope, dt, B, C, H, W, pad_size, kernel_size,
                                           max_displacement, stride1, stride2, FN2_CORR_AUTO, current_stream(input1)), op);
    return 1;
}

// Half tensors on maps wider than 64 px (Sintel-size conv3): there is no tiled half kernel for that corner (the narrow one holds
// whole rows of <= 64 px), and the general kernel takes milliseconds.  The fp32 column-window kernel on widened copies is
// ~25x faster and a superset numerically (exact products of the half operands, fp32 sums; the reference sums in half,
// correlation_cuda_kernel.cu:229): widen, run, narrow -- three elementwise passes around a 0.3 ms kernel instead of 9 ms.
// Shared by backward and backward_fused.  `go`: the contiguous half gradient.  false = not this corner (or a shape the
// fp32 launchers decline): the caller's half path takes it.
// The shapes that corner covers (pure predicate: callers test it BEFORE preparing operands for it).
static bool half_wide_applies(int dt, int C, int H, int W, int pad_size, int kernel_size, int max_displacement, int stride1, int stride2)
{
    return dt == FN2_F16 && W > 64 && kernel_size == 1 && stride1 == 1 && stride2 == 2 && pad_size == max_displacement &&
           max_displacement == 20 && C % 64 == 0 && H % 2 == 0 && W % 8 == 0;
}

static bool half_wide_backward(const at::Tensor &a, const at::Tensor &b, const at::Tensor &go, at::Tensor &gradInput1, at::Tensor &gradInput2,
                               int dt, int B, int C, int H, int W, int pad_size, int kernel_size, int max_displacement, int stride1,
                               int stride2)
{
    if (!half_wide_applies(dt, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2))
        return false;
    at::Tensor a32 = a.to(at::kFloat), b32 = b.to(at::kFloat), go32 = go.to(at::kFloat);
    at::Tensor g1 = at::empty_like(a32), g2 = at::empty_like(b32);
    const int rc = fn2_correlation_backward(a32.data_ptr(), b32.data_ptr(), go32.data_ptr(), g1.data_ptr(), g2.data_ptr(), FN2_F32,
                                            B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, current_stream(a));
    if (rc != FN2_OK) return false;
    gradInput1.copy_(g1);
    gradInput2.copy_(g2);
    return true;
}

// correlation_backward_cuda (correlation_cuda.cc:89-167)
int correlation_backward_hip(at::Tensor &input1, at::Tensor &input2, at::Tensor &rInput1, at::Tensor &rInput2,
                             at::Tensor &gradOutput, at::Tensor &gradInput1, at::Tensor &gradInput2, int pad_size,
                             int kernel_size, int max_displacement, int stride1, int stride2, int corr_type_multiply)
{
    (void)rInput1; (void)rInput2; (void)corr_type_multiply;
    const char *op = "correlation_cuda.backward";
    check_gpu(input1, op, "input1");
    check_same(input1, input2, op, "input2");
    check_same(input1, gradOutput, op, "gradOutput");
    check_same(input1, gradInput1, op, "gradInput1");
    check_same(input1, gradInput2, op, "gradInput2");
    TORCH_CHECK(input1.dim() == 4 && input1.sizes() == input2.sizes(), op, ": inputs must be 4-D and equally shaped");
    const int dt = dtype_of(input1, op);
    const int B = input1.size(0), C = input1.size(1), H = input1.size(2), W = input1.size(3);
    int nOut = 0, oH = 0, oW = 0;
    check_rc(fn2_correlation_output_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &nOut, &oH,
                                          &oW), op);
    TORCH_CHECK(gradOutput.dim() == 4 && gradOutput.size(0) == B && gradOutput.size(1) == nOut &&
                    gradOutput.size(2) == oH && gradOutput.size(3) == oW,
                op, ": gradOutput has shape ", gradOutput.sizes(), ", expected [", B, ", ", nOut, ", ", oH, ", ", oW, "]");
    c10::DeviceGuard guard(input1.device());
    at::Tensor a = input1.contiguous(), b = input2.contiguous();
    at::Tensor go = gradOutput.contiguous(); // the reference assumes contiguity without checking (SURVEY.md b)
    gradInput1.resize_({B, C, H, W});        // correlation_cuda.cc:108-109; fully written, no fill_(0)
    gradInput2.resize_({B, C, H, W});
    TORCH_CHECK(gradInput1.is_contiguous() && gradInput2.is_contiguous(), op, ": gradInputs must be contiguous");
    if (half_wide_backward(a, b, go, gradInput1, gradInput2, dt, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2))
        return 1;
    check_rc(fn2_correlation_backward(a.data_ptr(), b.data_ptr(), go.data_ptr(), gradInput1.data_ptr(),
                                      gradInput2.data_ptr(), dt, B, C, H, W, pad_size, kernel_size, max_displacement,
                                      stride1, stride2, current_stream(input1)), op);
    return 1;
}

// N1, training half, not in the reference module: gradients of the correlation branch of
// cat((conv_redir, LeakyReLU(Correlation(input1, input2))), 1) (FlowNetC.py:86-87, :92).  `buffer` is the concat buffer forward_fused
// wrote, `gradBuffer` the gradient wrt it (same shape): the slice of gradBuffer is read in place and the activation's derivative
// comes from the sign of the stored output -- what autograd does with a contiguous copy of the slice and leaky_relu_backward.
int correlation_backward_fused_hip(at::Tensor &input1, at::Tensor &input2, at::Tensor &buffer, at::Tensor &gradBuffer,
                                   int channel_offset, double negative_slope, at::Tensor &gradInput1, at::Tensor &gradInput2,
                                   int pad_size, int kernel_size, int max_displacement, int stride1, int stride2)
{
    const char *op = "correlation_cuda.backward_fused";
    check_gpu(input1, op, "input1");
    check_same(input1, input2, op, "input2");
    check_same(input1, buffer, op, "buffer");
    check_same(input1, gradBuffer, op, "gradBuffer");
    check_same(input1, gradInput1, op, "gradInput1");
    check_same(input1, gradInput2, op, "gradInput2");
    TORCH_CHECK(input1.dim() == 4 && input1.sizes() == input2.sizes(), op, ": inputs must be 4-D and equally shaped");
    const int dt = dtype_of(input1, op);
    const int B = input1.size(0), C = input1.size(1), H = input1.size(2), W = input1.size(3);
    int nOut = 0, oH = 0, oW = 0;
    check_rc(fn2_correlation_output_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &nOut, &oH,
                                          &oW), op);
    TORCH_CHECK(buffer.dim() == 4 && buffer.is_contiguous() && buffer.size(0) == B && buffer.size(2) == oH &&
                    buffer.size(3) == oW && channel_offset >= 0 && channel_offset + nOut <= buffer.size(1),
                op, ": buffer ", buffer.sizes(), " does not hold ", nOut, " channels of ", oH, "x", oW, " at channel ", channel_offset);
    TORCH_CHECK(gradBuffer.sizes() == buffer.sizes(), op, ": gradBuffer ", gradBuffer.sizes(), " must have the buffer's shape ", buffer.sizes());
    c10::DeviceGuard guard(input1.device());
    at::Tensor a = input1.contiguous(), b = input2.contiguous();
    at::Tensor gb = gradBuffer.contiguous();          // what autograd hands over is contiguous already; a view would be copied here
    gradInput1.resize_({B, C, H, W});
    gradInput2.resize_({B, C, H, W});
    TORCH_CHECK(gradInput1.is_contiguous() && gradInput2.is_contiguous(), op, ": gradInputs must be contiguous");
    // the activation's derivative is read off the SIGN of the stored output: only an increasing activation keeps it
    // (fn2_correlation_backward_fused enforces the same; checked here so that no branch below can run without it)
    TORCH_CHECK(negative_slope > 0, op, ": negative_slope must be > 0, got ", negative_slope);
    if (half_wide_applies(dt, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)) {
        // the masked gradient as autograd's leaky_relu_backward forms it for half tensors (fp32 product, rounded to half once), then
        // the same widened path as `backward`: fused and unfused training give the same gradients on Sintel-size maps too
        at::Tensor outs = buffer.narrow(1, channel_offset, nOut), gs = gb.narrow(1, channel_offset, nOut).to(at::kFloat);
        at::Tensor masked = at::where(outs > 0, gs, gs * (float)negative_slope).to(at::kHalf).contiguous();
        if (half_wide_backward(a, b, masked, gradInput1, gradInput2, dt, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2))
            return 1;
    }
    const size_t wsb = fn2_correlation_backward_fused_workspace_bytes(dt, B, H, W, pad_size, kernel_size, max_displacement, stride1, stride2);
    at::Tensor ws = at::empty({(int64_t)B, (int64_t)nOut, (int64_t)oH, (int64_t)oW}, input1.options());   // caching allocator: no synchronisation
    const int64_t off = (int64_t)channel_offset * oH * oW * buffer.element_size(), bs = buffer.size(1) * (int64_t)oH * oW;
    check_rc(fn2_correlation_backward_fused(a.data_ptr(), b.data_ptr(), static_cast<char *>(buffer.data_ptr()) + off, bs,
                                            static_cast<char *>(gb.data_ptr()) + off, bs, (float)negative_slope, ws.data_ptr(), wsb,
                                            gradInput1.data_ptr(), gradInput2.data_ptr(), dt, B, C, H, W, pad_size, kernel_size,
                                            max_displacement, stride1, stride2, FN2_CORR_AUTO, current_stream(input1)), op);
    return 1;
}

// The same two calls for the wrappers of this repository: outputs allocated here (one Python -> C++ transition per op instead of
// one per tensor; the reference's signatures above stay for code written against them)
at::Tensor correlation_forward_alloc(at::Tensor &input1, at::Tensor &input2, int pad_size, int kernel_size, int max_displacement,
                                     int stride1, int stride2, int corr_type_multiply)
{
    check_gpu(input1, "correlation_cuda.forward_alloc", "input1");
    c10::DeviceGuard guard(input1.device());
    at::Tensor scratch1 = at::empty({0}, input1.options()), scratch2 = at::empty({0}, input1.options()), output = at::empty({0}, input1.options());
    correlation_forward_hip(input1, input2, scratch1, scratch2, output, pad_size, kernel_size, max_displacement, stride1, stride2, corr_type_multiply);
    return output;
}

std::vector<at::Tensor> correlation_backward_alloc(at::Tensor &input1, at::Tensor &input2, at::Tensor &gradOutput, int pad_size,
                                                   int kernel_size, int max_displacement, int stride1, int stride2, int corr_type_multiply)
{
    check_gpu(input1, "correlation_cuda.backward_alloc", "input1");
    c10::DeviceGuard guard(input1.device());
    at::Tensor scratch1 = at::empty({0}, input1.options()), scratch2 = at::empty({0}, input1.options());
    at::Tensor g1 = at::empty({0}, input1.options()), g2 = at::empty({0}, input1.options());
    correlation_backward_hip(input1, input2, scratch1, scratch2, gradOutput, g1, g2, pad_size, kernel_size, max_displacement, stride1, stride2,
                             corr_type_multiply);
    return {g1, g2};
}

// ---- autograd Functions on the C++ side: `Correlation` / `CorrelationLeakyReLUCat` do no Python between
// `apply` and the launch, and the engine calls the backward without taking the GIL.  Same semantics as the Python Functions of
// networks/correlation_package/correlation.py (kept there for code written against CorrelationFunction.forward / .backward).
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

static void check_once_differentiable(const variable_list &grads, const char *op)
{
    for (const auto &g : grads)
        TORCH_CHECK(!(g.defined() && g.requires_grad() && at::GradMode::is_enabled()), op,
                    ": the backward of this layer is a HIP kernel and not differentiable a second time (create_graph=True)");
}

struct CorrelationOp : public torch::autograd::Function<CorrelationOp> {
    static at::Tensor forward(AutogradContext *ctx, const at::Tensor &input1, const at::Tensor &input2, int64_t pad_size, int64_t kernel_size,
                              int64_t max_displacement, int64_t stride1, int64_t stride2, int64_t corr_multiply)
    {
        ctx->save_for_backward({input1, input2});
        ctx->saved_data["p"] = std::vector<int64_t>{pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply};
        at::Tensor a = input1, b = input2;
        return correlation_forward_alloc(a, b, (int)pad_size, (int)kernel_size, (int)max_displacement, (int)stride1, (int)stride2, (int)corr_multiply);
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_outputs)
    {
        check_once_differentiable(grad_outputs, "CorrelationFunction.backward");
        auto saved = ctx->get_saved_variables();
        const auto p = ctx->saved_data["p"].toIntVector();
        at::Tensor a = saved[0], b = saved[1], go = grad_outputs[0];
        auto g = correlation_backward_alloc(a, b, go, (int)p[0], (int)p[1], (int)p[2], (int)p[3], (int)p[4], (int)p[5]);
        return {g[0], g[1], at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor correlation_apply(const at::Tensor &input1, const at::Tensor &input2, int64_t pad_size, int64_t kernel_size, int64_t max_displacement,
                             int64_t stride1, int64_t stride2, int64_t corr_multiply)
{
    return CorrelationOp::apply(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply);
}

// cat((redir, leaky_relu(corr(input1, input2), slope)), 1) (FlowNetC.py:86-87, :92) as one differentiable op
struct CorrelationLeakyReLUCatOp : public torch::autograd::Function<CorrelationLeakyReLUCatOp> {
    static at::Tensor forward(AutogradContext *ctx, const at::Tensor &input1, const at::Tensor &input2, const at::Tensor &redir, int64_t pad_size,
                              int64_t kernel_size, int64_t max_displacement, int64_t stride1, int64_t stride2, double negative_slope)
    {
        const char *op = "CorrelationLeakyReLUCat";
        check_gpu(input1, op, "input1");
        check_same(input1, redir, op, "redir");
        TORCH_CHECK(redir.dim() == 4, op, ": redir must be 4-D");
        TORCH_CHECK(stride2 > 0, op, ": stride2 must be positive");
        const int64_t r = max_displacement / stride2, n_out = (2 * r + 1) * (2 * r + 1), Cr = redir.size(1);
        c10::DeviceGuard guard(input1.device());
        at::Tensor buf = at::empty({redir.size(0), Cr + n_out, redir.size(2), redir.size(3)}, redir.options());
        buf.narrow(1, 0, Cr).copy_(redir);
        at::Tensor a = input1, b = input2;
        correlation_forward_fused_hip(a, b, buf, (int)Cr, negative_slope, (int)pad_size, (int)kernel_size, (int)max_displacement, (int)stride1, (int)stride2);
        ctx->save_for_backward({input1, input2, buf});
        ctx->saved_data["p"] = std::vector<int64_t>{pad_size, kernel_size, max_displacement, stride1, stride2, Cr};
        ctx->saved_data["slope"] = negative_slope;
        return buf;
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_outputs)
    {
        check_once_differentiable(grad_outputs, "CorrelationLeakyReLUCatFunction.backward");
        auto saved = ctx->get_saved_variables();
        const auto p = ctx->saved_data["p"].toIntVector();
        const double slope = ctx->saved_data["slope"].toDouble();
        at::Tensor a = saved[0], b = saved[1], buf = saved[2], gbuf = grad_outputs[0];
        at::Tensor g1, g2, gr;
        if (ctx->needs_input_grad(0) || ctx->needs_input_grad(1)) {
            c10::DeviceGuard guard(a.device());
            g1 = at::empty({0}, a.options());
            g2 = at::empty({0}, a.options());
            correlation_backward_fused_hip(a, b, buf, gbuf, (int)p[5], slope, g1, g2, (int)p[0], (int)p[1], (int)p[2], (int)p[3], (int)p[4]);
        }
        if (ctx->needs_input_grad(2)) gr = gbuf.narrow(1, 0, p[5]);
        return {g1, g2, gr, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor correlation_leakyrelu_cat_apply(const at::Tensor &input1, const at::Tensor &input2, const at::Tensor &redir, int64_t pad_size,
                                           int64_t kernel_size, int64_t max_displacement, int64_t stride1, int64_t stride2, double negative_slope)
{
    return CorrelationLeakyReLUCatOp::apply(input1, input2, redir, pad_size, kernel_size, max_displacement, stride1, stride2, negative_slope);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("apply", &correlation_apply, "CorrelationFunction.apply: differentiable, autograd node on the C++ side", py::arg("input1"), py::arg("input2"),
          py::arg("pad_size") = 3, py::arg("kernel_size") = 3, py::arg("max_displacement") = 20, py::arg("stride1") = 1, py::arg("stride2") = 2,
          py::arg("corr_multiply") = 1);
    m.def("leakyrelu_cat_apply", &correlation_leakyrelu_cat_apply, "CorrelationLeakyReLUCatFunction.apply: differentiable, autograd node on the C++ side");
    m.def("forward_alloc", &correlation_forward_alloc, "forward returning a freshly allocated output");
    m.def("backward_alloc", &correlation_backward_alloc, "backward returning freshly allocated gradients");
    m.doc() = "FlowNet2 correlation layer, gfx950 HIP kernels (drop-in for the reference correlation_cuda)";
    m.def("forward", &correlation_forward_hip, "Correlation forward (HIP, gfx950)");
    m.def("backward", &correlation_backward_hip, "Correlation backward (HIP, gfx950)");
    m.def("forward_fused", &correlation_forward_fused_hip,
          "Correlation forward + LeakyReLU written into a channel slice of a concat buffer");
    m.def("backward_fused", &correlation_backward_fused_hip,
          "Gradients of the correlation branch of cat((redir, LeakyReLU(corr))) from the concat gradient and the stored output");
}
