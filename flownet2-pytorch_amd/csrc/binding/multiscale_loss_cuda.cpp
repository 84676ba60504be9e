// multiscale_loss_cuda.cpp -- pybind module `multiscale_loss_cuda`: SURVEY.md 8f row N3, FlowNet2's training loss
// (reference losses.py:52-86, MultiScale with norm 'L1' or 'L2') and the EPE metric (:11-12) as ONE autograd node over
// fn2_multiscale_loss_fused / fn2_multiscale_scale_grads.  Not a module of the reference (its loss is plain PyTorch); named
// after the three it does have.  Round 6: the node lives here instead of in Python + ctypes -- forward =
// one launch (loss and metric written by the kernel's last workgroup), backward = one launch (five gradients scaled by the
// incoming gradient), no host arrays marshalled per call, the scratch memory cached per device.
#include "binding_common.h"

#include <mutex>
#include <unordered_map>

using namespace fn2b;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

// One primed workspace per (device, stream, size): zero-filled once, handed to the kernel with workspace_primed = 1 (the kernel
// leaves its ticket counter at zero).  Keyed by the stream because two streams may run the loss concurrently.
struct WsKey {
    int device;
    void *stream;
    size_t bytes;
    bool operator==(const WsKey &o) const { return device == o.device && stream == o.stream && bytes == o.bytes; }
};
struct WsHash {
    size_t operator()(const WsKey &k) const { return std::hash<void *>()(k.stream) ^ (std::hash<size_t>()(k.bytes) * 31u) ^ (size_t)k.device; }
};
std::mutex ws_mutex;
// (never destroyed: device tensors must not be freed by a static destructor after the HIP runtime has shut down)
auto &ws_cache = *new std::unordered_map<WsKey, at::Tensor, WsHash>();

at::Tensor primed_workspace(const at::Tensor &like, void *stream, size_t bytes)
{
    const WsKey key{(int)like.device().index(), stream, bytes};
    std::lock_guard<std::mutex> lock(ws_mutex);
    auto it = ws_cache.find(key);
    if (it != ws_cache.end()) return it->second;
    at::Tensor ws = at::zeros({(int64_t)((bytes + 3) / 4)}, like.options().dtype(at::kFloat));
    ws_cache.emplace(key, ws);
    return ws;
}

struct MultiScaleOp : public torch::autograd::Function<MultiScaleOp> {
    // returns {loss, epe}; epe is not differentiable
    // want_grads: decided by the caller of apply() (grad mode on and a prediction requires a gradient) -- inside forward grad mode is
    // off and, when nothing requires a gradient, the node has no edges to ask
    static variable_list forward(AutogradContext *ctx, const at::Tensor &target_, at::TensorList outputs, int64_t start_scale, double div_flow,
                                 std::vector<double> weights, int64_t norm, bool want_grads)
    {
        const char *op = "multiscale_loss_cuda.apply";
        const int n = (int)outputs.size();
        check_gpu(target_, op, "target");
        TORCH_CHECK(n >= 1 && n <= 6 && (int)weights.size() == n, op, ": 1..6 predictions with one weight each expected, got ", n, " / ", weights.size());
        TORCH_CHECK(target_.dim() == 4 && target_.size(1) == 2 && target_.scalar_type() == at::kFloat, op, ": target must be float32 B x 2 x H x W, got ",
                    target_.sizes());
        TORCH_CHECK(norm == 1 || norm == 2, op, ": norm must be 1 (L1) or 2 (L2)");
        c10::DeviceGuard guard(target_.device());
        at::Tensor target = target_.contiguous();
        const int B = target.size(0), H = target.size(2), W = target.size(3);
        const bool any_grad = want_grads;
        std::vector<at::Tensor> outs(n), grads;
        const float *optr[6] = {nullptr};
        float *gptr[6] = {nullptr};
        float w[6] = {0};
        for (int i = 0; i < n; ++i) {
            const int64_t k = start_scale << i;
            check_same(target, outputs[i], op, "a prediction");
            TORCH_CHECK(outputs[i].dim() == 4 && outputs[i].size(0) == B && outputs[i].size(1) == 2 && outputs[i].size(2) == H / k && outputs[i].size(3) == W / k,
                        op, ": prediction ", i, " has shape ", outputs[i].sizes(), ", expected [", B, ", 2, ", H / k, ", ", W / k, "]");
            outs[i] = outputs[i].contiguous();
            optr[i] = outs[i].data_ptr<float>();
            w[i] = (float)weights[i];
        }
        if (any_grad) {
            grads.resize(n);
            for (int i = 0; i < n; ++i) {
                grads[i] = at::empty_like(outs[i]);
                gptr[i] = grads[i].data_ptr<float>();
            }
        }
        at::Tensor res = at::empty({2 + 2 * n}, target.options());      // [loss, epe, sums...]
        void *stream = current_stream(target);
        const size_t wsb = fn2_multiscale_workspace_bytes(B, H, W, (int)start_scale, n);
        TORCH_CHECK(wsb > 0, op, ": unsupported geometry (start_scale ", start_scale, ", ", n, " scales)");
        at::Tensor ws = primed_workspace(target, stream, wsb);
        float *r = res.data_ptr<float>();
        check_rc(fn2_multiscale_loss_fused(optr, target.data_ptr<float>(), r + 2, r, any_grad ? gptr : nullptr, w, 1.0f, (int)norm, B, H, W,
                                           (int)start_scale, n, (float)div_flow, ws.data_ptr(), wsb, 1, stream), op);
        if (any_grad) ctx->save_for_backward(grads);
        ctx->saved_data["n"] = (int64_t)n;
        at::Tensor loss = res.select(0, 0), epe = res.select(0, 1);
        ctx->mark_non_differentiable({epe});
        return {loss, epe};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grad_outputs)
    {
        const char *op = "multiscale_loss_cuda.backward";
        const int n = (int)ctx->saved_data["n"].toInt();
        variable_list result(1 + n + 5);                       // target, n predictions, start_scale, div_flow, weights, norm, want_grads
        auto saved = ctx->get_saved_variables();
        if (saved.empty() || !grad_outputs[0].defined()) return result;
        at::Tensor g = grad_outputs[0];
        TORCH_CHECK(!(g.requires_grad() && at::GradMode::is_enabled()), op, ": not differentiable a second time (create_graph=True)");
        check_gpu(g, op, "grad_loss");
        c10::DeviceGuard guard(g.device());
        g = g.to(at::kFloat).contiguous();
        const float *in[6] = {nullptr};
        float *out[6] = {nullptr};
        int64_t numel[6] = {0};
        for (int i = 0; i < n; ++i) {
            if (!ctx->needs_input_grad(1 + i)) continue;        // numel 0: skipped by the kernel
            result[1 + i] = at::empty_like(saved[i]);
            in[i] = saved[i].data_ptr<float>();
            out[i] = result[1 + i].data_ptr<float>();
            numel[i] = saved[i].numel();
        }
        check_rc(fn2_multiscale_scale_grads(in, out, numel, n, g.data_ptr<float>(), current_stream(g)), op);
        return result;
    }
};

std::vector<at::Tensor> multiscale_apply(const at::Tensor &target, std::vector<at::Tensor> outputs, int64_t start_scale, double div_flow,
                                         std::vector<double> weights, int64_t norm)
{
    bool want = false;
    if (at::GradMode::is_enabled())
        for (const auto &o : outputs) want = want || o.requires_grad();
    return MultiScaleOp::apply(target, at::TensorList(outputs), start_scale, div_flow, weights, norm, want);
}

// the 2*n sums alone (tests, callers without autograd): sums[i] = sum |out_i - t_i|, sums[n+i] = sum of channel 2-norms
at::Tensor multiscale_sums(const at::Tensor &target, std::vector<at::Tensor> outputs, int64_t start_scale, double div_flow)
{
    const char *op = "multiscale_loss_cuda.sums";
    const int n = (int)outputs.size();
    check_gpu(target, op, "target");
    TORCH_CHECK(n >= 1 && n <= 6, op, ": 1..6 predictions expected");
    c10::DeviceGuard guard(target.device());
    at::Tensor t = target.contiguous();
    std::vector<at::Tensor> outs(n);
    const float *optr[6] = {nullptr};
    for (int i = 0; i < n; ++i) {
        check_same(t, outputs[i], op, "a prediction");
        outs[i] = outputs[i].contiguous();
        optr[i] = outs[i].data_ptr<float>();
    }
    const int B = t.size(0), H = t.size(2), W = t.size(3);
    at::Tensor sums = at::empty({2 * n}, t.options());
    const size_t wsb = fn2_multiscale_workspace_bytes(B, H, W, (int)start_scale, n);
    TORCH_CHECK(wsb > 0, op, ": unsupported geometry");
    at::Tensor ws = at::empty({(int64_t)((wsb + 3) / 4)}, t.options());
    check_rc(fn2_multiscale_loss(optr, t.data_ptr<float>(), sums.data_ptr<float>(), nullptr, nullptr, 1.0f, 1, B, H, W, (int)start_scale, n,
                                 (float)div_flow, ws.data_ptr(), wsb, current_stream(t)), op);
    return sums;
}

} // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "FlowNet2 MultiScale loss + EPE (losses.py:52-86), gfx950 HIP kernel, autograd node on the C++ side";
    m.def("apply", &multiscale_apply, "(loss, epe) = MultiScale(outputs, target): differentiable w.r.t. the predictions", py::arg("target"),
          py::arg("outputs"), py::arg("start_scale") = 4, py::arg("div_flow") = 0.05, py::arg("weights"), py::arg("norm") = 1);
    m.def("sums", &multiscale_sums, "the 2n raw sums of the pass (no autograd)");
}
