"""Gradient exchange for one-process-per-GPU data parallelism.

The reference trains with single-process ``nn.DataParallel`` (main.py:187-201): every step it re-broadcasts the 156.7 MB
of FlowNet2C parameters to all GPUs and reduces the gradients onto GPU 0 through the host thread.  Here every rank owns a
full replica (weights broadcast once, ``dist_utils.broadcast_state``) and the gradients are summed with RCCL all-reduces
that run WHILE the backward pass is still producing the rest:

* parameters are grouped, in reverse registration order (roughly the order their gradients become ready), into a few
  flat fp32 buckets.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is per-link bound and
  wants few, large messages: the default 48 MB bucket keeps FlowNet2C at four collectives per step;
* ``param.grad`` IS a view into its bucket (no copy in, no copy out);
* a post-accumulate-grad hook counts a bucket's parameters; when the last one is in, the bucket's all-reduce is
  launched on a side stream (after an event recorded on the compute stream), and ``finish()`` -- called before the
  optimizer step -- makes the compute stream wait for the side stream and scales by 1 / world.

On CPU (gloo, the tests) the same code runs without streams.
"""
import torch
import torch.distributed as dist


class BucketedGradAllReduce:
    def __init__(self, module, bucket_bytes=48 << 20, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # a group of ONE rank still sends every bucket through the backend (how the 1-GPU boxes execute the RCCL path);
        # only a process without any group skips the collectives
        self.collective = dist.is_initialized()
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        self.device = params[0].device
        self.on_gpu = self.device.type == "cuda"
        # ---- buckets of flat storage, filled in reverse order of registration
        self.buckets, self._bucket_of, self._home = [], {}, {}
        cur, cur_bytes = [], 0
        for p in reversed(params):
            if p.dtype != torch.float32 or p.device != self.device:
                raise ValueError("BucketedGradAllReduce handles fp32 parameters on one device")
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        self.reset()

    def _close(self, plist):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=torch.float32, device=self.device)
        off = 0
        for p in plist:
            view = flat[off:off + p.numel()].view_as(p)
            p.grad = view                                     # the gradient lives in the bucket
            self._bucket_of[p] = len(self.buckets)
            self._home[p] = (view, view.data_ptr())           # cached: the hook runs once per parameter and step
            off += p.numel()
        self.buckets.append({"flat": flat, "params": plist})

    def reset(self):
        """Before a backward pass: every bucket waits for all of its parameters."""
        self._pending = [len(b["params"]) for b in self.buckets]
        self._handles = []

    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()

    def _launch(self, i):
        flat = self.buckets[i]["flat"]
        if not self.collective:
            return
        if self.on_gpu:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            self._stream.wait_event(ready)
            with torch.cuda.stream(self._stream):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append((i, work))

    def _on_grad(self, p):
        i = self._bucket_of[p]
        view, ptr = self._home[p]
        if p.grad.data_ptr() != ptr:                   # something replaced .grad (e.g. set_to_none): re-home it
            view.copy_(p.grad)
            p.grad = view
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def _grad_ptr(self, p):
        """Address of the parameter's home in its bucket."""
        return self._home[p][1]

    def finish(self):
        """After backward, before the optimizer: wait for the collectives, average.  Buckets whose hooks never completed (a
        parameter without a gradient this step) are reduced here."""
        for i, n in enumerate(self._pending):
            if n > 0:
                self._pending[i] = 0
                self._launch(i)
        for _, work in self._handles:
            work.wait()
        if self.on_gpu and self.collective:
            torch.cuda.current_stream(self.device).wait_stream(self._stream)
        if self.world > 1:
            for b in self.buckets:
                b["flat"].mul_(1.0 / self.world)
        self._handles = []

    def remove(self):
        for h in self._hooks:
            h.remove()
