"""Data-parallel gradient reduction over RCCL (one process per GPU).

The reference trains on a single device; data parallelism is an addition with this contract:
every rank runs the reference step on its own shard (BatchNorm statistics and the TID-weighted
means stay per shard), then gradients are averaged, so an N-rank step equals the mean of the N
per-shard reference gradients.

Gradients live in a few persistent flat fp32 buckets, filled in the order the backward pass
produces them (last layer first).  A bucket's all-reduce is launched the moment its last gradient
has arrived, while the backward pass is still running, so only the final bucket (the first, small
layers) is exposed.  xGMI is point to point (a ring is bound by one ~150 GB/s link), so few large
messages beat many small ones: 70 MB of UNet gradients are three 32 MiB buckets.

Two ways a gradient reaches its bucket:
 * ``unet_amd`` models run their whole backward as one autograd node; they hand every parameter
   gradient to ``model.grad_sink(name, tensor)`` as soon as its kernel is queued.  The reducer
   installs itself as that sink.
 * any other ``nn.Module``: a post-accumulate-grad hook per parameter.
After ``finish()`` every ``.grad`` is a view into its bucket (averaged over the ranks).

One backward pass per ``finish()``: gradient accumulation over several backward passes is not supported (the
second pass raises).  ``zero_grad(set_to_none=False)`` is fine: a ``.grad`` that still aliases its bucket view is
dropped before the new gradient is handed to autograd, which would otherwise add the view to itself.
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("flat", "names", "views", "pending", "work")

    def __init__(self):
        self.flat: Optional[torch.Tensor] = None
        self.names: List[str] = []
        self.views: Dict[str, torch.Tensor] = {}
        self.pending = 0
        self.work = None


class _Done:
    def wait(self):
        return True


_DONE = _Done()


class GradReducer:
    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 32 << 20, process_group=None, collective=None):
        """``collective(flat)``: starts the sum over the ranks of one flat bucket in place and returns an
        object with ``wait()`` (or None when it has already finished).  Default: an asynchronous
        ``dist.all_reduce`` (RCCL: enqueued on the communicator's stream behind the kernels queued so far)."""
        self.module = module
        self.group = process_group
        self._collective = collective if collective is not None else (
            lambda flat: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.params = dict(named)
        # gradients appear in reverse registration order: bucket 0 holds the last layers
        self.buckets: List[_Bucket] = []
        cur, size = _Bucket(), 0
        layout = []
        for n, p in reversed(named):
            nbytes = p.numel() * 4
            if cur.names and size + nbytes > bucket_bytes:
                layout.append((cur, size))
                cur, size = _Bucket(), 0
            cur.names.append(n)
            size += nbytes
        if cur.names:
            layout.append((cur, size))
        self.bucket_of: Dict[str, _Bucket] = {}
        for b, size in layout:
            dev = self.params[b.names[0]].device
            b.flat = torch.zeros(size // 4, dtype=torch.float32, device=dev)
            off = 0
            for n in b.names:
                p = self.params[n]
                b.views[n] = b.flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self.bucket_of[n] = b
            self.buckets.append(b)
        self._armed = False
        self._hooks = []
        if hasattr(module, "grad_sink"):
            module.grad_sink = self._sink if self.world > 1 else None
        elif self.world > 1:
            for n, p in named:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(n)))

    # ------------------------------------------------------------------ filling
    def _arm(self):
        for b in self.buckets:
            b.pending = len(b.names)
            b.work = None
        self._armed = True

    def _arrived(self, b: _Bucket):
        b.pending -= 1
        if b.pending == 0:
            # the collective is ordered after everything queued so far on the current stream
            b.work = self._collective(b.flat) or _DONE

    def _sink(self, name: str, grad: torch.Tensor) -> torch.Tensor:
        """Called from inside a model's backward with a freshly produced gradient; returns the bucket
        view that now holds it (the model hands that view to autograd)."""
        if not self._armed:
            self._arm()
        b = self.bucket_of[name]
        if b.pending <= 0:
            raise RuntimeError(f"GradReducer: a second gradient for '{name}' arrived before finish(); one backward "
                               "pass per finish() (gradient accumulation is not supported)")
        view = b.views[name]
        view.copy_(grad.reshape(view.shape))
        self._arrived(b)
        p = self.params[name]
        if p.grad is not None and p.grad.data_ptr() == view.data_ptr():
            # left over from the last step (zero_grad(set_to_none=False) or none at all): autograd would do
            # ``p.grad += view`` on the same memory, doubling the bucket after its all-reduce was launched
            p.grad = None
        return view.detach()        # a fresh alias of the bucket memory, for autograd to keep as .grad

    def _make_hook(self, name):
        def hook(p):
            if not self._armed:
                self._arm()
            b = self.bucket_of[name]
            b.views[name].copy_(p.grad)
            self._arrived(b)
        return hook

    # ------------------------------------------------------------------ after backward
    def finish(self):
        """Wait for the collectives, average, and point every ``.grad`` at its bucket view.  A parameter
        that received no gradient in this step contributes zeros (its bucket is completed here)."""
        if self.world == 1:
            return
        if not self._armed:
            self._arm()
        for b in self.buckets:
            if b.work is None:
                for n in b.names:
                    if self.params[n].grad is None and b.pending:
                        b.views[n].zero_()
                    elif b.pending and self.params[n].grad is not None and \
                            self.params[n].grad.data_ptr() != b.views[n].data_ptr():
                        b.views[n].copy_(self.params[n].grad)
                b.pending = 0
                b.work = self._collective(b.flat) or _DONE
        inv = 1.0 / self.world
        for b in self.buckets:
            b.work.wait()
            b.flat.mul_(inv)
            for n in b.names:
                self.params[n].grad = b.views[n]
        self._armed = False

    def allreduce(self):
        """Average ``.grad`` over the ranks when nothing was launched during backward (plain call after
        ``loss.backward()``); with the sink or the hooks in place this only waits and averages."""
        self.finish()

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if hasattr(self.module, "grad_sink"):
            self.module.grad_sink = None


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
