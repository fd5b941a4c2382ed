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

A model may produce gradients on more than one HIP stream (the Swin-UNet backward runs its weight gradients on a
second stream).  Each gradient is copied into its bucket on the stream that produced it; a bucket filled from
several streams sends its all-reduce from one that is NOT the main stream (the stream of the step's first gradient),
behind events of the others, so the main data-gradient chain never waits for the weight gradients.  Gradients a
model completes only at the end of its backward (``module.late_grad_names()``: sums it takes with one launch at the
end) get a bucket of their own at the end of the list, so that they do not hold the others back.

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
    __slots__ = ("flat", "names", "views", "pending", "work", "streams", "done", "incoming")

    def __init__(self):
        self.flat: Optional[torch.Tensor] = None
        self.names: List[str] = []
        self.views: Dict[str, torch.Tensor] = {}
        self.pending = 0
        self.work = None
        self.streams: Dict[int, "torch.cuda.Stream"] = {}   # the HIP streams whose copies filled the bucket this step
        self.done = None                                     # event behind the collective's launch, when it left another stream
        self.incoming: Dict[str, torch.Tensor] = {}          # gradients handed to the sink, copied in together at launch


class _Done:
    def wait(self):
        return True


_DONE = _Done()


class GradReducer:
    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 32 << 20, process_group=None, collective=None,
                 force: bool = False):
        """``collective(flat)``: starts the sum over the ranks of one flat bucket in place and returns an
        object with ``wait()`` (or None when it has already finished).  Default: an asynchronous
        ``dist.all_reduce`` (RCCL: enqueued on the communicator's stream behind the kernels queued so far).
        ``force``: install the sink / hooks with a single rank too (timing the bucket path on one GPU with a
        collective that does nothing)."""
        self.module = module
        self.group = process_group
        self._collective = collective if collective is not None else (
            lambda flat: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.params = dict(named)
        late = set(module.late_grad_names()) if hasattr(module, "late_grad_names") else set()
        # gradients appear in reverse registration order: bucket 0 holds the last layers; the late ones come last
        self.buckets: List[_Bucket] = []
        layout = []
        for group in ([(n, p) for n, p in reversed(named) if n not in late],
                      [(n, p) for n, p in reversed(named) if n in late]):
            cur, size = _Bucket(), 0
            for n, p in group:
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
        self._main = None           # the HIP stream of the step's first gradient
        self._hooks = []
        self._active = self.world > 1 or force
        if hasattr(module, "grad_sink"):
            module.grad_sink = self._sink if self._active else None
        elif self._active:
            for n, p in named:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(n)))

    # ------------------------------------------------------------------ filling
    def _arm(self):
        for b in self.buckets:
            b.pending = len(b.names)
            b.work = None
            b.streams = {}
            b.done = None
            b.incoming = {}
        dev = self.buckets[0].flat.device if self.buckets else None
        self._main = torch.cuda.current_stream(dev) if (dev is not None and dev.type == "cuda") else None
        self._armed = True

    def _touched(self, b: _Bucket):
        if b.flat.is_cuda:
            st = torch.cuda.current_stream(b.flat.device)
            b.streams[st.cuda_stream] = st

    def _launch(self, b: _Bucket):
        """Starts the bucket's collective behind every copy that filled it.  One stream: the collective is ordered
        after everything queued so far on it.  Several: it leaves from one that is not the main stream, which
        waits for events of the others (the main stream is ahead of the side stream, never the other way round)."""
        streams = list(b.streams.values())
        cur = torch.cuda.current_stream(b.flat.device) if b.flat.is_cuda else None
        if cur is None or not streams or all(s.cuda_stream == cur.cuda_stream for s in streams):
            self._fill(b)
            b.work = self._collective(b.flat) or _DONE
            return
        main_id = self._main.cuda_stream if self._main is not None else cur.cuda_stream
        aux = next((s for s in streams if s.cuda_stream != main_id), cur)
        others = {s.cuda_stream: s for s in streams + [cur]}
        others.pop(aux.cuda_stream, None)
        with torch.cuda.stream(aux):
            for s in others.values():
                ev = torch.cuda.Event()
                ev.record(s)
                aux.wait_event(ev)
            self._fill(b)
            b.work = self._collective(b.flat) or _DONE
            b.done = torch.cuda.Event()
            b.done.record(aux)

    @staticmethod
    def _fill(b: _Bucket):
        """The gradients the sink collected for this bucket, copied into their slots by one multi-tensor launch
        (a copy per gradient costs the host more than the step can spare: ~250 launches per Swin-UNet step)."""
        if b.incoming:
            names = list(b.incoming)
            torch._foreach_copy_([b.views[n] for n in names], [b.incoming[n].reshape(b.views[n].shape) for n in names])
            b.incoming = {}

    def _arrived(self, b: _Bucket):
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

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
        b.incoming[name] = grad.detach()            # copied into its slot when the bucket is complete (_fill)
        self._touched(b)                            # ... behind the stream that produced it
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
            self._touched(b)
            self._arrived(b)
        return hook

    # ------------------------------------------------------------------ after backward
    def finish(self):
        """Wait for the collectives, average, and point every ``.grad`` at its bucket view.  A parameter
        that received no gradient in this step contributes zeros (its bucket is completed here)."""
        if not self._active:
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
                self._touched(b)
                self._launch(b)
        inv = 1.0 / self.world
        for b in self.buckets:
            b.work.wait()
            if b.done is not None:          # launched from another stream: this one continues behind it
                torch.cuda.current_stream(b.flat.device).wait_event(b.done)
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
