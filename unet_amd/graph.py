"""A whole training step as one HIP graph.

A Swin-UNet step queues ~600 small kernels; at the reference's batch sizes (6 per step,
swin_unet_model.py:728) the Python side of those launches, not the GPU, sets the step time.
``GraphedTrainStep`` runs the step a few times eagerly, captures it once (forward, loss, backward,
global-norm clip and AdamW, with the side-stream weight gradients forked and joined inside the capture)
and from then on replays the graph: one launch per step.  The warm-up steps are undone (parameters,
buffers and optimizer state return to their values), so constructing the object does not train.
Shapes are fixed by the example batch;
new data are copied into the captured input buffers.  The optimizer must keep its step count and
learning rate on the device (``FusedAdamW(..., capturable=True)``).
"""
from typing import Callable, Optional

import torch

from .losses import _LossParts


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, x, y, w=None,
                 post: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None, warmup: int = 3):
        """``post(outputs, labels)``: applied between the model and the loss (the Swin trainer's bilinear
        resize of the half-resolution prediction to the label size, swin_unet_model.py:812-814)."""
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs an optimizer whose step count lives on the device: "
                             "FusedAdamW(..., capturable=True)")
        if not x.is_cuda:
            raise RuntimeError("GraphedTrainStep runs on the GPU only")
        self.model, self.criterion, self.optimizer, self.post = model, criterion, optimizer, post
        self.x, self.y = x.clone(), y.clone()
        self.w = None if w is None else w.clone()
        tensors = list(model.parameters()) + list(model.buffers())
        saved = [t.detach().clone() for t in tensors]
        opt_saved = optimizer.snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):      # plans, packed-weight tables and pointer tables settle here
                self._eager()
            with torch.no_grad():
                for t, s0 in zip(tensors, saved):
                    t.copy_(s0)
                optimizer.restore(opt_saved)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="relaxed"):
            self.loss, parts = self._eager()
        self._parts_src = parts._pending if isinstance(parts, _LossParts) else None
        self._parts = parts

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(self.x)
        if self.post is not None:
            out = self.post(out, self.y)
        res = self.criterion(out, self.y, self.w) if self.w is not None else self.criterion(out, self.y)
        loss, parts = res if isinstance(res, tuple) else (res, None)
        loss.backward()
        self.optimizer.step()
        return loss, parts

    def __call__(self, x, y, w=None):
        """One step on a new batch of the captured shape.  Returns (loss, parts) as the criterion does; both
        refer to buffers that the next call overwrites."""
        if x.shape != self.x.shape or y.shape != self.y.shape:
            raise ValueError(f"the captured step takes {tuple(self.x.shape)} / {tuple(self.y.shape)} batches")
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        if y.data_ptr() != self.y.data_ptr():
            self.y.copy_(y, non_blocking=True)
        if self.w is not None and w is not None and w.data_ptr() != self.w.data_ptr():
            self.w.copy_(w, non_blocking=True)
        self.optimizer.sync_hyper()
        self.graph.replay()
        parts = _LossParts(*self._parts_src) if self._parts_src is not None else self._parts
        return self.loss, parts
