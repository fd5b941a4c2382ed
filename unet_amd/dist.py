"""Data-parallel gradient reduction over RCCL (one process per GPU).

The reference trains on a single device; data parallelism is an addition with this contract:
every rank runs the reference step on its own shard (BatchNorm statistics and the TID-weighted
means stay per shard), then gradients are averaged, so an N-rank step equals the mean of the N
per-shard reference gradients.  Gradients are packed into a few large flat fp32 buckets and
all-reduced bucket by bucket: xGMI is point to point (a ring is bound by one ~150 GB/s link), so
few large messages beat many small ones; 70 MB of UNet gradients is two 64 MiB buckets.
"""
from typing import List

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 64 << 20):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in self.params:
            n = p.numel() * 4
            if cur and size + n > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            self.buckets.append(cur)

    def allreduce(self):
        """Average .grad of every parameter over the ranks.  Asynchronous per bucket: the copy-in of
        bucket k+1 overlaps the collective of bucket k.  Afterwards every .grad is a view into its
        bucket's (fresh) flat buffer: one scaling kernel per bucket, no copy back per parameter."""
        if self.world == 1:
            return
        works = []
        for bucket in self.buckets:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1).float() for g in grads])
            works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
        inv = 1.0 / self.world
        for work, flat, bucket in works:
            work.wait()
            flat.mul_(inv)
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
