"""Optimizer tail of the training step on the GPU: global-norm gradient clipping and AdamW.

Drop-ins for the two calls the reference trainers make after ``loss.backward()``
(enhanced_training_multi_loss.py:827-829, swin_unet_model.py:836-842):

    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)   ->  clip_grad_norm_(...)
    torch.optim.AdamW(model.parameters(), lr=..., weight_decay=...)     ->  FusedAdamW(...)

Both walk the whole parameter list in one kernel launch each (device arrays of pointers and a
chunk table), instead of one small launch per tensor.  ``FusedAdamW(max_grad_norm=1.0)`` fuses
the two: the norm kernel leaves the clip coefficient on the device and the update applies it
while it reads the gradients.  State keys (``step``, ``exp_avg``, ``exp_avg_sq``) are torch's, so
optimizer checkpoints interchange (``state_dict()`` writes the device-resident step count of the capturable
mode into every ``step`` entry; ``load_state_dict()`` seeds the device counter from the loaded one).  GPU tensors only.
"""
from typing import Iterable, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream


class _TensorList:
    """Device-side description of a list of fp32 tensors: pointer array, sizes, chunk table."""

    def __init__(self, tensors: List[torch.Tensor]):
        if not tensors:
            raise ValueError("empty tensor list")
        dev = tensors[0].device
        for t in tensors:
            if t.device != dev or t.dtype != torch.float32 or not t.is_cuda:
                raise RuntimeError("fused optimizer kernels take float32 CUDA tensors on one device")
            if not t.is_contiguous():
                raise RuntimeError("fused optimizer kernels take contiguous tensors")
        self.device = dev
        self.sizes_h = [t.numel() for t in tensors]
        chunk = _lib.load().unetk_optim_chunk_elems()
        table = [(i, c) for i, n in enumerate(self.sizes_h) for c in range((n + chunk - 1) // chunk)]
        self.nchunks = len(table)
        self.sizes = torch.tensor(self.sizes_h, dtype=torch.int64, device=dev)
        self.chunks = torch.tensor(np.asarray(table, dtype=np.int32).reshape(-1), dtype=torch.int32, device=dev)
        self.partials = torch.empty(self.nchunks, dtype=torch.float32, device=dev)

        self._tables = {}

    def pointers(self, tensors: List[torch.Tensor], slot: str = "") -> torch.Tensor:
        """Device array of the tensors' addresses.  The last table of each slot is kept (parameters
        and moments never move; re-allocated gradients usually come back at the same addresses);
        a changed table goes up from pinned memory without blocking the host, so the launch queue
        never drains.

        A table uploaded while a HIP graph is being captured (gradients re-allocated in the graph's pool)
        becomes a copy node that reads the pinned host block again on every replay: such tables are kept for
        the life of the process (``_CAPTURED``), so that a later eager step, which replaces the slot, cannot hand
        the block back to the allocator under a graph that still reads it."""
        key = tuple(t.data_ptr() for t in tensors)
        hit = self._tables.get(slot)
        if hit is None or hit[0] != key:
            host = torch.tensor(key, dtype=torch.int64).pin_memory()
            hit = (key, host.to(self.device, non_blocking=True), host)
            self._tables[slot] = hit
            if torch.cuda.is_current_stream_capturing():
                _CAPTURED.append(hit)
        return hit[1]


_CAPTURED: list = []        # (key, device table, pinned host block) of every upload recorded into a HIP graph


def _grads_of(params: List[torch.Tensor]) -> List[torch.Tensor]:
    return [p.grad for p in params]


def clip_grad_norm_(parameters: Iterable[torch.Tensor], max_norm: float, norm_type: float = 2.0) -> torch.Tensor:
    """In-place global-norm clip; returns the total norm (0-dim tensor on the device)."""
    if float(norm_type) != 2.0:
        raise ValueError("only the L2 norm is implemented (the reference uses the default)")
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return torch.zeros(())
    grads = _grads_of(params)
    tl = _TensorList(grads)
    gp = tl.pointers(grads, "g")
    out = torch.empty(2, dtype=torch.float32, device=tl.device)
    call("unetk_grad_norm", ptr(gp), ptr(tl.sizes), ptr(tl.chunks), tl.nchunks, float(max_norm), ptr(tl.partials),
         ptr(out), stream())
    call("unetk_scale_list", ptr(gp), ptr(tl.sizes), ptr(tl.chunks), tl.nchunks, ptr(out[1:]), stream())
    return out[0]


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction), one launch per group.

    ``max_grad_norm`` (optional) fuses ``clip_grad_norm_`` over ALL parameters of the optimizer
    into the step; the norm of the last step is kept in ``last_grad_norm`` (device tensor).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_grad_norm: Optional[float] = None, capturable: bool = False):
        """``capturable``: the step number and the learning rate are kept on the device, so that a step
        captured into a HIP graph (``unet_amd.graph.GraphedTrainStep``) replays correctly; a learning rate
        changed by a scheduler reaches the device through ``sync_hyper()``."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm = None
        self.capturable = capturable
        self._lists = {}
        self._fast = {}         # group index -> what step() keeps between calls (see there)
        self._hyper = {}        # group index -> (hyper double[5] on the device, dyn float[3], last pushed (lr, wd, betas))

    # ------------------------------------------------------------------ device-resident step / lr (capturable)
    def _hyper_for(self, gi, group, device):
        hit = self._hyper.get(gi)
        if hit is None:
            b1, b2 = group["betas"]
            step0 = 0.0
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    step0 = float(st["step"])
                    break
            hyper = torch.tensor([step0, group["lr"], group["weight_decay"], b1, b2], dtype=torch.float64, device=device)
            dyn = torch.zeros(3, dtype=torch.float32, device=device)
            hit = self._hyper[gi] = [hyper, dyn, (group["lr"], group["weight_decay"], b1, b2)]
        return hit

    def sync_hyper(self):
        """Pushes learning rates / decays changed on the host (a scheduler) to the device copies."""
        for gi, group in enumerate(self.param_groups):
            hit = self._hyper.get(gi)
            if hit is None:
                continue
            now = (group["lr"], group["weight_decay"], group["betas"][0], group["betas"][1])
            if now != hit[2]:
                hit[0][1:].copy_(torch.tensor(now, dtype=torch.float64), non_blocking=False)
                hit[2] = now

    # ------------------------------------------------------------------ checkpoints
    def _pull_device_steps(self):
        """Capturable mode: graph replays advance only the device counter; bring the host-side ``step`` entries
        (what torch's state_dict() saves) up to it."""
        for gi, group in enumerate(self.param_groups):
            hit = self._hyper.get(gi)
            if hit is None:
                continue
            n = float(hit[0][0].item())
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"].fill_(n)

    def state_dict(self):
        self._pull_device_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._fast.clear()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            for p in group["params"]:
                st = self.state.get(p)
                if st and torch.is_tensor(st.get("step")) and st["step"].is_cuda:
                    st["step"] = st["step"].detach().cpu().to(torch.float32)      # host-side counter, as step() makes it
            hit = self._hyper.get(gi)
            if hit is not None:         # the same device tensors a captured graph reads: updated in place
                step0 = next((float(self.state[p]["step"]) for p in group["params"] if self.state.get(p)), 0.0)
                now = (group["lr"], group["weight_decay"], b1, b2)
                hit[0].copy_(torch.tensor((step0,) + now, dtype=torch.float64))
                hit[2] = now

    def snapshot(self):
        """Copies of the moments and step counts, for ``restore`` (a graph capture's warm-up steps are undone).
        In capturable mode the device-resident counters are created here, from the state's step counts (zero, or
        what a checkpoint brought), so that they are part of the snapshot."""
        if self.capturable:
            for gi, group in enumerate(self.param_groups):
                if group["params"]:
                    self._hyper_for(gi, group, group["params"][0].device)
        snap = {}
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    snap[p] = {k: v.clone() for k, v in st.items()}
        return snap, {gi: h[0][0:1].clone() for gi, h in self._hyper.items()}

    @torch.no_grad()
    def restore(self, snap):
        """Puts the optimizer state back in place (same tensors, old contents; state that did not exist at the
        snapshot returns to zero)."""
        states, steps = snap
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                old = states.get(p)
                for k, v in st.items():
                    if old is not None:
                        v.copy_(old[k])
                    else:
                        v.zero_()
        for gi, h in self._hyper.items():
            if gi in steps:
                h[0][0:1].copy_(steps[gi])
            else:       # a counter made after the snapshot: back to the snapshot's step count
                old = next((states[p] for p in self.param_groups[gi]["params"] if p in states), None)
                h[0][0:1].fill_(float(old["step"]) if old is not None else 0.0)

    def device_step_count(self, gi=0):
        hit = self._hyper.get(gi)
        return None if hit is None else int(hit[0][0].item())

    def _list_for(self, key, tensors):
        sig = tuple(t.numel() for t in tensors)
        hit = self._lists.get(key)
        if hit is None or hit[0] != sig:
            hit = (sig, _TensorList(tensors))
            self._lists[key] = hit
        return hit[1]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        coef = None
        if self.max_grad_norm is not None:
            allp = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            if allp:
                grads = _grads_of(allp)
                tl = self._list_for("norm", grads)
                out = torch.empty(2, dtype=torch.float32, device=tl.device)
                gp = tl.pointers(grads, "g")
                call("unetk_grad_norm", ptr(gp), ptr(tl.sizes), ptr(tl.chunks), tl.nchunks,
                     float(self.max_grad_norm), ptr(tl.partials), ptr(out), stream())
                self.last_grad_norm = out[0]
                coef = out[1:]
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            fast = self._fast.get(gi)
            if fast is not None and fast[0] == len(params) and fast[1] is params[0] and fast[2] is params[-1] and \
                    self.state[params[-1]].get("step") is fast[3] and \
                    fast[8] == (params[0].data_ptr(), params[-1].data_ptr()):
                # the same parameters as last step (the usual case): ONE step counter shared by the group's states,
                # the address tables of parameters and moments kept (they never move)
                _, _, _, step_t, tl, pp, mp, vp, _ = fast
                step = int(step_t) + 1
                step_t += 1
            else:
                for p in params:
                    st = self.state[p]
                    if not st:
                        st["step"] = torch.zeros((), dtype=torch.float32)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                steps = {int(self.state[p]["step"]) for p in params}
                if len(steps) != 1:
                    raise RuntimeError("parameters of one group must share their step count")
                step = steps.pop() + 1
                step_t = self.state[params[0]]["step"]
                step_t += 1
                for p in params[1:]:         # one counter object for the whole group from now on
                    self.state[p]["step"] = step_t
                tl = self._list_for(gi, params)
                pp = tl.pointers(params, "p")
                mp = tl.pointers([self.state[p]["exp_avg"] for p in params], "m")
                vp = tl.pointers([self.state[p]["exp_avg_sq"] for p in params], "v")
                self._fast[gi] = (len(params), params[0], params[-1], step_t, tl, pp, mp, vp,
                                  (params[0].data_ptr(), params[-1].data_ptr()))
            b1, b2 = group["betas"]
            gp = tl.pointers(_grads_of(params), "g")
            if self.capturable:
                # (the host-side counters above only count the calls made outside graph replays; the device
                # counter is the truth: device_step_count())
                hyper, dyn, _ = self._hyper_for(gi, group, tl.device)
                call("unetk_adamw_step_device", ptr(pp), ptr(gp), ptr(mp), ptr(vp), ptr(tl.sizes), ptr(tl.chunks),
                     tl.nchunks, ptr(hyper), ptr(dyn), float(group["eps"]), ptr(coef), 1, float(b1), float(b2), stream())
                continue
            call("unetk_adamw_step", ptr(pp), ptr(gp), ptr(mp), ptr(vp), ptr(tl.sizes), ptr(tl.chunks), tl.nchunks, float(group["lr"]), float(b1), float(b2),
                 float(group["eps"]), float(group["weight_decay"]), step, ptr(coef), 1, stream())
        return loss
