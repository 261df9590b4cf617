"""Batch-parallel training over the GPUs of one node: one process per GPU, RCCL (torch.distributed
backend "nccl" on ROCm) over xGMI.

Replaces the reference's DistributedDataParallel wrap (training/trainers/trainer.py:51-60,
find_unused_parameters=True), its per-batch / per-epoch scalar all-reduces (training/ddp.py:18-47)
and the 2-per-metric scalar all-reduces (training/metrics.py:27-28) with:

 * GradReducer -- gradients live as views into a few large flat buckets laid out in reverse
   registration order (= the order backward produces them: decoder -> flow stack -> posterior ->
   encoder).  A post-accumulate hook counts arrivals; when a bucket is complete its all-reduce is
   issued asynchronously from the autograd thread, so RCCL's stream runs under the rest of backward.
   Buckets default to 32 MiB: xGMI is point-to-point (7 links x ~153 GB/s), a ring is bound by one
   link, so few large messages beat many 25 MB DDP buckets for the <= 70 MB of gradients here.
   Parameters that received no gradient simply contribute zeros (no graph walk needed).
 * PackedScalars -- every per-step scalar that must be global (done-early flag, metric sums, count)
   travels in ONE small all-reduce.

The path has no other exchange step: every op is per-sentence.
"""
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Same initial weights on every rank (the reference relies on identical seeds, train_worker.py:48)."""
    if world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=group)


class GradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20, group=None,
                 flatten_params: bool = False):
        """flatten_params: also move the parameters themselves into flat buffers with the buckets' layout
        (p.data becomes a view; names, shapes and state-dict keys are unchanged), so that FusedAdam can update
        a whole bucket in one kernel.  Call after the model is on its device; do not `.to()` it afterwards."""
        self.group = group
        self.world = world_size(group)
        self.params = [p for p in params if p.requires_grad]
        self.buckets: List[torch.Tensor] = []
        self.flat_params: List[Optional[torch.Tensor]] = []
        self._bucket_of: Dict[int, int] = {}
        self._bucket_size: List[int] = []
        order = list(reversed(self.params))
        cur: List[torch.nn.Parameter] = []
        cur_bytes = 0
        groups: List[List[torch.nn.Parameter]] = []
        for p in order:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for bi, ps in enumerate(groups):
            # 64-element alignment keeps every view 256-byte aligned for fp32
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += (p.numel() + 63) // 64 * 64
            flat = torch.zeros(total, dtype=ps[0].dtype, device=ps[0].device)
            pflat = None
            if flatten_params:
                pflat = torch.zeros(total, dtype=ps[0].dtype, device=ps[0].device)
                with torch.no_grad():
                    for p, o in zip(ps, offs):
                        view = pflat[o:o + p.numel()].view_as(p)
                        view.copy_(p.data)
                        p.data = view
            self.flat_params.append(pflat)
            for p, o in zip(ps, offs):
                p.grad = flat[o:o + p.numel()].view_as(p)
                self._bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._on_grad)
            self.buckets.append(flat)
            self._bucket_size.append(len(ps))
        self._arrived = [0] * len(self.buckets)
        self._handles: List[Optional[object]] = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        # overlap = True: buckets are all-reduced from the grad hooks while backward is still running.
        # overlap = False (HIP-graph replay of the step, where Python hooks do not run): finish() issues
        # every bucket after backward.
        self.overlap = True

    # -- hooks ------------------------------------------------------------------------------------
    def _on_grad(self, p: torch.nn.Parameter) -> None:
        bi = self._bucket_of[id(p)]
        flat = self.buckets[bi]
        if p.grad.data_ptr() < flat.data_ptr() or p.grad.data_ptr() >= flat.data_ptr() + flat.numel() * flat.element_size():
            # someone replaced .grad (zero_grad(set_to_none=True)): copy back into the bucket view
            raise RuntimeError("GradReducer: parameter .grad was re-allocated; call reducer.zero_grad() "
                               "instead of optimizer.zero_grad(set_to_none=True)")
        if not self.overlap:
            return
        self._arrived[bi] += 1
        if self._arrived[bi] == self._bucket_size[bi]:
            self._launch(bi)

    def _launch(self, bi: int) -> None:
        if self._launched[bi]:
            return
        self._launched[bi] = True
        if self.world > 1:
            self._handles[bi] = dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True)

    # -- step API ---------------------------------------------------------------------------------
    def finish(self) -> None:
        """Call after loss.backward(): flushes incomplete buckets, waits, averages."""
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for bi, h in enumerate(self._handles):
            if h is not None:
                h.wait()
            if self.world > 1:
                self.buckets[bi].div_(self.world)
        self._arrived = [0] * len(self.buckets)
        self._handles = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

    def zero_grad(self) -> None:
        for flat in self.buckets:
            flat.zero_()

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm over all buckets (identical on every rank after finish())."""
        return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(b.float()) for b in self.buckets]))

    def clip_(self, max_norm: Optional[float], max_value: Optional[float]) -> torch.Tensor:
        """clip_grad_norm_(max_norm) then clip_grad_value_(max_value) on the flat buckets
        (training/trainers/trainer.py:286-289) -- two passes over a few tensors instead of a
        multi-tensor walk over every parameter."""
        norm = self.grad_norm()
        if max_norm is not None:
            coef = (max_norm / (norm + 1e-6)).clamp(max=1.0)
            for b in self.buckets:
                b.mul_(coef.to(b.dtype))
        if max_value is not None:
            for b in self.buckets:
                b.clamp_(-max_value, max_value)
        return norm


class FusedAdam:
    """torch.optim.Adam (the reference's `Optimizer: "adam"`, nn/optimizer.py:11-33) + the two gradient clips
    + zero_grad of DefaultTrainer._train_batch (trainer.py:286-289, 359-367) over the reducer's flat buckets:
    two kernel launches per bucket (libmivae `mivae_sumsq_partial`, `mivae_adam_clip_step`) instead of four
    multi-tensor walks over every parameter.  The step counter lives on the device, so the whole update can sit
    inside a HIP graph.  Needs GradReducer(..., flatten_params=True) and fp32 parameters."""

    def __init__(self, reducer: "GradReducer", lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, amsgrad: bool = False, bf16_shadow: bool = False):
        from . import _lib as L
        if amsgrad:
            raise NotImplementedError("FusedAdam: amsgrad is not supported")
        if any(fp is None for fp in reducer.flat_params):
            raise ValueError("FusedAdam needs GradReducer(..., flatten_params=True)")
        if any(b.dtype != torch.float32 for b in reducer.buckets):
            raise ValueError("FusedAdam: fp32 parameters only")
        L.require_gpu(*reducer.buckets)
        self._L = L
        self.reducer = reducer
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        dev = reducer.buckets[0].device
        self.exp_avg = [torch.zeros_like(b) for b in reducer.buckets]
        self.exp_avg_sq = [torch.zeros_like(b) for b in reducer.buckets]
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._nblk = [L.query("mivae_sumsq_blocks", b.numel()) for b in reducer.buckets]
        self._partial = torch.zeros(max(sum(self._nblk), 1), dtype=torch.float32, device=dev)
        # bf16_shadow: the update kernel also writes a bf16 copy of every parameter; each parameter gets a
        # `_mivae_lp` view into it, which nn_bracket.Linear uses as its compute copy under bf16 autocast
        # (no per-layer weight casts).  After loading weights by any other route call refresh_shadow().
        self.shadow = None
        if bf16_shadow:
            self.shadow = [torch.empty(b.numel(), dtype=torch.bfloat16, device=dev) for b in reducer.buckets]
            for p in reducer.params:
                bi = reducer._bucket_of[id(p)]
                off = (p.data_ptr() - reducer.flat_params[bi].data_ptr()) // 4
                p._mivae_lp = self.shadow[bi][off:off + p.numel()].view_as(p)
            self.refresh_shadow()

    def refresh_shadow(self) -> None:
        if self.shadow is not None:
            with torch.no_grad():
                for sh, fp in zip(self.shadow, self.reducer.flat_params):
                    sh.copy_(fp)

    def step(self, max_norm: Optional[float] = None, max_value: Optional[float] = None) -> torch.Tensor:
        """Clip (global L2 norm, then value), Adam update, zero the gradient buckets.  Returns the pre-clip
        gradient norm as a device scalar."""
        L = self._L
        st = L.stream()
        off = 0
        for b, nb in zip(self.reducer.buckets, self._nblk):
            L.call("mivae_sumsq_partial", L.ptr(b), self._partial.data_ptr() + 4 * off, b.numel(), st)
            off += nb
        for i, b in enumerate(self.reducer.buckets):
            L.call("mivae_adam_clip_step", L.ptr(self.reducer.flat_params[i]),
                   None if self.shadow is None else L.ptr(self.shadow[i]), L.ptr(b), L.ptr(self.exp_avg[i]),
                   L.ptr(self.exp_avg_sq[i]), L.ptr(self._partial), off, L.ptr(self.step_count),
                   L.ptr(self.grad_norm) if i == 0 else None, b.numel(), self.lr, self.betas[0], self.betas[1],
                   self.eps, self.weight_decay, float(max_norm or 0.0), float(max_value or 0.0), st)
        L.call("mivae_scalar_add", L.ptr(self.step_count), 1.0, st)
        return self.grad_norm

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.reducer.zero_grad()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd) -> None:
        self.step_count.copy_(sd["step"])
        for dst, src in zip(self.exp_avg, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            dst.copy_(src)


class PackedScalars:
    """One all-reduce for every per-step global scalar (replaces ddp.py:25-26 + metrics.py:27-28)."""

    def __init__(self, names: List[str], device, group=None):
        self.names = list(names)
        self.group = group
        self.buf = torch.zeros(len(self.names) + 1, dtype=torch.float32, device=device)

    def reduce(self, values: Dict[str, torch.Tensor], count: float = 1.0, async_op: bool = False):
        vals = [values[n].detach().float().reshape(()) if isinstance(values[n], torch.Tensor)
                else torch.tensor(float(values[n]), device=self.buf.device) for n in self.names]
        self.buf.copy_(torch.stack(vals + [torch.tensor(float(count), device=self.buf.device)]))
        if world_size(self.group) > 1:
            return dist.all_reduce(self.buf, group=self.group, async_op=async_op)
        return None

    def means(self) -> Dict[str, float]:
        host = self.buf.tolist()                 # the only device->host sync
        cnt = max(host[-1], 1e-30)
        return {n: v / cnt for n, v in zip(self.names, host[:-1])}
