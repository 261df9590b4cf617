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
import contextlib
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
        # overlap = True: a bucket is all-reduced as soon as its last contribution of the step has been enqueued, while
        # backward is still running (from the grad hooks / sink notifications; under HIP-graph capture the same calls
        # are captured into the graph).  overlap = False: finish() issues every bucket after backward.
        self.overlap = True
        # Contributions per bucket and step = autograd accumulations (post-accumulate hooks) + gradient-sink writes that
        # bypass autograd (notify_sink).  The count is LEARNED from the first finished step, during which nothing is
        # launched early; call relearn() when the set of modules taking part in a step changes.
        self._expected: Optional[List[int]] = None
        self._gradptr_bucket = {p.grad.data_ptr(): self._bucket_of[id(p)] for p in self.params}
        self.before_launch = None            # callable(lo, hi): fold deferred partials of the bucket's address range
        # fold_average: finish() leaves the SUM in the buckets; the optimizer kernel multiplies by 1 / world
        # (dp.FusedAdam, mivae_adam_clip_step grad_scale) instead of one div_ launch per bucket here
        self.fold_average = False
        self.compute_stream = None           # torch.cuda.Stream of the step (set by step.TrainStep); None: current stream

    # -- arrivals ---------------------------------------------------------------------------------
    def _on_grad(self, p: torch.nn.Parameter) -> None:
        bi = self._bucket_of[id(p)]
        flat = self.buckets[bi]
        if p.grad.data_ptr() < flat.data_ptr() or p.grad.data_ptr() >= flat.data_ptr() + flat.numel() * flat.element_size():
            # someone replaced .grad (zero_grad(set_to_none=True)): copy back into the bucket view
            raise RuntimeError("GradReducer: parameter .grad was re-allocated; call reducer.zero_grad() "
                               "instead of optimizer.zero_grad(set_to_none=True)")
        self._arrive(bi)

    def notify_sink(self, grad: torch.Tensor) -> None:
        """A kernel that adds into `grad` (a view into a bucket) behind autograd's back has been enqueued."""
        bi = self._gradptr_bucket.get(grad.data_ptr())
        if bi is not None:
            self._arrive(bi)

    def _arrive(self, bi: int) -> None:
        self._arrived[bi] += 1
        if self._launched[bi] and self.overlap:
            if self.world > 1:
                raise RuntimeError("GradReducer: a gradient contribution arrived after its bucket was all-reduced "
                                   "(the modules taking part in the step changed: call reducer.relearn())")
            import sys
            print(f"[GradReducer] late contribution to bucket {bi} (arrived {self._arrived[bi]}, expected "
                  f"{None if self._expected is None else self._expected[bi]}); counting again", file=sys.stderr)
            self._expected = None            # single rank: nothing was exchanged, just count again
            return
        if self.overlap and self._expected is not None and self._arrived[bi] == self._expected[bi]:
            self._launch(bi)

    def relearn(self) -> None:
        self._expected = None

    def reset(self) -> None:
        self._arrived = [0] * len(self.buckets)
        self._handles = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

    def _launch(self, bi: int) -> None:
        if self._launched[bi]:
            return
        self._launched[bi] = True
        # A grad hook runs on the stream of its AccumulateGrad node (the stream the parameter was created on), which need
        # not be the stream the step's kernels -- the sink kernels in particular -- are enqueued on.  Everything issued
        # here goes to the step's compute stream, after joining the hook's stream; under HIP-graph capture the two are
        # separate branches of the graph and nothing else would order a bucket's fold after the partials it reads.
        cs = self.compute_stream
        ctx = contextlib.nullcontext()
        if cs is not None:
            cur = torch.cuda.current_stream(cs.device)
            if cur != cs:
                cs.wait_stream(cur)
                ctx = torch.cuda.stream(cs)
        with ctx:
            if self.before_launch is not None:
                flat = self.buckets[bi]
                self.before_launch(flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size())
            if self.world > 1:
                self._handles[bi] = dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group,
                                                    async_op=True)

    # -- step API ---------------------------------------------------------------------------------
    def finish(self) -> None:
        """Call after loss.backward(): flushes incomplete buckets, waits, averages (unless fold_average)."""
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for bi, h in enumerate(self._handles):
            if h is not None:
                h.wait()
            if self.world > 1 and not self.fold_average:
                self.buckets[bi].div_(self.world)
        if self._expected is None:
            self._expected = list(self._arrived)
        self.reset()

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
                 weight_decay: float = 0.0, amsgrad: bool = False, bf16_shadow: bool = False,
                 model: Optional[torch.nn.Module] = None):
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
        # (no per-layer weight casts).  Pass `model` to have model.load_state_dict() refresh the shadow by itself; after
        # writing weights by any other route call refresh_shadow().
        self.shadow = None
        if bf16_shadow:
            self.shadow = [torch.empty(b.numel(), dtype=torch.bfloat16, device=dev) for b in reducer.buckets]
            for p in reducer.params:
                bi = reducer._bucket_of[id(p)]
                off = (p.data_ptr() - reducer.flat_params[bi].data_ptr()) // 4
                p._mivae_lp = self.shadow[bi][off:off + p.numel()].view_as(p)
            self.refresh_shadow()
            if model is not None:        # weights loaded later (checkpoint restore) must reach the shadow too
                model.register_load_state_dict_post_hook(lambda module, incompatible: self.refresh_shadow())

    def refresh_shadow(self) -> None:
        if self.shadow is not None:
            with torch.no_grad():
                for sh, fp in zip(self.shadow, self.reducer.flat_params):
                    sh.copy_(fp)

    def step(self, max_norm: Optional[float] = None, max_value: Optional[float] = None, bump: bool = True) -> torch.Tensor:
        """Clip (global L2 norm, then value), Adam update, zero the gradient buckets.  Returns the pre-clip
        gradient norm as a device scalar."""
        L = self._L
        st = L.stream()
        # a reducer that leaves the rank SUM in the buckets (fold_average) gets its averaging here
        gscale = 1.0 / self.reducer.world if (self.reducer.fold_average and self.reducer.world > 1) else 1.0
        off = 0
        for b, nb in zip(self.reducer.buckets, self._nblk):
            L.call("mivae_sumsq_partial", L.ptr(b), self._partial.data_ptr() + 4 * off, b.numel(), st)
            off += nb
        for i, b in enumerate(self.reducer.buckets):
            L.call("mivae_adam_clip_step", L.ptr(self.reducer.flat_params[i]),
                   None if self.shadow is None else L.ptr(self.shadow[i]), L.ptr(b), L.ptr(self.exp_avg[i]),
                   L.ptr(self.exp_avg_sq[i]), L.ptr(self._partial), off, L.ptr(self.step_count),
                   L.ptr(self.grad_norm) if i == 0 else None, b.numel(), self.lr, self.betas[0], self.betas[1],
                   self.eps, self.weight_decay, float(max_norm or 0.0), float(max_value or 0.0), gscale, st)
        if bump:                   # bump=False: the caller advances step_count itself (TrainStep folds it into another launch)
            L.call("mivae_scalar_add", L.ptr(self.step_count), 1.0, st)
        from . import ops
        ops.params_changed()       # raw-pointer writes: torch's version counters did not move
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
        self.refresh_shadow()            # checkpoints are usually loaded together with new weights


class PackedScalars:
    """One all-reduce for every per-step global scalar (replaces ddp.py:25-26 + metrics.py:27-28).  Device scalars are
    gathered with one stack + one index_copy_; host numbers sit in a device-side template that is re-uploaded only
    when they change (synchronously, from pageable memory: a pinned staging buffer reused every step could be
    overwritten while an earlier step's copy is still queued), so a steady-state step issues no host-to-device copy."""

    def __init__(self, names: List[str], device, group=None):
        self.names = list(names)
        self.group = group
        device = torch.device(device)
        self.buf = torch.zeros(len(self.names) + 1, dtype=torch.float32, device=device)
        self._tmpl = torch.zeros_like(self.buf)
        self._tmpl_host: Optional[tuple] = None
        self._buf_is_tmpl = False                # buf's host slots currently equal the template's (single rank only)
        self._idx_cache: Dict[tuple, torch.Tensor] = {}

    def reduce(self, values: Dict[str, torch.Tensor], count: float = 1.0, async_op: bool = False):
        tens_i, tens_v, host = [], [], []
        for i, n in enumerate(self.names):
            v = values[n]
            if isinstance(v, torch.Tensor):
                tens_i.append(i)
                tens_v.append(v.detach().float().reshape(()))
                host.append(0.0)
            else:
                host.append(float(v))
        host.append(float(count))
        host = tuple(host)
        fresh = host != self._tmpl_host
        if fresh:
            self._tmpl.copy_(torch.tensor(host, dtype=torch.float32))
            self._tmpl_host = host
        multi = world_size(self.group) > 1
        # device scalars that sit side by side in one buffer (TrainStep's {batch-loss, batch-kl, batch-nll}) and fill the
        # leading slots travel as ONE strided copy; on a single rank the host slots are rewritten only when they change
        # (no all-reduce touches the buffer), so a steady-state step issues one launch here
        k = len(tens_v)
        packed = (k > 0 and tens_i == list(range(k)) and all(t.dim() == 0 and t.dtype == torch.float32 for t in tens_v)
                  and tens_v[0].device == self.buf.device
                  and all(t.untyped_storage().data_ptr() == tens_v[0].untyped_storage().data_ptr()
                          and t.storage_offset() == tens_v[0].storage_offset() + i for i, t in enumerate(tens_v)))
        if packed:
            if multi or fresh or not self._buf_is_tmpl:
                self.buf.copy_(self._tmpl)
            self.buf[:k].copy_(torch.as_strided(tens_v[0], (k,), (1,)))
            self._buf_is_tmpl = not multi
        else:
            self.buf.copy_(self._tmpl)
            self._buf_is_tmpl = False
            if tens_i:
                key = tuple(tens_i)
                idx = self._idx_cache.get(key)
                if idx is None:
                    idx = self._idx_cache[key] = torch.tensor(tens_i, dtype=torch.int64, device=self.buf.device)
                self.buf.index_copy_(0, idx, torch.stack(tens_v).to(self.buf.device))
        if world_size(self.group) > 1:
            return dist.all_reduce(self.buf, group=self.group, async_op=async_op)
        return None

    def means(self) -> Dict[str, float]:
        host = self.buf.tolist()                 # the only device->host sync
        cnt = max(host[-1], 1e-30)
        return {n: v / cnt for n, v in zip(self.names, host[:-1])}


def spawn_local_ranks(argv: List[str], nranks: int, extra_env: Optional[Dict[str, str]] = None) -> int:
    """Start `nranks` copies of `argv` on this node, one per GPU, with the torch.distributed environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) the reference's launcher sets up with mp.spawn
    (training/ddp.py:14-15, train_worker.py).  Must be called BEFORE the calling process touches the GPU: the parent
    only waits.  When a rank fails the others are terminated (by PID).  Returns the largest exit code."""
    import os
    import socket
    import subprocess
    import time
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen(list(argv), env=env))
    rc = 0
    alive = set(range(nranks))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            rc = max(rc, abs(code))
            if code != 0:                        # a dead rank would leave the others waiting in a collective
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc
