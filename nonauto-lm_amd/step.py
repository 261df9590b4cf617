"""One training step (the reference's `DefaultTrainer._train_batch`, training/trainers/trainer.py:277-303)
as two replayable HIP graphs around the gradient all-reduce.

The step is ~700 small kernels at the C2 shape and host-launch bound in eager mode, so the fixed-shape
step is captured once (torch.cuda.graph = hipGraph on ROCm) and replayed:

    graph A: forward ELBO (autocast) + backward into the flat gradient buckets
    [world > 1: RCCL all-reduce of the flat buckets -- a handful of large messages, not captured]
    graph B: average, clip (norm 5 / value 2), Adam, zero the buckets, pack the step's scalars

In eager mode (graph=False) the all-reduce is issued from the grad hooks and overlaps backward
(dp.GradReducer); under replay Python hooks do not run, and the few-MB all-reduce sits between the
two graphs instead.
"""
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import dp


class TrainStep:
    def __init__(self, model, reducer: dp.GradReducer, optimizer, tokens: torch.Tensor, amp_dtype=None,
                 max_norm: float = 5.0, max_value: float = 2.0, graph: bool = True, warmup: int = 3):
        self.model, self.reducer, self.opt = model, reducer, optimizer
        self.tokens = tokens                      # static input buffer: copy_ new batches into it
        self.amp_dtype = amp_dtype
        self.max_norm, self.max_value = max_norm, max_value
        self.world = dp.world_size()
        self.info: Optional[Dict[str, torch.Tensor]] = None
        self.graph_a = self.graph_b = None
        self.grad_sink = True
        if graph:
            self._capture(warmup)

    # ---- pieces ---------------------------------------------------------------------------------
    def _fwd_bwd(self):
        from . import ops
        ops.GRAD_SINK = self.grad_sink       # weight-gradient kernels add straight into the flat buckets
        if self.grad_sink:
            ops.begin_deferred_sinks()       # ... the small-Linear partials in one launch at the end of backward
        try:
            info = self._fwd_bwd_inner()
            ops.flush_grad_sinks()
            return info
        finally:
            ops.GRAD_SINK = False
            ops._DEFER = None

    def _fwd_bwd_inner(self):
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            out = self.model(self.tokens, self.tokens)
        info = out["loss_info"]
        info["batch-loss"].backward()
        return {k: info[k] for k in ("batch-loss", "batch-kl", "batch-nll")}

    def _update(self):
        if isinstance(self.opt, dp.FusedAdam):      # clip + Adam + zero_grad in one pass over the flat buckets
            self.opt.step(self.max_norm, self.max_value)
            return
        self.reducer.clip_(self.max_norm, self.max_value)
        self.opt.step()
        self.reducer.zero_grad()

    def _capture(self, warmup: int):
        self.reducer.overlap = False
        self.model.track_metrics = False
        # the AccumulateGrad nodes (kept alive by the reducer's hooks) were created on the default stream while
        # capture runs on torch's capture stream: intentional, the buckets are only touched inside the graphs
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):             # warm-up on a side stream (allocator, lazy init, autotune)
            for _ in range(warmup):
                self._fwd_bwd()
                self.reducer.finish()
                self._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # thread_local: RCCL's watchdog thread may query events while this thread captures
        try:
            graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_a, capture_error_mode="thread_local"):
                self.info = self._fwd_bwd()
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b, pool=graph_a.pool(), capture_error_mode="thread_local"):
                self._update()
            self.graph_a, self.graph_b = graph_a, graph_b
        except RuntimeError as err:                 # keep training: same step, launched eagerly
            import sys
            print(f"[TrainStep] HIP-graph capture failed ({err}); falling back to eager launches", file=sys.stderr)
            torch.cuda.synchronize()
            self.graph_a = self.graph_b = None
            self.reducer.overlap = True
            self.reducer.zero_grad()

    # ---- one step -------------------------------------------------------------------------------
    def __call__(self) -> Dict[str, torch.Tensor]:
        if self.graph_a is None:
            self.info = self._fwd_bwd()
            self.reducer.finish()
            self._update()
            return self.info
        self.graph_a.replay()
        if self.world > 1:
            self.reducer.finish()                 # bucket all-reduces + averaging, outside the graphs
        self.graph_b.replay()
        return self.info
