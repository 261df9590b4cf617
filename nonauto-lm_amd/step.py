"""One training step (the reference's `DefaultTrainer._train_batch`, training/trainers/trainer.py:277-303)
as two replayable HIP graphs.

The step is ~200 kernels at the C2 shape and host-launch bound in eager mode, so the fixed-shape step is captured
once (torch.cuda.graph = hipGraph on ROCm) and replayed:

    graph A: forward ELBO (autocast) + backward into the flat gradient buckets.  With more than one rank the
             bucket all-reduces (RCCL, SUM) are captured INSIDE this graph: during capture the reducer's grad hooks
             fire exactly as in eager mode and issue each bucket's all-reduce on RCCL's stream as soon as the bucket
             is complete, so on replay the collectives run under the rest of backward (decoder bucket first, while
             the flow stack / posterior / encoder gradients are still being computed).  If RCCL refuses the capture,
             or the captured graph does not leave rank-identical buckets (checked once after capture), the step falls
             back to: graph A without collectives, all-reduce between the graphs.
    graph B: clip (norm 5 / value 2) + Adam + zero the buckets (dp.FusedAdam: the 1 / world averaging is a factor
             inside that kernel, no pass of its own).

Host-side state the capture would otherwise freeze or corrupt:
  * the scheduler weights (nn/weight_scheduler.py:33-60) live in a 2-float device tensor the ELBO kernels read
    (`model.device_weights`); every call writes the current weights there before the replay and then steps the host
    schedulers exactly as VAELmModel.forward does (models/base/model.py:180-184);
  * the warm-up iterations (allocator, lazy init, autotune) and the capture are side-effect free: parameters,
    optimizer state, scheduler state and the RNG offsets are snapshotted before and restored after.  Data-dependent
    initialisation (ActNorm, flows.py) is triggered by one plain forward BEFORE the snapshot, as the reference does it
    on its first training batch;
  * every device table whose address a captured launch bakes in (ops._PTR_CACHE, ops._JOB_TABLES, ops._PART_BUFS)
    is pinned on this object, and the caches stop evicting while any captured step is alive (ops.pin_tables).
"""
import copy
import os
import sys
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import dp, ops


class TrainStep:
    def __init__(self, model, reducer: dp.GradReducer, optimizer, tokens: torch.Tensor, amp_dtype=None,
                 max_norm: float = 5.0, max_value: float = 2.0, graph: bool = True, warmup: int = 3,
                 eps: Optional[torch.Tensor] = None):
        self.model, self.reducer, self.opt = model, reducer, optimizer
        self.tokens = tokens                      # static input buffer: copy_ new batches into it
        self.eps = eps                            # optional static buffer of injected posterior noise (parity runs)
        self.amp_dtype = amp_dtype
        self.max_norm, self.max_value = max_norm, max_value
        self.world = dp.world_size()
        self.info: Optional[Dict[str, torch.Tensor]] = None
        self.graph_a = self.graph_b = None
        self.comm_in_graph = False
        self.grad_sink = True
        self._seed = None
        # device-side metric sums (batch-nll, batch-kl) updated by ONE launch inside graph B; the model's RunningMean
        # accumulators point at them (was: two clones + two adds launched after every replay)
        self._acc = torch.zeros(2, dtype=torch.float32, device=tokens.device) if tokens.is_cuda else None
        self._acc_views = None if self._acc is None else (self._acc[0], self._acc[1])
        self._want_track = False
        self._pinned = None
        # scheduler weights on the device (read by mivae_elbo_fwd / _bwd)
        self._w_dev = torch.zeros(2, dtype=torch.float32, device=tokens.device)
        self._w_host = (0.0, 0.0)                 # what _w_dev holds (uploads happen only when a weight changes)
        if isinstance(optimizer, dp.FusedAdam):   # averaging folded into the update kernel
            reducer.fold_average = True
        if graph:
            self._capture(warmup)

    # ---- pieces ---------------------------------------------------------------------------------
    def _fwd_bwd(self):
        from . import ops
        ops.GRAD_SINK = self.grad_sink       # weight-gradient kernels add straight into the flat buckets
        if self.grad_sink:
            ops.begin_deferred_sinks()       # ... the small-Linear partials in one launch per bucket
            ops.SINK_NOTIFY = self.reducer.notify_sink          # sinks bypass autograd: tell the reducer they arrived
            self.reducer.before_launch = ops.flush_grad_sinks   # a bucket's partials are folded before its all-reduce
        self.reducer.compute_stream = torch.cuda.current_stream() if self.tokens.is_cuda else None
        try:
            info = self._fwd_bwd_inner()
            ops.flush_grad_sinks()
            return info
        finally:
            ops.GRAD_SINK = False
            ops._DEFER = None
            ops._PENDING = None
            ops._QUEUE = []
            ops._TWIN = None
            ops.SINK_NOTIFY = None
            self.reducer.before_launch = None
            self.reducer.compute_stream = None

    def _fwd_bwd_inner(self):
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            out = self.model(self.tokens, self.tokens, eps=self.eps)
        info = out["loss_info"]
        loss = info["batch-loss"]
        if self._seed is None or self._seed.device != loss.device or self._seed.dtype != loss.dtype:
            self._seed = torch.ones((), dtype=loss.dtype, device=loss.device)
        loss.backward(gradient=self._seed)       # a cached seed: no ones_like launch per step
        return {k: info[k] for k in ("batch-loss", "batch-kl", "batch-nll")}

    def _update(self, bump: bool = True):
        if isinstance(self.opt, dp.FusedAdam):      # clip + Adam + zero_grad in one pass over the flat buckets
            self.opt.step(self.max_norm, self.max_value, bump=bump)
            return
        self.reducer.clip_(self.max_norm, self.max_value)
        self.opt.step()
        self.reducer.zero_grad()

    def _push_weights(self):
        # by-value fills: a pinned staging buffer reused every step would be overwritten by the host while the
        # asynchronous copy of an earlier step is still queued behind the GPU's backlog
        m = self.model
        w = (float(m._recon_scheduler.weight), float(m._kl_scheduler.weight))
        if w == self._w_host:                     # constant schedulers: nothing to upload, no launch
            return
        if w[0] != self._w_host[0]:
            self._w_dev[0:1].fill_(w[0])
        if w[1] != self._w_host[1]:
            self._w_dev[1:2].fill_(w[1])
        self._w_host = w

    # ---- snapshot of everything a warm-up / capture iteration touches ---------------------------
    def _snapshot(self):
        from . import ops
        m = self.model
        snap = {
            "params": [p.detach().clone() for p in m.parameters()],
            "buffers": [b.detach().clone() for b in m.buffers()],
            "sched": (copy.deepcopy(m._recon_scheduler), copy.deepcopy(m._kl_scheduler)),
            "rng": torch.cuda.get_rng_state(self.tokens.device) if self.tokens.is_cuda else None,
            "philox": ops.philox_state(self.tokens.device).clone() if self.tokens.is_cuda else None,
        }
        if isinstance(self.opt, dp.FusedAdam):
            snap["opt"] = {"step": self.opt.step_count.clone(), "m": [t.clone() for t in self.opt.exp_avg],
                           "v": [t.clone() for t in self.opt.exp_avg_sq]}
        else:      # stock optimizer: its state tensors are baked into graph B by address, so they are restored in place
            snap["opt"] = {id(p): {k: v.clone() for k, v in st.items() if isinstance(v, torch.Tensor)}
                           for p, st in self.opt.state.items()}
        return snap

    def _restore(self, snap):
        from . import ops
        m = self.model
        with torch.no_grad():
            for p, s in zip(m.parameters(), snap["params"]):
                p.copy_(s)
            for b, s in zip(m.buffers(), snap["buffers"]):
                b.copy_(s)
            if isinstance(self.opt, dp.FusedAdam):
                self.opt.step_count.copy_(snap["opt"]["step"])
                for d, s in zip(self.opt.exp_avg, snap["opt"]["m"]):
                    d.copy_(s)
                for d, s in zip(self.opt.exp_avg_sq, snap["opt"]["v"]):
                    d.copy_(s)
                self.opt.refresh_shadow()
            else:
                for p, st in self.opt.state.items():
                    saved = snap["opt"].get(id(p), {})
                    for k, v in st.items():
                        if isinstance(v, torch.Tensor):
                            v.copy_(saved[k]) if k in saved else v.zero_()
        m._recon_scheduler, m._kl_scheduler = snap["sched"]
        if snap["rng"] is not None:
            torch.cuda.set_rng_state(snap["rng"], self.tokens.device)
            ops.philox_state(self.tokens.device).copy_(snap["philox"])
        self.reducer.zero_grad()

    def _capture(self, warmup: int):
        from . import ops
        m = self.model
        m.device_weights = self._w_dev
        track = m.track_metrics
        m.track_metrics = False                   # the accumulators are fed by graph B (see _acc), counted in __call__
        self._want_track = bool(track)
        # the AccumulateGrad nodes (kept alive by the reducer's hooks) were created on the default stream while
        # capture runs on torch's capture stream: intentional, the buckets are only touched inside the graphs
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        if m.training:                            # data-dependent initialisation (ActNorm) before the snapshot
            with torch.no_grad(), torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
                snap_sched = (copy.deepcopy(m._recon_scheduler), copy.deepcopy(m._kl_scheduler))
                self._push_weights()
                m(self.tokens, self.tokens, eps=self.eps)
                m._recon_scheduler, m._kl_scheduler = snap_sched
        snap = self._snapshot()
        ops.pin_tables(True)
        # collectives inside graph A (and, on one rank, just the per-bucket ordering of the deferred folds, so that the
        # path is exercised by the single-GPU tests too)
        want_comm = os.environ.get("MIVAE_GRAPH_COMM", "1") != "0"
        self.reducer.overlap = want_comm
        self.reducer.relearn()
        self._push_weights()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):             # warm-up on a side stream (allocator, lazy init, autotune)
            # at least two iterations after relearn(): the first learns the per-bucket contribution counts, the second is
            # the first one that folds per bucket and so builds (and caches) the per-bucket job tables -- a table first
            # needed inside the capture cannot be uploaded there and would bake the one-by-one fold into graph A
            for _ in range(max(warmup, 2)):
                self._fwd_bwd()
                self.reducer.finish()
                self._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ok = False
        if want_comm:
            ok = self._try_capture(comm=True)
            if ok and self.world > 1 and not self._ranks_agree_after_replay():
                print("[TrainStep] captured all-reduce left rank-divergent buckets; using the uncaptured all-reduce",
                      file=sys.stderr)
                self.graph_a = self.graph_b = None
                ok = False
        if not ok:
            ok = self._try_capture(comm=False)
        if ok:
            self._pinned = ops.pinned_tables()    # the tables the captured launches read
        else:                                     # keep training: same step, launched eagerly
            self.graph_a = self.graph_b = None
            self.reducer.overlap = True
            ops.pin_tables(False)
        # Outside the replays the model is an ordinary module again (validation / evaluation forwards between training
        # steps, models/base/model.py:200-210): its plain forward feeds the running metrics itself and reads the
        # schedulers' CURRENT weights by value.  The captured kernels keep reading self._w_dev (the address is baked in),
        # which __call__ refreshes before every replay; replays never run the Python forward, so neither flag is read.
        m.track_metrics = track
        m.device_weights = None
        self._track = track
        self._restore(snap)

    def _try_capture(self, comm: bool) -> bool:
        # thread_local: RCCL's watchdog thread may query events while this thread captures
        self.reducer.overlap = comm
        self.reducer.reset()
        try:
            graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_a, capture_error_mode="thread_local"):
                self.info = self._fwd_bwd()
                if comm:
                    self.reducer.finish()         # remaining buckets + join RCCL's stream: all inside the capture
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b, pool=graph_a.pool(), capture_error_mode="thread_local"):
                track = self._want_track and self._acc is not None
                fold = track and isinstance(self.opt, dp.FusedAdam)     # the optimizer's step counter rides along
                self._update(bump=not fold)
                if track:     # running sums += batch-nll, batch-kl; step counter += 1: ONE single-thread launch
                    import ctypes
                    from . import _lib as L
                    self._acc_src = [self.info["batch-nll"].detach().float(), self.info["batch-kl"].detach().float()]
                    dst = [v.data_ptr() for v in self._acc_views] + ([self.opt.step_count.data_ptr()] if fold else [])
                    src = [t.data_ptr() for t in self._acc_src] + ([None] if fold else [])
                    inc = [0.0, 0.0] + ([1.0] if fold else [])
                    L.call("mivae_accumulate_scalars", (ctypes.c_void_p * len(dst))(*dst),
                           (ctypes.c_void_p * len(src))(*src), (ctypes.c_float * len(inc))(*inc), len(dst), L.stream())
            self.graph_a, self.graph_b, self.comm_in_graph = graph_a, graph_b, comm
            return True
        except RuntimeError as err:
            print(f"[TrainStep] HIP-graph capture {'with' if comm else 'without'} collectives failed ({err})",
                  file=sys.stderr)
            torch.cuda.synchronize()
            self.reducer.reset()
            self.reducer.zero_grad()
            self.reducer.overlap = False
            return False

    def _ranks_agree_after_replay(self) -> bool:
        """One replay of graph A (its collectives included): every rank must end up with the same buckets."""
        self.graph_a.replay()
        sums = torch.stack([b.double().abs().sum() for b in self.reducer.buckets])
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi)) and bool(torch.isfinite(sums).all())
        self.reducer.zero_grad()
        flag = torch.tensor([1.0 if same else 0.0], device=sums.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)     # one verdict for all ranks
        return bool(flag.item() > 0)

    # ---- one step -------------------------------------------------------------------------------
    def __call__(self) -> Dict[str, torch.Tensor]:
        if self.graph_a is None:
            self.info = self._fwd_bwd()
            self.reducer.finish()
            self._update()
            return self.info
        m = self.model
        if self._track:                           # (re)attach the model's running sums to the graph's accumulators
            for name, view in zip(("nll", "avg-kl"), self._acc_views):
                rm = m._metrics[name]
                if rm.total is not view:
                    if isinstance(rm.total, torch.Tensor):
                        view.copy_(rm.total)
                    else:
                        view.fill_(0.0 if rm.total is None else float(rm.total))
                    rm.total = view
        self._push_weights()                      # the weights this step is computed with
        self.graph_a.replay()
        if self.world > 1 and not self.comm_in_graph:
            self.reducer.finish()                 # bucket all-reduces, outside the graphs
        self.graph_b.replay()
        ops.params_changed()                      # the replayed optimizer kernel rewrote the parameters
        if m.training:                            # what VAELmModel.forward does after the loss (model.py:180-184)
            m.kl_scheduler_step()
            m._recon_scheduler.step()
        if self._track:                           # graph B added batch-nll / batch-kl into the running sums
            m._metrics["nll"].count += 1
            m._metrics["avg-kl"].count += 1
            m._metrics["avg-recon-weight"].update(m._recon_scheduler.weight)
            m._metrics["avg-kl-weight"].update(m._kl_scheduler.weight)
        return self.info
