"""Timing of a Glow-ordered flow list (ActNorm, inv-linear, NICE) x K: the one-launch mixed stack (csrc/mixed_stack.hip)
vs the per-flow launch sets, in the evaluation and sampling directions (no gradients) and as a TRAINING step (forward +
backward of sum(log p) under autograd: the training path of the mixed stack vs per-flow autograd, host-synchronised wall
time per iteration and kernel launches per iteration counted by the profiler).  Prints one JSON line per shape."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonauto_lm_amd as M


def build(D, H, K):
    flows = []
    for _ in range(K):
        a = M.Flow.from_params(type="actnorm", input_size=D)
        a._is_initted = True
        flows += [a, M.InvertibleLinear.from_params(type="inv-linear", input_size=D),
                  M.Flow.from_params(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H]))]
    return M.FlowPrior(D, flows).cuda()


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


SHAPES = [(16, 32, 2, 64, 32), (64, 64, 4, 256, 64), (128, 128, 4, 256, 64)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(sys.argv[1])]]
for D, H, K, B, S in SHAPES:
    torch.manual_seed(0)
    prior = build(D, H, K)
    z = torch.randn(B, S, D, device="cuda")
    mask = torch.ones(B, S, dtype=torch.long, device="cuda")
    out = {}
    with torch.no_grad():
        for name, fused in (("one_launch_us", True), ("per_flow_us", False)):
            prior.allow_fused = fused
            out[name] = round(timed(lambda: prior.forward_pass(z, mask)), 1)
            out[name.replace("_us", "_sampling_us")] = round(timed(lambda: prior.backward_pass(z, mask)), 1)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        from nonauto_lm_amd import flows as F
        steps = []
        for f in prior._flows:
            if isinstance(f, F.ActNorm):
                steps.append((M.ops.MIXED_ACTNORM, f._log_scale, f._bias))
            elif isinstance(f, F.NICECoupling):
                p, Hh, act = f.fused_spec(z)
                steps.append((M.ops.MIXED_NICE, Hh, act, p))
            else:
                steps.append((M.ops.MIXED_LINEAR, f._weight))
        steps = M.ops.MixedPlan(steps)
        for _ in range(3):
            M.ops.mixed_stack(z, mask, steps)
        ev0.record()
        for _ in range(20):
            M.ops.mixed_stack(z, mask, steps)
        ev1.record()
        torch.cuda.synchronize()
        out["kernel_plus_segment_sum_us"] = round(ev0.elapsed_time(ev1) / 20 * 1e3, 1)
    # training step: forward + backward under autograd
    from torch.profiler import profile, ProfilerActivity
    zg = z.clone().requires_grad_(True)

    def train_step():
        for p in prior.parameters():
            p.grad = None
        zg.grad = None
        prior.forward_pass(zg, mask)[1].sum().backward()
    for name, fused in (("train_mixed", True), ("train_per_flow", False)):
        prior.allow_fused = fused
        out[name + "_us"] = round(timed(train_step, 30), 1)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            train_step()
            torch.cuda.synchronize()
        out[name + "_launches"] = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
    print(json.dumps({"D": D, "H": H, "flows": 3 * K, "tokens": B * S, **out}))
