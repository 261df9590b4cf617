"""Kernel census of one training step (forward + backward of sum(log p)) of a Glow-ordered mixed flow list through the
mixed-stack training path: kernel name -> launches, from the torch profiler.  Usage: python tools/mixed_train_census.py [D H K B S]"""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonauto_lm_amd as M  # noqa: E402

D, H, K, B, S = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (64, 64, 4, 256, 64)
torch.manual_seed(0)
flows = []
for _ in range(K):
    a = M.Flow.from_params(type="actnorm", input_size=D)
    a._is_initted = True
    flows += [a, M.InvertibleLinear.from_params(type="inv-linear", input_size=D),
              M.Flow.from_params(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H]))]
prior = M.FlowPrior(D, flows).cuda()
z = torch.randn(B, S, D, device="cuda", requires_grad=True)
mask = torch.ones(B, S, dtype=torch.long, device="cuda")


def step():
    for p in prior.parameters():
        p.grad = None
    z.grad = None
    prior.forward_pass(z, mask)[1].sum().backward()


for fused in (True, False):
    prior.allow_fused = fused
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    cnt, dur = collections.Counter(), collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            name = e.name.replace("void ", "").replace("(anonymous namespace)::", "")[:70]
            cnt[name] += 1
            dur[name] += e.device_time if hasattr(e, "device_time") else e.cuda_time
    print(f"== {'mixed-stack training path' if fused else 'per-flow autograd'}: {sum(cnt.values())} launches, "
          f"{sum(dur.values()):.0f} us of GPU time")
    for name, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:40]:
        print(f"  {n:4d} x {dur[name] / n:8.1f} us  {name}")
