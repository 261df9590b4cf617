"""Which ops launch the kernels of one eager C2 training step: torch.profiler with Python stacks, grouped by
(kernel, first frame inside this package).  python tools/op_census.py [config]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import nonauto_lm_amd as M
from nonauto_lm_amd import dp
from nonauto_lm_amd.step import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
B, S, D, H, K, dtype_name, ragged = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
torch.manual_seed(13)
model = M.VAELmModel.from_params(vocab=M.SimpleVocab(bench.VOCAB), **bench.model_config(D, H, K)).to(dev).train()
model.training_outputs_only = True
reducer = dp.GradReducer(model.parameters(), flatten_params=True)
opt = dp.FusedAdam(reducer, lr=1e-4, bf16_shadow=True, model=model)
tokens = bench.synthetic_tokens(B, S, ragged, 13).to(dev)
step = TrainStep(model, reducer, opt, tokens, torch.bfloat16, 5.0, 2.0, graph=False)
for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
# map: for every CPU op event with a stack, count device kernels launched under it (leaf ops only)
rows = collections.Counter()
times = collections.Counter()
for e in ev:
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    if any(c.kernels for c in e.cpu_children):      # not a leaf launcher
        continue
    frame = "?"
    for f in (e.stack or []):
        if "nonauto" in f and "site-packages" not in f:
            frame = f.split("nonauto-lm_amd/")[-1][:70]
            break
    for k in e.kernels:
        key = (e.name[:40], k.name[:60], frame)
        rows[key] += 1
        times[key] += k.duration
tot = sum(rows.values())
print("kernels in one step:", tot)
for key, n in sorted(rows.items(), key=lambda kv: -times[kv[0]]):
    print(f"{n:3d} {times[key]:7.1f}us  {key[0]:40s} {key[1]:60s} {key[2]}")
