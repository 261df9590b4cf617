"""Which framework ops launch the small elementwise / copy kernels of one training step (eager, torch.profiler)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, nonauto_lm_amd as M
from nonauto_lm_amd import dp
from nonauto_lm_amd.step import TrainStep
from torch.profiler import profile, ProfilerActivity
B, S, D, H, K, dtype_name, ragged = bench.CONFIGS["c2"]
dev = torch.device("cuda", 0)
torch.manual_seed(13)
model = M.VAELmModel.from_params(vocab=M.SimpleVocab(bench.VOCAB), **bench.model_config(D, H, K)).to(dev).train()
model.training_outputs_only = True
reducer = dp.GradReducer(model.parameters(), flatten_params=True)
opt = dp.FusedAdam(reducer, lr=1e-4, bf16_shadow=True)
tokens = bench.synthetic_tokens(B, S, ragged, 13).to(dev)
step = TrainStep(model, reducer, opt, tokens, torch.bfloat16, 5.0, 2.0, graph=False)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
for t, c, k, sh in rows[:70]:
    print(f"{t:8.1f} us  x{c:3d}  {k:45s} {sh}")
