"""IAF training direction (sequential MADE recurrence) at the C2 latent shape: one-launch forward + reverse-recurrence
backward vs the composed D-pass loop.  python tools/bench_maf.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonauto_lm_amd as M
from nonauto_lm_amd import ops

def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

B, S, D, H = 256, 64, 64, 64
flow = M.IAFlow(M.MADE(D, [H], permute_order=True, activation="elu"), parity=False).cuda()
z = torch.randn(B, S, D, device="cuda", requires_grad=True)
mask = torch.ones(B, S, dtype=torch.long, device="cuda")
def step():
    x, ld = flow(z, mask)
    (x.sum() + ld.sum()).backward()
t_k = timed(step)
orig = ops.maf_sequential_grad_supported
ops.maf_sequential_grad_supported = lambda w: False
t_c = timed(step, 3)
ops.maf_sequential_grad_supported = orig
print(json.dumps({"shape": [B, S, D, H], "kernel_fwd_bwd_us": round(t_k, 1), "composed_loop_fwd_bwd_us": round(t_c, 1)}))
