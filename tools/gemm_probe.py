"""Timing probe: split-K batched weight-gradient products dW (Fo, Fi) = dy^T x over N tokens, by split count G."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonauto_lm_amd as M
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for N, Fo, Fi in ((32768, 512, 256), (32768, 256, 512), (32768, 256, 256), (16384, 256, 128), (32768, 1024, 512)):
    dy = torch.randn(N, Fo, device="cuda").to(torch.bfloat16)
    x = torch.randn(N, Fi, device="cuda").to(torch.bfloat16)
    line = f"N={N} {Fo}x{Fi}: single {t(lambda: dy.t().matmul(x)):6.1f}"
    for G in (2, 4, 8, 16, 32, 64, 128):
        f = lambda: M.ops.colsum(torch.bmm(dy.view(G, N // G, Fo).transpose(1, 2), x.view(G, N // G, Fi), out_dtype=torch.float32).view(G, -1))
        line += f" | G{G} {t(f):6.1f}"
    print(line)
