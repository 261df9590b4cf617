"""Timing probe for the d W product of the fused projection + NLL backward: split-K batched variants."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonauto_lm_amd as M
N, V, E = 16384, 8192, 64
dl = torch.randn(N, V, device="cuda").to(torch.bfloat16)
h = torch.randn(N, E, device="cuda").to(torch.bfloat16)
ha = torch.cat([h, torch.ones(N, 8, device="cuda", dtype=torch.bfloat16)], 1)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
ref = dl.double().t() @ ha.double()
for G in (4, 8, 16, 32):
    f = lambda: torch.bmm(dl.view(G, N // G, V).transpose(1, 2), ha.view(G, N // G, E + 8))
    p = f()
    err = float((M.ops.colsum(p.view(G, -1)).view(V, E + 8).double() - ref).abs().max() / ref.abs().max())
    msg = f"G={G:2d} bf16 partials: bmm {t(f):6.1f} us, colsum {t(lambda: M.ops.colsum(p.view(G, -1))):5.1f} us, err {err:.2e}"
    try:
        f32 = lambda: torch.bmm(dl.view(G, N // G, V).transpose(1, 2), ha.view(G, N // G, E + 8), out_dtype=torch.float32)
        p32 = f32()
        e32 = float((M.ops.colsum(p32.view(G, -1)).view(V, E + 8).double() - ref).abs().max() / ref.abs().max())
        msg += f" | fp32 partials: bmm {t(f32):6.1f} us, colsum {t(lambda: M.ops.colsum(p32.view(G, -1))):5.1f} us, err {e32:.2e}"
    except Exception as ex:
        msg += f" | fp32 out: {type(ex).__name__} {str(ex)[:60]}"
    print(msg)
one = dl.t().matmul(ha)
print("single GEMM err", float((one.double() - ref).abs().max() / ref.abs().max()))
