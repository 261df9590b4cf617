"""Timing of mivae_embedding_bwd vs torch's embedding_dense_backward at the bench shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonauto_lm_amd as M
L = M._lib
N, V, E = 16384, 8192, 64
ids = torch.randint(4, V, (N,), device="cuda")
dy = torch.randn(N, E, device="cuda")
dw = torch.zeros(V, E, device="cuda")
def t(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("mivae_embedding_bwd us:", t(lambda: L.call("mivae_embedding_bwd", L.ptr(dy), L.ptr(ids), L.ptr(dw), N, V, E, 0, 1, L.stream())))
print("torch embedding_dense_backward us:", t(lambda: torch.ops.aten.embedding_dense_backward(dy, ids, V, 0, False)))
