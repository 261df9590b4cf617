"""Timing of the fused vocabulary projection + NLL entry points (csrc/nll.hip) at a BASELINE shape, each captured 10x
in a HIP graph and replayed between two events (as bench.py times its kernel table).  One JSON line.
usage: python tools/bench_lce.py [B S E V]      (default: config 2 = 256 64 64 8192)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonauto_lm_amd import _lib as L, ops  # noqa: E402

B, S, E, V = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (256, 64, 64, 8192)
N = B * S
dev = torch.device("cuda")
torch.manual_seed(0)
h = torch.randn(N, E, device=dev).to(torch.bfloat16)
W = (torch.randn(V, E, device=dev) / E ** 0.5).to(torch.bfloat16)
bias = 0.1 * torch.randn(V, device=dev)
tokens = torch.randint(1, V, (B, S), device=dev)
w = torch.ones(N, device=dev)
e = lambda *shape, dt=torch.float32: torch.empty(*shape, dtype=dt, device=dev)
wp = e(L.query("mivae_lce_pack_elems", V, E), dt=torch.bfloat16)
wt = e(V * E, dt=torch.bfloat16)
L.call("mivae_lce_pack2", L.ptr(W), L.ptr(bias), L.ptr(wp), L.ptr(wt), V, E, L.stream())
per_seq, aux, dper = e(B), e(3, N), torch.ones(B, device=dev)
lce_ws = e(max(L.query("mivae_lce_workspace_floats", N, V), 1))
dh_ws = e(max(L.query("mivae_lce_dh_workspace_floats", N, E, V), 1))
dh = e(N, E, dt=torch.bfloat16)
slabs = L.query("mivae_lce_dw_slabs", N, E, V)
part = e(max(slabs, 1), V, E + 8)
dl = None if slabs else e(N, V, dt=torch.bfloat16)     # E > 64: the token-major pass stores d logits for a library product
calls = {
    "lce_fwd": lambda: L.call("mivae_lce_fwd", L.ptr(h), L.ptr(wp), L.ptr(tokens), L.ptr(w), L.ptr(per_seq), L.ptr(aux[0]),
                              L.ptr(aux[1]), L.ptr(lce_ws), B, S, E, V, 0.0, L.stream()),
    "lce_bwd_dh": lambda: L.call("mivae_lce_bwd_dh", L.ptr(h), L.ptr(wp), L.ptr(wt), L.ptr(tokens), L.ptr(aux[0]),
                                 L.ptr(aux[1]), L.ptr(dper), L.ptr(dl), L.ptr(dh_ws), L.ptr(dh), B, S, E, V, 0.0, L.stream()),
}
if slabs:
    calls["lce_dw"] = lambda: L.call("mivae_lce_dw", L.ptr(h), L.ptr(wp), L.ptr(tokens), L.ptr(aux[0]), L.ptr(aux[1]),
                                     L.ptr(dper), L.ptr(part), B, S, E, V, 0.0, L.stream())
out = {"shape": [B, S, E, V], "dbg": os.environ.get("MIVAE_LCE_DBG", "0")}   # (read by -DMIVAE_LCE_ABLATION builds only)
side = torch.cuda.Stream()
for name, fn in calls.items():
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(10):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        out[name + "_us"] = round(1e3 * a.elapsed_time(b) / 50, 2)
out["dh_checksum"] = float(dh.float().abs().sum())
out["dw_checksum"] = float(part.sum(0)[:, :E].abs().sum()) if slabs else None
print(json.dumps(out))
