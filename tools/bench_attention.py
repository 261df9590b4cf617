"""Times libmivae's attention kernels against the library's fused attention on the three BASELINE shapes."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonauto_lm_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, B, S, heads, hd in (("c2", 256, 64, 4, 16), ("c3", 256, 128, 4, 64), ("c5", 64, 512, 4, 128)):
    W = heads * hd
    q, k, v = (torch.randn(B, S, W, device="cuda").to(torch.bfloat16).requires_grad_(True) for _ in range(3))
    lens = torch.randint(S // 2, S + 1, (B,), device="cuda")
    mask = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).long()
    dout = torch.randn(B, S, W, device="cuda").to(torch.bfloat16)
    bias = ops.attention_bias(mask, False, torch.bfloat16)
    sp = lambda x: x.view(B, S, heads, hd).transpose(1, 2)

    def mine():
        out = ops.attention(q, k, v, mask, heads)
        out.backward(dout)

    def lib():
        out = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=bias).transpose(1, 2).reshape(B, S, W)
        out.backward(dout)

    def mine_fwd():
        with torch.no_grad():
            ops.attention(q, k, v, mask, heads)

    def lib_fwd():
        with torch.no_grad():
            F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=bias)

    print(name, "fwd+bwd us: mivae %.1f  library %.1f | fwd only: mivae %.1f  library %.1f" %
          (timeit(mine), timeit(lib), timeit(mine_fwd), timeit(lib_fwd)), flush=True)
