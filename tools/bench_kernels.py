"""Kernel-level micro-bench of the fused NICE stack (fwd / bwd / inverse) and the loss-path kernels on
the BASELINE shapes; prints one JSON line per (config, kernel).  GPU only.
    python tools/bench_kernels.py [c2 c3 ...] [--iters 30]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nonauto_lm_amd as M  # noqa: E402
from nonauto_lm_amd import ops  # noqa: E402

CONFIGS = {"c1": (32, 32, 64, 64, 4, torch.float32), "c2": (256, 64, 64, 64, 4, torch.bfloat16),
           "c2f": (256, 64, 64, 64, 4, torch.float32),
           "c3": (256, 128, 256, 256, 16, torch.bfloat16), "c3s": (64, 128, 256, 256, 16, torch.bfloat16),
           "c4": (256, 1, 64, 64, 4, torch.float32),
           "c5h": (64, 512, 256, 256, 32, torch.bfloat16), "c5": (64, 512, 512, 512, 32, torch.bfloat16)}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
    return ts[len(ts) // 2]


def main():
    argv = sys.argv[1:]
    iters = 30
    if "--iters" in argv:
        i = argv.index("--iters")
        iters = int(argv[i + 1])
        del argv[i:i + 2]
    names = [a for a in argv if not a.startswith("--")] or ["c2", "c3"]
    dev = torch.device("cuda")
    for name in names:
        B, S, D, H, K, dt = CONFIGS[name]
        e = 2 if dt == torch.bfloat16 else 4
        N, h = B * S, D // 2
        torch.manual_seed(0)
        params = []
        for _ in range(K):
            for _net in range(2):
                params += [torch.randn(H, h, device=dev) * (0.5 / h ** 0.5), torch.randn(H, device=dev) * 0.1,
                           torch.randn(h, H, device=dev) * (0.5 / H ** 0.5), torch.randn(h, device=dev) * 0.1]
        z = torch.randn(B, S, D, device=dev).to(dt)
        lens = torch.randint(max(2, S // 4), S + 1, (B,)) if name.startswith("c5") else torch.full((B,), S)
        mask = (torch.arange(S)[None, :] < lens[:, None]).long().to(dev)
        if S == 1:
            z, mask = z.view(B, D), None
        w_bytes = K * 2 * (2 * h * H + H + h) * e
        ops.profile_kernels(True)
        with torch.no_grad():
            t_tot_f = timed(lambda: ops.nice_stack(z, mask, params, H, "gelu", base=True), iters)
        prof = ops.profile_kernels(False)
        evs = prof["nice_stack_fwd"][3:]
        t_f = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)[len(evs) // 2]
        zg = z.clone().requires_grad_(True)
        pg = [p.clone().requires_grad_(True) for p in params]

        def fb():
            zK, lp = ops.nice_stack(zg, mask, pg, H, "gelu", base=True)
            (lp.sum() + zK.float().sum()).backward()
        ops.profile_kernels(True)
        t_tot_fb = timed(fb, iters)
        prof = ops.profile_kernels(False)
        evs = prof["nice_stack_bwd"][3:]
        t_b = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)[len(evs) // 2]
        by_f = 2 * N * D * e + 12 * N + w_bytes
        by_b = 3 * N * D * e + 12 * N + 2 * w_bytes
        fl = 4 * K * N * D * H
        for kn, us, by, f in (("stack_fwd", t_f, by_f, fl), ("stack_bwd", t_b, by_b, 2 * fl)):
            print(json.dumps({"config": name, "kernel": kn, "us": round(us, 2), "algoGBps": round(by / us / 1e3, 1),
                              "frac_hbm": round(by / us / 1e3 / 8000, 4), "TFLOPs": round(f / us / 1e6, 1)}))
        print(json.dumps({"config": name, "op": "nice_stack fwd (pack+kernel+segsum)", "us": round(t_tot_f, 1),
                          "op2": "fwd+bwd incl. dW GEMMs", "us2": round(t_tot_fb, 1)}), flush=True)
        # loss path kernels
        mu = torch.randn(B, S, D, device=dev).to(dt)
        sg = (torch.rand(B, S, D, device=dev) + 0.5).to(dt)
        eps = torch.randn(B, 1, S, D, device=dev).to(dt)
        with torch.no_grad():
            t_r = timed(lambda: ops.reparam_logq(mu, sg, eps, mask if S > 1 else None, 1), iters)
            t_g = timed(lambda: ops.gauss_logprob(z, mask, 0.0, 1.0), iters)
        print(json.dumps({"config": name, "op": "reparam_logq fwd(+segsum)", "us": round(t_r, 1),
                          "algoGBps": round(4 * N * D * e / t_r / 1e3, 1),
                          "op2": "gauss_logprob fwd(+segsum)", "us2": round(t_g, 1),
                          "algoGBps2": round(N * D * e / t_g / 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
