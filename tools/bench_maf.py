"""Timing of the MAF / IAF kernels (8f-2) at the C2 latent shape: B=256, S=64, D=64, one hidden layer H.
    python tools/bench_maf.py [--hidden 64] [--dtype bf16]
Prints one JSON line: affine tail fwd/bwd (HBM roofline), sequential direction kernel vs the composed D-pass loop."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonauto_lm_amd as M  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--dim", type=int, default=64)
    args = ap.parse_args()
    M._lib.lib()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    e = 2 if dt == torch.bfloat16 else 4
    B, S, D, H = args.batch, args.seq, args.dim, args.hidden
    N = B * S
    torch.manual_seed(13)
    flow = M.MAFlow(M.MADE(D, [H]), parity=True).cuda()
    z = torch.randn(B, S, D, device="cuda", dtype=dt)
    st = (0.3 * torch.randn(B, S, 2 * D, device="cuda")).to(dt)
    mask = torch.ones(B, S, dtype=torch.int64, device="cuda")
    out, tok = torch.empty_like(z), torch.empty(N, device="cuda")
    dz, dst = torch.empty_like(z), torch.empty_like(st)
    L = M._lib
    code = L.dtype_code(dt)

    def aff_fwd():
        L.call("mivae_maf_affine_fwd", L.ptr(z), L.ptr(st), L.ptr(mask), L.ptr(out), L.ptr(tok), N, D, 1, code, L.stream())

    def aff_bwd():
        L.call("mivae_maf_affine_bwd", L.ptr(out), L.ptr(tok), L.ptr(z), L.ptr(st), L.ptr(mask), L.ptr(dz), L.ptr(dst),
               N, D, 1, code, L.stream())

    l1, l2 = flow._made.masked_layers()
    packed = M.ops.maf_sequential_pack(l1.weight, l1.weight_mask, l1.bias, l2.weight, l2.weight_mask, l2.bias)
    w1t, b1, w2, b2, Hp = packed
    act = M.ops.MAF_ACT_IDS["elu"]

    def seq():
        L.call("mivae_maf_sequential", L.ptr(z), L.ptr(w1t), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(mask), L.ptr(out),
               L.ptr(tok), N, D, Hp, act, 1, code, L.stream())

    def composed():
        with torch.no_grad():
            x = torch.zeros_like(z)
            ld = torch.zeros(B, S, device="cuda", dtype=dt)
            zz = z.flip(dims=(-1,))
            for i in range(D):
                s_t = flow._made(x.clone())
                x[..., i] = (zz[..., i] - s_t[..., D + i]) * torch.exp(-s_t[..., i])
                ld = ld + s_t[..., i]
            return x * mask.unsqueeze(-1), ld.mul(mask).sum(-1)

    flow = flow.to(dt)
    t_f, t_b = timeit(aff_fwd, 50), timeit(aff_bwd, 50)
    t_s, t_c = timeit(seq, 20), timeit(composed, 3, 1)
    by_f = N * D * e * 4 + N * 12            # z + st(2) + out, mask + tok
    by_b = N * D * e * 7 + N * 12            # dout + z + s read, dz + dst(2) written (+ t half not read)
    flops = N * D * Hp * 8                   # act ~2 + 3 fma per (token, feature, hidden unit), rough
    print(json.dumps({
        "shape": {"B": B, "S": S, "D": D, "H": H, "Hp": Hp, "dtype": args.dtype},
        "maf_affine_fwd": {"us": round(t_f, 2), "algorithmic_bytes": by_f, "GBps": round(by_f / t_f / 1e3, 1)},
        "maf_affine_bwd": {"us": round(t_b, 2), "algorithmic_bytes": by_b, "GBps": round(by_b / t_b / 1e3, 1)},
        "maf_sequential": {"us": round(t_s, 2), "valu_gflops_est": round(flops / t_s / 1e3, 1),
                           "composed_loop_us": round(t_c, 1), "speedup": round(t_c / t_s, 1)}}))


if __name__ == "__main__":
    main()
