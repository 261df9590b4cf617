"""Roofline evidence for the HBM-bound kernels of the latent path at the BASELINE latent shapes: every
entry point is timed alone through the C ABI (captured 10x into a HIP graph, replayed between two HIP events: pure kernel
time) and reported against its ALGORITHMIC bytes -- GB/s, fraction of the 8 TB/s datasheet peak and of the copy bandwidth
measured on this GPU in the same run.  One JSON line per (config, kernel).  GPU only.

    python tools/bench_hbm_kernels.py [c2 c3 c5 ...] [--plain]      (--plain: eager launches, for rocprofv3 --pmc passes)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nonauto_lm_amd as M  # noqa: E402
from nonauto_lm_amd import _lib as L, ops  # noqa: E402


def main():
    argv = sys.argv[1:]
    plain = "--plain" in argv
    names = [a for a in argv if not a.startswith("--")] or ["c3", "c5"]
    dev = torch.device("cuda:0")
    hbm = bench.hbm_copy_bandwidth(dev, mib=64 if plain else 1024)
    for name in names:
        B, S, D, H, K, dtype_name, ragged = bench.CONFIGS[name]
        dt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
        e = 2 if dt == torch.bfloat16 else 4
        code = L.dtype_code(dt)
        tokens = bench.synthetic_tokens(B, S, ragged, 13).to(dev)
        mask = ops.tokens_mask(tokens)
        N, h = B * S, D // 2
        n_real = int((tokens != 0).sum())
        g = torch.Generator().manual_seed(5)
        rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev).to(dt)
        z, s_, t_, dout = rnd(B, S, D), rnd(B, S, h) * 0.3, rnd(B, S, h), rnd(B, S, D)
        mu_t, pre_t = rnd(B, S, D), rnd(B, S, D)
        out, dz = torch.empty_like(z), torch.empty_like(z)
        ds, dtt = torch.empty_like(s_), torch.empty_like(t_)
        tok = torch.empty(N, dtype=torch.float32, device=dev)
        dlp, dld = torch.randn(B, device=dev), torch.randn(B, device=dev)
        ls, bias = (torch.randn(D, device=dev) * 0.05), torch.randn(D, device=dev) * 0.1
        ld = torch.empty(B, dtype=torch.float32, device=dev)
        dpar = torch.empty(2, D, dtype=torch.float32, device=dev)
        ws = torch.empty(L.query("mivae_col_workspace_floats", N, D, code), dtype=torch.float32, device=dev)
        mean, std = torch.empty(D, device=dev), torch.empty(D, device=dev)
        w = torch.linalg.qr(torch.randn(D, D, generator=g))[0].to(dev).to(dt).contiguous()
        nsl = L.query("mivae_rowgemm_wgrad_slabs", N, D)
        part = torch.empty(nsl, D * D, dtype=torch.float32, device=dev)
        sg_out, used = torch.empty_like(z), torch.empty(2, dtype=torch.int64, device=dev)
        dmu, dsg = torch.empty_like(z), torch.empty_like(z)
        state = ops.philox_state(dev)
        st = L.stream
        mb = 8 * N                       # the int64 mask, read once
        R = n_real                       # algorithmic rows of the masked kernels: padding tokens carry no information
        calls = {
            # name: (call, algorithmic bytes)
            "posterior_fwd (sigma head + sample + log q, in-kernel noise)": (
                lambda: L.call("mivae_posterior_fwd", L.ptr(mu_t), L.ptr(pre_t), 1, 1e-4, 5.0, None, L.ptr(state), L.ptr(used),
                               L.ptr(mask), L.ptr(out), L.ptr(sg_out), L.ptr(tok), B, 1, S, D, code, st()),
                4 * R * D * e + mb + 4 * N),
            "posterior_fwd (noise read from HBM instead: VALU share of the in-kernel draw)": (
                lambda: L.call("mivae_posterior_fwd", L.ptr(mu_t), L.ptr(pre_t), 1, 1e-4, 5.0, L.ptr(dout), None, None,
                               L.ptr(mask), L.ptr(out), L.ptr(sg_out), L.ptr(tok), B, 1, S, D, code, st()),
                5 * R * D * e + mb + 4 * N),
            "posterior_bwd": (
                lambda: L.call("mivae_posterior_bwd", L.ptr(dout), L.ptr(dlp), 0, L.ptr(pre_t), 1, 1e-4, 5.0, None, L.ptr(used),
                               L.ptr(mask), L.ptr(dmu), L.ptr(dsg), B, 1, S, D, code, st()),
                4 * R * D * e + mb),
            "gauss_logprob_fwd": (
                lambda: L.call("mivae_gauss_logprob_fwd", L.ptr(z), L.ptr(mask), L.ptr(tok), B, 1, S, D, 0.0, 1.0, code, st()),
                R * D * e + mb + 4 * N),
            "gauss_logprob_bwd": (
                lambda: L.call("mivae_gauss_logprob_bwd", L.ptr(z), L.ptr(mask), L.ptr(dlp), L.ptr(dz), B, 1, S, D, 0.0, 1.0,
                               code, st()),
                2 * R * D * e + mb),
            "nice_affine_fwd": (
                lambda: L.call("mivae_nice_affine_fwd", L.ptr(z), L.ptr(s_), L.ptr(t_), L.ptr(mask), L.ptr(out), L.ptr(tok), B, 1,
                               S, D, code, 0, st()),
                (2 * D + 2 * h) * R * e + mb + 4 * N),
            "nice_affine_bwd": (
                lambda: L.call("mivae_nice_affine_bwd", L.ptr(dout), L.ptr(dld), L.ptr(z), L.ptr(s_), L.ptr(t_), L.ptr(mask),
                               L.ptr(dz), L.ptr(ds), L.ptr(dtt), B, 1, S, D, code, 0, st()),
                (3 * D + 4 * h) * R * e + mb),
            "actnorm_fwd": (
                lambda: L.call("mivae_actnorm_fwd", L.ptr(z), L.ptr(ls), L.ptr(bias), L.ptr(mask), L.ptr(out), L.ptr(ld), B, 1, S,
                               D, B, 1, code, 0, st()),
                2 * R * D * e + mb),
            "actnorm_bwd (+ d log_scale, d bias)": (
                lambda: L.call("mivae_actnorm_bwd", L.ptr(dout), L.ptr(z), L.ptr(ls), L.ptr(mask), L.ptr(dz), L.ptr(dpar), L.ptr(ws),
                               B, 1, S, D, code, st()),
                3 * R * D * e + mb),
            "actnorm_stats (two passes)": (
                lambda: L.call("mivae_actnorm_stats", L.ptr(z), L.ptr(mean), L.ptr(std), L.ptr(ws), N, D, code, st()),
                2 * N * D * e),
            "rowgemm (inv-linear z W^T)": (
                lambda: L.call("mivae_rowgemm", L.ptr(z), L.ptr(w), L.ptr(out), N, D, code, st()),
                2 * N * D * e + D * D * e),
            "rowgemm_wgrad (slab partials)": (
                lambda: L.call("mivae_rowgemm_wgrad", L.ptr(z), L.ptr(dout), L.ptr(part), N, D, code, st()),
                2 * N * D * e + nsl * D * D * 4),
        }
        res = bench._time_calls({k: v[0] for k, v in calls.items()}, 10 if plain else 100, not plain)
        for k, (_, by) in calls.items():
            if k not in res:
                continue
            us = res[k]["avg_us"]
            gbps = by / us / 1e3
            print(json.dumps({"config": name, "kernel": k, "tokens": N, "real_tokens": n_real, "D": D, "dtype": dtype_name,
                              "avg_us": round(us, 2), "algorithmic_bytes": by, "GBps": round(gbps, 1),
                              "frac_of_8TBps": round(gbps / 8000, 3), "frac_of_measured_copy": round(gbps / hbm, 3),
                              "hbm_measured_GBps": round(hbm, 1)}), flush=True)


if __name__ == "__main__":
    main()
