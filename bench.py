"""bench.py -- train sentences/sec of the ELBO fwd+bwd step on synthetic token batches (BASELINE.json),
plus the roofline fraction of the dominant hand-written kernel and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5|c1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = the reference's `_train_batch` (training/trainers/trainer.py:277-303): forward ELBO of the
non-autoregressive VAE (transformer encoder/decoder on stock PyTorch-ROCm ops, latent path on the HIP
kernels), backward, bucketed RCCL gradient all-reduce overlapped with backward, clip (norm 5, value 2),
Adam step.  Inputs are resident in HBM before the timed region.  Weak scaling: the per-GPU batch is
fixed; value = global sentences / max-over-ranks time.

Workload at N=1 = BASELINE configs[1] ("C2"): B=256, S=64, D=64, K=4 NICE-FF steps (H=64), bf16.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0
# dense MFMA peaks by compute dtype (MI355X_MICROARCH.md: bf16 ~2.5 PF; f32-input MFMA = the vector peak)
MFMA_PEAK_TFLOPS = {"bf16": MFMA_BF16_PEAK_TFLOPS, "fp32": 157.3}

CONFIGS = {
    #        B    S    D    H    K   dtype   ragged
    "c1": (32, 32, 64, 64, 4, "f32", False),
    "c2": (256, 64, 64, 64, 4, "bf16", False),
    "c3": (256, 128, 256, 256, 16, "bf16", False),
    "c5": (64, 512, 512, 512, 32, "bf16", True),
}
VOCAB = 8192


def model_config(D, H, K):
    return dict(
        type="nonauto",
        embedder=dict(embedding_dim=D),
        encoder=dict(type="transformer", input_size=D, hidden_size=2 * D, num_layers=2, num_heads=4),
        decoder=dict(type="nonauto-transformer", input_size=D, hidden_size=2 * D, num_layers=2, num_heads=4),
        posterior=dict(type="default", input_size=D, features=D, samples=1),
        prior=dict(type="flow", features=D, flows=[
            dict(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H],
                                            activation="gelu", dropout=0.0)) for _ in range(K)]),
        kl_loss=dict(type="default"),
        recon_scheduler=dict(type="constant", weight=1.0),
        kl_scheduler=dict(type="constant", weight=1.0),
    )


def synthetic_tokens(B, S, ragged, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(max(2, S // 4), S + 1, (B,), generator=g) if ragged else torch.full((B,), S)
    tok = torch.randint(4, VOCAB, (B, S), generator=g)
    return tok * (torch.arange(S)[None, :] < lens[:, None])


def cpu_baseline(model, tokens, D, H, K, budget_s=20.0):
    """The same ELBO fwd+bwd step on the host cores: stock PyTorch encoder/decoder + the CPU oracle's
    latent path (oracle/latent_path.py, pinned against the reference's golden vectors), fp32.
    Bounded sample: a slice of the batch sized so a few steps take ~10-30 s."""
    import nonauto_lm_amd as M
    from oracle import latent_path as O
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    m = M.VAELmModel.from_params(vocab=M.SimpleVocab(VOCAB), **model_config(D, H, K)).train()
    m.load_state_dict({k: v.detach().float().cpu() for k, v in model.state_dict().items()})
    B = min(tokens.size(0), 64)
    tok = tokens[:B].cpu()

    def step():
        m.zero_grad(set_to_none=True)
        mask = O.tokens_mask(tok)
        enc = m._encoder.encoder(m._encoder._preprocess_embedding(m._embedder(tok)), mask)
        mu = m._posterior._mu_net(enc)
        sigma = m._posterior._sigma_net(enc)
        eps = torch.randn(B, 1, *mu.shape[1:])
        z, mrep = O.posterior_sample(mu, sigma, eps, mask, 1)
        logq = O.posterior_log_prob(z, mu, sigma, mrep, 1)
        logits = m._vocab_projection(m._decoder(z, mrep))
        recon = O.label_smoothing_nll(logits, tok, mrep.float(), 0.0, True)
        sd = dict(m._prior.named_parameters())
        ps = [O.NiceParams.from_state_dict(sd, f"_flows.{k}.") for k in range(K)]
        _, logp = O.flow_prior_forward(z, O.nice_steps(ps), mrep)
        out = O.elbo_tail(recon, O.kl_default(logq, logp), B, 1)
        out["batch-loss"].backward()

    step()
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < budget_s / 2 and n < 20):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": B / dt, "unit": "sentences/s", "cores": cores, "kind": "port",
            "sample": f"{n} fp32 steps of the same model on a {B}-sentence slice of the batch "
                      f"({dt * 1e3:.1f} ms/step); latent path = oracle/latent_path.py, enc/dec = stock PyTorch CPU"}


def flow_kernel_times(model, tokens, dtype, reps=50, graph=True):
    """Average GPU duration of the dominant hand-written launches (reconstruction NLL forward / backward over the
    (N, V) logits, fused NICE stack forward, its backward, the weight-gradient kernel pair) on the bench shape: each launch is captured alone into a
    HIP graph through the C ABI and replayed `reps` times between two HIP events on the launch stream,
    so no host launch gap is included."""
    import nonauto_lm_amd as M
    from nonauto_lm_amd import _lib as L, ops
    from nonauto_lm_amd.flows import fused_stack_spec
    dev = tokens.device
    B, S = tokens.shape
    prior = model._prior
    D = prior._features
    h = D // 2
    z = torch.randn(B, S, D, device=dev).to(dtype)
    mask = ops.tokens_mask(tokens)
    spec = fused_stack_spec(prior._flows, z)
    if spec is None:                # not an all-NICE-FF stack the fused kernels take: the step ran the composed path
        return {}
    params, H, act = spec
    K = len(params) // 8
    code, N = L.dtype_code(dtype), B * S
    big = ops.nice_big_supported(z, H)             # wide latents (97 <= D/2 <= 256, bf16): csrc/nice_big.hip
    hp, Hp = (h + 31) // 32 * 32, (H + 31) // 32 * 32
    e = lambda *shape, dt=dtype: torch.empty(*shape, dtype=dt, device=dev)
    z_out, hist, tok = e(B, S, D), e(K, N, h), e(N, dt=torch.float32)
    dz_out, dlog, dz_in = torch.randn(B, S, D, device=dev).to(dtype), torch.randn(N, device=dev), e(B, S, D)
    plist = [p.detach() for p in params]
    if big:
        z = z * mask.unsqueeze(-1).to(dtype)       # what the posterior hands over: zeros at padding
        pk_f, pk_b, pk_c, pk_n = (ops.nice_big_pack(plist, K, h, H, kind) for kind in ("fwd", "bias", "chain", "nets"))
        ptab = ops._ptr_table(plist, dev)
        dsdt = e(K, 2, N, h)
        ws = e(L.query("mivae_nice_big_dw_workspace_floats", N, h, H, K), dt=torch.float32)
        hid = e(L.query("mivae_nice_big_hidden_elems", N, h, H, K)) if ops._nice_big_save_hidden(h) else None
    else:
        npad = L.query("mivae_nice_stack_npad", N)
        packed = ops.nice_pack(plist, K, h, H, dtype)
        Gt, Ht, dSt, x0t = e(K, 2, Hp, npad), e(K, 2, Hp, npad), e(K, 2, hp, npad), e(hp, npad)
        ws = e(L.query("mivae_nice_dw_workspace_floats", N, h, H, K), dt=torch.float32)
    dW1, db1, dW2, db2 = (e(K, 2, H, h, dt=torch.float32), e(K, 2, H, dt=torch.float32),
                          e(K, 2, h, H, dt=torch.float32), e(K, 2, h, dt=torch.float32))
    actid = ops.ACT_IDS[act]
    V = VOCAB
    logits, dlogits = torch.randn(B, S, V, device=dev).to(dtype), e(B, S, V)
    w = mask.float()
    per_seq, aux, dper = e(B, dt=torch.float32), e(3, N, dt=torch.float32), torch.ones(B, device=dev)
    E = model._vocab_projection.in_features
    hdec = torch.randn(N, E, device=dev).to(torch.bfloat16)
    w16 = model._vocab_projection.weight.detach().to(torch.bfloat16).contiguous()
    pbias = model._vocab_projection.bias.detach().float().contiguous()
    use_lce = dtype == torch.bfloat16 and ops.linear_nll_supported(hdec, w16)
    if use_lce:
        wp16 = torch.empty(L.query("mivae_lce_pack_elems", V, E), dtype=torch.bfloat16, device=dev)
        wt16 = torch.empty(V * E, dtype=torch.bfloat16, device=dev)
        L.call("mivae_lce_pack2", L.ptr(w16), L.ptr(pbias), L.ptr(wp16), L.ptr(wt16), V, E, L.stream())
        w16 = wp16
        dh_ws = e(max(L.query("mivae_lce_dh_workspace_floats", N, E, V), 1), dt=torch.float32)
        dh16 = torch.empty(N, E, dtype=torch.bfloat16, device=dev)
    lce_ws = e(max(L.query("mivae_lce_workspace_floats", N, V), 1), dt=torch.float32)
    dl16 = torch.empty(N, V, dtype=torch.bfloat16, device=dev)
    calls = {}
    if use_lce:
        calls.update({
            "lce_fwd": lambda: L.call("mivae_lce_fwd", L.ptr(hdec), L.ptr(w16), L.ptr(tokens), L.ptr(w),
                                      L.ptr(per_seq), L.ptr(aux[0]), L.ptr(aux[1]), L.ptr(lce_ws), B, S, E, V, 0.0, L.stream()),
            "lce_bwd": lambda: L.call("mivae_lce_bwd", L.ptr(hdec), L.ptr(w16), L.ptr(tokens), L.ptr(aux[0]),
                                      L.ptr(aux[1]), L.ptr(dper), L.ptr(dl16), B, S, E, V, 0.0, L.stream()),
            "lce_bwd_dh": lambda: L.call("mivae_lce_bwd_dh", L.ptr(hdec), L.ptr(w16), L.ptr(wt16), L.ptr(tokens), L.ptr(aux[0]),
                                         L.ptr(aux[1]), L.ptr(dper), L.ptr(dl16), L.ptr(dh_ws), L.ptr(dh16), B, S, E, V, 0.0,
                                         L.stream()),
        })
    calls.update({
        "nll_fwd": lambda: L.call("mivae_nll_fwd", L.ptr(logits), L.ptr(tokens), L.ptr(w), L.ptr(per_seq), L.ptr(aux[0]),
                                  L.ptr(aux[1]), L.ptr(aux[2]), B, S, V, 0.0, code, L.stream()),
        "nll_bwd": lambda: L.call("mivae_nll_bwd", L.ptr(logits), L.ptr(tokens), L.ptr(aux[0]), L.ptr(aux[1]), L.ptr(dper),
                                  L.ptr(dlogits), B, S, V, 0.0, code, L.stream()),
    })
    if big:      # one entry point each: forward (k_big_chain<FWD>), backward (k_big_chain<CHAIN> + k_big_nets + k_big_dw + reduce)
        calls.update({
            "nice_stack_fwd": lambda: L.call("mivae_nice_big_fwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask),
                                             L.ptr(pk_f), L.ptr(pk_b), L.ptr(tok), B, 1, S, D, H, K, actid, 0, 1, 0.0, 1.0,
                                             L.stream()),
            "nice_stack_bwd": lambda: L.call("mivae_nice_big_bwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask),
                                             L.ptr(ptab), L.ptr(pk_c), L.ptr(pk_n), L.ptr(pk_b), L.ptr(dz_out), L.ptr(dlog),
                                             L.ptr(dz_in), L.ptr(dsdt), L.ptr(hid), L.ptr(ws), L.ptr(dW1), L.ptr(db1), L.ptr(dW2),
                                             L.ptr(db2), None, B, 1, S, D, H, K, actid, 1, 0.0, 1.0, L.stream()),
        })
    else:
        calls.update({
            "nice_stack_fwd": lambda: L.call("mivae_nice_stack_fwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask),
                                             L.ptr(packed), L.ptr(tok), B, 1, S, D, H, K, actid, 0, 1, 0.0, 1.0, code, L.stream()),
            "nice_stack_bwd": lambda: L.call("mivae_nice_stack_bwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask),
                                             L.ptr(packed), L.ptr(dz_out), L.ptr(dlog), L.ptr(dz_in), L.ptr(Gt), L.ptr(Ht),
                                             L.ptr(dSt), L.ptr(x0t), N, 1, 1, D, H, K, actid, 1, 0.0, 1.0, code, L.stream()),
            "nice_dw": lambda: L.call("mivae_nice_dw", L.ptr(Gt), L.ptr(Ht), L.ptr(dSt), L.ptr(x0t), L.ptr(ws), L.ptr(dW1),
                                      L.ptr(db1), L.ptr(dW2), L.ptr(db2), N, h, H, K, code, L.stream()),
        })
    out = {}
    for name, fn in calls.items():
        fn()                                   # eager once (module load, LDS opt-in), also produces the inputs of the next
        torch.cuda.synchronize()
        if not graph:                          # plain launches (the rocprofv3 PMC passes use this)
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            continue
        inner = 10                             # back-to-back launches inside one graph: no host gap between them
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(inner):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        outer = max(reps // inner, 1)
        a.record()
        for _ in range(outer):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        out[name] = {"launches": outer * inner, "avg_us": 1e3 * a.elapsed_time(b) / (outer * inner)}
    return out


def pmc_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/<round>/pmc_traffic.json,
    produced by tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 corrections applied); None if
    that kernel/config was not collected."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(config, {}).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)    # ~0.4 s timed window at C2
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--torch-adam", action="store_true", help="stock torch.optim.Adam + separate clip passes")
    ap.add_argument("--no-sdpa", action="store_true", help="explicit masked-softmax attention in the bracket")
    ap.add_argument("--eager", action="store_true", help="no HIP-graph replay of the step (hook-overlapped all-reduce)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches the GPU) and starts one
        # rank per GPU; under torch.distributed.run the ranks already exist and RANK is set
        ndev = torch.cuda.device_count()           # does not initialise the GPU
        if ndev < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible")
        from nonauto_lm_amd.dp import spawn_local_ranks
        sys.exit(spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import nonauto_lm_amd as M
    from nonauto_lm_amd import dp, ops
    from nonauto_lm_amd.step import TrainStep
    M._lib.lib()                                   # the HIP extension is mandatory
    if args.no_sdpa:
        M.nn_bracket.USE_FUSED_ATTENTION = False

    B, S, D, H, K, dtype_name, ragged = CONFIGS[args.config]
    amp_dtype = torch.bfloat16 if dtype_name == "bf16" else None
    torch.manual_seed(13)                          # identical init on every rank (train_worker.py:48)
    model = M.VAELmModel.from_params(vocab=M.SimpleVocab(VOCAB), **model_config(D, H, K)).to(dev).train()
    model.training_outputs_only = True             # skip probs/preds, unused by the loss
    dp.broadcast_parameters(model)
    reducer = dp.GradReducer(model.parameters(), bucket_bytes=args.bucket_mb << 20, flatten_params=not args.torch_adam)
    if args.torch_adam:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, foreach=True, capturable=not args.eager)
    else:                                          # clip + Adam + zero_grad fused over the flat buckets (libmivae)
        opt = dp.FusedAdam(reducer, lr=1e-4, bf16_shadow=amp_dtype == torch.bfloat16, model=model)
    scalars = dp.PackedScalars(["batch-loss", "batch-kl", "batch-nll", "done"], dev)
    tokens = synthetic_tokens(B, S, ragged, 13 + rank).to(dev)     # per-rank shard of the global batch

    train_step = TrainStep(model, reducer, opt, tokens, amp_dtype, 5.0, 2.0, graph=not args.eager)

    def step():
        info = train_step()
        scalars.reduce({"batch-loss": info["batch-loss"], "batch-kl": info["batch-kl"],
                        "batch-nll": info["batch-nll"], "done": 0.0})
        return info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    loss = scalars.means()["batch-loss"]
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        e = 2 if dtype_name == "bf16" else 4
        N = B * S
        h = D // 2
        n_real = int((tokens != 0).sum().item())
        # NonAutoModel.decode fuses projection + NLL under bf16 autocast when the shapes allow it
        fused_lce = dtype_name == "bf16" and model.training_outputs_only and \
            ops.linear_nll_supported(torch.empty(1, D, device=dev), model._vocab_projection.weight)
        w_bytes = K * 2 * (2 * h * H + H + h) * e
        # algorithmic bytes per launch (DESIGN.md "Kernels"): fused K-step forward reads z and
        # writes z' once (+ int64 mask, per-token value, packed weights); backward reads z', dz', writes dz.
        algo = {
            "nice_stack_fwd": 2 * N * D * e + 8 * N + 4 * N + w_bytes,
            "nice_stack_bwd": 3 * N * D * e + 8 * N + 4 * N + 2 * w_bytes,
            # weight gradients: the side outputs (K x 2 x (2H + h) x N) + x0^T in, fp32 gradients out
            "nice_dw": (K * 2 * (2 * H + h) + h) * N * e + 2 * w_bytes,
            # reconstruction NLL: forward reads the logits once; backward reads them and writes d logits
            # (rows of padding tokens are skipped: count real tokens only)
            "nll_fwd": n_real * VOCAB * e + 16 * N + 12 * N,
            "nll_bwd": (n_real + N) * VOCAB * e + 16 * N,
            # fused projection + NLL: the forward reads h and W only (exp / MFMA bound, not HBM);
            # the backward writes d logits once and reads nothing of that size
            "lce_fwd": N * D * 2 + VOCAB * D * 2 + 32 * N,
            "lce_bwd": N * VOCAB * 2 + N * D * 2 + VOCAB * D * 2 + 16 * N,
            # + d h in the same pass: W^T image read, d h written (slice partials are scratch, not algorithmic)
            "lce_bwd_dh": N * VOCAB * 2 + N * D * 2 + 2 * VOCAB * D * 2 + 16 * N + N * D * 2,
        }
        in_step = {"nll_fwd", "nll_bwd"} if not fused_lce else \
            {"lce_fwd", "lce_bwd_dh" if ops.LCE_FUSED_DH else "lce_bwd"}
        flops = {"nll_fwd": 0, "nll_bwd": 0, "lce_fwd": 2 * N * VOCAB * D, "lce_bwd": 2 * N * VOCAB * D,
                 "lce_bwd_dh": 4 * N * VOCAB * D,
                 # one step = 2 nets x 2 GEMMs x 2 h H flops per token; the backward's algorithmic count is twice the
                 # forward's (d input + d weight per GEMM).  Small-latent path: mivae_nice_stack_bwd = d input half,
                 # mivae_nice_dw = d weight half; wide path: mivae_nice_big_bwd does both (recomputed activations are
                 # not counted).
                 "nice_stack_fwd": 4 * K * N * D * H, "nice_stack_bwd": 4 * K * N * D * H, "nice_dw": 4 * K * N * D * H}
        kern = flow_kernel_times(model, tokens, torch.bfloat16 if dtype_name == "bf16" else torch.float32)
        if "nice_stack_bwd" in kern and "nice_dw" not in kern:      # wide path: one entry point for the whole backward
            flops["nice_stack_bwd"] = 8 * K * N * D * H
            algo["nice_stack_bwd"] += 2 * w_bytes
        kernel_timing = ("each entry point (C ABI) captured 10x back to back in a HIP graph, the graph replayed 5x between "
                         "two HIP events on the launch stream; same process, bench shape, trained weights")
        roofline = None
        roofline_flow = []
        if kern:
            # dominant = the slowest hand-written launch that the timed step actually runs
            skip = {"nll_fwd", "nll_bwd", "lce_fwd", "lce_bwd", "lce_bwd_dh"} - in_step
            dom = max((k for k in kern if k in algo and k not in skip), key=lambda k: kern[k]["avg_us"])
            mfma_peak = MFMA_PEAK_TFLOPS[dtype_name]

            def roof(k, us_k, by, fl):
                """roofline block of one kernel: bound = whichever of (algorithmic flops at the MFMA peak, algorithmic
                bytes at the HBM peak) takes longer at this shape"""
                if fl / (mfma_peak * 1e12) > by / (HBM_PEAK_GBS * 1e9):
                    ach = fl / (us_k * 1e-6) / 1e12
                    return {"kernel": k, "bound": "mfma", "achieved": round(ach, 1), "peak": mfma_peak, "unit": "TFLOP/s",
                            "frac": round(ach / mfma_peak, 4), "traffic": pmc_traffic(args.config, k),
                            "avg_us": round(us_k, 2), "algorithmic_flops": fl, "algorithmic_bytes": by}
                ach = by / (us_k * 1e-6) / 1e9
                return {"kernel": k, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args.config, k),
                        "avg_us": round(us_k, 2), "algorithmic_bytes": by,
                        "mfma_tflops": round(fl / (us_k * 1e-6) / 1e12, 2)}

            roofline = roof(dom, kern[dom]["avg_us"], algo[dom], flops[dom])
            # the north-star kernels: the fused flow stack forward / backward (the small-latent path's backward is
            # the pair mivae_nice_stack_bwd + mivae_nice_dw)
            for k in ("nice_stack_fwd", "nice_stack_bwd"):
                if k not in kern:
                    continue
                pair = k == "nice_stack_bwd" and "nice_dw" in kern
                roofline_flow.append(roof(k + ("+nice_dw" if pair else ""),
                                          kern[k]["avg_us"] + (kern["nice_dw"]["avg_us"] if pair else 0.0),
                                          algo[k] + (algo["nice_dw"] if pair else 0),
                                          flops[k] + (flops["nice_dw"] if pair else 0)))
            for k in kern:
                if k not in algo:
                    continue
                kern[k]["GBps_algorithmic"] = round(algo[k] / (kern[k]["avg_us"] * 1e-6) / 1e9, 1)
                kern[k]["avg_us"] = round(kern[k]["avg_us"], 2)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(model, tokens, D, H, K)
        line = {
            "metric": "train sentences/sec (ELBO fwd+bwd)", "value": round(world * B * args.steps / elapsed, 1),
            "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": f"{args.config}: NAR VAE B={B}/GPU S={S} D={D} K={K} NICE-FF H={H} V={VOCAB} "
                                   f"2+2 transformer layers, fwd+bwd+allreduce+clip+Adam",
                       "global_batch": world * B, "seq_len": S, "latent_dim": D, "flow_steps": K,
                       "parallelism": f"dp{world}"},
            "roofline": roofline, "roofline_flow": roofline_flow, "kernels": kern, "kernel_timing": kernel_timing,
            "step_mode": "hipgraph-replay" if train_step.graph_a is not None else "eager",
            "allreduce": ("none (1 rank)" if world == 1 else "captured in graph A, overlapped with backward"
                          if train_step.comm_in_graph else "between the graphs" if train_step.graph_a is not None
                          else "from the grad hooks, overlapped with backward"), "cpu_baseline": cpu, "final_loss": round(loss, 4),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
