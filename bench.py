"""bench.py -- train sentences/sec of the ELBO fwd+bwd step on synthetic token batches (BASELINE.json),
plus the roofline fraction of the dominant hand-written kernel and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5|c1] [--no-others]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = the reference's `_train_batch` (training/trainers/trainer.py:277-303): forward ELBO of the VAE
(encoder/decoder bracket + latent path on the HIP kernels), backward, bucketed RCCL gradient all-reduce overlapped with
backward, clip (norm 5, value 2), Adam step.  Inputs are resident in HBM before the timed region.  Weak scaling: the
per-GPU batch is fixed; value = global sentences / max-over-ranks time (--strong splits the 1-GPU batch over the ranks).

Workload of the headline line = BASELINE configs[1] ("C2"): B=256, S=64, D=64, K=4 NICE-FF steps (H=64), bf16.
With the default --config the other BASELINE configs (C3, C4, C5) are timed afterwards with fewer steps and embedded
in the same JSON line as "other_configs" (steps, ms/step and the flow kernels' rooflines; no CPU leg).

Roofline accounting: algorithmic flops / bytes are counted over REAL tokens (padding tokens of a ragged
batch do no algorithmic work: the kernels skip or zero them); side outputs a design writes for itself are traffic, not
algorithmic bytes -- every entry carries `wasted_traffic` = PMC traffic / algorithmic bytes when a PMC pass exists.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable (float4 copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0
# dense MFMA peaks by compute dtype (MI355X_MICROARCH.md: bf16 ~2.5 PF; f32-input MFMA = the vector peak)
MFMA_PEAK_TFLOPS = {"bf16": MFMA_BF16_PEAK_TFLOPS, "fp32": 157.3, "f32": 157.3}

CONFIGS = {
    #        B    S    D    H    K   dtype   ragged
    "c1": (32, 32, 64, 64, 4, "f32", False),
    "c2": (256, 64, 64, 64, 4, "bf16", False),
    "c3": (256, 128, 256, 256, 16, "bf16", False),
    "c4": (256, 64, 64, 64, 4, "bf16", False),      # autoregressive family: latent (B, D), LSTM encoder / LSTM-cell decoder
    "c5": (64, 512, 512, 512, 32, "bf16", True),
}
FAMILY = {"c4": "auto"}        # everything else: "nonauto"
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


def auto_model_config(D, H, K, S):
    """BASELINE config 4: AutoModel = LSTM encoder, `auto-recurrent` LSTM-cell decoder with teacher forcing
    1.0, sentence-level latent (B, D) through K NICE-FF steps (auto_lm/model.py:19-232, auto_lm/priors/flow_prior.py)."""
    return dict(
        type="auto",
        embedder=dict(embedding_dim=D),
        encoder=dict(type="recurrent", core=dict(type="lstm", input_size=D, hidden_size=D)),
        decoder=dict(type="auto-recurrent", cell=dict(type="lstm", input_size=D, hidden_size=D),
                     embedder=dict(embedding_dim=D), max_timesteps=S, teacher_forcing_ratio=1.0),
        posterior=dict(type="default", input_size=D, features=D, samples=1),
        prior=dict(type="flow", features=D, flows=[
            dict(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H],
                                            activation="gelu", dropout=0.0)) for _ in range(K)]),
        kl_loss=dict(type="default"),
        recon_scheduler=dict(type="constant", weight=1.0),
        kl_scheduler=dict(type="constant", weight=1.0),
    )


def config_model_params(name):
    B, S, D, H, K, _, _ = CONFIGS[name]
    return auto_model_config(D, H, K, S) if FAMILY.get(name) == "auto" else model_config(D, H, K)


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


def _time_calls(calls, reps, graph):
    """Average GPU duration of each entry point: captured `inner` times back to back into a HIP graph through the C ABI
    and the graph replayed between two HIP events on the launch stream, so no host launch gap is included."""
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
        del g
    return out


def flow_kernel_times(model, tokens, dtype, reps=50, graph=True, family="nonauto", loss_kernels=True):
    """Average GPU duration of the dominant hand-written launches (reconstruction NLL forward / backward over the
    (N, V) logits, fused NICE stack forward, its backward, the weight-gradient kernel pair, the posterior's
    reparameterisation kernels) on the bench shape.  family "auto": the latent is (B, D) = one token per sentence."""
    import nonauto_lm_amd as M
    from nonauto_lm_amd import _lib as L, ops
    from nonauto_lm_amd.flows import fused_stack_spec
    dev = tokens.device
    B, S_tok = tokens.shape
    S = 1 if family == "auto" else S_tok
    prior = model._prior
    D = prior._features
    h = D // 2
    z = torch.randn(B, S, D, device=dev).to(dtype)
    mask = ops.tokens_mask(tokens) if family != "auto" else None
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
    calls = {}
    if loss_kernels and family != "auto":
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
            calls.update({
                "lce_fwd": lambda: L.call("mivae_lce_fwd", L.ptr(hdec), L.ptr(w16), L.ptr(tokens), L.ptr(w),
                                          L.ptr(per_seq), L.ptr(aux[0]), L.ptr(aux[1]), L.ptr(lce_ws), B, S, E, V, 0.0, L.stream()),
                "lce_bwd": lambda: L.call("mivae_lce_bwd", L.ptr(hdec), L.ptr(w16), L.ptr(tokens), L.ptr(aux[0]),
                                          L.ptr(aux[1]), L.ptr(dper), L.ptr(dl16), B, S, E, V, 0.0, L.stream()),
            })
            # the training step's route: d W / d b from mivae_lce_dw (recomputed tiles), so the token-major pass writes no
            # (N, V) gradient; without that kernel (E > 64) the pass also stores d logits for a library product
            dw_slabs = L.query("mivae_lce_dw_slabs", N, E, V) if ops.LCE_RECOMPUTE_DW else 0
            dl_arg = None if dw_slabs else dl16
            calls["lce_bwd_dh"] = lambda: L.call("mivae_lce_bwd_dh", L.ptr(hdec), L.ptr(w16), L.ptr(wt16), L.ptr(tokens),
                                                 L.ptr(aux[0]), L.ptr(aux[1]), L.ptr(dper), L.ptr(dl_arg), L.ptr(dh_ws),
                                                 L.ptr(dh16), B, S, E, V, 0.0, L.stream())
            if dw_slabs:
                dw_part = torch.empty(dw_slabs, V, E + 8, dtype=torch.float32, device=dev)
                calls["lce_dw"] = lambda: L.call("mivae_lce_dw", L.ptr(hdec), L.ptr(w16), L.ptr(tokens), L.ptr(aux[0]),
                                                 L.ptr(aux[1]), L.ptr(dper), L.ptr(dw_part), B, S, E, V, 0.0, L.stream())
        calls.update({
            "nll_fwd": lambda: L.call("mivae_nll_fwd", L.ptr(logits), L.ptr(tokens), L.ptr(w), L.ptr(per_seq), L.ptr(aux[0]),
                                      L.ptr(aux[1]), L.ptr(aux[2]), B, S, V, 0.0, code, L.stream()),
            "nll_bwd": lambda: L.call("mivae_nll_bwd", L.ptr(logits), L.ptr(tokens), L.ptr(aux[0]), L.ptr(aux[1]), L.ptr(dper),
                                      L.ptr(dlogits), B, S, V, 0.0, code, L.stream()),
        })
    if big:      # one entry point each: forward (k_big_chain<FWD>), backward (chain / nets / weight-gradient kernels)
        calls.update({
            "nice_stack_fwd": lambda: L.call("mivae_nice_big_fwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask),
                                             L.ptr(pk_f), L.ptr(pk_b), L.ptr(tok), B, 1, S, D, H, K, actid, 0, 1, 0.0, 1.0,
                                             L.stream()),
            "nice_stack_bwd": lambda: L.call("mivae_nice_big_bwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask),
                                             L.ptr(ptab), L.ptr(pk_c), L.ptr(pk_n), L.ptr(pk_b), L.ptr(dz_out), L.ptr(dlog[:B]),
                                             L.ptr(dz_in), L.ptr(dsdt), L.ptr(hid), L.ptr(ws), L.ptr(dW1), L.ptr(db1), L.ptr(dW2),
                                             L.ptr(db2), None, B, 1, S, D, H, K, actid, 1, 0.0, 1.0, L.stream()),
        })
    else:
        calls.update({
            "nice_stack_fwd": lambda: L.call("mivae_nice_stack_fwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask),
                                             L.ptr(packed), L.ptr(tok), B, 1, S, D, H, K, actid, 0, 1, 0.0, 1.0, code, L.stream()),
            "nice_stack_bwd": lambda: L.call("mivae_nice_stack_bwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask),
                                             L.ptr(packed), L.ptr(dz_out), L.ptr(dlog[:B]), L.ptr(dz_in), L.ptr(Gt), L.ptr(Ht),
                                             L.ptr(dSt), L.ptr(x0t), B, 1, S, D, H, K, actid, 1, 0.0, 1.0, code, L.stream()),
            "nice_dw": lambda: L.call("mivae_nice_dw", L.ptr(Gt), L.ptr(Ht), L.ptr(dSt), L.ptr(x0t), L.ptr(ws), L.ptr(dW1),
                                      L.ptr(db1), L.ptr(dW2), L.ptr(db2), N, h, H, K, code, L.stream()),
        })
    # posterior: sigma head + sample + log q (a3 tail, a4, a5), in-kernel noise -- the other HBM-bound kernel of the path
    mu_t, pre_t = torch.randn(B, S, D, device=dev).to(dtype), torch.randn(B, S, D, device=dev).to(dtype)
    zq, sgq, tokq, used = e(B, S, D), e(B, S, D), e(N, dt=torch.float32), torch.empty(2, dtype=torch.int64, device=dev)
    dmu, dsg = e(B, S, D), e(B, S, D)
    dlq = torch.randn(B, device=dev)
    state = ops.philox_state(dev)
    calls.update({
        "reparam_fwd": lambda: L.call("mivae_posterior_fwd", L.ptr(mu_t), L.ptr(pre_t), 1, 1e-4, 5.0, None, L.ptr(state),
                                      L.ptr(used), L.ptr(mask), L.ptr(zq), L.ptr(sgq), L.ptr(tokq), B, 1, S, D, code,
                                      L.stream()),
        "reparam_bwd": lambda: L.call("mivae_posterior_bwd", L.ptr(dz_out), L.ptr(dlq), 0, L.ptr(pre_t), 1, 1e-4, 5.0, None,
                                      L.ptr(used), L.ptr(mask), L.ptr(dmu), L.ptr(dsg), B, 1, S, D, code, L.stream()),
    })
    return _time_calls(calls, reps, graph)


def hbm_copy_bandwidth(dev, mib=1024, reps=10, mode=1):
    """GB/s (read + write bytes) of libmivae's plain streaming copy kernel (non-temporal 16-byte accesses) on this GPU:
    the measured denominator.  1 GiB per buffer, so that the 256 MiB Infinity Cache cannot serve the reads (a 256 MiB copy
    reads 6.4 TB/s on this box, a 1 GiB one 5.5 TB/s; torch's copy_ 5.4 TB/s)."""
    from nonauto_lm_amd import _lib as L
    n = mib << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    b = torch.empty_like(a)
    fn = lambda: L.call("mivae_hbm_copy", L.ptr(a), L.ptr(b), n, mode, L.stream())
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pmc_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    produced by tools/pmc_traffic.py / tools/collect_profiles.sh: separate FETCH_SIZE / WRITE_SIZE runs, gfx950
    corrections applied); None if that kernel/config was not collected.  Not measured in this run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(config, {}).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def kernel_rooflines(name, model, tokens, kern, hbm_measured):
    """Roofline blocks for the timed entry points of config `name` (see the module docstring for the accounting)."""
    from nonauto_lm_amd import ops
    B, S_tok, D, H, K, dtype_name, _ = CONFIGS[name]
    auto = FAMILY.get(name) == "auto"
    S = 1 if auto else S_tok
    e = 2 if dtype_name == "bf16" else 4
    N, h = B * S, D // 2
    n_real = N if auto else int((tokens != 0).sum().item())
    dev = tokens.device
    fused_lce = (not auto) and dtype_name == "bf16" and model.training_outputs_only and \
        ops.linear_nll_supported(torch.empty(1, D, device=dev), model._vocab_projection.weight)
    w_bytes = K * 2 * (2 * h * H + H + h) * e
    mask_bytes = 0 if auto else 8 * N
    big = "nice_dw" not in kern and "nice_stack_bwd" in kern
    # algorithmic bytes per launch: the fused K-step forward reads z and writes z' once (+ int64 mask,
    # per-token value, weights); the whole backward (input AND weight gradients) reads z', dz', writes dz, reads W and
    # writes dW: 3 N D e + 2 W.  Padding rows are neither read nor computed: real tokens only.
    algo = {
        "nice_stack_fwd": 2 * n_real * D * e + mask_bytes + 4 * N + w_bytes,
        "nice_stack_bwd_total": 3 * n_real * D * e + mask_bytes + 4 * N + 2 * w_bytes,
        "nll_fwd": n_real * VOCAB * e + 16 * N + 12 * N,
        "nll_bwd": (n_real + N) * VOCAB * e + 16 * N,
        "lce_fwd": N * D * 2 + VOCAB * D * 2 + 32 * N,
        "lce_bwd": N * VOCAB * 2 + N * D * 2 + VOCAB * D * 2 + 16 * N,
        # with mivae_lce_dw in the step the token-major pass stores no (N, V) gradient: h, W, W^T in, d h out
        "lce_bwd_dh": (0 if "lce_dw" in kern else N * VOCAB * 2) + N * D * 2 + 2 * VOCAB * D * 2 + 16 * N + N * D * 2,
        "lce_dw": N * D * 2 + VOCAB * D * 2 + 16 * N + VOCAB * (D + 1) * 4,
        # mu, pre-sigma in, z and sigma out (noise drawn in-kernel) + per-token log q; backward: dz, pre in, d mu, d pre out
        "reparam_fwd": 4 * n_real * D * e + mask_bytes + 4 * N,
        "reparam_bwd": 4 * n_real * D * e + mask_bytes,
    }
    flops = {"nll_fwd": 0, "nll_bwd": 0, "lce_fwd": 2 * N * VOCAB * D, "lce_bwd": 2 * N * VOCAB * D,
             # logits recompute + the d h (d W) product: 2 N V E flops each
             "lce_bwd_dh": 4 * N * VOCAB * D, "lce_dw": 4 * N * VOCAB * D, "reparam_fwd": 0, "reparam_bwd": 0,
             # one step = 2 nets x 2 GEMMs x 2 h H flops per REAL token; the backward's algorithmic count is twice the
             # forward's (d input + d weight per GEMM); recomputed activations are not algorithmic
             "nice_stack_fwd": 4 * K * n_real * D * H, "nice_stack_bwd_total": 8 * K * n_real * D * H}
    in_step = {"nll_fwd", "nll_bwd"} if not fused_lce else {"lce_fwd", "lce_bwd_dh" if ops.lce_fused_dh(D) else "lce_bwd"}
    if fused_lce and "lce_dw" in kern:
        in_step.add("lce_dw")
    mfma_peak = MFMA_PEAK_TFLOPS[dtype_name]

    def roof(label, pmc_keys, us_k, by, fl, note=None):
        traffic = None
        vals = [pmc_traffic(name, k) for k in pmc_keys]
        if vals and all(v is not None for v in vals):
            traffic = sum(vals)
        blk = {"kernel": label, "avg_us": round(us_k, 2), "algorithmic_bytes": by, "algorithmic_flops": fl,
               "traffic": traffic, "wasted_traffic": None if traffic is None else round(traffic / by, 2),
               "hbm_GBps": round(by / (us_k * 1e-6) / 1e9, 1), "mfma_TFLOPs": round(fl / (us_k * 1e-6) / 1e12, 2)}
        if fl / (mfma_peak * 1e12) > by / (HBM_PEAK_GBS * 1e9):
            ach = fl / (us_k * 1e-6) / 1e12
            blk.update({"bound": "mfma", "achieved": round(ach, 1), "peak": mfma_peak, "unit": "TFLOP/s",
                        "frac": round(ach / mfma_peak, 4)})
        else:
            ach = by / (us_k * 1e-6) / 1e9
            blk.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4),
                        "frac_of_measured_copy": None if not hbm_measured else round(ach / hbm_measured, 4)})
        if note:
            blk["note"] = note
        return blk

    skip = {"nll_fwd", "nll_bwd", "lce_fwd", "lce_bwd", "lce_bwd_dh", "lce_dw"} - in_step
    blocks = {}
    for k in kern:
        if k in ("nice_stack_bwd", "nice_dw") or k in skip or k not in algo:
            continue
        note = None
        if k == "nice_stack_fwd" and big and h <= 128:
            # PMC (profiles/r02/c3_flow_pmc.json): MFMA busy 26 %, VALU 52 % -- the erf-GELU on the VALU, not the matrix
            # pipe, paces this shape; the issue model caps the MFMA fraction at ~0.54 here
            blk = roof(k, [k], kern[k]["avg_us"], algo[k], flops[k],
                       "VALU-issue bound at D/2 <= 128 (fp32 erf-GELU beside the MFMAs): issue-model ceiling ~0.54 of the MFMA peak")
            blk["bound"] = "valu-issue"
            blk["issue_model_ceiling_frac"] = 0.54
            blocks[k] = blk
        else:
            blocks[k] = roof(k, [k], kern[k]["avg_us"], algo[k], flops[k], note)
    if "nice_stack_bwd" in kern:
        pair = "nice_dw" in kern
        us = kern["nice_stack_bwd"]["avg_us"] + (kern["nice_dw"]["avg_us"] if pair else 0.0)
        blocks["nice_stack_bwd"] = roof("nice_stack_bwd" + ("+nice_dw" if pair else ""),
                                        ["nice_stack_bwd"] + (["nice_dw"] if pair else []), us,
                                        algo["nice_stack_bwd_total"], flops["nice_stack_bwd_total"])
    return blocks, in_step, n_real


def run_config(name, steps, warmup, args, rank, world, dev, hbm_measured, want_cpu):
    """Build the model of config `name`, time `steps` training steps after `warmup`, then the kernel rooflines."""
    import nonauto_lm_amd as M
    from nonauto_lm_amd import dp
    from nonauto_lm_amd.step import TrainStep
    B, S, D, H, K, dtype_name, ragged = CONFIGS[name]
    if args.strong:                                # strong scaling: the 1-GPU batch split over the ranks
        if B % world:
            sys.exit(f"bench.py --strong: batch {B} does not split over {world} ranks")
        B //= world
    auto = FAMILY.get(name) == "auto"
    amp_dtype = torch.bfloat16 if dtype_name == "bf16" else None
    torch.manual_seed(13)                          # identical init on every rank (train_worker.py:48)
    model = M.VAELmModel.from_params(vocab=M.SimpleVocab(VOCAB), **config_model_params(name)).to(dev).train()
    model.training_outputs_only = True             # skip probs/preds, unused by the loss
    dp.broadcast_parameters(model)
    reducer = dp.GradReducer(model.parameters(), bucket_bytes=args.bucket_mb << 20, flatten_params=not args.torch_adam)
    # the stock packed-sequence LSTM of the C4 encoder reads the lengths on the host (nn/recurrent.py:28-35): a device ->
    # host sync inside forward, which a HIP-graph capture cannot contain -> that config steps eagerly
    eager = args.eager or auto
    if args.torch_adam:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, foreach=True, capturable=not eager)
    else:                                          # clip + Adam + zero_grad fused over the flat buckets (libmivae)
        opt = dp.FusedAdam(reducer, lr=1e-4, bf16_shadow=amp_dtype == torch.bfloat16, model=model)
    scalars = dp.PackedScalars(["batch-loss", "batch-kl", "batch-nll", "done"], dev)
    tokens = synthetic_tokens(B, S, ragged, 13 + rank).to(dev)     # per-rank shard of the global batch
    train_step = TrainStep(model, reducer, opt, tokens, amp_dtype, 5.0, 2.0, graph=not eager)

    def step():
        info = train_step()
        scalars.reduce({"batch-loss": info["batch-loss"], "batch-kl": info["batch-kl"],
                        "batch-nll": info["batch-nll"], "done": 0.0})
        return info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        step()
        marks[i + 1].record()                      # per-step GPU time without a host sync (for the median)
    barrier()
    elapsed = time.perf_counter() - t0
    loss = scalars.means()["batch-loss"]
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    res = {"value": round(world * B * steps / elapsed, 1), "ms_per_step": round(1e3 * elapsed / steps, 3),
           "ms_per_step_median": round(per[len(per) // 2], 3), "steps": steps, "warmup": warmup,
           "step_mode": "hipgraph-replay" if train_step.graph_a is not None else "eager",
           "allreduce": ("none (1 rank)" if world == 1 else "captured in graph A, overlapped with backward"
                         if train_step.comm_in_graph else "between the graphs" if train_step.graph_a is not None
                         else "from the grad hooks, overlapped with backward"),
           "final_loss": round(loss, 4), "dtype": dtype_name,
           "workload": (f"{name}: AR VAE (LSTM encoder, LSTM-cell decoder, teacher forcing 1.0) B={B}/GPU S={S} latent (B, {D}) "
                        f"K={K} NICE-FF H={H} V={VOCAB}, fwd+bwd+allreduce+clip+Adam" if auto else
                        f"{name}: NAR VAE B={B}/GPU S={S} D={D} K={K} NICE-FF H={H} V={VOCAB} "
                        f"2+2 transformer layers{', ragged lengths U[S/4, S]' if ragged else ''}, fwd+bwd+allreduce+clip+Adam"),
           "global_batch": world * B}
    if rank == 0:
        kern = flow_kernel_times(model, tokens, torch.bfloat16 if dtype_name == "bf16" else torch.float32,
                                 family="auto" if auto else "nonauto", loss_kernels=not auto)
        blocks, in_step, n_real = kernel_rooflines(name, model, tokens, kern, hbm_measured)
        res["n_real_tokens"] = n_real
        res["kernels"] = {k: round(v["avg_us"], 2) for k, v in kern.items()}
        res["roofline_flow"] = [blocks[k] for k in ("nice_stack_fwd", "nice_stack_bwd", "reparam_fwd", "reparam_bwd")
                                if k in blocks]
        cand = [k for k in blocks if k in in_step or k.startswith("nice_stack")]
        if cand:
            dom = max(cand, key=lambda k: blocks[k]["avg_us"])
            res["roofline"] = blocks[dom]
        res["kernel_blocks"] = blocks
        if want_cpu and not auto:
            res["cpu_baseline"] = cpu_baseline(model, tokens, D, H, K)
    # drop everything this config holds on the GPU before the next one is built
    train_step.graph_a = train_step.graph_b = None
    del train_step, opt, reducer, model, scalars
    gc.collect()
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)    # ~0.3 s timed window at C2
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="only the named config (no other_configs block)")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--torch-adam", action="store_true", help="stock torch.optim.Adam + separate clip passes")
    ap.add_argument("--no-sdpa", action="store_true", help="explicit masked-softmax attention in the bracket")
    ap.add_argument("--eager", action="store_true", help="no HIP-graph replay of the step (hook-overlapped all-reduce)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the config's 1-GPU batch is split over the ranks (default: weak, fixed per-GPU batch)")
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
    M._lib.lib()                                   # the HIP extension is mandatory
    if args.no_sdpa:
        M.nn_bracket.USE_FUSED_ATTENTION = False

    hbm_measured = hbm_copy_bandwidth(dev) if rank == 0 else None
    main_res = run_config(args.config, args.steps, args.warmup, args, rank, world, dev, hbm_measured,
                          want_cpu=not args.no_cpu_baseline and world == 1)
    others = {}
    if args.config == "c2" and not args.no_others:
        # the other BASELINE configs, driver-run in the same line: fewer steps (their steps are 5-15x longer), no CPU leg.
        # A failure here must not cost the headline line.
        for other in ("c3", "c4", "c5"):
            try:
                r = run_config(other, max(20, min(args.steps, 60)), max(3, min(args.warmup, 10)), args, rank, world, dev,
                               hbm_measured, want_cpu=False)
                r.pop("kernel_blocks", None)
                others[other] = r
            except Exception as err:                       # noqa: BLE001 -- reported in the line, never fatal
                others[other] = {"error": f"{type(err).__name__}: {err}"[:300]}
                if world > 1:                              # ranks must agree on what ran: stop the extras everywhere
                    break

    if rank == 0:
        B, S, D, H, K, dtype_name, ragged = CONFIGS[args.config]
        line = {
            "metric": "train sentences/sec (ELBO fwd+bwd)", "value": main_res["value"],
            "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "ms_per_step_median": main_res["ms_per_step_median"],
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": main_res["workload"], "global_batch": main_res["global_batch"], "seq_len": S, "latent_dim": D,
                       "flow_steps": K, "parallelism": f"dp{world}"},
            "roofline": main_res.get("roofline"), "roofline_flow": main_res.get("roofline_flow"),
            "hbm_measured_GBps": None if hbm_measured is None else round(hbm_measured, 1),
            "hbm_measured_how": "libmivae mivae_hbm_copy (16 B / lane non-temporal streaming copy), 1 GiB, read + write bytes, this run",
            "traffic_source": "profiles/pmc_traffic.json: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same "
                              "entry points (not measured in this run)",
            "kernels": main_res.get("kernel_blocks"),
            "kernel_timing": ("each entry point (C ABI) captured 10x back to back in a HIP graph, the graph replayed 5x between "
                              "two HIP events on the launch stream; same process, bench shape, trained weights; flops / bytes "
                              "counted over real (non-padding) tokens"),
            "n_real_tokens": main_res.get("n_real_tokens"),
            "step_mode": main_res["step_mode"], "allreduce": main_res["allreduce"],
            "cpu_baseline": main_res.get("cpu_baseline"), "final_loss": main_res["final_loss"],
            "other_configs": others,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
