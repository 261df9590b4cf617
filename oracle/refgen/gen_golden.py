"""Golden-vector generator: runs the REAL reference (read-only, /root/reference) on CPU and writes
small .npz fixtures (inputs, weights, outputs, gradients) to tests/golden/.

Runs only in the build container (the GPU box has no /root/reference):
    PYTORCH_JIT=0 python oracle/refgen/gen_golden.py

`PYTORCH_JIT=0`: torch 2.10 cannot deepcopy the reference's torch.jit.script'ed `gelu`
(couplings.py:138 deep-copies the coupling); with the JIT off the same Python expression
(activation.py:16) runs eagerly.  `vae_lm.training` is imported first to dodge the reference's
circular import.  Fixtures are data only: no reference source text is stored.
"""
import os
import sys

os.environ.setdefault("PYTORCH_JIT", "0")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import plumbing_stubs  # noqa: E402

plumbing_stubs.install()

import vae_lm.training  # noqa: E402,F401  (must come first, F10)
from vae_lm.models.base.flows import ActNorm, NICECoupling  # noqa: E402
from vae_lm.models.base.flows.couplings import (  # noqa: E402
    AttentionCoupling, FeedForwardCoupling, RecurrentCoupling)
from vae_lm.nn.recurrent import LSTMCore, GRUCore  # noqa: E402
from vae_lm.models.base.flows.invertible_flows import (  # noqa: E402
    InvertibleLinearFlow, InvertibleMultiHeadFlow)
from vae_lm.models.base.flows.made import MADE  # noqa: E402
from vae_lm.models.base.flows.maf import MAFlow  # noqa: E402
from vae_lm.models.base.flows.iaf import IAFlow  # noqa: E402
from vae_lm.models.base import LatentSample  # noqa: E402
from vae_lm.models.base.losses import LabelSmoothingNLL  # noqa: E402
from vae_lm.models.base.model import VAELmModel  # noqa: E402
import vae_lm.models.nonauto_lm  # noqa: E402,F401
import vae_lm.models.auto_lm  # noqa: E402,F401
from vae_lm.models.nonauto_lm.priors import FlowPrior as NAFlowPrior  # noqa: E402
from vae_lm.models.nonauto_lm.posteriors import DefaultPosterior as NAPosterior  # noqa: E402
from vae_lm.models.auto_lm.priors import FlowPrior as AFlowPrior  # noqa: E402
from vae_lm.models.auto_lm.posteriors import DefaultPosterior as APosterior  # noqa: E402
from vae_lm.nn.kl_loss import DefaultKLLoss, InfoVAEKLLoss  # noqa: E402
from vae_lm.nn.utils import get_tokens_mask  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"{name}: {len(arrays)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def make_mask(kind, B, S, gen):
    if kind == "full":
        return torch.ones(B, S, dtype=torch.int64)
    if kind == "ragged":
        lens = torch.randint(1, S + 1, (B,), generator=gen)
        lens[0] = 1          # length-1 edge case
        lens[-1] = S
        return (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
    if kind == "holes":      # arbitrary (non-prefix) 0/1 mask
        m = (torch.rand(B, S, generator=gen) > 0.35).to(torch.int64)
        m[:, 0] = 1
        return m
    raise ValueError(kind)


def state(module, prefix=""):
    # the sinusoid tables (`_pos_enc`, 5000 x E constants, transformer.py:184-193) are recomputable
    return {prefix + k: v.detach().clone() for k, v in module.state_dict().items()
            if "_pos_enc" not in k}


def grads(module, prefix="grad."):
    return {prefix + k: p.grad for k, p in module.named_parameters() if p.grad is not None}


def jitter(module, gen, scale=0.15):
    """Move weights off the default init so s is not ~0 (makes exp/log-det tests meaningful)."""
    with torch.no_grad():
        for p in module.parameters():
            if not p.requires_grad:      # frozen sinusoid tables (`_pos_enc`) are constants, not weights
                continue
            p.add_(scale * torch.randn(p.shape, generator=gen))


# ----------------------------------------------------------------------------- MAF / IAF / MADE (8f-2)
def gen_maf(name, kind, B, S, D, hidden, K, mask_kind, seed, permute=True, activation="elu"):
    """K stacked MAFlow / IAFlow steps (maf.py:8-67, iaf.py:6-23) with alternating parity.  `S=None` records the
    2-D (auto) case, training direction only: the reference's sequential direction builds a (B, D) log-det for
    2-D input (maf.py:52) and fails to broadcast unless B == D."""
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    cls = MAFlow if kind == "maf" else IAFlow
    flows = torch.nn.ModuleList(
        [cls(MADE(D, list(hidden), permute_order=permute, activation=activation), parity=bool(k % 2))
         for k in range(K)])
    jitter(flows, gen, scale=0.1)
    shape = (B, D) if S is None else (B, S, D)
    z0 = torch.randn(*shape, generator=gen, requires_grad=True)
    mask = None if S is None else make_mask(mask_kind, B, S, gen)
    out = {"z0": z0, "activation": np.array(activation), "kind": np.array(kind), "K": np.int64(K),
           "permute": np.int64(permute), "hidden": np.asarray(hidden, dtype=np.int64)}
    if mask is not None:
        out["mask"] = mask
    out.update(state(flows, "sd."))
    if S is None and kind == "iaf":
        raise ValueError("2-D IAF training direction is not runnable in the reference")
    z, total = z0, 0.0
    for k, flow in enumerate(flows):
        z, ld = flow(z, mask)
        out[f"fwd.z{k}"], out[f"fwd.ld{k}"] = z, ld
        total = total + ld
    cz = torch.randn(z.shape, generator=gen)
    cl = torch.randn(total.shape, generator=gen)
    out["cot.z"], out["cot.ld"] = cz, cl
    ((z * cz).sum() + (total * cl).sum()).backward()
    out["grad.z0"] = z0.grad
    out.update(grads(flows, "grad.sd."))
    if S is not None or kind == "iaf":
        eps = torch.randn(*shape, generator=gen)
        if mask is not None:
            eps = eps * mask.unsqueeze(-1)
        out["inv.eps"] = eps
        with torch.no_grad():
            zz = eps
            for k, flow in enumerate(list(flows)[::-1]):
                zz, ld = flow.backward(zz, None if mask is None else mask.float())
                out[f"inv.z{k}"], out[f"inv.ld{k}"] = zz, ld
    save(name, **out)


# ----------------------------------------------------------------------------- NICE / FlowPrior
def gen_nice(name, B, S, D, H, K, mask_kind, seed, activation="gelu", hidden_layers=1):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    flows = [NICECoupling(FeedForwardCoupling(D // 2, [H] * hidden_layers, activation=activation))
             for _ in range(K)]
    prior = NAFlowPrior(D, flows)
    jitter(prior, gen)
    # make s_net != t_net (the reference deep-copies one module into both, F11)
    z0 = torch.randn(B, S, D, generator=gen, requires_grad=True)
    mask = make_mask(mask_kind, B, S, gen)
    out = {"z0": z0, "mask": mask, "activation": np.array(activation)}
    out.update(state(prior, "sd."))
    # per-step forward
    z = z0
    for k, flow in enumerate(prior._flows):
        z, ld = flow(z, mask)
        out[f"fwd.z{k}"], out[f"fwd.ld{k}"] = z, ld
    zK, logp = prior.forward_pass(z0, mask)
    out["fwd.zK"], out["fwd.logp"] = zK, logp
    cz = torch.randn(zK.shape, generator=gen)
    cl = torch.randn(logp.shape, generator=gen)
    out["cot.z"], out["cot.logp"] = cz, cl
    ((zK * cz).sum() + (logp * cl).sum()).backward()
    out["grad.z0"] = z0.grad
    out.update(grads(prior, "grad.sd."))
    # inverse direction (sampling): per-step flow.backward in reversed order + backward_pass
    eps = torch.randn(B, S, D, generator=gen) * mask.unsqueeze(-1)
    out["inv.eps"] = eps
    with torch.no_grad():
        z = eps
        for k, flow in enumerate(list(prior._flows)[::-1]):
            z, ld = flow.backward(z, mask.float())
            out[f"inv.z{k}"], out[f"inv.ld{k}"] = z, ld
        zi, lpi = prior.backward_pass(eps, mask.float())
        out["inv.zK"], out["inv.logp"] = zi, lpi
        # round trip z0 -> zK -> z0
        back, _ = prior.backward_pass(zK.detach(), mask.float())
        out["inv.roundtrip"] = back
    save(name, **out)


def gen_nice_auto(name, B, D, H, K, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    flows = [NICECoupling(FeedForwardCoupling(D // 2, [H])) for _ in range(K)]
    prior = AFlowPrior(D, flows, mu=0.1, std=1.3)
    jitter(prior, gen)
    z0 = torch.randn(B, D, generator=gen, requires_grad=True)
    out = {"z0": z0, "mu": np.float32(0.1), "std": np.float32(1.3)}
    out.update(state(prior, "sd."))
    z = z0
    for k, flow in enumerate(prior._flows):
        z, ld = flow(z)
        out[f"fwd.z{k}"], out[f"fwd.ld{k}"] = z, ld
    zK, logp = prior.forward_pass(z0)
    out["fwd.zK"], out["fwd.logp"] = zK, logp
    cz = torch.randn(zK.shape, generator=gen)
    cl = torch.randn(logp.shape, generator=gen)
    out["cot.z"], out["cot.logp"] = cz, cl
    ((zK * cz).sum() + (logp * cl).sum()).backward()
    out["grad.z0"] = z0.grad
    out.update(grads(prior, "grad.sd."))
    eps = torch.randn(B, D, generator=gen)
    out["inv.eps"] = eps
    with torch.no_grad():
        # The reference's auto backward_pass calls flow.backward(z) without the mask argument that
        # NICECoupling.backward requires (auto flow_prior.py:63 vs couplings.py:164) and raises
        # TypeError; record the per-step inverse with mask=None, which is what it intends.
        z, tot = eps, torch.zeros(B)
        for flow in list(prior._flows)[::-1]:
            z, ld = flow.backward(z, None)
            tot = tot + ld
        out["inv.zK"] = z
        out["inv.logp"] = prior.base_dist.log_prob(eps).sum(-1) + tot
    save(name, **out)


def gen_nice_coupling(name, kind, B, S, D, K, mask_kind, seed):
    """a11: NICE steps whose s/t networks are attention or recurrent couplings (couplings.py:54-119)."""
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    h = D // 2
    if kind == "attention":
        make = lambda: AttentionCoupling(h, 2 * h, 1, 2)                      # noqa: E731
    elif kind == "lstm":
        make = lambda: RecurrentCoupling(LSTMCore(h, 12))                     # noqa: E731
    else:
        make = lambda: RecurrentCoupling(GRUCore(h, 10), hidden_sizes=[14])   # noqa: E731
    prior = NAFlowPrior(D, [NICECoupling(make()) for _ in range(K)])
    jitter(prior, gen, 0.05)
    z0 = torch.randn(B, S, D, generator=gen, requires_grad=True)
    mask = make_mask(mask_kind, B, S, gen)
    out = {"z0": z0, "mask": mask, "kind": np.array(kind)}
    out.update(state(prior, "sd."))
    zK, logp = prior.forward_pass(z0, mask)
    out["fwd.zK"], out["fwd.logp"] = zK, logp
    cz, cl = torch.randn(zK.shape, generator=gen), torch.randn(logp.shape, generator=gen)
    out["cot.z"], out["cot.logp"] = cz, cl
    ((zK * cz).sum() + (logp * cl).sum()).backward()
    out["grad.z0"] = z0.grad
    out.update(grads(prior, "grad.sd."))
    eps = torch.randn(B, S, D, generator=gen) * mask.unsqueeze(-1)
    out["inv.eps"] = eps
    with torch.no_grad():
        zi, lpi = prior.backward_pass(eps, mask)
        out["inv.zK"], out["inv.logp"] = zi, lpi
    save(name, **out)


def gen_infovae(name, seed):
    """a19: InfoVAEKLLoss (kl_loss.py:52-149) on 3-D and 2-D latents, rbf and imq kernels; the prior draw of
    `_compute_mmd` (kl_loss.py:102) is recorded by replaying the global RNG."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for tag, shape, prior in (("3d", (6, 5, 8), NAFlowPrior(8, [], mu=0.0, std=1.2)),
                              ("2d", (7, 12), AFlowPrior(12, [], mu=0.0, std=0.8))):
        z = torch.randn(*shape, generator=gen, requires_grad=True)
        logq = torch.randn(shape[0], generator=gen) - 8
        logp = torch.randn(shape[0], generator=gen) - 9
        out[f"{tag}.z"], out[f"{tag}.logq"], out[f"{tag}.logp"] = z, logq, logp
        out[f"{tag}.std"] = np.float32(prior.std)
        for kernel in ("rbf", "imq"):
            for fb in (None, 0.4):
                mod = InfoVAEKLLoss(alpha=0.3, reg_weight=50, prior=prior, kernel_type=kernel, free_bits_alpha=fb)
                torch.manual_seed(seed + 7)
                prior_z = prior.base_dist.sample(z.size()[:-1])
                torch.manual_seed(seed + 7)
                res = mod(logq, logp, latent=z)
                key = f"{tag}.{kernel}.fb{fb}"
                out[key + ".prior_z"] = prior_z
                for k, v in res.items():
                    out[f"{key}.{k}"] = v
                g, = torch.autograd.grad(res["loss"].sum(), z)
                out[key + ".grad_z"] = g
    save(name, **out)


# ----------------------------------------------------------------------------- ActNorm
def gen_actnorm(name, B, S, D, mask_kind, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    flow = ActNorm(D)
    out = {"pre.log_scale": flow._log_scale.detach().clone(), "pre.bias": flow._bias.detach().clone()}
    if S is None:
        z0 = (torch.randn(B, D, generator=gen) * 1.7 + 0.4).requires_grad_(True)
        mask = None
    else:
        z0 = (torch.randn(B, S, D, generator=gen) * 1.7 + 0.4).requires_grad_(True)
        mask = make_mask(mask_kind, B, S, gen)
        # reference feeds already-masked latents (posterior.py:130-132); init stats include the zeros
        z0 = (z0.detach() * mask.unsqueeze(-1)).requires_grad_(True)
        out["mask"] = mask
    out["z0"] = z0
    z1, ld1 = flow(z0, mask)          # first call: data-dependent init happens here
    out["post.log_scale"], out["post.bias"] = flow._log_scale.detach().clone(), flow._bias.detach().clone()
    out["fwd.z"], out["fwd.ld"] = z1, ld1
    cz = torch.randn(z1.shape, generator=gen)
    cl = torch.randn(ld1.shape, generator=gen)
    out["cot.z"], out["cot.ld"] = cz, cl
    ((z1 * cz).sum() + (ld1 * cl).sum()).backward()
    out["grad.z0"] = z0.grad
    out["grad.log_scale"], out["grad.bias"] = flow._log_scale.grad, flow._bias.grad
    with torch.no_grad():
        zi, ldi = flow.backward(z1.detach(), mask)
        out["inv.z"], out["inv.ld"] = zi, ldi
    save(name, **out)


# ----------------------------------------------------------------------------- InvLinear
def gen_invlinear(name, B, S, D, heads, mask_kind, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    flow = InvertibleLinearFlow(D) if heads == 1 else InvertibleMultiHeadFlow(D, heads)
    with torch.no_grad():
        flow._weight.add_(0.1 * torch.randn(flow._weight.shape, generator=gen))
    z0 = torch.randn(B, S, D, generator=gen, requires_grad=True)
    mask = make_mask(mask_kind, B, S, gen)
    out = {"z0": z0, "mask": mask, "weight": flow._weight.detach().clone(), "heads": np.int64(heads)}
    z1, ld1 = flow(z0, mask)
    out["fwd.z"], out["fwd.ld"] = z1, ld1
    cz = torch.randn(z1.shape, generator=gen)
    cl = torch.randn(ld1.shape, generator=gen)
    out["cot.z"], out["cot.ld"] = cz, cl
    ((z1 * cz).sum() + (ld1 * cl).sum()).backward()
    out["grad.z0"], out["grad.weight"] = z0.grad, flow._weight.grad
    with torch.no_grad():
        zi, ldi = flow.backward(z1.detach(), mask)
        out["inv.z"], out["inv.ld"] = zi, ldi
    save(name, **out)


# ----------------------------------------------------------------------------- mixed stack
def gen_mixed(name, B, S, D, H, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    flows = [ActNorm(D), NICECoupling(FeedForwardCoupling(D // 2, [H])), InvertibleLinearFlow(D),
             NICECoupling(FeedForwardCoupling(D // 2, [H])), ActNorm(D)]
    prior = NAFlowPrior(D, flows)
    jitter(prior, gen, 0.05)
    mask = make_mask("ragged", B, S, gen)
    z0 = (torch.randn(B, S, D, generator=gen) * mask.unsqueeze(-1)).requires_grad_(True)
    out = {"z0": z0, "mask": mask}
    out.update(state(prior, "pre.sd."))
    zK, logp = prior.forward_pass(z0, mask)     # actnorm init happens in this call
    out.update(state(prior, "sd."))
    out["fwd.zK"], out["fwd.logp"] = zK, logp
    cl = torch.randn(logp.shape, generator=gen)
    out["cot.logp"] = cl
    (logp * cl).sum().backward()
    out["grad.z0"] = z0.grad
    out.update(grads(prior, "grad.sd."))
    with torch.no_grad():
        zi, lpi = prior.backward_pass(zK.detach(), mask)
        out["inv.zK"], out["inv.logp"] = zi, lpi
    save(name, **out)


# ----------------------------------------------------------------------------- posterior
def gen_posterior(name, B, S, E, D, n, mask_kind, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    auto = S is None
    post = (APosterior if auto else NAPosterior)(E, D, samples=n)
    enc = torch.randn(*((B, E) if auto else (B, S, E)), generator=gen, requires_grad=True)
    out = {"enc": enc, "samples": np.int64(n)}
    out.update(state(post, "sd."))
    mask = None
    if not auto:
        mask = make_mask(mask_kind, B, S, gen)
        out["mask"] = mask
    # The reference draws eps from the global RNG as the first draw of forward(): replay it.
    torch.manual_seed(seed + 1000)
    shape = (B, n) if auto else (B, n, S)
    eps = torch.distributions.Normal(torch.zeros(D), torch.ones(D)).sample(shape)
    torch.manual_seed(seed + 1000)
    latent, logq = post(enc) if auto else post(enc, mask)
    out["eps"], out["z"], out["mu"], out["sigma"], out["logq"] = eps, latent.z, latent.mu, latent.sigma, logq
    z_replay = (latent.mu.unsqueeze(1) + eps * latent.sigma.unsqueeze(1)).reshape(latent.z.shape)
    if mask is not None:
        z_replay = z_replay * mask.repeat_interleave(n, 0).unsqueeze(-1)
    assert torch.equal(z_replay, latent.z), "eps replay does not reproduce the reference sample"
    cz = torch.randn(latent.z.shape, generator=gen)
    cl = torch.randn(logq.shape, generator=gen)
    out["cot.z"], out["cot.logq"] = cz, cl
    latent.mu.retain_grad(), latent.sigma.retain_grad()
    ((latent.z * cz).sum() + (logq * cl).sum()).backward()
    out["grad.enc"], out["grad.mu"], out["grad.sigma"] = enc.grad, latent.mu.grad, latent.sigma.grad
    out.update(grads(post, "grad.sd."))
    # deterministic (random=False) path
    with torch.no_grad():
        lat0, logq0 = post(enc, random=False) if auto else post(enc, mask, random=False)
        out["det.z"], out["det.logq"] = lat0.z, logq0
    save(name, **out)


# ----------------------------------------------------------------------------- prior sampling / mutual information
def gen_prior_sample(name, batch, lengths, samples, D, H, K, seed, mu=0.0, std=1.0):
    """DefaultPrior.sample / FlowPrior.sample (nonauto prior.py:105-131, flow_prior.py:15-24): the prefix mask built
    from lengths, the masked base draw, its log-prob, and (K > 0) the flow-inverted sample with its log-prob."""
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    from vae_lm.models.nonauto_lm.priors import DefaultPrior as NADefaultPrior
    if K > 0:
        flows = [NICECoupling(FeedForwardCoupling(D // 2, [H], activation="gelu")) for _ in range(K)]
        prior = NAFlowPrior(D, flows, mu=mu, std=std)
        jitter(prior, gen)
    else:
        prior = NADefaultPrior(D, mu=mu, std=std)
    out = {"batch": np.int64(batch), "lengths": np.asarray(lengths, dtype=np.int64), "samples": np.int64(samples),
           "mu": np.float64(mu), "std": np.float64(std)}
    out.update(state(prior, "sd."))
    with torch.no_grad():
        # the base draw is the first use of the global RNG inside sample(): replay it
        torch.manual_seed(seed + 2000)
        eps, base_logp, mask = NADefaultPrior.sample(prior, batch, lengths, samples)
        out["eps"], out["base.logp"], out["mask"] = eps, base_logp, mask
        torch.manual_seed(seed + 2000)
        z, logp, mask2 = prior.sample(batch, lengths, samples)
        assert torch.equal(mask, mask2)
        out["z"], out["logp"] = z, logp
    save(name, **out)


def gen_mutual_info(name, B, S, E, D, n, mask_kind, seed):
    """Posterior.calc_mutual_info (nonauto posterior.py:174-222) on the sample the posterior just drew."""
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    post = NAPosterior(E, D, samples=n)
    enc = torch.randn(B, S, E, generator=gen)
    mask = make_mask(mask_kind, B, S, gen)
    out = {"enc": enc, "mask": mask, "samples": np.int64(n)}
    out.update(state(post, "sd."))
    torch.manual_seed(seed + 1000)
    eps = torch.distributions.Normal(torch.zeros(D), torch.ones(D)).sample((B, n, S))
    torch.manual_seed(seed + 1000)
    with torch.no_grad():
        latent, logq = post(enc, mask)
        rep_mask = mask.repeat_interleave(n, 0).float()
        mi = post.calc_mutual_info(latent.z, logq, rep_mask)
    out["eps"], out["z"], out["mu"], out["sigma"], out["logq"], out["mi"] = eps, latent.z, latent.mu, latent.sigma, logq, mi
    save(name, **out)


# ----------------------------------------------------------------------------- KL / NLL
def gen_kl_nll(name, seed):
    gen = torch.Generator().manual_seed(seed)
    out = {}
    q = torch.randn(12, generator=gen) * 5 - 20
    p = torch.randn(12, generator=gen) * 5 - 22
    out["logq"], out["logp"] = q, p
    out["kl.none"] = DefaultKLLoss()(q, p)["loss"]
    out["kl.fb0.5"] = DefaultKLLoss(free_bits_alpha=0.5)(q, p)["loss"]
    out["kl.fb3"] = DefaultKLLoss(free_bits_alpha=3.0)(q, p)["loss"]
    B, S, V = 5, 7, 11
    logits = torch.randn(B, S, V, generator=gen, requires_grad=True)
    target = torch.randint(1, V, (B, S), generator=gen)
    target[0, 3:] = 0
    target[2, 1:] = 0
    weights = get_tokens_mask(target).float()
    out["nll.logits"], out["nll.target"], out["nll.weights"] = logits, target, weights
    for sm in (0.0, 0.1):
        loss_mod = LabelSmoothingNLL(sm, size_average=False)
        loss_mod.train()
        val = loss_mod(logits, target, weights=weights)
        out[f"nll.train.s{sm}"] = val
        g, = torch.autograd.grad(val.sum(), logits)
        out[f"nll.train.s{sm}.grad"] = g
        loss_mod.eval()
        out[f"nll.eval.s{sm}"] = loss_mod(logits, target, weights=weights)
    save(name, **out)


# ----------------------------------------------------------------------------- full model ELBO
def model_config(D, H, K, V, n, iwae, free_bits, label_smoothing):
    return dict(
        type="nonauto",
        embedder=dict(embedding_dim=D),
        encoder=dict(type="transformer", input_size=D, hidden_size=2 * D, num_layers=2, num_heads=4),
        decoder=dict(type="nonauto-transformer", input_size=D, hidden_size=2 * D, num_layers=2, num_heads=4),
        posterior=dict(type="default", input_size=D, features=D, samples=n),
        prior=dict(type="flow", features=D, flows=[
            dict(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H],
                                            activation="gelu")) for _ in range(K)]),
        kl_loss=dict(type="default", free_bits_alpha=free_bits),
        recon_scheduler=dict(type="constant", weight=1.0),
        kl_scheduler=dict(type="constant", weight=0.7),
        iwae=iwae,
        label_smoothing=label_smoothing,
    )


def gen_model(name, B, S, D, H, K, V, n, iwae, free_bits, label_smoothing, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    vocab = plumbing_stubs.Vocabulary(V)
    cfg = model_config(D, H, K, V, n, iwae, free_bits, label_smoothing)
    model = VAELmModel.from_params(vocab=vocab, **cfg)
    model.train()
    lens = torch.randint(2, S + 1, (B,), generator=gen)
    lens[0] = S
    tokens = torch.randint(4, V, (B, S), generator=gen)
    tokens = tokens * (torch.arange(S)[None, :] < lens[:, None])
    rec = {}
    orig_post, orig_prior = model.sample_from_posterior, model._get_prior_log_prob

    def post_hook(encoded, random=True):
        res = orig_post(encoded, random=random)
        rec["enc"] = encoded.output
        rec["logq"] = res.log_prob
        rec["z"], rec["mu"], rec["sigma"] = res.latent.z, res.latent.mu, res.latent.sigma
        return res

    def prior_hook(z, mask=None):
        res = orig_prior(z, mask)
        rec["logp"] = res
        return res

    model.sample_from_posterior, model._get_prior_log_prob = post_hook, prior_hook
    torch.manual_seed(seed + 1000)
    eps = torch.distributions.Normal(torch.zeros(D), torch.ones(D)).sample((B, n, S))
    torch.manual_seed(seed + 1000)
    outd = model(tokens, tokens)
    info = outd["loss_info"]
    info["batch-loss"].backward()
    out = {"tokens": tokens, "eps": eps, "samples": np.int64(n), "iwae": np.int64(iwae),
           "free_bits": np.float32(-1.0 if free_bits is None else free_bits),
           "label_smoothing": np.float32(label_smoothing), "kl_weight": np.float32(0.7),
           "recon_weight": np.float32(1.0), "H": np.int64(H), "K": np.int64(K), "V": np.int64(V)}
    out.update(state(model, "sd."))
    for k in ("enc", "logq", "logp", "z", "mu", "sigma"):
        out["rec." + k] = rec[k]
    out["rec.logits"] = outd["logits"]
    out["rec.preds"] = outd["preds"]
    for k in ("batch-kl", "batch-nll", "batch-loss"):
        out["info." + k] = info[k]
    out.update(grads(model, "grad.sd."))
    save(name, **out)


def auto_config(D, E, H, K, n, iwae, free_bits, label_smoothing, kl="default"):
    return dict(
        type="auto",
        embedder=dict(embedding_dim=E),
        encoder=dict(type="recurrent", core=dict(type="lstm", input_size=E, hidden_size=E)),
        decoder=dict(type="auto-recurrent", cell=dict(type="lstm", input_size=E, hidden_size=D),
                     embedder=dict(embedding_dim=E), max_timesteps=12, teacher_forcing_ratio=1.0),
        posterior=dict(type="default", input_size=E, features=D, samples=n),
        prior=dict(type="flow", features=D, flows=[
            dict(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H],
                                            activation="gelu")) for _ in range(K)]),
        kl_loss=(dict(type="default", free_bits_alpha=free_bits) if kl == "default" else
                 dict(type="info-vae", alpha=0.3, reg_weight=20, kernel_type="rbf", free_bits_alpha=free_bits)),
        recon_scheduler=dict(type="constant", weight=1.0),
        kl_scheduler=dict(type="constant", weight=0.7),
        iwae=iwae,
        label_smoothing=label_smoothing,
    )


def gen_model_auto(name, B, S, D, E, H, K, V, n, iwae, free_bits, label_smoothing, seed):
    """C4: AutoModel (auto_lm/model.py) with an LSTM encoder and the LSTM-cell decoder, teacher forcing 1.0 and no
    word dropout (so the host RNG draws of decoder.py:119-121 do not change the inputs)."""
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    vocab = plumbing_stubs.Vocabulary(V)
    model = VAELmModel.from_params(vocab=vocab, **auto_config(D, E, H, K, n, iwae, free_bits, label_smoothing))
    model.train()
    lens = torch.randint(3, S + 1, (B,), generator=gen)
    lens[0] = S
    tokens = torch.randint(4, V, (B, S), generator=gen)
    tokens[:, 0] = vocab.token_to_index("<sos>", namespace="target")
    tokens = tokens * (torch.arange(S)[None, :] < lens[:, None])
    rec = {}
    orig_post, orig_prior = model.sample_from_posterior, model._get_prior_log_prob

    def post_hook(encoded, random=True):
        res = orig_post(encoded, random=random)
        rec["ctx"], rec["logq"] = encoded.ctx, res.log_prob
        rec["z"], rec["mu"], rec["sigma"] = res.latent.z, res.latent.mu, res.latent.sigma
        return res

    def prior_hook(z, mask=None):
        res = orig_prior(z, mask)
        rec["logp"] = res
        return res

    model.sample_from_posterior, model._get_prior_log_prob = post_hook, prior_hook
    torch.manual_seed(seed + 1000)
    eps = torch.distributions.Normal(torch.zeros(D), torch.ones(D)).sample((B, n))
    torch.manual_seed(seed + 1000)
    outd = model(tokens, tokens)
    info = outd["loss_info"]
    info["batch-loss"].backward()
    out = {"tokens": tokens, "eps": eps, "samples": np.int64(n), "iwae": np.int64(iwae),
           "free_bits": np.float32(-1.0 if free_bits is None else free_bits),
           "label_smoothing": np.float32(label_smoothing), "kl_weight": np.float32(0.7),
           "recon_weight": np.float32(1.0), "H": np.int64(H), "K": np.int64(K), "V": np.int64(V),
           "E": np.int64(E), "D": np.int64(D)}
    out.update(state(model, "sd."))
    for k in ("ctx", "logq", "logp", "z", "mu", "sigma"):
        out["rec." + k] = rec[k]
    out["rec.logits"] = outd["logits"]
    out["rec.preds"] = outd["preds"]
    for k in ("batch-kl", "batch-nll", "batch-loss"):
        out["info." + k] = info[k]
    out.update(grads(model, "grad.sd."))
    save(name, **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    if only:                         # regenerate selected fixtures: python gen_golden.py name [name ...]
        _save = save

        def save(name, **arrays):    # noqa: F811
            if name in only:
                _save(name, **arrays)
    gen_nice_coupling("nice_attention", "attention", 4, 7, 16, 2, "ragged", 21)
    gen_nice_coupling("nice_lstm", "lstm", 4, 7, 16, 2, "ragged", 22)
    gen_nice_coupling("nice_gru_ff", "gru", 3, 5, 12, 2, "ragged", 23)
    gen_infovae("infovae", 24)
    gen_model_auto("model_auto_n1", 6, 9, 16, 24, 32, 2, 40, 1, False, None, 0.0, 25)
    gen_model_auto("model_auto_n2_iwae", 5, 8, 32, 16, 32, 3, 40, 2, True, 0.5, 0.1, 26)
    gen_maf("maf_tiny_ragged", "maf", 4, 6, 8, [16], 3, "ragged", 31)
    gen_maf("maf_d16_natural_holes", "maf", 3, 5, 16, [32], 2, "holes", 32, permute=False, activation="tanh")
    gen_maf("maf_d64_2layer", "maf", 4, 9, 64, [64, 48], 2, "ragged", 33, activation="relu")
    gen_maf("maf_auto_2d", "maf", 5, None, 16, [32], 2, None, 34)
    gen_maf("iaf_tiny_ragged", "iaf", 4, 6, 8, [16], 2, "ragged", 35)
    gen_maf("iaf_d16_natural", "iaf", 3, 5, 16, [24], 2, "full", 36, permute=False, activation="gelu")
    gen_prior_sample("prior_sample_default", 4, [5, 2, 7, 1], 3, 8, 0, 0, 41, mu=0.25, std=1.5)
    gen_prior_sample("prior_sample_one_length", 3, [6], 2, 8, 0, 0, 42)
    gen_prior_sample("prior_sample_flow", 4, [9, 3, 1, 6], 2, 16, 32, 3, 43)
    gen_mutual_info("mutual_info_n1", 5, 7, 12, 8, 1, "ragged", 44)
    gen_mutual_info("mutual_info_n3", 4, 6, 12, 8, 3, "holes", 45)
    if only:
        sys.exit(0)
    gen_nice("nice_tiny_full", 4, 6, 8, 16, 4, "full", 13)
    gen_nice("nice_tiny_ragged", 4, 6, 8, 16, 4, "ragged", 14)
    gen_nice("nice_tiny_holes", 4, 6, 8, 16, 4, "holes", 15)
    gen_nice("nice_d64_ragged", 8, 16, 64, 64, 4, "ragged", 13)
    gen_nice("nice_d64_full", 8, 16, 64, 64, 4, "full", 14)
    gen_nice("nice_d128_h96_holes", 3, 37, 128, 96, 3, "holes", 16)
    gen_nice("nice_relu_2layer", 4, 6, 8, 16, 2, "ragged", 17, activation="relu", hidden_layers=2)
    gen_nice_auto("nice_auto_2d", 3, 16, 16, 2, 13)
    gen_nice_auto("nice_auto_d64", 9, 64, 64, 4, 14)
    gen_actnorm("actnorm_3d_ragged", 4, 6, 8, "ragged", 13)
    gen_actnorm("actnorm_3d_d64", 8, 16, 64, "holes", 14)
    gen_actnorm("actnorm_2d", 7, None, 16, None, 15)
    gen_invlinear("invlinear_d8", 4, 6, 8, 1, "ragged", 13)
    gen_invlinear("invlinear_d64", 8, 16, 64, 1, "holes", 14)
    gen_invlinear("invmultihead_d64_h4", 8, 16, 64, 4, "ragged", 15)
    gen_mixed("mixed_stack_d16", 4, 9, 16, 32, 13)
    gen_posterior("posterior_n1_ragged", 4, 6, 12, 8, 1, "ragged", 13)
    gen_posterior("posterior_n3_holes", 4, 6, 12, 8, 3, "holes", 14)
    gen_posterior("posterior_d64_n2", 8, 16, 64, 64, 2, "ragged", 15)
    gen_posterior("posterior_auto_n2", 5, None, 12, 16, 2, None, 16)
    gen_kl_nll("kl_nll", 13)
    gen_model("model_n1", 6, 12, 32, 32, 2, 50, 1, False, None, 0.0, 13)
    gen_model("model_n3_iwae_fb", 5, 10, 32, 64, 3, 50, 3, True, 0.5, 0.1, 14)
