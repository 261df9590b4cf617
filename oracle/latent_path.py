"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch CPU restatement (plain PyTorch ops, fp32 by default, fp64 on request) of the
latent hot path of Nemexur/nonauto-lm: posterior reparameterised sampling + log q, the flow stack
(forward / inverse with masked log-det), base log p, KL / free bits and the ELBO tail.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / reported baseline.  The product path (`nonauto-lm_amd/`) never
imports it and has no CPU fallback.

Parity status: PINNED.  Every function below is checked in `tests/test_oracle_golden.py` against
`tests/golden/*.npz`, which were produced by importing the real reference from /root/reference in
the build container (`oracle/refgen/gen_golden.py`, committed).  The reference ships no tests or
fixtures of its own, so those recorded outputs are the only pin.

Citations are file:line in /root/reference.
"""
import math
from typing import List, Optional, Sequence, Tuple

import torch

LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------- integer helpers
def tokens_mask(tokens: torch.Tensor) -> torch.Tensor:
    """int64 0/1 padding mask, id 0 = pad.  vae_lm/nn/utils.py:12-14."""
    return (tokens != 0).to(torch.int64)


def last_token_index(mask: torch.Tensor) -> torch.Tensor:
    """Index of the last non-pad token per row.  vae_lm/models/base/encoders/encoder.py:36."""
    return mask.sum(dim=1).to(torch.int64) - 1


def gather_last(output: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """ctx = output[b, last_idx[b]].  encoder.py:35-37."""
    idx = last_token_index(mask)
    return output[torch.arange(mask.size(0)), idx]


def prefix_mask(lengths: Sequence[int], batch: int) -> torch.Tensor:
    """Float prefix mask from lengths.  nonauto_lm/priors/prior.py:111-122."""
    lens = torch.as_tensor(list(lengths), dtype=torch.int64)
    if lens.numel() == 1:
        lens = lens.expand(batch)
    pos = torch.arange(int(lens.max()))
    return (pos[None, :] < lens[:, None]).to(torch.float32)


# ----------------------------------------------------------------------------- activations
def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """x * 0.5 * (1 + erf(x / sqrt 2)).  vae_lm/nn/activation.py:16."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def swish(x: torch.Tensor) -> torch.Tensor:
    """activation.py:23."""
    return x * torch.sigmoid(x)


ACTIVATIONS = {
    # subset of activation.py:59-77 the fused kernels implement
    "linear": lambda x: x,
    "gelu": gelu_erf,
    "swish": swish,
    "relu": torch.relu,
    "tanh": torch.tanh,
    "sigmoid": torch.sigmoid,
    "elu": torch.nn.functional.elu,
}


# ----------------------------------------------------------------------------- MADE / MAF / IAF (8f-2)
def made_masks(input_size: int, hidden_sizes: Sequence[int], permute_order: bool = True, seed: int = 13):
    """Connectivity masks of MADE, one (in, out) 0/1 array per MaskedLinear.  made.py:93-118.

    numpy RandomState(seed): input order = permutation (or arange), hidden degrees =
    randint(min(previous degrees), input_size - 1); hidden masks use <=, the output mask uses < and is
    replicated for the two (s, t) output groups."""
    import numpy as np
    rng = np.random.RandomState(seed)
    deg = {-1: rng.permutation(input_size) if permute_order else np.arange(input_size)}
    n = len(hidden_sizes)
    for layer in range(n):
        deg[layer] = rng.randint(deg[layer - 1].min(), input_size - 1, size=hidden_sizes[layer])
    masks = [deg[layer - 1][:, None] <= deg[layer][None, :] for layer in range(n)]
    last = deg[n - 1][:, None] < deg[-1][None, :]
    masks.append(np.concatenate([last, last], axis=1))
    return [torch.from_numpy(m.astype("float32")) for m in masks]


def made_forward(z, weights, biases, weight_masks, activation: str = "elu"):
    """MADE.forward: Linear(x, W * weight_mask, b) + activation per layer, no activation after the
    last.  made.py:39-40, :73-83, :124-127.  `weight_masks` are the (out, in) buffers of the state dict."""
    act = ACTIVATIONS[activation]
    x = z
    for i, (w, b, m) in enumerate(zip(weights, biases, weight_masks)):
        x = x @ (w * m).t() + b
        if i + 1 < len(weights):
            x = act(x)
    return x


def maf_forward(z, made, parity: bool, mask=None):
    """MAFlow.forward (density direction, one MADE pass).  maf.py:29-45."""
    D = z.size(-1)
    st = made(z)
    s, t = st[..., :D], st[..., D:]
    out = z * torch.exp(s) + t
    if parity:
        out = out.flip(dims=(-1,))
    log_det = s.sum(dim=-1)
    if mask is not None:
        out = out * mask.unsqueeze(-1)
        log_det = (log_det * mask).sum(dim=-1)
    return out, log_det


def maf_sequential(z, made, parity: bool, mask=None):
    """MAFlow.backward (sampling direction; IAFlow's training direction, iaf.py:21-23): D sequential MADE
    passes in NATURAL feature order whatever the MADE ordering is, log-det accumulated with +s.
    maf.py:47-62.  For 2-D input the reference builds a (B, D) log-det (maf.py:52) that only broadcasts when
    B == D; the intended per-row sum is returned here."""
    D = z.size(-1)
    x = torch.zeros_like(z)
    log_det = torch.zeros(z.shape[:-1], dtype=z.dtype)
    zz = z.flip(dims=(-1,)) if parity else z
    cols = []
    for i in range(D):
        st = made(x)
        s, t = st[..., :D], st[..., D:]
        xi = (zz[..., i] - t[..., i]) * torch.exp(-s[..., i])
        cols.append(xi)
        x = torch.cat([torch.stack(cols, dim=-1), torch.zeros_like(z[..., i + 1:])], dim=-1)
        log_det = log_det + s[..., i]
    if mask is not None:
        x = x * mask.unsqueeze(-1)
        log_det = (log_det * mask).sum(dim=-1)
    return x, log_det


# ----------------------------------------------------------------------------- coupling MLP
def ff_coupling(x: torch.Tensor, w_hidden: List[torch.Tensor], b_hidden: List[torch.Tensor],
                w_out: torch.Tensor, b_out: torch.Tensor, activation: str = "gelu") -> torch.Tensor:
    """FeedForwardCoupling with dropout 0: act(Linear) per hidden layer, then Linear.

    couplings.py:50-51 + nn/feedforward.py:93-98.  Weights are torch.nn.Linear layout (out, in).
    """
    act = ACTIVATIONS[activation]
    for w, b in zip(w_hidden, b_hidden):
        x = act(x @ w.t() + b)
    return x @ w_out.t() + b_out


class NiceParams:
    """Weights of one NICE step with feed-forward s/t nets (state-dict order of the reference)."""

    def __init__(self, s_w, s_b, s_wo, s_bo, t_w, t_b, t_wo, t_bo, activation="gelu"):
        self.s_w, self.s_b, self.s_wo, self.s_bo = list(s_w), list(s_b), s_wo, s_bo
        self.t_w, self.t_b, self.t_wo, self.t_bo = list(t_w), list(t_b), t_wo, t_bo
        self.activation = activation

    @classmethod
    def from_state_dict(cls, sd, prefix, activation="gelu", dtype=torch.float32):
        """Keys `<prefix>_s_net._feedforward._linear_layers.{i}.weight` ...."""
        def net(which):
            ws, bs, i = [], [], 0
            while f"{prefix}{which}._feedforward._linear_layers.{i}.weight" in sd:
                ws.append(torch.as_tensor(sd[f"{prefix}{which}._feedforward._linear_layers.{i}.weight"]).to(dtype))
                bs.append(torch.as_tensor(sd[f"{prefix}{which}._feedforward._linear_layers.{i}.bias"]).to(dtype))
                i += 1
            wo = torch.as_tensor(sd[f"{prefix}{which}._feedforward._output.weight"]).to(dtype)
            bo = torch.as_tensor(sd[f"{prefix}{which}._feedforward._output.bias"]).to(dtype)
            return ws, bs, wo, bo
        return cls(*net("_s_net"), *net("_t_net"), activation=activation)

    def tensors(self):
        return [*self.s_w, *self.s_b, self.s_wo, self.s_bo, *self.t_w, *self.t_b, self.t_wo, self.t_bo]


def nice_affine(z: torch.Tensor, s: torch.Tensor, t: torch.Tensor,
                mask: Optional[torch.Tensor], inverse: bool = False):
    """Affine tail of NICECoupling given s, t.  couplings.py:151-162 (fwd), :172-183 (inverse).

    Halves are never swapped between steps (couplings.py:185-188).
    """
    h = z.size(-1) // 2
    x0, x1 = z[..., :h], z[..., h:]
    if not inverse:
        y1 = torch.exp(s) * x1 + t
        log_det = s.sum(dim=-1)
    else:
        y1 = (x1 - t) * torch.exp(-s)
        log_det = (-s).sum(dim=-1)
    out = torch.cat([x0, y1], dim=-1)
    if mask is not None:
        out = out * mask.unsqueeze(-1)
        log_det = (log_det * mask).sum(dim=-1)
    return out, log_det


def nice_step(z, p: NiceParams, mask=None, inverse=False):
    """NICECoupling.forward / .backward with feed-forward nets.  couplings.py:141-183."""
    h = z.size(-1) // 2
    x0 = z[..., :h]
    s = ff_coupling(x0, p.s_w, p.s_b, p.s_wo, p.s_bo, p.activation)
    t = ff_coupling(x0, p.t_w, p.t_b, p.t_wo, p.t_bo, p.activation)
    return nice_affine(z, s, t, mask, inverse)


# ----------------------------------------------------------------------------- actnorm
def actnorm_init(z: torch.Tensor, log_scale: torch.Tensor, bias: torch.Tensor):
    """Data-dependent init, padding included, unbiased std.  actnorm.py:22-28."""
    flat = z.reshape(-1, z.size(-1))
    std = flat.std(dim=0)
    mean = flat.mean(dim=0)
    inv_std = 1.0 / (std + 1e-6)
    return log_scale + inv_std.log(), (bias - mean) * inv_std


def actnorm_step(z, log_scale, bias, mask=None, inverse=False):
    """ActNorm.forward (actnorm.py:31-39) / .backward (:45-54).  log_det has shape (1,) when the
    input is 2-D or mask is None, (B,) otherwise (token count is an integer sum of the mask)."""
    if not inverse:
        out = z * log_scale.exp() + bias
        if mask is not None:
            out = out * mask.unsqueeze(-1)
        log_det = log_scale.sum(dim=-1, keepdim=True)
    else:
        out = z - bias
        if mask is not None:
            out = out * mask.unsqueeze(-1)
        out = out / (log_scale.exp() + 1e-13)
        log_det = (-log_scale).sum(dim=-1, keepdim=True)
    if out.dim() > 2 and mask is not None:
        log_det = log_det * mask.reshape(mask.size(0), -1).sum(dim=1)
    return out, log_det


# ----------------------------------------------------------------------------- invertible linear
def invlinear_step(z, weight, mask, inverse=False, num_heads=1):
    """InvertibleLinearFlow (invertible_flows.py:26-44) and, with num_heads>1,
    InvertibleMultiHeadFlow (:61-83).  No mask multiply on the output; log_det = slogdet * count
    (* heads).  The inverse uses an fp64 matrix inverse cast back to fp32 (:37, :74)."""
    w = weight
    if inverse:
        w = torch.inverse(weight.double()).to(weight.dtype)
    if num_heads > 1:
        lead = z.shape[:-1]
        zz = z.reshape(*lead, num_heads, z.size(-1) // num_heads)
        out = (zz @ w.t()).reshape(*lead, z.size(-1))
    else:
        out = z @ w.t()
    log_det = torch.linalg.slogdet(w)[1]
    if z.dim() > 2:
        log_det = log_det * (mask.reshape(mask.size(0), -1).sum(dim=1) * num_heads)
    return out, log_det


# ----------------------------------------------------------------------------- base distribution
def gaussian_log_prob(z, mu: float, std: float):
    """Normal(mu, std).log_prob(z) elementwise (torch.distributions.Normal)."""
    var = std * std
    return -((z - mu) ** 2) / (2.0 * var) - math.log(std) - 0.5 * LOG_2PI


def masked_base_log_prob(z, mask, mu=0.0, std=1.0):
    """DefaultPrior.masked_log_prob.  nonauto_lm/priors/prior.py:148-158; auto prior.py:93."""
    lp = gaussian_log_prob(z, mu, std)
    if mask is not None:
        lp = lp * mask.unsqueeze(-1)
    return lp.reshape(lp.size(0), -1).sum(dim=1)


def flow_prior_forward(z0, steps, mask=None, mu=0.0, std=1.0):
    """FlowPrior.forward_pass.  nonauto flow_prior.py:37-52, auto flow_prior.py:36-50.

    `steps` is a list of callables (z, mask, inverse) -> (z, log_det).  Returns (z_K, log p).
    """
    z = z0
    total = z0.new_zeros(z0.size(0))
    for step in steps:
        z, ld = step(z, mask, False)
        total = total + ld
    return z, masked_base_log_prob(z, mask, mu, std) + total


def flow_prior_inverse(eps, steps, mask=None, mu=0.0, std=1.0):
    """FlowPrior.backward_pass (sampling direction).  nonauto flow_prior.py:54-70."""
    z = eps
    log_prob = masked_base_log_prob(eps, mask, mu, std)
    total = eps.new_zeros(eps.size(0))
    for step in reversed(list(steps)):
        z, ld = step(z, mask, True)
        total = total + ld
    return z, log_prob + total


def nice_steps(params: List[NiceParams]):
    return [(lambda z, m, inv, p=p: nice_step(z, p, m, inv)) for p in params]


# ----------------------------------------------------------------------------- posterior
def sigma_head(pre: torch.Tensor) -> torch.Tensor:
    """Hardtanh(Softplus(pre), 1e-4, 5).  posterior.py:23-27."""
    return torch.clamp(torch.nn.functional.softplus(pre), min=1e-4, max=5.0)


def posterior_sample(mu, sigma, eps, mask, samples: int):
    """Posterior.sample with injected eps.  nonauto posterior.py:119-133; auto :111-119.

    mu, sigma: (B, [S,] D); eps: (B, n, [S,] D); mask (B, S) int64 or None (auto).
    Returns z ((B n), [S,] D) and the repeated mask."""
    z = mu.unsqueeze(1) + eps * sigma.unsqueeze(1)
    z = z.reshape(-1, *z.shape[2:])
    if mask is not None:
        mask = mask.repeat_interleave(samples, dim=0)
        z = z * mask.unsqueeze(-1)
    return z, mask


def posterior_log_prob(z, mu, sigma, mask, samples: int):
    """Posterior.log_probability.  nonauto posterior.py:157-172; auto :141-156."""
    mu_r = mu.repeat_interleave(samples, dim=0)
    sg_r = sigma.repeat_interleave(samples, dim=0)
    lp = -0.5 * ((z - mu_r).pow(2) * sg_r.pow(2).reciprocal() + 2 * sg_r.log() + LOG_2PI)
    if mask is not None:
        lp = lp * mask.unsqueeze(-1)
    return lp.reshape(lp.size(0), -1).sum(dim=1)


def prior_sample_from_noise(noise, lengths: Sequence[int], batch: int, samples: int, steps=None, mu=0.0, std=1.0):
    """DefaultPrior.sample / FlowPrior.sample with the base draw injected.  nonauto_lm/priors/prior.py:105-131,
    flow_prior.py:15-24.  noise (batch, samples, max_len, D) = base_dist.sample((batch, samples, max_len)).
    Returns (z ((B n), L, D), log_prob (B n), mask ((B n), L)); steps = None: the base sample itself."""
    mask = prefix_mask(lengths, batch)
    eps = noise * mask[:, None, :, None]
    eps = eps.reshape(batch * samples, *eps.shape[2:])
    mask = mask.repeat_interleave(samples, dim=0)
    if not steps:
        return eps, masked_base_log_prob(eps, mask, mu, std), mask
    z, logp = flow_prior_inverse(eps, steps, mask, mu, std)
    return z, logp, mask


def mutual_info(z, log_prob, mask, mu, sigma, samples: int):
    """Posterior.calc_mutual_info.  nonauto_lm/posteriors/posterior.py:174-222: log q(z) of the aggregate posterior by
    comparing every latent code with every sentence's (mu, sigma), logsumexp - log(B n)."""
    mu_r = mu.repeat_interleave(samples, dim=0)
    sg_r = sigma.repeat_interleave(samples, dim=0)
    dens = -0.5 * ((z.unsqueeze(1) - mu_r).pow(2) * sg_r.pow(2).reciprocal() + 2 * sg_r.log() + LOG_2PI)
    dens = dens * mask.to(dens.dtype)[:, None, :, None]
    dens = dens.reshape(dens.size(0), dens.size(1), -1).sum(-1)
    log_qz = torch.logsumexp(dens, dim=-1) - math.log(mu_r.size(0))
    return (log_prob - log_qz).mean()


# ----------------------------------------------------------------------------- KL / ELBO tail
def kl_default(log_q, log_p, free_bits_alpha: Optional[float] = None):
    """DefaultKLLoss.  nn/kl_loss.py:38-49, free bits :14-21."""
    kl = log_q - log_p
    if free_bits_alpha is not None:
        kl = kl.clamp(min=free_bits_alpha)
    return kl


def infovae_kl(log_q, log_p, latent, prior_z, prior_std: float, alpha: float, reg_weight: float,
               kernel_type: str = "rbf", free_bits_alpha: Optional[float] = None):
    """InfoVAEKLLoss.forward with the prior draw injected.  nn/kl_loss.py:75-149 (direct (B, B, F) form).

    Returns dict(loss, mmd-part, kl-part)."""
    def kernel(a, b):
        a = a.unsqueeze(1)                                               # kl_loss.py:114
        z_dim = b.size(-1) if a.dim() == 3 else b.size(-2) + b.size(-1)   # :127 / :142 (sic)
        c = 2.0 * z_dim * prior_std
        diff2 = (a - b).pow(2).reshape(a.size(0), a.size(0), -1)
        if kernel_type == "rbf":
            return torch.exp(-diff2.mean(-1) / c)                        # :130-132
        k = c / (1e-13 + c + diff2.sum(-1))                              # :143-145
        return k.sum(-1) - k.diag()                                      # :148

    batch = log_q.size(0)
    kl = log_q - log_p
    mmd = kernel(prior_z, prior_z).mean(-1) + kernel(latent, latent).mean(-1) - 2 * kernel(prior_z, latent).mean(-1)
    loss = (1.0 - alpha) * kl + (alpha + reg_weight - 1.0) / (batch * (batch - 1)) * mmd
    if free_bits_alpha is not None:
        loss = loss.clamp(min=free_bits_alpha)
    return {"loss": loss, "mmd-part": mmd, "kl-part": kl}


def elbo_tail(recon, kl, batch: int, samples: int, recon_weight=1.0, kl_weight=1.0, iwae=False):
    """ELBO assembly of VAELmModel.forward.  models/base/model.py:156-193.

    recon, kl: (B*n,).  Returns dict with batch-loss, batch-kl, batch-nll (scalars)."""
    recon = recon.view(batch, samples)
    kl = kl.view(batch, samples)
    loss = recon_weight * recon + kl_weight * kl
    if iwae:
        w = loss.softmax(dim=-1)
        loss = (loss * w).sum(dim=-1)
    return {"batch-loss": loss.mean(), "batch-kl": kl.mean(), "batch-nll": recon.mean()}


def label_smoothing_nll(logits, target, weights, smoothing: float = 0.0, training: bool = True):
    """LabelSmoothingNLL(size_average=False).  models/base/losses.py:128-169."""
    v = logits.size(-1)
    logp = torch.log_softmax(logits.reshape(-1, v), dim=-1)
    tgt = target.reshape(-1, 1).long()
    if smoothing is not None and smoothing > 0.0 and training:
        q = torch.full_like(logp, smoothing / v)
        q.scatter_add_(-1, tgt, torch.full_like(tgt, 1.0 - smoothing, dtype=logp.dtype))
        nll = -(logp * q).sum(-1, keepdim=True)
    else:
        nll = -logp.gather(1, tgt)
    nll = nll.view(*target.shape) * weights
    denom = weights.reshape(weights.size(0), -1).sum(1).clamp(min=1e-13)
    return nll.reshape(nll.size(0), -1).sum(1) / denom
