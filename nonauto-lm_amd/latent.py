"""Prior / Posterior / KLLoss modules of the latent hot path behind the reference's module API.

nonauto: vae_lm/models/nonauto_lm/{priors,posteriors}; auto: vae_lm/models/auto_lm/{priors,posteriors};
KL: vae_lm/nn/kl_loss.py.  The `mask`-less 2-D (auto) and masked 3-D (nonauto) variants share one
implementation: the kernels treat a 2-D latent as S = 1 without a mask.
"""
import math
from typing import Dict, List, NamedTuple, Optional, Tuple

import torch
import torch.distributions as D

from . import ops
from .flows import Flow, fused_stack_spec, mixed_stack_run, mixed_stack_train_run
from .nn_bracket import Linear, linear_group
from .registrable import Registrable


class LatentSample(NamedTuple):
    """models/base/model.py:20-25."""
    z: torch.Tensor
    mu: torch.Tensor
    sigma: torch.Tensor


class PriorSample(NamedTuple):
    latent: torch.Tensor
    log_prob: torch.Tensor
    mask: torch.Tensor


class PosteriorSample(NamedTuple):
    latent: LatentSample
    log_prob: torch.Tensor


def _device_of(module: torch.nn.Module) -> torch.device:
    for p in module.parameters():
        return p.device
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


# ----------------------------------------------------------------------------------- priors
class Prior(torch.nn.Module, Registrable):
    """nonauto priors/prior.py:10-83 (the auto variant drops `lengths` / `mask`)."""

    def __init__(self, features: int) -> None:
        super().__init__()
        self._features = features
        self.register_buffer("_device_probe", torch.empty(0), persistent=False)

    @property
    def device(self):
        return self._device_probe.device


@Prior.register("default")
class DefaultPrior(Prior):
    """N(mu, std) prior with masked log-prob and sampling (nonauto prior.py:86-158, auto :53-93)."""

    def __init__(self, features: int, mu: float = 0.0, std: float = 1.0) -> None:
        super().__init__(features)
        self.mu = mu
        self.std = std

    @property
    def base_dist(self):
        dev = self.device
        return D.Normal(loc=torch.full((self._features,), float(self.mu), device=dev),
                        scale=torch.full((self._features,), float(self.std), device=dev))

    def sample(self, batch: int, lengths: Optional[List[int]] = None, samples: int = 1):
        dev = self.device
        if lengths is None:      # auto: (B n, D) epsilon and its log-prob (auto prior.py:72-82)
            eps = self.base_dist.sample((batch, samples)).reshape(batch * samples, self._features)
            return eps, self.masked_log_prob(eps, None)
        lens = torch.as_tensor(list(lengths), dtype=torch.int64, device=dev)
        if lens.size(0) == 1:
            lens = lens.expand(batch)
        max_len = int(lens.max().item())
        # integer prefix mask from lengths (prior.py:116-122), kept 0/1
        mask = (torch.arange(max_len, device=dev)[None, :] < lens[:, None]).float()
        eps = self.base_dist.sample((batch, samples, max_len)) * mask[:, None, :, None]
        eps = eps.reshape(batch * samples, max_len, self._features)
        mask = mask.repeat_interleave(samples, dim=0)
        return eps, self.masked_log_prob(eps, mask, size_average=True), mask

    def log_probability(self, posterior_sample: LatentSample, mask: Optional[torch.Tensor] = None):
        return self.masked_log_prob(posterior_sample.z, mask, size_average=True)

    def masked_log_prob(self, latent, mask, size_average: bool = False):
        return ops.gauss_logprob(latent, mask, float(self.mu), float(self.std))

    def token_log_prob(self, posterior_sample: LatentSample, mask: Optional[torch.Tensor] = None):
        """log_probability() before its per-sentence reduction: ((B n), S) fp32, or None when this prior has no
        per-token form (the ELBO tail kernel takes the per-token values and reduces them itself)."""
        return ops.gauss_logprob_tok(posterior_sample.z, mask, float(self.mu), float(self.std))


@Prior.register("flow")
class FlowPrior(DefaultPrior):
    """FlowPrior (nonauto flow_prior.py:9-75, auto flow_prior.py:9-70).

    When every flow is a fusable NICE-FF step the K steps, the masked log-det sum and the base
    log-prob run as ONE kernel launch (+ the per-sentence reduction); otherwise each flow runs its
    own kernels in the reference's Python loop."""

    def __init__(self, features: int, flows: List[Flow], mu: float = 0.0, std: float = 1.0) -> None:
        super().__init__(features, mu, std)
        self._flows = torch.nn.ModuleList(flows)
        self.allow_fused = True

    def sample(self, batch: int, lengths: Optional[List[int]] = None, samples: int = 1):
        if lengths is None:
            eps, _ = super().sample(batch, None, samples)
            return self.backward_pass(eps, None)
        eps, _, mask = super().sample(batch, lengths, samples)
        z, log_prob = self.backward_pass(eps, mask)
        return z, log_prob, mask

    def log_probability(self, posterior_sample: LatentSample, mask: Optional[torch.Tensor] = None):
        return self.forward_pass(posterior_sample.z, mask)[1]

    def token_log_prob(self, posterior_sample: LatentSample, mask: Optional[torch.Tensor] = None):
        z0 = posterior_sample.z
        spec = fused_stack_spec(self._flows, z0) if self.allow_fused else None
        if spec is None or not (mask is not None or z0.dim() == 2):
            return None                     # flow lists the fused stack does not take: per-sentence path
        params, H, act = spec
        return ops.nice_stack_tok(z0, mask, params, H, act, inverse=False, base=True, mu=float(self.mu),
                                  std=float(self.std))[1]

    def forward_pass(self, z0, mask=None):
        spec = fused_stack_spec(self._flows, z0) if self.allow_fused else None
        if spec is not None and (mask is not None or z0.dim() == 2):
            params, H, act = spec
            return ops.nice_stack(z0, mask, params, H, act, inverse=False, base=True,
                                  mu=float(self.mu), std=float(self.std))
        mixed = mixed_stack_run(self._flows, z0, mask, False) if self.allow_fused else None
        if mixed is None and self.allow_fused:       # training: one launch each way + the parameter reductions
            mixed = mixed_stack_train_run(self._flows, z0, mask)
        if mixed is not None:       # a mixed list in one launch (no-grad evaluation, or the training path)
            z, total = mixed
            return z, self.masked_log_prob(z, mask, size_average=True) + total
        z = z0
        total = None
        for flow in self._flows:
            z, log_det = flow(z, mask)
            total = log_det if total is None else total + log_det
        log_prob = self.masked_log_prob(z, mask, size_average=True)
        return z, log_prob + total

    def backward_pass(self, z0, mask=None):
        """Sampling direction: flows reversed, each `flow.backward` (flow_prior.py:54-70)."""
        log_prob = self.masked_log_prob(z0, mask, size_average=True)
        spec = fused_stack_spec(self._flows, z0) if self.allow_fused else None
        if spec is not None and (mask is not None or z0.dim() == 2):
            params, H, act = spec
            z, total = ops.nice_stack(z0, mask, params, H, act, inverse=True, base=False)
            return z, log_prob + total
        mixed = mixed_stack_run(self._flows, z0, mask, True) if self.allow_fused else None
        if mixed is not None:
            return mixed[0], log_prob + mixed[1]
        z = z0
        total = None
        for flow in list(self._flows)[::-1]:
            z, log_det = flow.backward(z, mask)
            total = log_det if total is None else total + log_det
        return z, log_prob + total

    @classmethod
    def from_params(cls, **params):
        flows = [Flow.from_params(**flow) for flow in params.pop("flows")]
        return cls(flows=flows, **params)


# ----------------------------------------------------------------------------------- posterior
class Posterior(torch.nn.Module, Registrable):
    """Diagonal-Gaussian posterior (nonauto posterior.py:13-222, auto posterior.py:13-200)."""

    def __init__(self, input_size: int, features: int, samples: int = 1) -> None:
        super().__init__()
        self.samples = samples
        self._features = features
        self._mu_net = Linear(input_size, features)
        self._sigma_net = torch.nn.Sequential(
            Linear(input_size, features), torch.nn.Softplus(),
            torch.nn.Hardtanh(min_val=1e-4, max_val=5.0))
        self._mu = None
        self._sigma = None
        self.rng_pending = False     # an in-kernel draw whose generator step was left to the caller (sample_tokens)

    @property
    def base_dist(self):
        dev = self._mu_net.weight.device
        return D.Normal(loc=torch.zeros(self._features, device=dev), scale=torch.ones(self._features, device=dev))

    def _draw(self, mu, random: bool, eps: Optional[torch.Tensor]):
        if eps is not None:
            return eps
        if not random:
            return None                              # kernel treats a null eps as zeros
        lead = (mu.size(0), self.samples) + tuple(mu.shape[1:])
        return torch.randn(lead, device=mu.device, dtype=mu.dtype)

    def sample_tokens(self, encoded, mask, random=True, eps=None, defer_rng: bool = False):
        """(z ((B n), ...), repeated mask, per-token log q ((B n), S)): a3 + a4 + a5 as two launches -- the grouped
        mu | sigma heads, then ONE kernel for the sigma head's Hardtanh(Softplus(.)), the reparameterised sample, the
        mask multiply and the per-token log q (posterior.py:19-27, 86-172); the noise is injected (`eps`), drawn inside
        that kernel (Philox) or absent (`random=False`).  defer_rng: with in-kernel noise, leave the generator's step to the
        caller (`self.rng_pending` says whether one is due: ops.latent_tail(..., rng_advance=True) or ops.rng_advance)."""
        # a3: the two heads share `encoded` -> one grouped launch forward, one for the input gradient
        mu, pre = linear_group(encoded, [self._mu_net, self._sigma_net[0]])
        self._mu = mu
        sp, ht = self._sigma_net[1], self._sigma_net[2]
        fused_head = (pre.is_cuda and pre.dtype in (torch.float32, torch.bfloat16) and isinstance(sp, torch.nn.Softplus)
                      and sp.beta == 1 and sp.threshold == 20 and isinstance(ht, torch.nn.Hardtanh))
        if random and eps is None and not ops.PHILOX:
            eps = self._draw(mu, random, None)
        rng = bool(random and eps is None)
        self.rng_pending = rng and defer_rng
        if fused_head:
            z, self._sigma, tok = ops.posterior_tok(mu, pre, eps, mask, self.samples, True, ht.min_val, ht.max_val, rng,
                                                    advance=not defer_rng)
        else:
            z, self._sigma, tok = ops.posterior_tok(mu, ht(sp(pre)), eps, mask, self.samples, False, rng=rng,
                                                    advance=not defer_rng)
        rep_mask = mask if mask is None or self.samples == 1 else mask.repeat_interleave(self.samples, dim=0)
        return z, rep_mask, tok

    def sample_and_log_prob(self, encoded, mask, random=True, eps=None):
        z, rep_mask, tok = self.sample_tokens(encoded, mask, random, eps)
        return z, rep_mask, ops._SegmentSum.apply(tok, tok.size(0), tok.size(1))

    def calc_mutual_info(self, z, log_prob, mask=None):
        """I(x, z) estimator (posterior.py:174-222).  On the GPU the (B n)^2 pairwise log-densities come from one tiled
        kernel (ops.pair_gauss_logdens: no (B n, B n, S, D) intermediate, 1 GiB at C2); evaluation metric, no gradients."""
        n = self.samples
        if z.is_cuda and not (torch.is_grad_enabled() and (z.requires_grad or self._mu.requires_grad)):
            dens = ops.pair_gauss_logdens(z, self._mu, self._sigma, mask, n)
            log_qz = torch.logsumexp(dens, dim=-1) - math.log(dens.size(1))
            return (log_prob - log_qz).mean()
        mu = self._mu.repeat_interleave(n, 0).float()
        sigma = self._sigma.repeat_interleave(n, 0).float()
        dens = -0.5 * ((z.float().unsqueeze(1) - mu).pow(2) * sigma.pow(2).reciprocal()
                       + 2 * sigma.log() + math.log(2 * math.pi))
        if mask is not None:
            dens = dens * mask.float()[:, None, :, None]
        dens = dens.reshape(dens.size(0), dens.size(1), -1).sum(-1)
        log_qz = torch.logsumexp(dens, dim=-1) - math.log(mu.size(0))
        return (log_prob - log_qz).mean()


@Posterior.register("default")
class DefaultPosterior(Posterior):
    """forward(encoded, mask, random=True) for nonauto (posterior.py:241-251); the auto variant is
    forward(encoded, random=True) (auto posterior.py:219-227) -- pass mask=None."""

    def forward(self, encoded, mask=None, random: bool = True, eps: Optional[torch.Tensor] = None):
        if isinstance(mask, bool):        # auto call style: forward(encoded, random)
            mask, random = None, mask
        z, _, log_prob = self.sample_and_log_prob(encoded, mask, random, eps)
        return LatentSample(z, self._mu, self._sigma), log_prob

    def backward(self, z, mask=None):
        return z, ops.gauss_logprob(z, mask, 0.0, 1.0)


# ----------------------------------------------------------------------------------- KL losses
class KLLoss(torch.nn.Module, Registrable):
    def __init__(self, free_bits_alpha: float = None, **kwargs) -> None:
        super().__init__()
        self._free_bits_alpha = free_bits_alpha

    def forward(self, posterior_log_prob, prior_log_prob, **kwargs) -> Dict[str, torch.Tensor]:
        raise NotImplementedError()


@KLLoss.register("default")
class DefaultKLLoss(KLLoss):
    """nn/kl_loss.py:34-49."""

    def __init__(self, free_bits_alpha: float = None, prior=None, **kwargs) -> None:
        super().__init__(free_bits_alpha)

    def forward(self, posterior_log_prob, prior_log_prob, **kwargs):
        return {"loss": ops.kl_default(posterior_log_prob, prior_log_prob, self._free_bits_alpha)}


@KLLoss.register("info-vae")
class InfoVAEKLLoss(KLLoss):
    """InfoVAE objective (nn/kl_loss.py:52-149): (1 - alpha) KL + (alpha + reg - 1) / (B (B-1)) MMD.

    The reference materialises the (B, B, S*D) difference tensor (1 GiB at B=256, S=D=64); here the pairwise
    squared distances come from the tiled kernels of csrc/pairwise.hip (mivae_pair_sqdist / _bwd: the reference's own
    difference arithmetic, streamed through LDS, fixed summation order), no (B, B, F) intermediate.  Reference quirks kept: `z_dim` is D for 2-D latents and
    S + D for 3-D latents (kl_loss.py:127,142); the IMQ kernel returns row sums without the diagonal, whose
    `.mean(-1)` is then a scalar (kl_loss.py:106,148).  `prior_z` may be injected (tests); otherwise drawn from the
    prior's base distribution like kl_loss.py:102."""

    def __init__(self, alpha: float, reg_weight: int, prior, kernel_type: str = "rbf",
                 free_bits_alpha: float = None) -> None:
        super().__init__(free_bits_alpha)
        self._alpha = alpha
        self._reg_weight = reg_weight
        self._kernel_type = kernel_type
        # not a registered submodule in spirit (the model owns the prior), but the reference assigns it as an
        # attribute, which registers it: state-dict keys `_kl_loss._prior.*` exist there too.
        self._prior = prior

    def forward(self, posterior_log_prob, prior_log_prob, latent, prior_z=None, **kwargs):
        batch = posterior_log_prob.size(0)
        kl = posterior_log_prob - prior_log_prob
        mmd = self._compute_mmd(latent, prior_z)
        loss = (1.0 - self._alpha) * kl + (self._alpha + self._reg_weight - 1.0) / (batch * (batch - 1)) * mmd
        if self._free_bits_alpha is not None:
            loss = loss.clamp(min=self._free_bits_alpha)
        return {"loss": loss, "mmd-part": mmd, "kl-part": kl}

    def _compute_mmd(self, z, prior_z=None):
        if prior_z is None:
            prior_z = self._prior.base_dist.sample(z.size()[:-1]).to(z.device)
        z_dim = z.size(-1) if z.dim() == 2 else z.size(-2) + z.size(-1)
        a, b = prior_z.reshape(prior_z.size(0), -1).float(), z.reshape(z.size(0), -1).float()
        return (self._kernel(a, a, z_dim).mean(-1) + self._kernel(b, b, z_dim).mean(-1)
                - 2 * self._kernel(a, b, z_dim).mean(-1))

    def _kernel(self, x1, x2, z_dim):
        sq = ops.pair_sqdist(x1, x2)        # tiled HIP kernel: sum_f (x1_i - x2_j)^2, no (B, B, F) intermediate
        scale = 2.0 * z_dim * self._prior.std
        if self._kernel_type == "rbf":
            return torch.exp(-(sq / x1.size(-1)) / scale)           # mean over features, kl_loss.py:130-132
        if self._kernel_type == "imq":
            kernel = scale / (1e-13 + scale + sq)                    # kl_loss.py:143-145
            return kernel.sum(-1) - kernel.diag()
        raise Exception("Undefined kernel type.")
