"""ELBO assembly (VAELmModel.forward, vae_lm/models/base/model.py:120-211) and the
non-autoregressive model (vae_lm/models/nonauto_lm/model.py) around the HIP latent path.

Output dictionary keys, `loss_info` keys and state-dict keys follow the reference.  Differences that
do not change results: metrics are accumulated on the device and read back only in `get_metrics`
(the reference forces >= 5 device->host syncs and 10 scalar all-reduces per forward);
under torch.distributed they are reduced in one packed all-reduce by `dp.PackedScalars`.
"""
from typing import Dict, Iterable, List, Optional

import torch

from . import ops
from .latent import DefaultKLLoss, KLLoss, LatentSample, Posterior, PosteriorSample, Prior, PriorSample
from .nn_bracket import Decoder, Embedder, Encoder, EncoderOutput, LabelSmoothingNLL
from .registrable import Registrable


class WeightScheduler(Registrable):
    """nn/weight_scheduler.py: host-side scalar weights for the reconstruction and KL terms."""

    @property
    def weight(self) -> float:
        raise NotImplementedError()

    def step(self) -> None:
        pass


@WeightScheduler.register("constant")
class ConstantWeightScheduler(WeightScheduler):
    def __init__(self, weight: float = 1.0) -> None:
        self._weight = weight

    @property
    def weight(self) -> float:
        return self._weight


@WeightScheduler.register("linear")
class LinearWeightScheduler(WeightScheduler):
    """weight_scheduler.py:33-60."""

    def __init__(self, zero_weight_steps: int, annealing_steps: int) -> None:
        self._step = 0
        self._weight = 0.0
        self._zero_weight_steps = zero_weight_steps
        self._annealing_steps = annealing_steps

    @property
    def weight(self) -> float:
        return self._weight

    def step(self) -> None:
        self._step += 1
        if not (self._zero_weight_steps > 0 and self._step <= self._zero_weight_steps):
            self._weight = min(1.0, (self._step - self._zero_weight_steps) / self._annealing_steps)


class RunningMean:
    """Device-side running sum / count; `value()` is the only host sync."""

    def __init__(self):
        self.total, self.count = None, 0

    def update(self, v):
        v = v.detach().float() if isinstance(v, torch.Tensor) else float(v)
        self.total = v if self.total is None else self.total + v
        self.count += 1

    def value(self, reset=False):
        tot = self.total
        out = 0.0 if self.count == 0 else float(tot.item() if isinstance(tot, torch.Tensor) else tot) / self.count
        if reset:
            self.total, self.count = None, 0
        return out


class VAELmModel(torch.nn.Module, Registrable):
    def __init__(self, vocab, recon_scheduler: WeightScheduler, kl_scheduler: WeightScheduler,
                 kl_loss: KLLoss, iwae: bool = False, recon_weight: float = 1.0,
                 label_smoothing: float = 0.0) -> None:
        super().__init__()
        self.vocab = vocab
        self._iwae = iwae
        self._recon_scheduler = recon_scheduler
        self._kl_scheduler = kl_scheduler
        self._kl_loss = kl_loss
        self._recon_weight = recon_weight
        self._loss = LabelSmoothingNLL(label_smoothing, size_average=False)
        self._metrics = {k: RunningMean() for k in ("nll", "avg-recon-weight", "avg-kl-weight", "avg-kl")}
        self._end_index = self.vocab.token_to_index("<eos>")
        self.training_outputs_only = False   # skip probs/preds (unused by the loss) when True
        self.track_metrics = True            # False under HIP-graph capture (host-side accumulators)
        self.device_weights = None           # optional fp32 device tensor {recon weight, kl weight}, see forward()

    @property
    def is_kl_used(self) -> bool:
        return self._kl_scheduler.weight != 0

    def forward(self, src_tokens, tgt_tokens=None, manual_kl_step: bool = False, random: bool = True,
                eps: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        batch = src_tokens.size(0)
        n = self.nsamples_posterior
        encoded = self.encode(src_tokens)
        # Launch-count pass over the latent path: with the default KL the per-sentence reductions of log q / log p, the
        # KL and the ELBO tail are ONE kernel each way (ops.latent_tail) fed by the per-token values of the posterior
        # and prior kernels; any other combination takes the module-by-module route below.  Same arithmetic, same order.
        tok_logq = None
        if (tgt_tokens is not None and type(self._kl_loss) is DefaultKLLoss and src_tokens.is_cuda
                and hasattr(self._posterior, "sample_tokens")):
            x_post, m_post = self._posterior_inputs(encoded)
            tail_ok = ops.latent_tail_supported(batch, n, x_post.size(1) if x_post.dim() == 3 else 1)
            # the generator's step after an in-kernel draw rides in the ELBO tail kernel (no launch of its own)
            z, _, tok_logq = self._posterior.sample_tokens(x_post, m_post, random=random, eps=eps, defer_rng=tail_ok)
            latent, posterior_log_prob = LatentSample(z, self._posterior._mu, self._posterior._sigma), None
            if not tail_ok:
                posterior_log_prob = ops._SegmentSum.apply(tok_logq, tok_logq.size(0), tok_logq.size(1))
                tok_logq = None
        else:
            latent, posterior_log_prob = self.sample_from_posterior(encoded, random=random, eps=eps)
        tgt_mask = None
        if tgt_tokens is not None:
            # the usual call reconstructs the source: one mask kernel, and no repeat copies for a single sample
            tgt_mask = encoded.mask if tgt_tokens is src_tokens else ops.tokens_mask(tgt_tokens)
            if n != 1:
                tgt_mask = tgt_mask.repeat_interleave(n, dim=0)
                tgt_tokens = tgt_tokens.repeat_interleave(n, dim=0)
        decoded = {k: v.view(batch, n, *v.shape[1:])
                   for k, v in self.decode(latent.z, tgt_mask, target=tgt_tokens).items()}
        tok_logp = None
        if tok_logq is not None:
            tok_logp = self._prior_token_log_prob(latent, tgt_mask)
            if tok_logp is None:            # a prior without a per-token form: reduce log q now, go module by module
                posterior_log_prob = ops._SegmentSum.apply(tok_logq, tok_logq.size(0), tok_logq.size(1))
                if self._posterior.rng_pending:
                    ops.rng_advance(tok_logq.device)
        if tok_logp is None:
            prior_log_prob = self._get_prior_log_prob(latent, tgt_mask)
        if tgt_tokens is None:
            return {"source": src_tokens, "latent": latent.z, **decoded}
        recon_error = decoded.pop("loss")                                  # (B, n)
        w_r, w_kl = self._recon_scheduler.weight, self._kl_scheduler.weight
        # device_weights (set by step.TrainStep): the two weights live in device memory, written by the host before
        # every step, so that a HIP-graph replay of this forward does not freeze them at their capture-time values
        if tok_logp is not None:
            kl_out = {}
            (batch_loss, batch_kl, batch_nll), _ = ops.latent_tail(
                tok_logq, tok_logp, recon_error, batch, n, self._kl_loss._free_bits_alpha, w_r, w_kl, self._iwae,
                weights_dev=self.device_weights, rng_advance=self._posterior.rng_pending)
        else:
            kl_out = {f"batch-{k}": v.view(batch, n)
                      for k, v in self._kl_loss(posterior_log_prob, prior_log_prob, latent=latent.z).items()}
            kl_loss = kl_out.pop("batch-loss")
            batch_loss, batch_kl, batch_nll = ops.elbo_tail(recon_error, kl_loss, batch, n, w_r, w_kl, self._iwae,
                                                            weights_dev=self.device_weights)
        if not manual_kl_step and self.training:
            self.kl_scheduler_step()
        if self.training:
            self._recon_scheduler.step()
        # the reported weights are the schedulers' AFTER their step (models/base/model.py:180-189, 204-208)
        w_r, w_kl = self._recon_scheduler.weight, self._kl_scheduler.weight
        out = {
            "loss_info": {"recon-weight": w_r, "kl-weight": w_kl, "batch-kl": batch_kl,
                          "batch-nll": batch_nll, "batch-loss": batch_loss,
                          **{k: v.mean() for k, v in kl_out.items()}},
            "source": src_tokens, "target": tgt_tokens, "latent": latent.z, **decoded,
        }
        if self.track_metrics:
            self.update_metrics(batch_nll, batch_kl, w_r, w_kl)
        return out

    def update_metrics(self, batch_nll, batch_kl, w_r, w_kl) -> None:
        """models/base/model.py:200-210 (device-side running sums; no host sync)."""
        self._metrics["nll"].update(batch_nll)
        self._metrics["avg-recon-weight"].update(w_r)
        self._metrics["avg-kl-weight"].update(w_kl)
        self._metrics["avg-kl"].update(batch_kl)

    def kl_scheduler_step(self) -> None:
        self._kl_scheduler.step()

    def get_metrics(self, reset: bool = False) -> Dict[str, float]:
        import math
        nll = self._metrics["nll"].value(reset)
        return {"ppl": math.exp(nll) if nll != 0 else 0.0, "avg-nll": nll,
                **{k: self._metrics[k].value(reset) for k in ("avg-recon-weight", "avg-kl-weight", "avg-kl")}}

    def sample(self, samples: int, lengths: List[int] = None):
        prior_sample = self.sample_from_prior(samples, lengths)
        return self.decode(prior_sample.latent, prior_sample.mask), prior_sample.log_prob


@VAELmModel.register("nonauto")
class NonAutoModel(VAELmModel):
    """vae_lm/models/nonauto_lm/model.py:20-225."""

    def __init__(self, vocab, embedder: Embedder, encoder: Encoder, decoder: Decoder, posterior: Posterior,
                 prior: Prior, kl_loss: KLLoss, recon_scheduler: WeightScheduler,
                 kl_scheduler: WeightScheduler, iwae: bool = False, label_smoothing: float = 0.0) -> None:
        super().__init__(vocab=vocab, kl_loss=kl_loss, recon_scheduler=recon_scheduler,
                         kl_scheduler=kl_scheduler, iwae=iwae, label_smoothing=label_smoothing)
        self._embedder = embedder
        self._encoder = encoder
        self._decoder = decoder
        self._posterior = posterior
        self._prior = prior
        self._vocab_projection = torch.nn.Linear(self._decoder.get_output_size(),
                                                 vocab.get_vocab_size(namespace="target"))

    @property
    def nsamples_posterior(self) -> int:
        return self._posterior.samples

    def set_samples(self, samples: int) -> None:
        self._posterior.samples = samples

    def encode(self, tokens) -> EncoderOutput:
        return self._encoder(self._embedder(tokens), ops.tokens_mask(tokens))

    def decode(self, z, mask, target=None):
        """`mask` is the token mask of `target` when a target is given (forward() passes it that way)."""
        decoded = self._decoder(z, mask)
        if (self.training_outputs_only and target is not None and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16
                and ops.linear_nll_supported(decoded, self._vocab_projection.weight)):
            # training step under bf16 autocast: projection + NLL fused, the (N, V) logits are never written
            # (SURVEY 8f-1); `logits` / `probs` / `preds` are not produced in this mode
            proj = self._vocab_projection
            weights = ops.mask_f32(mask if mask is not None and mask.shape == target.shape else ops.tokens_mask(target))
            smoothing = self._loss._smoothing if (self._loss._smoothing and self.training) else 0.0
            return {"loss": ops.linear_nll(decoded, proj.weight, proj.bias, target, weights, smoothing,
                                           getattr(proj.weight, "_mivae_lp", None))}
        logits = self._vocab_projection(decoded)
        out = {"logits": logits}
        if not (self.training_outputs_only and target is not None):
            out["probs"] = torch.softmax(logits, dim=-1)
            out["preds"] = torch.argmax(out["probs"], dim=-1)
        if target is not None:
            weights = ops.tokens_mask(target).float()
            out["loss"] = self._loss(logits, target, weights=weights)
        return out

    def sample_from_prior(self, samples: int, lengths: List[int] = None) -> PriorSample:
        return PriorSample(*self._prior.sample(samples, lengths))

    def sample_from_posterior(self, encoded: EncoderOutput, random: bool = True, eps=None) -> PosteriorSample:
        return PosteriorSample(*self._posterior(encoded.output, encoded.mask, random=random, eps=eps))

    def _get_prior_log_prob(self, z: LatentSample, mask=None):
        return self._prior.log_probability(z, mask)

    def _posterior_inputs(self, encoded: EncoderOutput):
        return encoded.output, encoded.mask

    def _prior_token_log_prob(self, z: LatentSample, mask=None):
        fn = getattr(self._prior, "token_log_prob", None)
        return None if fn is None else fn(z, mask)

    def calc_mutual_info(self, src_tokens, random: bool = True):
        encoded = self.encode(src_tokens)
        latent, log_q = self.sample_from_posterior(encoded, random=True)
        mask = encoded.mask.repeat_interleave(self.nsamples_posterior, dim=0)
        return self._posterior.calc_mutual_info(latent.z, log_q, mask=mask)

    def encoder_parameters(self) -> Iterable[torch.nn.Parameter]:
        from itertools import chain
        return chain(self._encoder.parameters(), self._posterior.parameters())

    def decoder_parameters(self) -> Iterable[torch.nn.Parameter]:
        return self._decoder.parameters()

    @classmethod
    def from_params(cls, vocab, **params):
        params = dict(params)
        embedder = Embedder.from_params(vocab=vocab, **params.pop("embedder"))
        encoder = Encoder.from_params(**params.pop("encoder"))
        decoder = Decoder.from_params(**params.pop("decoder"))
        posterior = Posterior.from_params(**params.pop("posterior"))
        prior = Prior.from_params(**params.pop("prior"))
        kl_loss = KLLoss.from_params(prior=prior, **params.pop("kl_loss"))
        recon_scheduler = WeightScheduler.from_params(**params.pop("recon_scheduler"))
        kl_scheduler = WeightScheduler.from_params(**params.pop("kl_scheduler"))
        return cls(vocab=vocab, embedder=embedder, encoder=encoder, decoder=decoder, posterior=posterior,
                   prior=prior, kl_loss=kl_loss, recon_scheduler=recon_scheduler,
                   kl_scheduler=kl_scheduler, **params)


class SimpleVocab:
    """Stand-in for torch_nlp_utils Vocabulary: sizes and the few reserved indices the model reads."""

    def __init__(self, size: int):
        self._size = size

    def token_to_index(self, token, namespace=None):
        return {"<eos>": 2, "<sos>": 1, "@@UNKNOWN@@": 3}.get(token, 3)

    def get_vocab_size(self, namespace=None):
        return self._size
