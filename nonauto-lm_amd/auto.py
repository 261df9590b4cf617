"""Autoregressive variant (BASELINE config C4): sentence-level latent (B, D) through the same HIP latent path.

Mirrors vae_lm/models/auto_lm/: `AutoModel` (model.py:19-232, registry name "auto") and its LSTM-cell
decoder (decoders/decoder.py:11-160, decoders/rnn.py:12-105, registry name "auto-recurrent").  The encoder /
decoder bracket runs on stock PyTorch-ROCm ops (out of scope of SURVEY 8); what goes through libmivae is
everything between them: token mask (a1), last-token context gather (a2), reparameterised sample + log q on the
2-D latent (a4/a5), the flow prior (a6-a10, a15; the fused NICE stack treats (B, D) as S = 1), KL and the ELBO
tail (a16/a17) and the reconstruction NLL (a18).  Beam search (decoders/beam_search.py) is inference-only and not
built: generation without a target is greedy (the reference's own default is beam_size = 1, decoder.py:37).
"""
import random as _random
from itertools import chain
from typing import Dict, Iterable, List, Tuple

import torch

from . import ops
from .latent import KLLoss, Posterior, PosteriorSample, Prior, PriorSample
from .model import VAELmModel, WeightScheduler
from .nn_bracket import Activation, Embedder, Encoder, FeedForward, RecurrentCell
from .registrable import Registrable


class AutoDecoder(torch.nn.Module, Registrable):
    """Token-by-token decoder from a sentence latent (decoders/decoder.py:11-160)."""

    def __init__(self, embedder: Embedder, sos_index: int, eos_index: int, unk_index: int, max_timesteps: int,
                 teacher_forcing_ratio: float, word_dropout: float = None, beam_size: int = 1,
                 skip_z: bool = False) -> None:
        super().__init__()
        if not 0 <= teacher_forcing_ratio <= 1:
            raise ValueError("teacher_forcing_ratio should be in between 0 and 1.")
        self._embedder = embedder
        self._sos_index, self._eos_index, self._unk_index = sos_index, eos_index, unk_index
        self._max_timesteps = max_timesteps
        self._teacher_forcing_ratio = teacher_forcing_ratio
        self._word_dropout = word_dropout or 0.0
        self._skip_z = skip_z

    def forward(self, z, target=None):
        state = {"latent": z, "hidden": z, "cell": torch.zeros_like(z)}          # decoder.py:129-135
        return self._forward_loop(state, target) if target is not None else self._generate(state)

    def _forward_loop(self, state, target):
        logits_all, preds_all = [], []
        prediction = torch.full((target.size(0),), self._sos_index, dtype=torch.long, device=target.device)
        for step in range(target.size(1) - 1):
            logits, state = self.decoder_step(self._choose_input(prediction, target[:, step]), state)
            prediction = torch.argmax(logits, dim=-1)     # argmax(softmax(x)) == argmax(x), decoder.py:60-61
            logits_all.append(logits.unsqueeze(1))
            preds_all.append(prediction.unsqueeze(1))
        return torch.cat(logits_all, dim=1), torch.cat(preds_all, dim=1)

    def _generate(self, state):
        logits_all, preds_all = [], []
        prediction = torch.full((state["latent"].size(0),), self._sos_index, dtype=torch.long,
                                device=state["latent"].device)
        for _ in range(self._max_timesteps):
            logits, state = self.decoder_step(prediction, state)
            prediction = torch.argmax(logits, dim=-1)
            logits_all.append(logits.unsqueeze(1))
            preds_all.append(prediction.unsqueeze(1))
        return torch.cat(logits_all, dim=1), torch.cat(preds_all, dim=1)

    def _choose_input(self, prediction, target):
        if not self.training:
            return target
        # two host RNG draws per step, in the reference's order (decoder.py:119-121)
        use_word_dropout = _random.random() < self._word_dropout
        use_teacher_forcing = _random.random() < self._teacher_forcing_ratio
        if use_word_dropout:
            return torch.full_like(target, self._unk_index)
        return target if use_teacher_forcing else prediction

    def decoder_step(self, step_input, state):
        raise NotImplementedError()


@AutoDecoder.register("auto-recurrent")
class RecurrentDecoder(AutoDecoder):
    """decoders/rnn.py:12-105: LSTM cell (hidden initialised with z) + projection to the vocabulary."""

    def __init__(self, cell: RecurrentCell, embedder: Embedder, num_classes: int, sos_index: int, eos_index: int,
                 unk_index: int, max_timesteps: int, teacher_forcing_ratio: float, word_dropout: float = None,
                 hidden_sizes: List[int] = None, ff_dropout: float = 0.2, ff_activation: str = "elu",
                 beam_size: int = 2, skip_z: bool = False) -> None:
        super().__init__(embedder=embedder, sos_index=sos_index, eos_index=eos_index, unk_index=unk_index,
                         max_timesteps=max_timesteps, teacher_forcing_ratio=teacher_forcing_ratio,
                         word_dropout=word_dropout, beam_size=beam_size, skip_z=skip_z)
        self._cell = cell
        self._num_classes = num_classes
        width = cell.get_output_size() + (cell.get_input_size() if skip_z else 0)
        if hidden_sizes is not None:
            self._projection = FeedForward(input_size=width, num_layers=len(hidden_sizes), hidden_sizes=hidden_sizes,
                                           activations=Activation.by_name(ff_activation), dropout=ff_dropout,
                                           output_size=num_classes)
        else:
            self._projection = torch.nn.Linear(width, num_classes)

    def get_input_size(self) -> int:
        return self._cell.get_input_size()

    def get_output_size(self) -> int:
        return self._num_classes

    def decoder_step(self, step_input, state):
        hidden, cell = self._cell(self._embedder(step_input), (state["hidden"], state["cell"]))
        state["hidden"], state["cell"] = hidden, cell
        feats = torch.cat((hidden, state["latent"]), dim=-1) if self._skip_z else hidden
        return self._projection(feats), state

    @classmethod
    def from_params(cls, vocab, **params):
        cell = RecurrentCell.from_params(**params.pop("cell"))
        embedder = Embedder.from_params(vocab=vocab, **params.pop("embedder"))
        return cls(cell=cell, embedder=embedder, num_classes=vocab.get_vocab_size(namespace="target"),
                   sos_index=vocab.token_to_index("<sos>", namespace="target"),
                   eos_index=vocab.token_to_index("<eos>", namespace="target"),
                   unk_index=vocab.token_to_index("@@UNKNOWN@@", namespace="target"), **params)


@VAELmModel.register("auto")
class AutoModel(VAELmModel):
    """vae_lm/models/auto_lm/model.py:19-232."""

    def __init__(self, vocab, embedder: Embedder, encoder: Encoder, decoder: AutoDecoder, posterior: Posterior,
                 prior: Prior, kl_loss: KLLoss, recon_scheduler: WeightScheduler, kl_scheduler: WeightScheduler,
                 iwae: bool = False, label_smoothing: float = 0.0) -> None:
        super().__init__(vocab=vocab, kl_loss=kl_loss, recon_scheduler=recon_scheduler, kl_scheduler=kl_scheduler,
                         iwae=iwae, label_smoothing=label_smoothing)
        self._embedder = embedder
        self._encoder = encoder
        self._decoder = decoder
        self._posterior = posterior
        self._prior = prior

    @property
    def nsamples_posterior(self) -> int:
        return self._posterior.samples

    def set_samples(self, samples: int) -> None:
        self._posterior.samples = samples

    def encode(self, tokens):
        return self._encoder(self._embedder(tokens), ops.tokens_mask(tokens))

    def decode(self, z, mask, target=None) -> Dict[str, torch.Tensor]:
        logits, predictions = self._decoder(z, target)
        out = {"logits": logits, "preds": predictions}
        if not (self.training_outputs_only and target is not None):
            out["probs"] = torch.softmax(logits, dim=-1)
        if target is not None:
            # the <sos> column is given, not predicted (model.py:113-121)
            weights = ops.tokens_mask(target).float()
            out["loss"] = self._loss(logits, target[:, 1:].contiguous(), weights=weights[:, 1:].contiguous())
        return out

    def sample_from_prior(self, samples: int, lengths: List[int] = None) -> PriorSample:
        z, log_prob = self._prior.sample(samples)[:2]
        return PriorSample(z, log_prob, None)

    def sample_from_posterior(self, encoded, random: bool = True, eps=None) -> PosteriorSample:
        return PosteriorSample(*self._posterior(encoded.ctx, None, random=random, eps=eps))

    def _get_prior_log_prob(self, z, mask=None):
        return self._prior.log_probability(z, None)

    def _posterior_inputs(self, encoded):
        return encoded.ctx, None

    def _prior_token_log_prob(self, z, mask=None):
        fn = getattr(self._prior, "token_log_prob", None)
        return None if fn is None else fn(z, None)

    def calc_mutual_info(self, src_tokens, random: bool = True):
        encoded = self.encode(src_tokens)
        latent, log_q = self.sample_from_posterior(encoded, random=True)
        return self._posterior.calc_mutual_info(latent.z, log_q)

    def encoder_parameters(self) -> Iterable[torch.nn.Parameter]:
        return chain(self._encoder.parameters(), self._posterior.parameters(), self._prior.parameters())

    def decoder_parameters(self) -> Iterable[torch.nn.Parameter]:
        return self._decoder.parameters()

    @classmethod
    def from_params(cls, vocab, **params):
        params = dict(params)
        embedder = Embedder.from_params(vocab=vocab, **params.pop("embedder"))
        encoder = Encoder.from_params(**params.pop("encoder"))
        decoder = AutoDecoder.from_params(vocab=vocab, **params.pop("decoder"))
        posterior = Posterior.from_params(**params.pop("posterior"))
        prior = Prior.from_params(**params.pop("prior"))
        kl_loss = KLLoss.from_params(prior=prior, **params.pop("kl_loss"))
        recon_scheduler = WeightScheduler.from_params(**params.pop("recon_scheduler"))
        kl_scheduler = WeightScheduler.from_params(**params.pop("kl_scheduler"))
        return cls(vocab=vocab, embedder=embedder, encoder=encoder, decoder=decoder, posterior=posterior, prior=prior,
                   kl_loss=kl_loss, recon_scheduler=recon_scheduler, kl_scheduler=kl_scheduler, **params)
