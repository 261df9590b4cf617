"""nonauto_lm_amd: MI355X-native latent hot path of Nemexur/nonauto-lm (flow stack, posterior
reparameterisation + log q, base log p, KL / ELBO) behind the reference's module API.

HIP kernels + C ABI: csrc/ -> lib/libmivae.so (include/mivae.h); host mirror: this package."""
from . import _lib, ops
from .flows import (MADE, ActNorm, AttentionCoupling, Coupling, FeedForwardCoupling, Flow, IAFlow, InvertibleLinear,
                    InvertibleLinearFlow, InvertibleMultiHeadFlow, MAFlow, MaskedLinear, NICECoupling,
                    RecurrentCoupling)
from .latent import (DefaultKLLoss, DefaultPosterior, DefaultPrior, FlowPrior, InfoVAEKLLoss, KLLoss, LatentSample,
                     Posterior, PosteriorSample, Prior, PriorSample)
from .model import NonAutoModel, SimpleVocab, VAELmModel, WeightScheduler
from .auto import AutoDecoder, AutoModel, RecurrentDecoder
from .nn_bracket import (Activation, Embedder, Encoder, FeedForward, LabelSmoothingNLL, RecurrentCell, RecurrentCore,
                         RecurrentEncoder)
from .registrable import FromParams, Registrable

__all__ = [n for n in dir() if not n.startswith("_")]
