"""Flow modules of the latent hot path, HIP-backed, behind the reference's `Flow` API.

Mirrors vae_lm/models/base/flows/: same class names, registry names, constructor signatures,
`forward(z, mask=None) -> (z', log_det)` / `backward(z, mask=None)` contract (flow.py:7-20) and
state-dict keys.  All arithmetic on z runs in libmivae kernels (ops.py); tensors must be on the GPU.
"""
from copy import deepcopy
from typing import List, Optional, Tuple, Union

import torch

from . import ops
from .nn_bracket import Activation, FeedForward, PositionalEncoding, RecurrentCore, TransformerDecoderLayer
from .registrable import Registrable


class Flow(torch.nn.Module, Registrable):
    """Generic generative flow (flow.py:7-20)."""

    def forward(self, z, mask=None):
        raise NotImplementedError()

    def backward(self, z, mask=None):
        raise NotImplementedError()

    def get_input_size(self):
        return None

    def get_output_size(self):
        return None


class Coupling(Flow, Registrable):
    """Coupling network base (couplings.py:14-23)."""

    def __init__(self, input_size: int) -> None:
        super().__init__()
        self._input_size = input_size

    def get_input_size(self) -> int:
        return self._input_size

    def get_output_size(self) -> int:
        return self._input_size


@Coupling.register("feed-forward")
class FeedForwardCoupling(Coupling):
    """couplings.py:26-51.  Standalone calls run the stock MLP; inside NICECoupling the
    one-hidden-layer case is executed by the fused MFMA kernel and this forward is bypassed."""

    def __init__(self, input_size: int, hidden_sizes: Union[int, List[int]], activation: str = "gelu",
                 dropout: float = 0.0) -> None:
        super().__init__(input_size)
        hidden_sizes = hidden_sizes if isinstance(hidden_sizes, list) else [hidden_sizes]
        self._hidden_sizes = list(hidden_sizes)
        self._activation_name = activation
        self._dropout_p = float(dropout)
        self._feedforward = FeedForward(input_size=input_size, num_layers=len(hidden_sizes),
                                        hidden_sizes=list(hidden_sizes),
                                        activations=Activation.by_name(activation), dropout=dropout,
                                        output_size=input_size)
        self._dropout = torch.nn.Dropout(dropout) if dropout > 0.0 else torch.nn.Identity()

    def forward(self, z, mask=None):
        return self._feedforward(self._dropout(z))

    def fused_params(self) -> Optional[List[torch.Tensor]]:
        """[W1, b1, W2, b2] when the fused kernel applies (one hidden layer, supported activation,
        dropout inactive), else None."""
        if len(self._hidden_sizes) != 1 or self._activation_name not in ops.ACT_IDS:
            return None
        if self._dropout_p > 0.0 and self.training:
            return None
        lin, out = self._feedforward._linear_layers[0], self._feedforward._output
        return [lin.weight, lin.bias, out.weight, out.bias]


@Coupling.register("attention")
class AttentionCoupling(Coupling):
    """couplings.py:54-81 (a11): positional encoding + causal self-attention layers of width h on x0.
    The network itself stays on library ops; inside NICECoupling its s/t outputs
    feed the HIP affine + masked log-det kernel."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int, num_heads: int,
                 dropout: float = 0.0) -> None:
        super().__init__(input_size)
        self._pos_enc = PositionalEncoding(input_size)
        self._layers = torch.nn.ModuleList(
            [TransformerDecoderLayer(input_size, hidden_size, num_heads, dropout=dropout) for _ in range(num_layers)])
        self._dropout = torch.nn.Dropout(dropout) if dropout > 0.0 else torch.nn.Identity()

    def forward(self, z, mask=None):
        z = self._pos_enc(self._dropout(z)) * mask.unsqueeze(-1).to(z.dtype)
        for layer in self._layers:
            z = layer(z, mask)
        return z


@Coupling.register("recurrent")
class RecurrentCoupling(Coupling):
    """couplings.py:84-119 (a11): recurrent core over x0 + projection back to h features."""

    def __init__(self, core: RecurrentCore, hidden_sizes: List[int] = None, ff_activation: str = "gelu",
                 dropout: float = 0.0) -> None:
        super().__init__(core.get_input_size())
        self._core = core
        if hidden_sizes is None:
            self._projection = torch.nn.Linear(core.get_output_size(), core.get_input_size())
        else:
            self._projection = FeedForward(input_size=core.get_output_size(), num_layers=len(hidden_sizes),
                                           hidden_sizes=hidden_sizes, activations=Activation.by_name(ff_activation),
                                           dropout=dropout, output_size=core.get_input_size())
        self._dropout = torch.nn.Dropout(dropout) if dropout > 0.0 else torch.nn.Identity()

    def forward(self, z, mask=None):
        return self._projection(self._core(self._dropout(z), mask))

    @classmethod
    def from_params(cls, **params):
        core = RecurrentCore.from_params(**params.pop("core"))
        return cls(core, **params)


@Flow.register("nice")
class NICECoupling(Flow):
    """Affine coupling (couplings.py:122-193).  Halves are never swapped (couplings.py:185-188)."""

    def __init__(self, coupling: Coupling) -> None:
        super().__init__()
        assert coupling.get_input_size() // 2, "Input features should be divesable by 2."
        self._s_net = deepcopy(coupling)   # identical initial weights, like the reference (:138-139)
        self._t_net = deepcopy(coupling)

    # ---- fused path -------------------------------------------------------------------------------
    def fused_spec(self, z: torch.Tensor):
        """(params[8], H, activation) if this step can run in the fused stack kernel for `z`."""
        if not (isinstance(self._s_net, FeedForwardCoupling) and isinstance(self._t_net, FeedForwardCoupling)):
            return None
        ps, pt = self._s_net.fused_params(), self._t_net.fused_params()
        if ps is None or pt is None:
            return None
        if z.size(-1) != 2 * self._s_net.get_input_size():
            return None
        if z.size(-1) > ops.FUSED_MAX_D and not ops.nice_big_supported(z, self._s_net._hidden_sizes[0]):
            return None
        if z.dtype not in (torch.float32, torch.bfloat16):
            return None
        if any(p.dtype != torch.float32 for p in ps + pt):
            return None
        return ps + pt, self._s_net._hidden_sizes[0], self._s_net._activation_name

    def _run(self, z, mask, inverse):
        spec = self.fused_spec(z)
        if spec is not None:
            params, H, act = spec
            return ops.nice_stack(z, mask, params, H, act, inverse=inverse)
        # composed path (attention / recurrent / multi-layer couplings, active dropout, D > 256):
        # coupling nets on library ops, affine + masked log-det tail in the HIP kernel.
        h = z.size(-1) // 2
        x0 = z[..., :h]
        s = self._s_net(x0, mask)
        t = self._t_net(x0, mask)
        return ops.nice_affine(z, s, t, mask, inverse=inverse)

    def forward(self, z, mask=None):
        return self._run(z, mask, False)

    def backward(self, z, mask=None):
        return self._run(z, mask, True)

    @classmethod
    def from_params(cls, **params):
        coupling = Coupling.from_params(**params.pop("coupling"))
        return cls(coupling=coupling, **params)


@Flow.register("actnorm")
class ActNorm(Flow):
    """actnorm.py:6-54, including the data-dependent first-call init (:22-28, padding included).

    Deliberate deviation (documented in DESIGN.md): under torch.distributed the init statistics are
    all-reduced so every rank starts from identical parameters (the reference lets ranks diverge,
    SURVEY F7).  `_is_initted` stays a plain attribute, as in the reference."""

    def __init__(self, input_size: int) -> None:
        super().__init__()
        self._input_size = input_size
        self._log_scale = torch.nn.Parameter(torch.empty(input_size))
        self._bias = torch.nn.Parameter(torch.empty(input_size))
        self._is_initted = False
        self.reset_parameters()

    def reset_parameters(self) -> None:
        torch.nn.init.normal_(self._log_scale, mean=0, std=0.05)
        torch.nn.init.constant_(self._bias, 0.0)

    @torch.no_grad()
    def _data_init(self, z):
        mean, std = ops.actnorm_stats(z)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # pool (count, sum, sum of squares) across ranks: identical parameters everywhere
            n = float(z.numel() // z.size(-1))
            stats = torch.stack([mean * n, (std * std) * (n - 1) + mean * mean * n])
            cnt = torch.tensor([n], device=z.device)
            dist.all_reduce(stats)
            dist.all_reduce(cnt)
            tot = cnt.item()
            mean = stats[0] / tot
            std = ((stats[1] - tot * mean * mean) / (tot - 1)).clamp(min=0).sqrt()
        inv_std = 1.0 / (std + 1e-6)
        self._log_scale.data.add_(inv_std.log())
        self._bias.data.add_(-mean).mul_(inv_std)
        self._is_initted = True

    def forward(self, z, mask=None):
        if not self._is_initted:
            self._data_init(z)
        return ops.actnorm(z, self._log_scale, self._bias, mask, inverse=False)

    def backward(self, z, mask=None):
        return ops.actnorm(z, self._log_scale, self._bias, mask, inverse=True)


class MaskedLinear(torch.nn.Linear):
    """Linear layer with a 0/1 mask on its weights (made.py:8-40); `weight_mask` is a buffer of the state dict."""

    def __init__(self, in_features: int, out_features: int, bias=True) -> None:
        super().__init__(in_features, out_features, bias)
        self.register_buffer("weight_mask", torch.ones(out_features, in_features))

    @property
    def mask(self) -> torch.Tensor:
        return self.weight_mask

    @mask.setter
    def mask(self, value: torch.Tensor) -> None:
        self.weight_mask.data.copy_(value.T)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight * self.weight_mask, self.bias)


class MADE(torch.nn.Module):
    """Masked autoencoder for distribution estimation (made.py:43-127): MaskedLinear / activation stack with
    2 * input_size outputs ([s | t]); masks from numpy RandomState(13) exactly like `_update_masks` (:93-122)."""

    def __init__(self, input_size: int, hidden_sizes: List[int], permute_order: bool = True,
                 activation: str = "elu") -> None:
        super().__init__()
        self._input_size = input_size
        self._output_size = input_size * 2
        self._hidden_sizes = list(hidden_sizes)
        self._activation_name = activation
        nets = []
        hs = [input_size] + self._hidden_sizes + [self._output_size]
        for h0, h1 in zip(hs, hs[1:]):
            nets.extend([MaskedLinear(h0, h1), Activation.by_name(activation)])
        nets.pop()
        self._nets = torch.nn.Sequential(*nets)
        self._permute_order = permute_order
        self._seed = 13
        self._masks = {}
        self._update_masks()

    def get_input_size(self) -> int:
        return self._input_size

    def _update_masks(self) -> None:
        import numpy as np
        n = len(self._hidden_sizes)
        rng = np.random.RandomState(self._seed)
        self._masks[-1] = rng.permutation(self._input_size) if self._permute_order else np.arange(self._input_size)
        for layer in range(n):
            self._masks[layer] = rng.randint(self._masks[layer - 1].min(), self._input_size - 1,
                                             size=self._hidden_sizes[layer])
        masks = [self._masks[layer - 1][:, None] <= self._masks[layer][None, :] for layer in range(n)]
        last = self._masks[n - 1][:, None] < self._masks[-1][None, :]
        masks.append(np.concatenate([last] * (self._output_size // self._input_size), axis=1))
        for layer, m in zip(self.masked_layers(), masks):
            layer.mask = torch.from_numpy(m.astype("float32"))

    def masked_layers(self) -> List[MaskedLinear]:
        return [m for m in self._nets if isinstance(m, MaskedLinear)]

    def forward(self, z):
        return self._nets(z)


@Flow.register("maf")
class MAFlow(Flow):
    """Masked autoregressive flow (maf.py:8-67).  `forward` = one MADE pass (library GEMMs) + the HIP affine /
    flip / masked log-det tail.  `backward` = the D-step sequential direction: one `mivae_maf_sequential` launch
    for a one-hidden-layer MADE when no gradient is needed, else the reference's loop on library ops.

    Kept from the reference: features are visited in natural order whatever the MADE ordering is, and the
    sequential direction accumulates the log-det with +s (maf.py:54-58).  Deviations: the log-det is created on
    z's device (the reference builds it on the CPU, maf.py:52, and cannot run this direction on a GPU), and for 2-D
    input it is the per-row sum (B,) (the reference's (B, D) buffer only broadcasts when B == D)."""

    def __init__(self, made: MADE, parity: bool) -> None:
        super().__init__()
        self._made = made
        self._dim = made.get_input_size()
        self._parity = parity

    def _density(self, z, mask=None):
        return ops.maf_affine(z, self._made(z), mask, self._parity)

    def _sequential(self, z, mask=None):
        layers = self._made.masked_layers()
        needs_grad = torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self._made.parameters()))
        if (len(layers) == 2 and self._made._activation_name in ops.MAF_ACT_IDS
                and z.dtype in (torch.float32, torch.bfloat16)):
            l1, l2 = layers
            if needs_grad:      # training direction of IAF: one-launch forward + reverse-recurrence backward
                if ops.maf_sequential_grad_supported(l1.weight) and l1.weight.dtype == torch.float32:
                    return ops.maf_sequential(z, None, mask, self._made._activation_name, self._parity,
                                              params=(l1.weight, l1.weight_mask, l1.bias, l2.weight, l2.weight_mask,
                                                      l2.bias))
            else:
                packed = ops.maf_sequential_pack(l1.weight, l1.weight_mask, l1.bias, l2.weight, l2.weight_mask,
                                                 l2.bias)
                if packed is not None:
                    return ops.maf_sequential(z, packed, mask, self._made._activation_name, self._parity)
        # composed path (training direction of IAF, deeper MADEs): the reference's loop, maf.py:51-62
        from . import _lib as L
        L.require_gpu(z, mask)
        D = self._dim
        x = torch.zeros_like(z)
        log_det = torch.zeros(z.shape[:-1], dtype=z.dtype, device=z.device)
        zz = z.flip(dims=(-1,)) if self._parity else z
        for i in range(D):
            st = self._made(x.clone())
            x[..., i] = (zz[..., i] - st[..., D + i]) * torch.exp(-st[..., i])
            log_det = log_det + st[..., i]
        if mask is not None:
            x = x * mask.unsqueeze(-1)
            log_det = log_det.mul(mask).sum(dim=-1)
        return x, log_det

    def forward(self, z, mask=None):
        return self._density(z, mask)

    def backward(self, z, mask=None):
        return self._sequential(z, mask)

    @classmethod
    def from_params(cls, **params):
        # maf.py:64-67 builds `MAFlow(...)` whatever `cls` is, so a config with type "iaf" yields a MAFlow in the
        # reference; reproduced (IAFlow is reachable by direct construction only, there as here).
        made = MADE(**params.pop("made"))
        return MAFlow(made=made, **params)


@Flow.register("iaf")
class IAFlow(MAFlow):
    """Inverse autoregressive flow (iaf.py:6-23): MAF with the two directions swapped."""

    def forward(self, z, mask=None):
        return self._sequential(z, mask)

    def backward(self, z, mask=None):
        return self._density(z, mask)


class InvertibleLinear(Flow, Registrable):
    """Registry of invertible linear flows (invertible_flows.py:11-12)."""


class _InvLinearBase(Flow):
    def _linear_flow(self, z, mask, inverse, heads):
        from . import _lib as L
        L.require_gpu(z, mask)
        w = self._weight
        if inverse:
            w = torch.inverse(w.double()).float()        # invertible_flows.py:37
        # the head split is the identity on the memory image: (N, D) == (N heads, D / heads) row major, so both flows
        # are ONE product with a small square matrix on the matrix cores (csrc/invlinear.hip)
        lead = z.shape[:-1]
        hs = z.size(-1) // heads
        if hs <= ops.ROWGEMM_MAX and z.dtype in (torch.float32, torch.bfloat16):
            out = ops.rowgemm(z.reshape(*lead, heads, hs), w).reshape(*lead, z.size(-1))
        else:       # matrices wider than the MFMA kernel's 512 / other latent dtypes: the library product
            out = torch.matmul(z.reshape(*lead, heads, hs), w.to(z.dtype).t()).reshape(*lead, z.size(-1))
        logdet = ops.slogdet(w)                       # invertible_flows.py:30-33: torch.slogdet(w)[1]
        if z.dim() > 2:
            cnt = ops.mask_count(mask, z.size(0))        # integer token count (HIP), bit exact
            logdet = logdet * (cnt * heads)
        return out, logdet


@InvertibleLinear.register("inv-linear")
class InvertibleLinearFlow(_InvLinearBase):
    """invertible_flows.py:16-44: z W^T, no mask multiply, log_det = slogdet(W) * token count."""

    def __init__(self, input_size: int):
        super().__init__()
        self._weight = torch.nn.Parameter(torch.empty(input_size, input_size))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        torch.nn.init.orthogonal_(self._weight)

    def forward(self, z, mask=None):
        return self._linear_flow(z, mask, False, 1)

    def backward(self, z, mask=None):
        return self._linear_flow(z, mask, True, 1)


@InvertibleLinear.register("inv-multi-head")
class InvertibleMultiHeadFlow(_InvLinearBase):
    """invertible_flows.py:48-83: one (D/heads)^2 matrix shared by all heads."""

    def __init__(self, input_size: int, num_heads: int) -> None:
        super().__init__()
        assert input_size % num_heads == 0
        self._num_heads = num_heads
        self._head_size = input_size // num_heads
        self._weight = torch.nn.Parameter(torch.empty(self._head_size, self._head_size))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        torch.nn.init.orthogonal_(self._weight)

    def forward(self, z, mask=None):
        return self._linear_flow(z, mask, False, self._num_heads)

    def backward(self, z, mask=None):
        return self._linear_flow(z, mask, True, self._num_heads)


def fused_stack_spec(flows, z):
    """If every flow is a NICE-FF step with the same (H, activation) the whole list runs as one
    fused launch: returns (params[K*8], H, activation), else None."""
    params, H0, act0 = [], None, None
    for flow in flows:
        if not isinstance(flow, NICECoupling):
            return None
        spec = flow.fused_spec(z)
        if spec is None:
            return None
        p, H, act = spec
        if H0 is None:
            H0, act0 = H, act
        elif (H, act) != (H0, act0):
            return None
        params += p
    if not params:
        return None
    return params, H0, act0


def _mixed_plan(flows, z, inverse: bool):
    """(MixedPlan, constant) for the list, cached on the first flow while no parameter changes (storage address,
    in-place version counter, and ops.PARAM_EPOCH for the raw-pointer writes of FusedAdam / replayed graphs): the matrix inverse and the slogdet of the linear steps are rocSOLVER calls that cost
    far more than the flow kernel.  `constant` = sum over steps of the parameter-only log-determinant per token
    (ActNorm: sum log_scale; linear: slogdet x heads), a device scalar, or None when nothing contributes."""
    D, three = z.size(-1), z.dim() == 3
    key = (D, three, z.dtype, z.device, ops.PARAM_EPOCH[0],
           tuple((p.data_ptr(), p._version) for f in flows for p in f.parameters()),
           tuple(getattr(f, "_is_initted", True) for f in flows), tuple(f.training for f in flows))
    cache = flows[0].__dict__.setdefault("_mixed_plan_cache", {})
    hit = cache.get(inverse)
    if hit is not None and hit[0] == key:
        return hit[1]
    plan = None
    steps, const, hmax, ok = [], None, 0, D % 2 == 0
    for flow in (flows[::-1] if inverse else flows):
        if not ok:
            break
        if isinstance(flow, NICECoupling):
            spec = flow.fused_spec(z)
            if spec is None:
                ok = False
                break
            params, H, act = spec
            hmax = max(hmax, H)
            steps.append((ops.MIXED_NICE, H, act, params))
            continue
        if isinstance(flow, ActNorm):
            if not flow._is_initted or flow._log_scale.dtype != torch.float32 or flow._input_size != D:
                ok = False
                break
            steps.append((ops.MIXED_ACTNORM, flow._log_scale, flow._bias))
            c = flow._log_scale.detach().sum()
            c = -c if inverse else c
        elif isinstance(flow, _InvLinearBase):
            heads = getattr(flow, "_num_heads", 1)
            w = flow._weight.detach()
            if w.dtype != torch.float32 or w.size(0) * heads != D:
                ok = False
                break
            if inverse:
                w = torch.inverse(w.double()).float()            # invertible_flows.py:37
            steps.append((ops.MIXED_LINEAR, w))
            c = torch.linalg.slogdet(w)[1]
            if three:                                            # 2-D inputs: no head factor, like the per-flow modules
                c = c * heads
        else:
            ok = False
            break
        const = c if const is None else const + c
    if ok and steps and ops.mixed_stack_supported(D, hmax) and len(steps) <= 24:
        plan = (ops.MixedPlan(steps), const)
    cache[inverse] = (key, plan)
    return plan


def mixed_stack_run(flows, z, mask, inverse: bool):
    """One-launch evaluation of a flow list that mixes NICE-FF, ActNorm and invertible-linear steps (flow_prior.py:37-70
    run through csrc/mixed_stack.hip).  Returns (z', log_det) like the per-flow loop, or None when the list does not
    qualify: a gradient is needed, an unsupported flow / width, an ActNorm that still has to see its first batch, or a
    coupling-only list (which has its own fused kernels)."""
    flows = list(flows)
    if not flows or z.dtype not in (torch.float32, torch.bfloat16) or z.dim() not in (2, 3):
        return None
    if z.dim() == 3 and mask is None:
        return None
    if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for f in flows for p in f.parameters())):
        return None
    if all(isinstance(f, NICECoupling) for f in flows):
        return None
    planned = _mixed_plan(flows, z, bool(inverse))
    if planned is None:
        return None
    plan, const = planned
    out, val = ops.mixed_stack(z, mask, plan, inverse=inverse)
    if const is not None:
        # (sum log_scale) x tokens, slogdet x tokens x heads; a 2-D input counts once
        val = val + (const * ops.mask_count(mask, z.size(0)) if z.dim() == 3 else const)
    return out, val


def mixed_stack_train_run(flows, z, mask):
    """The density direction of a mixed NICE-FF / ActNorm / invertible-linear list WITH gradients (training): one launch
    forward, one launch for the z-gradient chain (csrc/mixed_stack.hip), parameter gradients as a fixed handful of
    reductions (ops._MixedStackTrain).  The parameter-only log-determinants (sum log_scale, slogdet W x heads) x token
    count are formed here from the live parameters, so autograd differentiates them.  Returns (z', log_det) like the
    per-flow loop of flow_prior.py:37-52, or None when the list does not qualify (then that loop runs)."""
    flows = list(flows)
    if not flows or not z.is_cuda or z.dtype not in (torch.float32, torch.bfloat16) or z.dim() not in (2, 3):
        return None
    if (z.dim() == 3 and mask is None) or z.size(-1) % 2 or not torch.is_grad_enabled():
        return None
    if all(isinstance(f, NICECoupling) for f in flows) or len(flows) > 24:
        return None
    D, three = z.size(-1), z.dim() == 3
    steps, hidden, scales, mats, heads_of = [], [], [], [], []
    for flow in flows:
        if isinstance(flow, NICECoupling):
            spec = flow.fused_spec(z)
            if spec is None or D > ops.FUSED_MAX_D:
                return None
            params, H, act = spec
            hidden.append(H)
            steps.append((ops.MIXED_NICE, H, act, params))
        elif isinstance(flow, ActNorm):
            if not flow._is_initted or flow._log_scale.dtype != torch.float32 or flow._input_size != D:
                return None
            steps.append((ops.MIXED_ACTNORM, flow._log_scale, flow._bias))
            scales.append(flow._log_scale)
        elif isinstance(flow, _InvLinearBase):
            heads = getattr(flow, "_num_heads", 1)
            w = flow._weight
            if w.dtype != torch.float32 or w.size(0) * heads != D:
                return None
            steps.append((ops.MIXED_LINEAR, w))
            mats.append(w)
            heads_of.append(float(heads) if three else 1.0)       # 2-D inputs: no head factor, like the per-flow modules
        else:
            return None
    if not ops.mixed_stack_train_supported(D, hidden, bool(mats)):
        return None
    out, val = ops.mixed_stack_train(z, mask, steps)
    # parameter-only log-determinants per token: sum log_scale (actnorm.py:35-38), slogdet W x heads
    # (invertible_flows.py:30-33, :64-66) -- every slogdet of the list in one launch, the scales in one reduction
    const = None
    if scales:
        const = (scales[0] if len(scales) == 1 else torch.cat(scales)).sum()
    if mats:
        ld = ops.slogdet_batch(mats)
        if any(hf != 1.0 for hf in heads_of):
            ld = ld * torch.tensor(heads_of, dtype=torch.float32).to(ld.device, non_blocking=True)
        const = ld.sum() if const is None else const + ld.sum()
    if const is not None:
        val = val + (const * ops.mask_count(mask, z.size(0)) if three else const)
    return out, val
