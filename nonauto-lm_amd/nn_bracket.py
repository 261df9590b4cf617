"""Stock-PyTorch pieces that bracket the latent hot path (OUT OF SCOPE rows):
Activation registry, FeedForward, transformer encoder / non-autoregressive decoder, embedder and the
label-smoothing NLL.  They run on PyTorch-ROCm library ops (rocBLAS / MIOpen); no HIP kernels here.
Parameter names follow the reference so its checkpoints load (state-dict keys).
"""
import math
import os
from typing import Callable, Dict, List, NamedTuple, Union

import torch
import torch.nn.functional as F

from . import ops
from .registrable import FromParams, Registrable


# ----------------------------------------------------------------------------------- linear
class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b with a token-split weight gradient.

    The stock weight-gradient GEMM of a small Linear over many tokens (M = out <= 256, N = in <= 256,
    K = tokens >= 16k) is dispatched by hipBLASLt to a single 64x64 output tile, i.e. ONE workgroup
    walks the whole token axis (106 us per layer at 16k tokens on MI355X, 27 such layers per step).
    Splitting the token axis into G batches (one batched GEMM + a G-way sum) spreads it over the chip.
    Same arithmetic, different (fixed) summation order."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_lp, b_lp):
        # w_lp / b_lp: low-precision compute copies of the fp32 parameters (bf16 autocast), or None.  The
        # gradients go straight to the fp32 parameters in fp32 (no cast nodes, no bf16 rounding of dW).
        w = weight if w_lp is None else w_lp
        b = bias if b_lp is None else b_lp
        ctx.has_bias = bias is not None
        ctx.param_dtype = weight.dtype
        ctx.params = (weight, bias)
        ctx.small = ops.linear_small_supported(x, w) and (b is None or b.dtype == torch.bfloat16)
        if ctx.small:                               # libmivae MFMA kernels: one read of x, one write of y
            x2 = x.contiguous().view(-1, x.size(-1))
            ctx.save_for_backward(x2, w)
            ctx.xshape = x.shape
            return ops.linear_fwd(x2, w, b).view(*x.shape[:-1], w.size(0))
        ctx.save_for_backward(x, w)
        with torch.autocast("cuda", enabled=False):
            return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors[:2]
        gx = gw = gb = None
        if ctx.small and gy.dtype == torch.bfloat16:
            g2 = gy.contiguous().view(-1, gy.size(-1))
            if ctx.needs_input_grad[0]:
                gx = ops.linear_dgrad(g2, weight).view(ctx.xshape)
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                if weight.size(0) <= 128 and weight.size(1) <= 128:
                    if ctx.has_bias and all(ctx.needs_input_grad[1:3]) and ops.grad_sink_ok(*ctx.params):
                        ops.linear_wgrad(x, g2, sink=(ctx.params[0].grad, ctx.params[1].grad),
                                         hold=getattr(ctx, "hold_wgrad", False))
                        return gx, None, None, None, None
                    gw, gb = ops.linear_wgrad(x, g2)
                    gw, gb = gw.to(ctx.param_dtype), (gb.to(ctx.param_dtype) if ctx.has_bias else None)
                    return gx, gw, gb, None, None
            else:
                return gx, None, None, None, None
        if ctx.needs_input_grad[0] and gx is None:
            gx = gy.matmul(weight.to(gy.dtype))
        x2 = x.reshape(-1, x.size(-1))
        g2 = gy.reshape(-1, gy.size(-1))
        if g2.dtype != x2.dtype:
            x2 = x2.to(g2.dtype)
        n = x2.size(0)
        # gradient sinks (training step): fold the partials straight into the parameters' .grad -- no temporaries, no
        # cast, no accumulate launch per parameter
        sink = (ctx.has_bias and all(ctx.needs_input_grad[1:3]) and ops.grad_sink_ok(*ctx.params) and g2.is_cuda
                and g2.dtype in (torch.float32, torch.bfloat16) and n >= 1024 and g2.size(1) <= 1024)
        if ctx.needs_input_grad[1]:
            G = 1                                   # 16 token slabs: flat optimum for 128..1024-wide layers (MI355X)
            while G < 16 and n % (2 * G) == 0 and n // (2 * G) >= 128:
                G *= 2
            if G > 1 and sink and g2.is_contiguous() and x2.is_contiguous() and \
                    ops.wide_linear_wgrad_deferred(g2, x2, ctx.params[0].grad, ctx.params[1].grad, G):
                return gx, None, None, None, None   # partials queued: folded with the bucket's other partials
            if G > 1:
                ga, xa = g2.view(G, n // G, -1).transpose(1, 2), x2.view(G, n // G, -1)
                try:                                # fp32 partials: the fold below is then exact fp32 accumulation
                    part = torch.bmm(ga, xa, out_dtype=torch.float32)
                except (TypeError, RuntimeError):
                    part = torch.bmm(ga, xa)
                if sink and part.dtype == torch.float32:
                    w, b = ctx.params
                    ops.colsum_into(part.view(G, -1), w.grad.view(-1))
                    ops.colsum_into(g2, b.grad)
                    ops.sink_arrived(w.grad, b.grad)
                    return gx, None, None, None, None
                if part.is_cuda and part.dtype in (torch.float32, torch.bfloat16):
                    gw = ops.colsum(part.view(G, -1)).view(part.shape[1:])      # fp32 G-way sum, fixed order
                else:
                    gw = part.sum(0, dtype=torch.float32)
            else:
                gw = g2.t().matmul(x2).float()
            gw = gw.to(ctx.param_dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if g2.is_cuda and g2.dtype in (torch.float32, torch.bfloat16) and n >= 1024 and g2.size(1) <= 1024:
                gb = ops.colsum(g2).to(ctx.param_dtype)
            else:
                gb = g2.sum(0, dtype=torch.float32).to(ctx.param_dtype)
        return gx, gw, gb, None, None


class _ReluLinearFn(torch.autograd.Function):
    """relu(x W^T + b) for a layer on the library GEMM (wider than the MFMA small-Linear kernels): bias and ReLU ride on
    the GEMM's epilogue (`torch._addmm_activation`: one launch instead of GEMM + relu, 35 -> 20 us at 32k x 256 x 512);
    the backward masks the incoming gradient with the saved output and is `_SplitKLinearFn`'s from there on."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_lp, b_lp):
        w = weight if w_lp is None else w_lp
        b = bias if b_lp is None else b_lp
        ctx.has_bias, ctx.param_dtype, ctx.params, ctx.small = True, weight.dtype, (weight, bias), False
        with torch.autocast("cuda", enabled=False):
            y = torch._addmm_activation(b, x.reshape(-1, x.size(-1)), w.t(), use_gelu=False)
        ctx.save_for_backward(x, w, y)
        return y.view(*x.shape[:-1], w.size(0))

    @staticmethod
    def backward(ctx, gy):
        y = ctx.saved_tensors[2]
        gy = torch.ops.aten.threshold_backward(gy.reshape(y.shape), y, 0).view(gy.shape)
        return _SplitKLinearFn.backward(ctx, gy)


class _LinearResidualFn(torch.autograd.Function):
    """res + (x W^T + b) for a small bf16 layer: the residual add is the epilogue of the libmivae product (the same sum
    torch forms from the bf16-rounded product); the backward is `_SplitKLinearFn`'s plus the pass-through to `res`."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_lp, b_lp, res):       # argument order of _SplitKLinearFn, the residual last
        ctx.has_bias, ctx.param_dtype, ctx.params, ctx.small = True, weight.dtype, (weight, bias), True
        ctx.hold_wgrad = True       # attention's output projection: its weight gradient shares the q / k / v group's launch
        x2 = x.contiguous().view(-1, x.size(-1))
        ctx.save_for_backward(x2, w_lp)
        ctx.xshape = x.shape
        return ops.linear_fwd_residual(x2, w_lp, b_lp, res.contiguous().view(-1, w_lp.size(0))).view(res.shape)

    @staticmethod
    def backward(ctx, gy):
        g16 = ops.to_bf16(gy.contiguous())       # the cast autograd makes for `res + y` (or the LayerNorm backward's twin)
        gx, gw, gb, _, _ = _SplitKLinearFn.backward(ctx, g16)
        return gx, gw, gb, None, None, (gy if ctx.needs_input_grad[5] else None)


class Linear(torch.nn.Linear):
    """torch.nn.Linear (same parameters / state-dict keys) whose backward splits the token axis of
    the weight-gradient GEMM when there are many tokens and few features."""

    def forward(self, x):
        if x.is_cuda and torch.is_grad_enabled() and x.numel() // x.size(-1) >= 4096 and self.out_features <= 512:
            if torch.is_autocast_enabled():
                dt = torch.get_autocast_dtype("cuda")
                w_lp = getattr(self.weight, "_mivae_lp", None)        # bf16 shadow kept by dp.FusedAdam
                if w_lp is None or w_lp.dtype != dt:
                    w_lp = self.weight.detach().to(dt)
                b_lp = None
                if self.bias is not None:
                    b_lp = getattr(self.bias, "_mivae_lp", None)
                    if b_lp is None or b_lp.dtype != dt:
                        b_lp = self.bias.detach().to(dt)
                return _SplitKLinearFn.apply(x.to(dt), self.weight, self.bias, w_lp, b_lp)
            return _SplitKLinearFn.apply(x, self.weight, self.bias, None, None)
        return super().forward(x)

    def forward_residual(self, x, res):
        """res + self(x); for a small layer on the bf16-autocast training route the add is the product's epilogue."""
        if (x.is_cuda and torch.is_grad_enabled() and x.dtype == torch.bfloat16 and self.bias is not None
                and x.numel() // x.size(-1) >= 4096 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16
                and res.dtype in (torch.float32, torch.bfloat16) and res.is_cuda
                and res.shape == (*x.shape[:-1], self.out_features)):
            w_lp, b_lp = _lp(self.weight), _lp(self.bias)
            if ops.linear_small_supported(x, w_lp):
                return _LinearResidualFn.apply(x, self.weight, self.bias, w_lp, b_lp, res)
        return res + self(x)

    def forward_relu(self, x):
        """relu(self(x)); on the wide bf16-autocast training route bias + ReLU are the GEMM's epilogue."""
        if (x.is_cuda and torch.is_grad_enabled() and x.numel() // x.size(-1) >= 4096 and self.out_features <= 1024
                and self.bias is not None and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and hasattr(torch, "_addmm_activation")):
            w_lp, b_lp = _lp(self.weight), _lp(self.bias)
            x16 = x.to(torch.bfloat16)
            if not ops.linear_small_supported(x16, w_lp):
                return _ReluLinearFn.apply(x16, self.weight, self.bias, w_lp, b_lp)
        return torch.relu(self(x))


class _GroupLinearFn(torch.autograd.Function):
    """G same-shape Linear layers on one input (q / k / v of self-attention; mu / sigma heads): one forward launch,
    one input-gradient launch (a single contraction over G * Fo), weight gradients per layer into the sinks."""

    @staticmethod
    def forward(ctx, x, G, *flat):
        ws, bs, wl, bl = flat[0:G], flat[G:2 * G], flat[2 * G:3 * G], flat[3 * G:4 * G]
        x2 = x.contiguous().view(-1, x.size(-1))
        ctx.save_for_backward(x2, *wl)
        ctx.params = list(zip(ws, bs))
        ctx.G, ctx.xshape, ctx.param_dtype = G, x.shape, ws[0].dtype
        ys = ops.linear_fwd_group(x2, list(wl), list(bl))
        return tuple(y.view(*x.shape[:-1], y.size(-1)) for y in ys)

    @staticmethod
    def backward(ctx, *gys):
        x2, *wl = ctx.saved_tensors
        G = ctx.G
        Fo = wl[0].size(0)
        gs = [torch.zeros(x2.size(0), Fo, dtype=torch.bfloat16, device=x2.device) if g is None
              else g.contiguous().view(-1, Fo).to(torch.bfloat16) for g in gys]
        gx = ops.linear_dgrad_group(gs, wl).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        if all(ops.grad_sink_ok(w, b) for w, b in ctx.params) and \
                ops.linear_wgrad_group_deferred(x2, gs, [(w.grad, b.grad) for w, b in ctx.params]):
            return (gx, None, *([None] * (4 * G)))
        gws, gbs = [], []
        for g, (w, b) in zip(gs, ctx.params):
            if ops.grad_sink_ok(w, b):
                ops.linear_wgrad(x2, g, sink=(w.grad, b.grad))
                gws.append(None)
                gbs.append(None)
            else:
                gw, gb = ops.linear_wgrad(x2, g)
                gws.append(gw.to(ctx.param_dtype))
                gbs.append(gb.to(ctx.param_dtype))
        return (gx, None, *gws, *gbs, *([None] * (2 * G)))


class _WideGroupLinearFn(torch.autograd.Function):
    """G same-shape Linear layers too wide for the MFMA small-Linear kernels on one input (q / k / v of self-attention at
    256 / 512 features): ONE library product against the row-concatenated bf16 weights, the outputs are the column blocks
    of its (N, G * Fo) result (the attention kernels take the row stride).  Backward: one input-gradient product over
    K = G * Fo -- no per-layer products, no accumulation launches -- on the packed dq | dk | dv the attention backward
    hands back, and one batched weight-gradient product whose partials every layer's fold job reads by column block."""

    @staticmethod
    def forward(ctx, x, G, *flat):
        ws, bs = flat[:G], flat[G:2 * G]
        wcat = torch.cat([_lp(w) for w in ws], 0)
        bcat = torch.cat([_lp(b) for b in bs], 0)
        x2 = x.contiguous().view(-1, x.size(-1))
        with torch.autocast("cuda", enabled=False):
            y = F.linear(x2, wcat, bcat)
        ctx.save_for_backward(x2, wcat)
        ctx.params, ctx.G, ctx.xshape, ctx.param_dtype = list(zip(ws, bs)), G, x.shape, ws[0].dtype
        Fo = ws[0].size(0)
        y3 = y.view(*x.shape[:-1], G * Fo)
        return tuple(y3[..., i * Fo:(i + 1) * Fo] for i in range(G))

    @staticmethod
    def backward(ctx, *gys):
        x2, wcat = ctx.saved_tensors
        G, N = ctx.G, x2.size(0)
        Fo = wcat.size(0) // G
        g0 = gys[0]
        packed = all(g is not None and g.dtype == torch.bfloat16 and g.dim() >= 2 and g.stride(-1) == 1
                     and g.stride(-2) == G * Fo and g.numel() == N * Fo for g in gys)
        if packed:
            es = g0.element_size()
            packed = all(g.data_ptr() == g0.data_ptr() + i * Fo * es for i, g in enumerate(gys)) and (
                g0.dim() == 2 or g0.stride(0) == g0.size(1) * G * Fo)
        if packed:               # the column blocks of one (N, G * Fo) buffer (ops.attention's packed gradients)
            gcat = torch.as_strided(g0, (N, G * Fo), (G * Fo, 1), g0.storage_offset())
        else:
            gcat = torch.cat([torch.zeros(N, Fo, dtype=torch.bfloat16, device=x2.device) if g is None
                              else g.reshape(N, Fo).to(torch.bfloat16) for g in gys], 1)
        gx = gcat.matmul(wcat).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        S = 1                                       # token slabs of the weight-gradient product (as _SplitKLinearFn)
        while S < 16 and N % (2 * S) == 0 and N // (2 * S) >= 128:
            S *= 2
        if S > 1 and all(ops.grad_sink_ok(w, b) for w, b in ctx.params) and \
                ops.wide_group_wgrad_deferred(gcat, x2, [(w.grad, b.grad) for w, b in ctx.params], S):
            return (gx, None, *([None] * (2 * G)))
        gw = gcat.t().matmul(x2).float()
        gb = ops.colsum(gcat)
        gws = [gw[i * Fo:(i + 1) * Fo].to(ctx.param_dtype) for i in range(G)]
        gbs = [gb[i * Fo:(i + 1) * Fo].to(ctx.param_dtype) for i in range(G)]
        return (gx, None, *gws, *gbs)


WIDE_GROUP = os.environ.get("MIVAE_WIDE_GROUP", "1") != "0"


def linear_group(x, layers, strided_ok=False):
    """[layer(x) for layer in layers] for same-shape `Linear` layers sharing their input; under bf16 autocast with
    many tokens the group runs as one forward and one input-gradient launch (libmivae), otherwise layer by layer.
    `strided_ok`: the caller takes outputs that are column blocks of one buffer (row stride = the group's width), which
    lets layers too wide for the libmivae kernels run as one library product (`_WideGroupLinearFn`)."""
    if (x.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled() and len(layers) >= 2
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.numel() // x.size(-1) >= 4096
            and all(isinstance(l, Linear) and l.bias is not None for l in layers)
            and all(l.out_features <= 128 and l.in_features <= 128 for l in layers)):
        wl, bl = [], []
        for l in layers:
            w_lp = getattr(l.weight, "_mivae_lp", None)
            b_lp = getattr(l.bias, "_mivae_lp", None)
            wl.append(w_lp if w_lp is not None and w_lp.dtype == torch.bfloat16 else l.weight.detach().to(torch.bfloat16))
            bl.append(b_lp if b_lp is not None and b_lp.dtype == torch.bfloat16 else l.bias.detach().to(torch.bfloat16))
        x16 = x.to(torch.bfloat16)
        if ops.linear_group_supported(x16, wl):
            return list(_GroupLinearFn.apply(x16, len(layers), *[l.weight for l in layers], *[l.bias for l in layers],
                                             *wl, *bl))
    if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and len(layers) >= 2):
        x = x.to(torch.bfloat16)        # one cast for the group instead of one per layer
        l0 = layers[0]
        if (WIDE_GROUP and strided_ok and torch.is_grad_enabled() and x.numel() // x.size(-1) >= 4096 and len(layers) <= 4
                and all(isinstance(l, Linear) and l.bias is not None and l.weight.shape == l0.weight.shape
                        and l.weight.dtype == torch.float32 for l in layers)
                and l0.out_features % 8 == 0 and l0.out_features <= 1024 and l0.in_features <= 1024):
            return list(_WideGroupLinearFn.apply(x, len(layers), *[l.weight for l in layers],
                                                 *[l.bias for l in layers]))
    return [l(x) for l in layers]


def _lp(p):
    """bf16 compute copy of an fp32 parameter: the shadow dp.FusedAdam keeps, else a cast."""
    q = getattr(p, "_mivae_lp", None)
    return q if q is not None and q.dtype == torch.bfloat16 else p.detach().to(torch.bfloat16)


class _FfnBlockFn(torch.autograd.Function):
    """x + W2 relu(W1 LN(x) + b1) + b2 (pre-norm feed-forward sublayer): forward in one libmivae launch; backward on
    the small-Linear / LayerNorm kernels with the tensors the forward saved (no recomputation)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, eps):
        x2 = x.contiguous().view(-1, x.size(-1))
        w1l, b1l, w2l, b2l = _lp(w1), _lp(b1), _lp(w2), _lp(b2)
        y, xn, hmid, stats = ops.ffn_fwd(x2, gamma, beta, w1l, b1l, w2l, b2l, eps)
        ctx.save_for_backward(x2, gamma, xn, hmid, stats, w1l, w2l)
        ctx.params = (gamma, beta, w1, b1, w2, b2)
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, xn, hmid, stats, w1l, w2l = ctx.saved_tensors
        g, be, w1, b1, w2, b2 = ctx.params
        dy2 = dy.contiguous().view(-1, dy.size(-1))
        dy16 = ops.to_bf16(dy2)                  # fp32 stream: the bf16 twin the LayerNorm backward wrote, if any
        dh = ops.linear_dgrad_relu(dy16, w2l, hmid)                        # (N, H), relu' applied on store
        grads = {}
        if (ops.grad_sink_ok(w1, b1) and ops.grad_sink_ok(w2, b2)
                and ops.linear_wgrad_pair_deferred(hmid, dy16, (w2.grad, b2.grad), xn, dh, (w1.grad, b1.grad))):
            dxn = ops.linear_dgrad(dh, w1l)          # both weight gradients are queued: one launch
            dx, gg, gb_ = ops.layer_norm_bwd(dxn, x2, gamma, stats, (g, be), add=dy2)
            return dx.view(ctx.xshape), gg, gb_, None, None, None, None, None
        for name, (xin, gout, w, b) in (("2", (hmid, dy16, w2, b2)), ("1", (xn, dh, w1, b1))):
            small = w.size(0) <= 128 and w.size(1) <= 128
            if small and ops.grad_sink_ok(w, b):
                ops.linear_wgrad(xin, gout, sink=(w.grad, b.grad))
                grads[name] = (None, None)
            elif small:
                gw, gb = ops.linear_wgrad(xin, gout)
                grads[name] = (gw.to(w.dtype), gb.to(b.dtype))
            else:                                                          # wide layer: library products
                grads[name] = (gout.t().matmul(xin).to(w.dtype), ops.colsum(gout).to(b.dtype))
        dxn = ops.linear_dgrad(dh, w1l)
        dx, gg, gb_ = ops.layer_norm_bwd(dxn, x2, gamma, stats, (g, be), add=dy2)    # + the residual branch, same store
        dx = dx.view(ctx.xshape)
        return dx, gg, gb_, grads["1"][0], grads["1"][1], grads["2"][0], grads["2"][1], None


def ffn_block_supported(x, norm, ffn, sub) -> bool:
    """bf16-autocast training bracket, inactive dropout, shapes the fused forward is instantiated for."""
    def off(d):
        return not isinstance(d, torch.nn.Dropout) or d.p == 0 or not d.training
    D = x.size(-1)
    return bool(x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.numel() // D >= 4096
                and isinstance(norm, LayerNorm) and norm.elementwise_affine and norm.bias is not None
                and norm.weight.dtype == torch.float32 and tuple(norm.normalized_shape) == (D,)
                and (D, ffn._w_1.out_features) in ops.FFN_DIMS and ffn._w_1.in_features == D
                and ffn._w_2.out_features == D and ffn._w_1.bias is not None and ffn._w_2.bias is not None
                and off(ffn._dropout) and off(sub._dropout))


# ----------------------------------------------------------------------------------- activations
def _gelu(x):
    # erf GELU, nn/activation.py:16: x * 0.5 * (1 + erf(x / sqrt 2)) -- the library's exact (erf) gelu is that
    # expression as one kernel forward and one backward instead of four + six elementwise launches
    return F.gelu(x)


ACTIVATION_FNS: Dict[str, Callable] = {
    "linear": lambda x: x, "gelu": _gelu, "swish": lambda x: x * torch.sigmoid(x), "relu": F.relu,
    "relu6": F.relu6, "elu": F.elu, "leaky_relu": F.leaky_relu, "hardtanh": F.hardtanh,
    "sigmoid": torch.sigmoid, "tanh": torch.tanh, "log_sigmoid": F.logsigmoid, "softplus": F.softplus,
    "softshrink": F.softshrink, "softsign": F.softsign, "tanhshrink": F.tanhshrink,
}


class Activation(torch.nn.Module):
    """Named activation (nn/activation.py:26-56)."""

    def __init__(self, name: str) -> None:
        super().__init__()
        if name not in ACTIVATION_FNS:
            raise Exception("Inavalid activation name.")
        self._name = name

    def forward(self, x):
        return ACTIVATION_FNS[self._name](x)

    def __repr__(self):
        return f"Activation({self._name})"

    @classmethod
    def by_name(cls, name: str) -> "Activation":
        return cls(name)


class FeedForward(torch.nn.Module, FromParams):
    """Linear/activation/dropout stack + output Linear (nn/feedforward.py:54-98)."""

    def __init__(self, input_size: int, num_layers: int, hidden_sizes: Union[int, List[int]],
                 activations, dropout: Union[float, List[float]] = 0.0, output_size: int = None):
        super().__init__()
        hidden_sizes = hidden_sizes if isinstance(hidden_sizes, list) else [hidden_sizes] * num_layers
        activations = activations if isinstance(activations, list) else [activations] * num_layers
        dropout = dropout if isinstance(dropout, list) else [dropout] * num_layers
        for name, seq in (("hidden_dims", hidden_sizes), ("activations", activations), ("dropout", dropout)):
            if len(seq) != num_layers:
                raise ValueError(f"len({name}) ({len(seq)}) != num_layers ({num_layers})")
        self._input_size = input_size
        self._output_size = output_size or hidden_sizes[-1]
        dims = [input_size] + hidden_sizes
        self._linear_layers = torch.nn.ModuleList(
            [Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self._activations = torch.nn.ModuleList(activations)
        self._dropout = torch.nn.ModuleList([torch.nn.Dropout(p) for p in dropout])
        self._output = Linear(hidden_sizes[-1], self._output_size)

    def forward(self, x):
        for lin, act, drop in zip(self._linear_layers, self._activations, self._dropout):
            x = drop(act(lin(x)))
        return self._output(x)

    def get_input_size(self):
        return self._input_size

    def get_output_size(self):
        return self._output_size


# ----------------------------------------------------------------------------------- transformer
def _maybe_dropout(p):
    return torch.nn.Dropout(p) if p and p > 0 else torch.nn.Identity()


# bf16 training bracket: use the library's fused attention kernel (set MIVAE_NO_SDPA=1 or this flag to False to keep
# the explicit masked-softmax path, e.g. to profile a HIP-graph replayed step with rocprofv3, which aborts on the
# graph-captured library attention kernel on ROCm 7.2)
USE_FUSED_ATTENTION = os.environ.get("MIVAE_NO_SDPA", "0") != "1"
USE_FUSED_FFN = True     # pre-norm feed-forward sublayer forward in one libmivae launch (bf16 autocast)
# libmivae's own attention kernels (csrc/attention.hip: head sizes 16 / 32 / 64 / 128, S <= 2048) instead of the
# library's; MIVAE_LIB_ATTENTION=1 keeps the library kernel for comparisons
USE_MIVAE_ATTENTION = os.environ.get("MIVAE_LIB_ATTENTION", "0") != "1"


class MultiHeadAttention(torch.nn.Module):
    """nn/transformer.py:7-117: separate q/k/v projections, masked_fill(-1e13) on padded keys,
    optional causal (tril) mask.  Scores are materialised like the reference (S <= 512 here)."""

    def __init__(self, num_heads, hidden_size, bias=True, dropout=0.1, output_size=None,
                 attention_fill_value=-1e13, mask_tril=False):
        super().__init__()
        assert hidden_size % num_heads == 0
        self._attn_size = hidden_size // num_heads
        self._num_heads = num_heads
        self._projections = torch.nn.ModuleDict({
            "query": Linear(hidden_size, hidden_size, bias=bias),
            "key": Linear(hidden_size, hidden_size, bias=bias),
            "value": Linear(hidden_size, hidden_size, bias=bias)})
        self._output = Linear(hidden_size, output_size or hidden_size)
        self._fill = attention_fill_value
        self._mask_tril = mask_tril
        self._dropout = _maybe_dropout(dropout)

    def _split(self, x):
        b, s, _ = x.shape
        return x.view(b, s, self._num_heads, self._attn_size).transpose(1, 2)

    def fused_ok(self, x) -> bool:
        """the library's fused attention applies (bf16 autocast bracket on the GPU, inactive attention dropout)"""
        return bool(USE_FUSED_ATTENTION and x.is_cuda and torch.is_autocast_enabled()
                    and torch.get_autocast_dtype("cuda") == torch.bfloat16
                    and not (self.training and isinstance(self._dropout, torch.nn.Dropout) and self._dropout.p > 0))

    def kernel_ok(self, x, mask) -> bool:
        """libmivae's attention kernels apply: fused_ok, a supported head size / length and a (B, S) token mask"""
        return bool(USE_MIVAE_ATTENTION and self.fused_ok(x) and x.dim() == 3
                    and (mask is None or mask.shape == x.shape[:2])
                    and ops.attention_shape_supported(x.size(0), x.size(1), self._num_heads * self._attn_size,
                                                      self._num_heads))

    def _project_out(self, t, residual):
        return self._output(t) if residual is None else self._output.forward_residual(t, residual)

    def forward(self, query, key, value, mask=None, attn_bias=None, residual=None):
        """`residual` (optional): returns residual + attention(...), the add folded into the output projection"""
        if query is key and key is value:        # self-attention: the three projections share their input
            q, k, v = linear_group(query, [self._projections[n] for n in ("query", "key", "value")], strided_ok=True)
        else:
            q = self._projections["query"](query)
            k = self._projections["key"](key)
            v = self._projections["value"](value)
        if (q.dtype == torch.bfloat16 and q.shape == k.shape and k.shape == v.shape and self.kernel_ok(q, mask)):
            # scores, reference fill (-1e13), softmax and probs @ v per head in libmivae's kernels, straight on the
            # (B, S, heads * d) projections (no head transposes, no bias tensor)
            return self._project_out(ops.attention(q, k, v, mask, self._num_heads, self._mask_tril), residual)
        q, k, v = self._split(q), self._split(k), self._split(v)
        S = q.size(2)
        if (USE_FUSED_ATTENTION and q.is_cuda and q.dtype == torch.bfloat16 and torch.is_autocast_enabled()
                and not (self.training and isinstance(self._dropout, torch.nn.Dropout) and self._dropout.p > 0)):
            # bf16 training bracket: the library's fused attention (same key-padding x causal mask, -inf instead of
            # -1e13 fill: identical for every row that has a valid key, i.e. always with prefix masks)
            if attn_bias is not None and mask is not None:
                # 0 / -inf bias built once per encoder / decoder call by ops.attention_bias (one launch for all layers)
                out = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_bias)
            elif mask is not None:
                out = F.scaled_dot_product_attention(q, k, v, attn_mask=ops.attention_bias(mask, self._mask_tril, q.dtype))
            else:
                out = F.scaled_dot_product_attention(q, k, v, is_causal=self._mask_tril)
            return self._project_out(out.transpose(1, 2).reshape(q.size(0), S, -1), residual)
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self._attn_size)
        if mask is None:
            keep = torch.ones(q.size(0), 1, 1, S, dtype=torch.bool, device=q.device)
        else:
            keep = mask.ne(0)[:, None, None, :]
        if self._mask_tril:
            keep = keep & torch.ones(S, S, dtype=torch.bool, device=q.device).tril()[None, None]
        scores = scores.masked_fill(~keep, self._fill)
        probs = self._dropout(torch.softmax(scores, dim=-1))
        out = torch.matmul(probs, v).transpose(1, 2).reshape(q.size(0), S, -1)
        return self._project_out(out, residual)


class PositionwiseFeedForward(torch.nn.Module):
    def __init__(self, input_size, hidden_size, dropout=0.0):
        super().__init__()
        self._w_1 = Linear(input_size, hidden_size)
        self._w_2 = Linear(hidden_size, input_size)
        self._dropout = _maybe_dropout(dropout)

    def forward(self, x):
        return self._w_2(self._dropout(self._w_1.forward_relu(x)))


class LayerNorm(torch.nn.LayerNorm):
    """torch.nn.LayerNorm (same parameters / state-dict keys) on the libmivae row kernels when the input is a
    GPU fp32/bf16 tensor; under bf16 autocast the output is written as bf16 directly (the next op is always a
    projection), otherwise in the input's dtype.  Anything else falls through to the stock op."""

    def kernel_dtype(self, x):
        """output dtype of the libmivae route for `x`, or None when the stock op has to run"""
        D = x.size(-1)
        if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and self.elementwise_affine
                and self.bias is not None and len(self.normalized_shape) == 1 and D % 4 == 0
                and D <= ops.LAYERNORM_MAX_D and self.weight.dtype == torch.float32 and x.numel() > 0):
            if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
                return torch.bfloat16
            return x.dtype
        return None

    def forward(self, x):
        out_dtype = self.kernel_dtype(x)
        if out_dtype is not None:
            return ops.layer_norm(x, self.weight, self.bias, self.eps, out_dtype)
        return super().forward(x)


class SublayerConnection(torch.nn.Module):
    """Pre-norm residual (nn/transformer.py:147-162)."""

    def __init__(self, input_size, dropout=0.1):
        super().__init__()
        self._norm = LayerNorm(input_size)
        self._dropout = _maybe_dropout(dropout)

    def forward(self, x, sublayer, residual_aware=False):
        """`residual_aware`: `sublayer(xn, residual)` returns residual + f(xn) itself (used when the dropout is inactive,
        so that the add can ride on the sublayer's last product)"""
        norm = self._norm
        fold = residual_aware and not (self.training and isinstance(self._dropout, torch.nn.Dropout) and self._dropout.p > 0)
        if isinstance(norm, LayerNorm) and torch.is_grad_enabled() and x.requires_grad:
            out_dtype = norm.kernel_dtype(x)
            if out_dtype is not None:
                # training: the residual branch's gradient is folded into the LayerNorm backward kernel (no add launch)
                res, xn = ops.layer_norm_residual(x, norm.weight, norm.bias, norm.eps, out_dtype)
                if fold:
                    return sublayer(xn, res)
                return res + self._dropout(sublayer(xn))
        if fold:
            return sublayer(norm(x), x)
        return x + self._dropout(sublayer(norm(x)))


class PositionalEncoding(torch.nn.Module):
    """Sinusoid table kept as a frozen Parameter named `_pos_enc` (nn/transformer.py:181-202)."""

    def __init__(self, hidden_size, dropout=0.0, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, hidden_size)
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, hidden_size, 2, dtype=torch.float) * -(math.log(10000.0) / hidden_size))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self._pos_enc = torch.nn.Parameter(pe.unsqueeze(0), requires_grad=False)
        self._dropout = _maybe_dropout(dropout)

    def forward(self, tokens):
        return self._dropout(tokens + self._pos_enc[:, :tokens.size(1)].to(tokens.dtype))

    def add_and_mask(self, tokens, mask):
        """forward(tokens) * mask[..., None]: one launch when the dropout is inactive and the input is a GPU fp32 / bf16
        (B, S, D) tensor, else the two stock expressions."""
        drop = self._dropout
        if (tokens.is_cuda and tokens.dim() == 3 and tokens.dtype in (torch.float32, torch.bfloat16) and mask is not None
                and mask.shape == tokens.shape[:2]
                and (not isinstance(drop, torch.nn.Dropout) or drop.p == 0 or not self.training)):
            return ops.posenc_mask(tokens, self._pos_enc.detach(), mask)
        return self.forward(tokens) * mask.unsqueeze(-1).to(tokens.dtype)


class _TransformerLayer(torch.nn.Module):
    def __init__(self, input_size, hidden_size, num_heads, dropout=0.0, causal=False):
        super().__init__()
        self._attn = MultiHeadAttention(num_heads, input_size, dropout=dropout, mask_tril=causal)
        self._pos_ffn = PositionwiseFeedForward(input_size, hidden_size, dropout=dropout)
        self._sublayers = torch.nn.ModuleDict({
            "attn": SublayerConnection(input_size, dropout=dropout),
            "pos_ffn": SublayerConnection(input_size, dropout=dropout)})

    def attention_bias(self, x, mask):
        """the layer stack's shared attention bias for (x, mask), or None when the explicit masked-softmax path runs"""
        if mask is None or not self._attn.fused_ok(x) or self._attn.kernel_ok(x, mask):
            return None
        return ops.attention_bias(mask, self._attn._mask_tril, torch.bfloat16)

    def forward(self, x, mask, attn_bias=None):
        x = self._sublayers["attn"](x, lambda y, r=None: self._attn(y, y, y, mask, attn_bias, residual=r),
                                    residual_aware=True)
        sub = self._sublayers["pos_ffn"]
        if USE_FUSED_FFN and ffn_block_supported(x, sub._norm, self._pos_ffn, sub):
            f = self._pos_ffn
            return _FfnBlockFn.apply(x, sub._norm.weight, sub._norm.bias, f._w_1.weight, f._w_1.bias, f._w_2.weight,
                                     f._w_2.bias, sub._norm.eps)
        return sub(x, self._pos_ffn)


class TransformerEncoderLayer(_TransformerLayer):
    def __init__(self, input_size, hidden_size, num_heads, dropout=0.0):
        super().__init__(input_size, hidden_size, num_heads, dropout, causal=False)


class TransformerDecoderLayer(_TransformerLayer):
    """Self-attention with a causal x padding mask (nn/transformer.py:275)."""

    def __init__(self, input_size, hidden_size, num_heads, dropout=0.0):
        super().__init__(input_size, hidden_size, num_heads, dropout, causal=True)


class EncoderOutput(NamedTuple):
    output: torch.Tensor
    ctx: torch.Tensor
    mask: torch.Tensor


class Embedder(torch.nn.Module, FromParams):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=0, **kwargs):
        super().__init__()
        self._embedding = torch.nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx, **kwargs)

    def forward(self, tokens):
        e = self._embedding
        if e.max_norm is None and not e.scale_grad_by_freq and not e.sparse:
            return ops.embedding(tokens, e.weight, e.padding_idx)      # stock lookup, libmivae weight gradient
        return e(tokens)

    @classmethod
    def from_params(cls, vocab, **kwargs):
        return cls(num_embeddings=vocab.get_vocab_size(), **kwargs)


class Encoder(torch.nn.Module, Registrable):
    """encoders/encoder.py:16-54; the last-token gather (a2) goes through the HIP kernel."""

    def __init__(self, input_size):
        super().__init__()
        self._input_size = input_size

    def forward(self, tokens, mask):
        from . import ops
        output = self.encoder(self._preprocess_embedding(tokens), mask)
        ctx, _ = ops.gather_last(output, mask)
        return EncoderOutput(output, ctx, mask)

    def _preprocess_embedding(self, x):
        return x

    def get_input_size(self):
        return self._input_size


@Encoder.register("transformer")
class TransformerEncoder(Encoder):
    def __init__(self, input_size, hidden_size, num_layers, num_heads, dropout=0.0):
        super().__init__(input_size)
        self._pos_enc = PositionalEncoding(input_size)
        self._layers = torch.nn.ModuleList(
            [TransformerEncoderLayer(input_size, hidden_size, num_heads, dropout) for _ in range(num_layers)])
        self._dropout = _maybe_dropout(dropout)

    def get_output_size(self):
        return self._input_size

    def _preprocess_embedding(self, x):
        return self._dropout(x)

    def encoder(self, x, mask):
        x = self._pos_enc.add_and_mask(x, mask)
        bias = self._layers[0].attention_bias(x, mask) if len(self._layers) else None
        for layer in self._layers:
            x = layer(x, mask, bias)
        if x.is_cuda and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16) and mask.shape == x.shape[:2]:
            return ops.mask_mul(x, mask)
        return x * mask.unsqueeze(-1).to(x.dtype)


# ----------------------------------------------------------------------------------- recurrent bracket
class RecurrentCore(torch.nn.Module, Registrable):
    """RNN / LSTM / GRU over padded sequences (nn/recurrent.py:8-38): pack by mask lengths, run the library
    recurrent module (MIOpen), pad back to the mask's length.  Bracket code (stock ops), not part of the path."""

    def __init__(self, module: torch.nn.Module, input_size: int) -> None:
        super().__init__()
        self._module = module
        self._input_size = input_size
        self._num_directions = 2 if getattr(module, "bidirectional", False) else 1

    def forward(self, tokens, mask):
        from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
        lengths = mask.sum(dim=1).long()
        packed = pack_padded_sequence(tokens, lengths.detach().cpu(), batch_first=True, enforce_sorted=False)
        encoded, _ = self._module(packed)
        output, _ = pad_packed_sequence(encoded, batch_first=True, total_length=mask.size(1))
        if (output.dtype == torch.float16 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            # this PyTorch-ROCm build runs the MIOpen recurrent kernels in fp16 whatever the autocast dtype is; the
            # latent path takes fp32 / bf16 (bug fix for BASELINE config 4 under bf16 autocast, bracket otherwise frozen)
            output = output.to(torch.bfloat16)
        return output

    def get_input_size(self) -> int:
        return self._input_size

    def get_output_size(self) -> int:
        return self._module.hidden_size * self._num_directions


@RecurrentCore.register("rnn")
class RNNCore(RecurrentCore):
    def __init__(self, input_size, hidden_size, num_layers=1, nonlinearity="tanh", bias=True, dropout=0.0,
                 bidirectional=False):
        super().__init__(torch.nn.RNN(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                                      nonlinearity=nonlinearity, bias=bias, batch_first=True, dropout=dropout,
                                      bidirectional=bidirectional), input_size)


@RecurrentCore.register("lstm")
class LSTMCore(RecurrentCore):
    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, dropout=0.0, bidirectional=False,
                 stateful=False):
        super().__init__(torch.nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                                       bias=bias, batch_first=True, dropout=dropout,
                                       bidirectional=bidirectional), input_size)


@RecurrentCore.register("gru")
class GRUCore(RecurrentCore):
    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, dropout=0.0, bidirectional=False,
                 stateful=False):
        super().__init__(torch.nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                                      bias=bias, batch_first=True, dropout=dropout,
                                      bidirectional=bidirectional), input_size)


class RecurrentCell(torch.nn.Module, Registrable):
    """One-step cells for the autoregressive decoder (nn/recurrent.py:126-177)."""

    def __init__(self, module: torch.nn.RNNCellBase) -> None:
        super().__init__()
        self._module = module

    def forward(self, input_tensor, hx=None):
        return self._module(input_tensor, hx)

    def get_input_size(self) -> int:
        return self._module.input_size

    def get_output_size(self) -> int:
        return self._module.hidden_size


@RecurrentCell.register("lstm")
class LSTMCell(RecurrentCell):
    def __init__(self, input_size, hidden_size, bias=True):
        super().__init__(torch.nn.LSTMCell(input_size=input_size, hidden_size=hidden_size, bias=bias))


@RecurrentCell.register("gru")
class GRUCell(RecurrentCell):
    def __init__(self, input_size, hidden_size, bias=True):
        super().__init__(torch.nn.GRUCell(input_size=input_size, hidden_size=hidden_size, bias=bias))


@Encoder.register("recurrent")
class RecurrentEncoder(Encoder):
    """encoders/rnn.py:9-69: recurrent core (+ optional feed-forward), output masked."""

    def __init__(self, core: RecurrentCore, hidden_sizes: List[int] = None, ff_activation: str = "elu",
                 dropout: float = 0.0) -> None:
        super().__init__(core.get_input_size())
        self._core = core
        self._dropout = _maybe_dropout(dropout)
        if hidden_sizes is None:
            self._feedforward = torch.nn.Identity()
            self._output_size = core.get_output_size()
        else:
            self._feedforward = FeedForward(input_size=core.get_output_size(), num_layers=len(hidden_sizes),
                                            hidden_sizes=hidden_sizes, activations=Activation.by_name(ff_activation),
                                            dropout=dropout)
            self._output_size = self._feedforward.get_output_size()

    def get_output_size(self) -> int:
        return self._output_size

    def _preprocess_embedding(self, x):
        return self._dropout(x)

    def encoder(self, x, mask):
        out = self._feedforward(self._core(x, mask))
        return out * mask.unsqueeze(-1).to(out.dtype)

    @classmethod
    def from_params(cls, **params):
        core = RecurrentCore.from_params(**params.pop("core"))
        return cls(core, **params)


class Decoder(torch.nn.Module, Registrable):
    def __init__(self, skip_z=False):
        super().__init__()
        self._skip_z = skip_z

    def forward(self, z, mask):
        out = self.decoder(z, mask)
        return torch.cat([out, z], dim=-1) if self._skip_z else out


@Decoder.register("nonauto-transformer")
class TransformerDecoder(Decoder):
    def __init__(self, input_size, hidden_size, num_layers, num_heads, dropout=0.0, skip_z=False):
        super().__init__(skip_z)
        self._input_size = input_size
        self._pos_enc = PositionalEncoding(input_size)
        self._layers = torch.nn.ModuleList(
            [TransformerDecoderLayer(input_size, hidden_size, num_heads, dropout) for _ in range(num_layers)])
        self._dropout = _maybe_dropout(dropout)

    def get_input_size(self):
        return self._input_size

    def get_output_size(self):
        return self._input_size * (2 if self._skip_z else 1)

    def decoder(self, z, mask):
        z = self._pos_enc.add_and_mask(self._dropout(z), mask)
        bias = self._layers[0].attention_bias(z, mask) if len(self._layers) else None
        for layer in self._layers:
            z = layer(z, mask, bias)
        return z


class LabelSmoothingNLL(torch.nn.Module):
    """Per-sequence token-mean NLL with optional label smoothing (models/base/losses.py:106-169)."""

    def __init__(self, smoothing=0.0, size_average=True):
        super().__init__()
        self._smoothing = smoothing
        self._size_average = size_average

    def forward(self, logits, target, weights):
        smoothing = self._smoothing if (self._smoothing is not None and self._smoothing > 0.0 and self.training) else 0.0
        per_seq = ops.label_smoothing_nll(logits, target, weights, smoothing)
        if self._size_average:
            wsum = weights.reshape(weights.size(0), -1).sum(1)
            return per_seq.sum() / wsum.gt(0).float().sum().clamp(min=1e-13)
        return per_seq
