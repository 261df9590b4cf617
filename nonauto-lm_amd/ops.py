"""torch.autograd wrappers around the libmivae C ABI (include/mivae.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and autograd bookkeeping.
Every numeric result on this path comes from a hand-written HIP kernel; inputs must live on the GPU.
"""
import math
from typing import Optional, Sequence, Tuple

import os

import torch

from . import _lib as L

ACT_IDS = {"linear": 0, "gelu": 1, "swish": 2, "relu": 3, "tanh": 4, "sigmoid": 5}
FUSED_MAX_D = 256        # small-D kernels (nice_stack.hip); bf16 latents up to D = 512 run nice_big.hip

# Optional per-launch timing of named kernels with HIP events recorded on the launch stream (used by
# bench.py for the roofline figure; no host sync is added here).
_PROFILE = None


def profile_kernels(enable: bool):
    """Start (True) / stop (False) collecting (start, end) event pairs; returns the collected dict."""
    global _PROFILE
    prev = _PROFILE
    _PROFILE = {} if enable else None
    return prev


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _PROFILE is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        if _PROFILE is not None:
            self.ev[1].record(torch.cuda.current_stream())
            _PROFILE.setdefault(self.name, []).append(self.ev)
        return False


def _mask_i64(mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Masks are int64 0/1 in the reference (nn/utils.py:12-14); the sampling path passes the same
    0/1 values as floats (priors/prior.py:116-122)."""
    if mask is None:
        return None
    if mask.dtype != torch.int64:
        mask = mask.to(torch.int64)
    return mask.contiguous()


def _shape3(z: torch.Tensor) -> Tuple[int, int, int]:
    if z.dim() == 3:
        return z.size(0), z.size(1), z.size(2)
    if z.dim() == 2:
        return z.size(0), 1, z.size(1)
    raise ValueError(f"latent must be (B, S, D) or (B, D), got {tuple(z.shape)}")


# ----------------------------------------------------------------------------------- integer helpers
def tokens_mask(tokens: torch.Tensor) -> torch.Tensor:
    """a1: x.ne(0).long()  (vae_lm/nn/utils.py:12-14)."""
    L.require_gpu(tokens)
    tokens = tokens.contiguous()
    if tokens.dtype != torch.int64:
        tokens = tokens.to(torch.int64)
    out = torch.empty_like(tokens)
    outf = torch.empty(tokens.shape, dtype=torch.float32, device=tokens.device)
    L.call("mivae_tokens_mask_f32", L.ptr(tokens), L.ptr(out), L.ptr(outf), tokens.numel(), L.stream())
    out._mivae_f32 = outf          # the same mask as fp32 token weights (mask_f32(): no cast launch for the loss)
    return out


def mask_f32(mask: torch.Tensor) -> torch.Tensor:
    """mask.float(); free when `mask` is the tensor tokens_mask() returned (its fp32 twin was written by the same kernel
    and the integer mask is never modified in place on this path: checked through the version counter)."""
    tw = getattr(mask, "_mivae_f32", None)
    if tw is not None and mask._version == 0 and tw.shape == mask.shape:
        return tw
    return mask.float()


class _GatherLast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, mask):
        B, S, E = output.shape
        output = output.contiguous()
        ctxv = torch.empty(B, E, dtype=output.dtype, device=output.device)
        idx = torch.empty(B, dtype=torch.int64, device=output.device)
        L.call("mivae_gather_last", L.ptr(output), L.ptr(mask), L.ptr(ctxv), L.ptr(idx), B, S, E,
               L.dtype_code(output.dtype), L.stream())
        ctx.save_for_backward(idx)
        ctx.shape = (B, S, E)
        ctx.mark_non_differentiable(idx)
        return ctxv, idx

    @staticmethod
    def backward(ctx, g, _gidx):
        (idx,) = ctx.saved_tensors
        B, S, E = ctx.shape
        grad = g.new_zeros(B, S, E)
        pos = torch.where(idx < 0, idx + S, idx)
        grad[torch.arange(B, device=g.device), pos] = g   # scatter of B rows: plumbing, not hot path
        return grad, None


def gather_last(output: torch.Tensor, mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """a2: ctx = output[b, mask.sum(1) - 1]  (encoders/encoder.py:35-37).  Returns (ctx, last_idx)."""
    L.require_gpu(output, mask)
    return _GatherLast.apply(output, _mask_i64(mask))


class _PosEncMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pe, mask):
        B, S, D = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        L.call("mivae_posenc_mask", L.ptr(x), L.ptr(pe), L.ptr(mask), L.ptr(y), B, S, D, 0, L.dtype_code(x.dtype), L.stream())
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        B, S, D = dy.shape
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.call("mivae_posenc_mask", L.ptr(dy), None, L.ptr(mask), L.ptr(dx), B, S, D, 1, L.dtype_code(dy.dtype), L.stream())
        return dx, None, None


def posenc_mask(x: torch.Tensor, pe: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """(x + pe[:, :S]) * mask[..., None] in one launch (bracket entry of the transformer encoder / decoder); pe is the
    fp32 (1, max_len, D) sinusoid table."""
    L.require_gpu(x, pe, mask)
    if pe.dtype != torch.float32 or not pe.is_contiguous() or pe.size(-1) != x.size(-1) or pe.size(-2) < x.size(1):
        raise ValueError("posenc_mask: pe must be a contiguous fp32 (1, >= S, D) table")
    return _PosEncMask.apply(x, pe, _mask_i64(mask))


class _MaskMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        B, S, D = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        L.call("mivae_posenc_mask", L.ptr(x), None, L.ptr(mask), L.ptr(y), B, S, D, 1, L.dtype_code(x.dtype), L.stream())
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        B, S, D = dy.shape
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.call("mivae_posenc_mask", L.ptr(dy), None, L.ptr(mask), L.ptr(dx), B, S, D, 1, L.dtype_code(dy.dtype), L.stream())
        return dx, None


def mask_mul(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """x * mask[..., None] for x (B, S, D) fp32 / bf16 and an integer (B, S) mask: one launch, no mask cast."""
    L.require_gpu(x, mask)
    return _MaskMul.apply(x, _mask_i64(mask))


def attention_bias(mask: torch.Tensor, causal: bool, dtype: torch.dtype) -> torch.Tensor:
    """Additive attention bias of a (B, S) token mask: 0 / -inf, (B, 1, 1, S) or (B, 1, S, S) when causal (one launch;
    the bracket builds it once per encoder / decoder call and hands it to every layer)."""
    L.require_gpu(mask)
    mask = _mask_i64(mask)
    B, S = mask.shape
    out = torch.empty((B, 1, S, S) if causal else (B, 1, 1, S), dtype=dtype, device=mask.device)
    L.call("mivae_attn_bias", L.ptr(mask), L.ptr(out), B, S, int(bool(causal)), L.dtype_code(dtype), L.stream())
    return out


def attention_shape_supported(B: int, S: int, W: int, heads: int) -> bool:
    """head size W / heads in {16, 32, 64, 128}, 1 <= S <= 2048, B and heads within the launch grid"""
    return bool(heads > 0 and W % heads == 0 and 0 < B <= 65535 and heads <= 65535
                and L.query("mivae_attention_supported", W // heads, S))


def attention_supported(x: torch.Tensor, heads: int) -> bool:
    """libmivae's attention kernels take (B, S, heads * d) bf16 GPU operands of a supported shape"""
    return bool(x.is_cuda and x.dim() == 3 and x.dtype == torch.bfloat16
                and attention_shape_supported(x.size(0), x.size(1), x.size(2), heads))


def _att_operand(t):
    """(B, S, W) bf16 view the kernels can address: unit feature stride, 16-byte aligned rows"""
    if t.stride(2) != 1 or t.stride(1) % 8 or t.stride(0) % 8 or t.data_ptr() % 16 or t.stride(1) < t.size(2):
        t = t.contiguous()
    return t


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, heads, causal):
        q, k, v = _att_operand(q), _att_operand(k), _att_operand(v)
        B, S, W = q.shape
        hd = W // heads
        scale = 1.0 / math.sqrt(hd)
        need = any(ctx.needs_input_grad[:3])
        out = torch.empty(B, S, W, dtype=torch.bfloat16, device=q.device)
        lse = torch.empty(B, heads, S, dtype=torch.float32, device=q.device) if need else None
        L.call("mivae_attention_fwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), L.ptr(out), L.ptr(lse), B, S, heads, hd,
               q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), int(causal), scale,
               L.stream())
        if need:
            ctx.save_for_backward(q, k, v, mask, out, lse)
            ctx.geo = (heads, hd, int(causal), scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, mask, out, lse = ctx.saved_tensors
        heads, hd, causal, scale = ctx.geo
        B, S, W = q.shape
        dout = dout.contiguous()
        if dout.dtype != torch.bfloat16:
            dout = dout.to(torch.bfloat16)
        delta = torch.empty_like(lse)
        es = q.element_size()
        packed = (q.stride(1) == 3 * W and k.stride(1) == 3 * W and v.stride(1) == 3 * W
                  and k.data_ptr() == q.data_ptr() + W * es and v.data_ptr() == k.data_ptr() + W * es)
        if packed:
            # q, k, v are the column blocks of one fused q|k|v projection: hand its backward ONE (B, S, 3 W) gradient
            # (dq | dk | dv per token), the K-concatenated operand of a single input-gradient product
            g = torch.empty(B, S, 3 * W, dtype=torch.bfloat16, device=q.device)
            dq, dk, dv = g[..., :W], g[..., W:2 * W], g[..., 2 * W:]
        else:
            g = torch.empty(3, B, S, W, dtype=torch.bfloat16, device=q.device)
            dq, dk, dv = g[0], g[1], g[2]
        L.call("mivae_attention_bwd", L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), L.ptr(out), L.ptr(dout), L.ptr(lse),
               L.ptr(delta), L.ptr(dq), L.ptr(dk), L.ptr(dv), B, S, heads, hd, q.stride(0), q.stride(1),
               k.stride(0), k.stride(1), v.stride(0), v.stride(1), 3 * W if packed else W, causal, scale, L.stream())
        return dq, dk, dv, None, None, None


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: Optional[torch.Tensor], heads: int,
              causal: bool = False) -> torch.Tensor:
    """softmax(masked_fill(q k^T / sqrt(d), mask == 0 [or key > query], -1e13)) v per head for q, k, v (B, S, heads * d)
    bf16 (the projections' outputs, any row stride) and a (B, S) token mask: MultiHeadAttention.forward's core
    (nn/transformer.py:62-117) without the score tensor, forward and backward on libmivae's kernels."""
    L.require_gpu(q, k, v, mask)
    if not (attention_supported(q, heads) and k.shape == q.shape and v.shape == q.shape
            and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16):
        raise ValueError(f"ops.attention: unsupported operands {tuple(q.shape)} {q.dtype} with {heads} heads")
    if mask is not None:
        mask = _mask_i64(mask)
        if mask.shape != q.shape[:2]:
            raise ValueError(f"ops.attention: mask {tuple(mask.shape)} does not match (B, S) = {tuple(q.shape[:2])}")
    return _Attention.apply(q, k, v, mask, int(heads), bool(causal))


def segment_sum(tok: torch.Tensor, nseg: int, S: int, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty(nseg, dtype=torch.float32, device=tok.device)
    L.call("mivae_segment_sum", L.ptr(tok), L.ptr(add), L.ptr(out), nseg, S, L.stream())
    return out


def mask_count(mask: torch.Tensor, nseg: int, mask_div: int = 1) -> torch.Tensor:
    mask = _mask_i64(mask)
    out = torch.empty(nseg, dtype=torch.int64, device=mask.device)
    L.call("mivae_mask_count", L.ptr(mask), L.ptr(out), nseg, mask_div, mask.size(-1), L.stream())
    return out


# ----------------------------------------------------------------------------------- posterior
class _ReparamLogQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, sigma, eps, mask, n):
        B, S, D = _shape3(mu)
        # under autocast the sigma head (softplus) comes out fp32 while mu is bf16: z's dtype follows mu
        mu_c, sg_c = mu.contiguous(), sigma.contiguous().to(mu.dtype)
        eps_c = None if eps is None else eps.contiguous().to(mu.dtype)
        ctx.sigma_dtype = sigma.dtype
        z = torch.empty((B * n, *mu.shape[1:]), dtype=mu.dtype, device=mu.device)
        tok = torch.empty(B * n * S, dtype=torch.float32, device=mu.device)
        L.call("mivae_reparam_logq_fwd", L.ptr(mu_c), L.ptr(sg_c), L.ptr(eps_c), L.ptr(mask), L.ptr(z),
               L.ptr(tok), B, n, S, D, L.dtype_code(mu.dtype), L.stream())
        logq = segment_sum(tok, B * n, S)
        ctx.save_for_backward(sg_c, eps_c, mask)
        ctx.dims = (B, n, S, D)
        return z, logq

    @staticmethod
    def backward(ctx, dz, dlogq):
        sg, eps, mask = ctx.saved_tensors
        B, n, S, D = ctx.dims
        dz = None if dz is None else dz.contiguous()
        dlogq = None if dlogq is None else dlogq.contiguous().float()
        dmu = torch.empty_like(sg)
        dsg = torch.empty_like(sg)
        L.call("mivae_reparam_logq_bwd", L.ptr(dz), L.ptr(dlogq), L.ptr(sg), L.ptr(eps), L.ptr(mask),
               L.ptr(dmu), L.ptr(dsg), B, n, S, D, L.dtype_code(sg.dtype), L.stream())
        return dmu, dsg.to(ctx.sigma_dtype), None, None, None


# In-kernel posterior noise (Philox): one generator state {seed, offset} per device, living in device memory (a captured
# graph advances it by itself).  It follows torch's CUDA generator: whenever that generator's (seed, offset) is not
# what this module left behind -- torch.manual_seed(), or any torch random op in between -- the device state is
# re-based on it, so torch.manual_seed() makes runs reproducible; the seed is mixed with the rank so that every rank
# draws different noise.  Every eager call also advances torch's offset (host-side bookkeeping only).
PHILOX = True
_PHILOX_STATE = {}


def philox_state(device) -> torch.Tensor:
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    ent = _PHILOX_STATE.get(key)
    if ent is not None and torch.cuda.is_current_stream_capturing():
        return ent["state"]                 # no generator queries inside a capture: the device state carries on
    gen = torch.cuda.default_generators[key]
    cur = (gen.initial_seed(), gen.get_offset())
    if ent is None or ent["next"] != cur:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        mixed = (cur[0] + rank * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF
        st = torch.tensor([mixed, cur[1] << 16], dtype=torch.int64).to(device)
        if ent is None:
            ent = _PHILOX_STATE[key] = {"state": st}
        else:                               # keep the address (it may be baked into a captured graph)
            ent["state"].copy_(st)
    gen.set_offset(cur[1] + 4)
    ent["next"] = (cur[0], cur[1] + 4)
    return ent["state"]


def _philox_peek(device) -> torch.Tensor:
    """the device generator state without the host-side bookkeeping of philox_state() (same tensor)"""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    ent = _PHILOX_STATE.get(key)
    return ent["state"] if ent is not None else philox_state(device)


class _ReparamLogQRng(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, sigma, mask, n):
        B, S, D = _shape3(mu)
        mu_c, sg_c = mu.contiguous(), sigma.contiguous().to(mu.dtype)
        ctx.sigma_dtype = sigma.dtype
        z = torch.empty((B * n, *mu.shape[1:]), dtype=mu.dtype, device=mu.device)
        tok = torch.empty(B * n * S, dtype=torch.float32, device=mu.device)
        used = torch.empty(2, dtype=torch.int64, device=mu.device)
        L.call("mivae_reparam_logq_fwd_rng", L.ptr(mu_c), L.ptr(sg_c), L.ptr(philox_state(mu.device)), L.ptr(used),
               L.ptr(mask), L.ptr(z), L.ptr(tok), B, n, S, D, L.dtype_code(mu.dtype), L.stream())
        logq = segment_sum(tok, B * n, S)
        ctx.save_for_backward(sg_c, used, mask)
        ctx.dims = (B, n, S, D)
        return z, logq

    @staticmethod
    def backward(ctx, dz, dlogq):
        sg, used, mask = ctx.saved_tensors
        B, n, S, D = ctx.dims
        dz = None if dz is None else dz.contiguous()
        dlogq = None if dlogq is None else dlogq.contiguous().float()
        dmu = torch.empty_like(sg)
        dsg = torch.empty_like(sg)
        L.call("mivae_reparam_logq_bwd_rng", L.ptr(dz), L.ptr(dlogq), L.ptr(sg), L.ptr(used), L.ptr(mask),
               L.ptr(dmu), L.ptr(dsg), B, n, S, D, L.dtype_code(sg.dtype), L.stream())
        return dmu, dsg.to(ctx.sigma_dtype), None, None


def _tok_cotangent(dtok, Bn: int, S: int):
    """(tensor, per_token flag) for the kernels from the cotangent of a (Bn, S) per-token output.  The ELBO tail's
    backward hands back ONE value per sentence as an expanded (stride-0) view; the kernels index it per sentence, so no
    (Bn, S) copy is made.  Anything else is a genuine per-token cotangent."""
    if dtok is None:
        return None, False
    if (dtok.dim() == 2 and dtok.shape == (Bn, S) and dtok.dtype == torch.float32
            and (S == 1 or dtok.stride(1) == 0) and (Bn == 1 or dtok.stride(0) == 1)):
        return dtok[:, 0], False
    return dtok.contiguous().float().reshape(-1), True


class _PosteriorTok(torch.autograd.Function):
    """mivae_posterior_fwd / _bwd: (mu, sigma or the sigma head's pre-activation) -> (z, sigma, per-token log q (Bn, S)),
    noise injected (eps), drawn in-kernel (rng) or absent; ONE launch each way."""

    @staticmethod
    def forward(ctx, mu, sg, eps, mask, n, is_pre, lo, hi, rng, advance):
        B, S, D = _shape3(mu)
        mu_c, sg_c = mu.contiguous(), sg.contiguous().to(mu.dtype)
        eps_c = None if eps is None else eps.contiguous().to(mu.dtype)
        ctx.sg_dtype = sg.dtype
        z = torch.empty((B * n, *mu.shape[1:]), dtype=mu.dtype, device=mu.device)
        sigma = torch.empty_like(mu_c) if is_pre else sg_c
        tok = torch.empty(B * n, S, dtype=torch.float32, device=mu.device)
        used = state = None
        if rng and eps_c is None:
            used = torch.empty(2, dtype=torch.int64, device=mu.device)
            state = philox_state(mu.device)
        L.call("mivae_posterior_fwd", L.ptr(mu_c), L.ptr(sg_c), int(is_pre), lo, hi, L.ptr(eps_c), L.ptr(state), L.ptr(used),
               L.ptr(mask), L.ptr(z), L.ptr(sigma) if is_pre else None, L.ptr(tok), B, n, S, D, L.dtype_code(mu.dtype),
               L.stream())
        if state is not None and advance:   # the draw is taken now; advance = False: the ELBO tail of this forward does it
            L.call("mivae_rng_advance", L.ptr(state), L.stream())
        ctx.save_for_backward(sg_c, eps_c, used, mask)
        ctx.meta = (B, n, S, D, int(is_pre), lo, hi)
        ctx.set_materialize_grads(False)     # unused outputs (sigma, usually) come back as None, not as zero tensors
        return z, sigma, tok

    @staticmethod
    def backward(ctx, dz, dsigma_up, dtok):
        sg, eps, used, mask = ctx.saved_tensors
        B, n, S, D, is_pre, lo, hi = ctx.meta
        dz = None if dz is None else dz.contiguous().to(sg.dtype)
        dq, per_tok = _tok_cotangent(dtok, B * n, S)
        dmu = torch.empty_like(sg)
        dsg = torch.empty_like(sg)
        L.call("mivae_posterior_bwd", L.ptr(dz), L.ptr(dq), int(per_tok), L.ptr(sg), is_pre, lo, hi, L.ptr(eps), L.ptr(used),
               L.ptr(mask), L.ptr(dmu), L.ptr(dsg), B, n, S, D, L.dtype_code(sg.dtype), L.stream())
        if dsigma_up is not None:           # someone differentiates through the returned sigma itself (not the ELBO path)
            if is_pre:
                extra = torch.empty_like(sg)
                L.call("mivae_sigma_head_bwd", L.ptr(sg), L.ptr(dsigma_up.contiguous().to(sg.dtype)), L.ptr(extra),
                       sg.numel(), lo, hi, L.dtype_code(sg.dtype), L.stream())
                dsg = dsg + extra
            else:
                dsg = dsg + dsigma_up.to(sg.dtype)
        return dmu, dsg.to(ctx.sg_dtype), None, None, None, None, None, None, None, None


def posterior_tok(mu, sigma_or_pre, eps, mask, samples: int, is_pre: bool = False, lo: float = 1e-4, hi: float = 5.0,
                  rng: bool = False, advance: bool = True):
    """a3 tail + a4 + a5 (posterior.py:23-27, 86-172) in one launch: returns (z ((B n), ...), sigma (B, ...), per-token
    log q ((B n), S)).  `sigma_or_pre` is sigma, or (is_pre) the sigma head's pre-activation; noise = `eps` if given, else
    drawn inside the kernel when `rng`, else zero.  With `rng` the generator state moves on by one draw right after the
    kernel (one single-thread launch); `advance=False` leaves that to the caller -- latent_tail(..., rng_advance=True)
    of the same forward does it inside the ELBO tail kernel, rng_advance() by itself."""
    L.require_gpu(mu, sigma_or_pre, eps, mask)
    return _PosteriorTok.apply(mu, sigma_or_pre, eps, _mask_i64(mask), int(samples), bool(is_pre), float(lo), float(hi),
                               bool(rng), bool(advance))


def rng_advance(device) -> None:
    """one draw taken from the in-kernel noise generator of `device` (see posterior_tok(advance=False))"""
    L.call("mivae_rng_advance", L.ptr(_philox_peek(device)), L.stream())


class _SigmaHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre, lo, hi):
        pre = pre.contiguous()
        sigma = torch.empty_like(pre)
        L.call("mivae_sigma_head_fwd", L.ptr(pre), L.ptr(sigma), pre.numel(), lo, hi, L.dtype_code(pre.dtype), L.stream())
        ctx.save_for_backward(pre)
        ctx.bounds = (lo, hi)
        return sigma

    @staticmethod
    def backward(ctx, dsigma):
        (pre,) = ctx.saved_tensors
        lo, hi = ctx.bounds
        dsigma = dsigma.contiguous().to(pre.dtype)
        dpre = torch.empty_like(pre)
        L.call("mivae_sigma_head_bwd", L.ptr(pre), L.ptr(dsigma), L.ptr(dpre), pre.numel(), lo, hi,
               L.dtype_code(pre.dtype), L.stream())
        return dpre, None, None


def sigma_head(pre: torch.Tensor, lo: float = 1e-4, hi: float = 5.0) -> torch.Tensor:
    """a3: sigma = Hardtanh(Softplus(pre), lo, hi) (posterior.py:23-27) in one launch, in `pre`'s dtype (fp32 / bf16;
    computed in fp32)."""
    L.require_gpu(pre)
    return _SigmaHead.apply(pre, float(lo), float(hi))


def reparam_logq_rng(mu, sigma, mask, samples: int):
    """reparam_logq with eps ~ N(0, 1) drawn inside the kernels (no randn launch, no eps tensor)."""
    L.require_gpu(mu, sigma, mask)
    return _ReparamLogQRng.apply(mu, sigma, _mask_i64(mask), int(samples))


def philox_normal(used: torch.Tensor, count: int) -> torch.Tensor:
    out = torch.empty(count, dtype=torch.float32, device=used.device)
    L.call("mivae_philox_normal", L.ptr(used), L.ptr(out), count, L.stream())
    return out


def reparam_logq(mu, sigma, eps, mask, samples: int):
    """a4 + a5: z = (mu + eps * sigma) * mask (rearranged to (B n) ...), log q(z) per sample.

    posterior.py:119-133 and :157-172 (nonauto); auto :111-119, :141-156 (mask None, 2-D)."""
    L.require_gpu(mu, sigma, eps, mask)
    return _ReparamLogQ.apply(mu, sigma, eps, _mask_i64(mask), int(samples))


# ----------------------------------------------------------------------------------- base log-prob
class _GaussLogProb(torch.autograd.Function):
    """per-token base log-prob (Bn, S); the cotangent comes back per sentence (expanded view) or per token"""

    @staticmethod
    def forward(ctx, z, mask, mask_div, mu, std):
        Bn, S, D = _shape3(z)
        z = z.contiguous()
        tok = torch.empty(Bn, S, dtype=torch.float32, device=z.device)
        L.call("mivae_gauss_logprob_fwd", L.ptr(z), L.ptr(mask), L.ptr(tok), Bn, mask_div, S, D,
               float(mu), float(std), L.dtype_code(z.dtype), L.stream())
        ctx.save_for_backward(z, mask)
        ctx.meta = (Bn, S, D, mask_div, float(mu), float(std))
        return tok

    @staticmethod
    def backward(ctx, dtok):
        z, mask = ctx.saved_tensors
        Bn, S, D, mask_div, mu, std = ctx.meta
        if dtok is None:
            return None, None, None, None, None
        dz = torch.empty_like(z)
        dl, per_tok = _tok_cotangent(dtok, Bn, S)
        if per_tok:       # one "sentence" per token: the mask is indexed per token (un-repeated masks are expanded first)
            if mask is not None and mask_div != 1:
                mask = mask.repeat_interleave(mask_div, dim=0).contiguous()
            L.call("mivae_gauss_logprob_bwd", L.ptr(z), L.ptr(mask), L.ptr(dl), L.ptr(dz), Bn * S, 1, 1, D, mu, std,
                   L.dtype_code(z.dtype), L.stream())
        else:
            L.call("mivae_gauss_logprob_bwd", L.ptr(z), L.ptr(mask), L.ptr(dl.contiguous()), L.ptr(dz), Bn, mask_div, S, D,
                   mu, std, L.dtype_code(z.dtype), L.stream())
        return dz, None, None, None, None


def gauss_logprob_tok(z, mask, mu: float, std: float, mask_div: int = 1):
    """a15 per token: Normal(mu, std).log_prob(z) summed over the features, times the mask -> (Bn, S) fp32."""
    L.require_gpu(z, mask)
    return _GaussLogProb.apply(z, _mask_i64(mask), mask_div, mu, std)


def gauss_logprob(z, mask, mu: float, std: float, mask_div: int = 1):
    """a15: sum over all but batch of Normal(mu, std).log_prob(z) * mask  (priors/prior.py:148-158)."""
    Bn, S, _ = _shape3(z)
    return _SegmentSum.apply(gauss_logprob_tok(z, mask, mu, std, mask_div), Bn, S)


# ----------------------------------------------------------------------------------- NICE affine tail
class _NiceAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, s, t, mask, inverse):
        Bn, S, D = _shape3(z)
        z, s, t = z.contiguous(), s.contiguous().to(z.dtype), t.contiguous().to(z.dtype)
        out = torch.empty_like(z)
        tok = torch.empty(Bn * S, dtype=torch.float32, device=z.device)
        L.call("mivae_nice_affine_fwd", L.ptr(z), L.ptr(s), L.ptr(t), L.ptr(mask), L.ptr(out), L.ptr(tok),
               Bn, 1, S, D, L.dtype_code(z.dtype), int(inverse), L.stream())
        ctx.save_for_backward(z, s, t, mask)
        ctx.meta = (Bn, S, D, int(inverse))
        return out, tok

    @staticmethod
    def backward(ctx, dout, dtok):
        z, s, t, mask = ctx.saved_tensors
        Bn, S, D, inverse = ctx.meta
        dz, ds, dt = torch.empty_like(z), torch.empty_like(s), torch.empty_like(t)
        if dout is None:
            dout = torch.zeros_like(z)
        # the kernel takes d(log_det) per *segment*; per-token cotangents are handled with S = 1
        dld = None if dtok is None else dtok.contiguous().float()
        L.call("mivae_nice_affine_bwd", L.ptr(dout.contiguous()), L.ptr(dld), L.ptr(z), L.ptr(s), L.ptr(t),
               L.ptr(mask), L.ptr(dz), L.ptr(ds), L.ptr(dt), Bn * S, 1, 1, D, L.dtype_code(z.dtype),
               inverse, L.stream())
        return dz, ds, dt, None, None


def nice_affine(z, s, t, mask, inverse: bool = False):
    """a8/a9 tail: y1 = exp(s) x1 + t (or inverse), cat, mask multiply, masked log-det.

    couplings.py:151-162 / :172-183.  Returns (z', log_det) with the reference's log_det shape:
    (B,) for masked 3-D input and for 2-D input, (B, S) for 3-D input without a mask."""
    L.require_gpu(z, s, t, mask)
    mask = _mask_i64(mask)
    Bn, S, D = _shape3(z)
    out, tok = _NiceAffine.apply(z, s, t, mask, inverse)
    if z.dim() == 2:
        return out, tok
    if mask is None:
        return out, tok.view(Bn, S)
    return out, _SegmentSum.apply(tok, Bn, S)


# ----------------------------------------------------------------------------------- MAF / IAF (8f-2)
MAF_ACT_IDS = dict(ACT_IDS, elu=6)
MAF_HIDDEN_PER_LANE = 16


class _MafAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, st, mask, parity):
        Bn, S, D = _shape3(z)
        z, st = z.contiguous(), st.contiguous().to(z.dtype)
        out = torch.empty_like(z)
        tok = torch.empty(Bn * S, dtype=torch.float32, device=z.device)
        L.call("mivae_maf_affine_fwd", L.ptr(z), L.ptr(st), L.ptr(mask), L.ptr(out), L.ptr(tok), Bn * S, D,
               int(parity), L.dtype_code(z.dtype), L.stream())
        ctx.save_for_backward(z, st, mask)
        ctx.meta = (Bn * S, D, int(parity))
        return out, tok

    @staticmethod
    def backward(ctx, dout, dtok):
        z, st, mask = ctx.saved_tensors
        rows, D, parity = ctx.meta
        dz, dst = torch.empty_like(z), torch.empty_like(st)
        dout = None if dout is None else dout.contiguous().to(z.dtype)
        dtok = None if dtok is None else dtok.contiguous().float()
        L.call("mivae_maf_affine_bwd", L.ptr(dout), L.ptr(dtok), L.ptr(z), L.ptr(st), L.ptr(mask), L.ptr(dz),
               L.ptr(dst), rows, D, parity, L.dtype_code(z.dtype), L.stream())
        return dz, dst, None, None


def _tok_logdet(tok, z, mask, Bn, S):
    """Per-token log-dets -> the reference's shape: (B,) for masked 3-D and for 2-D input, (B, S) otherwise."""
    if z.dim() == 2:
        return tok
    if mask is None:
        return tok.view(Bn, S)
    return _SegmentSum.apply(tok, Bn, S)


def maf_affine(z, st, mask, parity: bool):
    """Tail of MAFlow.forward given st = MADE(z) (maf.py:35-45): flip_if_parity(z * exp(s) + t) * mask and the masked
    log-det sum(s)."""
    L.require_gpu(z, st, mask)
    mask = _mask_i64(mask)
    Bn, S, D = _shape3(z)
    if st.shape != (*z.shape[:-1], 2 * D):
        raise ValueError(f"MADE output must be {(*z.shape[:-1], 2 * D)}, got {tuple(st.shape)}")
    if mask is not None and mask.numel() != Bn * S:
        raise ValueError("mask must be (B, S) for a (B, S, D) latent")
    out, tok = _MafAffine.apply(z, st, mask, bool(parity))
    return out, _tok_logdet(tok, z, mask, Bn, S)


def maf_sequential_pack(w1, m1, b1, w2, m2, b2):
    """Packed fp32 weights of a one-hidden-layer MADE for mivae_maf_sequential (layout in include/mivae.h), or None
    when the shape is unsupported.  w*/m* are the (out, in) weight and weight_mask tensors (made.py:39-40)."""
    H, D = w1.shape
    lanes = 1
    while lanes * MAF_HIDDEN_PER_LANE < H:
        lanes *= 2
    Hp = lanes * MAF_HIDDEN_PER_LANE
    if not L.query("mivae_maf_sequential_supported", D, Hp):
        return None
    w1t = torch.zeros(D, Hp, dtype=torch.float32, device=w1.device)
    w1t[:, :H] = (w1.detach().float() * m1).t()
    b1p = torch.zeros(Hp, dtype=torch.float32, device=w1.device)
    b1p[:H] = b1.detach().float()
    w2p = torch.zeros(2 * D, Hp, dtype=torch.float32, device=w1.device)
    w2p[:, :H] = w2.detach().float() * m2
    return w1t, b1p, w2p, b2.detach().float().contiguous(), Hp


class _MafSequential(torch.autograd.Function):
    """mivae_maf_sequential with autograd: the backward walks the recurrence in reverse (mivae_maf_sequential_bwd) and
    needs nothing but the forward's output."""

    @staticmethod
    def forward(ctx, z, mask, act, parity, w1, m1, b1, w2, m2, b2):
        Bn, S, D = _shape3(z)
        packed = maf_sequential_pack(w1, m1, b1, w2, m2, b2)
        w1t, b1p, w2p, b2p, Hp = packed
        zc = z.contiguous()
        out = torch.empty_like(zc)
        tok = torch.empty(Bn * S, dtype=torch.float32, device=z.device)
        L.call("mivae_maf_sequential", L.ptr(zc), L.ptr(w1t), L.ptr(b1p), L.ptr(w2p), L.ptr(b2p), L.ptr(mask), L.ptr(out),
               L.ptr(tok), Bn * S, D, Hp, act, int(parity), L.dtype_code(z.dtype), L.stream())
        ctx.save_for_backward(out, mask, w1t, b1p, w2p, b2p, m1, m2)
        ctx.meta = (Bn * S, D, Hp, w1.size(0), act, int(parity), [t.dtype for t in (w1, b1, w2, b2)])
        return out, tok

    @staticmethod
    def backward(ctx, dout, dtok):
        out, mask, w1t, b1p, w2p, b2p, m1, m2 = ctx.saved_tensors
        rows, D, Hp, H, act, parity, dts = ctx.meta
        dout = None if dout is None else dout.contiguous().to(out.dtype)
        dtok = None if dtok is None else dtok.contiguous().float()
        dz = torch.empty_like(out)
        nb = L.query("mivae_maf_sequential_bwd_blocks", rows)
        P = 3 * D * Hp + Hp + 2 * D
        part = torch.empty(nb, P, dtype=torch.float32, device=out.device)
        L.call("mivae_maf_sequential_bwd", L.ptr(out), L.ptr(dout), L.ptr(dtok), L.ptr(w1t), L.ptr(b1p), L.ptr(w2p),
               L.ptr(b2p), L.ptr(mask), L.ptr(dz), L.ptr(part), rows, D, Hp, act, parity, L.dtype_code(out.dtype),
               L.stream())
        sums = colsum(part)
        gw1t = sums[:D * Hp].view(D, Hp)
        gw2 = sums[D * Hp:3 * D * Hp].view(2 * D, Hp)
        gb1 = sums[3 * D * Hp:3 * D * Hp + Hp]
        gb2 = sums[3 * D * Hp + Hp:]
        # the kernel differentiates the MASKED weights (made.py:39-40: weight * weight_mask)
        dw1 = (gw1t[:, :H].t() * m1).to(dts[0])
        dw2 = (gw2[:, :H] * m2).to(dts[2])
        return dz, None, None, None, dw1, None, gb1[:H].to(dts[1]), dw2, None, gb2.to(dts[3])


def maf_sequential_grad_supported(w1) -> bool:
    H, D = w1.shape
    lanes = 1
    while lanes * MAF_HIDDEN_PER_LANE < H:
        lanes *= 2
    return bool(L.query("mivae_maf_sequential_bwd_supported", D, lanes * MAF_HIDDEN_PER_LANE))


def maf_sequential(z, packed, mask, activation: str, parity: bool, params=None):
    """MAFlow.backward / IAFlow.forward (maf.py:47-62) in one launch.  `packed` (maf_sequential_pack): no autograd
    (sampling / inference).  `params` = (w1, m1, b1, w2, m2, b2) instead: differentiable w.r.t. z and the MADE
    parameters (IAF training), backward = one reverse-recurrence launch + a fixed-order fold."""
    L.require_gpu(z, mask)
    mask = _mask_i64(mask)
    Bn, S, D = _shape3(z)
    if mask is not None and mask.numel() != Bn * S:
        raise ValueError("mask must be (B, S) for a (B, S, D) latent")
    if params is not None:
        out, tok = _MafSequential.apply(z, mask, MAF_ACT_IDS[activation], bool(parity), *params)
        if z.dim() == 2:
            return out, tok
        if mask is None:
            return out, tok.view(Bn, S)
        return out, _SegmentSum.apply(tok, Bn, S)
    w1t, b1, w2, b2, Hp = packed
    zc = z.detach().contiguous()
    out = torch.empty_like(zc)
    tok = torch.empty(Bn * S, dtype=torch.float32, device=z.device)
    L.call("mivae_maf_sequential", L.ptr(zc), L.ptr(w1t), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(mask), L.ptr(out),
           L.ptr(tok), Bn * S, D, Hp, MAF_ACT_IDS[activation], int(parity), L.dtype_code(z.dtype), L.stream())
    if z.dim() == 2:
        return out, tok
    if mask is None:
        return out, tok.view(Bn, S)
    return out, segment_sum(tok, Bn, S)


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, nseg, S):
        ctx.meta = (nseg, S, tok.dim())
        return segment_sum(tok.contiguous(), nseg, S)

    @staticmethod
    def backward(ctx, g):
        nseg, S, dim = ctx.meta
        ge = g.contiguous().reshape(nseg, 1).expand(nseg, S)   # stride-0 view: the producing kernel indexes it per sentence
        return (ge if dim == 2 else ge.reshape(-1)), None, None


# ----------------------------------------------------------------------------------- ActNorm
class _ActNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, log_scale, bias, mask, inverse, count_weighted):
        Bn, S, D = _shape3(z)
        z = z.contiguous()
        ls, b = log_scale.contiguous().float(), bias.contiguous().float()
        out = torch.empty_like(z)
        nld = Bn if count_weighted else 1
        ld = torch.empty(nld, dtype=torch.float32, device=z.device)
        L.call("mivae_actnorm_fwd", L.ptr(z), L.ptr(ls), L.ptr(b), L.ptr(mask), L.ptr(out), L.ptr(ld),
               Bn, 1, S, D, nld, int(count_weighted), L.dtype_code(z.dtype), int(inverse), L.stream())
        ctx.save_for_backward(z, ls, mask)
        ctx.meta = (Bn, S, D, int(inverse), bool(count_weighted))
        return out, ld

    @staticmethod
    def backward(ctx, dout, dld):
        z, ls, mask = ctx.saved_tensors
        Bn, S, D, inverse, cw = ctx.meta
        if inverse:
            raise RuntimeError("ActNorm: autograd through the inverse (sampling) direction is not "
                               "supported; the reference only uses it under no_grad")
        if dout is None:
            dout = torch.zeros_like(z)
        dz = torch.empty_like(z)
        dparams = torch.empty(2, D, dtype=torch.float32, device=z.device)
        nws = L.query("mivae_col_workspace_floats", Bn * S, D, L.dtype_code(z.dtype))
        ws = torch.empty(nws, dtype=torch.float32, device=z.device)
        L.call("mivae_actnorm_bwd", L.ptr(dout.contiguous()), L.ptr(z), L.ptr(ls), L.ptr(mask), L.ptr(dz),
               L.ptr(dparams), L.ptr(ws), Bn, 1, S, D, L.dtype_code(z.dtype), L.stream())
        dls, db = dparams[0], dparams[1]
        if dld is not None:
            # log_det = sum(log_scale) * count  ->  every d gets sum_b dld[b] * count[b]
            if cw:
                cnt = mask_count(mask, Bn).float()
                dls = dls + (dld.float() * cnt).sum()
            else:
                dls = dls + dld.float().sum()
        return dz, dls, db, None, None, None


def actnorm(z, log_scale, bias, mask, inverse: bool = False):
    """a12/a13: ActNorm.forward (actnorm.py:31-39) / .backward (:45-54) after initialisation."""
    L.require_gpu(z, log_scale, bias, mask)
    mask = _mask_i64(mask)
    cw = z.dim() > 2 and mask is not None
    return _ActNorm.apply(z, log_scale, bias, mask, inverse, cw)


def actnorm_stats(z) -> Tuple[torch.Tensor, torch.Tensor]:
    """Column mean and unbiased std over z.view(-1, D) (actnorm.py:23-24), two passes, fp32."""
    L.require_gpu(z)
    D = z.size(-1)
    z2 = z.contiguous().view(-1, D)
    rows = z2.size(0)
    mean = torch.empty(D, dtype=torch.float32, device=z.device)
    std = torch.empty(D, dtype=torch.float32, device=z.device)
    ws = torch.empty(L.query("mivae_col_workspace_floats", rows, D, L.dtype_code(z.dtype)),
                     dtype=torch.float32, device=z.device)
    L.call("mivae_actnorm_stats", L.ptr(z2), L.ptr(mean), L.ptr(std), L.ptr(ws), rows, D,
           L.dtype_code(z.dtype), L.stream())
    return mean, std


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        D = x.size(-1)
        x2 = x.contiguous().view(-1, D)
        rows = x2.size(0)
        y = torch.empty(rows, D, dtype=out_dtype, device=x.device)
        stats = torch.empty(2, rows, dtype=torch.float32, device=x.device)
        L.call("mivae_layernorm_fwd", L.ptr(x2), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(stats[0]), L.ptr(stats[1]),
               rows, D, float(eps), L.dtype_code(x2.dtype), L.dtype_code(out_dtype), L.stream())
        ctx.save_for_backward(x2, weight, stats)
        ctx.shape = x.shape
        ctx.params = (weight, bias)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, stats = ctx.saved_tensors
        rows, D = x2.shape
        dx, gw, gb = layer_norm_bwd(dy.contiguous().view(rows, D), x2, weight, stats, ctx.params)
        return dx.view(ctx.shape), gw, gb, None, None


class _LayerNormResidual(torch.autograd.Function):
    """(x, LayerNorm(x)) for a pre-norm residual sublayer y = x + f(LN(x)): the backward receives the residual branch's
    gradient together with the norm's and writes their sum in the LayerNorm backward kernel's one store."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        D = x.size(-1)
        x2 = x.contiguous().view(-1, D)
        rows = x2.size(0)
        y = torch.empty(rows, D, dtype=out_dtype, device=x.device)
        stats = torch.empty(2, rows, dtype=torch.float32, device=x.device)
        L.call("mivae_layernorm_fwd", L.ptr(x2), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(stats[0]), L.ptr(stats[1]),
               rows, D, float(eps), L.dtype_code(x2.dtype), L.dtype_code(out_dtype), L.stream())
        ctx.save_for_backward(x2, weight, stats)
        ctx.shape = x.shape
        ctx.params = (weight, bias)
        return x, y.view(x.shape)

    @staticmethod
    def backward(ctx, dres, dy):
        x2, weight, stats = ctx.saved_tensors
        rows, D = x2.shape
        if dy is None:
            return dres, None, None, None, None
        add = None if dres is None else dres.contiguous().view(rows, D)
        dx, gw, gb = layer_norm_bwd(dy.contiguous().view(rows, D), x2, weight, stats, ctx.params, add=add)
        return dx.view(ctx.shape), gw, gb, None, None


def layer_norm_residual(x, weight, bias, eps: float, out_dtype):
    L.require_gpu(x, weight, bias)
    return _LayerNormResidual.apply(x, weight, bias, float(eps), out_dtype)


def layer_norm_bwd(dy2, x2, weight, stats, params, add=None):
    """(dx, d gamma, d beta) of layer_norm for dy2, x2 (rows, D); with gradient sinks active the parameter gradients
    go straight into params' .grad (deferred: folded by flush_grad_sinks) and None is returned for them.
    `add` (rows, D): gradient of a residual branch around the norm, included in dx (inside the kernel's store on the
    deferred-sink route the training step takes, else by a separate add)."""
    if add is not None:
        if add.dtype != x2.dtype or not add.is_contiguous():
            add = add.to(x2.dtype).contiguous()
        dx, gw, gb = _layer_norm_bwd(dy2, x2, weight, stats, params, add)
        return dx, gw, gb
    return _layer_norm_bwd(dy2, x2, weight, stats, params, None)


# bf16 twin of the latest fp32 residual-stream gradient (written by the LayerNorm backward kernel).  The cache holds the
# fp32 tensor itself, so its storage cannot be handed to another tensor while the entry lives; a consumer gets the twin
# only for a tensor with the same storage start, shape, dtype and version -- i.e. the same, unmodified values.
BF16_TWIN = True
_TWIN = None


def _set_twin(t32, t16) -> None:
    global _TWIN
    _TWIN = (t32, t32._version, t16)


def to_bf16(t: torch.Tensor) -> torch.Tensor:
    """t.to(bfloat16), without a launch when `t` is the fp32 gradient whose bf16 twin the LayerNorm backward just wrote"""
    if t.dtype == torch.bfloat16:
        return t
    tw = _TWIN
    if (tw is not None and t.dtype == torch.float32 and t.data_ptr() == tw[0].data_ptr() and t.numel() == tw[0].numel()
            and t.is_contiguous() and t._version == tw[1] and tw[0]._version == tw[1]):
        return tw[2].view(t.shape)
    return t.to(torch.bfloat16)


def _layer_norm_bwd(dy2, x2, weight, stats, params, add):
    rows, D = x2.shape
    dx = torch.empty_like(x2)
    ws = torch.empty(L.query("mivae_layernorm_workspace_floats", rows, D), dtype=torch.float32, device=x2.device)
    if grad_sink_ok(*params):
        gw, gb = params[0].grad, params[1].grad
        key = (gw.data_ptr(), rows, -D)
        if _DEFER is not None and key not in _DEFER:      # partials now, folded with the Linear ones after backward
            wsp = _PART_BUFS.get(key)
            if wsp is None or wsp.device != x2.device:
                wsp = _PART_BUFS[key] = torch.empty_like(ws)
            if add is not None and BF16_TWIN and x2.dtype == torch.float32 and D <= 128:     # widths of the small-Linear kernels,
                                                                                        # whose backward asks for it (to_bf16)
                # fp32 residual stream: dx also lands as bf16, for the previous sublayer's projection backward
                dx16 = torch.empty(rows, D, dtype=torch.bfloat16, device=x2.device)
                nblk = L.query("mivae_layernorm_bwd_partial_add_twin", L.ptr(dy2), L.ptr(x2), L.ptr(weight), L.ptr(stats[0]),
                               L.ptr(stats[1]), L.ptr(add), L.ptr(dx), L.ptr(dx16), L.ptr(wsp), rows, D,
                               L.dtype_code(x2.dtype), L.dtype_code(dy2.dtype), L.stream())
                _set_twin(dx, dx16)
            elif add is not None:
                nblk = L.query("mivae_layernorm_bwd_partial_add", L.ptr(dy2), L.ptr(x2), L.ptr(weight), L.ptr(stats[0]),
                               L.ptr(stats[1]), L.ptr(add), L.ptr(dx), L.ptr(wsp), rows, D, L.dtype_code(x2.dtype),
                               L.dtype_code(dy2.dtype), L.stream())
            else:
                nblk = L.query("mivae_layernorm_bwd_partial", L.ptr(dy2), L.ptr(x2), L.ptr(weight), L.ptr(stats[0]),
                               L.ptr(stats[1]), L.ptr(dx), L.ptr(wsp), rows, D, L.dtype_code(x2.dtype),
                               L.dtype_code(dy2.dtype), L.stream())
            if nblk <= 0:
                raise RuntimeError(f"mivae_layernorm_bwd_partial failed ({nblk}): {L.lib().mivae_last_error().decode()}")
            _DEFER[key] = (wsp[:nblk * 2 * D].view(nblk, 2 * D), gw, gb, D)
            sink_arrived(gw, gb)
            return dx, None, None
        L.call("mivae_layernorm_bwd_sink", L.ptr(dy2), L.ptr(x2), L.ptr(weight), L.ptr(stats[0]), L.ptr(stats[1]),
               L.ptr(dx), L.ptr(gw), L.ptr(gb), L.ptr(ws), rows, D, L.dtype_code(x2.dtype), L.dtype_code(dy2.dtype),
               L.stream())
        sink_arrived(gw, gb)
        return (dx if add is None else dx + add), None, None
    dgb = torch.empty(2, D, dtype=torch.float32, device=x2.device)
    L.call("mivae_layernorm_bwd", L.ptr(dy2), L.ptr(x2), L.ptr(weight), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(dx),
           L.ptr(dgb), L.ptr(ws), rows, D, L.dtype_code(x2.dtype), L.dtype_code(dy2.dtype), L.stream())
    return (dx if add is None else dx + add), dgb[0], dgb[1]


FFN_DIMS = ((64, 128), (64, 256), (128, 128), (128, 256))


def ffn_fwd(x2, gamma, beta, w1, b1, w2, b2, eps: float):
    """y = x + W2 relu(W1 LN(x) + b1) + b2 in one launch (pre-norm feed-forward sublayer, nn/transformer.py:120-162);
    returns (y, xn, hmid, stats) with the tensors the backward needs.  x2 (N, D) fp32 / bf16; w*, b* bf16."""
    N, D = x2.shape
    H = w1.size(0)
    y = torch.empty_like(x2)
    xn = torch.empty(N, D, dtype=torch.bfloat16, device=x2.device)
    hmid = torch.empty(N, H, dtype=torch.bfloat16, device=x2.device)
    stats = torch.empty(2, N, dtype=torch.float32, device=x2.device)
    L.call("mivae_ffn_fwd", L.ptr(x2), L.ptr(gamma), L.ptr(beta), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(y),
           L.ptr(xn), L.ptr(hmid), L.ptr(stats[0]), L.ptr(stats[1]), N, D, H, float(eps), L.dtype_code(x2.dtype),
           L.stream())
    return y, xn, hmid, stats


LAYERNORM_MAX_D = 1024


def layer_norm(x, weight, bias, eps: float = 1e-5, out_dtype=None):
    """torch.nn.functional.layer_norm over the last axis (fp32 affine parameters); `out_dtype` lets the fp32
    residual stream be normalised straight into bf16 for the projection that follows."""
    L.require_gpu(x, weight, bias)
    return _LayerNorm.apply(x, weight, bias, eps, out_dtype or x.dtype)


# Gradient sinks: when enabled (step.TrainStep turns it on), the weight-gradient kernels of the custom Functions add
# their results straight into `param.grad` (the reducer's flat-bucket views) and return None for those inputs, so
# autograd launches no per-parameter accumulation kernel.  Post-accumulate hooks do not fire for such parameters
# (GradReducer.finish() flushes every bucket anyway).
GRAD_SINK = False


def grad_sink_ok(*params) -> bool:
    if not GRAD_SINK:
        return False
    for p in params:
        g = getattr(p, "grad", None)
        if g is None or g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous():
            return False
    return True


# Deferred sinks: between begin_deferred_sinks() and flush_grad_sinks() (TrainStep brackets forward + backward with
# them) the [dW | db] partials of the small Linear layers stay in persistent per-layer buffers and ONE launch folds all
# of them into the gradient buckets at the end of backward (instead of one ~3.7 us launch per layer).
_DEFER = None
_PART_BUFS = {}
_JOB_TABLES = {}

# Device tables (pointer tables, job tables, partial buffers) whose ADDRESSES a captured HIP graph bakes in must never be
# freed while that graph can still be replayed: the caching allocator would hand the memory to another tensor and the
# replayed kernels would read garbage as gradient addresses.  step.TrainStep pins the caches (no eviction while any
# captured step is alive) and keeps its own references to their entries.
_PIN = 0


def pin_tables(on: bool) -> None:
    global _PIN
    _PIN = max(_PIN + (1 if on else -1), 0)


def pinned_tables():
    """Strong references to every cached device table / partial buffer (held by the TrainStep that captured them)."""
    return (list(_PTR_CACHE.values()), list(_JOB_TABLES.values()), list(_PART_BUFS.values()))


# SINK_NOTIFY(grad_tensor): set by step.TrainStep to dp.GradReducer.notify_sink -- "a contribution to this .grad has been
# enqueued on the current stream" (sinks bypass autograd's accumulation, so the reducer's grad hooks do not fire for them).
# Bumped whenever parameters are rewritten through raw pointers (FusedAdam's kernel, graph replays): such writes do
# not move torch's per-tensor version counters, so caches derived from parameter values key on this as well.
PARAM_EPOCH = [0]


def params_changed() -> None:
    PARAM_EPOCH[0] += 1


SINK_NOTIFY = None


def sink_arrived(*grads) -> None:
    if SINK_NOTIFY is not None:
        for g in grads:
            if g is not None:
                SINK_NOTIFY(g)


# Queued weight gradients.  In deferred-sink mode the small-Linear weight-gradient problems of a backward pass are not
# launched where autograd reaches them: they feed nothing but the gradient buckets, so they are queued (inputs kept
# alive, partial buffers registered with the fold, arrival reported to the reducer) and issued together -- up to 16
# problems per launch, mivae_linear_wgrad_multi -- by flush_grad_sinks() right before anything is folded or reduced,
# on the same stream: partials -> fold -> reduce keeps its order.  WGRAD_QUEUE = False restores launch-at-once (with the
# hand-paired launches: feed-forward pair, output projection + q / k / v group).
WGRAD_QUEUE = True
_QUEUE = []               # (x2, [dy...], [partial...])
_SINGLE_SHAPES = {(32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64)}      # (Fo, Fi) of the multi launch
_PENDING = None           # WGRAD_QUEUE off: the output projection's launch waiting for its q / k / v group


def _launch_one(x2, gs, parts):
    N, Fi = x2.shape
    Fo = gs[0].size(1)
    if len(gs) == 1:
        L.call("mivae_linear_wgrad", L.ptr(x2), L.ptr(gs[0]), L.ptr(parts[0]), N, Fi, Fo, L.stream())
    else:
        L.call("mivae_linear_wgrad_group", L.ptr(x2), _ptr_array(gs), _ptr_array(parts), len(gs), N, Fi, Fo, L.stream())


def _launch_pending() -> None:
    global _PENDING, _QUEUE
    if _PENDING is not None:
        x2, g2, key, part, sink = _PENDING
        _PENDING = None
        _launch_one(x2, [g2], [part])
        if _DEFER is not None:
            _DEFER[key] = (part, sink[0], sink[1], g2.size(1) * x2.size(1))
    if not _QUEUE:
        return
    import ctypes
    probs, _QUEUE = _QUEUE, []
    by = {}
    for pr in probs:
        by.setdefault((pr[0].size(0), len(pr[1]) > 1), []).append(pr)
    for (N, group), lst in by.items():
        for i in range(0, len(lst), 16):
            chunk = lst[i:i + 16]
            flat = []
            for x2, gs, parts in chunk:
                flat += [gs[0].size(1), x2.size(1), len(gs), x2.data_ptr()]
                flat += [g.data_ptr() for g in gs] + [0] * (4 - len(gs))
                flat += [p_.data_ptr() for p_ in parts] + [0] * (4 - len(parts))
            desc = (ctypes.c_int64 * len(flat))(*flat)
            rc = L.query("mivae_linear_wgrad_multi", desc, len(chunk), N, int(group), L.stream())
            if rc == -3:                       # a shape the multi launch is not instantiated for: one by one
                for x2, gs, parts in chunk:
                    _launch_one(x2, gs, parts)
            elif rc != 0:
                raise RuntimeError(f"mivae_linear_wgrad_multi failed ({rc}): {L.lib().mivae_last_error().decode()}")


def begin_deferred_sinks():
    global _DEFER, _PENDING, _QUEUE, _TWIN
    _DEFER = {}
    _PENDING = None
    _QUEUE = []
    _TWIN = None


def deferred_jobs_in_range(defer, lo: int, hi: int):
    """Keys of the deferred fold jobs (part, g0, g1, split) that write into the address range [lo, hi).  A combined
    [dW | db] / [d gamma | d beta] job has TWO destinations (g0 = weight.grad, g1 = bias.grad) and a byte-limit bucket
    split can put them into different buckets (bias at the end of bucket i, weight at the start of bucket i + 1): the job
    is selected when EITHER lies in the range, so it is folded whole before the EARLIER of its two buckets is reduced --
    the other bucket cannot have gone out yet, because the job's arrival was counted for it at the same time and its own
    flush would have selected the job."""
    return [k for k, j in defer.items() if lo <= j[1].data_ptr() < hi or lo <= j[2].data_ptr() < hi]


def flush_grad_sinks(lo: Optional[int] = None, hi: Optional[int] = None):
    """Fold the queued [dW | db] partials into their gradient buffers.  With an address range only the jobs with a
    destination in [lo, hi) are folded and the queue stays open (the reducer flushes a bucket right before its
    all-reduce); without one everything left is folded and deferred mode ends."""
    global _DEFER
    if _DEFER is None:
        return
    _launch_pending()
    if lo is None:
        jobs, _DEFER = _DEFER, None
    else:
        jobs = {k: _DEFER.pop(k) for k in deferred_jobs_in_range(_DEFER, lo, hi)}
    if not jobs:
        return
    rows = [(part.data_ptr(), g0.data_ptr(), g1.data_ptr(), part.size(0), part.size(1), split, part.stride(0))
            for part, g0, g1, split in jobs.values()]
    key = tuple(rows)
    table = _JOB_TABLES.get(key)
    if table is None:
        dev = next(iter(jobs.values()))[0].device
        if torch.cuda.is_current_stream_capturing():      # no host-to-device copy inside a capture: fold one by one
            for part, g0, g1, split in jobs.values():
                part = part if part.stride(0) == part.size(1) else part.contiguous()
                ws = torch.empty(max(1, L.query("mivae_col_workspace_floats", part.size(0), part.size(1), 0)),
                                 dtype=torch.float32, device=dev)
                L.call("mivae_colsum_sink", L.ptr(part), L.ptr(g0), L.ptr(g1), split, L.ptr(ws), part.size(0), part.size(1), 0,
                       L.stream())
            return
        if len(_JOB_TABLES) > 16 and _PIN == 0:
            _JOB_TABLES.clear()
        table = torch.tensor(rows, dtype=torch.int64).to(dev)
        _JOB_TABLES[key] = table
    L.call("mivae_colsum_sink_multi", L.ptr(table), len(rows), max(r[4] for r in rows), L.stream())


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight, padding_idx):
        ctx.save_for_backward(ids)
        ctx.weight = weight
        ctx.pad = -1 if padding_idx is None else int(padding_idx)
        return torch.nn.functional.embedding(ids, weight.detach())

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        w = ctx.weight
        V, E = w.shape
        dy2 = dy.contiguous().view(-1, E)
        idc = ids.contiguous().view(-1)
        if grad_sink_ok(w):
            L.call("mivae_embedding_bwd", L.ptr(dy2), L.ptr(idc), L.ptr(w.grad), idc.numel(), V, E, ctx.pad, 1, L.stream())
            sink_arrived(w.grad)
            return None, None, None
        dw = torch.empty(V, E, dtype=torch.float32, device=dy.device)
        L.call("mivae_embedding_bwd", L.ptr(dy2), L.ptr(idc), L.ptr(dw), idc.numel(), V, E, ctx.pad, 0, L.stream())
        return None, dw, None


def embedding(ids, weight, padding_idx=None):
    """torch.nn.functional.embedding whose weight gradient is the deterministic sort-free libmivae scatter-add
    (fp32 weights of width <= 512 on the GPU); anything else goes to the stock op."""
    if (weight.is_cuda and weight.dtype == torch.float32 and weight.size(1) <= 512 and ids.dtype == torch.int64
            and torch.is_grad_enabled() and weight.requires_grad and weight.is_contiguous()):
        return _Embedding.apply(ids, weight, padding_idx)
    return torch.nn.functional.embedding(ids, weight, padding_idx)


LINEAR_DIMS = (32, 64, 128, 256)


def linear_small_supported(x, weight) -> bool:
    """bf16 GPU tensors, many tokens, feature counts the MFMA Linear kernels are instantiated for."""
    # 256-wide layers stay on the library: at that width its 256x256 macro tiles run the product in ~20 us where the
    # one-tile-per-wave kernels here take 43 us (forward) / 250 us (input gradient: W transposed per block), C3 profile
    return bool(x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
                and weight.size(0) in LINEAR_DIMS and weight.size(1) in LINEAR_DIMS
                and max(weight.size(0), weight.size(1)) <= 128 and weight.is_contiguous()
                and weight.data_ptr() % 16 == 0)


def linear_fwd(x2, w16, b16):
    """y = x2 w16^T + b16 for x2 (N, Fi) bf16 contiguous (a3 posterior heads, bracket projections)."""
    N, Fi = x2.shape
    Fo = w16.size(0)
    y = torch.empty(N, Fo, dtype=torch.bfloat16, device=x2.device)
    L.call("mivae_linear_fwd", L.ptr(x2), L.ptr(w16), L.ptr(b16), L.ptr(y), N, Fi, Fo, L.stream())
    return y


def linear_fwd_residual(x2, w16, b16, res2):
    """res2 + (x2 w16^T + b16) for x2 (N, Fi) bf16 and a residual (N, Fo) fp32 / bf16, the add as the product's epilogue
    (the same sum torch forms from the bf16-rounded product)."""
    N, Fi = x2.shape
    Fo = w16.size(0)
    out = torch.empty_like(res2)
    L.call("mivae_linear_fwd_residual", L.ptr(x2), L.ptr(w16), L.ptr(b16), L.ptr(res2), L.ptr(out), N, Fi, Fo,
           L.dtype_code(res2.dtype), L.stream())
    return out


def linear_dgrad(g2, w16):
    N, Fo = g2.shape
    Fi = w16.size(1)
    dx = torch.empty(N, Fi, dtype=torch.bfloat16, device=g2.device)
    L.call("mivae_linear_dgrad", L.ptr(g2), L.ptr(w16), L.ptr(dx), N, Fi, Fo, L.stream())
    return dx


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def linear_group_supported(x, weights) -> bool:
    """2..4 same-shape layers on one input, each eligible for the MFMA small-Linear kernels, G * Fo <= 256."""
    w0 = weights[0]
    return bool(2 <= len(weights) <= 4 and all(w.shape == w0.shape for w in weights)
                and all(linear_small_supported(x, w) for w in weights) and len(weights) * w0.size(0) <= 256)


def linear_fwd_group(x2, ws16, bs16):
    """[x2 w^T + b for (w, b) in the group] in one launch: q / k / v projections, mu / sigma heads (a3)."""
    N, Fi = x2.shape
    Fo = ws16[0].size(0)
    ys = [torch.empty(N, Fo, dtype=torch.bfloat16, device=x2.device) for _ in ws16]
    L.call("mivae_linear_fwd_group", L.ptr(x2), _ptr_array(ws16), _ptr_array(bs16), _ptr_array(ys), len(ws16), N, Fi, Fo,
           L.stream())
    return ys


def linear_dgrad_group(gs, ws16):
    """sum_g gs[g] @ ws16[g] as one contraction."""
    N, Fo = gs[0].shape
    Fi = ws16[0].size(1)
    dx = torch.empty(N, Fi, dtype=torch.bfloat16, device=gs[0].device)
    L.call("mivae_linear_dgrad_group", _ptr_array(gs), _ptr_array(ws16), L.ptr(dx), len(gs), N, Fi, Fo, L.stream())
    return dx


def linear_wgrad_group_deferred(x2, gs, sinks) -> bool:
    """[dW | db] partials of a group of same-shape layers sharing x2 in ONE launch, queued for flush_grad_sinks().
    Only in deferred-sink mode and for the instantiated shapes; returns False if the caller must go layer by layer."""
    N, Fi = x2.shape
    Fo = gs[0].size(1)
    rows = L.query("mivae_linear_wgrad_rows", N)
    D = Fo * Fi + Fo
    if _DEFER is None or rows > 512 or Fi != Fo or Fi not in (32, 64) or not 2 <= len(gs) <= 4:
        return False
    keys = [(s0.data_ptr(), rows, D) for s0, _ in sinks]
    if any(k in _DEFER for k in keys):
        return False
    parts = []
    for k in keys:
        part = _PART_BUFS.get(k)
        if part is None or part.device != x2.device:
            part = _PART_BUFS[k] = torch.empty(rows, D, dtype=torch.float32, device=x2.device)
        parts.append(part)
    global _PENDING
    pend = _PENDING
    if WGRAD_QUEUE and len(gs) in (2, 3):
        _QUEUE.append((x2, list(gs), parts))
    elif (pend is not None and len(gs) == 3 and pend[0].shape == (N, Fi) and pend[1].shape == (N, Fo)
            and pend[2] not in keys):
        _PENDING = None          # the held-back single layer (output projection) rides along: one launch for four layers
        L.call("mivae_linear_wgrad_single_group", L.ptr(pend[0]), L.ptr(pend[1]), L.ptr(pend[3]), L.ptr(x2), _ptr_array(gs),
               _ptr_array(parts), len(gs), N, Fi, L.stream())
        _DEFER[pend[2]] = (pend[3], pend[4][0], pend[4][1], Fo * Fi)
    else:
        _launch_pending()
        L.call("mivae_linear_wgrad_group", L.ptr(x2), _ptr_array(gs), _ptr_array(parts), len(gs), N, Fi, Fo, L.stream())
    for k, part, (s0, s1) in zip(keys, parts, sinks):
        _DEFER[k] = (part, s0, s1, Fo * Fi)
        sink_arrived(s0, s1)
    return True


def wide_linear_wgrad_deferred(g2, x2, w_grad, b_grad, G: int) -> bool:
    """Weight / bias gradient of a Linear too wide for the MFMA small-Linear kernels, in deferred-sink mode: the token-slab
    partials of d W (one batched library GEMM into a persistent buffer) and the per-block partial column sums of d y are
    queued, and flush_grad_sinks() folds them -- with every other queued partial of the bucket -- in one launch.
    Returns False when the caller has to fold by itself (no deferred mode, layer used twice in the step, odd shapes)."""
    n, Fo = g2.shape
    Fi = x2.size(1)
    if _DEFER is None or n % G or G > 512:
        return False
    code = L.dtype_code(g2.dtype)
    kw, kb = (w_grad.data_ptr(), G, Fo * Fi), (b_grad.data_ptr(), n, -Fo)
    if kw in _DEFER or kb in _DEFER:
        return False
    nws = L.query("mivae_col_workspace_floats", n, Fo, code)
    if nws <= 0 or nws // Fo > 2 * 512:            # more row blocks than the fold kernel takes per job
        return False
    part = _PART_BUFS.get(kw)
    if part is None or part.device != g2.device:
        part = _PART_BUFS[kw] = torch.empty(G, Fo, Fi, dtype=torch.float32, device=g2.device)
    wsb = _PART_BUFS.get(kb)
    if wsb is None or wsb.device != g2.device:
        wsb = _PART_BUFS[kb] = torch.empty(nws, dtype=torch.float32, device=g2.device)
    ga, xa = g2.view(G, n // G, Fo).transpose(1, 2), x2.view(G, n // G, Fi)
    try:
        torch.bmm(ga, xa, out_dtype=torch.float32, out=part)
    except (TypeError, RuntimeError):
        part.copy_(torch.bmm(ga, xa, out_dtype=torch.float32))
    nblk = L.query("mivae_colsum_partial", L.ptr(g2), L.ptr(wsb), n, Fo, code, L.stream())
    if nblk <= 0:
        raise RuntimeError(f"mivae_colsum_partial failed ({nblk}): {L.lib().mivae_last_error().decode()}")
    if nblk > 512:
        raise RuntimeError("wide_linear_wgrad_deferred: partial count exceeds the fold kernel's limit")
    wflat = w_grad.view(-1)
    _DEFER[kw] = (part.view(G, Fo * Fi), wflat, wflat, Fo * Fi)
    _DEFER[kb] = (wsb[:nblk * Fo].view(nblk, Fo), b_grad, b_grad, Fo)
    sink_arrived(w_grad, b_grad)
    return True


def wide_group_wgrad_deferred(gcat, x2, sinks, G: int) -> bool:
    """wide_linear_wgrad_deferred for a group of same-shape layers sharing x2 whose output gradients are the column
    blocks of gcat (n, len(sinks) * Fo): ONE batched product for the token-slab partials of all d W, one partial column
    sum for all d b; every layer's fold job reads its column block of the shared partials (row stride in the job)."""
    n, Fi = x2.shape
    L3 = gcat.size(1)
    ng = len(sinks)
    Fo = L3 // ng
    if _DEFER is None or n % G or G > 512 or not gcat.is_contiguous() or not x2.is_contiguous():
        return False
    code = L.dtype_code(gcat.dtype)
    kw, kb = (sinks[0][0].data_ptr(), G, -L3 * Fi), (sinks[0][1].data_ptr(), n, -L3)
    kws = [(w.data_ptr(), G, Fo * Fi) for w, _ in sinks]
    kbs = [(b.data_ptr(), n, -Fo) for _, b in sinks]
    if any(k in _DEFER for k in kws + kbs):
        return False
    nws = L.query("mivae_col_workspace_floats", n, L3, code)
    if nws <= 0 or nws // L3 > 2 * 512:
        return False
    part = _PART_BUFS.get(kw)
    if part is None or part.device != gcat.device:
        part = _PART_BUFS[kw] = torch.empty(G, L3, Fi, dtype=torch.float32, device=gcat.device)
    wsb = _PART_BUFS.get(kb)
    if wsb is None or wsb.device != gcat.device:
        wsb = _PART_BUFS[kb] = torch.empty(nws, dtype=torch.float32, device=gcat.device)
    ga, xa = gcat.view(G, n // G, L3).transpose(1, 2), x2.view(G, n // G, Fi)
    try:
        torch.bmm(ga, xa, out_dtype=torch.float32, out=part)
    except (TypeError, RuntimeError):
        part.copy_(torch.bmm(ga, xa, out_dtype=torch.float32))
    nblk = L.query("mivae_colsum_partial", L.ptr(gcat), L.ptr(wsb), n, L3, code, L.stream())
    if nblk <= 0:
        raise RuntimeError(f"mivae_colsum_partial failed ({nblk}): {L.lib().mivae_last_error().decode()}")
    if nblk > 512:
        raise RuntimeError("wide_group_wgrad_deferred: partial count exceeds the fold kernel's limit")
    pw = part.view(G, ng, Fo * Fi)
    pb = wsb[:nblk * L3].view(nblk, ng, Fo)
    for i, (w_grad, b_grad) in enumerate(sinks):
        wflat = w_grad.view(-1)
        _DEFER[kws[i]] = (pw[:, i], wflat, wflat, Fo * Fi)         # (G, Fo * Fi) with row stride ng * Fo * Fi
        _DEFER[kbs[i]] = (pb[:, i], b_grad, b_grad, Fo)
        sink_arrived(w_grad, b_grad)
    return True


def linear_dgrad_relu(g2, w16, relu_out):
    """(g2 @ w16) * (relu_out > 0): input gradient of a Linear whose input was a relu output, one launch."""
    N, Fo = g2.shape
    Fi = w16.size(1)
    dx = torch.empty(N, Fi, dtype=torch.bfloat16, device=g2.device)
    L.call("mivae_linear_dgrad_relu", L.ptr(g2), L.ptr(w16), L.ptr(relu_out), L.ptr(dx), N, Fi, Fo, L.stream())
    return dx


def linear_wgrad_pair_deferred(x0, g0, sink0, x1, g1, sink1) -> bool:
    """linear_wgrad(x0, g0, sink=sink0) and linear_wgrad(x1, g1, sink=sink1) -- the two layers of a feed-forward
    sublayer -- as ONE launch in deferred-sink mode; False when the caller has to issue them separately."""
    N = x0.size(0)
    (Fi0, Fo0), (Fi1, Fo1) = (x0.size(1), g0.size(1)), (x1.size(1), g1.size(1))
    rows = L.query("mivae_linear_wgrad_rows", N)
    if _DEFER is None or rows > 512 or (Fi0, Fo0, Fi1, Fo1) != (128, 64, 64, 128) or x1.size(0) != N:
        return False
    k0, k1 = (sink0[0].data_ptr(), rows, Fo0 * Fi0 + Fo0), (sink1[0].data_ptr(), rows, Fo1 * Fi1 + Fo1)
    if k0 in _DEFER or k1 in _DEFER or k0 == k1:
        return False
    parts = []
    for k in (k0, k1):
        part = _PART_BUFS.get(k)
        if part is None or part.device != x0.device:
            part = _PART_BUFS[k] = torch.empty(rows, k[2], dtype=torch.float32, device=x0.device)
        parts.append(part)
    if WGRAD_QUEUE:
        _QUEUE.append((x0, [g0], [parts[0]]))
        _QUEUE.append((x1, [g1], [parts[1]]))
    else:
        L.call("mivae_linear_wgrad_pair", L.ptr(x0), L.ptr(g0), L.ptr(parts[0]), Fi0, Fo0, L.ptr(x1), L.ptr(g1),
               L.ptr(parts[1]), Fi1, Fo1, N, L.stream())
    _DEFER[k0] = (parts[0], sink0[0], sink0[1], Fo0 * Fi0)
    _DEFER[k1] = (parts[1], sink1[0], sink1[1], Fo1 * Fi1)
    sink_arrived(sink0[0], sink0[1])
    sink_arrived(sink1[0], sink1[1])
    return True


def linear_wgrad(x2, g2, sink=None, hold=False):
    """(dW (Fo, Fi), db (Fo)) fp32 from x2 (N, Fi), g2 (N, Fo) bf16; Fi, Fo <= 128.  `sink` = (weight.grad,
    bias.grad): the sums are added there instead and None is returned.  `hold` (deferred sinks, square 32 / 64-wide
    layers): the launch may wait for the q / k / v group of the same attention sublayer and share its launch."""
    global _PENDING
    N, Fi = x2.shape
    Fo = g2.size(1)
    rows = L.query("mivae_linear_wgrad_rows", N)
    if sink is not None and _DEFER is not None and rows <= 512:
        key = (sink[0].data_ptr(), rows, Fo * Fi + Fo)
        if not WGRAD_QUEUE:
            _launch_pending()                      # at most one launch is held back at a time
        if key not in _DEFER:                      # a layer applied twice in one step takes the direct path below
            part = _PART_BUFS.get(key)
            if part is None or part.device != x2.device:
                part = _PART_BUFS[key] = torch.empty(rows, Fo * Fi + Fo, dtype=torch.float32, device=x2.device)
            if WGRAD_QUEUE and (Fo, Fi) in _SINGLE_SHAPES:
                _QUEUE.append((x2, [g2], [part]))
                _DEFER[key] = (part, sink[0], sink[1], Fo * Fi)
                sink_arrived(sink[0], sink[1])
                return None
            if hold and Fi == Fo and Fi in (32, 64):
                _PENDING = (x2, g2, key, part, sink)
                sink_arrived(sink[0], sink[1])
                return None
            L.call("mivae_linear_wgrad", L.ptr(x2), L.ptr(g2), L.ptr(part), N, Fi, Fo, L.stream())
            _DEFER[key] = (part, sink[0], sink[1], Fo * Fi)
            sink_arrived(sink[0], sink[1])
            return None
    part = torch.empty(rows, Fo * Fi + Fo, dtype=torch.float32, device=x2.device)
    L.call("mivae_linear_wgrad", L.ptr(x2), L.ptr(g2), L.ptr(part), N, Fi, Fo, L.stream())
    if sink is not None:
        D = Fo * Fi + Fo
        ws = torch.empty(L.query("mivae_col_workspace_floats", rows, D, 0), dtype=torch.float32, device=x2.device)
        L.call("mivae_colsum_sink", L.ptr(part), L.ptr(sink[0]), L.ptr(sink[1]), Fo * Fi, L.ptr(ws), rows, D, 0, L.stream())
        sink_arrived(sink[0], sink[1])
        return None
    sums = colsum(part)
    return sums[:Fo * Fi].view(Fo, Fi), sums[Fo * Fi:]


def colsum(x: torch.Tensor) -> torch.Tensor:
    """fp32 column sums of a (rows, D) fp32/bf16 matrix, fixed summation order (bias gradient of a Linear)."""
    L.require_gpu(x)
    x = x.contiguous()
    rows, D = x.shape
    code = L.dtype_code(x.dtype)
    out = torch.empty(D, dtype=torch.float32, device=x.device)
    ws = torch.empty(L.query("mivae_col_workspace_floats", rows, D, code), dtype=torch.float32, device=x.device)
    L.call("mivae_colsum", L.ptr(x), L.ptr(out), L.ptr(ws), rows, D, code, L.stream())
    return out


def colsum_into(x: torch.Tensor, dst: torch.Tensor) -> None:
    """dst (D,) fp32 += column sums of x (rows, D) fp32/bf16, fixed order: a gradient written straight into its
    parameter's .grad (no temporary, no accumulate launch)."""
    L.require_gpu(x, dst)
    x = x.contiguous()
    rows, D = x.shape
    if dst.dtype != torch.float32 or not dst.is_contiguous() or dst.numel() != D:
        raise ValueError("colsum_into: destination must be a contiguous fp32 tensor of D elements")
    code = L.dtype_code(x.dtype)
    ws = torch.empty(L.query("mivae_col_workspace_floats", rows, D, code), dtype=torch.float32, device=x.device)
    L.call("mivae_colsum_sink", L.ptr(x), L.ptr(dst), L.ptr(dst), D, L.ptr(ws), rows, D, code, L.stream())


# ----------------------------------------------------------------------------------- pairwise distances (InfoVAE MMD)
class _PairSqDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        B1, F = a.shape
        B2 = b.size(0)
        out = torch.empty(B1, B2, dtype=torch.float32, device=a.device)
        L.call("mivae_pair_sqdist", L.ptr(a), L.ptr(b), L.ptr(out), B1, B2, F, L.dtype_code(a.dtype), L.stream())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B1, F = a.shape
        B2 = b.size(0)
        g = g.contiguous().float()
        da = torch.empty(B1, F, dtype=torch.float32, device=a.device) if ctx.needs_input_grad[0] else None
        db = torch.empty(B2, F, dtype=torch.float32, device=a.device) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            L.call("mivae_pair_sqdist_bwd", L.ptr(a), L.ptr(b), L.ptr(g), L.ptr(da), L.ptr(db), B1, B2, F,
                   L.dtype_code(a.dtype), 0, L.stream())
        return (None if da is None else da.to(a.dtype)), (None if db is None else db.to(b.dtype))


def pair_gauss_logdens(z, mu, sigma, mask, samples: int) -> torch.Tensor:
    """(Bn, Bn) fp32 log-densities of every latent code z[i] under every sentence's posterior N(mu[j / n], sigma[j / n]),
    masked by z[i]'s tokens: the pairwise term of calc_mutual_info (posterior.py:174-222) in one launch, no gradients."""
    L.require_gpu(z, mu, sigma, mask)
    Bn, S, D = _shape3(z)
    dt = z.dtype if z.dtype in (torch.float32, torch.bfloat16) else torch.float32
    zc, mc, sc = (t.detach().to(dt).contiguous() for t in (z, mu, sigma))
    if mc.shape[0] * samples != Bn or mc.shape[1:] != zc.shape[1:] or sc.shape != mc.shape:
        raise ValueError("pair_gauss_logdens: mu / sigma must be (Bn / samples, ...) like z")
    mask = _mask_i64(mask)
    out = torch.empty(Bn, Bn, dtype=torch.float32, device=z.device)
    L.call("mivae_pair_gauss_logdens", L.ptr(zc), L.ptr(mc), L.ptr(sc), L.ptr(mask), L.ptr(out), Bn, int(samples), S, D,
           L.dtype_code(dt), L.stream())
    return out


def pair_sqdist(a, b):
    """(B1, B2) fp32 squared distances between the rows of a (B1, F) and b (B2, F), fp32 or bf16 on the GPU, with
    gradients: the pairwise term of the InfoVAE MMD (nn/kl_loss.py:124-149) without the (B, B, F) tensor."""
    L.require_gpu(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.size(1) != b.size(1) or a.dtype != b.dtype:
        raise ValueError("pair_sqdist: (B1, F) and (B2, F) matrices of one dtype")
    if a.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("pair_sqdist: fp32 or bf16")
    return _PairSqDist.apply(a.contiguous(), b.contiguous())


# ----------------------------------------------------------------------------------- invertible linear flows
class _RowGemm(torch.autograd.Function):
    """Y = X M^T for X (R, hs) and a small square M (hs, hs): mivae_rowgemm forward, the same kernel on M^T for d X,
    mivae_rowgemm_wgrad + fixed-order slab sum for d M."""

    @staticmethod
    def forward(ctx, x2, m):
        R, hs = x2.shape
        mk = m.detach().to(x2.dtype).contiguous()
        y = torch.empty_like(x2)
        L.call("mivae_rowgemm", L.ptr(x2), L.ptr(mk), L.ptr(y), R, hs, L.dtype_code(x2.dtype), L.stream())
        ctx.save_for_backward(x2, mk)
        ctx.mdtype = m.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, mk = ctx.saved_tensors
        R, hs = x2.shape
        dy = dy.contiguous()
        dx = dm = None
        code = L.dtype_code(x2.dtype)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            L.call("mivae_rowgemm", L.ptr(dy), L.ptr(mk.t().contiguous()), L.ptr(dx), R, hs, code, L.stream())
        if ctx.needs_input_grad[1] and R > 0:
            nslab = L.query("mivae_rowgemm_wgrad_slabs", R, hs)
            part = torch.empty(nslab, hs * hs, dtype=torch.float32, device=x2.device)
            L.call("mivae_rowgemm_wgrad", L.ptr(x2), L.ptr(dy), L.ptr(part), R, hs, code, L.stream())
            dm = colsum(part).view(hs, hs).to(ctx.mdtype)
        return dx, dm


ROWGEMM_MAX = 512


def rowgemm(x, m):
    """x (..., hs) times m (hs, hs) transposed, on the matrix cores (a14: invertible_flows.py:29, :39, :61-63, :75-77).
    x fp32 or bf16 on the GPU, hs <= 512; gradients for both."""
    L.require_gpu(x, m)
    hs = m.size(0)
    if m.dim() != 2 or m.size(1) != hs or x.size(-1) != hs or hs > ROWGEMM_MAX:
        raise ValueError(f"rowgemm: x (..., {x.size(-1)}) against m {tuple(m.shape)} (square, <= {ROWGEMM_MAX})")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("rowgemm: fp32 or bf16 latents")
    x2 = x.contiguous().view(-1, hs)
    return _RowGemm.apply(x2, m).view(x.shape)


# ----------------------------------------------------------------------------------- log |det| of the flow matrices
SLOGDET_MAX = 128


class _SlogdetInv(torch.autograd.Function):
    """log |det M_i| of a list of small square fp32 matrices in ONE launch (mivae_slogdet_inv); the same launch leaves
    M_i^-T, which IS the gradient, so the backward is one scaling per matrix (the library route: an LU factorisation +
    ~15 index launches per matrix forward, two triangular solves backward)."""

    @staticmethod
    def forward(ctx, *mats):
        import ctypes
        ms = [m.detach().contiguous() for m in mats]
        dev = ms[0].device
        out = torch.empty(len(ms), dtype=torch.float32, device=dev)
        need = [ctx.needs_input_grad[i] for i in range(len(ms))]
        inv = [torch.empty_like(m) if g else None for m, g in zip(ms, need)]
        sizes = (ctypes.c_int64 * len(ms))(*[m.size(0) for m in ms])
        L.call("mivae_slogdet_inv", _ptr_array(ms), sizes, len(ms), L.ptr(out), _ptr_array(inv), L.stream())
        ctx.inv = inv
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(None if t is None else g[i] * t for i, t in enumerate(ctx.inv))


def slogdet_supported(m) -> bool:
    return bool(m.is_cuda and m.dtype == torch.float32 and m.dim() == 2 and m.size(0) == m.size(1)
                and 1 <= m.size(0) <= SLOGDET_MAX)


def slogdet_batch(mats: Sequence[torch.Tensor]) -> torch.Tensor:
    """(len(mats),) fp32: log |det M_i| with gradients (torch.slogdet(M)[1], invertible_flows.py:30-33), up to 24
    matrices of <= 128 rows per launch."""
    mats = list(mats)
    L.require_gpu(*mats)
    if not all(slogdet_supported(m) for m in mats):
        raise ValueError("slogdet_batch: square fp32 CUDA matrices of 1..128 rows")
    outs = [_SlogdetInv.apply(*mats[i:i + 24]) for i in range(0, len(mats), 24)]
    return outs[0] if len(outs) == 1 else torch.cat(outs)


def slogdet(m: torch.Tensor) -> torch.Tensor:
    """log |det m| (0-d) with gradient: the HIP kernel for fp32 CUDA matrices of <= 128 rows, else torch.linalg.slogdet."""
    if slogdet_supported(m):
        return slogdet_batch([m])[0]
    return torch.linalg.slogdet(m)[1]


# ----------------------------------------------------------------------------------- KL / ELBO
class _KL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logq, logp, free_bits):
        q, p = logq.contiguous().float(), logp.contiguous().float()
        kl = torch.empty_like(q)
        use = free_bits is not None
        L.call("mivae_kl_fwd", L.ptr(q), L.ptr(p), L.ptr(kl), q.numel(), int(use),
               float(free_bits) if use else 0.0, L.stream())
        ctx.save_for_backward(q, p)
        ctx.fb = free_bits
        return kl

    @staticmethod
    def backward(ctx, dkl):
        q, p = ctx.saved_tensors
        dq, dp = torch.empty_like(q), torch.empty_like(p)
        use = ctx.fb is not None
        L.call("mivae_kl_bwd", L.ptr(q), L.ptr(p), L.ptr(dkl.contiguous().float()), L.ptr(dq), L.ptr(dp),
               q.numel(), int(use), float(ctx.fb) if use else 0.0, L.stream())
        return dq, dp, None


def kl_default(logq, logp, free_bits: Optional[float] = None):
    """a16: DefaultKLLoss (nn/kl_loss.py:38-49): clamp(log q - log p, min=free_bits_alpha)."""
    L.require_gpu(logq, logp)
    return _KL.apply(logq, logp, free_bits)


class _Elbo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recon, kl, B, n, wr, wk, iwae, wdev):
        r, k = recon.contiguous().float(), kl.contiguous().float()
        out = torch.empty(3, dtype=torch.float32, device=r.device)
        L.call("mivae_elbo_fwd", L.ptr(r), L.ptr(k), L.ptr(out), B, n, float(wr), float(wk), int(iwae), L.ptr(wdev),
               L.stream())
        ctx.save_for_backward(r, k)
        ctx.wdev = wdev
        ctx.meta = (B, n, float(wr), float(wk), int(iwae))
        return out

    @staticmethod
    def backward(ctx, g):
        r, k = ctx.saved_tensors
        B, n, wr, wk, iwae = ctx.meta
        dr, dk = torch.empty_like(r), torch.empty_like(k)
        L.call("mivae_elbo_bwd", L.ptr(r), L.ptr(k), L.ptr(g.contiguous().float()), L.ptr(dr), L.ptr(dk),
               B, n, wr, wk, iwae, L.ptr(ctx.wdev), L.stream())
        return dr, dk, None, None, None, None, None, None


def elbo_tail(recon, kl, batch: int, samples: int, recon_weight: float, kl_weight: float, iwae: bool,
              weights_dev: Optional[torch.Tensor] = None):
    """a17: (batch-loss, batch-kl, batch-nll) of VAELmModel.forward (models/base/model.py:171-193).
    weights_dev: optional fp32 device tensor {recon weight, kl weight} the kernels read instead of the two floats
    (the step harness keeps the scheduler weights there so that a captured step replays with the current ones)."""
    L.require_gpu(recon, kl)
    if weights_dev is not None:
        L.require_gpu(weights_dev)
        if weights_dev.dtype != torch.float32 or weights_dev.numel() < 2 or not weights_dev.is_contiguous():
            raise ValueError("elbo_tail: weights_dev must be a contiguous fp32 tensor of 2 elements")
    out = _Elbo.apply(recon.reshape(-1), kl.reshape(-1), batch, samples, recon_weight, kl_weight, iwae, weights_dev)
    return out[0], out[1], out[2]


class _LatentTail(torch.autograd.Function):
    """mivae_latent_tail_fwd / _bwd: per-token log q, log p + per-sentence reconstruction error -> batch-loss, batch-kl,
    batch-nll (three device scalars, views of one 3-vector) and (non-differentiable) per-sentence log q, log p, kl.
    The scalars are separate outputs so that the backward receives their cotangents as they are (None when unused):
    indexing one 3-vector output instead costs a zeros + copy launch per step in autograd's select backward."""

    @staticmethod
    def forward(ctx, tok_q, tok_p, recon, B, n, fb, wr, wk, iwae, wdev, rng_state):
        Bn, S = tok_q.shape
        tq, tp, r = tok_q.contiguous(), tok_p.contiguous(), recon.contiguous().float()
        dev = tq.device
        out = torch.empty(3, dtype=torch.float32, device=dev)
        stats = torch.empty(3, Bn, dtype=torch.float32, device=dev)      # log q, log p, kl
        use = fb is not None
        L.call("mivae_latent_tail_fwd", L.ptr(tq), L.ptr(tp), L.ptr(r), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(stats[2]),
               L.ptr(out), B, n, S, int(use), float(fb) if use else 0.0, float(wr), float(wk), int(iwae), L.ptr(wdev),
               L.ptr(rng_state), L.stream())
        ctx.save_for_backward(r, stats)
        ctx.wdev = wdev
        ctx.meta = (B, n, S, fb, float(wr), float(wk), int(iwae))
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2], stats

    @staticmethod
    def backward(ctx, g_loss, g_kl, g_nll, _gstats):
        r, stats = ctx.saved_tensors
        B, n, S, fb, wr, wk, iwae = ctx.meta
        Bn = B * n
        d = torch.empty(3, Bn, dtype=torch.float32, device=r.device)     # d recon, d log q, d log p
        use = fb is not None
        gs = [None if g is None else g.contiguous().float() for g in (g_loss, g_kl, g_nll)]
        L.call("mivae_latent_tail_bwd", L.ptr(r), L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(gs[0]), L.ptr(gs[1]), L.ptr(gs[2]),
               L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]), B, n, int(use), float(fb) if use else 0.0, wr, wk, iwae,
               L.ptr(ctx.wdev), L.stream())
        # per-sentence cotangents of the per-token inputs as expanded views: the producers' kernels index them per sentence
        return (d[1].view(Bn, 1).expand(Bn, S), d[2].view(Bn, 1).expand(Bn, S), d[0], None, None, None, None, None,
                None, None, None)


def latent_tail_supported(batch: int, samples: int, S: int) -> bool:
    return bool(L.query("mivae_latent_tail_supported", int(batch), int(samples), int(S)))


def latent_tail(tok_logq, tok_logp, recon, batch: int, samples: int, free_bits: Optional[float], recon_weight: float,
                kl_weight: float, iwae: bool, weights_dev: Optional[torch.Tensor] = None, rng_advance: bool = False):
    """The reductions of a5 / a15, a16 and a17 in one launch each way: per-token log q and log p ((B n), S) + the
    reconstruction error (B n) -> ((batch-loss, batch-kl, batch-nll), (log q, log p, kl) per sentence).
    posterior.py:157-172, priors/prior.py:148-158, nn/kl_loss.py:38-49, models/base/model.py:171-193.
    rng_advance: also take the draw a posterior_tok(..., rng=True, advance=False) of this forward left pending."""
    L.require_gpu(tok_logq, tok_logp, recon)
    if tok_logq.shape != tok_logp.shape or tok_logq.dim() != 2 or tok_logq.size(0) != batch * samples:
        raise ValueError("latent_tail: per-token log q / log p must both be ((B n), S)")
    if tok_logq.dtype != torch.float32 or tok_logp.dtype != torch.float32:
        raise TypeError("latent_tail: per-token values are fp32")
    state = _philox_peek(tok_logq.device) if rng_advance else None     # the posterior's deferred generator step
    o0, o1, o2, stats = _LatentTail.apply(tok_logq, tok_logp, recon.reshape(-1), int(batch), int(samples), free_bits,
                                          recon_weight, kl_weight, iwae, weights_dev, state)
    return (o0, o1, o2), (stats[0], stats[1], stats[2])


# ----------------------------------------------------------------------------------- reconstruction NLL
class _NLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weights, smoothing):
        Bn, V = logits.size(0), logits.size(-1)
        S = target.numel() // max(Bn, 1)
        lg = logits.contiguous()
        if lg.dtype not in (torch.float32, torch.bfloat16):
            lg = lg.float()
        tgt = target.reshape(Bn, S).contiguous().long()
        w = weights.reshape(Bn, S).contiguous().float()
        dev = lg.device
        per_seq = torch.empty(Bn, dtype=torch.float32, device=dev)
        aux = torch.empty(3, Bn * S, dtype=torch.float32, device=dev)      # lse, tokscale, scratch
        with _timed("nll_fwd"):
            L.call("mivae_nll_fwd", L.ptr(lg), L.ptr(tgt), L.ptr(w), L.ptr(per_seq), L.ptr(aux[0]), L.ptr(aux[1]),
                   L.ptr(aux[2]), Bn, S, V, float(smoothing), L.dtype_code(lg.dtype), L.stream())
        ctx.save_for_backward(lg, tgt, aux)
        ctx.meta = (Bn, S, V, float(smoothing), logits.dtype)
        return per_seq

    @staticmethod
    def backward(ctx, dper):
        lg, tgt, aux = ctx.saved_tensors
        Bn, S, V, smoothing, in_dtype = ctx.meta
        dl = torch.empty_like(lg)
        with _timed("nll_bwd"):
            L.call("mivae_nll_bwd", L.ptr(lg), L.ptr(tgt), L.ptr(aux[0]), L.ptr(aux[1]),
                   L.ptr(dper.contiguous().float()), L.ptr(dl), Bn, S, V, smoothing, L.dtype_code(lg.dtype),
                   L.stream())
        return dl.to(in_dtype), None, None, None


def label_smoothing_nll(logits, target, weights, smoothing: float = 0.0):
    """a18: per-sequence weighted token-mean NLL, LabelSmoothingNLL.forward with size_average=False
    (models/base/losses.py:128-164).  logits (Bn, ..., V); target, weights (Bn, ...)."""
    L.require_gpu(logits, target, weights)
    return _NLL.apply(logits, target, weights, smoothing or 0.0)


class _LinearNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, weight, bias, w_lp, target, weights, smoothing):
        Bn, E = h.size(0), h.size(-1)
        V = weight.size(0)
        S = target.numel() // max(Bn, 1)
        N = Bn * S
        h2 = h.contiguous().view(N, E)
        if h2.dtype != torch.bfloat16:
            h2 = h2.to(torch.bfloat16)
        w16 = w_lp if (w_lp is not None and w_lp.dtype == torch.bfloat16) else weight.detach().to(torch.bfloat16)
        w16 = w16.contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        # fragment-ordered weight + bias image for the two kernels
        wp = torch.empty(L.query("mivae_lce_pack_elems", V, E), dtype=torch.bfloat16, device=h.device)
        wtp = torch.empty(V * E, dtype=torch.bfloat16, device=h.device) if lce_fused_dh(E) and h.requires_grad else None
        L.call("mivae_lce_pack2", L.ptr(w16), L.ptr(b32), L.ptr(wp), L.ptr(wtp), V, E, L.stream())
        tgt = target.reshape(Bn, S).contiguous().long()
        w = weights.reshape(Bn, S).contiguous().float()
        dev = h.device
        per_seq = torch.empty(Bn, dtype=torch.float32, device=dev)
        aux = torch.empty(2, N, dtype=torch.float32, device=dev)            # lse, tokscale
        ws = torch.empty(L.query("mivae_lce_workspace_floats", N, V), dtype=torch.float32, device=dev)
        with _timed("lce_fwd"):
            L.call("mivae_lce_fwd", L.ptr(h2), L.ptr(wp), L.ptr(tgt), L.ptr(w), L.ptr(per_seq), L.ptr(aux[0]),
                   L.ptr(aux[1]), L.ptr(ws), Bn, S, E, V, float(smoothing), L.stream())
        ctx.save_for_backward(h2, w16, wp, tgt, aux)
        ctx.wtp = wtp
        ctx.meta = (Bn, S, E, V, float(smoothing), h.shape, h.dtype, weight.dtype, bias is not None)
        ctx.params = (weight, bias)
        return per_seq

    @staticmethod
    def backward(ctx, dper):
        h2, w16, wp, tgt, aux = ctx.saved_tensors
        Bn, S, E, V, smoothing, hshape, hdtype, wdtype, has_bias = ctx.meta
        N = Bn * S
        dpc = dper.contiguous().float()
        want_dh, want_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        want_db = has_bias and ctx.needs_input_grad[2]
        wpar, bpar = ctx.params
        # d W / d b without a stored (N, V) gradient: mivae_lce_dw recomputes the logits tiles per 64 vocabulary rows and
        # contracts the bf16 d logits with h in registers.  Then the token-major pass only has to produce d h.
        slabs = L.query("mivae_lce_dw_slabs", N, E, V) if (LCE_RECOMPUTE_DW and want_dw and want_db) else 0
        fused_dh = ctx.wtp is not None and want_dh
        need_dl = slabs == 0 and (want_dw or want_db)
        dl = None
        if need_dl or (want_dh and not fused_dh):
            dl = torch.empty(N, V, dtype=torch.bfloat16, device=h2.device)
        dh = dw = db = None
        if fused_dh:
            # d h = dl W on the matrix cores inside the same pass (no (N, V) x (V, E) library GEMM, no re-read of dl)
            dh2 = torch.empty(N, E, dtype=torch.bfloat16, device=h2.device)
            ws = torch.empty(L.query("mivae_lce_dh_workspace_floats", N, E, V), dtype=torch.float32, device=h2.device)
            with _timed("lce_bwd"):
                L.call("mivae_lce_bwd_dh", L.ptr(h2), L.ptr(wp), L.ptr(ctx.wtp), L.ptr(tgt), L.ptr(aux[0]), L.ptr(aux[1]),
                       L.ptr(dpc), L.ptr(dl), L.ptr(ws), L.ptr(dh2), Bn, S, E, V, smoothing, L.stream())
            dh = dh2.view(hshape).to(hdtype)
        elif dl is not None:
            with _timed("lce_bwd"):
                L.call("mivae_lce_bwd", L.ptr(h2), L.ptr(wp), L.ptr(tgt), L.ptr(aux[0]), L.ptr(aux[1]),
                       L.ptr(dpc), L.ptr(dl), Bn, S, E, V, smoothing, L.stream())
            if want_dh:
                dh = dl.matmul(w16).view(hshape).to(hdtype)
        if slabs:
            part = torch.empty(slabs, V, E + 8, dtype=torch.float32, device=h2.device)
            with _timed("lce_dw"):
                L.call("mivae_lce_dw", L.ptr(h2), L.ptr(wp), L.ptr(tgt), L.ptr(aux[0]), L.ptr(aux[1]), L.ptr(dpc),
                       L.ptr(part), Bn, S, E, V, smoothing, L.stream())
            if grad_sink_ok(wpar, bpar) and wpar.grad.shape == (V, E):
                L.call("mivae_fold_aug_sink", L.ptr(part), slabs, V, E, 8, L.ptr(wpar.grad), L.ptr(bpar.grad), L.stream())
                sink_arrived(wpar.grad, bpar.grad)
                return dh, None, None, None, None, None, None
            # returned gradients: the same fixed-order fold as the sink route, into zeroed temporaries (bit-equal to it)
            dwt = torch.zeros(V * E + V, dtype=torch.float32, device=h2.device)
            dw_t, db_t = dwt[:V * E].view(V, E), dwt[V * E:]
            L.call("mivae_fold_aug_sink", L.ptr(part), slabs, V, E, 8, L.ptr(dw_t), L.ptr(db_t), L.stream())
            return dh, dw_t.to(wdtype), db_t.to(wdtype), None, None, None, None
        n = h2.size(0)
        if (want_dw and want_db and n % 8 == 0 and n // 8 >= 1024 and grad_sink_ok(wpar, bpar)
                and wpar.grad.shape == (V, E)):
            # gradient sinks: [h | 1] lives in a persistent buffer whose ones block is written once, the 8 token-slab
            # partials of dl^T [h | 1] are folded straight into weight.grad / bias.grad (one launch, no accumulations)
            key = ("lce_haug", n, E, h2.device)
            h_aug = _PART_BUFS.get(key)
            if h_aug is None:
                h_aug = _PART_BUFS[key] = torch.ones(n, E + 8, dtype=torch.bfloat16, device=h2.device)
            h_aug[:, :E].copy_(h2)
            try:
                part = torch.bmm(dl.view(8, n // 8, V).transpose(1, 2), h_aug.view(8, n // 8, E + 8),
                                 out_dtype=torch.float32)
            except (TypeError, RuntimeError):
                part = None
            if part is not None:
                L.call("mivae_fold_aug_sink", L.ptr(part), 8, V, E, 8, L.ptr(wpar.grad), L.ptr(bpar.grad), L.stream())
                sink_arrived(wpar.grad, bpar.grad)
                return dh, None, None, None, None, None, None
        if want_dw and want_db:
            # one dense product for both: d W = dl^T h and d b = dl^T 1 (a ones block appended to h)
            h_aug = torch.nn.functional.pad(h2, (0, 8), value=1.0)          # one launch (was ones + cat)
            G = 8 if (n % 8 == 0 and n // 8 >= 1024) else (4 if (n % 4 == 0 and n // 4 >= 1024) else 1)
            dwa = None
            if G > 1:
                # the token axis is the contraction: one (V, E + 8) output tile per workgroup walks all N tokens in the
                # library's single GEMM (102 us at C2); split into G batches with fp32 partials, folded in fixed order
                try:
                    part = torch.bmm(dl.view(G, n // G, V).transpose(1, 2), h_aug.view(G, n // G, E + 8),
                                     out_dtype=torch.float32)
                    dwa = colsum(part.view(G, -1)).view(V, E + 8)
                except (TypeError, RuntimeError):
                    dwa = None
            if dwa is None:
                dwa = dl.t().matmul(h_aug)
            dw, db = dwa[:, :E].to(wdtype), dwa[:, E].to(wdtype)
        elif want_dw:
            dw = dl.t().matmul(h2).to(wdtype)
        elif want_db:
            db = dl.sum(0, dtype=torch.float32).to(wdtype)
        return dh, dw, db, None, None, None, None


LCE_FEATURES = (32, 64, 128, 256)
LCE_FUSED_DH = True        # d h inside the lce backward pass (mivae_lce_bwd_dh); False: library GEMM on d logits
LCE_FUSED_DH_MAX_E = 128   # at E = 256 the d h accumulators (128 registers) leave one wave per SIMD: the pass that only stores
                           # d logits (232 us at C3) + the library product beat the fused pass (526 us); C3 step 4.95 -> 4.88 ms


def lce_fused_dh(E: int) -> bool:
    return bool(LCE_FUSED_DH and E <= LCE_FUSED_DH_MAX_E)
LCE_RECOMPUTE_DW = os.environ.get("MIVAE_LCE_DW", "1") != "0"   # d W / d b by mivae_lce_dw (E <= 64): no stored d logits


def linear_nll_supported(h, weight) -> bool:
    return bool(h.is_cuda and h.size(-1) in LCE_FEATURES and weight.size(0) % 64 == 0 and weight.size(0) >= 64
                and weight.size(1) == h.size(-1))


def linear_nll(h, weight, bias, target, weights, smoothing: float = 0.0, w_lp=None):
    """8f-1: LabelSmoothingNLL(size_average=False) of logits = h W^T + b without materialising the logits
    (nonauto_lm/model.py:105-112 + models/base/losses.py:128-164).  bf16 operands, fp32 accumulation.
    h (Bn, ..., E); weight (V, E) fp32 parameter (`w_lp`: its bf16 shadow, if one is kept); bias (V) or None."""
    L.require_gpu(h, weight, target, weights)
    return _LinearNLL.apply(h, weight, bias, w_lp, target, weights, smoothing or 0.0)


# ----------------------------------------------------------------------------------- fused NICE stack
_PTR_CACHE = {}


def _ptr_table(params: Sequence[torch.Tensor], device) -> torch.Tensor:
    key = (device, tuple(p.data_ptr() for p in params))
    tab = _PTR_CACHE.get(key)
    if tab is None:
        if len(_PTR_CACHE) > 64 and _PIN == 0:
            _PTR_CACHE.clear()
        tab = torch.tensor(key[1], dtype=torch.int64).to(device)
        _PTR_CACHE[key] = tab
    return tab


def nice_pack(params: Sequence[torch.Tensor], K: int, h: int, H: int, dtype, inverse: bool = False) -> torch.Tensor:
    """Repack K x {W1s,b1s,W2s,b2s,W1t,b1t,W2t,b2t} (fp32 nn.Linear layout) into MFMA fragment order."""
    dev = params[0].device
    for p in params:
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise TypeError("NICE parameters must be contiguous float32 tensors")
    code = L.dtype_code(dtype)
    nbytes = L.query("mivae_nice_pack_bytes", h, H, code)
    packed = torch.empty(K * nbytes, dtype=torch.uint8, device=dev)
    L.call("mivae_nice_pack", L.ptr(_ptr_table(params, dev)), L.ptr(packed), K, h, H, code, int(inverse), L.stream())
    return packed


class _NiceStack(torch.autograd.Function):
    """K fused NICE-FF steps (+ optional base log-prob).  params: K x 8 tensors, see nice_pack."""

    @staticmethod
    def forward(ctx, z, mask, H, act, inverse, base, mu, std, *params):
        Bn, S, D = _shape3(z)
        h = D // 2
        K = len(params) // 8
        z = z.contiguous()
        code = L.dtype_code(z.dtype)
        packed = nice_pack(params, K, h, H, z.dtype, inverse)
        need_grad = (not inverse) and any(ctx.needs_input_grad)
        N = Bn * S
        z_out = torch.empty_like(z)
        hist = torch.empty(K, N, h, dtype=z.dtype, device=z.device) if need_grad else None
        tok = torch.empty(Bn, S, dtype=torch.float32, device=z.device)
        with _timed("nice_stack_inv" if inverse else "nice_stack_fwd"):
            L.call("mivae_nice_stack_fwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask), L.ptr(packed),
                   L.ptr(tok), Bn, 1, S, D, H, K, act, int(inverse), int(base), float(mu), float(std), code,
                   L.stream())
        ctx.meta = (Bn, S, D, H, K, act, int(base), float(mu), float(std), code)
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.params = params
        ctx.set_materialize_grads(False)     # an unused z_K (the ELBO only takes the per-token values) comes back as None
        if need_grad:
            ctx.save_for_backward(z, z_out, hist, mask, packed)
        return z_out, tok

    @staticmethod
    def backward(ctx, dz_out, dtok):
        z, z_out, hist, mask, packed = ctx.saved_tensors
        Bn, S, D, H, K, act, base, mu, std, code = ctx.meta
        h, N = D // 2, Bn * S
        hp, Hp = (h + 31) // 32 * 32, (H + 31) // 32 * 32
        npad = L.query("mivae_nice_stack_npad", N)
        dt = z.dtype
        Gt = torch.empty(K, 2, Hp, npad, dtype=dt, device=z.device)
        Ht = torch.empty(K, 2, Hp, npad, dtype=dt, device=z.device)
        dSt = torch.empty(K, 2, hp, npad, dtype=dt, device=z.device)
        x0t = torch.empty(hp, npad, dtype=dt, device=z.device)
        dz_in = torch.empty_like(z)
        # `dlogp` is one cotangent per sentence (the expanded view _SegmentSum / the ELBO tail hand back), or -- a genuine
        # per-token cotangent -- one per token with S = 1 segments
        dlog, per_tok = _tok_cotangent(dtok, Bn, S)
        seg = (N, 1, 1) if per_tok else (Bn, 1, S)
        dzo = None if dz_out is None else dz_out.contiguous()
        with _timed("nice_stack_bwd"):
            L.call("mivae_nice_stack_bwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask), L.ptr(packed),
                   L.ptr(dzo), L.ptr(dlog), L.ptr(dz_in), L.ptr(Gt), L.ptr(Ht), L.ptr(dSt), L.ptr(x0t), *seg,
                   D, H, K, act, base, mu, std, code, L.stream())
        # Weight gradients: split-K NT MFMA products over the token axis on the emitted
        # [feature][token] tensors, reduced in fixed order (k_dw_partial / k_dw_reduce).
        ws = torch.empty(L.query("mivae_nice_dw_workspace_floats", N, h, H, K), dtype=torch.float32, device=z.device)
        if grad_sink_ok(*ctx.params) and all(ctx.needs_input_grad[8:]):
            table = _ptr_table([p.grad for p in ctx.params], z.device)
            with _timed("nice_dw"):
                L.call("mivae_nice_dw_sink", L.ptr(Gt), L.ptr(Ht), L.ptr(dSt), L.ptr(x0t), L.ptr(ws), L.ptr(table),
                       N, h, H, K, code, L.stream())
            sink_arrived(*[p.grad for p in ctx.params])
            return (dz_in, None, None, None, None, None, None, None, *([None] * len(ctx.params)))
        dW1 = torch.empty(K, 2, H, h, dtype=torch.float32, device=z.device)
        db1 = torch.empty(K, 2, H, dtype=torch.float32, device=z.device)
        dW2 = torch.empty(K, 2, h, H, dtype=torch.float32, device=z.device)
        db2 = torch.empty(K, 2, h, dtype=torch.float32, device=z.device)
        with _timed("nice_dw"):
            L.call("mivae_nice_dw", L.ptr(Gt), L.ptr(Ht), L.ptr(dSt), L.ptr(x0t), L.ptr(ws), L.ptr(dW1), L.ptr(db1),
                   L.ptr(dW2), L.ptr(db2), N, h, H, K, code, L.stream())
        grads = []
        for k in range(K):
            for net in range(2):
                grads += [dW1[k, net], db1[k, net], dW2[k, net], db2[k, net]]
        return (dz_in, None, None, None, None, None, None, None, *grads)


BIG_STREAMS = {"fwd": 0, "inv": 1, "chain": 2, "nets": 3, "bias": 4}


def nice_big_supported(z, H: int) -> bool:
    """bf16 latents with 96 < D/2 <= 256 run the MFMA-bound kernels of csrc/nice_big.hip (C3, C5)."""
    return bool(z.is_cuda and z.dtype == torch.bfloat16
                and L.query("mivae_nice_big_supported", z.size(-1) // 2, int(H), L.dtype_code(z.dtype)))


def nice_big_pack(params: Sequence[torch.Tensor], K: int, h: int, H: int, kind: str) -> torch.Tensor:
    dev = params[0].device
    for p in params:
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise TypeError("NICE parameters must be contiguous float32 tensors")
    code = BIG_STREAMS[kind]
    out = torch.empty(L.query("mivae_nice_big_pack_bytes", K, h, H, code), dtype=torch.uint8, device=dev)
    L.call("mivae_nice_big_pack", L.ptr(_ptr_table(params, dev)), L.ptr(out), K, h, H, code, L.stream())
    return out


def nice_big_pack_multi(params: Sequence[torch.Tensor], K: int, h: int, H: int, kinds: Sequence[str]):
    """Several packed streams of the same weights in ONE launch: {kind: uint8 tensor}."""
    dev = params[0].device
    for p in params:
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise TypeError("NICE parameters must be contiguous float32 tensors")
    outs = {k: torch.empty(L.query("mivae_nice_big_pack_bytes", K, h, H, BIG_STREAMS[k]), dtype=torch.uint8, device=dev)
            for k in kinds}
    by_code = {BIG_STREAMS[k]: t for k, t in outs.items()}
    L.call("mivae_nice_big_pack_multi", L.ptr(_ptr_table(params, dev)), _ptr_array([by_code.get(c) for c in range(5)]),
           K, h, H, L.stream())
    return outs


# Weight-gradient route of the wide-latent backward.  True: the nets launch keeps the hidden activations / d pre in HBM
# (2 K 2 N Hp elements) and the weight gradients contract them directly (k_big_dw2); False: they are recomputed from x0
# and d s / d t (k_big_dw, no extra memory); "auto": save when D/2 > 128 -- there the recompute kernel needs two passes
# (register budget) and saving wins (C5: nets + dW 4.8 ms vs 6.2 ms), while at D/2 <= 128 the extra 2 x K 2 N Hp HBM
# round trip costs more than the recompute it removes (C3: 952 us vs 886 us).
NICE_BIG_SAVE_HIDDEN = {"0": False, "1": True}.get(os.environ.get("MIVAE_BIG_SAVE", ""), "auto")   # env: A/B runs


def _nice_big_save_hidden(h: int) -> bool:
    return h > 128 if NICE_BIG_SAVE_HIDDEN == "auto" else bool(NICE_BIG_SAVE_HIDDEN)


class _NiceStackBig(torch.autograd.Function):
    """K fused NICE-FF steps (+ optional base log-prob) on the wide-latent kernels.  params as _NiceStack."""

    @staticmethod
    def forward(ctx, z, mask, H, act, inverse, base, mu, std, *params):
        Bn, S, D = _shape3(z)
        h = D // 2
        K = len(params) // 8
        z = z.contiguous()
        N = Bn * S
        need_grad = (not inverse) and any(ctx.needs_input_grad)
        # every stream this call (and its backward) reads, packed in one launch
        kinds = ["inv" if inverse else "fwd", "bias"] + (["chain", "nets"] if need_grad else [])
        packed = nice_big_pack_multi(params, K, h, H, kinds)
        pk, pb = packed[kinds[0]], packed["bias"]
        z_out = torch.empty_like(z)
        hist = torch.empty(K, N, h, dtype=z.dtype, device=z.device) if need_grad else None
        tok = torch.empty(Bn, S, dtype=torch.float32, device=z.device)
        with _timed("nice_stack_inv" if inverse else "nice_stack_fwd"):
            L.call("mivae_nice_big_fwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask), L.ptr(pk), L.ptr(pb),
                   L.ptr(tok), Bn, 1, S, D, H, K, act, int(inverse), int(base), float(mu), float(std), L.stream())
        ctx.meta = (Bn, S, D, H, K, act, int(base), float(mu), float(std))
        ctx.params = params
        ctx.set_materialize_grads(False)
        if need_grad:
            ctx.save_for_backward(z, z_out, hist, mask, pb, packed["chain"], packed["nets"])
        return z_out, tok

    @staticmethod
    def backward(ctx, dz_out, dtok):
        z, z_out, hist, mask, pb, pc, pn = ctx.saved_tensors
        Bn, S, D, H, K, act, base, mu, std = ctx.meta
        h, N = D // 2, Bn * S
        dev = z.device
        params = ctx.params
        dz_in = torch.empty_like(z)
        dsdt = torch.empty(K, 2, N, h, dtype=z.dtype, device=dev)
        ws = torch.empty(L.query("mivae_nice_big_dw_workspace_floats", N, h, H, K), dtype=torch.float32, device=dev)
        # hidden activations + d pre of every (step, net, token), saved by the nets launch for the weight gradients
        hid = torch.empty(L.query("mivae_nice_big_hidden_elems", N, h, H, K), dtype=z.dtype, device=dev) \
            if _nice_big_save_hidden(h) else None
        dlog, per_tok = _tok_cotangent(dtok, Bn, S)
        seg = (N, 1, 1) if per_tok else (Bn, 1, S)       # per-token cotangents: one "sentence" per token
        dzo = None if dz_out is None else dz_out.contiguous()
        sink = grad_sink_ok(*params) and all(ctx.needs_input_grad[8:])
        if sink:
            table = _ptr_table([p.grad for p in params], dev)
            dW1 = db1 = dW2 = db2 = None
        else:
            table = None
            dW1 = torch.empty(K, 2, H, h, dtype=torch.float32, device=dev)
            db1 = torch.empty(K, 2, H, dtype=torch.float32, device=dev)
            dW2 = torch.empty(K, 2, h, H, dtype=torch.float32, device=dev)
            db2 = torch.empty(K, 2, h, dtype=torch.float32, device=dev)
        with _timed("nice_stack_bwd"):
            L.call("mivae_nice_big_bwd", L.ptr(z), L.ptr(z_out), L.ptr(hist), L.ptr(mask), L.ptr(_ptr_table(params, dev)),
                   L.ptr(pc), L.ptr(pn), L.ptr(pb),
                   L.ptr(dzo), L.ptr(dlog), L.ptr(dz_in), L.ptr(dsdt), L.ptr(hid), L.ptr(ws), L.ptr(dW1), L.ptr(db1), L.ptr(dW2),
                   L.ptr(db2), L.ptr(table), *seg, D, H, K, act, base, mu, std, L.stream())
        if sink:
            sink_arrived(*[p.grad for p in params])
            return (dz_in, None, None, None, None, None, None, None, *([None] * len(params)))
        grads = []
        for k in range(K):
            for net in range(2):
                grads += [dW1[k, net], db1[k, net], dW2[k, net], db2[k, net]]
        return (dz_in, None, None, None, None, None, None, None, *grads)


def nice_stack(z, mask, params: Sequence[torch.Tensor], H: int, activation: str, inverse: bool = False,
               base: bool = False, mu: float = 0.0, std: float = 1.0):
    """Fused K-step NICE-FF stack (a6-a10): returns (z_K, val) where val is the masked log-det summed
    over steps (+ the base log-prob when `base`), reduced like the reference: (B,) for masked 3-D and
    2-D inputs, (B, S) for unmasked 3-D inputs."""
    L.require_gpu(z, mask, *params)
    mask = _mask_i64(mask)
    Bn, S, D = _shape3(z)
    z_out, tok = nice_stack_tok(z, mask, params, H, activation, inverse, base, mu, std)
    if z.dim() == 2:
        return z_out, tok.view(Bn)
    if mask is None and not base:
        return z_out, tok
    return z_out, _SegmentSum.apply(tok, Bn, S)


def nice_stack_tok(z, mask, params: Sequence[torch.Tensor], H: int, activation: str, inverse: bool = False,
                   base: bool = False, mu: float = 0.0, std: float = 1.0):
    """nice_stack without the per-sentence reduction: (z_K, per-token value (Bn, S) fp32)."""
    L.require_gpu(z, mask, *params)
    mask = _mask_i64(mask)
    fn = _NiceStackBig if nice_big_supported(z, H) else _NiceStack
    return fn.apply(z, mask, int(H), ACT_IDS[activation], bool(inverse), bool(base), float(mu), float(std), *params)


# ----------------------------------------------------------------------------------- mixed flow stack (no grad)
MIXED_NICE, MIXED_ACTNORM, MIXED_LINEAR = 0, 1, 2


def mixed_stack_supported(D: int, Hmax: int) -> bool:
    return bool(L.query("mivae_mixed_stack_supported", int(D), int(Hmax)))


class MixedPlan:
    """Host step table of mivae_mixed_stack for one flow list (built once, reusable while the parameters' storage is
    unchanged: the table holds device pointers)."""

    def __init__(self, steps):
        import ctypes
        table, self.keep = [], []
        for st in steps:
            kind = st[0]
            if kind == MIXED_NICE:
                row = [kind, int(st[1]), ACT_IDS[st[2]], 0]
                ps = list(st[3])
                assert len(ps) == 8
            elif kind == MIXED_ACTNORM:
                row, ps = [kind, 0, 0, 0], [st[1], st[2]]
            elif kind == MIXED_LINEAR:
                row, ps = [kind, 0, 0, int(st[1].size(-1))], [st[1]]
                assert st[1].dim() == 2 and st[1].size(0) == st[1].size(1)
            else:
                raise ValueError(f"mixed_stack: unknown step kind {kind}")
            for p in ps:
                L.require_gpu(p)
                if p.dtype != torch.float32:
                    raise TypeError("mixed_stack: parameters must be fp32")
            ps = [p.detach().contiguous() for p in ps]
            self.keep += ps
            table += row + [p.data_ptr() for p in ps] + [0] * (8 - len(ps))
        self.K = len(steps)
        self.host = (ctypes.c_int64 * len(table))(*table)
        self.addr = ctypes.addressof(self.host)


def mixed_stack(z, mask, steps, inverse: bool = False):
    """Any list of NICE-FF / ActNorm / invertible-linear steps in ONE launch (csrc/mixed_stack.hip), without
    gradients: the evaluation / sampling form of FlowPrior.forward_pass / backward_pass (flow_prior.py:37-70) for
    lists that are not coupling-only.  `steps` (or a MixedPlan made from them) in execution order, each one of
      (MIXED_NICE, H, activation name, [W1s, b1s, W2s, b2s, W1t, b1t, W2t, b2t]),
      (MIXED_ACTNORM, log_scale, bias),  (MIXED_LINEAR, matrix (hs, hs))      -- fp32 CUDA tensors.
    Returns (z_out, val): val = the masked sum of the coupling log-scales (negated for `inverse`), reduced like the
    reference ((B,) for masked 3-D and 2-D inputs, (B, S) for unmasked 3-D inputs).  The parameter-only
    log-determinants are the caller's to add."""
    L.require_gpu(z, mask)
    mask = _mask_i64(mask)
    Bn, S, D = _shape3(z)
    plan = steps if isinstance(steps, MixedPlan) else MixedPlan(steps)
    zc = z.detach().contiguous()
    z_out = torch.empty_like(zc)
    tok = torch.empty(Bn * S, dtype=torch.float32, device=z.device)
    mask_div = 1 if mask is None else Bn // mask.size(0)
    L.call("mivae_mixed_stack", L.ptr(zc), L.ptr(z_out), L.ptr(mask), L.ptr(tok), plan.addr, plan.K,
           Bn, S, mask_div, D, int(bool(inverse)), L.dtype_code(z.dtype), L.stream())
    if z.dim() == 2:
        return z_out, tok
    if mask is None:
        return z_out, tok.view(Bn, S)
    return z_out, segment_sum(tok, Bn, S)


# ----------------------------------------------------------------------------------- mixed flow stack, training path
class _MixedStackTrain(torch.autograd.Function):
    """A mixed NICE-FF / ActNorm / invertible-linear list in the density direction under autograd: ONE launch forward
    (mivae_mixed_stack_train_fwd, which keeps the input of every step) and ONE launch for the whole z-gradient chain
    (mivae_mixed_stack_bwd); the parameter gradients are reductions over all tokens and follow as a fixed handful of
    launches whatever the number of steps: one fold for every ActNorm step together, the split-K pair of
    csrc/nice_stack.hip for every coupling step together, and one row-slab product + fold per linear step.
    `meta` = [(kind, H, activation name, head size, number of parameters)] in list order, `params` flat."""

    @staticmethod
    def forward(ctx, z, mask, meta, *params):
        Bn, S, D = _shape3(z)
        N = Bn * S
        z = z.contiguous()
        steps, at = [], 0
        for kind, H, act, hs, cnt in meta:
            ps = list(params[at:at + cnt])
            at += cnt
            steps.append((kind, H, act, ps) if kind == MIXED_NICE else (kind, *ps))
        plan = MixedPlan(steps)
        z_out = torch.empty_like(z)
        tok = torch.empty(Bn, S, dtype=torch.float32, device=z.device)
        hist = torch.empty(plan.K, N, D, dtype=torch.float32, device=z.device)
        L.call("mivae_mixed_stack_train_fwd", L.ptr(z), L.ptr(z_out), L.ptr(mask), L.ptr(tok), L.ptr(hist), plan.addr,
               plan.K, Bn, S, 1, D, L.dtype_code(z.dtype), L.stream())
        ctx.plan, ctx.meta, ctx.shape = plan, meta, (Bn, S, D, z.dtype)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(hist, mask)
        return z_out, tok

    @staticmethod
    def backward(ctx, dz_out, dtok):
        hist, mask = ctx.saved_tensors
        plan, meta = ctx.plan, ctx.meta
        Bn, S, D, dtype = ctx.shape
        N, h, dev = Bn * S, D // 2, hist.device
        nice = [i for i, m in enumerate(meta) if m[0] == MIXED_NICE]
        acts = [i for i, m in enumerate(meta) if m[0] == MIXED_ACTNORM]
        lins = [i for i, m in enumerate(meta) if m[0] == MIXED_LINEAR]
        H = meta[nice[0]][1] if nice else 0
        hp, Hp = (h + 31) // 32 * 32, (H + 31) // 32 * 32
        npad = L.query("mivae_nice_stack_npad", N)
        blocks = L.query("mivae_mixed_stack_bwd_blocks", N, D, H, H, int(bool(lins)))
        f32 = dict(dtype=torch.float32, device=dev)
        an_part = torch.empty(blocks, len(acts) * 2 * D, **f32) if acts else None
        gcot = torch.empty(len(lins), N, D, **f32) if lins else None
        Gt = torch.empty(len(nice), 2, Hp, npad, **f32) if nice else None
        Ht = torch.empty(len(nice), 2, Hp, npad, **f32) if nice else None
        dSt = torch.empty(len(nice), 2, hp, npad, **f32) if nice else None
        x0t = torch.empty(len(nice), hp, npad, **f32) if nice else None
        dz_in = torch.empty(Bn, S, D, dtype=dtype, device=dev)
        dlog, per_tok = (None, False) if dtok is None else _tok_cotangent(dtok, Bn, S)
        dzo = None if dz_out is None else dz_out.contiguous()
        L.call("mivae_mixed_stack_bwd", L.ptr(hist), L.ptr(dzo), L.ptr(dlog), int(per_tok), L.ptr(mask), plan.addr, plan.K,
               L.ptr(dz_in), L.ptr(gcot), L.ptr(an_part), L.ptr(Gt), L.ptr(Ht), L.ptr(dSt), L.ptr(x0t), Bn, S, 1, D,
               L.dtype_code(dtype), L.stream())
        grads = [None] * sum(m[4] for m in meta)
        first = [0]
        for m in meta:
            first.append(first[-1] + m[4])
        if acts:
            folded = colsum(an_part).view(len(acts), 2, D)
            for j, i in enumerate(acts):
                grads[first[i]], grads[first[i] + 1] = folded[j, 0], folded[j, 1]
        for j, i in enumerate(lins):
            hs = meta[i][3]
            R = N * (D // hs)
            nslab = L.query("mivae_rowgemm_wgrad_slabs", R, hs)
            part = torch.empty(nslab, hs * hs, **f32)
            L.call("mivae_rowgemm_wgrad", L.ptr(hist[i]), L.ptr(gcot[j]), L.ptr(part), R, hs, L.dtype_code(torch.float32),
                   L.stream())
            grads[first[i]] = colsum(part).view(hs, hs)
        if nice:
            K = len(nice)
            ws = torch.empty(L.query("mivae_nice_dw_workspace_floats", N, h, H, K), **f32)
            dW1, db1 = torch.empty(K, 2, H, h, **f32), torch.empty(K, 2, H, **f32)
            dW2, db2 = torch.empty(K, 2, h, H, **f32), torch.empty(K, 2, h, **f32)
            L.call("mivae_nice_dw_steps", L.ptr(Gt), L.ptr(Ht), L.ptr(dSt), L.ptr(x0t), L.ptr(ws), L.ptr(dW1), L.ptr(db1),
                   L.ptr(dW2), L.ptr(db2), N, h, H, K, L.dtype_code(torch.float32), L.stream())
            for j, i in enumerate(nice):
                for net in range(2):
                    grads[first[i] + 4 * net: first[i] + 4 * net + 4] = [dW1[j, net], db1[j, net], dW2[j, net], db2[j, net]]
        return (dz_in if ctx.needs_input_grad[0] else None, None, None, *grads)


def mixed_stack_train_supported(D: int, hidden: Sequence[int], any_linear: bool) -> bool:
    """True when the list fits mivae_mixed_stack_bwd: D even <= 128, one coupling hidden width <= 128."""
    hmax, hmin = (max(hidden), min(hidden)) if hidden else (0, 0)
    return mixed_stack_supported(D, hmax) and L.query("mivae_mixed_stack_bwd_blocks", 1, int(D), hmax, hmin,
                                                      int(any_linear)) > 0


def mixed_stack_train(z, mask, steps):
    """mixed_stack() in the density direction WITH gradients for z and every step parameter (steps as for mixed_stack,
    holding the live fp32 parameters).  Returns (z_out, val) reduced like mixed_stack."""
    L.require_gpu(z, mask)
    mask = _mask_i64(mask)
    Bn, S, D = _shape3(z)
    meta, flat = [], []
    for st in steps:
        if st[0] == MIXED_NICE:
            meta.append((MIXED_NICE, int(st[1]), st[2], 0, 8))
            flat += list(st[3])
        elif st[0] == MIXED_ACTNORM:
            meta.append((MIXED_ACTNORM, 0, None, 0, 2))
            flat += [st[1], st[2]]
        else:
            meta.append((MIXED_LINEAR, 0, None, int(st[1].size(-1)), 1))
            flat.append(st[1])
    z3 = z if z.dim() == 3 else z.unsqueeze(1)
    z_out, tok = _MixedStackTrain.apply(z3, mask, tuple(meta), *flat)
    if z.dim() == 2:
        return z_out.view(Bn, D), tok.view(Bn)
    if mask is None:
        return z_out, tok
    return z_out, _SegmentSum.apply(tok, Bn, S)
