"""Round-2 kernels of the transformer bracket and of the step's gradient plumbing, through the C ABI:
  * libmivae attention kernels (csrc/attention.hip) against the reference's attention core restated in fp32 torch:
    MultiHeadAttention.forward, nn/transformer.py:62-117 (scores / sqrt(d), masked_fill(-1e13), softmax, probs @ v).
    bf16 operands: outputs within 2e-2 absolute / relative of the fp32 restatement on the same bf16-rounded inputs
    (the probabilities are rounded to bf16 before the second product, like every fused attention);
  * residual epilogue of the output projection, fused wide q|k|v projection, paired / queued weight gradients,
    the bf16 twin of the residual-stream gradient, the vocabulary projection's gradient sink: each against the
    unfused form it replaces (bit-equal where the arithmetic is the same, stated tolerance otherwise)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, mask, heads, causal):
    B, S, W = q.shape
    hd = W // heads
    sp = lambda x: x.view(B, S, heads, hd).transpose(1, 2)
    qh, kh, vh = sp(q), sp(k), sp(v)
    scores = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(hd)
    keep = torch.ones(B, 1, 1, S, dtype=torch.bool, device=q.device) if mask is None else mask.ne(0)[:, None, None, :]
    if causal:
        keep = keep & torch.ones(S, S, dtype=torch.bool, device=q.device).tril()[None, None]
    scores = scores.masked_fill(~keep, -1e13)
    return torch.matmul(torch.softmax(scores, dim=-1), vh).transpose(1, 2).reshape(B, S, W)


def _case(B, S, heads, hd, causal, masked, seed=0, packed=False, empty_row=False):
    from nonauto_lm_amd import ops
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = heads * hd
    x = torch.randn(B, S, 3 * W, device="cuda", generator=g).to(torch.bfloat16)
    mask = None
    if masked:
        lens = torch.randint(1, S + 1, (B,), device="cuda", generator=g)
        lens[0] = S
        mask = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).long()
        if empty_row:
            mask[-1] = 0          # a sentence without tokens: the reference attends uniformly (finite fill)
    dout = torch.randn(B, S, W, device="cuda", generator=g).to(torch.bfloat16)
    split = lambda t: (t[..., :W], t[..., W:2 * W], t[..., 2 * W:])
    if packed:        # q, k, v stay column blocks of one (B, S, 3 W) tensor (row stride 3 W)
        leaves = [x.clone().requires_grad_(True)]
        out = ops.attention(*split(leaves[0]), mask, heads, causal)
        refl = [x.float().requires_grad_(True)]
        ref = _ref(*split(refl[0]), mask, heads, causal)
    else:
        leaves = [t.clone().requires_grad_(True) for t in split(x)]
        out = ops.attention(*leaves, mask, heads, causal)
        refl = [t.float().clone().requires_grad_(True) for t in split(x)]
        ref = _ref(*refl, mask, heads, causal)
    out.backward(dout)
    ref.backward(dout.float())
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)
    for i, (a, b) in enumerate(zip(leaves, refl)):
        scale = b.grad.abs().max().item()
        err = (a.grad.float() - b.grad).abs().max().item()
        assert err <= 2e-2 * max(scale, 1.0), f"grad {i}: max err {err} (scale {scale})"


@pytest.mark.parametrize("hd", [16, 32, 64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_attention_matches_reference_core(hd, causal):
    _case(3, 64, 2, hd, causal, masked=True, seed=hd + causal)


@pytest.mark.parametrize("S", [1, 7, 33, 100, 128, 257])
def test_attention_ragged_lengths(S):
    _case(2, S, 4, 16, False, masked=True, seed=S)
    _case(2, S, 2, 64, True, masked=True, seed=S + 1)


def test_attention_no_mask_and_packed_rows():
    _case(2, 96, 4, 32, False, masked=False, seed=5)
    _case(2, 96, 4, 32, False, masked=True, seed=6, packed=True)


def test_attention_row_without_valid_key_is_uniform():
    """mask == 0 everywhere in a sentence: -1e13 everywhere -> uniform weights over all S keys (forward), that weight
    into dV and nothing into dQ / dK (masked_fill's gradient), exactly like the reference expression"""
    _case(3, 40, 2, 16, False, masked=True, seed=9, empty_row=True)
    _case(3, 70, 2, 64, False, masked=True, seed=10, empty_row=True)


def test_attention_config_shapes():
    _case(4, 64, 4, 16, False, masked=True, seed=11)      # BASELINE config 2
    _case(4, 128, 4, 64, False, masked=True, seed=12)     # config 3
    _case(2, 512, 4, 128, False, masked=True, seed=13)    # config 5


def test_attention_rejects_unsupported():
    from nonauto_lm_amd import ops
    x = torch.zeros(2, 8, 48, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.attention(x, x, x, None, 2)                   # head size 24
    with pytest.raises(ValueError):
        ops.attention(x.float(), x.float(), x.float(), None, 3)


@pytest.mark.parametrize("causal", [False, True])
def test_multi_head_attention_module_uses_kernels_and_matches_explicit_path(causal, monkeypatch):
    """nn_bracket.MultiHeadAttention under bf16 autocast: libmivae kernel route == the explicit masked-softmax
    expression of the reference (same module, fused routes switched off), outputs and parameter gradients"""
    from nonauto_lm_amd import nn_bracket as nb, ops
    torch.manual_seed(3)
    B, S, D, heads = 4, 48, 64, 4
    mha = nb.MultiHeadAttention(heads, D, dropout=0.0, mask_tril=causal).cuda()
    x = torch.randn(B, S, D, device="cuda")
    lens = torch.tensor([48, 17, 33, 1], device="cuda")
    mask = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).long()
    calls = []
    real = ops.attention
    monkeypatch.setattr(ops, "attention", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run(fused):
        monkeypatch.setattr(nb, "USE_FUSED_ATTENTION", fused)
        mha.zero_grad(set_to_none=True)
        xx = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mha(xx, xx, xx, mask)
        (y.float() * mask[..., None]).square().sum().backward()
        return y.float(), xx.grad.float(), [p.grad.float().clone() for p in mha.parameters()]

    y1, g1, p1 = run(True)
    assert calls, "the kernel route did not run"
    n = len(calls)
    y0, g0, p0 = run(False)
    assert len(calls) == n
    torch.testing.assert_close(y1, y0, rtol=3e-2, atol=3e-2)
    assert (g1 - g0).abs().max() <= 3e-2 * max(g0.abs().max().item(), 1.0)
    for a, b in zip(p1, p0):
        assert (a - b).abs().max() <= 3e-2 * max(b.abs().max().item(), 1.0)


@pytest.mark.parametrize("res_dtype", [torch.float32, torch.bfloat16])
def test_linear_residual_epilogue_bit_equal(res_dtype):
    """Linear.forward_residual (residual add as the epilogue of the libmivae product) == res + layer(x) bit for bit,
    same gradients (the transformer layer's attention sublayer uses it for its output projection)"""
    from nonauto_lm_amd import nn_bracket as nb
    torch.manual_seed(5)
    lin = nb.Linear(64, 64).cuda()
    x = torch.randn(64, 64, 64, device="cuda").to(torch.bfloat16)
    res = torch.randn(64, 64, 64, device="cuda").to(res_dtype)
    outs = []
    for fused in (True, False):
        lin.zero_grad(set_to_none=True)
        xx, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = lin.forward_residual(xx, rr) if fused else rr + lin(xx)
        assert y.dtype == res_dtype
        y.float().square().sum().backward()
        outs.append((y, xx.grad, rr.grad, lin.weight.grad.clone(), lin.bias.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_wide_qkv_group_matches_layerwise(monkeypatch):
    """MultiHeadAttention at 256 features under bf16 autocast with gradient sinks: the fused q|k|v product (one forward
    GEMM, packed dq|dk|dv -> one input-gradient GEMM, one batched weight-gradient product folded by column block) ==
    the three separate projections, outputs and every parameter gradient"""
    from nonauto_lm_amd import nn_bracket as nb, ops
    torch.manual_seed(7)
    B, S, D, heads = 64, 64, 256, 4
    mha = nb.MultiHeadAttention(heads, D, dropout=0.0).cuda()
    x = torch.randn(B, S, D, device="cuda")
    lens = torch.randint(1, S + 1, (B,), device="cuda")
    mask = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).long()
    res = {}
    for wide in (True, False):
        monkeypatch.setattr(nb, "WIDE_GROUP", wide)
        for sink in (False, True):
            for p in mha.parameters():
                p.grad = torch.zeros_like(p)
            xx = x.clone().requires_grad_(True)
            ops.GRAD_SINK = sink
            if sink:
                ops.begin_deferred_sinks()
            try:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = mha(xx, xx, xx, mask)
                (y.float() * mask[..., None]).square().sum().backward()
                ops.flush_grad_sinks()
            finally:
                ops.GRAD_SINK = False
                ops._DEFER = None
            res[(wide, sink)] = (y.float(), xx.grad.float(), [p.grad.clone() for p in mha.parameters()])
    ref = res[(False, False)]
    for key in ((True, False), (True, True), (False, True)):
        y, gx, gp = res[key]
        torch.testing.assert_close(y, ref[0], rtol=2e-2, atol=2e-2)
        assert (gx - ref[1]).abs().max() <= 2e-2 * max(ref[1].abs().max().item(), 1.0), key
        for a, b in zip(gp, ref[2]):
            assert (a - b).abs().max() <= 2e-2 * max(b.abs().max().item(), 1.0), key


def test_linear_wgrad_pair_bit_equal_to_two_launches(monkeypatch):
    """mivae_linear_wgrad_pair (both weight gradients of a feed-forward sublayer in one launch, deferred sinks) folds to
    exactly what two mivae_linear_wgrad launches fold to"""
    from nonauto_lm_amd import ops
    torch.manual_seed(11)
    N, D, H = 16384, 64, 128
    hmid, dy = torch.randn(N, H, device="cuda").to(torch.bfloat16), torch.randn(N, D, device="cuda").to(torch.bfloat16)
    xn, dh = torch.randn(N, D, device="cuda").to(torch.bfloat16), torch.randn(N, H, device="cuda").to(torch.bfloat16)
    outs = []
    for pair, queue in ((True, True), (True, False), (False, True), (False, False)):    # queued multi launch / hand-paired
        monkeypatch.setattr(ops, "WGRAD_QUEUE", queue)
        gw2, gb2 = torch.zeros(D, H, device="cuda"), torch.zeros(D, device="cuda")
        gw1, gb1 = torch.zeros(H, D, device="cuda"), torch.zeros(H, device="cuda")
        ops.begin_deferred_sinks()
        try:
            if pair:
                assert ops.linear_wgrad_pair_deferred(hmid, dy, (gw2, gb2), xn, dh, (gw1, gb1))
            else:
                ops.linear_wgrad(hmid, dy, sink=(gw2, gb2))
                ops.linear_wgrad(xn, dh, sink=(gw1, gb1))
            ops.flush_grad_sinks()
        finally:
            ops._DEFER = None
        outs.append((gw2, gb2, gw1, gb1))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    ref = dy.float().t() @ hmid.float()
    assert (outs[0][0] - ref).abs().max() <= 1e-3 * ref.abs().max()


def test_queued_weight_gradients_match_immediate_launches(monkeypatch):
    """a transformer layer's backward with deferred sinks: the queued weight gradients (one multi-problem launch per kind at
    flush) give bit-identical parameter gradients to launching them where autograd reaches them"""
    from nonauto_lm_amd import nn_bracket as nb, ops
    torch.manual_seed(13)
    layer = nb._TransformerLayer(64, 128, 4, dropout=0.0).cuda()
    x = torch.randn(256, 64, 64, device="cuda")
    mask = torch.ones(256, 64, dtype=torch.long, device="cuda")
    grads = []
    for queue in (True, False):
        monkeypatch.setattr(ops, "WGRAD_QUEUE", queue)
        for p in layer.parameters():
            p.grad = torch.zeros_like(p)
        xx = x.clone().requires_grad_(True)
        ops.GRAD_SINK = True
        ops.begin_deferred_sinks()
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = layer(xx, mask)
            y.float().square().mean().backward()
            ops.flush_grad_sinks()
        finally:
            ops.GRAD_SINK = False
            ops._DEFER = None
        grads.append([p.grad.clone() for p in layer.parameters()] + [xx.grad.clone()])
    assert any(g.abs().max() > 0 for g in grads[0][:-1])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_bf16_twin_of_residual_gradient(monkeypatch):
    """encoder-style fp32 residual stream: with the bf16 twin written by the LayerNorm backward the layer's backward is
    bit-identical to casting the fp32 gradient (and the twin is only handed out for the very tensor it mirrors)"""
    from nonauto_lm_amd import nn_bracket as nb, ops
    torch.manual_seed(17)
    layers = torch.nn.ModuleList([nb._TransformerLayer(64, 128, 4, dropout=0.0) for _ in range(2)]).cuda()
    x = torch.randn(256, 64, 64, device="cuda")
    mask = torch.ones(256, 64, dtype=torch.long, device="cuda")
    grads = []
    for twin in (True, False):
        monkeypatch.setattr(ops, "BF16_TWIN", twin)
        for p in layers.parameters():
            p.grad = torch.zeros_like(p)
        xx = x.clone().requires_grad_(True)
        ops.GRAD_SINK = True
        ops.begin_deferred_sinks()
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = xx
                for l in layers:
                    y = l(y, mask)
            y.float().square().mean().backward()
            ops.flush_grad_sinks()
        finally:
            ops.GRAD_SINK = False
            ops._DEFER = None
        grads.append([p.grad.clone() for p in layers.parameters()] + [xx.grad.clone()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    t32 = torch.randn(8, 64, device="cuda")
    ops._set_twin(t32, t32.to(torch.bfloat16))
    other = torch.randn(8, 64, device="cuda")
    assert ops.to_bf16(other).data_ptr() != ops._TWIN[2].data_ptr()
    assert ops.to_bf16(t32).data_ptr() == ops._TWIN[2].data_ptr()
    t32.add_(1.0)                                         # modified in place: the twin is stale, a fresh cast is returned
    assert ops.to_bf16(t32).data_ptr() != ops._TWIN[2].data_ptr()
    ops._TWIN = None


def test_linear_nll_weight_gradient_sink_matches_autograd():
    """ops.linear_nll (fused vocabulary projection + NLL) with gradient sinks: dW / db folded straight into .grad by
    mivae_fold_aug_sink == the gradients autograd accumulates without sinks (same partial products, same fold order)"""
    from nonauto_lm_amd import ops
    torch.manual_seed(19)
    Bn, S, E, V = 128, 64, 64, 1024
    h = torch.randn(Bn, S, E, device="cuda")
    W = (torch.randn(V, E, device="cuda") * 0.1).requires_grad_(True)
    b = torch.zeros(V, device="cuda").requires_grad_(True)
    tgt = torch.randint(4, V, (Bn, S), device="cuda")
    wts = torch.ones(Bn, S, device="cuda")
    res = []
    for sink in (False, True):
        W.grad, b.grad = torch.zeros_like(W), torch.zeros_like(b)
        hh = h.clone().requires_grad_(True)
        ops.GRAD_SINK = sink
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = ops.linear_nll(hh.to(torch.bfloat16), W, b, tgt, wts, 0.1)
            loss.sum().backward()
        finally:
            ops.GRAD_SINK = False
        res.append((W.grad.clone(), b.grad.clone(), hh.grad.clone()))
    assert res[0][0].abs().max() > 0 and res[0][1].abs().max() > 0
    for a, c in zip(*res):
        torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-6)
