"""One-launch mixed flow stack (csrc/mixed_stack.hip: NICE-FF / ActNorm / invertible-linear steps in any order,
evaluation and sampling directions) against the CPU oracle and against the per-flow HIP path."""
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import latent_path as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def M():
    import importlib
    return importlib.import_module("nonauto_lm_amd")


def _build(M, D, H, act, heads, layout, seed, wstd=0.25):
    """flows from a layout string: a = ActNorm, n = NICE, l = inv-linear, m = inv-multi-head."""
    torch.manual_seed(seed)
    flows = []
    for ch in layout:
        if ch == "a":
            f = M.Flow.from_params(type="actnorm", input_size=D)
            f._log_scale.data.normal_(0, 0.3)
            f._bias.data.normal_(0, 0.5)
            f._is_initted = True
        elif ch == "n":
            f = M.Flow.from_params(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H],
                                                              activation=act))
            for p in f.parameters():
                p.data.normal_(0, wstd)
        elif ch == "l":
            f = M.InvertibleLinear.from_params(type="inv-linear", input_size=D)
        else:
            f = M.InvertibleLinear.from_params(type="inv-multi-head", input_size=D, num_heads=heads)
        flows.append(f)
    return M.FlowPrior(D, flows).to(DEV)


def _oracle_steps(prior):
    steps = []
    for f in prior._flows:
        name = type(f).__name__
        if name == "ActNorm":
            ls, b = f._log_scale.detach().cpu(), f._bias.detach().cpu()
            steps.append(lambda z, m, inv, ls=ls, b=b: O.actnorm_step(z, ls, b, m, inv))
        elif name == "NICECoupling":
            sd = {k: v.detach().cpu() for k, v in f.state_dict().items()}
            p = O.NiceParams.from_state_dict(sd, "", activation=f._s_net._activation_name)
            steps.append(lambda z, m, inv, p=p: O.nice_step(z, p, m, inv))
        else:
            w = f._weight.detach().cpu()
            heads = getattr(f, "_num_heads", 1)
            steps.append(lambda z, m, inv, w=w, heads=heads: O.invlinear_step(z, w, m, inv, heads))
    return steps


def _count_calls(M, monkeypatch):
    calls = []
    real = M.ops.mixed_stack

    def counted(*a, **k):
        calls.append(1)
        return real(*a, **k)
    monkeypatch.setattr(M.ops, "mixed_stack", counted)
    return calls


CASES = [
    # D, H, act, heads, layout, B, S
    (16, 24, "gelu", 2, "analna", 5, 11),
    (64, 64, "gelu", 4, "anmanl", 9, 37),
    (64, 128, "swish", 8, "mnnla", 3, 64),
    (96, 96, "tanh", 3, "anln", 4, 19),
    (128, 128, "gelu", 4, "alnmn", 6, 33),
    (6, 5, "relu", 3, "lnam", 7, 5),
]


@pytest.mark.parametrize("D,H,act,heads,layout,B,S", CASES)
def test_mixed_stack_both_directions_vs_oracle(M, monkeypatch, D, H, act, heads, layout, B, S):
    prior = _build(M, D, H, act, heads, layout, seed=D + H)
    calls = _count_calls(M, monkeypatch)
    gen = torch.Generator().manual_seed(S)
    z = torch.randn(B, S, D, generator=gen)
    lengths = torch.randint(0, S + 1, (B,), generator=gen)
    lengths[0] = S
    mask = (torch.arange(S)[None, :] < lengths[:, None]).long()
    steps = _oracle_steps(prior)
    ref_z, ref_lp = O.flow_prior_forward(z, steps, mask)
    inv_z, inv_lp = O.flow_prior_inverse(z, steps, mask)
    with torch.no_grad():
        got_z, got_lp = prior.forward_pass(z.to(DEV), mask.to(DEV))
        gi_z, gi_lp = prior.backward_pass(z.to(DEV), mask.to(DEV))
    assert len(calls) == 2, "the one-launch path did not run"
    assert rel_err(got_z, ref_z) < 2e-5 and rel_err(got_lp, ref_lp) < 2e-5
    assert rel_err(gi_z, inv_z) < 2e-5 and rel_err(gi_lp, inv_lp) < 2e-5
    # same numbers as the per-flow HIP path
    prior.allow_fused = False
    with torch.no_grad():
        pz, plp = prior.forward_pass(z.to(DEV), mask.to(DEV))
    assert len(calls) == 2
    assert rel_err(got_z, pz) < 1e-5 and rel_err(got_lp, plp) < 1e-5


def test_mixed_stack_two_dim_and_bf16(M, monkeypatch):
    calls = _count_calls(M, monkeypatch)
    prior = _build(M, 32, 48, "gelu", 2, "anlnm", seed=5, wstd=0.08)
    z = torch.randn(77, 32, generator=torch.Generator().manual_seed(0))
    steps = _oracle_steps(prior)
    ref_z, ref_lp = O.flow_prior_forward(z, steps, None)
    with torch.no_grad():
        got_z, got_lp = prior.forward_pass(z.to(DEV), None)
    assert len(calls) == 1
    assert rel_err(got_z, ref_z) < 2e-5 and rel_err(got_lp, ref_lp) < 2e-5
    # bf16 latents: fp32 arithmetic inside, bf16 at the two ends
    B, S = 4, 21
    zb = torch.randn(B, S, 32, generator=torch.Generator().manual_seed(1)).bfloat16()
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, 9:] = 0
    ref_z, ref_lp = O.flow_prior_forward(zb.float(), steps, mask)
    with torch.no_grad():
        got_z, got_lp = prior.forward_pass(zb.to(DEV), mask.to(DEV))
    assert len(calls) == 2 and got_z.dtype == torch.bfloat16
    assert rel_err(got_z.float(), ref_z) < 1e-2 and rel_err(got_lp, ref_lp) < 1e-2


def test_mixed_stack_golden_no_grad(M, monkeypatch):
    """mixed_stack_d16 (ActNorm, NICE, inv-linear, NICE, ActNorm) from the reference, both directions in one launch."""
    g = load_golden("mixed_stack_d16")
    D = 16
    H = g["sd._flows.1._s_net._feedforward._linear_layers.0.weight"].shape[0]
    nice = lambda: dict(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H]))
    flows = [M.Flow.from_params(type="actnorm", input_size=D), M.Flow.from_params(**nice()),
             M.InvertibleLinear.from_params(type="inv-linear", input_size=D), M.Flow.from_params(**nice()),
             M.Flow.from_params(type="actnorm", input_size=D)]
    prior = M.FlowPrior(D, flows)
    sd = {k[len("sd."):]: v for k, v in g.items() if k.startswith("sd.")}       # post-init parameters
    prior.load_state_dict(sd, strict=False)
    for k in (0, 4):
        prior._flows[k]._is_initted = True
    prior = prior.to(DEV)
    calls = _count_calls(M, monkeypatch)
    z0, mask = g["z0"].to(DEV), g["mask"].to(DEV)
    with torch.no_grad():
        zK, logp = prior.forward_pass(z0, mask)
        zi, lpi = prior.backward_pass(g["fwd.zK"].to(DEV), mask)
    assert len(calls) == 2
    assert rel_err(zK, g["fwd.zK"]) < 1e-5 and rel_err(logp, g["fwd.logp"]) < 1e-5
    assert rel_err(zi, g["inv.zK"]) < 1e-5 and rel_err(lpi, g["inv.logp"]) < 1e-5


def test_mixed_stack_not_used_when_gradients_are_needed_or_uninitialised(M, monkeypatch):
    calls = _count_calls(M, monkeypatch)
    prior = _build(M, 16, 16, "gelu", 2, "ana", seed=1)
    z = torch.randn(3, 7, 16, device=DEV)
    mask = torch.ones(3, 7, dtype=torch.long, device=DEV)
    prior.forward_pass(z, mask)                      # parameters require grad: per-flow autograd path
    assert not calls
    prior._flows[0]._is_initted = False
    with torch.no_grad():
        prior.forward_pass(z, mask)                  # first ActNorm batch: the data-dependent init has to run
    assert not calls and prior._flows[0]._is_initted
    with torch.no_grad():
        prior.forward_pass(z, mask)
    assert len(calls) == 1


def test_mixed_stack_entry_point_rejects_bad_tables(M):
    import ctypes
    L = M._lib
    z = torch.randn(2, 4, 16, device=DEV)
    tab = (ctypes.c_int64 * 12)(7, 0, 0, 0, *([0] * 8))
    rc = L.lib().mivae_mixed_stack(z.data_ptr(), z.data_ptr(), None, None, ctypes.addressof(tab), 1, 2, 4, 1, 16, 0, 0, None)
    assert rc != 0 and b"unknown kind" in L.lib().mivae_last_error()
    tab = (ctypes.c_int64 * 12)(2, 0, 0, 5, z.data_ptr(), *([0] * 7))
    rc = L.lib().mivae_mixed_stack(z.data_ptr(), z.data_ptr(), None, None, ctypes.addressof(tab), 1, 2, 4, 1, 16, 0, 0, None)
    assert rc != 0 and b"does not divide" in L.lib().mivae_last_error()
    assert not M.ops.mixed_stack_supported(130, 64) and not M.ops.mixed_stack_supported(64, 256)
    assert M.ops.mixed_stack_supported(128, 128)


def test_mixed_stack_plan_cache_follows_parameter_updates(M, monkeypatch):
    prior = _build(M, 16, 16, "gelu", 2, "aln", seed=2)
    z = torch.randn(3, 7, 16, device=DEV)
    mask = torch.ones(3, 7, dtype=torch.long, device=DEV)
    slog = []
    real = torch.linalg.slogdet
    monkeypatch.setattr(torch.linalg, "slogdet", lambda w: (slog.append(1), real(w))[1])
    with torch.no_grad():
        a = prior.forward_pass(z, mask)
        b = prior.forward_pass(z, mask)
        assert len(slog) == 1 and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])    # second call: cached plan
        prior._flows[1]._weight.mul_(1.5)                                               # in-place update
        c = prior.forward_pass(z, mask)
        assert len(slog) == 2
    steps = _oracle_steps(prior)
    ref_z, ref_lp = O.flow_prior_forward(z.cpu(), steps, mask.cpu())
    assert rel_err(c[0], ref_z) < 2e-5 and rel_err(c[1], ref_lp) < 2e-5


def test_mixed_stack_plan_cache_sees_fused_adam_updates(M):
    """FusedAdam writes parameters through raw pointers (no torch version bump): the cached inverse / slogdet of a linear
    step must still be refreshed."""
    from nonauto_lm_amd import dp
    prior = _build(M, 16, 16, "gelu", 2, "anl", seed=4)
    reducer = dp.GradReducer(prior.parameters(), flatten_params=True)
    opt = dp.FusedAdam(reducer, lr=5e-2)
    z = torch.randn(3, 7, 16, device=DEV)
    mask = torch.ones(3, 7, dtype=torch.long, device=DEV)
    with torch.no_grad():
        before = prior.backward_pass(z, mask)
    zz, lp = prior.forward_pass(z, mask)
    lp.sum().backward()
    reducer.finish()
    opt.step()
    with torch.no_grad():
        after = prior.backward_pass(z, mask)
    steps = _oracle_steps(prior)
    ref_z, ref_lp = O.flow_prior_inverse(z.cpu(), steps, mask.cpu())
    assert rel_err(after[0], ref_z) < 2e-5 and rel_err(after[1], ref_lp) < 2e-5
    assert not torch.equal(before[0], after[0])


# ------------------------------------------------------------------------------------ training path
def _oracle_steps_grad(prior):
    """oracle steps on CPU leaf copies of the prior's parameters: (steps, {parameter name: leaf})"""
    leaves = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in prior.named_parameters()}
    steps = []
    for i, f in enumerate(prior._flows):
        name, pre = type(f).__name__, f"_flows.{i}."
        if name == "ActNorm":
            steps.append(lambda z, m, inv, ls=leaves[pre + "_log_scale"], b=leaves[pre + "_bias"]: O.actnorm_step(z, ls, b, m, inv))
        elif name == "NICECoupling":
            p = O.NiceParams.from_state_dict(leaves, pre, activation=f._s_net._activation_name)
            steps.append(lambda z, m, inv, p=p: O.nice_step(z, p, m, inv))
        else:
            heads = getattr(f, "_num_heads", 1)
            steps.append(lambda z, m, inv, w=leaves[pre + "_weight"], heads=heads: O.invlinear_step(z, w, m, inv, heads))
    return steps, leaves


def _count_train_calls(M, monkeypatch):
    calls = []
    real = M.ops.mixed_stack_train

    def counted(*a, **k):
        calls.append(1)
        return real(*a, **k)
    monkeypatch.setattr(M.ops, "mixed_stack_train", counted)
    return calls


@pytest.mark.parametrize("D,H,act,heads,layout,B,S", CASES + [(64, 64, "gelu", 4, "nanlnmna", 300, 33)])
def test_mixed_stack_training_path_vs_oracle(M, monkeypatch, D, H, act, heads, layout, B, S):
    """FlowPrior.forward_pass of a mixed list UNDER AUTOGRAD (flow_prior.py:37-52 through actnorm.py:30-39,
    invertible_flows.py:26-33 / :58-66, couplings.py:141-162): mivae_mixed_stack_train_fwd + mivae_mixed_stack_bwd and the
    parameter reductions against the CPU oracle's autograd -- z_K, log p, d z0 and the gradient of every parameter
    (fp32: 2e-5 outputs, 1e-4 gradients; the widths cover one and two 32-feature tiles, odd half widths, 4- and 2-wave
    blocks, ragged lengths with empty sentences, multi-block reductions)."""
    prior = _build(M, D, H, act, heads, layout, seed=D + H, wstd=0.25 if B < 100 else 0.1)
    calls = _count_train_calls(M, monkeypatch)
    gen = torch.Generator().manual_seed(S)
    z = torch.randn(B, S, D, generator=gen)
    lengths = torch.randint(0, S + 1, (B,), generator=gen)
    lengths[0] = S
    mask = (torch.arange(S)[None, :] < lengths[:, None]).long()
    cz, cl = torch.randn(B, S, D, generator=gen), torch.randn(B, generator=gen)
    steps, leaves = _oracle_steps_grad(prior)
    zr = z.clone().requires_grad_(True)
    ref_z, ref_lp = O.flow_prior_forward(zr, steps, mask)
    ((ref_z * cz).sum() + (ref_lp * cl).sum()).backward()
    zg = z.to(DEV).requires_grad_(True)
    got_z, got_lp = prior.forward_pass(zg, mask.to(DEV))
    assert len(calls) == 1, "the training path did not run"
    assert rel_err(got_z, ref_z) < 2e-5 and rel_err(got_lp, ref_lp) < 2e-5
    ((got_z * cz.to(DEV)).sum() + (got_lp * cl.to(DEV)).sum()).backward()
    assert rel_err(zg.grad, zr.grad) < 1e-4, rel_err(zg.grad, zr.grad)
    worst = {k: rel_err(p.grad, leaves[k].grad) for k, p in prior.named_parameters()}
    assert max(worst.values()) < 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:4]


def test_mixed_stack_training_path_two_dim_bf16_and_fallback(M, monkeypatch):
    calls = _count_train_calls(M, monkeypatch)
    # 2-D latents (the auto-regressive model's (B, D) code): no mask, parameter log-dets count once
    prior = _build(M, 32, 48, "gelu", 2, "anlnm", seed=5, wstd=0.08)
    z = torch.randn(77, 32, generator=torch.Generator().manual_seed(0))
    steps, leaves = _oracle_steps_grad(prior)
    zr = z.clone().requires_grad_(True)
    ref_z, ref_lp = O.flow_prior_forward(zr, steps, None)
    (ref_z.square().sum() + ref_lp.sum()).backward()
    zg = z.to(DEV).requires_grad_(True)
    got_z, got_lp = prior.forward_pass(zg, None)
    (got_z.square().sum() + got_lp.sum()).backward()
    assert len(calls) == 1
    assert rel_err(got_z, ref_z) < 2e-5 and rel_err(got_lp, ref_lp) < 2e-5 and rel_err(zg.grad, zr.grad) < 1e-4
    worst = {k: rel_err(p.grad, leaves[k].grad) for k, p in prior.named_parameters()}
    assert max(worst.values()) < 1e-4, worst
    # bf16 latents: fp32 arithmetic inside, bf16 at the two ends
    B, S = 4, 21
    zb = torch.randn(B, S, 32, generator=torch.Generator().manual_seed(1)).bfloat16()
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, 9:] = 0
    for p in prior.parameters():
        p.grad = None
    steps, leaves = _oracle_steps_grad(prior)
    zr = zb.float().requires_grad_(True)
    ref_z, ref_lp = O.flow_prior_forward(zr, steps, mask)
    ref_lp.sum().backward()
    zg = zb.to(DEV).requires_grad_(True)
    got_z, got_lp = prior.forward_pass(zg, mask.to(DEV))
    got_lp.sum().backward()
    assert len(calls) == 2 and got_z.dtype == torch.bfloat16 and zg.grad.dtype == torch.bfloat16
    assert rel_err(got_lp, ref_lp) < 1e-2 and rel_err(zg.grad.float(), zr.grad) < 1e-2
    worst = {k: rel_err(p.grad, leaves[k].grad) for k, p in prior.named_parameters()}
    assert max(worst.values()) < 1e-2, worst
    # coupling steps of different hidden widths: the per-flow loop runs, same numbers
    prior2 = _build(M, 16, 16, "gelu", 2, "ana", seed=1)
    wide = M.Flow.from_params(type="nice", coupling=dict(type="feed-forward", input_size=8, hidden_sizes=[24])).to(DEV)
    prior2._flows.append(wide)
    z2 = torch.randn(3, 7, 16, device=DEV, requires_grad=True)
    prior2.forward_pass(z2, torch.ones(3, 7, dtype=torch.long, device=DEV))[1].sum().backward()
    assert len(calls) == 2 and z2.grad is not None
