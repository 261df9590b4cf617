"""GPU parity of the wide-latent NICE stack kernels (csrc/nice_big.hip: BASELINE configs C3 D=256 K=16 and
C5 D=512 K=32 ragged) against the fp32 CPU oracle (oracle/latent_path.py, pinned to the reference's goldens).

bf16 latents: the oracle is evaluated in fp32 on the same bf16-rounded inputs and weights.  Tolerances (tensor-level
max-norm relative error, conftest.rel_err) are per shape, 2x the error measured on MI355X and recorded next to each bound
(BIG_BOUNDS): z_K 2-3e-3 (one bf16 rounding per coupling), log p 2-7e-5 (fp32 accumulation), gradients 2-8e-3 (d s / d t
and the hidden-layer cotangents are rounded to bf16 once before the weight-gradient products).  fp32 latents of the same widths run the composed path (library MLP +
HIP affine / log-det tail) and keep the fp32 bound 1e-5 (gradients 5e-5)."""
import pytest
import torch

from conftest import rel_err
from oracle import latent_path as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def M():
    import nonauto_lm_amd as pkg
    pkg._lib.lib()
    return pkg


def make_case(B, S, D, H, K, act, seed, ragged, round_bf16):
    gen = torch.Generator().manual_seed(seed)
    h = D // 2
    rb = (lambda t: t.to(torch.bfloat16).float()) if round_bf16 else (lambda t: t)
    mk = lambda *s: torch.randn(*s, generator=gen)
    ps, flat = [], []
    for _ in range(K):
        w = [rb(mk(H, h) * (0.5 / h ** 0.5)), mk(H) * 0.1, rb(mk(h, H) * (0.5 / H ** 0.5)), mk(h) * 0.05,
             rb(mk(H, h) * (0.5 / h ** 0.5)), mk(H) * 0.1, rb(mk(h, H) * (0.5 / H ** 0.5)), mk(h) * 0.05]
        flat += w
        ps.append(O.NiceParams([w[0]], [w[1]], w[2], w[3], [w[4]], [w[5]], w[6], w[7], activation=act))
    z0 = rb(mk(B, S, D))
    if ragged:
        lens = torch.randint(1, S + 1, (B,), generator=gen)
        lens[0] = S                       # one full sentence, one of length 1, the rest random
        if B > 1:
            lens[1] = 1
        if B > 2:
            lens[2] = max(1, S - 200)     # a sentence whose tail covers whole 128-token blocks of padding
    else:
        lens = torch.full((B,), S)
    mask = (torch.arange(S)[None, :] < lens[:, None]).long()
    z0 = z0 * mask.unsqueeze(-1)          # what the posterior hands over: zeros at padding
    cz, cl = rb(mk(B, S, D)), mk(B)
    return ps, flat, z0, mask, cz, cl


def oracle_run(ps, flat, z0, mask, cz, cl):
    zr = z0.clone().requires_grad_(True)
    for t in flat:
        t.requires_grad_(True)
    zK, lp = O.flow_prior_forward(zr, O.nice_steps(ps), mask, 0.0, 1.0)
    ((zK * cz).sum() + (lp * cl).sum()).backward()
    grads = [t.grad.clone() for t in flat]
    for t in flat:
        t.requires_grad_(False)
        t.grad = None
    return zK.detach(), lp.detach(), zr.grad, grads


BIG_SHAPES = [
    # B, S, D, H, K, act, ragged
    (4, 128, 256, 256, 16, "gelu", False),     # C3 shape family
    (3, 50, 208, 200, 3, "gelu", True),        # h = 104, H = 200: padded feature / hidden tiles, N % 128 != 0
    (2, 37, 256, 96, 2, "tanh", True),         # generic-activation instantiation
    (2, 512, 512, 512, 32, "gelu", True),      # C5 shape family: ragged lengths incl. all-padding 128-token blocks
    (5, 64, 400, 320, 4, "relu", True),        # h = 200 on the HT = 8 instantiation with padded tiles
]


# per-shape bounds = 2x the error measured on MI355X (recorded in the comment of each row), not a blanket 2e-2 / 3e-2
BIG_BOUNDS = {
    # measured:        z 2.2e-3, log p 6.9e-5, dz 3.3e-3, d params 7.5e-3, inverse 3.7e-3
    (256, 256, 16): dict(z=5e-3, logp=1.5e-4, dz=7e-3, dparams=1.5e-2, inverse=8e-3),
    # measured:        z 2.2e-3, log p 2.6e-5, dz 3.9e-3, d params 6.0e-3, inverse 2.4e-3
    (208, 200, 3): dict(z=5e-3, logp=6e-5, dz=8e-3, dparams=1.2e-2, inverse=5e-3),
    # measured:        z 2.0e-3, log p 1.8e-5, dz 2.2e-3, d params 4.7e-3, inverse 2.6e-3
    (256, 96, 2): dict(z=4e-3, logp=4e-5, dz=5e-3, dparams=1e-2, inverse=6e-3),
    # measured:        z 2.4e-3, log p 1.8e-5, dz 3.2e-3, d params 7.0e-3, inverse 1.5e-2 (32 inverse couplings in bf16)
    (512, 512, 32): dict(z=5e-3, logp=4e-5, dz=7e-3, dparams=1.5e-2, inverse=3e-2),
    # measured:        z 3.0e-3, log p 1.8e-5, dz 3.3e-3, d params 4.6e-3, inverse 2.7e-3
    (400, 320, 4): dict(z=6e-3, logp=4e-5, dz=7e-3, dparams=1e-2, inverse=6e-3),
}


@pytest.mark.parametrize("B,S,D,H,K,act,ragged", BIG_SHAPES)
def test_nice_big_bf16_vs_fp32_oracle(M, B, S, D, H, K, act, ragged):
    ps, flat, z0, mask, cz, cl = make_case(B, S, D, H, K, act, 100 + D + K, ragged, True)
    zK_r, lp_r, dz_r, g_r = oracle_run(ps, flat, z0, mask, cz, cl)
    assert M.ops.nice_big_supported(z0.to(DEV).bfloat16(), H)
    gp = [t.to(DEV).clone().requires_grad_(True) for t in flat]
    zg = z0.to(DEV).bfloat16().requires_grad_(True)
    zK, lp = M.ops.nice_stack(zg, mask.to(DEV), gp, H, act, inverse=False, base=True, mu=0.0, std=1.0)
    assert zK.dtype == torch.bfloat16 and lp.dtype == torch.float32
    errs = {"z": rel_err(zK.float(), zK_r), "logp": rel_err(lp, lp_r)}
    # padding positions are exact zeros (couplings.py:159-161)
    assert float((zK.float().cpu() * (1 - mask).unsqueeze(-1)).abs().max()) == 0.0
    ((zK.float() * cz.to(DEV)).sum() + (lp * cl.to(DEV)).sum()).backward()
    errs["dz"] = rel_err(zg.grad.float(), dz_r)
    assert float((zg.grad.float().cpu() * (1 - mask).unsqueeze(-1)).abs().max()) == 0.0
    errs["dparams"] = max(rel_err(a.grad, b) for a, b in zip(gp, g_r))
    with torch.no_grad():
        zi, _ = M.ops.nice_stack(zK_r.to(DEV).bfloat16(), mask.to(DEV), [t.detach() for t in gp], H, act, inverse=True)
        zi_r2, _ = O.flow_prior_inverse(zK_r.to(torch.bfloat16).float(), O.nice_steps(ps), mask.float())
        errs["inverse"] = rel_err(zi.float(), zi_r2)
    bound = BIG_BOUNDS[(D, H, K)]
    print(f"nice_big bf16 vs fp32 oracle D={D} H={H} K={K} {act}: measured", {k: f"{v:.2e}" for k, v in errs.items()},
          "bounds", bound)
    for k, v in errs.items():
        assert v < bound[k], (k, v, bound[k])


def test_nice_big_grad_sink_matches_returned_grads(M):
    """mivae_nice_big_bwd with a gradient-pointer table (adds into .grad) == the returned-gradient form, bit for bit."""
    ps, flat, z0, mask, cz, cl = make_case(3, 70, 256, 160, 2, "gelu", 7, True, True)
    outs = []
    for sink in (False, True):
        gp = [t.to(DEV).clone().requires_grad_(True) for t in flat]
        if sink:
            for p in gp:
                p.grad = torch.zeros_like(p)
        zg = z0.to(DEV).bfloat16().requires_grad_(True)
        M.ops.GRAD_SINK = sink
        try:
            zK, lp = M.ops.nice_stack(zg, mask.to(DEV), gp, 160, "gelu", base=True)
            ((zK.float() * cz.to(DEV)).sum() + (lp * cl.to(DEV)).sum()).backward()
        finally:
            M.ops.GRAD_SINK = False
        outs.append([p.grad.clone() for p in gp] + [zg.grad.clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,S,D,H,K", [(2, 96, 512, 512, 4), (3, 40, 256, 256, 3)])
def test_nice_wide_fp32_composed_path(M, B, S, D, H, K):
    """fp32 latents at D = 256 / 512 through the module API (fused small-D kernel for D = 256, composed path for
    D = 512): fp32 bound."""
    ps, flat, z0, mask, cz, cl = make_case(B, S, D, H, K, "gelu", 5 + D, True, False)
    zK_r, lp_r, dz_r, g_r = oracle_run(ps, flat, z0, mask, cz, cl)
    prior = M.Prior.from_params(type="flow", features=D, flows=[
        dict(type="nice", coupling=dict(type="feed-forward", input_size=D // 2, hidden_sizes=[H], activation="gelu"))
        for _ in range(K)]).to(DEV)
    with torch.no_grad():
        for k in range(K):
            f = prior._flows[k]
            for net, off in ((f._s_net, 0), (f._t_net, 4)):
                ff = net._feedforward
                ff._linear_layers[0].weight.copy_(flat[8 * k + off]); ff._linear_layers[0].bias.copy_(flat[8 * k + off + 1])
                ff._output.weight.copy_(flat[8 * k + off + 2]); ff._output.bias.copy_(flat[8 * k + off + 3])
    zg = z0.to(DEV).requires_grad_(True)
    zK, lp = prior.forward_pass(zg, mask.to(DEV))
    assert rel_err(zK, zK_r) < 1e-5 and rel_err(lp, lp_r) < 1e-5
    ((zK * cz.to(DEV)).sum() + (lp * cl.to(DEV)).sum()).backward()
    assert rel_err(zg.grad, dz_r) < 5e-5
    names = ["_linear_layers.0.weight", "_linear_layers.0.bias", "_output.weight", "_output.bias"]
    for k in range(K):
        f = prior._flows[k]
        for net, off in ((f._s_net, 0), (f._t_net, 4)):
            sd = dict(net._feedforward.named_parameters())
            for i, nm in enumerate(names):
                assert rel_err(sd[nm].grad, g_r[8 * k + off + i]) < 5e-5, (k, off, nm)


@pytest.mark.parametrize("D,H", [(256, 160), (512, 288)])
def test_nice_big_saved_hidden_matches_recompute(M, D, H):
    """The two weight-gradient routes of mivae_nice_big_bwd -- hidden activations / d pre saved by the nets launch and
    contracted directly (k_big_dw2), or recomputed from x0 and d s / d t (k_big_dw) -- agree to bf16 rounding (the saved
    route sums the bf16-rounded d pre for the bias gradient, the recompute route its fp32 value); dz is the same kernel either
    way (bit equal).  The default picks by width (saved for D/2 > 128), so the oracle tests above cover the recompute
    route at D = 256 and the saved route at D = 512; this test runs both routes at both widths."""
    ps, flat, z0, mask, cz, cl = make_case(3, 70, D, H, 2, "gelu", 11, True, True)
    outs = []
    for save in (True, False):
        gp = [t.to(DEV).clone().requires_grad_(True) for t in flat]
        zg = z0.to(DEV).bfloat16().requires_grad_(True)
        M.ops.NICE_BIG_SAVE_HIDDEN = save
        try:
            zK, lp = M.ops.nice_stack(zg, mask.to(DEV), gp, H, "gelu", base=True)
            ((zK.float() * cz.to(DEV)).sum() + (lp * cl.to(DEV)).sum()).backward()
        finally:
            M.ops.NICE_BIG_SAVE_HIDDEN = "auto"
        outs.append(([p.grad.clone() for p in gp], zg.grad.clone()))
    assert torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][0], outs[1][0]):
        assert rel_err(a, b) < 1e-2, rel_err(a, b)
