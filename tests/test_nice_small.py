"""Backward of small bf16 NICE stacks with the coupling nets of all steps side by side (k_stack_bwd_par,
csrc/nice_small.hip, behind mivae_nice_stack_bwd) against (a) the fp32 CPU oracle's autograd on the same bf16-rounded
inputs and weights (couplings.py:141-162 + nn/feedforward.py:93-98 under FlowPrior.forward_pass, flow_prior.py:37-52)
and (b) the task-serial kernel of nice_stack.hip it replaces for these shapes (MIVAE_NICE_BWD_PAR=0), which must agree
to the bf16 rounding of one intermediate (act').

Tolerances: bf16 latents, fp32 accumulation.  Bounds are 2x the errors measured on MI355X (printed by the test)."""
import os

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


def _case(B, S, D, H, K, act, seed, full=False, holes=False):
    gen = torch.Generator().manual_seed(seed)
    h = D // 2
    rb = lambda t: t.to(torch.bfloat16).float()
    ps, flat = [], []
    for _ in range(K):
        mk = lambda *s: torch.randn(*s, generator=gen)
        w = [rb(mk(H, h) * (0.5 / h ** 0.5)), mk(H) * 0.1, rb(mk(h, H) * (0.5 / H ** 0.5)), mk(h) * 0.1,
             rb(mk(H, h) * (0.5 / h ** 0.5)), mk(H) * 0.1, rb(mk(h, H) * (0.5 / H ** 0.5)), mk(h) * 0.1]
        flat += w
        ps.append(O.NiceParams([w[0]], [w[1]], w[2], w[3], [w[4]], [w[5]], w[6], w[7], activation=act))
    z0 = rb(torch.randn(B, S, D, generator=gen))
    lens = torch.full((B,), S) if full else torch.randint(1, S + 1, (B,), generator=gen)
    if holes:                       # whole sentences of padding: all-padding 32-token tiles
        lens[::3] = 0
    mask = (torch.arange(S)[None, :] < lens[:, None]).long()
    cz = rb(torch.randn(B, S, D, generator=gen))
    cl = torch.randn(B, generator=gen)
    return ps, flat, z0, mask, cz, cl


def _run_gpu(M, flat, z0, mask, cz, cl, H, act, one_pass, use_cz=True, base=True):
    os.environ["MIVAE_NICE_BWD_PAR"] = "1" if one_pass else "0"      # 0: the task-serial kernel of nice_stack.hip
    try:
        gp = [t.detach().to(DEV).clone().requires_grad_(True) for t in flat]
        zg = z0.to(DEV).bfloat16().requires_grad_(True)
        zK, lp = M.ops.nice_stack(zg, mask.to(DEV), gp, H, act, inverse=False, base=base, mu=0.1, std=1.3)
        loss = (lp * cl.to(DEV)).sum()
        if use_cz:
            loss = loss + (zK.float() * cz.to(DEV)).sum()
        loss.backward()
        torch.cuda.synchronize()
        return zK, lp, zg.grad, [p.grad for p in gp]
    finally:
        os.environ.pop("MIVAE_NICE_BWD_PAR", None)


@pytest.mark.parametrize("B,S,D,H,K,act,kw", [
    (16, 64, 64, 64, 4, "gelu", {}),                    # the C2 latent shape family, ragged
    (256, 64, 64, 64, 4, "gelu", {"full": True}),       # C2 itself: 512 tiles on 256 blocks, two tiles per block
    (700, 64, 64, 64, 4, "gelu", {}),                   # more tiles than 2 x 256 blocks: uneven tile counts per block
    (9, 37, 64, 64, 4, "gelu", {"holes": True}),        # all-padding tiles, N not a multiple of 32
    (5, 19, 48, 40, 3, "swish", {}),                    # h = 24, H = 40: padded feature / hidden columns
    (4, 33, 16, 16, 1, "tanh", {}),                     # one step, one hidden chunk
    (7, 21, 32, 64, 2, "relu", {}),
    (6, 50, 64, 32, 4, "sigmoid", {}),
    (256, 1, 64, 64, 4, "gelu", {"full": True}),        # C4: a (B, D) latent as S = 1
])
def test_parallel_nets_backward_vs_oracle_and_task_serial_kernel(M, B, S, D, H, K, act, kw):
    ps, flat, z0, mask, cz, cl = _case(B, S, D, H, K, act, seed=B * 100 + D + K, **kw)
    zr = z0.clone().requires_grad_(True)
    for p in ps:
        for t in p.tensors():
            t.requires_grad_(True)
    zK_r, lp_r = O.flow_prior_forward(zr, O.nice_steps(ps), mask, 0.1, 1.3)
    ((zK_r * cz).sum() + (lp_r * cl).sum()).backward()
    ref_flat = []
    for p in ps:
        ref_flat += [p.s_w[0], p.s_b[0], p.s_wo, p.s_bo, p.t_w[0], p.t_b[0], p.t_wo, p.t_bo]
    zK, lp, dz, grads = _run_gpu(M, flat, z0, mask, cz, cl, H, act, one_pass=True)
    zK2, lp2, dz2, grads2 = _run_gpu(M, flat, z0, mask, cz, cl, H, act, one_pass=False)
    assert torch.equal(zK, zK2) and torch.equal(lp, lp2)          # same forward kernel
    e_dz = rel_err(dz.float(), zr.grad)
    e_w = max(rel_err(a, b.grad) for a, b in zip(grads, ref_flat))
    e_dz2 = rel_err(dz.float(), dz2.float())
    e_w2 = max(rel_err(a, b) for a, b in zip(grads, grads2))
    print(f"parallel-nets bwd {B}x{S}x{D} H{H} K{K} {act}: vs oracle dz {e_dz:.2e} dW {e_w:.2e}; vs task-serial kernel dz {e_dz2:.2e} dW {e_w2:.2e}")
    assert not torch.isnan(dz.float()).any()
    assert e_dz < 1.1e-2 and e_w < 1.7e-2
    assert e_dz2 < 8e-3 and e_w2 < 1.2e-2
    # masked tokens get an exact zero gradient
    assert float((dz.float() * (1 - mask.to(DEV).float()).unsqueeze(-1)).abs().max()) == 0.0


def test_parallel_nets_backward_without_z_cotangent_and_per_token_cotangent(M):
    """dz_out = NULL (the ELBO only takes the per-token values) and a genuine per-token cotangent (S = 1 segments)."""
    B, S, D, H, K = 8, 40, 64, 64, 4
    ps, flat, z0, mask, cz, cl = _case(B, S, D, H, K, "gelu", seed=5)
    for one_pass in (True, False):
        os.environ["MIVAE_NICE_BWD_PAR"] = "1" if one_pass else "0"
        try:
            gp = [t.detach().to(DEV).clone().requires_grad_(True) for t in flat]
            zg = z0.to(DEV).bfloat16().requires_grad_(True)
            _, tok = M.ops.nice_stack_tok(zg, mask.to(DEV), gp, H, "gelu", False, True, 0.0, 1.0)
            wt = torch.linspace(-1, 1, B * S, device=DEV).view(B, S)
            (tok * wt).sum().backward()
            res = (zg.grad.float(), [p.grad for p in gp])
        finally:
            os.environ.pop("MIVAE_NICE_BWD_PAR", None)
        if one_pass:
            first = res
    assert rel_err(first[0], res[0]) < 1.6e-2
    assert max(rel_err(a, b) for a, b in zip(first[1], res[1])) < 1e-2


def test_parallel_nets_backward_random_shapes(M):
    """Seeded sweep over small shapes (odd token counts, h and H that are not multiples of 32, K = 1 .. 4, every
    activation): the parallel-nets kernel against the task-serial one on the same inputs."""
    import random
    rnd = random.Random(20240607)
    acts = ["gelu", "relu", "tanh", "sigmoid", "swish"]
    worst = (0.0, 0.0)
    for it in range(24):
        B, S = rnd.randint(1, 12), rnd.randint(1, 70)
        D = rnd.choice([16, 32, 48, 64])
        H = rnd.choice([8, 16, 24, 40, 48, 64])
        K = rnd.randint(1, 4)
        act = rnd.choice(acts)
        ps, flat, z0, mask, cz, cl = _case(B, S, D, H, K, act, seed=1000 + it, holes=(it % 5 == 0))
        zK, lp, dz, grads = _run_gpu(M, flat, z0, mask, cz, cl, H, act, one_pass=True)
        zK2, lp2, dz2, grads2 = _run_gpu(M, flat, z0, mask, cz, cl, H, act, one_pass=False)
        assert torch.equal(zK, zK2) and torch.equal(lp, lp2)
        assert not torch.isnan(dz.float()).any() and all(not torch.isnan(g).any() for g in grads)
        e_dz = rel_err(dz.float(), dz2.float())
        e_w = max(rel_err(a, b) for a, b in zip(grads, grads2))
        worst = (max(worst[0], e_dz), max(worst[1], e_w))
        assert e_dz < 8e-3 and e_w < 1.2e-2, (B, S, D, H, K, act, e_dz, e_w)
        assert float((dz.float() * (1 - mask.to(DEV).float()).unsqueeze(-1)).abs().max()) == 0.0
    print(f"parallel-nets bwd, 24 random shapes: worst dz {worst[0]:.2e} dW {worst[1]:.2e} vs the task-serial kernel")
