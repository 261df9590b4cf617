"""Pins the CPU oracle (oracle/latent_path.py) against golden vectors recorded from the real
reference (oracle/refgen/gen_golden.py).  CPU only; tolerance fp32 <= 1e-6 relative (max-norm)."""
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import latent_path as O

TOL = 2e-6

NICE_CASES = ["nice_tiny_full", "nice_tiny_ragged", "nice_tiny_holes", "nice_d64_ragged",
              "nice_d64_full", "nice_d128_h96_holes", "nice_relu_2layer"]


def nice_params(g, prefix="sd._flows."):
    act = str(g["activation"]) if "activation" in g else "gelu"
    ps, k = [], 0
    while f"{prefix}{k}._s_net._feedforward._output.weight" in g:
        ps.append(O.NiceParams.from_state_dict(g, f"{prefix}{k}.", activation=act))
        k += 1
    return ps


@pytest.mark.parametrize("name", NICE_CASES)
def test_nice_forward_steps_and_logp(name):
    g = load_golden(name)
    ps = nice_params(g)
    z, mask = g["z0"], g["mask"]
    for k, p in enumerate(ps):
        z, ld = O.nice_step(z, p, mask)
        assert rel_err(z, g[f"fwd.z{k}"]) < TOL
        assert rel_err(ld, g[f"fwd.ld{k}"]) < TOL
    zK, logp = O.flow_prior_forward(g["z0"], O.nice_steps(ps), mask)
    assert rel_err(zK, g["fwd.zK"]) < TOL
    assert rel_err(logp, g["fwd.logp"]) < TOL


@pytest.mark.parametrize("name", NICE_CASES)
def test_nice_inverse_and_roundtrip(name):
    g = load_golden(name)
    ps = nice_params(g)
    mask = g["mask"]
    z = g["inv.eps"]
    for k, p in enumerate(reversed(ps)):
        z, ld = O.nice_step(z, p, mask.float(), inverse=True)
        assert rel_err(z, g[f"inv.z{k}"]) < TOL
        assert rel_err(ld, g[f"inv.ld{k}"]) < TOL
    zi, lpi = O.flow_prior_inverse(g["inv.eps"], O.nice_steps(ps), mask.float())
    assert rel_err(zi, g["inv.zK"]) < TOL
    assert rel_err(lpi, g["inv.logp"]) < TOL
    back, _ = O.flow_prior_inverse(g["fwd.zK"], O.nice_steps(ps), mask.float())
    assert rel_err(back, g["inv.roundtrip"]) < 1e-5


@pytest.mark.parametrize("name", NICE_CASES)
def test_nice_gradients(name):
    g = load_golden(name)
    ps = nice_params(g)
    z0 = g["z0"].clone().requires_grad_(True)
    for p in ps:
        for t in p.tensors():
            t.requires_grad_(True)
    zK, logp = O.flow_prior_forward(z0, O.nice_steps(ps), g["mask"])
    ((zK * g["cot.z"]).sum() + (logp * g["cot.logp"]).sum()).backward()
    assert rel_err(z0.grad, g["grad.z0"]) < 1e-5
    for k, p in enumerate(ps):
        pre = f"grad.sd._flows.{k}."
        assert rel_err(p.s_w[0].grad, g[pre + "_s_net._feedforward._linear_layers.0.weight"]) < 1e-5
        assert rel_err(p.t_wo.grad, g[pre + "_t_net._feedforward._output.weight"]) < 1e-5
        assert rel_err(p.s_bo.grad, g[pre + "_s_net._feedforward._output.bias"]) < 1e-5


@pytest.mark.parametrize("name", ["nice_auto_2d", "nice_auto_d64"])
def test_nice_auto_unmasked(name):
    g = load_golden(name)
    ps = nice_params(g)
    mu, std = float(g["mu"]), float(g["std"])
    z = g["z0"]
    for k, p in enumerate(ps):
        z, ld = O.nice_step(z, p, None)
        assert rel_err(z, g[f"fwd.z{k}"]) < TOL and rel_err(ld, g[f"fwd.ld{k}"]) < TOL
    zK, logp = O.flow_prior_forward(g["z0"], O.nice_steps(ps), None, mu, std)
    assert rel_err(logp, g["fwd.logp"]) < TOL
    zi, lpi = O.flow_prior_inverse(g["inv.eps"], O.nice_steps(ps), None, mu, std)
    assert rel_err(zi, g["inv.zK"]) < TOL and rel_err(lpi, g["inv.logp"]) < TOL


@pytest.mark.parametrize("name", ["actnorm_3d_ragged", "actnorm_3d_d64", "actnorm_2d"])
def test_actnorm(name):
    g = load_golden(name)
    mask = g.get("mask")
    ls, b = O.actnorm_init(g["z0"], g["pre.log_scale"], g["pre.bias"])
    assert rel_err(ls, g["post.log_scale"]) < TOL and rel_err(b, g["post.bias"]) < TOL
    ls = g["post.log_scale"].clone().requires_grad_(True)
    b = g["post.bias"].clone().requires_grad_(True)
    z0 = g["z0"].clone().requires_grad_(True)
    z, ld = O.actnorm_step(z0, ls, b, mask)
    assert z.shape == g["fwd.z"].shape and ld.shape == g["fwd.ld"].shape
    assert rel_err(z, g["fwd.z"]) < TOL and rel_err(ld, g["fwd.ld"]) < TOL
    ((z * g["cot.z"]).sum() + (ld * g["cot.ld"]).sum()).backward()
    assert rel_err(z0.grad, g["grad.z0"]) < 1e-5
    assert rel_err(ls.grad, g["grad.log_scale"]) < 1e-5 and rel_err(b.grad, g["grad.bias"]) < 1e-5
    zi, ldi = O.actnorm_step(g["fwd.z"], g["post.log_scale"], g["post.bias"], mask, inverse=True)
    assert rel_err(zi, g["inv.z"]) < TOL and rel_err(ldi, g["inv.ld"]) < TOL


@pytest.mark.parametrize("name", ["invlinear_d8", "invlinear_d64", "invmultihead_d64_h4"])
def test_invlinear(name):
    g = load_golden(name)
    heads = int(g["heads"])
    w = g["weight"].clone().requires_grad_(True)
    z0 = g["z0"].clone().requires_grad_(True)
    z, ld = O.invlinear_step(z0, w, g["mask"], num_heads=heads)
    assert rel_err(z, g["fwd.z"]) < TOL and rel_err(ld, g["fwd.ld"]) < 1e-5
    ((z * g["cot.z"]).sum() + (ld * g["cot.ld"]).sum()).backward()
    assert rel_err(z0.grad, g["grad.z0"]) < 1e-5 and rel_err(w.grad, g["grad.weight"]) < 1e-5
    zi, ldi = O.invlinear_step(g["fwd.z"], g["weight"], g["mask"], inverse=True, num_heads=heads)
    assert rel_err(zi, g["inv.z"]) < 1e-5 and rel_err(ldi, g["inv.ld"]) < 1e-5


def mixed_steps(g, prefix="sd."):
    acts = lambda k: (lambda z, m, inv: O.actnorm_step(z, g[f"{prefix}_flows.{k}._log_scale"],
                                                        g[f"{prefix}_flows.{k}._bias"], m, inv))
    nice = lambda k: (lambda z, m, inv: O.nice_step(
        z, O.NiceParams.from_state_dict(g, f"{prefix}_flows.{k}."), m, inv))
    lin = lambda k: (lambda z, m, inv: O.invlinear_step(z, g[f"{prefix}_flows.{k}._weight"], m, inv))
    return [acts(0), nice(1), lin(2), nice(3), acts(4)]


def test_mixed_stack():
    g = load_golden("mixed_stack_d16")
    # data-dependent init of both actnorm layers, replayed in stack order (actnorm.py:22-28)
    z = g["z0"]
    ls0, b0 = O.actnorm_init(z, g["pre.sd._flows.0._log_scale"], g["pre.sd._flows.0._bias"])
    assert rel_err(ls0, g["sd._flows.0._log_scale"]) < TOL and rel_err(b0, g["sd._flows.0._bias"]) < TOL
    steps = mixed_steps(g)
    for st in steps[:4]:
        z, _ = st(z, g["mask"], False)
    ls4, b4 = O.actnorm_init(z, g["pre.sd._flows.4._log_scale"], g["pre.sd._flows.4._bias"])
    assert rel_err(ls4, g["sd._flows.4._log_scale"]) < 1e-5 and rel_err(b4, g["sd._flows.4._bias"]) < 1e-5
    zK, logp = O.flow_prior_forward(g["z0"], steps, g["mask"])
    assert rel_err(zK, g["fwd.zK"]) < 1e-5 and rel_err(logp, g["fwd.logp"]) < 1e-5
    zi, lpi = O.flow_prior_inverse(g["fwd.zK"], steps, g["mask"])
    assert rel_err(zi, g["inv.zK"]) < 1e-5 and rel_err(lpi, g["inv.logp"]) < 1e-5


@pytest.mark.parametrize("name", ["posterior_n1_ragged", "posterior_n3_holes", "posterior_d64_n2",
                                  "posterior_auto_n2"])
def test_posterior(name):
    g = load_golden(name)
    n = int(g["samples"])
    enc = g["enc"].clone().requires_grad_(True)
    wm, bm = g["sd._mu_net.weight"], g["sd._mu_net.bias"]
    ws, bs = g["sd._sigma_net.0.weight"], g["sd._sigma_net.0.bias"]
    mu = enc @ wm.t() + bm
    sigma = O.sigma_head(enc @ ws.t() + bs)
    mu.retain_grad(), sigma.retain_grad()
    assert rel_err(mu, g["mu"]) < TOL and rel_err(sigma, g["sigma"]) < TOL
    z, mrep = O.posterior_sample(mu, sigma, g["eps"], g.get("mask"), n)
    logq = O.posterior_log_prob(z, mu, sigma, mrep, n)
    assert rel_err(z, g["z"]) < TOL and rel_err(logq, g["logq"]) < TOL
    ((z * g["cot.z"]).sum() + (logq * g["cot.logq"]).sum()).backward()
    assert rel_err(enc.grad, g["grad.enc"]) < 1e-5
    assert rel_err(mu.grad, g["grad.mu"]) < 1e-4      # reference value is a rounding-noise cancellation
    assert rel_err(sigma.grad, g["grad.sigma"]) < 1e-5
    z0, _ = O.posterior_sample(mu, sigma, torch.zeros_like(g["eps"]), g.get("mask"), n)
    assert rel_err(z0, g["det.z"]) < TOL
    assert rel_err(O.posterior_log_prob(z0, mu, sigma, mrep, n), g["det.logq"]) < TOL


def test_kl_and_nll():
    g = load_golden("kl_nll")
    assert torch.equal(O.kl_default(g["logq"], g["logp"]), g["kl.none"])
    assert torch.equal(O.kl_default(g["logq"], g["logp"], 0.5), g["kl.fb0.5"])
    assert torch.equal(O.kl_default(g["logq"], g["logp"], 3.0), g["kl.fb3"])
    assert torch.equal(O.tokens_mask(g["nll.target"]).float(), g["nll.weights"])
    for sm in (0.0, 0.1):
        for mode in ("train", "eval"):
            got = O.label_smoothing_nll(g["nll.logits"], g["nll.target"], g["nll.weights"], sm,
                                        training=(mode == "train"))
            assert rel_err(got, g[f"nll.{mode}.s{sm}"]) < TOL


@pytest.mark.parametrize("name", ["model_n1", "model_n3_iwae_fb"])
def test_full_model_latent_path_and_elbo(name):
    """Replays the hot path inside a full reference NonAutoModel.forward from its recorded
    encoder output / logits: posterior -> flow prior -> KL -> ELBO tail (model.py:136-193)."""
    g = load_golden(name)
    n, iwae = int(g["samples"]), bool(int(g["iwae"]))
    fb = float(g["free_bits"])
    fb = None if fb < 0 else fb
    tokens = g["tokens"]
    mask = O.tokens_mask(tokens)
    B = tokens.size(0)
    enc = g["rec.enc"]
    mu = enc @ g["sd._posterior._mu_net.weight"].t() + g["sd._posterior._mu_net.bias"]
    sigma = O.sigma_head(enc @ g["sd._posterior._sigma_net.0.weight"].t() + g["sd._posterior._sigma_net.0.bias"])
    z, mrep = O.posterior_sample(mu, sigma, g["eps"], mask, n)
    logq = O.posterior_log_prob(z, mu, sigma, mrep, n)
    assert rel_err(z, g["rec.z"]) < TOL and rel_err(logq, g["rec.logq"]) < TOL
    ps = nice_params(g, "sd._prior._flows.")
    assert len(ps) == int(g["K"])
    _, logp = O.flow_prior_forward(z, O.nice_steps(ps), mrep)
    assert rel_err(logp, g["rec.logp"]) < TOL
    tgt = tokens.repeat_interleave(n, 0)
    recon = O.label_smoothing_nll(g["rec.logits"].reshape(B * n, *g["rec.logits"].shape[2:]), tgt,
                                  O.tokens_mask(tgt).float(), float(g["label_smoothing"]), True)
    kl = O.kl_default(logq, logp, fb)
    out = O.elbo_tail(recon, kl, B, n, float(g["recon_weight"]), float(g["kl_weight"]), iwae)
    for k in ("batch-loss", "batch-kl", "batch-nll"):
        assert rel_err(out[k], g["info." + k]) < 1e-5, k


def test_infovae_kl():
    """a19: oracle restatement of InfoVAEKLLoss vs the reference's recorded outputs (prior draw replayed)."""
    g = load_golden("infovae")
    for tag in ("3d", "2d"):
        for kernel in ("rbf", "imq"):
            for fb in (None, 0.4):
                key = f"{tag}.{kernel}.fb{fb}"
                z = g[f"{tag}.z"].clone().requires_grad_(True)
                out = O.infovae_kl(g[f"{tag}.logq"], g[f"{tag}.logp"], z, g[key + ".prior_z"], float(g[f"{tag}.std"]),
                                   0.3, 50, kernel, fb)
                for k in ("loss", "mmd-part", "kl-part"):
                    assert rel_err(out[k], g[f"{key}.{k}"]) < TOL, (key, k)
                gz, = torch.autograd.grad(out["loss"].sum(), z)
                assert rel_err(gz, g[key + ".grad_z"]) < 1e-5, key


@pytest.mark.parametrize("name", ["model_auto_n1", "model_auto_n2_iwae"])
def test_auto_model_latent_path_and_elbo(name):
    """C4: the hot path inside a full reference AutoModel.forward (sentence-level (B, D) latent) replayed from its
    recorded encoder context / logits: posterior -> flow prior -> KL -> NLL (without <sos>) -> ELBO tail
    (auto_lm/model.py:71-121, base/model.py:136-193)."""
    g = load_golden(name)
    n, iwae = int(g["samples"]), bool(int(g["iwae"]))
    fb = float(g["free_bits"])
    fb = None if fb < 0 else fb
    tokens = g["tokens"]
    B = tokens.size(0)
    ctx = g["rec.ctx"]
    mu = ctx @ g["sd._posterior._mu_net.weight"].t() + g["sd._posterior._mu_net.bias"]
    sigma = O.sigma_head(ctx @ g["sd._posterior._sigma_net.0.weight"].t() + g["sd._posterior._sigma_net.0.bias"])
    z, _ = O.posterior_sample(mu, sigma, g["eps"], None, n)
    logq = O.posterior_log_prob(z, mu, sigma, None, n)
    assert rel_err(z, g["rec.z"]) < TOL and rel_err(logq, g["rec.logq"]) < TOL
    ps = nice_params(g, "sd._prior._flows.")
    _, logp = O.flow_prior_forward(z, O.nice_steps(ps), None)
    assert rel_err(logp, g["rec.logp"]) < TOL
    tgt = tokens.repeat_interleave(n, 0)
    logits = g["rec.logits"].reshape(B * n, *g["rec.logits"].shape[2:])
    recon = O.label_smoothing_nll(logits, tgt[:, 1:], O.tokens_mask(tgt).float()[:, 1:], float(g["label_smoothing"]), True)
    out = O.elbo_tail(recon, O.kl_default(logq, logp, fb), B, n, float(g["recon_weight"]), float(g["kl_weight"]), iwae)
    for k in ("batch-loss", "batch-kl", "batch-nll"):
        assert rel_err(out[k], g["info." + k]) < 1e-5, k


def test_integer_helpers_bit_exact():
    g = load_golden("model_n1")
    tokens = g["tokens"]
    m = O.tokens_mask(tokens)
    assert m.dtype == torch.int64 and torch.equal(m, (tokens != 0).long())
    idx = O.last_token_index(m)
    assert torch.equal(idx, m.sum(1) - 1)
    pm = O.prefix_mask([3, 1, 5], 3)
    assert pm.tolist() == [[1, 1, 1, 0, 0], [1, 0, 0, 0, 0], [1, 1, 1, 1, 1]]
    assert O.prefix_mask([2], 3).shape == (3, 2)


# ------------------------------------------------------------------------------------- MAF / IAF / MADE (8f-2)
MAF_CASES = ["maf_tiny_ragged", "maf_d16_natural_holes", "maf_d64_2layer", "maf_auto_2d", "iaf_tiny_ragged",
             "iaf_d16_natural"]


def made_fns(g, requires_grad=False):
    """One oracle MADE closure per recorded flow + the tensors it uses (state-dict layout of made.py:72-83)."""
    act = str(g["activation"])
    fns, params = [], []
    for k in range(int(g["K"])):
        ws, bs, ms, i = [], [], [], 0
        while f"sd.{k}._made._nets.{2 * i}.weight" in g:
            ws.append(g[f"sd.{k}._made._nets.{2 * i}.weight"].clone().requires_grad_(requires_grad))
            bs.append(g[f"sd.{k}._made._nets.{2 * i}.bias"].clone().requires_grad_(requires_grad))
            ms.append(g[f"sd.{k}._made._nets.{2 * i}.weight_mask"])
            i += 1
        fns.append(lambda x, ws=ws, bs=bs, ms=ms: O.made_forward(x, ws, bs, ms, act))
        params.append((ws, bs))
    return fns, params


@pytest.mark.parametrize("name", MAF_CASES)
def test_made_masks_match_recorded_buffers(name):
    g = load_golden(name)
    D = g["z0"].shape[-1]
    masks = O.made_masks(D, [int(h) for h in g["hidden"]], bool(int(g["permute"])))
    for i, m in enumerate(masks):
        assert torch.equal(m.t(), g[f"sd.0._made._nets.{2 * i}.weight_mask"])      # bit exact (made.py:37)


@pytest.mark.parametrize("name", MAF_CASES)
def test_maf_iaf_steps_and_gradients(name):
    g = load_golden(name)
    fns, params = made_fns(g, requires_grad=True)
    iaf = str(g["kind"]) == "iaf"
    mask = g["mask"] if "mask" in g else None
    z0 = g["z0"].clone().requires_grad_(True)
    z, total = z0, 0.0
    for k, fn in enumerate(fns):
        step = O.maf_sequential if iaf else O.maf_forward
        z, ld = step(z, fn, bool(k % 2), mask)
        assert rel_err(z, g[f"fwd.z{k}"]) < TOL
        assert rel_err(ld, g[f"fwd.ld{k}"]) < TOL
        total = total + ld
    ((z * g["cot.z"]).sum() + (total * g["cot.ld"]).sum()).backward()
    assert rel_err(z0.grad, g["grad.z0"]) < 1e-5
    for k, (ws, bs) in enumerate(params):
        for i, (w, b) in enumerate(zip(ws, bs)):
            assert rel_err(w.grad, g[f"grad.sd.{k}._made._nets.{2 * i}.weight"]) < 1e-5
            assert rel_err(b.grad, g[f"grad.sd.{k}._made._nets.{2 * i}.bias"]) < 1e-5


@pytest.mark.parametrize("name", [n for n in MAF_CASES if n != "maf_auto_2d"])
def test_maf_iaf_reverse_direction(name):
    g = load_golden(name)
    fns, _ = made_fns(g)
    iaf = str(g["kind"]) == "iaf"
    mask = g["mask"].float() if "mask" in g else None
    K = len(fns)
    z = g["inv.eps"]
    for k, fn in enumerate(reversed(fns)):
        step = O.maf_forward if iaf else O.maf_sequential
        z, ld = step(z, fn, bool((K - 1 - k) % 2), mask)
        assert rel_err(z, g[f"inv.z{k}"]) < TOL
        assert rel_err(ld, g[f"inv.ld{k}"]) < TOL


# ----------------------------------------------------------------------------- prior sampling / mutual information
PRIOR_SAMPLE_CASES = ["prior_sample_default", "prior_sample_one_length", "prior_sample_flow"]


def _recover_noise(g):
    """The recorded base sample is already masked; at real positions it IS the draw, elsewhere anything works."""
    B, n = int(g["batch"]), int(g["samples"])
    eps = g["eps"]
    return eps.reshape(B, n, *eps.shape[1:])


@pytest.mark.parametrize("name", PRIOR_SAMPLE_CASES)
def test_prior_sample_mask_and_logprob(name):
    """DefaultPrior.sample / FlowPrior.sample (prior.py:105-131, flow_prior.py:15-24) -- the prefix
    mask from lengths is bit-exact, base / flow log-probs and the inverted sample match the recorded reference."""
    g = load_golden(name)
    B, n = int(g["batch"]), int(g["samples"])
    lengths = g["lengths"].tolist()
    mu, std = float(g["mu"]), float(g["std"])
    eps, base_lp, mask = O.prior_sample_from_noise(_recover_noise(g), lengths, B, n, None, mu, std)
    assert torch.equal(mask, g["mask"])
    assert torch.equal(eps, g["eps"])
    assert rel_err(base_lp, g["base.logp"]) < TOL
    ps = nice_params(g)
    z, lp, _ = O.prior_sample_from_noise(_recover_noise(g), lengths, B, n, O.nice_steps(ps) if ps else None, mu, std)
    assert rel_err(z, g["z"]) < TOL and rel_err(lp, g["logp"]) < TOL
    assert float((z * (1 - mask).unsqueeze(-1)).abs().max()) == 0.0


@pytest.mark.parametrize("name", ["mutual_info_n1", "mutual_info_n3"])
def test_mutual_info(name):
    g = load_golden(name)
    n = int(g["samples"])
    mask = g["mask"].repeat_interleave(n, 0)
    mi = O.mutual_info(g["z"], g["logq"], mask, g["mu"], g["sigma"], n)
    assert rel_err(mi, g["mi"]) < TOL
