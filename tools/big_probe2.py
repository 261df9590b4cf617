"""Debug probe: which hidden chunk of which net is mis-read (only chunk c0 of net n0 has non-zero W2)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nonauto_lm_amd as M
from oracle import latent_path as O
from test_nice_big import make_case
from conftest import rel_err
B, S, D, H, K = 2, 128, 256, 256, 2
ps, flat, z0, mask, cz, cl = make_case(B, S, D, H, K, "gelu", 3, False, True)
for k0 in range(K):
  for n0 in range(2):
    res = []
    for c0 in range(H // 32):
        fl = [t.clone() for t in flat]
        for k in range(K):
            for n in range(2):
                w2 = fl[8 * k + 4 * n + 2]
                keep = torch.zeros_like(w2)
                if k == k0 and n == n0:
                    keep[:, 32 * c0:32 * c0 + 32] = 1
                fl[8 * k + 4 * n + 2] = w2 * keep
        pp = [O.NiceParams([fl[8*k]], [fl[8*k+1]], fl[8*k+2], fl[8*k+3], [fl[8*k+4]], [fl[8*k+5]], fl[8*k+6], fl[8*k+7]) for k in range(K)]
        zr, _ = O.flow_prior_forward(z0, O.nice_steps(pp), mask, 0.0, 1.0)
        zg, _ = M.ops.nice_stack(z0.cuda().bfloat16(), mask.cuda(), [t.cuda() for t in fl], H, "gelu", base=True)
        res.append(round(rel_err(zg.float(), zr), 4))
    print("step", k0, "net", n0, "err by chunk:", res)
# detail for step 0, net 0, chunk 0
fl = [t.clone() for t in flat]
for k in range(K):
    for n in range(2):
        w2 = fl[8 * k + 4 * n + 2]
        keep = torch.zeros_like(w2)
        if k == 0 and n == 0:
            keep[:, 0:32] = 1
        fl[8 * k + 4 * n + 2] = w2 * keep
pp = [O.NiceParams([fl[8*k]], [fl[8*k+1]], fl[8*k+2], fl[8*k+3], [fl[8*k+4]], [fl[8*k+5]], fl[8*k+6], fl[8*k+7]) for k in range(K)]
zr, _ = O.flow_prior_forward(z0, O.nice_steps(pp), mask, 0.0, 1.0)
for it in range(3):
    zg, _ = M.ops.nice_stack(z0.cuda().bfloat16(), mask.cuda(), [t.cuda() for t in fl], H, "gelu", base=True)
    e = (zg.float().cpu() - zr).abs().reshape(-1, D)
    N = e.shape[0]
    print("run", it, "max err by wave, block 0..1:", e.reshape(N // 16, 16, D).amax(dim=(1, 2)).reshape(-1, 8)[:2].tolist())
    print("   by token in wave:", [round(float(v), 3) for v in e.reshape(N // 16, 16, D).amax(dim=(0, 2))])
    print("   by feature tile:", [round(float(v), 3) for v in e.reshape(N, D // 16, 16).amax(dim=(0, 2))])
