"""Debug probe: wide-latent NICE stack forward vs the fp32 oracle, repeated (determinism + error level)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nonauto_lm_amd as M
from oracle import latent_path as O
from test_nice_big import make_case
from conftest import rel_err
B, S, D, H, K = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (4, 128, 256, 256, 16))]
ps, flat, z0, mask, cz, cl = make_case(B, S, D, H, K, "gelu", 100 + D + K, False, True)
zK_r, lp_r = O.flow_prior_forward(z0, O.nice_steps(ps), mask, 0.0, 1.0)
gp = [t.cuda() for t in flat]
outs = []
for it in range(4):
    zK, lp = M.ops.nice_stack(z0.cuda().bfloat16(), mask.cuda(), gp, H, "gelu", base=True)
    outs.append((zK.clone(), lp.clone()))
    print(it, "z err", rel_err(zK.float(), zK_r), "lp err", rel_err(lp, lp_r), "same as run 0:", torch.equal(zK, outs[0][0]), torch.equal(lp, outs[0][1]))
d = (outs[0][0].float().cpu() - zK_r).abs()
print("max abs err", float(d.max()), "at", [int(v) for v in torch.nonzero(d == d.max())[0]], "ref max", float(zK_r.abs().max()))
e = (outs[0][0].float().cpu() - zK_r).abs().reshape(-1, D)
N = e.shape[0]
byw = e.reshape(N // 16, 16, D).amax(dim=(1, 2)).reshape(-1, 8)
print("max err by wave (rows = blocks, first 4 blocks):\n", byw[:4])
print("max err by feature tile of 16:", e.reshape(N, D // 16, 16).amax(dim=(0, 2)))
print("max err by token within wave:", e.reshape(N // 16, 16, D).amax(dim=(0, 2)))
