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
def run(fl, tag):
    pp = [O.NiceParams([fl[8*k]], [fl[8*k+1]], fl[8*k+2], fl[8*k+3], [fl[8*k+4]], [fl[8*k+5]], fl[8*k+6], fl[8*k+7]) for k in range(K)]
    zr, _ = O.flow_prior_forward(z0, O.nice_steps(pp), mask, 0.0, 1.0)
    zg, _ = M.ops.nice_stack(z0.cuda().bfloat16(), mask.cuda(), [t.cuda() for t in fl], H, "gelu", base=True)
    print(tag, rel_err(zg.float(), zr))
def only_chunk0(fl):
    for k in range(K):
        for n in range(2):
            keep = torch.zeros_like(fl[8*k+4*n+2])
            if k == 0 and n == 0: keep[:, 0:32] = 1
            fl[8*k+4*n+2] = fl[8*k+4*n+2] * keep
    return fl
fl = only_chunk0([t.clone() for t in flat]); run(fl, "chunk0 only:")
fl = only_chunk0([t.clone() for t in flat]); fl[0] = fl[0] * 0; run(fl, "chunk0 only, W1s = 0 (hid = act(b1)):")
fl = only_chunk0([t.clone() for t in flat]); fl[1] = fl[1] * 0; run(fl, "chunk0 only, b1s = 0:")
fl = only_chunk0([t.clone() for t in flat]); fl[3] = fl[3] * 0; run(fl, "chunk0 only, b2s = 0:")
fl = only_chunk0([t.clone() for t in flat]); fl[1] = fl[1] * 0; fl[3] = fl[3] * 0; run(fl, "chunk0 only, b1s = b2s = 0:")
for ks in range(4):
    for u in range(2):
        fl = only_chunk0([t.clone() for t in flat])
        keep = torch.zeros_like(fl[0]); keep[16*u:16*u+16, 32*ks:32*ks+32] = 1
        fl[0] = fl[0] * keep
        run(fl, f"chunk0, W1s rows {16*u}..{16*u+15}, cols {32*ks}..{32*ks+31} only:")
