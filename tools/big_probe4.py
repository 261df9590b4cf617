import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nonauto_lm_amd as M
from nonauto_lm_amd import ops, _lib as L
from test_nice_big import make_case
B, S, D, H, K = 2, 128, 256, 256, 2
ps, flat, z0, mask, cz, cl = make_case(B, S, D, H, K, "gelu", 3, False, True)
gp = [t.cuda() for t in flat]
z = z0.cuda().bfloat16()
h = D // 2
pk = ops.nice_big_pack(gp, K, h, H, "fwd")
pb = ops.nice_big_pack(gp, K, h, H, "bias")
z_out = torch.zeros_like(z); tok = torch.zeros(B * S, device="cuda")
L.call("mivae_nice_big_fwd", L.ptr(z), L.ptr(z_out), None, L.ptr(mask.cuda()), L.ptr(pk), L.ptr(pb), L.ptr(tok), B, 1, S, D, H, K, 1, 0, 1, 0.0, 1.0, L.stream())
torch.cuda.synchronize()
raw = z_out.view(torch.uint8).reshape(-1)[:2 * 8192 + 4096].cpu()
ref = pk[:2 * 8192].cpu()
d = (raw[:2 * 8192] != ref).reshape(16, 1024).sum(1)
print("mismatching bytes per 1 KiB piece of items 0,1:", d.tolist())
db = (raw[2 * 8192:] != pb.view(torch.uint8).reshape(-1)[:4096].cpu()).reshape(4, 1024).sum(1)
print("mismatching bytes per 1 KiB piece of bias 0:", db.tolist())
