import sys, torch
sys.path.insert(0, '/root/repo')
import nonauto_lm_amd as M
from nonauto_lm_amd import ops
B,S,D,H,K=256,128,256,256,16
dev='cuda'
torch.manual_seed(0)
h=D//2
params=[]
for _ in range(K):
    for _n in range(2):
        params += [torch.randn(H,h,device=dev)*(0.5/h**0.5), torch.randn(H,device=dev)*0.1, torch.randn(h,H,device=dev)*(0.5/H**0.5), torch.randn(h,device=dev)*0.1]
z=torch.randn(B,S,D,device=dev).bfloat16()
mask=torch.ones(B,S,dtype=torch.long,device=dev)
with torch.no_grad():
    for _ in range(3):
        zo,tok=ops._NiceStackBig.apply(z,mask,H,1,False,True,0.0,1.0,*params)
torch.cuda.synchronize()
print("cycles/body enter, body, fold/net:", tok[-4:-1].tolist())
