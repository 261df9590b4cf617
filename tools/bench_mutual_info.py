import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nonauto_lm_amd as M
B, S, D = 256, 64, 64
post = M.DefaultPosterior(D, D, samples=1).cuda()
enc = torch.randn(B, S, D, device="cuda")
mask = torch.ones(B, S, dtype=torch.long, device="cuda")
with torch.no_grad():
    latent, logq = post(enc, mask)
    def t(fn, it=10):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
    k = t(lambda: post.calc_mutual_info(latent.z, logq, mask))
    torch.cuda.reset_peak_memory_stats(); post.calc_mutual_info(latent.z, logq, mask); pk = torch.cuda.max_memory_allocated() / 2**20
with torch.enable_grad():
    z = latent.z.clone().requires_grad_(True)
    e = t(lambda: post.calc_mutual_info(z, logq, mask), 3)
    torch.cuda.reset_peak_memory_stats(); post.calc_mutual_info(z, logq, mask); pe = torch.cuda.max_memory_allocated() / 2**20
print(f"kernel route {k:.2f} ms peak {pk:.0f} MiB | broadcast route {e:.2f} ms peak {pe:.0f} MiB")
