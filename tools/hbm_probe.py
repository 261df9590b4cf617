"""Reference points for the HBM-bound kernels: device fill (write-only), copy (read + write) and sum (read-only)
bandwidth of a 256 MiB buffer on this GPU, timed with HIP events (stock PyTorch kernels)."""
import torch
n = 128 * 1024 * 1024
a = torch.empty(n, dtype=torch.bfloat16, device="cuda")
b = torch.empty_like(a)


def timed(fn, bytes_moved, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    return round(us, 1), round(bytes_moved / us / 1e6, 2)


print("fill  256MiB (write)      us, TB/s:", timed(lambda: a.zero_(), 2 * n))
print("copy  256MiB (read+write) us, TB/s:", timed(lambda: b.copy_(a), 4 * n))
print("sum   256MiB (read)       us, TB/s:", timed(lambda: a.view(torch.int16).max(), 2 * n))
