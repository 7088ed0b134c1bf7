"""Device copy / fill / reduce rates of the box (torch kernels), for reading the roofline fractions: python tools/copy_rate.py"""
import torch
torch.cuda.init()
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    def timed(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    t_copy = timed(lambda: b.copy_(a))
    t_fill = timed(lambda: b.fill_(1.5))
    t_sum = timed(lambda: a.sum())
    t_axpy = timed(lambda: torch.add(a, b, alpha=2.0, out=b))
    print("%5d MB: copy %.2f TB/s (r+w)  fill %.2f TB/s (w)  sum %.2f TB/s (r)  axpy %.2f TB/s (2r+w)" % (
        mb, 2 * n * 8 / t_copy / 1e12, n * 8 / t_fill / 1e12, n * 8 / t_sum / 1e12, 3 * n * 8 / t_axpy / 1e12))
