"""Practical HBM rates on this box (reference point for the memory-bound kernels): torch copy / read-only sum / fill."""
import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for mb in (64, 256, 1024):
    x = torch.empty(mb * 1024 * 1024 // 2, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    dt = t(lambda: y.copy_(x)); print("copy  %5d MB: %.1f us  %.2f TB/s (read+write)" % (mb, dt * 1e6, 2 * x.numel() * 2 / dt / 1e12))
    dt = t(lambda: y.fill_(1.0)); print("fill  %5d MB: %.1f us  %.2f TB/s (write)" % (mb, dt * 1e6, x.numel() * 2 / dt / 1e12))
    xf = x.view(torch.int16)
    dt = t(lambda: xf.sum()); print("sum   %5d MB: %.1f us  %.2f TB/s (read)" % (mb, dt * 1e6, x.numel() * 2 / dt / 1e12))
    dt = t(lambda: torch.add(x, y, out=y)); print("add   %5d MB: %.1f us  %.2f TB/s (2 reads + write)" % (mb, dt * 1e6, 3 * x.numel() * 2 / dt / 1e12))
