#!/usr/bin/env python3
"""What the box's HBM delivers to plain streaming kernels (the practical ceiling behind the 8 TB/s spec the roofline is priced against):
device copies (read + write bytes), fills (write only) and sums (read only) of buffers far larger than the 256 MB Infinity Cache, and
of a 150 MB buffer -- the size of one layer's activations on the 18 k-node training batch -- right after it was written (cache-warm)."""
import torch

dev = torch.device("cuda:0")


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for mb in (150, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    tc = t_us(lambda: y.copy_(x))
    tf = t_us(lambda: y.fill_(1.0))
    ts = t_us(lambda: x.sum())
    by = n * 4
    print(f"{mb:5d} MB: copy {2 * by / tc / 1e6:6.2f} TB/s (r+w)   fill {by / tf / 1e6:6.2f} TB/s   sum {by / ts / 1e6:6.2f} TB/s")
