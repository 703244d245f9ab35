#!/usr/bin/env python3
"""Diagnostic (not product): what this GPU's HBM delivers to plain torch kernels -- read-only, copy (1 read : 1 write), and the
lean sweep's mix (about 85 % reads) -- as the practical ceiling beside the 8 TB/s spec the roofline fraction is priced on."""
import torch
dev = torch.device('cuda:0')
n = 1 << 27          # 1 GiB of float64
x = torch.rand(n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


t = timed(lambda: x.sum())
print('read-only (sum of 1 GiB):        %.2f TB/s' % (n * 8 / t / 1e12))
t = timed(lambda: y.copy_(x))
print('copy 1 GiB (read + write):       %.2f TB/s of total traffic' % (2 * n * 8 / t / 1e12))
t = timed(lambda: y.fill_(1.0))
print('write-only (fill 1 GiB):         %.2f TB/s' % (n * 8 / t / 1e12))
m = n // 6
t = timed(lambda: torch.add(x[:m], x[m:2 * m], out=y[:m]))
print('2 reads : 1 write (add):         %.2f TB/s of total traffic' % (3 * m * 8 / t / 1e12))
