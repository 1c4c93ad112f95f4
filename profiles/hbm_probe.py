#!/usr/bin/env python3
"""Achievable HBM rates on this box with plain ATen streaming kernels over 1 GiB buffers (write-only fill, read-only sum,
copy): the practical ceilings the write-heavy kernels (resize forward, GEMM epilogues) should be read against."""
import torch

n = 1 << 28
x = torch.empty(n, dtype=torch.float32, device='cuda')
y = torch.empty(n, dtype=torch.float32, device='cuda')


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


gb = n * 4 / 1e12
print(f'fill  (write 1 GiB)        {gb / timed(lambda: x.fill_(1.0)):.2f} TB/s')
print(f'sum   (read 1 GiB)         {gb / timed(lambda: x.sum()):.2f} TB/s')
print(f'copy  (read + write 1 GiB) {2 * gb / timed(lambda: y.copy_(x)):.2f} TB/s')
h = x.view(-1)[: n // 2].view(torch.bfloat16)
print(f'bf16 mul (read + write)    {2 * h.numel() * 2 / 1e12 / timed(lambda: h.mul_(1.0)):.2f} TB/s')
