#!/usr/bin/env python3
"""Microbenchmark + check of the depthwise 7x7 kernels (csrc/dwconv.hip) at the four stage shapes of config #3
(batch 16 = merged rough + precise pass).  VKAS_DW_OLD=1 selects the round-1 kernels for an A/B run.
The check compares with torch's grouped conv2d in fp32 on the GPU (development aid only, not a parity test)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vkit_ocr_model_adaptive_scaling_amd import ops  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd._lib import lib, check  # noqa: E402


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    p, st = ops._p, ops._stream
    quick = '--quick' in sys.argv  # stage-0 forward / dgrad timing only (ablation builds: results are wrong by design)
    shapes = ((16, 256, 256, 96),) if quick else ((16, 256, 256, 96), (16, 128, 128, 192), (16, 64, 64, 384), (16, 32, 32, 768),
                                                  (2, 37, 45, 40))
    for (B, H, W, C) in shapes:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).cuda()
        dy = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).cuda()
        w = (torch.randn(C, 1, 7, 7, generator=g) * 0.1).cuda()
        bias = (torch.randn(C, generator=g) * 0.1).cuda()
        wp = ops.pack_dw_weight(w, C, C, 0)
        y = torch.empty_like(x)
        M = B * H * W
        fwd = lambda: check(lib.vkas_dwconv7x7_fwd(p(x), C, p(wp), p(bias), None, 0, p(y), C, B, H, W, C, 1, st()), 'fwd')
        ms = timeit(fwd)
        if quick:
            print(f'{os.environ.get("VKAS_LIB_PATH", "libvkas.so"):40s} fwd {ms * 1e3:7.1f} us', flush=True)
            if '--trace' in sys.argv:  # -DDW_TRACE build: where a workgroup's tile iterations spend their cycles
                import numpy as np
                raw = ctypes.CDLL(os.environ['VKAS_LIB_PATH'])
                buf = np.zeros(8192 * 8, dtype=np.uint64)
                assert raw.vkas_dw_trace_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
                t = buf.reshape(-1, 8).astype(np.float64)
                t = t[t[:, 6] > 0]
                per = t[:, :6] / t[:, 6:7]
                names = ('fetch issue', 'matrix products (+ lgkmcnt)', 'barrier 1', 'result planes -> y', 'staging (+ lgkmcnt)',
                         'barrier 2')
                print(f'  {len(t)} workgroups, {np.median(t[:, 6]):.0f} tiles each; median cycles per tile:')
                for i, n in enumerate(names):
                    print(f'    {n:30s} {np.median(per[:, i]):8.0f}')
                print(f'    {"sum":30s} {np.median(per.sum(1)):8.0f}')
            continue
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, bias, padding=3, groups=C).permute(0, 2, 3, 1)
        err = float((y.float() - ref).norm() / ref.norm())
        print(f'B={B} {H}x{W} C={C}: fwd {ms * 1e3:7.1f} us  {2 * M * C * 2 / ms / 1e9:5.2f} TB/s  rel err {err:.2e}', flush=True)
        # dgrad form: flipped taps + residual addend
        wf = ops.pack_dw_weight(w, C, C, 1)
        dx = torch.empty_like(x)
        dg = lambda: check(lib.vkas_dwconv7x7_fwd(p(dy), C, p(wf), None, p(x), C, p(dx), C, B, H, W, C, 1, st()), 'dgrad')
        ms = timeit(dg)
        ref = F.conv2d(dy.float().permute(0, 3, 1, 2), w.flip(2, 3), None, padding=3, groups=C).permute(0, 2, 3, 1) + x.float()
        err = float((dx.float() - ref).norm() / ref.norm())
        print(f'      dgrad+res {ms * 1e3:7.1f} us  {3 * M * C * 2 / ms / 1e9:5.2f} TB/s  rel err {err:.2e}', flush=True)
        nb = lib.vkas_dwconv7x7_wgrad_ws_bytes(B, H, W, C)
        ws = torch.empty(nb // 4 + 4, device='cuda')
        gwb = torch.empty(50 * C, device='cuda')
        wg = lambda: check(lib.vkas_dwconv7x7_wgrad(p(x), C, p(dy), C, p(gwb[:49 * C]), p(gwb[49 * C:]), p(ws), nb, B, H, W, C,
                                                    1, st()), 'wgrad')
        ms = timeit(wg)
        xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(False)
        wr = w.clone().requires_grad_(True)
        F.conv2d(xr, wr, None, padding=3, groups=C).backward(dy.float().permute(0, 3, 1, 2).contiguous())
        got = gwb[:49 * C].view(49, C).t().reshape(C, 1, 7, 7)
        err = float((got - wr.grad).norm() / wr.grad.norm())
        berr = float((gwb[49 * C:] - dy.float().sum((0, 1, 2))).norm() / dy.float().sum((0, 1, 2)).norm())
        print(f'      wgrad     {ms * 1e3:7.1f} us  {2 * M * C * 2 / ms / 1e9:5.2f} TB/s  rel err {err:.2e} bias {berr:.2e}', flush=True)


if __name__ == '__main__':
    main()
