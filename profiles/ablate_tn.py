#!/usr/bin/env python3
"""Time vkas_conv_gemm_wgrad for one conv shape with several builds of libvkas, interleaved in one process.

usage: ablate_tn.py [--shape B,H,W,C,N,KH] [--rounds R] lib1.so lib2.so ..."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='8,512,512,384,384,3')
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('libs', nargs='+')
args = ap.parse_args()
B, H, W, C, N, KH = map(int, args.shape.split(','))
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((B, H, W, C), generator=g, device='cuda').bfloat16()
dy = torch.randn((B, H, W, N), generator=g, device='cuda').bfloat16()
K = KH * KH * C
gw = torch.zeros((N * K + N,), device='cuda', dtype=torch.float32)
geom = _lib.ConvGeom(B, H, W, H, W, C, C, KH, KH, 1, KH // 2)
libs = []
for p in args.libs:
    L = ctypes.CDLL(os.path.abspath(p))
    L.vkas_conv_gemm_wgrad.restype = ctypes.c_int
    L.vkas_conv_gemm_wgrad.argtypes = _lib._SIGS['vkas_conv_gemm_wgrad'][1]
    libs.append((os.path.basename(p), L))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flops = 2.0 * B * H * W * N * K
times = {n: [] for n, _ in libs}
ref = None
for r in range(args.rounds + 1):
    for name, L in libs:
        gw.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            rc = L.vkas_conv_gemm_wgrad(x.data_ptr(), ctypes.byref(geom), dy.data_ptr(), N, N, gw.data_ptr(),
                                        gw.data_ptr() + 4 * N * K, _lib.BF16, st)
            assert rc == 0, rc
        e.record()
        torch.cuda.synchronize()
        if r == 0:
            if ref is None:
                ref = gw.clone()
            else:
                err = float((gw - ref).norm() / ref.norm())
                print(f'{name}: rel diff vs first lib {err:.3e}')
        else:
            times[name].append(s.elapsed_time(e) / args.iters)
for name, ts in times.items():
    ts.sort()
    med = ts[len(ts) // 2]
    print(f'{name:28s} median {med:8.3f} ms  min {ts[0]:8.3f} ms  {flops / med / 1e9:8.1f} TFLOP/s (median)', flush=True)
