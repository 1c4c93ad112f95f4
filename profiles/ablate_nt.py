#!/usr/bin/env python3
"""Time vkas_conv_gemm_fwd for one conv shape with several builds of libvkas (build_variants/libvkas_*.so), interleaved
in one process (rounds x variants), HIP-event timed.  Used to A/B kernel variants and timing-only ablations.

usage: ablate_nt.py [--shape B,H,W,C,N,KH] [--rounds R] lib1.so lib2.so ..."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import _lib  # noqa: E402  (struct definitions + the shipped build)

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='8,512,512,384,384,3')
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--mode', type=int, default=0, help='0 plain store, 1 GELU dual store')
ap.add_argument('libs', nargs='+')
args = ap.parse_args()
B, H, W, C, N, KH = map(int, args.shape.split(','))
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((B, H, W, C), generator=g, device='cuda').bfloat16()
Bw = (torch.randn((N, KH * KH * C), generator=g, device='cuda') * 0.02).bfloat16()
out = torch.empty((B, H, W, N), device='cuda', dtype=torch.bfloat16)
geom = _lib.ConvGeom(B, H, W, H, W, C, C, KH, KH, 1, KH // 2)
epi = _lib.Epilogue()
epi.mode, epi.out, epi.ldo = args.mode, out.data_ptr(), N
out2 = torch.empty_like(out)
if args.mode == 1:
    epi.out2, epi.ldo2 = out2.data_ptr(), N
libs = []
for p in args.libs:
    L = ctypes.CDLL(os.path.abspath(p))
    L.vkas_conv_gemm_fwd.restype = ctypes.c_int
    L.vkas_conv_gemm_fwd.argtypes = _lib._SIGS['vkas_conv_gemm_fwd'][1]
    libs.append((os.path.basename(p), L))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flops = 2.0 * B * H * W * N * KH * KH * C
times = {n: [] for n, _ in libs}
for r in range(args.rounds + 1):
    for name, L in libs:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            rc = L.vkas_conv_gemm_fwd(x.data_ptr(), ctypes.byref(geom), Bw.data_ptr(), N, ctypes.byref(epi), _lib.BF16, st)
            assert rc == 0, rc
        e.record()
        torch.cuda.synchronize()
        if r > 0:
            times[name].append(s.elapsed_time(e) / args.iters)
for name, ts in times.items():
    ts.sort()
    med = ts[len(ts) // 2]
    print(f'{name:28s} median {med:8.3f} ms  min {ts[0]:8.3f} ms  {flops / med / 1e9:8.1f} TFLOP/s (median)', flush=True)
