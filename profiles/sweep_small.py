#!/usr/bin/env python3
"""Time the pointwise GEMM shapes of the forward-only configurations (BASELINE.json configs[1] and [4]: M = 400 ... 25 600
rows) through the C ABI, many launches between two events so that the launch floor is included but the event cost is not.
usage: python profiles/sweep_small.py [VKAS_NT_TILE is read by the library]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import _lib  # noqa: E402

lib = _lib.lib
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device='cuda').manual_seed(0)
# (M, N, K, epilogue): 1 = bias + GELU with two outputs, 2 = bias + layer scale + residual
SHAPES = [(6400, 1536, 384, 1), (6400, 384, 1536, 2), (1600, 3072, 768, 1), (1600, 768, 3072, 2),       # config 2, stages 2 / 3
          (7168, 2048, 512, 1), (7168, 512, 2048, 2), (12288, 2048, 512, 1), (12288, 512, 2048, 2),     # config 5, stage 2
          (1792, 4096, 1024, 1), (1792, 1024, 4096, 2), (3072, 4096, 1024, 1), (3072, 1024, 4096, 2),   # config 5, stage 3
          (25600, 768, 192, 1), (25600, 192, 768, 2), (28672, 1024, 256, 1), (28672, 256, 1024, 2)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]


def timed(fn, iters=20, rounds=5):
    ts = []
    for r in range(rounds + 1):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        if r:
            ts.append(s.elapsed_time(e) / iters)
    ts.sort()
    return ts[len(ts) // 2]


tag = f"NT_TILE={os.environ.get('VKAS_NT_TILE', 'auto')}"
for M, N, K, mode in SHAPES:
    x = torch.randn((M, K), generator=g, device='cuda').bfloat16()
    w = (torch.randn((N, K), generator=g, device='cuda') * 0.05).bfloat16()
    out = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
    out2 = torch.empty_like(out)
    aux = torch.randn((M, N), generator=g, device='cuda').bfloat16()
    bias = torch.zeros((N,), device='cuda')
    cs = torch.ones((N,), device='cuda')
    W = 32
    geom = _lib.ConvGeom(1, M // W, W, M // W, W, K, K, 1, 1, 1, 0)
    epi = _lib.Epilogue()
    epi.mode, epi.out, epi.ldo, epi.bias = mode, out.data_ptr(), N, bias.data_ptr()
    if mode == 1:
        epi.out2, epi.ldo2 = out2.data_ptr(), N
    if mode == 2:
        epi.out2, epi.ldo2, epi.aux, epi.ldaux, epi.colscale = out2.data_ptr(), N, aux.data_ptr(), N, cs.data_ptr()
        epi.rows_per_image = M

    def run():
        rc = lib.vkas_conv_gemm_fwd(x.data_ptr(), ctypes.byref(geom), w.data_ptr(), N, ctypes.byref(epi), _lib.BF16, st)
        assert rc == 0
    ms = timed(run)
    print(f'{tag} NT M={M:6d} N={N:5d} K={K:5d} mode={mode} tile={lib.vkas_conv_gemm_tile(0, M, N, K):4d} {ms * 1e3:8.1f} us '
          f'{2.0 * M * N * K / ms / 1e9:7.1f} TF', flush=True)
