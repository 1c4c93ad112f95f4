#!/usr/bin/env python3
"""Time the backbone / neck GEMM shapes of one train step (1x1 convs of the ConvNeXt MLPs and their gradients) through the
C ABI, one line per (shape, epilogue).  The tile is chosen by the library; run once per VKAS_NT_TILE / VKAS_TN_TILE
override (1 = 128x128, 128 / 192 / 224 = N extent of the 256-row tile; TN: 128 / 192 / 224) to compare tiles."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import _lib  # noqa: E402

lib = _lib.lib
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device='cuda').manual_seed(0)
NT = [(524288, 384, 96, 1), (524288, 96, 384, 2), (524288, 384, 96, 3), (524288, 96, 384, 0),
      (131072, 768, 192, 1), (131072, 192, 768, 2), (131072, 768, 192, 3), (131072, 192, 768, 0),
      (32768, 1536, 384, 1), (32768, 384, 1536, 2), (32768, 1536, 384, 3), (32768, 384, 1536, 0),
      (8192, 3072, 768, 1), (8192, 768, 3072, 2), (8192, 3072, 768, 3), (8192, 768, 3072, 0)]
TN = [(524288, 384, 96), (524288, 96, 384), (131072, 768, 192), (131072, 192, 768), (32768, 1536, 384),
      (32768, 384, 1536), (8192, 3072, 768), (8192, 768, 3072), (524288, 96, 864), (131072, 96, 864)]


def timed(fn, iters=5, rounds=5):
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


which = sys.argv[1] if len(sys.argv) > 1 else 'both'
if len(sys.argv) > 2 and sys.argv[2] == 'merged':  # backbone batch 16 (merged pass schedule)
    NT = [(2 * m, n, k, mode) for m, n, k, mode in NT]
    TN = [(2 * m, n, k) for m, n, k in TN]
tag = f"NT_TILE={os.environ.get('VKAS_NT_TILE', 'auto')} TN_TILE={os.environ.get('VKAS_TN_TILE', 'auto')}"
if which in ('nt', 'both'):
    for M, N, K, mode in NT:
        x = torch.randn((M, K), generator=g, device='cuda').bfloat16()
        w = (torch.randn((N, K), generator=g, device='cuda') * 0.05).bfloat16()
        out = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
        out2 = torch.empty_like(out)
        aux = torch.randn((M, N), generator=g, device='cuda').bfloat16()
        bias = torch.zeros((N,), device='cuda')
        cs = torch.ones((N,), device='cuda')
        geom = _lib.ConvGeom(1, M // 256, 256, M // 256, 256, K, K, 1, 1, 1, 0)
        epi = _lib.Epilogue()
        epi.mode, epi.out, epi.ldo, epi.bias = mode, out.data_ptr(), N, bias.data_ptr()
        nbytes = M * (K + N) * 2
        if mode == 1:
            epi.out2, epi.ldo2 = out2.data_ptr(), N
            nbytes += M * N * 2
        if mode == 2:
            epi.out2, epi.ldo2, epi.aux, epi.ldaux, epi.colscale = out2.data_ptr(), N, aux.data_ptr(), N, cs.data_ptr()
            epi.rows_per_image = M
            nbytes += 2 * M * N * 2
        if mode == 3:
            epi.aux, epi.ldaux, epi.bias = aux.data_ptr(), N, None
            nbytes += M * N * 2
        def run():
            rc = lib.vkas_conv_gemm_fwd(x.data_ptr(), ctypes.byref(geom), w.data_ptr(), N, ctypes.byref(epi), _lib.BF16, st)
            assert rc == 0
        ms = timed(run)
        print(f'{tag} NT M={M:7d} N={N:5d} K={K:5d} mode={mode} tile={lib.vkas_conv_gemm_tile(0, M, N, K):4d} {ms * 1e3:8.1f} us '
              f'{2.0 * M * N * K / ms / 1e9:7.1f} TF {nbytes / ms / 1e9:6.2f} TB/s', flush=True)
        del x, w, out, out2, aux
if which in ('tn', 'both'):
    for M, N, K in TN:
        KH = 3 if K == 864 else 1
        C = K // (KH * KH)
        W = 256
        x = torch.randn((M, C), generator=g, device='cuda').bfloat16()
        dy = torch.randn((M, N), generator=g, device='cuda').bfloat16()
        gw = torch.zeros((N * K + N,), device='cuda')
        geom = _lib.ConvGeom(1, M // W, W, M // W, W, C, C, KH, KH, 1, KH // 2)
        def run():
            fn = lib.vkas_conv_gemm_wgrad_gelu if os.environ.get('SWEEP_XG') else lib.vkas_conv_gemm_wgrad
            rc = fn(x.data_ptr(), ctypes.byref(geom), dy.data_ptr(), N, N, gw.data_ptr(),
                                          None if os.environ.get('SWEEP_NOBIAS') else gw.data_ptr() + 4 * N * K, _lib.BF16, st)
            assert rc == 0
        ms = timed(run)
        print(f'{tag} TN M={M:7d} N={N:5d} K={K:5d} tile={lib.vkas_conv_gemm_tile(1, M, N, K):4d} {ms * 1e3:8.1f} us '
              f'{2.0 * M * N * K / ms / 1e9:7.1f} TF {M * (C + N) * 2 / ms / 1e9:6.2f} TB/s', flush=True)
        del x, dy, gw
