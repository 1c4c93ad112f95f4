#!/usr/bin/env python3
"""Time the row kernels of a ConvNeXt layer (LayerNorm forward / backward, scale-residual backward) at the four stage shapes of
config #3 (merged schedule: 16 images) through the C ABI: microseconds and algorithmic TB/s per launch.  VKAS_LIB_PATH selects
a variant build (profiles/build_variant.sh with SRC=norm.hip)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import _lib  # noqa: E402

lib = _lib.lib
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(16 * 256 * 256, 96), (16 * 128 * 128, 192), (16 * 64 * 64, 384), (16 * 32 * 32, 768)]


def timed(fn, iters=10, rounds=5):
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


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


g = torch.Generator(device='cuda').manual_seed(0)
for M, C in SHAPES:
    x = torch.randn((M, C), generator=g, device='cuda').bfloat16()
    dy = torch.randn((M, C), generator=g, device='cuda').bfloat16()
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    gamma = torch.rand((C,), generator=g, device='cuda') + 0.5
    beta = torch.randn((C,), generator=g, device='cuda') * 0.1
    stats = torch.empty((M, 2), device='cuda')
    rs = torch.ones((16,), device='cuda')

    def fwd(gelu):
        assert lib.vkas_layernorm_fwd(p(x), C, p(gamma), p(beta), p(y), C, p(stats), M, C, C, gelu, _lib.BF16, st) == 0

    nb = lib.vkas_layernorm_bwd_ws_bytes(M, C)
    ws = torch.empty((nb // 4 + 4,), device='cuda')
    dgb = torch.empty((2 * C,), device='cuda')

    def bwd(gelu):
        assert lib.vkas_layernorm_bwd(p(x), C, p(gamma), p(beta), p(stats), p(dy), C, p(dx), C, p(dgb[:C]), p(dgb[C:]), p(ws),
                                      nb, M, C, C, gelu, _lib.BF16, st) == 0

    nb2 = lib.vkas_scale_res_bwd_ws_bytes(M, C)
    ws2 = torch.empty((nb2 // 4 + 4,), device='cuda')

    def srb():
        assert lib.vkas_scale_res_bwd(p(dy), C, p(x), C, p(gamma), p(rs), M // 16, p(dx), C, p(dgb[:C]), p(dgb[C:]), p(ws2), nb2,
                                      M, C, _lib.BF16, st) == 0

    fwd(0)
    e = 2.0 * M * C
    for name, fn, nbytes in (('ln_fwd', lambda: fwd(0), 2 * e), ('ln_fwd_gelu', lambda: fwd(1), 2 * e),
                             ('ln_bwd', lambda: bwd(0), 3 * e), ('ln_bwd_gelu', lambda: bwd(1), 3 * e),
                             ('scale_res_bwd', srb, 3 * e)):
        ms = timed(fn)
        print(f'M={M:8d} C={C:4d} {name:14s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s', flush=True)
    del x, dy, y, dx
