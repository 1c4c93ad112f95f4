#!/usr/bin/env python3
"""Times the row-slab 3x3 kernels of one train step at 8 x 512 x 512 through the C ABI: the fused head forwards
(conv3x3_slab_mfma_kernel<7,1> precise, <6,1> rough), the plain forward / input-gradient shapes (<6,0>: N = 384 with K = 9 x 384
and K = 9 x 192) and the weight gradients (conv3x3_wgrad_slab_kernel<8> N = 384, <7> N = 192).  Run once per library build
(VKAS_LIB_PATH=build_variants/...) to A/B a kernel change."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import _lib, ops  # noqa: E402

lib = _lib.lib
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device='cuda').manual_seed(0)
B, H, W = 8, 512, 512
M = B * H * W


def timed(fn, iters=3, rounds=3):
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


tag = os.path.basename(os.environ.get('VKAS_LIB_PATH', 'libvkas.so'))
x = torch.randn((B, H, W, 384), generator=g, device='cuda').bfloat16()
# fused head forwards
for name, chans, ocs in (('precise <7,1>', (192, 193, 194, 194), (1, 2, 4, 4)), ('rough <6,1>', (192, 192), (1, 1))):
    params = []
    for c, oc in zip(chans, ocs):
        params += [torch.randn((c, 384, 3, 3), generator=g, device='cuda') * 0.02, torch.zeros(c, device='cuda'),
                   torch.ones(c, device='cuda'), torch.zeros(c, device='cuda'),
                   torch.randn((oc, c), generator=g, device='cuda') * 0.05, torch.zeros(oc, device='cuda')]
    ms = timed(lambda: ops.HeadsFused.apply(x, True, *params))
    fl = 2.0 * M * sum(chans) * 9 * 384
    print(f'{tag:24s} heads fwd {name:14s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s', flush=True)
# plain forward / dgrad shapes
for N, Cin in ((384, 384), (384, 192)):
    xi = torch.randn((B, H, W, Cin), generator=g, device='cuda').bfloat16()
    w = (torch.randn((N, 9 * Cin), generator=g, device='cuda') * 0.02).bfloat16()
    out = torch.empty((B, H, W, N), device='cuda', dtype=torch.bfloat16)
    geom = _lib.ConvGeom(B, H, W, H, W, Cin, Cin, 3, 3, 1, 1)
    epi = _lib.Epilogue()
    epi.mode, epi.out, epi.ldo = _lib.EPI_NONE, out.data_ptr(), N

    def run():
        rc = lib.vkas_conv_gemm_fwd(xi.data_ptr(), ctypes.byref(geom), w.data_ptr(), N, ctypes.byref(epi), _lib.BF16, st)
        assert rc == 0
    ms = timed(run)
    print(f'{tag:24s} conv3x3 N={N} K=9x{Cin:3d} <6,0>   {ms:7.3f} ms  {2.0 * M * N * 9 * Cin / ms / 1e9:7.1f} TFLOP/s', flush=True)
# weight gradients
for N in (384, 192):
    dy = torch.randn((B, H, W, N), generator=g, device='cuda').bfloat16()
    gw = torch.zeros((N * 9 * 384 + N,), device='cuda')
    geom = _lib.ConvGeom(B, H, W, H, W, 384, 384, 3, 3, 1, 1)

    def run():
        rc = lib.vkas_conv_gemm_wgrad(x.data_ptr(), ctypes.byref(geom), dy.data_ptr(), N, N, gw.data_ptr(),
                                      gw.data_ptr() + 4 * N * 9 * 384, _lib.BF16, st)
        assert rc == 0
    ms = timed(run)
    print(f'{tag:24s} wgrad N={N} K=9x384          {ms:7.3f} ms  {2.0 * M * N * 9 * 384 / ms / 1e9:7.1f} TFLOP/s', flush=True)
