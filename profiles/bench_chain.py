#!/usr/bin/env python3
"""Microbenchmark of the fused ConvNeXt MLP kernels (csrc/mlp_chain.hip) at the stage-0 / stage-1 shapes of config #3.

    python profiles/bench_chain.py [libvkas variant .so ...]

Each library is loaded with ctypes directly (timing-only ablation builds from build_variants/ included), so several
variants can be compared in one GPU call.  Prints ms per launch and algorithmic TB/s; outputs of the first library are
the reference the others are compared with (max abs difference; ablation builds are expected to differ).
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p


def ptr(t):
    return None if t is None else P(t.data_ptr())


def run(path, M, C, keep=None, only=None):
    lib = ctypes.CDLL(path)
    lib.vkas_mlp_chain_image_elems.restype = ctypes.c_size_t
    lib.vkas_mlp_chain_image_elems.argtypes = [ctypes.c_int]
    L, I = ctypes.c_long, ctypes.c_int
    lib.vkas_mlp_chain_pack.argtypes = [P, P, P, I, I, P, I, P]
    lib.vkas_mlp_chain_fwd.argtypes = [P, L, P, P, P, L, P, P, I, P, L, P, L, P, L, L, I, I, P]
    lib.vkas_mlp_chain_bwd.argtypes = [P, L, P, P, L, P, L, P, L, L, I, I, P]
    lib.vkas_last_error.restype = ctypes.c_char_p
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(1)
    H4 = 4 * C
    w1 = (torch.randn(H4, C, generator=g) / C ** 0.5).to(dev)
    w2 = (torch.randn(C, H4, generator=g) / H4 ** 0.5).to(dev)
    b1 = (torch.randn(H4, generator=g) * 0.1).to(dev)
    b2 = (torch.randn(C, generator=g) * 0.1).to(dev)
    cs = torch.ones(C, device=dev)
    n = lib.vkas_mlp_chain_image_elems(C)
    img = torch.empty(n, dtype=torch.bfloat16, device=dev)
    imgt = torch.empty(n, dtype=torch.bfloat16, device=dev)
    st = P(torch.cuda.current_stream().cuda_stream)
    chk = lambda rc: (_ for _ in ()).throw(RuntimeError(lib.vkas_last_error())) if rc else None
    chk(lib.vkas_mlp_chain_pack(ptr(w1), ptr(w2), ptr(b1), C, 0, ptr(img), 1, st))
    chk(lib.vkas_mlp_chain_pack(ptr(w1), ptr(w2), None, C, 1, ptr(imgt), 1, st))
    yn = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
    x = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
    dz = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
    h = torch.empty(M, H4, dtype=torch.bfloat16, device=dev)
    z = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    dh = torch.empty(M, H4, dtype=torch.bfloat16, device=dev)
    dyn = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    rpi = M // 16 if M % 16 == 0 else M

    def fwd():
        chk(lib.vkas_mlp_chain_fwd(ptr(yn), C, ptr(img), ptr(b2), ptr(x), C, ptr(cs), None, rpi, ptr(h), H4, ptr(z), C,
                                   ptr(out), C, M, C, 1, st))

    def bwd():
        chk(lib.vkas_mlp_chain_bwd(ptr(dz), C, ptr(imgt), ptr(h), H4, ptr(dh), H4, ptr(dyn), C, M, C, 1, st))

    res = {}
    for name, fn, nbytes in (('fwd', fwd, M * (4 * C + H4) * 2), ('bwd', bwd, M * (2 * C + 2 * H4) * 2)):
        if only is not None and name != only:
            continue
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        res[name] = ms
        print(f'  {os.path.basename(path):28s} M={M:8d} C={C:3d} {name}: {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s  '
              f'{4.0 * M * C * H4 / ms / 1e9:7.1f} TFLOP/s', flush=True)
    outs = [t.float() for t in (h, z, out, dh, dyn)]
    if keep is not None:
        print('    max |diff| vs first library (h z out dh dyn):', ' '.join(f'{float((a - b).abs().max()):.3g}' for a, b in zip(outs, keep)))
    return outs


def main():
    libs = [a for a in sys.argv[1:] if not a.startswith('--')] or [os.path.join(ROOT, 'vkit_ocr_model_adaptive_scaling_amd', 'libvkas.so')]
    shapes = ((1048576, 96), (262144, 192), (1048576 + 40, 96))
    for a in sys.argv[1:]:  # --shapes=65536x384,16384x384  (VKAS_CHAIN_PAIR=0 in the environment: the one-wave-per-SIMD kernel at C > 256)
        if a.startswith('--shapes='):
            shapes = tuple(tuple(int(v) for v in sh.split('x')) for sh in a[9:].split(','))
    for M, C in shapes:
        keep = None
        for p in libs:
            o = run(p, M, C, keep)
            keep = keep or o


if __name__ == '__main__':
    main()
