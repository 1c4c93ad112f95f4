"""Op-level parity of the HIP kernels (through the C ABI, via the autograd ops) against fp64 CPU math.

Every case runs in fp32 mode (exact-fp32 kernels; tolerance 2e-5 norm-wise, 1e-4 of the peak value
element-wise) and in bf16 mode (MFMA kernels; inputs and parameters are pre-rounded to bf16 so the only
differences are output rounding and accumulation order: 4e-3 norm-wise, 2e-2 of the peak element-wise).
Integer index math (resize / pooling / gather) is exercised at awkward, non-square, non-power-of-two sizes.
"""
import math

import numpy as np
import pytest
import torch
from torch.nn import functional as F

from oracle import torch_oracle as O
from tests.golden import recipe
from tests.helpers import golden, rel_err

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: (2e-5, 1e-4), torch.bfloat16: (4e-3, 2e-2), torch.float16: (6e-4, 4e-3)}


def ops_mod():
    from vkit_ocr_model_adaptive_scaling_amd import ops
    return ops


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale)


def q(t, dtype):
    """Round to the storage dtype and come back as fp64 (what the kernel will actually see)."""
    return t.to(dtype).double()


def to_act(x_nchw64, dtype, ld_extra=0):
    """(B,C,H,W) fp64 CPU -> NHWC activation on the GPU, optionally as a channel slice of a wider buffer."""
    B, C, H, W = x_nchw64.shape
    Cp = (C + 7) // 8 * 8
    buf = torch.zeros((B, H, W, Cp + ld_extra), dtype=dtype, device='cuda')
    buf[..., :C] = x_nchw64.permute(0, 2, 3, 1).to(dtype).cuda()
    return buf[..., :Cp] if ld_extra else buf


def from_act(a, C):
    return a[..., :C].permute(0, 3, 1, 2).double().cpu()


def close(actual, expected, dtype, what=''):
    rt, mt = TOL[dtype]
    actual = actual.double().cpu()
    expected = expected.double().cpu()
    assert actual.shape == expected.shape, (what, actual.shape, expected.shape)
    assert torch.isfinite(actual).all(), what
    r = rel_err(actual, expected)
    peak = float(expected.abs().max())
    m = float((actual - expected).abs().max())
    assert r < rt, f'{what}: norm-wise rel err {r:.3e} >= {rt}'
    assert m <= mt * max(peak, 1e-30), f'{what}: max abs err {m:.3e} vs peak {peak:.3e}'


# ------------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # B, Cin, H, W, N, K, stride, pad
    (2, 96, 9, 13, 384, 1, 1, 0),
    (1, 384, 5, 7, 96, 1, 1, 0),
    (2, 16, 9, 11, 24, 3, 1, 1),
    (1, 96, 40, 36, 96, 3, 1, 1),
    (2, 64, 12, 20, 33, 3, 1, 1),
    (2, 33, 6, 10, 4, 1, 1, 0),
    (2, 32, 12, 20, 64, 2, 2, 0),
    (1, 3, 32, 64, 16, 4, 4, 0),
    (1, 3, 32, 64, 16, 2, 2, 0),
    (3, 200, 17, 9, 136, 1, 1, 0),
    (1, 64, 6, 5, 3072, 1, 1, 0),  # > 2048 output channels (stage-3 MLP width): bias-grad column sums in two passes
    # rows of whole 256-pixel tiles, M >= 16384: the 3x3 row-slab kernel (fwd and dgrad), channel-block tails, both halos
    (2, 72, 33, 256, 200, 3, 1, 1),
    (1, 64, 34, 512, 384, 3, 1, 1),
    # ... and M >= 65536 with >= 128 input channels: the wgrad row-slab kernel (n / channel tails, W = 64 and 192)
    (2, 136, 512, 64, 200, 3, 1, 1),
    (1, 384, 342, 192, 136, 3, 1, 1),
    (1, 128, 256, 256, 256, 3, 1, 1),  # 128-channel n tiles (the two above pad less with 112)
    (1, 136, 256, 256, 192, 3, 1, 1),  # 96-channel n tiles (the probability head's width: two full tiles, rotated dy rows)
    (2, 128, 512, 64, 280, 3, 1, 1),   # ... with an n tail (three tiles, the last 88 wide)
]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv(case, dtype):
    ops = ops_mod()
    B, Cin, H, W, N, K, stride, pad = case
    x = q(rnd((B, Cin, H, W), 1), dtype)
    w = q(rnd((N, Cin, K, K), 2, 1.0 / math.sqrt(Cin * K * K)), dtype)
    b = rnd((N,), 3, 0.1)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    cot = q(rnd(tuple(ref.shape), 4), dtype)
    (ref * cot).sum().backward()

    if Cin == 3:
        xa = ops.ImageToAct.apply(x.float().cuda(), dtype)
    else:
        xa = to_act(x, dtype, ld_extra=8).requires_grad_(True)
    wg = w.float().cuda().requires_grad_(True)
    bg = b.float().cuda().requires_grad_(True)
    wparam = wg if K > 1 else wg.view(N, Cin)  # nn.Linear layout for 1x1
    y = ops.Conv.apply(xa, wparam, bg, stride, pad, Cin != 3)
    close(from_act(y, N), ref.detach(), dtype, 'conv fwd')
    if (N + 7) // 8 * 8 != N:
        assert float(y[..., N:].abs().max()) == 0.0, 'pad channels must stay zero'
    Np = y.shape[3]
    cot_a = torch.zeros_like(y)
    cot_a[..., :N] = cot.permute(0, 2, 3, 1).to(dtype).cuda()
    y.backward(cot_a)
    close(wg.grad.view(N, Cin, K, K), wr.grad, dtype, 'conv wgrad')
    close(bg.grad, br.grad, dtype, 'conv bias grad')
    if Cin != 3:
        close(from_act(xa.grad, Cin), xr.grad, dtype, 'conv dgrad')


def test_rowslab_kernels_repeatable():
    """Race screen for the LDS-DMA row-slab kernels (counted vmcnt + raw barriers): the forward / dgrad kernel has no
    atomics, so 12 launches on the same operands must agree bit for bit; the weight-gradient kernel (fp32 atomics over the
    pixel splits) must agree to accumulation order."""
    ops = ops_mod()
    torch.manual_seed(3)
    x = torch.randn((2, 130, 512, 136), device='cuda').bfloat16().requires_grad_(True)   # M = 133120, Cp = 136
    w = (torch.randn((200, 136, 3, 3), device='cuda') * 0.03).requires_grad_(True)
    b = torch.randn((200,), device='cuda').requires_grad_(True)
    cot = torch.randn((2, 130, 512, 200), device='cuda').bfloat16()
    ys, gxs, gws = [], [], []
    for _ in range(12):
        x.grad = w.grad = b.grad = None
        y = ops.Conv.apply(x, w, b, 1, 1)
        y.backward(cot)
        ys.append(y.detach().clone())
        gxs.append(x.grad.clone())
        gws.append(w.grad.clone())
    for y, gx, gw in zip(ys[1:], gxs[1:], gws[1:]):
        assert torch.equal(y, ys[0])
        assert torch.equal(gx, gxs[0])
        assert rel_err(gw, gws[0]) < 1e-5


def test_conv_mfma_matches_simple_bitwise_shapes():
    """bf16: the MFMA kernels and the plain fp32-FMA kernels see identical inputs; results agree to accumulation order."""
    import os, subprocess, sys
    # a second process with VKAS_GEMM=simple computes the same case; compare through a file
    code = r'''
import torch, sys
from vkit_ocr_model_adaptive_scaling_amd import ops
g = torch.Generator().manual_seed(5)
x = torch.randn((2, 20, 28, 96), generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
w = (torch.randn((40, 96, 3, 3), generator=g) * 0.03).cuda().requires_grad_(True)
b = torch.randn((40,), generator=g).cuda().requires_grad_(True)
y = ops.Conv.apply(x, w, b, 1, 1)
y.backward(torch.ones_like(y))
torch.save({'y': y.float().cpu(), 'gx': x.grad.float().cpu(), 'gw': w.grad.cpu()}, sys.argv[1])
'''
    import tempfile
    outs = []
    for mode in ('mfma', 'simple'):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            env = dict(os.environ, VKAS_GEMM=mode)
            subprocess.run([sys.executable, '-c', code, f.name], check=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            outs.append(torch.load(f.name, weights_only=True))
    for k in ('y', 'gx', 'gw'):
        assert rel_err(outs[0][k], outs[1][k]) < 3e-3, k


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('variant', ['bias', 'nobias', 'gelu'])
@pytest.mark.parametrize('shape', [(376, 264, 192), (768, 376, 384)], ids=['tile192', 'tile384'])
def test_pointwise_wgrad_8wave_tiles(shape, variant, dtype):
    """Weight (and bias) gradient of a pointwise layer at the sizes that take the 8-wave 192-wide tile of gemm_tn_mfma_kernel
    (M >= 16 384 rows): the variant with bias sums (v_dot2c against ones), the one compiled without them (second staging set in
    flight) and the one that applies GELU to the staged operand - ragged M / N / K tails - against fp64 math on the stored
    operands."""
    ops = ops_mod()
    from vkit_ocr_model_adaptive_scaling_amd import _lib
    import ctypes
    g = torch.Generator().manual_seed(17)
    # tile384: N a multiple of 384 and 256 < K <= 384 take the 384 (N) x 128 (K) tile (no zero-padded K tile); K tail ragged
    M, (N, K, tile) = 16384 + 72, shape
    x = (torch.randn((M, K), generator=g) * 1.5).to(dtype).cuda()
    dy = torch.randn((M, N), generator=g).to(dtype).cuda()
    gw = torch.zeros((N * K + N,), device='cuda')
    geom = _lib.ConvGeom(1, 1, M, 1, M, K, K, 1, 1, 1, 0)
    assert _lib.lib.vkas_conv_gemm_kernel_id(1, ctypes.byref(geom), N, N, 0) == tile
    fn = _lib.lib.vkas_conv_gemm_wgrad_gelu if variant == 'gelu' else _lib.lib.vkas_conv_gemm_wgrad
    gb = None if variant == 'nobias' else gw.data_ptr() + 4 * N * K
    rc = fn(x.data_ptr(), ctypes.byref(geom), dy.data_ptr(), N, N, gw.data_ptr(), gb,
            _lib.BF16 if dtype == torch.bfloat16 else _lib.F16, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    xr = x.double().cpu()
    if variant == 'gelu':
        xr = O.gelu(xr).to(dtype).double()  # the kernel rounds gelu(x) to the storage type, as the forward's GEMM-b operand was
    ref_w = dy.double().cpu().t() @ xr
    close(gw[:N * K].view(N, K), ref_w, dtype, 'weight gradient')
    ref_b = torch.zeros(N, dtype=torch.float64) if variant == 'nobias' else dy.double().cpu().sum(0)
    assert rel_err(gw[N * K:].double().cpu(), ref_b) < 1e-5 if variant != 'nobias' else float(gw[N * K:].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'f32'])
def test_wgrad_ordered_is_one_split_and_reproducible(dtype):
    """vkas_conv_gemm_wgrad_ordered (the label-point heads' input-gradient terms, HeadsFused.backward): the shape of that call
    - few reduced rows, many output tiles - against fp64 math and against the split-M entry point, and bit-identical over
    repeated launches (every output tile is summed by one workgroup in row order and added once to the zeroed buffer)."""
    ops_mod()
    from vkit_ocr_model_adaptive_scaling_amd import _lib
    import ctypes
    g = torch.Generator().manual_seed(23)
    M, N, K = 584, 320, 1736  # reduced rows (the heads' z columns), output rows (label points), 9 * Cp columns
    x = torch.randn((M, K), generator=g).to(dtype).cuda()
    dy = torch.randn((M, N), generator=g).to(dtype).cuda()
    geom = _lib.ConvGeom(1, 1, M, 1, M, K, K, 1, 1, 1, 0)
    code = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for _ in range(6):
        gw = torch.zeros((N * K,), device='cuda')
        assert _lib.lib.vkas_conv_gemm_wgrad_ordered(x.data_ptr(), ctypes.byref(geom), dy.data_ptr(), N, N, gw.data_ptr(), code, st) == 0
        outs.append(gw)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:]), 'ordered weight gradient differs between launches'
    ref = dy.double().cpu().t() @ x.double().cpu()
    assert rel_err(outs[0].view(N, K).double().cpu(), ref) < 1e-5
    gs = torch.zeros((N * K,), device='cuda')
    assert _lib.lib.vkas_conv_gemm_wgrad(x.data_ptr(), ctypes.byref(geom), dy.data_ptr(), N, N, gs.data_ptr(), None, code, st) == 0
    assert rel_err(gs.double().cpu(), outs[0].double().cpu()) < 1e-6


def test_ring_kernel_matches_register_staged_kernel_bitwise():
    """gemm_nt_ring_kernel (launches of few tiles: LDS-DMA ring) against gemm_nt_mfma_kernel<2,2,4,4> in a second process with
    VKAS_NT_RING=0: same geometry decode, K order and epilogue, so the outputs are bit-identical - pointwise shapes with M / N / K
    tails, a 3x3 im2col geometry with padding, a strided patchify, ring depths 2 and 4, and the no-grad layer path that drops
    the pre-activation."""
    import os, subprocess, sys, tempfile
    code = r'''
import torch, sys
from vkit_ocr_model_adaptive_scaling_amd import ops, _lib
g = torch.Generator().manual_seed(9)
out = {}
for i, (B, H, W, C, N, k, s, p) in enumerate([(1, 37, 29, 520, 136, 1, 1, 0), (2, 40, 40, 384, 1536, 1, 1, 0),
                                               (1, 20, 28, 96, 40, 3, 1, 1), (1, 32, 24, 96, 192, 2, 2, 0)]):
    x = torch.randn((B, H, W, C), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn((N, C, k, k), generator=g) * 0.03).cuda()
    b = torch.randn((N,), generator=g).cuda()
    with torch.no_grad():
        out['conv%d' % i] = ops.Conv.apply(x, w, b, s, p).float().cpu()
C = 320
x = torch.randn((1, 12, 16, C), generator=g).to(torch.bfloat16).cuda()
ps = [torch.randn(C, 1, 7, 7, generator=g) * 0.15, torch.randn(C, generator=g) * 0.1, torch.ones(C), torch.zeros(C),
      torch.randn(4 * C, C, generator=g) / C ** 0.5, torch.randn(4 * C, generator=g) * 0.1,
      torch.randn(C, 4 * C, generator=g) * 0.5 / C ** 0.5, torch.randn(C, generator=g) * 0.1, torch.ones(C, 1, 1)]
ps = [t.cuda() for t in ps]
with torch.no_grad():
    out['layer_nograd'] = ops.ConvNextLayer.apply(x, *ps, None, False).float().cpu()
out['layer_keep'] = ops.ConvNextLayer.apply(x, *ps, None, True).float().cpu()
geom = _lib.ConvGeom(1, 12, 16, 12, 16, C, C, 1, 1, 1, 0)
out['kid'] = torch.tensor([_lib.lib.vkas_conv_gemm_kernel_id(0, __import__('ctypes').byref(geom), 4 * C, 0, 0)])
torch.save(out, sys.argv[1])
'''
    outs = {}
    for ring in ('0', '2', '4', ''):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            env = dict(os.environ)
            env.pop('VKAS_NT_RING', None)
            if ring:
                env['VKAS_NT_RING'] = ring
            subprocess.run([sys.executable, '-c', code, f.name], check=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            outs[ring] = torch.load(f.name, weights_only=True)
    assert int(outs['0']['kid']) == 1 and int(outs['2']['kid']) == 12 and int(outs['4']['kid']) == 14 and int(outs['']['kid']) in (12, 14)
    for ring in ('2', '4', ''):
        for k, v in outs['0'].items():
            if k != 'kid':
                assert torch.equal(v, outs[ring][k]), (ring, k, rel_err(v, outs[ring][k]))
    # dropping the pre-activation (and z) changes nothing in what the layer returns
    assert torch.equal(outs['']['layer_nograd'], outs['']['layer_keep'])


# --------------------------------------------------------------------------------------------- dwconv / layer
@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(2, 24, 19, 37), (1, 16, 8, 8), (2, 96, 40, 33), (1, 40, 5, 3),
                                   # every instantiation of the fused MLP kernel (C <= 32 / 64 / 96 / 128 / 192 / 256 / 384 /
                                   # 512) and one width beyond it (two-GEMM path)
                                   (1, 64, 7, 9), (2, 128, 12, 9), (1, 192, 9, 7), (1, 256, 6, 5), (1, 264, 4, 4),
                                   (2, 384, 9, 5), (1, 512, 6, 7), (1, 520, 3, 3)])
def test_convnext_layer(shape, dtype):
    """Whole ConvNextBlockLayer (dw7x7, LN, MLP, layer scale, stochastic depth mask, residual) fwd + bwd."""
    ops = ops_mod()
    B, C, H, W = shape
    sd = {
        'block.0.weight': q(rnd((C, 1, 7, 7), 10, 0.15), torch.float32), 'block.0.bias': rnd((C,), 11, 0.1),
        'block.2.weight': 1 + rnd((C,), 12, 0.1), 'block.2.bias': rnd((C,), 13, 0.1),
        'block.3.weight': q(rnd((4 * C, C), 14, 1 / math.sqrt(C)), dtype), 'block.3.bias': rnd((4 * C,), 15, 0.1),
        'block.5.weight': q(rnd((C, 4 * C), 16, 0.5 / math.sqrt(C)), dtype), 'block.5.bias': rnd((C,), 17, 0.1),
        'block_scale': (1 + rnd((C, 1, 1), 18, 0.2)),
    }
    x = q(rnd(shape, 19), dtype)
    mask = torch.tensor([1.25, 0.0][:B], dtype=torch.float64) if B > 1 else torch.tensor([1.25], dtype=torch.float64)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.convnext_layer(sdr, '', xr, mask.view(-1, 1, 1, 1))
    cot = q(rnd(shape, 20), dtype)
    (ref * cot).sum().backward()

    xa = to_act(x, dtype).requires_grad_(True)
    pg = {k: v.float().cuda().requires_grad_(True) for k, v in sd.items()}
    y = ops.ConvNextLayer.apply(xa, pg['block.0.weight'], pg['block.0.bias'], pg['block.2.weight'], pg['block.2.bias'],
                                pg['block.3.weight'], pg['block.3.bias'], pg['block.5.weight'], pg['block.5.bias'],
                                pg['block_scale'], mask.float().cuda())
    close(from_act(y, C), ref.detach(), dtype, 'layer fwd')
    y.backward(to_act(cot, dtype))
    # bf16: the chain dw -> LN -> GEMM -> GELU -> GEMM re-rounds intermediates; allow 3x the single-op tolerance
    gtol = {torch.float32: 5e-5, torch.bfloat16: 1.5e-2, torch.float16: 3e-3}[dtype]
    for k in sd:
        r = rel_err(pg[k].grad, sdr[k].grad)
        assert r < gtol, (k, r)
    r = rel_err(from_act(xa.grad, C), xr.grad)
    assert r < gtol, ('dx', r)


@pytest.mark.parametrize('C', [16, 96, 192, 320, 384, 512])
def test_mlp_chain_matches_two_gemm_path(C):
    """The fused MLP kernels against the two-GEMM layer path they replace (same bf16 inputs; the two differ only in
    where fp32 values are rounded to bf16): outputs and every gradient, ragged M (not a multiple of the 256/128-row tile),
    dropped sample."""
    ops = ops_mod()
    torch.manual_seed(C)
    B, H, W = 3, 37, 29
    x = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    names = ['dw_w', 'dw_b', 'ln_g', 'ln_b', 'w1', 'b1', 'w2', 'b2', 'scale']
    base = [torch.randn(C, 1, 7, 7) * 0.15, torch.randn(C) * 0.1, 1 + torch.randn(C) * 0.1, torch.randn(C) * 0.1,
            (torch.randn(4 * C, C) / math.sqrt(C)).to(torch.bfloat16).float(), torch.randn(4 * C) * 0.1,
            (torch.randn(C, 4 * C) * 0.5 / math.sqrt(C)).to(torch.bfloat16).float(), torch.randn(C) * 0.1,
            1 + torch.randn(C, 1, 1) * 0.2]
    mask = torch.tensor([1.25, 0.0, 1.0], device='cuda')
    cot = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    res = {}
    max_c, min_rows = ops._CHAIN_MAX_C, ops._CHAIN_PAIR_MIN_ROWS
    for chain in (True, False):
        ops._NO_CHAIN = not chain
        ops._CHAIN_MAX_C = 512  # the wide instantiations are not the default path (no faster than two GEMMs)
        ops._CHAIN_PAIR_MIN_ROWS = 0  # nor is the pair-split kernel at this few rows
        try:
            ps = [t.clone().cuda().requires_grad_(True) for t in base]
            xa = x.clone().requires_grad_(True)
            assert ops.mlp_chain_eligible(xa, C) == chain
            y = ops.ConvNextLayer.apply(xa, *ps, mask)
            y.backward(cot)
            res[chain] = [y.detach()] + [xa.grad] + [p.grad for p in ps]
        finally:
            ops._NO_CHAIN = False
            ops._CHAIN_MAX_C = max_c
            ops._CHAIN_PAIR_MIN_ROWS = min_rows
    for name, a, b in zip(['out', 'dx'] + names, res[True], res[False]):
        assert rel_err(a, b) < 1.5e-2, (name, rel_err(a, b))
    assert float(res[True][0][1].float().sub(x[1].float()).abs().max()) == 0.0, 'dropped sample must pass through'


@pytest.mark.parametrize('C', [24, 96, 192, 384])
def test_mlp_chain_fused_layernorm(C):
    """LayerNorm inside the fused MLP kernel (vkas_mlp_chain_ln_fwd) against LayerNorm as its own launch in front of it
    (vkas_layernorm_fwd + vkas_mlp_chain_fwd): the two sum a row's channels in a different order, so a normalised value may
    differ by one 16-bit rounding - outputs, every gradient (the saved yn / stats feed the W1 weight gradient and
    the LayerNorm backward), ragged M, and the no-grad call (nothing stored) bit-equal to the training call's output."""
    ops = ops_mod()
    torch.manual_seed(100 + C)
    B, H, W = 3, 37, 29
    x = (torch.randn(B, H, W, C, device='cuda') * 1.5 + 0.3).to(torch.bfloat16)
    base = [torch.randn(C, 1, 7, 7) * 0.15, torch.randn(C) * 0.1, 1 + torch.randn(C) * 0.1, torch.randn(C) * 0.1,
            (torch.randn(4 * C, C) / math.sqrt(C)).to(torch.bfloat16).float(), torch.randn(4 * C) * 0.1,
            (torch.randn(C, 4 * C) * 0.5 / math.sqrt(C)).to(torch.bfloat16).float(), torch.randn(C) * 0.1,
            1 + torch.randn(C, 1, 1) * 0.2]
    names = ['dw_w', 'dw_b', 'ln_g', 'ln_b', 'w1', 'b1', 'w2', 'b2', 'scale']
    mask = torch.tensor([1.25, 0.0, 1.0], device='cuda')
    cot = torch.randn(B, H, W, C, device='cuda').to(torch.bfloat16)
    res = {}
    min_rows, no_ln = ops._CHAIN_PAIR_MIN_ROWS, ops._NO_CHAIN_LN
    try:
        ops._CHAIN_PAIR_MIN_ROWS = 0
        for fused in (True, False):
            ops._NO_CHAIN_LN = not fused
            ps = [t.clone().cuda().requires_grad_(True) for t in base]
            xa = x.clone().requires_grad_(True)
            assert ops.mlp_chain_eligible(xa, C)
            y = ops.ConvNextLayer.apply(xa, *ps, mask)
            y.backward(cot)
            res[fused] = [y.detach()] + [xa.grad] + [p.grad for p in ps]
        ops._NO_CHAIN_LN = False
        with torch.no_grad():
            y_inf = ops.ConvNextLayer.apply(x, *[t.cuda() for t in base], mask, False)
    finally:
        ops._CHAIN_PAIR_MIN_ROWS = min_rows
        ops._NO_CHAIN_LN = no_ln
    for name, a, b in zip(['out', 'dx'] + names, res[True], res[False]):
        assert rel_err(a, b) < (3e-3 if name == 'out' else 8e-3), (name, rel_err(a, b))
    assert torch.equal(y_inf, res[True][0]), 'no-grad call must give the training call\'s output'


# ------------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('act', [False, True], ids=['ln', 'ln_gelu'])
@pytest.mark.parametrize('C', [33, 96, 194, 768, 1536])
def test_layernorm(C, act, dtype):
    ops = ops_mod()
    shape = (2, C, 5, 7)
    x = q(rnd(shape, 30, 2.0) + 0.5, dtype)
    g = 1 + rnd((C,), 31, 0.2)
    b = rnd((C,), 32, 0.2)
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = O.layer_norm_nchw(xr, gr, br)
    if act:
        ref = O.gelu(ref)
    cot = q(rnd(shape, 33), dtype)
    (ref * cot).sum().backward()
    xa = to_act(x, dtype, ld_extra=16).requires_grad_(True)
    gg, bg = g.float().cuda().requires_grad_(True), b.float().cuda().requires_grad_(True)
    y = ops.LayerNorm.apply(xa, gg, bg, act)
    close(from_act(y, C), ref.detach(), dtype, 'ln fwd')
    if y.shape[3] != C:
        assert float(y[..., C:].abs().max()) == 0.0
    y.backward(to_act(cot, dtype))
    close(from_act(xa.grad, C), xr.grad, dtype, 'ln dx')
    close(gg.grad, gr.grad, dtype, 'ln dgamma')
    close(bg.grad, br.grad, dtype, 'ln dbeta')


# ---------------------------------------------------------------------------------------------- resize / pool
@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('mode', [0, 1], ids=['bilinear', 'nearest'])
@pytest.mark.parametrize('case', recipe.RESIZE_CASES + ((32, 48, 8, 12), (7, 9, 7, 9)))
def test_resize(case, mode, dtype):
    ops = ops_mod()
    hi, wi, ho, wo = case
    x = q(rnd((2, 16, hi, wi), 40), dtype)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, size=(ho, wo), mode='bilinear' if mode == 0 else 'nearest')
    cot = q(rnd(tuple(ref.shape), 41), dtype)
    (ref * cot).sum().backward()
    xa = to_act(x, dtype, ld_extra=8).requires_grad_(True)
    y = ops.Resize.apply(xa, (ho, wo), mode)
    close(from_act(y, 16), ref.detach(), dtype, 'resize fwd')
    y.backward(to_act(cot, dtype))
    close(from_act(xa.grad, 16), xr.grad, dtype, 'resize bwd')


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('mode', [0, 1], ids=['bilinear', 'nearest'])
@pytest.mark.parametrize('case', [(16, 24, 64, 96), (12, 12, 96, 96), (10, 10, 48, 53), (18, 10, 60, 34)],
                         ids=['x4', 'x8', 'x4.8-x5.3', 'ragged-x3.3-x3.4'])
def test_resize_backward_large_ratio_two_pass(case, mode, dtype):
    """The necks resize levels 2 and 3 to level-0 size by x4 / x8 (upernext.py:191-195, fpn.py:138-142): at these ratios and map
    sizes the backward runs as two separable gathers through an fp32 workspace (vkas_resize_bwd_ws).  Against torch's own
    backward of F.interpolate; 96 channels in a 104-channel buffer (pixel stride > channels), ragged non-integer ratios too."""
    ops = ops_mod()
    from vkit_ocr_model_adaptive_scaling_amd._lib import lib
    hi, wi, ho, wo = case
    B, C = 4, 96
    assert lib.vkas_resize_bwd_ws_bytes(B, hi, wi, ho, wo, C) == B * ho * wi * C * 4  # the two-pass path is what runs here
    x = q(rnd((B, C, hi, wi), 50), dtype)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, size=(ho, wo), mode='bilinear' if mode == 0 else 'nearest')
    cot = q(rnd(tuple(ref.shape), 51), dtype)
    (ref * cot).sum().backward()
    xa = to_act(x, dtype, ld_extra=8).requires_grad_(True)
    y = ops.Resize.apply(xa, (ho, wo), mode)
    y.backward(to_act(cot, dtype))
    close(from_act(xa.grad, C), xr.grad, dtype, 'resize bwd (two-pass)')


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
def test_resize_matches_reference_goldens(dtype):
    """F.interpolate outputs stored by the golden generator (same cases the oracle is pinned on)."""
    ops = ops_mod()
    g = golden('ops')
    for (hi, wi, ho, wo) in recipe.RESIZE_CASES:
        a = torch.from_numpy(recipe.plain_tensor(7, (2, 3, hi, wi)))
        xa = to_act(a, torch.float32)
        for mode, name in ((0, 'bilinear'), (1, 'nearest')):
            y = ops.Resize.apply(xa, (ho, wo), mode)
            close(from_act(y, 3), torch.from_numpy(g[f'{name}_{hi}x{wi}_{ho}x{wo}']), torch.float32, name)
    for (hi, wi, s) in recipe.POOL_CASES:
        a = torch.from_numpy(recipe.plain_tensor(9, (2, 3, hi, wi)))
        y = ops.AdaptiveAvgPool.apply(to_act(a, torch.float32), s)
        close(from_act(y, 3), torch.from_numpy(g[f'avgpool_{hi}x{wi}_{s}']), torch.float32, 'avgpool')


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('mode', [0, 1], ids=['bilinear', 'nearest'])
def test_resize_add_inplace(mode, dtype):
    ops = ops_mod()
    dst = q(rnd((2, 8, 12, 20), 42), dtype)
    src = q(rnd((2, 8, 6, 10), 43), dtype)
    dr, sr = dst.clone().requires_grad_(True), src.clone().requires_grad_(True)
    ref = dr + F.interpolate(sr, size=(12, 20), mode='bilinear' if mode == 0 else 'nearest')
    cot = q(rnd((2, 8, 12, 20), 44), dtype)
    (ref * cot).sum().backward()
    da, sa = to_act(dst, dtype).requires_grad_(True), to_act(src, dtype).requires_grad_(True)
    y = ops.ResizeAdd.apply(da * 1, sa, mode)
    close(from_act(y, 8), ref.detach(), dtype, 'resize-add fwd')
    y.backward(to_act(cot, dtype))
    close(from_act(da.grad, 8), dr.grad, dtype, 'resize-add d dst')
    close(from_act(sa.grad, 8), sr.grad, dtype, 'resize-add d src')


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('case', recipe.POOL_CASES)
def test_adaptive_avgpool(case, dtype):
    ops = ops_mod()
    hi, wi, s = case
    x = q(rnd((2, 24, hi, wi), 45), dtype)
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.AdaptiveAvgPool2d(s)(xr)
    cot = q(rnd(tuple(ref.shape), 46), dtype)
    (ref * cot).sum().backward()
    xa = to_act(x, dtype).requires_grad_(True)
    y = ops.AdaptiveAvgPool.apply(xa, s)
    close(from_act(y, 24), ref.detach(), dtype, 'avgpool fwd')
    y.backward(to_act(cot, dtype))
    close(from_act(xa.grad, 24), xr.grad, dtype, 'avgpool bwd')


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
def test_adaptive_avgpools_one_node(dtype):
    """The PPM's four pooled branches as one autograd node (ops.AdaptiveAvgPools): outputs are those of the single-scale op and
    the input gradient is the sum of the branches' gradients (accumulated inside the kernels), also with an unused branch."""
    ops = ops_mod()
    scales = (1, 2, 3, 6)
    x = q(rnd((2, 24, 13, 10), 145), dtype)
    xr = x.clone().requires_grad_(True)
    refs = [torch.nn.AdaptiveAvgPool2d(s)(xr) for s in scales]
    cots = [q(rnd(tuple(r.shape), 146 + i), dtype) for i, r in enumerate(refs)]
    sum((r * c).sum() for r, c in zip(refs, cots)).backward()
    xa = to_act(x, dtype).requires_grad_(True)
    ys = ops.AdaptiveAvgPools.apply(xa, *scales)
    for y, r in zip(ys, refs):
        close(from_act(y, 24), r.detach(), dtype, 'avgpools fwd')
    torch.autograd.backward(ys, [to_act(c, dtype) for c in cots])
    close(from_act(xa.grad, 24), xr.grad, dtype, 'avgpools bwd')
    # a branch without gradient is skipped, not read
    xr2 = x.clone().requires_grad_(True)
    (torch.nn.AdaptiveAvgPool2d(3)(xr2) * cots[2]).sum().backward()
    xb = to_act(x, dtype).requires_grad_(True)
    yb = ops.AdaptiveAvgPools.apply(xb, *scales)
    yb[2].backward(to_act(cots[2], dtype))
    close(from_act(xb.grad, 24), xr2.grad, dtype, 'avgpools bwd, one branch')


def test_finalize_many_sums_partial_rows():
    """vkas_finalize_many: up to 8 second-stage column sums per launch (ops.finalize_many batches more), each over its own
    number of partial rows / columns / row pitch / first column, written or added onto the output."""
    ops = ops_mod()
    g = torch.Generator().manual_seed(77)
    items, expect = [], []
    for k in range(11):  # two launches
        P, n = int(torch.randint(1, 300, (1,), generator=g)), int(torch.randint(1, 200, (1,), generator=g))
        col0 = int(torch.randint(0, 40, (1,), generator=g))
        ld = col0 + n + int(torch.randint(0, 9, (1,), generator=g))
        ws = torch.randn(P, ld, generator=g).cuda()
        acc = k % 3 == 1
        out = torch.randn(n, generator=g).cuda() if acc else torch.full((n,), float('nan'), device='cuda')
        expect.append(ws[:, col0:col0 + n].double().sum(0) + (out.double() if acc else 0.0))
        items.append((ws.view(-1), col0, P, n, ld, out, acc))
    ops.finalize_many(items)
    for (_, _, P, n, _, out, _), ref in zip(items, expect):
        assert float((out.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (P, n)


def test_zero_arena_hands_out_clean_disjoint_slices():
    """ops.zeros_f32 / zero_arena_reset (the per-step scratch of the weight-gradient images): before the first reset every
    request is a fresh torch.zeros; after a reset requests are disjoint, zeroed slices of one buffer until it is exhausted, then
    fresh tensors again; a reset clears what the kernels left behind and grows the buffer to what the step asked for."""
    ops = ops_mod()
    dev = torch.device('cuda', 0)
    arena = ops._ZeroArena()
    old, ops._ZERO_ARENA = ops._ZERO_ARENA, arena
    try:
        a = ops.zeros_f32(1000, dev, True)
        assert arena.buf is None and float(a.abs().sum()) == 0.0
        ops.zero_arena_reset()
        assert arena.buf is not None and arena.buf.numel() == 1024
        b = ops.zeros_f32(1000, dev, True)
        assert b.data_ptr() == arena.buf.data_ptr() and float(b.abs().sum()) == 0.0
        c = ops.zeros_f32(100, dev, True)        # does not fit any more: a tensor of its own
        assert not (arena.buf.data_ptr() <= c.data_ptr() < arena.buf.data_ptr() + 4 * arena.buf.numel())
        d = ops.zeros_f32(10, dev, False)        # not step scratch: never from the arena
        assert not (arena.buf.data_ptr() <= d.data_ptr() < arena.buf.data_ptr() + 4 * arena.buf.numel())
        b.fill_(3.0)
        c.fill_(3.0)
        ops.zero_arena_reset()                   # asked for 1024 + 128 floats: the buffer grows, and is clean
        assert arena.buf.numel() == 1152 and float(arena.buf.abs().sum()) == 0.0
        e, f = ops.zeros_f32(1000, dev, True), ops.zeros_f32(100, dev, True)
        assert f.data_ptr() == e.data_ptr() + 4 * 1024
        e.fill_(1.0)
        assert float(f.abs().sum()) == 0.0
        ops.zero_arena_reset()
        assert float(arena.buf.abs().sum()) == 0.0
    finally:
        ops._ZERO_ARENA = old


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
def test_cat_and_tonchw(dtype):
    ops = ops_mod()
    a, b = q(rnd((2, 16, 6, 5), 47), dtype), q(rnd((2, 8, 6, 5), 48), dtype)
    aa, ba = to_act(a, dtype).requires_grad_(True), to_act(b, dtype, ld_extra=8).requires_grad_(True)
    cat = ops.Cat.apply(aa, ba)
    close(from_act(cat, 24), torch.cat([a, b], 1), dtype, 'cat')
    out = ops.ToNchw.apply(cat, 20)
    assert out.dtype == torch.float32 and tuple(out.shape) == (2, 20, 6, 5)
    close(out, torch.cat([a, b], 1)[:, :20], dtype, 'to nchw')
    cot = rnd((2, 20, 6, 5), 49).float().cuda()
    out.backward(cot)
    close(from_act(aa.grad, 16), cot[:, :16].double().cpu().to(dtype).double(), dtype, 'cat grad a')
    close(from_act(ba.grad, 8)[:, :4], cot[:, 16:20].double().cpu().to(dtype).double(), dtype, 'cat grad b')
    assert float(from_act(ba.grad, 8)[:, 4:].abs().max()) == 0.0


def test_softplus_tails():
    ops = ops_mod()
    g = golden('ops')
    t = torch.from_numpy(recipe.TAIL_POINTS * 6).float().cuda().requires_grad_(True)
    y = ops.Softplus.apply(t)
    assert np.allclose(y.detach().cpu().numpy(), g['softplus_tail'], rtol=2e-6, atol=1e-30)
    y.backward(torch.ones_like(y))
    tr = torch.from_numpy(recipe.TAIL_POINTS * 6).requires_grad_(True)
    torch.nn.Softplus()(tr).sum().backward()
    assert np.allclose(t.grad.cpu().numpy(), tr.grad.numpy(), rtol=2e-6, atol=1e-30)


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16', 'f16'])
def test_gelu_tails(dtype):
    """helper.gelu at the reference-generated tail points (-40 ... 40, golden `gelu_tail` = nn.GELU() of the imported
    reference), through the two kernels that apply it: LayerNorm(act_gelu) with gamma = 0, beta = points (the LN output is
    beta exactly) and the GELU epilogue of the implicit GEMM with zero weights, bias = points.  bf16 mode uses the
    polynomial Phi: |error| <= 5e-5 absolute plus the bf16 rounding of the result; the left tail must go to ~0, not
    grow with |x|."""
    ops = ops_mod()
    from vkit_ocr_model_adaptive_scaling_amd import _lib
    g = golden('ops')
    pts = recipe.TAIL_POINTS
    n = len(pts)
    Np = (n + 7) // 8 * 8
    rtol, atol = (2e-6, 1e-7) if dtype == torch.float32 else (4e-3, 1e-4)
    # LayerNorm(+GELU): y = gelu(0 * xhat + beta)
    x = to_act(rnd((2, n, 3, 5), 70), dtype)
    y = ops.LayerNorm.apply(x, torch.zeros(n, device='cuda'), torch.from_numpy(pts).float().cuda(), True)
    got = y[..., :n].double().cpu().reshape(-1, n)
    assert np.allclose(got.numpy(), np.broadcast_to(g['gelu_tail'], got.shape), rtol=rtol, atol=atol), got[0]
    assert float(got[:, :3].abs().max()) <= 5e-5, 'left tail must vanish'
    # GEMM epilogue: h = bf16/fp32(0 + bias), g = gelu(h)
    M = 16384 + 40  # MFMA tiles of both sizes incl. a ragged last tile
    xa = torch.zeros((1, 1, M, 8), dtype=dtype, device='cuda')
    Bw = torch.zeros((Np * 8,), dtype=dtype, device='cuda')
    bias = torch.zeros(Np, device='cuda')
    bias[:n] = torch.from_numpy(pts).float().cuda()
    h = torch.full((1, 1, M, Np), 7.0, dtype=dtype, device='cuda')
    gact = torch.full((1, 1, M, Np), 7.0, dtype=dtype, device='cuda')
    geom = _lib.ConvGeom(1, 1, M, 1, M, 8, 8, 1, 1, 1, 0)
    ops.conv_gemm(xa, geom, Bw, Np, h, _lib.EPI_GELU, bias=bias, out2=gact)
    hq = torch.from_numpy(pts).to(dtype).double()   # what the epilogue hands to GELU
    ref = O.gelu(hq).numpy()
    assert torch.equal(h[0, 0, :, :n].double().cpu(), hq.expand(M, n))
    got = gact[0, 0, :, :n].double().cpu().numpy()
    assert np.allclose(got, np.broadcast_to(ref, got.shape), rtol=rtol, atol=atol), got[0]
    assert np.abs(got[:, :3]).max() <= 5e-5
    if Np > n:
        assert float(gact[..., n:].abs().max()) == 0.0 and float(h[..., n:].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------- losses
PRIM_CASES = ['focal', 'focal_mask', 'focal_noalpha', 'dice', 'dice_mask', 'l1', 'l1_mask', 'smooth_l1', 'smooth_l1_mask',
              'l2', 'l2_mask']


@pytest.mark.parametrize('case', PRIM_CASES)
def test_primitive_losses_vs_oracle(case):
    """The exported primitive callables (loss_function/__init__.py:12-18) against the oracle's closed forms in fp64:
    value and d/dpred, with and without a mask, on a ragged (3, 1, 37, 53) map."""
    from vkit_ocr_model_adaptive_scaling_amd import loss_function as L
    shape = (3, 1, 37, 53)
    x = rnd(shape, 80, 2.0)
    t01 = (rnd(shape, 81) > 0.3).double()
    tr = rnd(shape, 82)
    m = (rnd(shape, 83) > -0.2).double() if case.endswith('_mask') else None
    kind = case.replace('_mask', '')
    xr = x.clone().requires_grad_(True)
    if kind in ('focal', 'focal_noalpha'):
        alpha = 0.25 if kind == 'focal' else -1.0
        fn, gt = L.FocalWithLogitsLossFunction(alpha=alpha, gamma=2), t01
        p = torch.sigmoid(xr)
        ce = torch.clamp(xr, min=0) - xr * gt + torch.log1p(torch.exp(-xr.abs()))
        e = ce * (1 - (p * gt + (1 - p) * (1 - gt))) ** 2
        if alpha >= 0:
            e = (alpha * gt + (1 - alpha) * (1 - gt)) * e
        ref = e.mean() if m is None else (e * m).sum() / (m.sum() + 1e-6)
        if m is None and alpha >= 0:
            assert abs(float(ref) - float(O.sigmoid_focal_mean(x, gt))) < 1e-12
    elif kind == 'dice':
        fn, gt = L.DiceLossFunction(), t01
        xr = torch.sigmoid(x).clone().requires_grad_(True)  # dice takes probabilities
        ref = O.dice(xr, gt) if m is None else O.dice(xr * m, gt * m)
    elif kind == 'l1':
        fn, gt = L.L1LossFunction(), tr
        e = (xr - gt).abs()
        ref = e.mean() if m is None else (e * m).sum() / (m.sum() + 1e-6)
    elif kind == 'smooth_l1':
        fn, gt = L.L1LossFunction(smooth=True, smooth_beta=2.5), tr
        ref = O.smooth_l1(xr, gt, 2.5, m)
    else:
        fn, gt = L.L2LossFunction(), tr
        ref = O.l2(xr, gt, m)
    ref.backward()
    xg = xr.detach().float().cuda().requires_grad_(True)
    gt_g = gt.float().cuda()
    gt_before = gt_g.clone()
    out = fn(xg, gt_g, None if m is None else m.float().cuda())
    assert out.shape == () and out.dtype == torch.float32
    assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (float(out), float(ref))
    out.backward()
    assert rel_err(xg.grad, xr.grad) < 2e-5
    assert torch.equal(gt_g, gt_before), 'targets must not be modified (dice.py:28-30 aliasing is not replicated)'


@pytest.mark.parametrize('hard', [False, True], ids=['soft', 'hard'])
def test_cross_entropy_vs_oracle(hard):
    from vkit_ocr_model_adaptive_scaling_amd import loss_function as L
    B, C, P = 3, 4, 29
    x = rnd((B, C, P), 90, 3.0)
    xr = x.clone().requires_grad_(True)
    if hard:
        gt = torch.randint(0, C, (B, P), generator=torch.Generator().manual_seed(91))
        ref = F.cross_entropy(xr, gt)
        gt_g = gt.cuda()
    else:
        gt = torch.softmax(rnd((B, C, P), 92), dim=1)
        ref = O.soft_cross_entropy(xr, gt)
        gt_g = gt.float().cuda()
    ref.backward()
    xg = x.float().cuda().requires_grad_(True)
    out = L.CrossEntropyWithLogitsLossFunction()(xg, gt_g)
    assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    out.backward()
    assert rel_err(xg.grad, xr.grad) < 2e-5
    if hard:
        with pytest.raises(IndexError):
            L.CrossEntropyWithLogitsLossFunction()(xg, torch.full((B, P), C, device='cuda'))


def test_loss_argument_rejection_on_gpu():
    """Mismatched batch / shapes / label points outside the map raise before any kernel indexes with them
    (the reference raises a broadcast / index error in these cases)."""
    from vkit_ocr_model_adaptive_scaling_amd import loss_function as L
    B, H, W, P = 2, 24, 28, 5
    box = L.Box(up=2, down=H - 3, left=2, right=W - 3)
    ch, cw = H - 4, W - 4
    z = lambda *s: torch.zeros(*s, device='cuda')
    rough = L.AdaptiveScalingRoughLossFunction(L.AdaptiveScalingRoughLossFunctionConifg())
    with pytest.raises((ValueError, AssertionError)):
        rough(z(B, 1, H, W), z(B, 1, H, W), z(B + 1, ch, cw), z(B + 1, ch, cw), (H, W), box)
    with pytest.raises(RuntimeError):
        rough(z(B, 1, H, W), z(B, 1, H, W), torch.zeros(B, ch, cw), z(B, ch, cw), (H, W), box)  # CPU labels
    precise = L.AdaptiveScalingPreciseLossFunction(L.AdaptiveScalingPreciseLossFunctionConifg())
    args = lambda py, px, off=None: (None, z(B, 1, H, W), z(B, 2, H, W), z(B, 4, H, W), z(B, 4, H, W), z(B, ch, cw),
                                     z(B, ch, cw), (H, W), box, py, px, z(B, P, 2) if off is None else off, z(B, P, 4),
                                     z(B, P, 3))
    ok = torch.zeros(B, P, dtype=torch.long, device='cuda')
    ops = ops_mod()
    float(precise(*args(ok, ok)))
    ops.check_deferred(wait=True)
    # label points outside the map: checked on the device, reported without stalling the step (at the next loss call or
    # by check_deferred); the kernel itself clamps, so nothing is read out of bounds meanwhile
    for bad in ((ok + H, ok), (ok, ok - 1)):
        float(precise(*args(*bad)))
        with pytest.raises(IndexError):
            ops.check_deferred(wait=True)
    precise(*args(ok, ok + W))
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        precise(*args(ok, ok))  # the next call reports the previous batch
    with pytest.raises(ValueError):
        precise(*args(ok[:1], ok[:1]))
    with pytest.raises(ValueError):
        precise(*args(ok, ok, z(B, P + 1, 2)))
    with pytest.raises(RuntimeError):
        precise(*args(ok.cpu(), ok))


@pytest.mark.parametrize('variant', ['plain', 'edge'])
def test_losses_vs_reference_goldens(variant):
    """Fused loss kernels vs the reference's own loss classes (goldens) incl. the masked-out / empty-mask edge cases."""
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import (
        Box, AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg,
        AdaptiveScalingPreciseLossFunction, AdaptiveScalingPreciseLossFunctionConifg)
    L = recipe.LOSS_TOY
    g = golden('losses')
    t = {k: torch.from_numpy(v) for k, v in recipe.loss_inputs(L, variant).items()}
    c = {k: (v.float().cuda() if v.dtype == torch.float64 else v.cuda()) for k, v in t.items()}
    box = Box(*L['core_box'])
    mf, hf = c['mask_feat'].requires_grad_(True), c['height_feat'].requires_grad_(True)
    rl = AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg())(
        mf, hf, c['gt_mask'], c['gt_score_rough'], L['shape'], box)
    rl.backward()
    assert abs(float(rl) - float(g[f'{variant}/rough_loss'])) < 2e-5 * abs(float(g[f'{variant}/rough_loss']))
    assert rel_err(mf.grad, g[f'{variant}/g_mask_feat']) < 1e-4
    assert rel_err(hf.grad, g[f'{variant}/g_height_feat']) < 1e-4
    p = {k: c[k].requires_grad_(True) for k in ('prob', 'offset', 'angle', 'dist')}
    pl = AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg())(
        None, p['prob'], p['offset'], p['angle'], p['dist'], c['gt_score_precise'], c['gt_mask'], L['shape'], box,
        c['py'], c['px'], c['gt_offsets'], c['gt_angles'], c['gt_dists'])
    pl.backward()
    assert abs(float(pl) - float(g[f'{variant}/precise_loss'])) < 2e-5 * abs(float(g[f'{variant}/precise_loss']))
    for k, v in p.items():
        assert rel_err(v.grad, g[f'{variant}/g_{k}']) < 1e-4, k


def test_precise_loss_duplicate_points_accumulate():
    """Two label points on the same pixel: gradients must add (atomic scatter), as advanced indexing backward does."""
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import (Box, AdaptiveScalingPreciseLossFunction,
                                                                   AdaptiveScalingPreciseLossFunctionConifg)
    B, H, W, P = 1, 16, 16, 4
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: torch.randn(*s, generator=g)
    prob, offset, angle, dist = mk(B, 1, H, W), mk(B, 2, H, W), mk(B, 4, H, W), mk(B, 4, H, W).abs() + 0.1
    py = torch.tensor([[5, 5, 7, 5]])
    px = torch.tensor([[6, 6, 2, 6]])
    gs, gm = torch.rand(B, 12, 12, generator=g), (torch.rand(B, 12, 12, generator=g) > 0.5).float()
    go, ga, gd = mk(B, P, 2), torch.softmax(mk(B, P, 4), -1), torch.rand(B, P, 3, generator=g)
    ref_in = [t.double().requires_grad_(True) for t in (prob, offset, angle, dist)]
    ref = O.precise_loss(*ref_in, gs.double(), gm.double(), (2, 13, 2, 13), py, px, go.double(), ga.double(), gd.double())
    ref.backward()
    dev_in = [t.cuda().requires_grad_(True) for t in (prob, offset, angle, dist)]
    out = AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg())(
        None, *dev_in, gs.cuda(), gm.cuda(), (H, W), Box(2, 13, 2, 13), py.cuda(), px.cuda(), go.cuda(), ga.cuda(), gd.cuda())
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-5 * abs(float(ref))
    for a, b in zip(dev_in, ref_in):
        assert rel_err(a.grad, b.grad) < 1e-4


# ------------------------------------------------------------------------------------------------- optimizer
def test_clip_adamw_matches_torch():
    import ctypes
    from vkit_ocr_model_adaptive_scaling_amd.training.optimizer import FlatAdamW
    torch.manual_seed(0)
    shapes = [(96, 3, 4, 4), (96,), (384, 96), (1, 1), (33, 7)]
    ref_params = [torch.nn.Parameter(torch.randn(s, dtype=torch.float64) * 0.1) for s in shapes]
    opt_ref = torch.optim.AdamW(ref_params, lr=8e-4, betas=(0.9, 0.999), weight_decay=0.01)
    params = [torch.nn.Parameter(p.detach().float().cuda()) for p in ref_params]
    opt = FlatAdamW(params, lr=8e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=2.5)
    for step in range(3):
        for i, (p, r) in enumerate(zip(params, ref_params)):
            gr = torch.randn(r.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(100 * step + i)) * (3.0 if step == 1 else 0.01)
            r.grad = gr.clone()
            p.grad.copy_(gr.float().cuda())
        torch.nn.utils.clip_grad_norm_(ref_params, 2.5)
        opt_ref.step()
        opt.step(lr=8e-4)
        opt.zero_grad()
    for p, r in zip(params, ref_params):
        assert rel_err(p.detach(), r.detach()) < 2e-6


# ------------------------------------------------------------------------------------------ fused head tail
def test_adamw_skips_parameters_without_gradient():
    """torch.optim.AdamW leaves parameters whose .grad is None untouched (no weight decay, no moment update) - e.g. the
    precise mask head under precise_enable_char_mask_head, which forward_precise never runs."""
    from vkit_ocr_model_adaptive_scaling_amd.training import FlatAdamW, FlatBuffers
    torch.manual_seed(5)
    mk = lambda: torch.nn.ModuleDict({'a': torch.nn.Linear(16, 8), 'unused': torch.nn.Linear(8, 8), 'b': torch.nn.Linear(8, 4)})
    m, r = mk().cuda(), mk().cuda()
    r.load_state_dict(m.state_dict())
    fb = FlatBuffers(m.named_parameters())
    opt = FlatAdamW(None, lr=1e-2, weight_decay=0.1, max_grad_norm=1.0, flat=fb)
    ropt = torch.optim.AdamW(r.parameters(), lr=1e-2, weight_decay=0.1)
    x = torch.randn(5, 16, device='cuda')
    for _ in range(3):
        m['b'](m['a'](x)).pow(2).sum().backward()
        opt.step()
        opt.zero_grad()
        r['b'](r['a'](x)).pow(2).sum().backward()
        torch.nn.utils.clip_grad_norm_(r.parameters(), 1.0)
        ropt.step()
        ropt.zero_grad(set_to_none=True)
    for (n, p), (_, q) in zip(m.named_parameters(), r.named_parameters()):
        assert rel_err(p, q) < 2e-6, n
    s0, k0 = fb.offsets['unused.weight']
    assert float(opt.exp_avg[s0:s0 + k0].abs().max()) == 0.0


@pytest.mark.parametrize('hw', [(128, 136), (66, 256)], ids=['generic', 'rowslab'])
@pytest.mark.parametrize('chans', [((40, 33), (1, 4)), ((192, 193, 194, 194), (1, 2, 4, 4)), ((96,), (3,)),
                                   ((256, 257, 258, 258), (1, 2, 4, 4)), ((384, 386), (1, 4))])  # Base / Large widths
def test_heads_fused_matches_unfused_and_fp64(chans, hw):
    """Conv3x3 + per-head LayerNorm + GELU + Linear(C -> oc) fused into the GEMM epilogue (bf16) vs the same math in fp64
    on the host, forward and every gradient (input, conv weight / bias, gamma, beta, projection weight / bias)."""
    ops = ops_mod()
    cs, ocs = chans
    B, Cin = 1, 64
    H, W = hw
    dtype = torch.bfloat16
    x = q(rnd((B, Cin, H, W), 50), dtype)
    convs = [(q(rnd((c, Cin, 3, 3), 51 + i, 1.0 / math.sqrt(Cin * 9)), dtype), rnd((c,), 61 + i, 0.1)) for i, c in enumerate(cs)]
    tails = [(1 + rnd((c,), 71 + i, 0.1), rnd((c,), 81 + i, 0.1), rnd((oc, c), 91 + i, 1.0 / math.sqrt(c)), rnd((oc,), 101 + i, 0.1))
             for i, (c, oc) in enumerate(zip(cs, ocs))]
    # fp64 reference
    xr = x.clone().requires_grad_(True)
    ref_params, ref_outs = [], []
    for (w, b), (g, bt, wp, bp) in zip(convs, tails):
        ps = [t.clone().requires_grad_(True) for t in (w, b, g, bt, wp, bp)]
        ref_params.append(ps)
        zz = F.conv2d(xr, ps[0], ps[1], padding=1)
        a = O.gelu(O.layer_norm_nchw(zz, ps[2], ps[3]))
        ref_outs.append(O.linear_nchw(a, ps[4], ps[5]))
    cots = [rnd(tuple(o.shape), 111 + i) for i, o in enumerate(ref_outs)]
    sum((o * c).sum() for o, c in zip(ref_outs, cots)).backward()
    # fused op
    xa = to_act(x, dtype).requires_grad_(True)
    dev = [[t.float().cuda().requires_grad_(True) for t in (w, b, g, bt, wp, bp)] for (w, b), (g, bt, wp, bp) in zip(convs, tails)]
    fused = [t for head in dev for t in head]  # per head: conv weight, conv bias, gamma, beta, wproj, bproj
    assert ops.HeadsFused.eligible(xa, cs, ocs)
    outs = ops.HeadsFused.apply(xa, True, *fused)
    loss = 0
    for o, oc, r, c in zip(outs, ocs, ref_outs, cots):
        assert tuple(o.shape) == (B, H, W, 8) and o.dtype == torch.float32
        assert float(o[..., oc:].abs().max()) == 0.0
        got = o[..., :oc].permute(0, 3, 1, 2)
        assert rel_err(got, r.detach()) < 8e-3, rel_err(got, r.detach())
        loss = loss + (got * c.float().cuda()).sum()
    loss.backward()
    assert rel_err(from_act(xa.grad, Cin), xr.grad) < 1.5e-2
    names = ('conv w', 'conv b', 'gamma', 'beta', 'proj w', 'proj b')
    for ps, rs in zip(dev, ref_params):
        for n, p, r in zip(names, ps, rs):
            e = rel_err(p.grad, r.grad)
            assert e < 2e-2, (n, e)
