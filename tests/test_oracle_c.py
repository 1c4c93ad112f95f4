"""The plain-C oracle (oracle/c/vkas_oracle.c) against the reference's fixtures and against the torch oracle: two
independently written restatements must agree to fp64 rounding.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import torch_oracle as O
from tests.golden import recipe
from tests.helpers import golden, rel_err


def t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def test_c_oracle_matches_reference_op_goldens():
    g = golden('ops')
    for (hi, wi, ho, wo) in recipe.RESIZE_CASES:
        a = recipe.plain_tensor(7, (2, 3, hi, wi))
        assert rel_err(t(C.bilinear(a, (ho, wo))), g[f'bilinear_{hi}x{wi}_{ho}x{wo}']) < 1e-12
        assert rel_err(t(C.nearest(a, (ho, wo))), g[f'nearest_{hi}x{wi}_{ho}x{wo}']) == 0.0
    for (hi, wi, s) in recipe.POOL_CASES:
        a = recipe.plain_tensor(9, (2, 3, hi, wi))
        assert rel_err(t(C.adaptive_avgpool(a, s)), g[f'avgpool_{hi}x{wi}_{s}']) < 1e-12
    assert np.allclose(C.gelu(recipe.TAIL_POINTS), g['gelu_tail'], rtol=1e-12, atol=1e-300)
    assert np.allclose(C.softplus(recipe.TAIL_POINTS * 6), g['softplus_tail'], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize('case', [(2, 5, 9, 11, 7, 3, 1, 1), (1, 3, 16, 24, 8, 4, 4, 0), (2, 6, 8, 8, 4, 2, 2, 0),
                                  (1, 12, 7, 5, 9, 1, 1, 0)])
def test_c_conv_vs_torch(case):
    B, Cin, H, W, N, K, stride, pad = case
    x, w, b = recipe.plain_tensor(1, (B, Cin, H, W)), recipe.plain_tensor(2, (N, Cin, K, K)), recipe.plain_tensor(3, (N,))
    ref = torch.nn.functional.conv2d(t(x), t(w), t(b), stride=stride, padding=pad)
    assert rel_err(t(C.conv2d(x, w, b, stride, pad)), ref) < 1e-13


def test_c_convnext_layer_vs_torch_oracle():
    Cn, shape = 12, (2, 12, 9, 7)
    mk = lambda s, sc, seed: recipe.plain_tensor(seed, s) * sc
    sd = {'block.0.weight': mk((Cn, 1, 7, 7), 0.2, 10), 'block.0.bias': mk((Cn,), 0.1, 11),
          'block.2.weight': 1 + mk((Cn,), 0.1, 12), 'block.2.bias': mk((Cn,), 0.1, 13),
          'block.3.weight': mk((4 * Cn, Cn), 0.3, 14), 'block.3.bias': mk((4 * Cn,), 0.1, 15),
          'block.5.weight': mk((Cn, 4 * Cn), 0.2, 16), 'block.5.bias': mk((Cn,), 0.1, 17),
          'block_scale': 1 + mk((Cn, 1, 1), 0.2, 18)}
    x = mk(shape, 1.0, 19)
    mask = np.array([1.25, 0.0])
    ref = O.convnext_layer({k: t(v) for k, v in sd.items()}, '', t(x), t(mask).view(-1, 1, 1, 1))
    assert rel_err(t(C.convnext_layer(x, sd, mask)), ref) < 1e-12


def test_c_rough_loss_vs_reference_golden():
    L = recipe.LOSS_TOY
    g = golden('losses')
    up, down, left, right = L['core_box']
    for variant in ('plain', 'edge'):
        d = recipe.loss_inputs(L, variant)
        m = d['mask_feat'][:, 0, up:down + 1, left:right + 1]
        h = d['height_feat'][:, 0, up:down + 1, left:right + 1]
        val = C.rough_loss(m, h, d['gt_mask'], d['gt_score_rough'])
        assert abs(val - float(g[f'{variant}/rough_loss'])) < 1e-8 * abs(val)
