"""Input recipes shared by the golden generator (make_golden.py) and the tests that consume the
fixtures.  Everything is derived from portable seeds (no stored inputs, no reference code)."""
import numpy as np

from vkit_ocr_model_adaptive_scaling_amd.utils import portable_rng as prng

CONVNEXT_TOY = dict(seed=11, std=0.15, plan=((16, 2), (32, 2), (64, 3), (128, 2)), shape=(2, 3, 64, 96))
CONVNEXT_TOY_P2 = dict(seed=12, std=0.15, plan=((16, 1), (32, 1), (64, 2), (128, 1)), shape=(1, 3, 32, 64))
NECK_TOY = dict(seed=21, std=0.2, in_channels_group=(16, 32, 64, 128), out_channels=64, batch=2, base_hw=(24, 40))
HEAD_TOY = dict(seed=31, std=0.1, in_channels=64, batch=2, hw=(12, 20))
HEAD_CASES = ((1, 2, 8.0), (2, 2, 0.0), (4, 2, 0.0), (4, 1, 0.0))  # (out_channels, upsampling_factor, init bias)
RESIZE_CASES = ((6, 6, 32, 32), (3, 3, 32, 32), (1, 1, 32, 32), (2, 2, 32, 32), (8, 12, 16, 24), (5, 7, 12, 20),
                (3, 5, 6, 10), (16, 16, 32, 32))
POOL_CASES = ((20, 20, 3), (20, 20, 6), (32, 32, 3), (32, 32, 1), (3, 5, 6), (8, 8, 2), (10, 16, 6))
TAIL_POINTS = np.array([-40.0, -9.0, -6.0, -3.0, -1.0, -0.5, -1e-3, 0.0, 1e-3, 0.5, 1.0, 3.0, 3.3333, 3.4, 6.0, 9.0, 40.0])
# the reference's own test widths that are not multiples of 8 (tests/test_fpn.py:16-28: out_channels=400 -> 4 x 100)
WIDTH400 = dict(seed=61, std=0.08, head_std=0.05, in_channels_group=(96, 192, 384, 768), out_channels=400, batch=1,
                base_hw=(16, 24))
LOSS_TOY = dict(seed=41, batch=2, shape=(80, 72), core_box=(10, 69, 6, 65), points=20)
FULL_MODEL = dict(seed=51, std=0.05, image=(1, 3, 256, 256), down_shape=(128, 128), core_box=(10, 117, 10, 117),
                  points=50)


def sample_indices(n: int, k: int = 32) -> np.ndarray:
    """Deterministic strided sample positions used for gradient summaries."""
    if n <= k:
        return np.arange(n)
    return (np.arange(k) * (n - 1)) // (k - 1)


def image(seed: int, shape) -> np.ndarray:
    """Raw 0..255 pixel values as float (tests/test_adaptive_scaling.py:126 of the reference)."""
    n = int(np.prod(shape))
    return prng.integers(seed, prng.key_from_name('image'), n, 0, 256).astype(np.float64).reshape(shape)


def plain_tensor(seed: int, shape, name: str = 'plain') -> np.ndarray:
    n = int(np.prod(shape))
    return prng.normal_like(seed, prng.key_from_name(name), n).reshape(shape)


def cotangent(seed: int, idx: int, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return prng.normal_like(seed, prng.key_from_name(f'cot{idx}'), n).reshape(shape)


def neck_features(n):
    feats = []
    h, w = n['base_hw']
    for i, c in enumerate(n['in_channels_group']):
        feats.append(plain_tensor(n['seed'], (n['batch'], c, h >> i, w >> i), f'feat{i}'))
    return feats


def head_input(h):
    return plain_tensor(h['seed'], (h['batch'], h['in_channels'], *h['hw']), 'head_in')


def _loss_targets(seed, batch, core_hw, full_hw, core_box, points):
    ch, cw = core_hw
    u = lambda name, shape: prng.uniform(seed, prng.key_from_name(name), int(np.prod(shape))).reshape(shape)
    up, down, left, right = core_box
    ang = u('angles', (batch, points, 4)) + 0.05
    return dict(
        gt_mask=(u('gt_mask', (batch, ch, cw)) > 0.5).astype(np.float64),
        gt_score_rough=u('gt_score_rough', (batch, ch, cw)) + 8.75,
        gt_score_precise=u('gt_score_precise', (batch, ch, cw)),
        py=prng.integers(seed, prng.key_from_name('py'), batch * points, up, down + 1).reshape(batch, points),
        px=prng.integers(seed, prng.key_from_name('px'), batch * points, left, right + 1).reshape(batch, points),
        gt_offsets=prng.integers(seed, prng.key_from_name('offs'), batch * points * 2, -20, 21).astype(
            np.float64).reshape(batch, points, 2),
        gt_angles=ang / ang.sum(axis=-1, keepdims=True),
        gt_dists=u('dists', (batch, points, 3)),
    )


def loss_inputs(L, variant: str):
    """Predictions + targets for the loss-only goldens.  ``edge`` adds: height predictions <= 1.1 (masked
    out before the clamp, loss_function/adaptive_scaling.py:112-114), score map values <= 1.1, large
    |logits|, and an image whose mask is all zero (sum/(0+eps) denominators)."""
    b = L['batch']
    h, w = L['shape']
    up, down, left, right = L['core_box']
    ch, cw = down - up + 1, right - left + 1
    seed = L['seed'] + (0 if variant == 'plain' else 1000)
    t = _loss_targets(seed, b, (ch, cw), (h, w), L['core_box'], L['points'])
    t['mask_feat'] = 2.0 * plain_tensor(seed, (b, 1, h, w), 'mask_feat')
    t['height_feat'] = 8.0 + 3.0 * plain_tensor(seed, (b, 1, h, w), 'height_feat')
    t['prob'] = 2.0 * plain_tensor(seed, (b, 1, h, w), 'prob')
    t['offset'] = 6.0 * plain_tensor(seed, (b, 2, h, w), 'offset')
    t['angle'] = 2.0 * plain_tensor(seed, (b, 4, h, w), 'angle')
    t['dist'] = np.abs(3.0 * plain_tensor(seed, (b, 4, h, w), 'dist')) + 0.01
    if variant == 'edge':
        t['height_feat'] = 1.0 + 0.5 * np.abs(plain_tensor(seed, (b, 1, h, w), 'height_feat'))  # many <= 1.1
        t['gt_score_rough'] = t['gt_score_rough'] - 8.0  # in [0.75, 1.75): many <= 1.1
        t['mask_feat'] = 25.0 * plain_tensor(seed, (b, 1, h, w), 'mask_feat')  # saturating logits
        t['gt_mask'][1] = 0.0
    return t


def full_model_inputs(Fm):
    seed = Fm['seed']
    up, down, left, right = Fm['core_box']
    ch, cw = down - up + 1, right - left + 1
    b = Fm['image'][0]
    t = _loss_targets(seed, b, (ch, cw), Fm['down_shape'], Fm['core_box'], Fm['points'])
    out = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in t.items()}
    out['image_rough'] = image(seed, Fm['image']).astype(np.float32)
    out['image_precise'] = image(seed + 1, Fm['image']).astype(np.float32)
    return out
