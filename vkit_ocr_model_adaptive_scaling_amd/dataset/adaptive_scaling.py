"""Batch side of ``vkit_open_model.dataset.adaptive_scaling`` (dataset/adaptive_scaling.py:39-66,282-368).

The reference draws its samples from a third-party synthetic-data pipeline (``vkit.pipeline``, absent here: SURVEY 8f); what
the hot path consumes is the COLLATED batch, whose schema this module reproduces from plain numpy records:

    batch['rough']   image (B,3,H,W) f32 0..255 | downsampled_mask (B,h,w) f32 | downsampled_score_map (B,h,w) f32
                     | downsampled_shape (h_full, w_full) | downsampled_core_box | rng_states
    batch['precise'] the same + downsampled_label_point_y / _x (B,P) i64 | up_left_offsets (B,P,2) i64
                     | corner_angles (B,P,4) f32 | corner_distances (B,P,3) f32

``SyntheticAdaptiveScalingIterableDataset`` yields seeded random records of that shape (the role ``bench.py`` and the tests
need a loader for); real data plugs in by yielding ``(RoughSample, PreciseSample)`` records from any other source."""
from typing import Any, Dict, Iterable, Iterator, List, Mapping, Optional, Sequence, Tuple

import attrs
import numpy as np
import torch
from torch.utils.data import IterableDataset, get_worker_info

from ..loss_function import Box


@attrs.define
class RoughSample:
    """dataset/adaptive_scaling.py:39-47 with the vkit element types replaced by their arrays."""
    image: np.ndarray                     # (H, W, 3) uint8
    downsampled_shape: Tuple[int, int]    # feature-map size the core box refers to
    downsampled_core_box: Any             # .up / .down / .left / .right, inclusive
    downsampled_mask: np.ndarray          # (h, w) bool / uint8, the core box's extent
    downsampled_score_map: np.ndarray     # (h, w) float32
    rng_state: Optional[Mapping] = None


@attrs.define
class PreciseSample:
    """dataset/adaptive_scaling.py:50-66; the per-character regression labels as arrays over the P label points."""
    image: np.ndarray
    downsampled_shape: Tuple[int, int]
    downsampled_core_box: Any
    downsampled_mask: np.ndarray
    downsampled_score_map: np.ndarray
    downsampled_label_point_y: np.ndarray  # (P,) int64
    downsampled_label_point_x: np.ndarray  # (P,) int64
    up_left_offsets: np.ndarray            # (P, 2) int64, (y, x)
    corner_angles: np.ndarray              # (P, 4) float32, a distribution over the four clockwise angles
    corner_distances: np.ndarray           # (P, 3) float32, the up-left distance trimmed
    rng_state: Optional[Mapping] = None


def _stack(arrays: Sequence[np.ndarray], dtype) -> torch.Tensor:
    out = torch.from_numpy(np.stack([np.ascontiguousarray(a, dtype=dtype) for a in arrays]))
    return out


def adaptive_scaling_dataset_collate_fn(batch: Iterable[Tuple[RoughSample, PreciseSample]]) -> Dict[str, Dict[str, Any]]:
    """dataset/adaptive_scaling.py:282-368: images (H,W,3) -> (3,H,W) float32, masks float32, the batch-wide shape / box
    taken from the last sample (all samples of a batch share them), rng states listed."""
    pairs = list(batch)
    if not pairs:
        raise ValueError('empty batch')
    rough = [r for r, _ in pairs]
    precise = [p for _, p in pairs]

    def common(samples) -> Dict[str, Any]:
        return {'image': _stack([s.image.transpose(2, 0, 1) for s in samples], np.float32),
                'downsampled_mask': _stack([s.downsampled_mask for s in samples], np.float32),
                'downsampled_score_map': _stack([s.downsampled_score_map for s in samples], np.float32)}

    r = common(rough)
    p = common(precise)
    p['downsampled_label_point_y'] = _stack([s.downsampled_label_point_y for s in precise], np.int64)
    p['downsampled_label_point_x'] = _stack([s.downsampled_label_point_x for s in precise], np.int64)
    p['up_left_offsets'] = _stack([s.up_left_offsets for s in precise], np.int64)
    p['corner_angles'] = _stack([s.corner_angles for s in precise], np.float32)
    p['corner_distances'] = _stack([s.corner_distances for s in precise], np.float32)
    for out, samples in ((r, rough), (p, precise)):
        out['downsampled_shape'] = samples[-1].downsampled_shape
        out['downsampled_core_box'] = samples[-1].downsampled_core_box
        out['rng_states'] = [s.rng_state for s in samples]
    return {'rough': r, 'precise': p}


class SyntheticAdaptiveScalingIterableDataset(IterableDataset):
    """``num_samples`` seeded random (RoughSample, PreciseSample) pairs of one image size: uniform pixels, a blob-free random
    mask / score map over the core box (the feature map minus ``margin`` pixels per side), ``num_label_points`` label points
    inside it (train.py:58: 200).  Sample i is the same whatever the worker layout."""

    def __init__(self, num_samples: int, image_hw: Tuple[int, int] = (1024, 1024), downsample: int = 2, margin: int = 10,
                 num_label_points: int = 200, rng_seed: int = 13371):
        super().__init__()
        H, W = image_hw
        if H % 32 or W % 32:
            raise ValueError('image sides must be multiples of 32')
        self.num_samples, self.image_hw, self.rng_seed = num_samples, (H, W), rng_seed
        self.down_hw = (H // downsample, W // downsample)
        dh, dw = self.down_hw
        if dh <= 2 * margin + 1 or dw <= 2 * margin + 1:
            raise ValueError('margin leaves no core box')
        self.box = Box(up=margin, down=dh - margin - 1, left=margin, right=dw - margin - 1)
        self.num_label_points = num_label_points

    def __len__(self):
        return self.num_samples

    def sample(self, index: int) -> Tuple[RoughSample, PreciseSample]:
        rng = np.random.default_rng([self.rng_seed, index])
        H, W = self.image_hw
        box = self.box
        ch, cw = box.down - box.up + 1, box.right - box.left + 1
        P = self.num_label_points

        def maps():
            return (rng.integers(0, 256, (H, W, 3), dtype=np.uint8), rng.random((ch, cw)) > 0.7,
                    rng.random((ch, cw), dtype=np.float32))
        img, mask, score = maps()
        rough = RoughSample(img, self.down_hw, box, mask, score * 40.0, {'seed': self.rng_seed, 'index': index})
        img, mask, score = maps()
        angles = rng.random((P, 4), dtype=np.float32) + 0.1
        precise = PreciseSample(
            img, self.down_hw, box, mask, score,
            rng.integers(box.up, box.down + 1, P), rng.integers(box.left, box.right + 1, P),
            rng.integers(-20, 21, (P, 2)), angles / angles.sum(1, keepdims=True),
            rng.random((P, 3), dtype=np.float32) * 30.0, {'seed': self.rng_seed, 'index': index})
        return rough, precise

    def __iter__(self) -> Iterator[Tuple[RoughSample, PreciseSample]]:
        info = get_worker_info()
        first, step = (0, 1) if info is None else (info.id, info.num_workers)
        for i in range(first, self.num_samples, step):
            yield self.sample(i)
