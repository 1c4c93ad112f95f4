"""Tensor side of the adaptive-scaling inference path (mirror of vkit_open_model/inferencing/adaptive_scaling.py:92-188,
295-396 and the shape rules of :95-107): pad-to-32, the short-side-720 rule, the no-grad model calls and the
sigmoid / threshold / softmax / padding-mask post-processing, which runs on the device (csrc/infer.hip).

Out of scope: the CPU geometry around it - polygons from masks, text-region flattening / stacking,
polygon building (vkit, cv2, scipy; third-party code that is absent here).  Images are plain (H, W, 3) uint8 arrays
instead of ``vkit.element.Image``; results carry numpy arrays instead of ``Mask`` / ``ScoreMap``.  The reference loads
a TorchScript file (``model_jit``, :85-90); so does this mirror (``torch.jit.save`` of ``torch.jit.script(model)``, see
model/scripting.py), and it also takes the scripted or the eager ``AdaptiveScaling`` module itself, or a state-dict file in
the reference's ``RestoreState`` schema.
"""
import ctypes
import math
from typing import Optional, Sequence, Tuple, Union

import attrs
import numpy as np
import torch

from .opt import pad_mat_to_make_divisible
from .graphs import GraphCache, param_stamp
from .. import ops
from .._lib import lib, check
from ..model import AdaptiveScaling, AdaptiveScalingConfig


@attrs.define
class AdaptiveScalingInferencingConfig:
    """inferencing/adaptive_scaling.py:41-58 (tensor-side fields, same names - including the reference's spelling
    ``legnth`` - and defaults).  ``model_jit``: the path of a TorchScript file (the reference's usage) or of a state-dict
    file (then ``model_config`` is needed), a scripted module, or an eager ``AdaptiveScaling`` module - which is then
    switched to eval mode and to ``compute_dtype`` IN PLACE (pass a copy to keep a training module as it is), a scripted one
    likewise (its recipe string is rewritten)."""
    model_jit: Union[str, AdaptiveScaling, torch.jit.ScriptModule, None] = None
    device: str = 'cuda'
    backbone_downsampling_factor: int = 32
    rough_head_upsampling_factor: int = 2
    rough_downsample_short_side_legnth: int = 720
    rough_char_mask_positive_thr: float = 0.5
    rough_valid_char_height_min: float = 3.0
    precise_head_upsampling_factor: int = 2
    precise_char_mask_positive_thr: float = 0.5
    model_config: Optional[AdaptiveScalingConfig] = None  # needed to rebuild the module from a state-dict file
    compute_dtype: torch.dtype = torch.float16            # BASELINE.json configs[4]
    # replay one captured HIP graph per (pass, padded shape) instead of enqueuing its few hundred launches from Python; the
    # first call of a shape runs eagerly (inferencing/graphs.py).  Same results bit for bit.
    use_hip_graphs: bool = True


@attrs.define
class AdaptiveScalingInferencingRoughInferResult:
    """:61-66"""
    resized_shape: Tuple[int, int]
    padded_image: np.ndarray
    rough_char_mask: np.ndarray              # (H/FDF, W/FDF) uint8
    rough_char_height_score_map: np.ndarray  # (H/FDF, W/FDF) float32


@attrs.define
class AdaptiveScalingInferencingPresiceInferResult:
    """:69-76 (the reference's spelling)"""
    padded_image: np.ndarray
    precise_char_mask: Optional[np.ndarray]
    precise_char_prob_score_map: np.ndarray
    precise_np_char_up_left_corner_offset: np.ndarray      # (H/FDF, W/FDF, 2)
    precise_np_char_corner_angle_distribution: np.ndarray  # (H/FDF, W/FDF, 4)
    precise_np_char_corner_distance: np.ndarray            # (H/FDF, W/FDF, 4)


def rough_resized_shape(height: int, width: int, short_side: int) -> Tuple[int, int]:
    """:95-107: when the shorter side exceeds ``short_side`` the image is shrunk so that it equals it (the other side
    keeps the aspect ratio, rounded like vkit's ``to_resized_image``)."""
    if min(height, width) <= short_side:
        return height, width
    if height < width:
        return short_side, round(short_side * width / height)
    return round(short_side * height / width), short_side


def _as_mat(image) -> np.ndarray:
    mat = getattr(image, 'mat', image)
    mat = np.asarray(mat)
    if mat.ndim != 3 or mat.shape[2] != 3:
        raise ValueError(f'expected an (H, W, 3) RGB image, got {mat.shape}')
    return mat


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class AdaptiveScalingInferencing:
    """:79-188,295-396 without the vkit geometry."""

    def __init__(self, config: AdaptiveScalingInferencingConfig):
        self.config = config
        self.graphs = GraphCache(enabled=config.use_hip_graphs)
        model = config.model_jit
        if isinstance(model, str):
            try:  # :85-90: a TorchScript file (the operator it calls is registered by importing this package)
                model = torch.jit.load(model, map_location=config.device)
            except RuntimeError:  # not a TorchScript archive: a state-dict / RestoreState file
                if config.model_config is None:
                    raise ValueError('model_config is required to rebuild the module from a state-dict file') from None
                sd = torch.load(model, map_location='cpu', weights_only=True)
                sd = sd.get('model_jit_state_dict', sd) if isinstance(sd, dict) else sd  # RestoreState schema, train.py:91-96
                module = AdaptiveScaling(config.model_config)
                module.load_state_dict(sd)
                model = module
        if isinstance(model, torch.jit.ScriptModule):
            # a scripted module carries its construction recipe as a string attribute (model/scripting.py); the storage type
            # in it is switched to config.compute_dtype, as set_compute_dtype does for an eager module below
            from ..model import scripting
            if not hasattr(model, '_script_spec'):
                raise TypeError('config.model_jit is a TorchScript module that this package did not script')
            model._script_spec = scripting.with_compute_dtype(model._script_spec, config.compute_dtype)
            self.model = model.to(config.device).eval()
            return
        if not isinstance(model, AdaptiveScaling):
            raise TypeError('config.model_jit must be a TorchScript file / module, an AdaptiveScaling module or a state-dict file')
        self.model = model.to(config.device).eval()
        self.model.set_compute_dtype(config.compute_dtype)

    # ---- shared pieces ------------------------------------------------------------------------------------------
    def _to_device(self, mats: Sequence[np.ndarray]) -> torch.Tensor:
        """(H, W, 3) uint8 arrays of one padded size -> (B, 3, H, W) fp32 on the device (:117-122)."""
        x = torch.from_numpy(np.stack([np.ascontiguousarray(m) for m in mats]))
        x = x.to(self.config.device, non_blocking=True)
        return x.permute(0, 3, 1, 2).float()

    @staticmethod
    def _valid(sizes: Sequence[Tuple[int, int]], fdf: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        vh = torch.tensor([math.ceil(h / fdf) for h, _ in sizes], dtype=torch.int32, device=device)
        vw = torch.tensor([math.ceil(w / fdf) for _, w in sizes], dtype=torch.int32, device=device)
        return vh, vw

    # ---- rough pass ----------------------------------------------------------------------------------------------
    def rough_infer(self, image, resize_fn=None) -> AdaptiveScalingInferencingRoughInferResult:
        """:92-188.  ``resize_fn(mat, height, width)`` performs the area-interpolation shrink of the 720 rule (cv2 in the
        reference); without it an image that needs shrinking is rejected - resampling pixels is host-side image I/O."""
        c = self.config
        mat = _as_mat(image)
        h, w = rough_resized_shape(mat.shape[0], mat.shape[1], c.rough_downsample_short_side_legnth)
        if (h, w) != mat.shape[:2]:
            if resize_fn is None:
                raise ValueError(f'image {mat.shape[:2]} exceeds the short-side limit {c.rough_downsample_short_side_legnth}: '
                                 f'pass resize_fn or an image already resized to {(h, w)}')
            mat = np.asarray(resize_fn(mat, h, w))
            assert mat.shape[:2] == (h, w)
        padded = pad_mat_to_make_divisible(mat, c.backbone_downsampling_factor)
        fdf = 4 // c.rough_head_upsampling_factor
        x = self._to_device([padded])
        H, W = padded.shape[0] // fdf, padded.shape[1] // fdf
        vh, vw = self._valid([(h, w)], fdf, x.device)
        thr, hmin = float(c.rough_char_mask_positive_thr), float(c.rough_valid_char_height_min)

        def rough_pass(x, vh, vw):  # the model call + the device post-processing: one HIP graph per padded shape
            mask_feat, height_feat = self.model.forward_rough(x)
            B = mask_feat.shape[0]
            assert tuple(mask_feat.shape) == (B, 1, H, W) and height_feat.shape == mask_feat.shape
            out_mask = torch.empty((B, H, W), dtype=torch.uint8, device=x.device)
            out_height = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
            check(lib.vkas_rough_postprocess(_ptr(mask_feat.contiguous()), _ptr(height_feat.contiguous()), B, H, W, _ptr(vh),
                                             _ptr(vw), thr, hmin, _ptr(out_mask), _ptr(out_height), ops._stream()),
                  'rough_postprocess')
            return out_mask, out_height

        with torch.no_grad():
            out_mask, out_height = self.graphs.run(('rough', thr, hmin), rough_pass, [x, vh, vw], param_stamp(self.model))
        return AdaptiveScalingInferencingRoughInferResult(
            resized_shape=(math.ceil(h / fdf), math.ceil(w / fdf)), padded_image=padded,
            rough_char_mask=out_mask[0].cpu().numpy(), rough_char_height_score_map=out_height[0].cpu().numpy())

    # ---- precise pass --------------------------------------------------------------------------------------------
    def precise_infer_batch(self, images: Sequence) -> Sequence[AdaptiveScalingInferencingPresiceInferResult]:
        """:295-396 for a batch of stacked-region pages: each is padded to x32, pages of one padded size share a model
        call (the reference feeds one page at a time)."""
        c = self.config
        mats = [_as_mat(im) for im in images]
        padded = [pad_mat_to_make_divisible(m, c.backbone_downsampling_factor) for m in mats]
        fdf = 4 // c.precise_head_upsampling_factor
        results = [None] * len(mats)
        groups = {}
        for i, p in enumerate(padded):
            groups.setdefault(p.shape[:2], []).append(i)
        for shape, idxs in groups.items():
            x = self._to_device([padded[i] for i in idxs])
            H, W = shape[0] // fdf, shape[1] // fdf
            vh, vw = self._valid([mats[i].shape[:2] for i in idxs], fdf, x.device)

            def precise_pass(x, vh, vw):
                prob, offset, angle, dist = self.model.forward_precise(x)
                B = prob.shape[0]
                assert tuple(prob.shape) == (B, 1, H, W)
                o_prob = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
                o_off = torch.empty((B, H, W, 2), dtype=torch.float32, device=x.device)
                o_ang = torch.empty((B, H, W, 4), dtype=torch.float32, device=x.device)
                o_dist = torch.empty((B, H, W, 4), dtype=torch.float32, device=x.device)
                check(lib.vkas_precise_postprocess(_ptr(prob.contiguous()), _ptr(offset.contiguous()), _ptr(angle.contiguous()),
                                                   _ptr(dist.contiguous()), B, H, W, _ptr(vh), _ptr(vw), _ptr(o_prob),
                                                   _ptr(o_off), _ptr(o_ang), _ptr(o_dist), ops._stream()), 'precise_postprocess')
                return o_prob, o_off, o_ang, o_dist

            with torch.no_grad():
                o_prob, o_off, o_ang, o_dist = self.graphs.run('precise', precise_pass, [x, vh, vw], param_stamp(self.model))
            o_prob, o_off, o_ang, o_dist = (t.cpu().numpy() for t in (o_prob, o_off, o_ang, o_dist))
            for k, i in enumerate(idxs):
                results[i] = AdaptiveScalingInferencingPresiceInferResult(
                    padded_image=padded[i], precise_char_mask=None, precise_char_prob_score_map=o_prob[k],
                    precise_np_char_up_left_corner_offset=o_off[k], precise_np_char_corner_angle_distribution=o_ang[k],
                    precise_np_char_corner_distance=o_dist[k])
        return results

    def precise_infer(self, image) -> AdaptiveScalingInferencingPresiceInferResult:
        return self.precise_infer_batch([image])[0]
