"""Data-parallel gradient reduction for the two-pass adaptive-scaling step.

The reference has no distributed code; torch's DistributedDataParallel cannot wrap this model (no ``forward()``, two
passes per step whose parameter sets differ, backbone gradients only final after the second backward).  This reducer
all-reduces contiguous ranges ("buckets") of the flat gradient buffer as soon as they are final:

  * after backward #1 (rough): the rough neck + rough heads range — it overlaps the entire precise pass;
  * during backward #2 (precise): the precise heads/neck range, then backbone stage 3, 2, 1, 0 + stem, each launched
    from a post-accumulate-grad hook when the last parameter of the bucket has received its gradient.

Backbone buckets are armed only for the second backward (their gradients are partial after the first).  Collectives
are ``torch.distributed.all_reduce(..., async_op=True)`` on slices of the flat buffer: with the ``nccl`` backend that
is RCCL over xGMI on its own stream, ordered after the compute stream at launch and joined by ``wait()`` before the
clip + AdamW kernels.  Gradients are averaged by pre-scaling the losses with 1/world_size.  Works with ``gloo`` on CPU
tensors as well (tests)."""
import os
import warnings
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .flat import FlatBuffers


class Bucket:

    def __init__(self, name: str, start: int, end: int, param_names: Sequence[str]):
        self.name, self.start, self.end = name, start, end
        self.param_names = list(param_names)
        self.waiting = set()   # parameters of this bucket that have not delivered their gradient in the armed backward
        self.armed = False


class BucketedGradReducer:

    def __init__(self, flat: FlatBuffers, bucket_prefixes: Sequence[Tuple[str, Tuple[str, ...]]],
                 process_group=None, always_reduce: Optional[bool] = None):
        """always_reduce (default: environment VKAS_FORCE_REDUCER=1): issue the collectives at world size 1 as well (a sum
        over one rank leaves the gradient as it is).  A test / measurement switch: it runs the whole N > 1 code path - RCCL
        library load, communicator, the side stream's ordering against the compute stream, all_reduce launched from a callback
        inside autograd's backward thread, wait() before the clip kernel - on the one GPU a build box has."""
        self.flat = flat
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if always_reduce is None:
            always_reduce = os.environ.get('VKAS_FORCE_REDUCER', '0') == '1'
        self.collective = self.world_size > 1 or (bool(always_reduce) and dist.is_initialized())
        self.collectives_issued = 0
        self.buckets: Dict[str, Bucket] = {}
        self._param_bucket: Dict[str, Bucket] = {}
        covered = set()
        for name, prefixes in bucket_prefixes:
            start, end = flat.range_of(tuple(prefixes))
            names = [n for n in flat.names if n.startswith(tuple(prefixes))]
            b = Bucket(name, start, end, names)
            self.buckets[name] = b
            for n in names:
                if n in covered:
                    raise ValueError(f'{n} belongs to two buckets')
                covered.add(n)
                self._param_bucket[n] = b
        missing = [n for n in flat.names if n not in covered]
        if missing:
            raise ValueError(f'parameters without a bucket: {missing[:4]}...')
        self._works: List = []
        self.launch_log: List[str] = []  # bucket names in launch order (inspected by tests)
        # like DistributedDataParallel at construction: every rank starts from rank 0's parameters, whatever its seed
        if self.world_size > 1:
            flat.broadcast_params(src=0, group=process_group)
        flat.subscribe(self._on_param_grad)  # fires for autograd accumulation and for the ops' direct delivery alike

    def _on_param_grad(self, name: str):
        b = self._param_bucket[name]
        if not b.armed:
            return
        # a delivery is reported per USE of a parameter by the ops' direct path (and once per backward by autograd's
        # accumulation hook): count each parameter once, so that a parameter used twice in one graph cannot fire its
        # bucket while another parameter's gradient is still missing.  (Its own later contributions would be missed by
        # an early launch all the same: a model that re-uses a parameter must put it in a bucket that is flushed.)
        b.waiting.discard(name)
        if not b.waiting:
            b.armed = False
            self._launch(b)

    def arm(self, bucket_names: Sequence[str], expected: Optional[Dict[str, Sequence[str]]] = None):
        """Arm buckets for the next backward: each launches when all its parameters have accumulated a gradient.
        ``expected`` names, per bucket, the parameters that will receive a gradient in this pass when that is a subset."""
        for bn in bucket_names:
            b = self.buckets[bn]
            b.waiting = set(expected[bn]) if expected and bn in expected else set(b.param_names)
            b.armed = len(b.waiting) > 0

    def _launch(self, b: Bucket):
        self.launch_log.append(b.name)
        if not self.collective:
            return
        self.collectives_issued += 1
        work = dist.all_reduce(self.flat.flat_grad[b.start:b.end], op=dist.ReduceOp.SUM, group=self.group,
                               async_op=True)
        self._works.append(work)

    def flush(self, bucket_names: Sequence[str] = ()):
        """Launch any still-armed bucket (a parameter that got no gradient keeps its bucket from firing) and
        explicitly named ones, then wait for every outstanding collective."""
        for b in self.buckets.values():
            if b.armed or b.name in bucket_names:
                b.armed = False
                self._launch(b)
        for w in self._works:
            w.wait()
        self._works.clear()


def adaptive_scaling_buckets(model) -> List[Tuple[str, Tuple[str, ...]]]:
    """Bucket layout for AdaptiveScaling: finalisation order rough -> precise -> backbone stages (reverse)."""
    n_blocks = len(model.backbone.blocks)
    buckets = [('rough', ('rough_neck.', 'rough_char_mask_head.', 'rough_char_height_head.')),
               ('precise', ('precise_neck.', 'precise_char_mask_head.', 'precise_char_prob_head.',
                            'precise_char_up_left_corner_offset_head.', 'precise_char_corner_angle_head.',
                            'precise_char_corner_distance_head.'))]
    for i in range(n_blocks - 1, 0, -1):
        buckets.append((f'backbone{i}', (f'backbone.blocks.{i}.',)))
    buckets.append(('backbone0', ('backbone.stem.', 'backbone.blocks.0.')))
    return buckets


class TwoPassStep:
    """One training step with the reference's semantics (train.py:397-478): rough forward/loss/backward, precise
    forward/loss/backward (gradients accumulate), gradient all-reduce, global-norm clip, AdamW."""

    def __init__(self, model, rough_loss_fn, precise_loss_fn, optimizer, reducer: Optional[BucketedGradReducer] = None,
                 merge_backbone: bool = False, label_point_forward: bool = False):
        """merge_backbone: run the backbone once over both batches (model.forward_both) and back-propagate
        rough_loss + precise_loss in one backward: the same gradients as the two accumulating passes (the sum is taken
        in a different order), half the backbone launches."""
        self.model, self.rough_loss_fn, self.precise_loss_fn = model, rough_loss_fn, precise_loss_fn
        self.optimizer, self.reducer = optimizer, reducer
        self.world = reducer.world_size if reducer is not None else 1
        self._backbone_buckets = [b for b in (reducer.buckets if reducer else {}) if b.startswith('backbone')]
        self.merge_backbone = merge_backbone
        # opt-in, NOT the reference's module API: the regression heads' forward at the label points only (the precise loss
        # reads nothing else of them); same losses and gradients, see ops.HeadsAtPoints
        self.label_point_forward = label_point_forward
        self._one: Optional[torch.Tensor] = None
        if getattr(model, 'compute_dtype', None) == torch.float16:
            # INTEGRATION.md "fp16 training": block_scale = 1e-6 (convnext.py:38) times an fp16 activation gradient underflows,
            # the residual branches then receive no gradient at all - measured, with and without a loss scale
            warnings.warn('TwoPassStep on a float16 model: gradients of the ConvNeXt residual branches underflow in fp16 '
                          'storage from the reference initialisation (block_scale = 1e-6); train in bfloat16 (the default) '
                          'and use float16 for inference only', RuntimeWarning, stacklevel=2)

    def _rough_loss(self, outs, b, scale):
        mask, height = outs
        return self.rough_loss_fn(mask, height, b['downsampled_mask'], b['downsampled_score_map'], b['downsampled_shape'],
                                  b['downsampled_core_box'], scale=scale)

    def _precise_loss(self, outs, b, scale):
        prob, offset, angle, dist_ = outs
        return self.precise_loss_fn(None, prob, offset, angle, dist_, b['downsampled_score_map'], b['downsampled_mask'],
                                    b['downsampled_shape'], b['downsampled_core_box'], b['downsampled_label_point_y'],
                                    b['downsampled_label_point_x'], b['up_left_offsets'], b['corner_angles'],
                                    b['corner_distances'], scale=scale)

    def __call__(self, rough_batch: dict, precise_batch: dict, lr: Optional[float] = None):
        scale = 0.5 / self.world  # train.py:413,451 (loss / 2), averaged over ranks
        r = self.reducer
        pts = ((precise_batch['downsampled_label_point_y'], precise_batch['downsampled_label_point_x'])
               if self.label_point_forward else None)
        if self.merge_backbone:
            rough_out, precise_out = self.model.forward_both(rough_batch['image'], precise_batch['image'],
                                                             precise_label_points=pts)
            rough_loss = self._rough_loss(rough_out, rough_batch, scale)
            precise_loss = self._precise_loss(precise_out, precise_batch, scale)
            if r is not None:
                r.arm(['rough', 'precise'] + self._backbone_buckets)
            # d(rough_loss + precise_loss) without building the sum: both roots seeded with a cached 1
            one = self._one
            if one is None or one.device != rough_loss.device or one.dtype != rough_loss.dtype or one.shape != rough_loss.shape:
                one = self._one = torch.ones_like(rough_loss)
            torch.autograd.backward((rough_loss, precise_loss), (one, one))
        else:
            rough_loss = self._rough_loss(self.model.forward_rough(rough_batch['image']), rough_batch, scale)
            if r is not None:
                r.arm(['rough'])
            rough_loss.backward()
            precise_loss = self._precise_loss(self.model.forward_precise(precise_batch['image'], label_points=pts),
                                              precise_batch, scale)
            if r is not None:
                r.arm(['precise'] + self._backbone_buckets)
            precise_loss.backward()
        if r is not None:
            r.flush()
        self.optimizer.step(lr=lr)
        self.optimizer.zero_grad()
        # the values the reference logs are loss / 2 (train.py:413-415,451-453): undo the 1 / world pre-scaling
        if self.world > 1:
            return rough_loss.detach() * self.world, precise_loss.detach() * self.world
        return rough_loss.detach(), precise_loss.detach()
