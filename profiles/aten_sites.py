#!/usr/bin/env python3
"""Which lines of this package (or which autograd nodes) still run torch-native device ops inside one train step: a
TorchDispatchMode that records every aten op touching a device tensor with the innermost Python frame inside the package
(ops the autograd engine runs by itself - gradient accumulation, slice / view backward - have no package frame and are
listed under the engine).  Complements aten_trace.py (torch.profiler resolves no Python stacks on this image)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
import bench  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,  # noqa: E402
                                                       AdaptiveScalingNeckHeadType)
from vkit_ocr_model_adaptive_scaling_amd.loss_function import (AdaptiveScalingRoughLossFunction,  # noqa: E402
                                                               AdaptiveScalingRoughLossFunctionConifg,
                                                               AdaptiveScalingPreciseLossFunction,
                                                               AdaptiveScalingPreciseLossFunctionConifg)
from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, FlatAdamW, TwoPassStep  # noqa: E402

PKG = 'vkit_ocr_model_adaptive_scaling_amd'
# ops that launch nothing: views, metadata, allocation without fill
_FREE = ('view', 'as_strided', 'slice', 'select', 'narrow', 'permute', 'transpose', 't.', 'expand', 'reshape', 'detach',
         'alias', 'empty', 'unsqueeze', 'squeeze', 'split', 'unbind', 'size', 'stride', 'is_', 'numel', 'dim', '_unsafe_view',
         'unflatten', 'flatten', 'chunk', 'set_', 'resize_', 'record_stream', 'lift_fresh', '_local_scalar_dense', 'item',
         'new_empty', 'result_type', 'can_cast', '_has_compatible_shallow_copy_type', 'storage_offset', 'sym_', 'prim::')


class Sites(TorchDispatchMode):

    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        short = name.replace('aten.', '')
        if any(short.startswith(f) for f in _FREE):
            return out
        flat = [a for a in list(args) + list((kwargs or {}).values()) + ([out] if isinstance(out, torch.Tensor) else [])
                if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in flat):
            return out
        frame = '(autograd engine)'
        for f in reversed(traceback.extract_stack(limit=40)):
            if PKG in f.filename or f.filename.endswith('bench.py'):
                frame = '%s:%d %s' % (f.filename.split(PKG)[-1], f.lineno, f.name)
                break
        self.counts[(short, frame)] += 1
        return out


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT)).to(dev).train()
    flat = FlatBuffers(model.named_parameters())
    opt = FlatAdamW(None, flat=flat)
    step = TwoPassStep(model, AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg()),
                       AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg()), opt, None,
                       merge_backbone=True)
    rough, precise = bench.synthetic_batches(2, (size, size), dev, 1)
    for _ in range(2):
        step(rough, precise, lr=1e-4)
    torch.cuda.synchronize()
    n = 2
    with Sites() as s:
        for _ in range(n):
            step(rough, precise, lr=1e-4)
        torch.cuda.synchronize()
    total = 0
    for (name, frame), c in sorted(s.counts.items(), key=lambda kv: -kv[1]):
        print('%6.1f / step  %-34s %s' % (c / n, name, frame))
        total += c
    print('%6.1f / step  total' % (total / n))


if __name__ == '__main__':
    main()
