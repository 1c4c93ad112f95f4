"""TorchScript bridge of the AdaptiveScaling mirror.

The reference's only caller scripts the model before training (experiment/adaptive_scaling/train.py:277-280:
``model_jit = torch.jit.script(model); model_jit = model_jit.to(device); del model``), saves / restores
``model_jit.state_dict()`` (train.py:599,314) and the inference class loads a TorchScript file
(vkit_open_model/inferencing/adaptive_scaling.py:85-90).  The HIP ops of this package are Python
``torch.autograd.Function``s over the C ABI, which TorchScript cannot compile, so the two exported methods
``forward_rough`` / ``forward_precise`` compile to ONE call of a dispatcher-registered operator

    vkas::adaptive_scaling_forward(Tensor x, Tensor[] params, str spec, int which, bool training) -> Tensor[]

whose kernel (registered for CompositeImplicitAutograd, i.e. autograd records what it runs) is the eager forward of
this package on those very parameter tensors.  Consequences:

* ``torch.jit.script(model)`` works, the scripted module owns the parameters under the reference's state-dict keys and
  shares them with the eager module it was scripted from; outputs and gradients are those of the eager module bit for bit;
* after ``del model`` (the reference's order) or ``torch.jit.load`` in another process the kernel rebuilds an eager module
  from ``spec`` (size / neck type / factors / storage type) and binds the scripted module's tensors into it - no copy;
* a saved file needs this package imported before ``torch.jit.load`` (that registers the operator, exactly as a C++
  extension would need ``torch.ops.load_library``); nothing in it falls back to stock torch math.
"""
import json
import weakref
from typing import Dict, List, Tuple

import torch

_SCHEMA = 'adaptive_scaling_forward(Tensor x, Tensor[] params, str spec, int which, bool training) -> Tensor[]'
_LIB = torch.library.Library('vkas', 'DEF')
_LIB.define(_SCHEMA)

_DTYPE_NAMES = {torch.bfloat16: 'bf16', torch.float16: 'f16', torch.float32: 'f32'}
_DTYPES = {v: k for k, v in _DTYPE_NAMES.items()}

# eager modules that serve scripted ones: weak references to live modules (scripted from an eager module that still
# exists) and the modules rebuilt from a spec (owned here; a handful at most - one per scripted / loaded model - and each
# keeps the parameter tensors it was built around alive until it is evicted)
_LIVE: List['weakref.ReferenceType'] = []
_REBUILT: Dict[Tuple[str, int], object] = {}
_MAX_REBUILT = 4


def make_spec(config, compute_dtype: torch.dtype) -> str:
    return json.dumps({'size': config.size.value, 'neck_head_type': config.neck_head_type.value,
                       'rough_upsampling_factor': config.rough_upsampling_factor,
                       'rough_init_char_height_output_bias': config.rough_init_char_height_output_bias,
                       'precise_upsampling_factor': config.precise_upsampling_factor,
                       'precise_enable_char_mask_head': config.precise_enable_char_mask_head,
                       'compute_dtype': _DTYPE_NAMES[compute_dtype]}, sort_keys=True)


def register_live(model) -> None:
    _LIVE[:] = [r for r in _LIVE if r() is not None]
    _LIVE.append(weakref.ref(model))


def _same_tensors(model, params) -> bool:
    own = model._script_params
    return (len(own) == len(params) and (not own or own[0] is params[0]) and own[-1] is params[-1]
            and all(a is b for a, b in zip(own, params)))


def _rebuild(spec: str, params):
    from .adaptive_scaling import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                   AdaptiveScalingNeckHeadType)
    d = json.loads(spec)
    config = AdaptiveScalingConfig(size=AdaptiveScalingSize(d['size']),
                                   neck_head_type=AdaptiveScalingNeckHeadType(d['neck_head_type']),
                                   rough_upsampling_factor=d['rough_upsampling_factor'],
                                   rough_init_char_height_output_bias=d['rough_init_char_height_output_bias'],
                                   precise_upsampling_factor=d['precise_upsampling_factor'],
                                   precise_enable_char_mask_head=d['precise_enable_char_mask_head'])
    with torch.device('meta'):  # structure only: every parameter is replaced by the caller's tensor below
        model = AdaptiveScaling(config, compute_dtype=_DTYPES[d['compute_dtype']])
    names = [n for n, _ in model.named_parameters()]
    if len(names) != len(params):
        raise RuntimeError(f'vkas::adaptive_scaling_forward: {len(params)} parameter tensors for a model with {len(names)}')
    for name, t in zip(names, params):
        owner = model
        *path, leaf = name.split('.')
        for part in path:
            owner = getattr(owner, part)
        if tuple(owner._parameters[leaf].shape) != tuple(t.shape):
            raise RuntimeError(f'vkas::adaptive_scaling_forward: parameter {name} has shape {tuple(t.shape)}, expected '
                               f'{tuple(owner._parameters[leaf].shape)}')
        owner._parameters[leaf] = t  # the caller's tensor itself (no copy, no new autograd leaf)
    model._script_params = list(params)
    return model


def _model_for(params, spec: str):
    for r in _LIVE:
        m = r()
        if m is not None and _same_tensors(m, params):
            return m
    key = (spec, params[0].data_ptr() if len(params) else 0)
    m = _REBUILT.get(key)
    if m is not None and _same_tensors(m, params):
        return m
    m = _rebuild(spec, params)
    if len(_REBUILT) >= _MAX_REBUILT:
        _REBUILT.pop(next(iter(_REBUILT)))
    _REBUILT[key] = m
    return m


def _adaptive_scaling_forward(x: torch.Tensor, params: List[torch.Tensor], spec: str, which: int, training: bool):
    model = _model_for(params, spec)
    if model.training != training:
        model.train(training)
    if which == 0:
        return list(model._forward_rough_eager(x))
    if which == 1:
        return list(model._forward_precise_eager(x))
    raise RuntimeError(f'vkas::adaptive_scaling_forward: unknown branch {which}')


# CompositeImplicitAutograd: the kernel is ordinary autograd-recording code (the package's autograd.Functions and a few
# views), so the scripted module trains exactly like the eager one - no separate backward formula to keep in step
_LIB.impl('adaptive_scaling_forward', _adaptive_scaling_forward, 'CompositeImplicitAutograd')
