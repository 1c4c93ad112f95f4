"""TorchScript bridge of the model mirror.

The reference's only caller scripts the model before training (experiment/adaptive_scaling/train.py:277-280:
``model_jit = torch.jit.script(model); model_jit = model_jit.to(device); del model``), saves / restores
``model_jit.state_dict()`` (train.py:599,314) and the inference class loads a TorchScript file
(vkit_open_model/inferencing/adaptive_scaling.py:85-90); the reference's own tests script - and call - the backbone, the
necks and the heads on their own (tests/test_convnext.py:53-63, tests/test_fpn.py:30,49, tests/test_upernext.py:30).  The HIP
ops of this package are Python ``torch.autograd.Function``s over the C ABI, which TorchScript cannot compile, so every
scriptable ``forward`` of the mirror compiles to ONE call of a dispatcher-registered operator

    vkas::adaptive_scaling_forward(Tensor x, Tensor[] params, str spec, int which, bool training) -> Tensor[]
    vkas::module_forward(Tensor[] inputs, Tensor[] params, str spec, bool training) -> Tensor[]

(the first for ``AdaptiveScaling.forward_rough`` / ``forward_precise``, the second for ``ConvNext``, ``ConvNextBlock``,
``ConvNextBlockLayer``, ``PpmBlock``, ``UperNextNeck``, ``UperNextHead``, ``FpnNeck``, ``FpnHead``) whose kernel - registered for
CompositeImplicitAutograd, i.e. autograd records what it runs - is the eager forward of this package on those very
parameter tensors:

* ``spec`` is the construction recipe (class, constructor arguments, storage type) the scripted module carries as a string
  attribute; the kernel keeps ONE parameter-less skeleton of the eager module per recipe (built on the meta device: no
  memory), binds the caller's tensors into it for the duration of the call and unbinds them afterwards.  So a scripted
  module never depends on, and never touches, the eager module it was scripted from (its train / eval flag and storage type
  included), works after ``del model`` and after ``torch.jit.load`` in another process, and nothing here keeps parameter
  memory alive once the scripted module is gone;
* outputs and gradients are those of the eager module bit for bit (it is the same code on the same tensors);
* a saved file needs this package imported before ``torch.jit.load`` (that registers the operators, exactly as a C++
  extension would need ``torch.ops.load_library``); nothing in it falls back to stock torch math.
"""
import json
from typing import Callable, Dict, List

import torch

_LIB = torch.library.Library('vkas', 'DEF')
_LIB.define('adaptive_scaling_forward(Tensor x, Tensor[] params, str spec, int which, bool training) -> Tensor[]')
_LIB.define('module_forward(Tensor[] inputs, Tensor[] params, str spec, bool training) -> Tensor[]')

_DTYPE_NAMES = {torch.bfloat16: 'bf16', torch.float16: 'f16', torch.float32: 'f32'}
_DTYPES = {v: k for k, v in _DTYPE_NAMES.items()}

# recipe -> (skeleton module on the meta device, [(owner module, parameter name, meta placeholder)] in named_parameters()
# order).  A handful of entries (one per scripted recipe); `clear()` drops them.
_SKELETONS: Dict[str, tuple] = {}
_BUILDERS: Dict[str, Callable] = {}


def make_spec(config, compute_dtype: torch.dtype) -> str:
    """Recipe of an AdaptiveScaling model (kept free of a 'cls' key: files saved by earlier versions carry this form)."""
    return json.dumps({'size': config.size.value, 'neck_head_type': config.neck_head_type.value,
                       'rough_upsampling_factor': config.rough_upsampling_factor,
                       'rough_init_char_height_output_bias': config.rough_init_char_height_output_bias,
                       'precise_upsampling_factor': config.precise_upsampling_factor,
                       'precise_enable_char_mask_head': config.precise_enable_char_mask_head,
                       'compute_dtype': _DTYPE_NAMES[compute_dtype]}, sort_keys=True)


def make_module_spec(cls_name: str, args: dict, compute_dtype: torch.dtype) -> str:
    return json.dumps({'cls': cls_name, 'args': args, 'compute_dtype': _DTYPE_NAMES[compute_dtype]}, sort_keys=True)


def init_script_state(module: torch.nn.Module, args: dict) -> None:
    """Called at the end of a scriptable module's __init__: what its compiled ``forward`` hands to vkas::module_forward."""
    module._script_args_json = json.dumps(args, sort_keys=True)
    module._script_params = [p for _, p in module.named_parameters()]
    module._script_spec = make_module_spec(type(module).__name__, args, module.compute_dtype)


def refresh_module_spec(module: torch.nn.Module) -> None:
    module._script_spec = make_module_spec(type(module).__name__, json.loads(module._script_args_json), module.compute_dtype)


def with_compute_dtype(spec: str, compute_dtype: torch.dtype) -> str:
    """The recipe ``spec`` with another storage type (what set_compute_dtype is to an eager module)."""
    d = json.loads(spec)
    d['compute_dtype'] = _DTYPE_NAMES[compute_dtype]
    return json.dumps(d, sort_keys=True)


def clear() -> None:
    """Drop the cached skeletons (they hold no parameter memory; rebuilt on the next scripted call)."""
    _SKELETONS.clear()


def _builders():
    if not _BUILDERS:
        from .convnext import ConvNext, ConvNextBlock, ConvNextBlockLayer
        from .upernext import PpmBlock, UperNextNeck, UperNextHead
        from .fpn import FpnNeck, FpnHead
        for cls in (ConvNext, ConvNextBlock, ConvNextBlockLayer, PpmBlock, UperNextNeck, UperNextHead, FpnNeck, FpnHead):
            _BUILDERS[cls.__name__] = cls
    return _BUILDERS


def _build(d: dict):
    if 'cls' not in d:
        from .adaptive_scaling import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                       AdaptiveScalingNeckHeadType)
        config = AdaptiveScalingConfig(size=AdaptiveScalingSize(d['size']),
                                       neck_head_type=AdaptiveScalingNeckHeadType(d['neck_head_type']),
                                       rough_upsampling_factor=d['rough_upsampling_factor'],
                                       rough_init_char_height_output_bias=d['rough_init_char_height_output_bias'],
                                       precise_upsampling_factor=d['precise_upsampling_factor'],
                                       precise_enable_char_mask_head=d['precise_enable_char_mask_head'])
        return AdaptiveScaling(config, compute_dtype=_DTYPES[d['compute_dtype']])
    cls = _builders().get(d['cls'])
    if cls is None:
        raise RuntimeError(f"vkas::module_forward: unknown module class {d['cls']!r}")
    args = dict(d['args'])
    for k, v in args.items():  # JSON turned the tuples into lists
        if isinstance(v, list):
            args[k] = tuple(tuple(e) if isinstance(e, list) else e for e in v)
    from .helper import set_compute_dtype
    return set_compute_dtype(cls(**args), _DTYPES[d['compute_dtype']])


def _skeleton(spec: str):
    ent = _SKELETONS.get(spec)
    if ent is None:
        with torch.device('meta'):  # structure only: every parameter is replaced by the caller's tensor per call
            model = _build(json.loads(spec))
        slots = []
        for name, p in model.named_parameters():
            owner = model
            *path, leaf = name.split('.')
            for part in path:
                owner = getattr(owner, part)
            slots.append((owner, leaf, p, name))
        ent = (model, slots)
        _SKELETONS[spec] = ent
    return ent


class _Bound:
    """Context: the caller's parameter tensors bound into the recipe's skeleton (no copy, no new autograd leaf)."""

    def __init__(self, who: str, params, spec: str, training: bool):
        self.model, self.slots = _skeleton(spec)
        if len(self.slots) != len(params):
            raise RuntimeError(f'{who}: {len(params)} parameter tensors for a module with {len(self.slots)}')
        for (owner, leaf, meta, name), t in zip(self.slots, params):
            if tuple(meta.shape) != tuple(t.shape):
                raise RuntimeError(f'{who}: parameter {name} has shape {tuple(t.shape)}, expected {tuple(meta.shape)}')
        self.params = params
        self.training = training

    def __enter__(self):
        m = self.model
        if getattr(m, '_script_busy', False):
            raise RuntimeError('vkas scripted forward re-entered for the same recipe (not re-entrant)')
        m._script_busy = True
        for (owner, leaf, _, _), t in zip(self.slots, self.params):
            owner._parameters[leaf] = t
        if hasattr(m, '_script_params'):
            m._script_params = list(self.params)
        if m.training != self.training:
            m.train(self.training)
        return m

    def __exit__(self, *exc):
        for owner, leaf, meta, _ in self.slots:
            owner._parameters[leaf] = meta
        if hasattr(self.model, '_script_params'):
            self.model._script_params = []
        self.model._script_busy = False
        return False


def _adaptive_scaling_forward(x: torch.Tensor, params: List[torch.Tensor], spec: str, which: int, training: bool):
    if which not in (0, 1):
        raise RuntimeError(f'vkas::adaptive_scaling_forward: unknown branch {which}')
    with _Bound('vkas::adaptive_scaling_forward', params, spec, training) as model:
        return list(model._forward_rough_eager(x) if which == 0 else model._forward_precise_eager(x))


def _module_forward(inputs: List[torch.Tensor], params: List[torch.Tensor], spec: str, training: bool):
    with _Bound('vkas::module_forward', params, spec, training) as model:
        return list(model._script_call(list(inputs)))


# CompositeImplicitAutograd: the kernels are ordinary autograd-recording code (the package's autograd.Functions and a few
# views), so a scripted module trains exactly like the eager one - no separate backward formula to keep in step
_LIB.impl('adaptive_scaling_forward', _adaptive_scaling_forward, 'CompositeImplicitAutograd')
_LIB.impl('module_forward', _module_forward, 'CompositeImplicitAutograd')
