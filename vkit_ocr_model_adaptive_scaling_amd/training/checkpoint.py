"""Checkpoint files in the reference's ``RestoreState`` schema (experiment/adaptive_scaling/train.py:91-96,304-330,
599-605,608-644): one ``torch.save``d dict with

    epoch_idx, model_jit_state_dict, optimizer_state_dict, optimizer_scheduler_state_dict

so that a run can be resumed in either code base.  The module's state-dict keys / shapes are identical to the reference's
(tests/test_cpu_host.py::test_state_dict_schema_matches_reference), so ``model_jit_state_dict`` is
``model.state_dict()`` verbatim; the fused optimizer's flat moment buffers are exported as a ``torch.optim.AdamW``
state dict (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` in ``model.parameters()`` order) and the closed-form
learning-rate rule as a ``CosineAnnealingWarmRestarts`` state dict.  Files are read with ``weights_only=True`` (nothing
in them is executed)."""
import math
from typing import Any, Dict, Mapping, Optional

import attrs
import torch

from .flat import FlatBuffers


@attrs.define
class RestoreState:
    """train.py:91-96"""
    epoch_idx: int
    model_jit_state_dict: Mapping[str, torch.Tensor]
    optimizer_state_dict: Mapping[str, Any]
    optimizer_scheduler_state_dict: Mapping[str, Any]


def optimizer_state_dict(optimizer) -> Dict[str, Any]:
    """FlatAdamW -> the dict ``torch.optim.AdamW(model.parameters()).state_dict()`` would hold."""
    flat: FlatBuffers = optimizer.flat
    state = {}
    if optimizer.step_count > 0:
        # one copy of each moment buffer to the host, sliced there.  torch.optim.AdamW holds no state for a parameter that never
        # received a gradient (the fused step skips those too: a head that a pass never runs, ...) and full state - step N, zero
        # moments - for one whose gradients were exactly zero: FlatAdamW.has_state records which parameters a step has updated
        m1, m2 = optimizer.exp_avg.detach().cpu(), optimizer.exp_avg_sq.detach().cpu()
        has_state = getattr(optimizer, 'has_state', None)  # an optimizer object without the record: every parameter has state
        for i, n in enumerate(flat.names):
            if has_state is not None and not has_state[i]:
                continue
            start, size = flat.offsets[n]
            shape = flat.params[i].shape
            state[i] = {'step': torch.tensor(float(optimizer.step_count)),
                        'exp_avg': m1[start:start + size].view(shape).clone(),
                        'exp_avg_sq': m2[start:start + size].view(shape).clone()}
    group = {'lr': optimizer.lr, 'betas': tuple(optimizer.betas), 'eps': optimizer.eps, 'weight_decay': optimizer.weight_decay,
             'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
             'initial_lr': optimizer.lr, 'params': list(range(len(flat.names)))}
    return {'state': state, 'param_groups': [group]}


def load_optimizer_state_dict(optimizer, sd: Mapping[str, Any]):
    """``torch.optim.AdamW`` state dict (ours or the reference's) -> the flat moment buffers and the step counter."""
    flat: FlatBuffers = optimizer.flat
    group = sd['param_groups'][0]
    if len(group['params']) != len(flat.names):
        raise ValueError(f"optimizer state holds {len(group['params'])} parameters, the model has {len(flat.names)}")
    optimizer.lr = float(group.get('lr', optimizer.lr))
    optimizer.betas = tuple(group.get('betas', optimizer.betas))
    optimizer.eps = float(group.get('eps', optimizer.eps))
    optimizer.weight_decay = float(group.get('weight_decay', optimizer.weight_decay))
    optimizer.exp_avg.zero_()
    optimizer.exp_avg_sq.zero_()
    step = 0
    optimizer.has_state = bytearray(len(flat.names))
    optimizer.restored_from_state = True
    with torch.no_grad():
        for i, pid in enumerate(group['params']):
            st = sd['state'].get(pid)
            if st is None:
                continue
            optimizer.has_state[i] = 1
            start, size = flat.offsets[flat.names[i]]
            if st['exp_avg'].numel() != size:
                raise ValueError(f'moment shape mismatch for {flat.names[i]}')
            optimizer.exp_avg[start:start + size].copy_(st['exp_avg'].reshape(-1))
            optimizer.exp_avg_sq[start:start + size].copy_(st['exp_avg_sq'].reshape(-1))
            step = max(step, int(float(st['step'])))
    optimizer.step_count = step  # the fused kernel keeps one bias-correction step for all parameters (train.py: all equal)


def scheduler_state_dict(epoch: float, base_lr: float, eta_min: float, t_0: int, t_mult: int, step_count: int = 0) -> Dict[str, Any]:
    """``CosineAnnealingWarmRestarts.state_dict()`` after ``scheduler.step(epoch)`` (train.py:290-298,475-477)."""
    from .optimizer import cosine_warm_restarts_lr
    if epoch >= t_0 and t_mult != 1:
        n = int(math.log(epoch / t_0 * (t_mult - 1) + 1, t_mult))
        t_cur, t_i = epoch - t_0 * (t_mult ** n - 1) / (t_mult - 1), t_0 * t_mult ** n
    elif epoch >= t_0:
        t_cur, t_i = epoch % t_0, t_0
    else:
        t_cur, t_i = epoch, t_0
    lr = cosine_warm_restarts_lr(epoch, base_lr, eta_min, t_0, t_mult)
    return {'T_0': t_0, 'T_i': t_i, 'T_mult': t_mult, 'eta_min': eta_min, 'T_cur': t_cur, 'base_lrs': [base_lr],
            'last_epoch': math.floor(epoch), '_step_count': step_count, '_get_lr_called_within_step': False,
            '_last_lr': [lr]}


def save_restore_state(path, epoch_idx: int, model: torch.nn.Module, optimizer, scheduler_state: Mapping[str, Any]):
    """train.py:599-605: ``torch.save(cattrs.unstructure(RestoreState(...)), path)``."""
    rs = RestoreState(epoch_idx=int(epoch_idx),
                      model_jit_state_dict={k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                      optimizer_state_dict=optimizer_state_dict(optimizer), optimizer_scheduler_state_dict=dict(scheduler_state))
    torch.save(attrs.asdict(rs, recurse=False), path)
    return rs


def load_restore_state(path, model: Optional[torch.nn.Module] = None, optimizer=None, flat: Optional[FlatBuffers] = None) -> RestoreState:
    """train.py:304-330: reads a RestoreState file and (optionally) restores the module's parameters and the optimizer.
    Parameters that live in flat buffers are written through ``load_state_dict`` (copy_ into the views), after which the
    packed-weight cache of the HIP ops is dropped."""
    raw = torch.load(path, map_location='cpu', weights_only=True)
    rs = RestoreState(epoch_idx=int(raw['epoch_idx']), model_jit_state_dict=raw['model_jit_state_dict'],
                      optimizer_state_dict=raw['optimizer_state_dict'],
                      optimizer_scheduler_state_dict=raw['optimizer_scheduler_state_dict'])
    if model is not None:
        model.load_state_dict(rs.model_jit_state_dict)
        FlatBuffers.notify_params_changed()
    if optimizer is not None:
        load_optimizer_state_dict(optimizer, rs.optimizer_state_dict)
    return rs


def build_model_from_state_dict_path(state_dict_path, model_config, device='cuda', compute_dtype=torch.bfloat16):
    """train.py:608-632 (``build_model_jit_from_state_dict_path``) without the TorchScript step: the eager module with the
    checkpoint's parameters, in eval mode."""
    from ..model import AdaptiveScaling
    model = AdaptiveScaling(model_config, compute_dtype=compute_dtype)
    load_restore_state(state_dict_path, model)
    return model.to(device).eval()
