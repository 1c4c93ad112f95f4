"""Optimizer side of the train step (train.py:287-298,468-478): global-norm clipping + AdamW on the flat buffers
(two HIP kernels), and the cosine-annealing-with-warm-restarts learning-rate rule evaluated at fractional epochs."""
import ctypes
import math
from typing import Iterable, Optional, Tuple

import torch
from torch import nn

from .flat import FlatBuffers
from .. import ops
from .._lib import lib, check


def cosine_warm_restarts_lr(epoch: float, base_lr: float, eta_min: float, t_0: int, t_mult: int) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingWarmRestarts.step(epoch) closed form, as the reference drives it with
    ``epoch_idx + (batch_idx - 1) / num_batches`` (train.py:475-477; T_0=10, T_mult=10, eta_min=8e-6 by default)."""
    if epoch < 0:
        raise ValueError('epoch must be non-negative')
    if epoch >= t_0:
        if t_mult == 1:
            t_cur, t_i = epoch % t_0, t_0
        else:
            n = int(math.log(epoch / t_0 * (t_mult - 1) + 1, t_mult))
            t_cur = epoch - t_0 * (t_mult ** n - 1) / (t_mult - 1)
            t_i = t_0 * t_mult ** n
    else:
        t_cur, t_i = epoch, t_0
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / t_i)) / 2


def all_or_none(flags: bytearray):
    """True if every flag is set, False if none is, None for a mix."""
    n = sum(flags)
    return True if n == len(flags) else (False if n == 0 else None)


class FlatAdamW:
    """AdamW(lr 8e-4, betas (0.9, 0.999), wd 0.01) with clip_grad_norm_(max_norm 2.5) (train.py:72-80,468-478)."""

    def __init__(self, params, lr: float = 8e-4, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, max_grad_norm: Optional[float] = 2.5, flat: Optional[FlatBuffers] = None):
        if flat is None:
            params = list(params)
            if params and isinstance(params[0], nn.Parameter):
                params = [(f'p{i}', p) for i, p in enumerate(params)]
            flat = FlatBuffers(params)
        self.flat = flat
        if not flat.flat_param.is_cuda:
            raise RuntimeError('FlatAdamW runs on the MI355X only (no CPU fallback)')
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.exp_avg = torch.zeros_like(flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(flat.flat_param)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=flat.flat_param.device)
        self.step_count = 0
        # which parameters hold optimizer state, i.e. were updated by some step so far: torch.optim.AdamW creates a parameter's
        # state at its first step with a gradient (an exactly-zero gradient included) and never for a parameter whose .grad
        # stayed None; the RestoreState export goes by this record (training/checkpoint.py)
        self.has_state = bytearray(len(flat.params))
        # hyper-parameters restored from a checkpoint win over a config record on resume (training/harness.py)
        self.restored_from_state = False

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0):
        f = self.flat
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.step_count += 1
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        sumsq = None
        if self.max_grad_norm is not None:
            check(lib.vkas_l2norm_sq(ptr(f.flat_grad), f.numel, ptr(self.sumsq), st), 'l2norm_sq')
            sumsq = ptr(self.sumsq)
        # torch.optim.AdamW leaves parameters without a gradient untouched (no decay, no moment update): e.g. the
        # precise mask head under precise_enable_char_mask_head, which forward_precise never runs.  One launch per
        # contiguous run of parameters that did receive a gradient (normally the whole buffer).
        # (no hook fired at all = the gradients were written into the flat buffer by hand: update everything)
        partial = all_or_none(f.touched) is None
        ranges = f.touched_ranges() if partial else [(0, f.numel)]
        for i in range(len(self.has_state)):
            if not partial or f.touched[i]:
                self.has_state[i] = 1
        for start, end in ranges:
            off = lambda t: ctypes.c_void_p(t.data_ptr() + 4 * start)
            check(lib.vkas_adamw_step(off(f.flat_param), off(f.flat_grad), off(self.exp_avg), off(self.exp_avg_sq),
                                      end - start, sumsq, float(self.max_grad_norm or 0.0), float(grad_scale),
                                      float(self.lr if lr is None else lr), self.betas[0], self.betas[1], self.eps,
                                      self.weight_decay, self.step_count, st), 'adamw_step')
        ops.refresh_packed_params()  # parameters changed behind autograd's version counters: rebuild their packed images

    def grad_norm(self) -> float:
        """Host-visible total norm of the last step (forces a sync; for logging only)."""
        return float(self.sumsq.sqrt())

    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.exp_avg.clone(), 'exp_avg_sq': self.exp_avg_sq.clone()}

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
