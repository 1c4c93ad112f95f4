"""Flat parameter / gradient storage.

All parameters of a model become views into one fp32 buffer and all ``.grad`` tensors views into a second one
(registration order, each tensor aligned to 16 bytes).  The gradient buffer is what RCCL all-reduces (contiguous
ranges = buckets) and what the fused clip + AdamW kernels walk; autograd accumulates into the views in place, so the
two backward passes of a step (train.py:416,454) sum without extra copies.  Device-agnostic (the CPU/gloo tests of the
reducer use it too).

Writing parameters.  The HIP ops cache packed (bf16, GEMM-layout) copies of the weights between the uses of one
parameter value (ops._PACK_CACHE) and recognise a change by the parameter's autograd version counter.  Writes that go
around that counter - anything through ``flat_param`` (checkpoint load, rank-0 broadcast) or through ``p.data`` (EMA
swaps, manual init) - must be followed by ``notify_params_changed()`` (= ``ops.invalidate_packed_params()``);
``load_flat`` / ``broadcast_params`` below and ``FlatAdamW.step`` do it themselves."""
from typing import Callable, Dict, Iterable, List, Tuple

import torch
from torch import nn

ALIGN = 4  # elements (16 bytes of fp32)


class FlatBuffers:

    def __init__(self, named_params: Iterable[Tuple[str, nn.Parameter]]):
        named_params = [(n, p) for n, p in named_params if p.requires_grad]
        if not named_params:
            raise ValueError('no trainable parameters')
        device = named_params[0][1].device
        self.names: List[str] = []
        self.params: List[nn.Parameter] = []
        self.offsets: Dict[str, Tuple[int, int]] = {}
        total = 0
        for n, p in named_params:
            if p.dtype != torch.float32:
                raise TypeError(f'{n}: master parameters must be fp32')
            self.offsets[n] = (total, p.numel())
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.names.append(n)
            self.params.append(p)
        self.numel = total
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for n, p in named_params:
                start, size = self.offsets[n]
                view = self.flat_param[start:start + size].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[start:start + size].view(p.shape)
        # which parameters received a gradient since the last zero_grad(): torch.optim.AdamW (train.py:287-298) skips
        # parameters whose .grad is None; here .grad is a persistent view, so the hook keeps the equivalent record
        self.touched = bytearray(len(self.params))
        self._subscribers: List[Callable[[str], None]] = []
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._touch_hook(i))
            # the HIP ops add weight gradients straight into this view (no temporary, no autograd add) and then call
            # grad_delivered(i) instead of handing a tensor back to autograd: ops.grad_sink()
            p._vkas_sink = (self, i)

    def _touch_hook(self, i: int):
        def hook(_param):
            self.grad_delivered(i)
        return hook

    def subscribe(self, callback: Callable[[str], None]):
        """callback(name) runs every time parameter ``name`` has received (a contribution to) its gradient, whether
        through autograd's accumulation or through the ops' direct path (the bucketed reducer listens here)."""
        self._subscribers.append(callback)

    def grad_delivered(self, i: int):
        self.touched[i] = 1
        for cb in self._subscribers:
            cb(self.names[i])

    def grad_view_ok(self, i: int) -> bool:
        """True while parameter i's .grad is still the view into the flat gradient buffer."""
        p = self.params[i]
        return p.grad is not None and p.grad.data_ptr() == self.flat_grad.data_ptr() + 4 * self.offsets[self.names[i]][0]

    def touched_ranges(self) -> List[Tuple[int, int]]:
        """Maximal contiguous [start, end) element ranges of parameters that got a gradient since zero_grad()."""
        out: List[Tuple[int, int]] = []
        for i, n in enumerate(self.names):
            if not self.touched[i]:
                continue
            start, size = self.offsets[n]
            end = start + (size + ALIGN - 1) // ALIGN * ALIGN
            if out and out[-1][1] == start:
                out[-1] = (out[-1][0], end)
            else:
                out.append((start, end))
        return out

    @staticmethod
    def notify_params_changed():
        """Drop every packed weight copy the HIP ops cached: call after writing parameters behind autograd's back."""
        from .. import ops
        ops.invalidate_packed_params()

    def load_flat(self, values: torch.Tensor):
        """Overwrite all parameters from a flat fp32 tensor in this buffer's layout (checkpoint restore)."""
        if values.numel() != self.numel:
            raise ValueError(f'expected {self.numel} elements, got {values.numel()}')
        with torch.no_grad():
            self.flat_param.copy_(values.reshape(-1))
        self.notify_params_changed()

    def broadcast_params(self, src: int = 0, group=None):
        """Make every rank hold rank ``src``'s parameters (what DistributedDataParallel does at construction)."""
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat_param, src=src, group=group)
        self.notify_params_changed()

    def zero_grad(self):
        self.flat_grad.zero_()
        if self.flat_grad.is_cuda:
            from .. import ops
            ops.zero_arena_reset()  # the step's weight-gradient scratch is dead by now: one fill re-arms it
        self.touched = bytearray(len(self.params))
        for n, p in zip(self.names, self.params):  # re-attach views if someone set .grad to None
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * self.offsets[n][0]:
                start, size = self.offsets[n]
                p.grad = self.flat_grad[start:start + size].view(p.shape)

    def range_of(self, prefixes: Tuple[str, ...]) -> Tuple[int, int]:
        """Smallest [start, end) element range covering every parameter whose name starts with one of ``prefixes``;
        raises if a foreign parameter lies inside (ranges must be contiguous to be all-reduced in one call)."""
        sel = [n for n in self.names if n.startswith(prefixes)]
        if not sel:
            raise KeyError(prefixes)
        start = min(self.offsets[n][0] for n in sel)
        end = max(self.offsets[n][0] + (self.offsets[n][1] + ALIGN - 1) // ALIGN * ALIGN for n in sel)
        inside = [n for n in self.names if start <= self.offsets[n][0] < end]
        if set(inside) != set(sel):
            raise ValueError(f'parameters {prefixes} are not contiguous in the flat buffer')
        return start, end
