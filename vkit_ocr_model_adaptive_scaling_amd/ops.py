"""Autograd ops of the adaptive-scaling hot path, backed by libvkas.so (HIP, gfx950).

Activations travel between ops as NHWC tensors of shape (B, H, W, Cp) in the compute dtype (bf16 or
fp32), Cp = channels rounded up to a multiple of 8 with zero pad channels; a channel slice of a wider
tensor (pixel stride ld > Cp) is accepted everywhere, which is how concatenation stays copy-free on
the consumer side.  Parameters stay fp32 in the reference's own layouts; each op converts what it
needs with the pack kernels.  Every op launches on torch's current HIP stream and allocates through
torch's caching allocator; nothing here falls back to torch math or to the CPU.
"""
import ctypes
import os
import weakref
from typing import Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import _lib
from ._lib import lib, check, ConvGeom, Epilogue

_FLOAT = torch.float32


def rup8(c: int) -> int:
    return (c + 7) // 8 * 8


_DTYPE_CODES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}
_MFMA_DTYPES = (torch.bfloat16, torch.float16)  # 16-bit storage types with MFMA kernels (fp32 accumulate)


def _dtc(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODES[dtype]
    except KeyError:
        raise TypeError(f'unsupported activation dtype {dtype}') from None


def _dt(t: torch.Tensor) -> int:
    return _dtc(t.dtype)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('vkas ops run on the MI355X only (tensor is on %s); there is no CPU fallback' % t.device)


def act_ok(t: torch.Tensor) -> bool:
    if t.dim() != 4 or t.shape[3] % 8 != 0:
        return False
    B, H, W, C = t.shape
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else (t.stride(0) if B > 1 else C))
    if t.stride(3) != 1 and C > 1:
        return False
    if ld < C or ld % 8 != 0 or t.data_ptr() % 16 != 0:
        return False
    if W > 1 and t.stride(2) != ld:
        return False
    if H > 1 and t.stride(1) != W * ld:
        return False
    if B > 1 and t.stride(0) != H * W * ld:
        return False
    return True


def act_ld(t: torch.Tensor) -> int:
    B, H, W, C = t.shape
    if W > 1:
        return t.stride(2)
    if H > 1:
        return t.stride(1)
    if B > 1:
        return t.stride(0)
    return C


def as_act(t: torch.Tensor) -> torch.Tensor:
    """Return t if it is a valid (possibly channel-sliced) NHWC activation, else a dense copy."""
    return t if act_ok(t) else t.contiguous()


def new_act(B, H, W, Cp, like: torch.Tensor) -> torch.Tensor:
    return torch.empty((B, H, W, Cp), dtype=like.dtype, device=like.device)


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(((nbytes + 3) // 4 + 4,), dtype=_FLOAT, device=device)


class _ZeroArena:
    """Zero-initialised fp32 scratch for the atomically accumulated weight-gradient images of one training step: the 15
    per-layer ``torch.zeros`` of a backward pass become slices of one buffer that a single fill clears once per step
    (``zero_arena_reset``, called by ``FlatBuffers.zero_grad``).  A slice is handed out once between two resets, so it is
    clean when its kernel starts and may be dirty afterwards; callers ask for arena memory only for buffers that do not
    outlive the step (their contents are unpacked / added into the flat gradient before the optimizer runs).  Without resets
    (no flat optimizer) every request falls back to ``torch.zeros``."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.off = 0      # floats handed out since the last reset
        self.want = 0     # floats asked for since the last reset (sizes the buffer at the next one)
        self.live = False

    def take(self, n: int, device) -> Optional[torch.Tensor]:
        n_al = -(-n // 64) * 64  # 256-byte aligned slices
        self.want += n_al
        if not self.live or self.buf.device != device or self.off + n_al > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n]
        self.off += n_al
        return t

    def reset(self):
        if self.want == 0 and self.off == 0:
            return  # nothing was asked for since the last reset: the buffer (if any) is still clean
        if self.buf is None or self.want > self.buf.numel():
            dev = self.buf.device if self.buf is not None else torch.device('cuda', torch.cuda.current_device())
            self.buf = None
            self.buf = torch.zeros((self.want,), dtype=_FLOAT, device=dev)
        else:
            self.buf.zero_()
        self.off, self.want, self.live = 0, 0, True


_ZERO_ARENA = _ZeroArena()
_NO_ZERO_ARENA = os.environ.get('VKAS_NO_ZERO_ARENA') is not None  # A/B switch


def zero_arena_reset():
    if not _NO_ZERO_ARENA:
        _ZERO_ARENA.reset()


def zeros_f32(n: int, device, step_scratch: bool) -> torch.Tensor:
    """n zeroed floats; step_scratch: the buffer is dead before the step's optimizer update (see _ZeroArena)."""
    if step_scratch and not _NO_ZERO_ARENA:
        t = _ZERO_ARENA.take(n, device)
        if t is not None:
            return t
    return torch.zeros((n,), dtype=_FLOAT, device=device)


# ---------------------------------------------------------------------------------------- raw wrappers
# Packed operands derived from parameters (GEMM weight layouts, padded vectors) are cached between the uses of one
# parameter value: a weight is used by the forward and the backward of both passes of a step, i.e. packed once per
# layout instead of four times.  An entry is valid while the source storage, its autograd version counter and the
# epoch below are unchanged; the fused optimizer writes parameters through the C ABI (no version bump) and bumps the
# epoch instead (training/optimizer.py).
_PACK_CACHE = {}
_PACK_EPOCH = [0]
# Recipes of the cached images that one vkas_pack_many launch can rebuild (conv / depthwise weights of leaf parameters):
# cache key -> (list of _lib.PackDesc, parameters as weak references, their data pointers when the recipe was recorded)
_PACK_PLAN = {}
_PACK_TABLE = [None]  # ((source, image) pointers of all entries, device descriptor table, device block starts, entries, workgroups)


def invalidate_packed_params():
    _PACK_EPOCH[0] += 1
    _PACK_CACHE.clear()
    _PACK_PLAN.clear()


def _plan_pack(k, descs, params):
    _PACK_PLAN[k] = (descs, tuple(weakref.ref(p) for p in params), tuple(p.data_ptr() for p in params))


def _conv_desc(w4: torch.Tensor, out: torch.Tensor, Np: int, Cp: int, mode: int, n_off: int, Nt: int, dtype) -> _lib.PackDesc:
    N, C, KH, KW = w4.shape
    return _lib.PackDesc(w4.data_ptr(), out.data_ptr(), Np * KH * KW * Cp, 0, N, C, KH, KW, Np, Cp, mode, n_off, Nt, _dtc(dtype))


def refresh_packed_params():
    """All parameters were just updated in place (the fused optimizer: training/optimizer.py).  Instead of dropping the
    packed operands and rebuilding them one launch at a time during the next step (~150 launches of a few microseconds),
    rebuild every image with a recorded recipe by ONE vkas_pack_many launch - the parameters and the images keep their
    addresses, so the descriptor table of a model is uploaded once - and drop only the rest."""
    _PACK_EPOCH[0] += 1
    live = {}
    for k, (descs, refs, ptrs) in _PACK_PLAN.items():
        params = [r() for r in refs]
        if k in _PACK_CACHE and all(p is not None and p.data_ptr() == q for p, q in zip(params, ptrs)):
            live[k] = (descs, refs, ptrs)
    for k in list(_PACK_CACHE):
        if k not in live:
            del _PACK_CACHE[k]
    _PACK_PLAN.clear()
    _PACK_PLAN.update(live)
    if not live:
        return
    keys = tuple(live)
    descs = [d for k in keys for d in live[k][0]]
    # the uploaded table is valid while it is byte for byte what would be uploaded now: the same entries pointing at the same
    # buffers in the same layouts (after an invalidate + lazy rebuild the allocator may hand a freed image block to another
    # layout of the same weight: equal pointers, different mode)
    sig = b''.join(bytes(d) for d in descs)
    tab = _PACK_TABLE[0]
    if tab is None or tab[0] != sig:
        arr = (_lib.PackDesc * len(descs))(*descs)
        starts = [0]
        for d in descs:
            starts.append(starts[-1] + lib.vkas_pack_many_blocks(ctypes.byref(d)))
        dev = _PACK_CACHE[keys[0]][2].device
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        tab = (sig, table, torch.tensor(starts, dtype=torch.int32, device=dev), len(descs), starts[-1])
        _PACK_TABLE[0] = tab
    check(lib.vkas_pack_many(_p(tab[1]), _p(tab[2]), tab[3], tab[4], _stream()), 'pack_many')
    for k in keys:
        stamp, refs, out = _PACK_CACHE[k]
        if isinstance(refs, tuple):  # _cached_pack_multi
            params = [r() for r in refs]
            stamp = (tuple(p._version for p in params), _PACK_EPOCH[0], tuple(p.data_ptr() for p in params))
        else:
            src = refs()
            stamp = (src._version, _PACK_EPOCH[0], src.data_ptr())
        _PACK_CACHE[k] = (stamp, refs, out)


def _cached_pack(src: torch.Tensor, key, build):
    # only leaf parameters have a stable identity; anything else (temporaries, concatenated head weights) is rebuilt
    if not (src.is_leaf and src.requires_grad):
        return build()
    k = (id(src), key)
    ent = _PACK_CACHE.get(k)
    stamp = (src._version, _PACK_EPOCH[0], src.data_ptr())
    if ent is not None and ent[0] == stamp and ent[1]() is src:  # the weak reference guards against a recycled id()
        return ent[2]
    out = build()
    _PACK_CACHE[k] = (stamp, weakref.ref(src), out)
    return out


def pack_conv_weight(w: torch.Tensor, Np: int, Cp: int, mode: int, dtype: torch.dtype) -> torch.Tensor:
    def build():
        w4 = w if w.dim() == 4 else w.view(w.shape[0], w.shape[1], 1, 1)
        N, C, KH, KW = w4.shape
        out = torch.empty((Np * KH * KW * Cp,), dtype=dtype, device=w.device)
        check(lib.vkas_pack_conv_weight(_p(w4.contiguous()), _p(out), N, C, KH, KW, Np, Cp, mode,
                                        _dtc(dtype), _stream()), 'pack_conv_weight')
        if w.is_leaf and w.requires_grad and w4.is_contiguous():
            _plan_pack((id(w), key), [_conv_desc(w4, out, Np, Cp, mode, 0, Np, dtype)], [w])
        return out
    key = ('conv', Np, Cp, mode, dtype)
    return _cached_pack(w, key, build)


def _cached_pack_pair(a: torch.Tensor, b: torch.Tensor, key, build):
    """_cached_pack for an operand derived from two parameters (the fused MLP's weight image)."""
    if not (a.is_leaf and a.requires_grad and b.is_leaf and b.requires_grad):
        return build()
    k = (id(a), id(b), key)
    ent = _PACK_CACHE.get(k)
    stamp = (a._version, b._version, _PACK_EPOCH[0], a.data_ptr(), b.data_ptr())
    if ent is not None and ent[0] == stamp and ent[1]() is a and ent[3]() is b:
        return ent[2]
    out = build()
    _PACK_CACHE[k] = (stamp, weakref.ref(a), out, weakref.ref(b))
    return out


_NO_CHAIN = os.environ.get('VKAS_NO_MLP_CHAIN') is not None  # A/B switch: force the two-GEMM layer path


# The kernels exist up to 512 channels.  Up to 256 channels and - round 4, the pair-split kernel with two 256-register waves per
# SIMD (csrc/mlp_chain.hip::mlp_chain_pair_kernel) - for 256 < C <= 384 with C % 16 == 0 (stage 2 of ConvNeXt-T / -S) the
# fused kernels beat the two-GEMM path (stage 2 of config #3: 0.23 + 0.21 ms per layer against 0.29 + 0.29); at 512 channels
# (ConvNeXt-Base) the one-wave-per-SIMD instantiation measured no faster than two GEMMs, so that width stays on the GEMMs.
_CHAIN_MAX_C = int(os.environ.get('VKAS_MLP_CHAIN_MAX_C', '384'))


def _chain_width_ok(C: int) -> bool:
    return C <= min(_CHAIN_MAX_C, 256) or (256 < C <= min(_CHAIN_MAX_C, 384) and C % 16 == 0) or (384 < C <= _CHAIN_MAX_C)


# The pair-split kernel walks the 4C hidden units of its 128 rows serially (48 chunks at C = 384: 85 us however few rows there
# are); below this many rows - the forward-only configurations, M = 6 400 at stage 2 of config #2 - the two ring GEMMs
# (csrc/gemm_mfma.hip::gemm_nt_ring_kernel: 25 + 20 us) are faster.
_NO_CHAIN_LN = os.environ.get('VKAS_NO_CHAIN_LN') is not None  # A/B switch: LayerNorm as its own launch in front of the fused MLP
_CHAIN_PAIR_MIN_ROWS = int(os.environ.get('VKAS_MLP_CHAIN_PAIR_MIN_ROWS', '16384'))


def mlp_chain_eligible(x: torch.Tensor, C: int) -> bool:
    """The fused ConvNeXt MLP kernels (csrc/mlp_chain.hip) cover 16-bit activations with C % 8 == 0, C <= 512."""
    if 256 < C <= 384 and x.shape[0] * x.shape[1] * x.shape[2] < _CHAIN_PAIR_MIN_ROWS:
        return False
    return (not _NO_CHAIN and x.dtype in _MFMA_DTYPES and x.shape[3] == C and _chain_width_ok(C)
            and lib.vkas_mlp_chain_image_elems(C) > 0)


def pack_mlp_chain(w1: torch.Tensor, w2: torch.Tensor, b1: Optional[torch.Tensor], C: int, mode: int,
                   dtype: torch.dtype) -> torch.Tensor:
    """Packed, pre-swizzled weight image of the fused MLP (mode 0 forward incl. b1, 1 backward)."""
    def build():
        img = torch.empty((lib.vkas_mlp_chain_image_elems(C),), dtype=dtype, device=w1.device)
        check(lib.vkas_mlp_chain_pack(_p(w1.contiguous()), _p(w2.contiguous()), _p(b1.contiguous()) if mode == 0 else None, C,
                                      mode, _p(img), _dtc(dtype), _stream()), 'mlp_chain_pack')
        return img
    if mode == 0:  # the forward image also depends on b1: key on (w1, w2) and stamp b1's version into the key
        return _cached_pack_pair(w1, w2, ('chain', C, mode, dtype, b1._version, b1.data_ptr()), build)
    return _cached_pack_pair(w1, w2, ('chain', C, mode, dtype), build)


def pack_dw_weight(w: torch.Tensor, C: int, Cp: int, flip: int) -> torch.Tensor:
    def build():
        out = torch.empty((lib.vkas_dw_weight_elems(Cp),), dtype=_FLOAT, device=w.device)
        check(lib.vkas_pack_dw_weight(_p(w.contiguous()), _p(out), C, Cp, flip, _stream()), 'pack_dw_weight')
        if w.is_leaf and w.requires_grad and w.is_contiguous():
            _plan_pack((id(w), key), [_lib.PackDesc(w.data_ptr(), out.data_ptr(), 105 * Cp, 1, 0, C, 0, 0, 0, Cp, flip, 0, 0, 0)], [w])
        return out
    key = ('dw', C, Cp, flip)
    return _cached_pack(w, key, build)


def pad_vector(v: Optional[torch.Tensor], npad: int) -> Optional[torch.Tensor]:
    if v is None:
        return None
    if v.dim() == 1 and v.numel() == npad and v.is_contiguous():
        return v

    def build():
        flat = v.reshape(-1)
        if flat.numel() == npad and flat.is_contiguous():
            return flat
        out = torch.empty((npad,), dtype=_FLOAT, device=v.device)
        check(lib.vkas_pad_vector(_p(flat.contiguous()), _p(out), flat.numel(), npad, _stream()), 'pad_vector')
        return out
    return _cached_pack(v, ('pad', npad), build)


def _geom(B, Hin, Win, Hout, Wout, Cp, ldx, KH, KW, stride, pad) -> ConvGeom:
    return ConvGeom(B, Hin, Win, Hout, Wout, Cp, ldx, KH, KW, stride, pad)


class LaunchTimer:
    """Optional per-launch HIP-event timing of the implicit-GEMM kernels (bench.py's roofline leg).  Events are
    recorded on torch's current stream, which is the stream the kernels are launched on."""

    def __init__(self):
        self.records = []  # (kind, start_event, end_event, flops, M, N, K, algorithmic bytes)

    def summary(self):
        out = {}
        for kind, s, e, flops, M, N, K, _ in self.records:
            d = out.setdefault(kind, {'launches': 0, 'ms': 0.0, 'flops': 0.0})
            d['launches'] += 1
            d['ms'] += s.elapsed_time(e)
            d['flops'] += flops
        return out


TIMER: Optional[LaunchTimer] = None


def nt_kernel_name(kid: int, head: bool) -> str:
    """rocprofv3-style short name (profiles/parse_pmc.py) of the kernel behind a vkas_conv_gemm_kernel_id() code."""
    if kid == 0:
        return 'gemm_nt_simple'
    if kid >= 1000:
        return 'conv3x3_slab_mfma_kernel<%d,%d>' % (kid - 1000, int(head))
    if 12 <= kid <= 14:
        return 'gemm_nt_ring_kernel<%d>' % (kid - 10)
    return 'gemm_nt_mfma_kernel<%s>' % {1: '2,2,4,4', 128: '4,2,4,4', 192: '4,2,4,6', 224: '4,2,4,7'}[kid]


def tn_kernel_name(kid: int) -> str:
    if kid == 0:
        return 'gemm_tn_simple'
    if kid >= 2000:
        return 'conv3x3_wgrad_slab_kernel<%d>' % (kid - 2000)
    return 'gemm_tn_mfma_kernel<%s>' % {128: '2,2,4,4', 192: '2,4,6,4', 224: '2,4,7,4', 384: '4,2,6,4'}[kid]


def _timed(kind, x, flops, M, N, K, fn, nbytes=0.0):
    if TIMER is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    TIMER.records.append((kind, s, e, flops, M, N, K, nbytes))
    return r


def conv_gemm(x: torch.Tensor, geom: ConvGeom, Bw: torch.Tensor, Np: int, out: torch.Tensor, mode=_lib.EPI_NONE,
              bias=None, out2=None, aux=None, colscale=None, rowscale=None, rows_per_image=0, patch=0, patch_hw=(0, 0),
              patch_Cp=0, nk=None, head=None):
    if head is not None:
        assert TIMER is None or True
    if TIMER is not None:
        # algorithmic FLOPs use the logical (unpadded) N and K when the caller knows them
        M = geom.B * geom.Hout * geom.Wout
        N, K = nk if nk is not None else (Np, geom.KH * geom.KW * geom.Cp)
        if x.dtype in _MFMA_DTYPES:
            wmax = max(head.np[i] for i in range(head.n_heads)) if head is not None else 0
            kind = nt_kernel_name(lib.vkas_conv_gemm_kernel_id(0, ctypes.byref(geom), Np, 0, wmax), head is not None)
        else:
            kind = 'gemm_nt_simple'
        es = x.element_size()
        nbytes = (geom.B * geom.Hin * geom.Win * geom.Cp + M * Np * ((out is not None) + (out2 is not None) + (aux is not None))) * es
        return _timed(kind, x, 2.0 * M * N * K, M, N, K,
                      lambda: _conv_gemm(x, geom, Bw, Np, out, mode, bias, out2, aux, colscale, rowscale, rows_per_image,
                                         patch, patch_hw, patch_Cp, head), nbytes)
    return _conv_gemm(x, geom, Bw, Np, out, mode, bias, out2, aux, colscale, rowscale, rows_per_image, patch, patch_hw,
                      patch_Cp, head)


def _conv_gemm(x, geom, Bw, Np, out, mode, bias, out2, aux, colscale, rowscale, rows_per_image, patch, patch_hw,
               patch_Cp, head=None):
    epi = Epilogue(mode, _p(bias).value if bias is not None else None, out.data_ptr() if out is not None else None,
                   act_ld(out) if out is not None else 0,
                   out2.data_ptr() if out2 is not None else None, act_ld(out2) if out2 is not None else 0,
                   aux.data_ptr() if aux is not None else None, act_ld(aux) if aux is not None else 0,
                   colscale.data_ptr() if colscale is not None else None,
                   rowscale.data_ptr() if rowscale is not None else None, rows_per_image, patch, patch_hw[0],
                   patch_hw[1], patch_Cp)
    if head is not None:
        epi.head = head
    check(lib.vkas_conv_gemm_fwd(_p(x), ctypes.byref(geom), _p(Bw), Np, ctypes.byref(epi), _dt(x), _stream()),
          'conv_gemm_fwd')
    return out


def grad_sink(param: Optional[torch.Tensor]):
    """(FlatBuffers, index) when ``param`` is a parameter whose .grad is a live view of a flat gradient buffer
    (training.FlatBuffers): the backward ops then add its gradient in place - the weight-gradient GEMM accumulates into the
    view directly when the packed layout equals the reference layout, else the unpack kernel does - and report the
    delivery to the buffer instead of returning a tensor for autograd to add."""
    s = getattr(param, '_vkas_sink', None) if param is not None else None
    return s if s is not None and s[0].grad_view_ok(s[1]) else None


def deliver_small_grads(pairs):
    """pairs: [(parameter, gradient tensor or None)].  Gradients of parameters with a flat .grad view (grad_sink) are
    added onto it by ONE vkas_accumulate_many launch and reported as delivered; returns the list to hand to autograd
    (None in those places, the tensor itself otherwise)."""
    out, todo = [], []
    for param, g in pairs:
        s = grad_sink(param) if g is not None else None
        if s is None or g.dtype != _FLOAT or g.numel() != param.numel():
            out.append(g)
        else:
            todo.append((s, param, g.contiguous()))
            out.append(None)
    for i in range(0, len(todo), 16):
        part = todo[i:i + 16]
        n = len(part)
        src = (ctypes.c_void_p * n)(*[g.data_ptr() for _, _, g in part])
        dst = (ctypes.c_void_p * n)(*[p.grad.data_ptr() for _, p, _ in part])
        cnt = (ctypes.c_int * n)(*[g.numel() for _, _, g in part])
        check(lib.vkas_accumulate_many(n, src, dst, cnt, _stream()), 'accumulate_many')
    for s, _, _ in todo:
        s[0].grad_delivered(s[1])
    return out


def conv_wgrad(x: torch.Tensor, geom: ConvGeom, dy: torch.Tensor, Np: int, nk=None, with_bias: bool = False,
               x_gelu: bool = False, gw_into: Optional[torch.Tensor] = None, gb_into: Optional[torch.Tensor] = None,
               step_scratch: bool = False, ordered: bool = False):
    """Weight gradient in the packed (Np, K) layout; with_bias also returns the fused bias gradient (Np,): the two
    live in one zero-filled buffer so a single memset covers both.  x_gelu: the input operand is gelu(x).
    gw_into / gb_into: existing fp32 buffers to ACCUMULATE into (the kernels add with atomics) instead of fresh zeros.
    step_scratch: the caller consumes the result before the step ends (zeros_f32).
    ordered: no split over the reduced rows (vkas_conv_gemm_wgrad_ordered) - for a result that is rounded to a 16-bit
    tensor further down, where the last-bit spread of atomically added partial sums would flip roundings from run to run."""
    assert not (ordered and (with_bias or x_gelu or gw_into is not None))
    K = geom.KH * geom.KW * geom.Cp
    n_new = (0 if gw_into is not None else Np * K) + (Np if with_bias and gb_into is None else 0)
    buf = zeros_f32(n_new, x.device, step_scratch) if n_new else None
    gw = gw_into if gw_into is not None else buf[:Np * K]
    gb = None
    if with_bias:
        gb = gb_into if gb_into is not None else buf[n_new - Np:]
    M = geom.B * geom.Hout * geom.Wout

    def run():
        if ordered:
            check(lib.vkas_conv_gemm_wgrad_ordered(_p(x), ctypes.byref(geom), _p(dy), act_ld(dy), Np, _p(gw), _dt(x), _stream()),
                  'conv_gemm_wgrad_ordered')
            return gw
        fn = lib.vkas_conv_gemm_wgrad_gelu if x_gelu else lib.vkas_conv_gemm_wgrad
        check(fn(_p(x), ctypes.byref(geom), _p(dy), act_ld(dy), Np, _p(gw), _p(gb), _dt(x), _stream()), 'conv_gemm_wgrad')
        return (gw, gb) if with_bias else gw
    if x.dtype in _MFMA_DTYPES:
        kid = lib.vkas_conv_gemm_kernel_id(1, ctypes.byref(geom), Np, act_ld(dy), 0)
        if x_gelu and kid >= 2000:  # the gelu-on-load variant exists for the generic kernel only
            kid = lib.vkas_conv_gemm_tile(1, M, Np, K)
        kind = tn_kernel_name(kid)
    else:
        kind = 'gemm_tn_simple'
    N, Kl = nk if nk is not None else (Np, K)
    nbytes = (geom.B * geom.Hin * geom.Win * geom.Cp + M * Np) * x.element_size() + 4.0 * Np * K
    return _timed(kind, x, 2.0 * M * N * Kl, M, N, Kl, run, nbytes)


def unpack_wgrad(gw: torch.Tensor, shape4, Np: int, Cp: int, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Packed (Np, KH, KW, Cp) weight gradient -> reference layout (N, C, KH, KW); ``into``: add onto that tensor."""
    N, C, KH, KW = shape4
    g = torch.empty((N, C, KH, KW), dtype=_FLOAT, device=gw.device) if into is None else into
    check(lib.vkas_unpack_conv_wgrad(_p(gw), _p(g), N, C, KH, KW, Np, Cp, int(into is not None), _stream()),
          'unpack_conv_wgrad')
    return g


def conv_param_grads(x, geom, dy, Np: int, weight: torch.Tensor, bias: Optional[torch.Tensor], x_gelu: bool = False):
    """Weight and bias gradient of an implicit-GEMM convolution / Linear for autograd: (gw, gb) in the reference layouts,
    with None where the gradient went straight into the parameter's flat .grad view (grad_sink)."""
    w4 = weight if weight.dim() == 4 else weight.view(weight.shape[0], weight.shape[1], 1, 1)
    N, C, KH, KW = w4.shape
    Cp = geom.Cp
    nk = (N, C * KH * KW)
    sw, sb = grad_sink(weight), grad_sink(bias)
    same_layout = KH == 1 and KW == 1 and Np == N and Cp == C  # packed (Np, K) rows are the reference rows
    gw_into = weight.grad.view(-1) if (sw is not None and same_layout) else None
    gb_into = bias.grad if (sb is not None and Np == N) else None
    # with flat gradient sinks nothing of the packed buffer survives the step: the weight image is unpacked into the sink and
    # a bias gradient that is returned to autograd is added onto the sink's view in place (never adopted as .grad)
    scratch = sw is not None and (bias is None or sb is not None)
    r = conv_wgrad(x, geom, dy, Np, nk=nk, with_bias=bias is not None, x_gelu=x_gelu, gw_into=gw_into, gb_into=gb_into,
                   step_scratch=scratch)
    gwp, gbp = r if bias is not None else (r, None)
    if gw_into is not None:
        gw = None
    elif sw is not None:
        unpack_wgrad(gwp, (N, C, KH, KW), Np, Cp, into=weight.grad.view(N, C, KH, KW))
        gw = None
    else:
        gw = unpack_wgrad(gwp, (N, C, KH, KW), Np, Cp).view(weight.shape)
    gb = None
    if bias is not None and gb_into is None:
        # a slice of the step's zero arena must not reach autograd: AccumulateGrad ADOPTS an incoming tensor when the
        # parameter's .grad is None (set_to_none), and the next arena reset would zero the adopted gradient
        gb = gbp[:N].clone() if scratch else gbp[:N]
    if sw is not None:
        sw[0].grad_delivered(sw[1])
    if gb_into is not None:
        sb[0].grad_delivered(sb[1])
    return gw, gb


def colsum(y: torch.Tensor, n_logical: int) -> torch.Tensor:
    B, H, W, Np = y.shape
    M = B * H * W
    out = torch.empty((Np,), dtype=_FLOAT, device=y.device)
    nbytes = lib.vkas_colsum_ws_bytes(M, Np)
    ws = _ws(nbytes, y.device)
    check(lib.vkas_colsum(_p(y), act_ld(y), M, Np, _p(out), 0, _p(ws), nbytes, _dt(y), _stream()), 'colsum')
    return out[:n_logical]


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, C: int, act_gelu: bool,
                  out: Optional[torch.Tensor] = None):
    B, H, W, Cp = x.shape
    M = B * H * W
    y = new_act(B, H, W, Cp, x) if out is None else out
    stats = torch.empty((M, 2), dtype=_FLOAT, device=x.device)
    gamma, beta = pad_vector(gamma, Cp), pad_vector(beta, Cp)
    check(lib.vkas_layernorm_fwd(_p(x), act_ld(x), _p(gamma), _p(beta), _p(y), act_ld(y), _p(stats), M, C, Cp,
                                 int(act_gelu), _dt(x), _stream()), 'layernorm_fwd')
    return y, stats


def layernorm_bwd(x, gamma, beta, stats, dy, C: int, act_gelu: bool, defer: bool = False):
    """defer: leave the per-workgroup partial rows of dgamma | dbeta in the workspace and return (dx, ws, parts) - the caller
    sums them (finalize_many), typically straight into the parameters' gradient views."""
    B, H, W, Cp = x.shape
    M = B * H * W
    dx = new_act(B, H, W, Cp, x)
    nbytes = lib.vkas_layernorm_bwd_ws_bytes(M, Cp)
    ws = _ws(nbytes, x.device)
    gamma, beta = pad_vector(gamma, Cp), pad_vector(beta, Cp)
    if defer:
        check(lib.vkas_layernorm_bwd(_p(x), act_ld(x), _p(gamma), _p(beta), _p(stats), _p(dy), act_ld(dy), _p(dx),
                                     act_ld(dx), None, None, _p(ws), nbytes, M, C, Cp, int(act_gelu), _dt(x),
                                     _stream()), 'layernorm_bwd')
        return dx, ws, lib.vkas_layernorm_bwd_parts(M, Cp)
    dgb = torch.empty((2 * Cp,), dtype=_FLOAT, device=x.device)  # contiguous pair -> one finalize launch
    dg, db = dgb[:Cp], dgb[Cp:]
    check(lib.vkas_layernorm_bwd(_p(x), act_ld(x), _p(gamma), _p(beta), _p(stats), _p(dy), act_ld(dy), _p(dx),
                                 act_ld(dx), _p(dg), _p(db), _p(ws), nbytes, M, C, Cp, int(act_gelu), _dt(x),
                                 _stream()), 'layernorm_bwd')
    return dx, dg[:C], db[:C]


_NO_FINALIZE_MANY = os.environ.get('VKAS_NO_FINALIZE_MANY') is not None  # A/B switch: per-op finalize + accumulate launches


def finalize_many(items):
    """items: [(workspace tensor, first column, partial rows, columns, row pitch, out tensor, accumulate)] -> ONE launch of
    vkas_finalize_many per 8 reductions (second-stage column sums, optionally added onto ``out``)."""
    for i in range(0, len(items), 8):
        part = items[i:i + 8]
        n = len(part)
        src = (ctypes.c_void_p * n)(*[ws.data_ptr() + 4 * col for ws, col, _, _, _, _, _ in part])
        P = (ctypes.c_long * n)(*[p for _, _, p, _, _, _, _ in part])
        nn = (ctypes.c_int * n)(*[c for _, _, _, c, _, _, _ in part])
        ld = (ctypes.c_int * n)(*[l for _, _, _, _, l, _, _ in part])
        dst = (ctypes.c_void_p * n)(*[o.data_ptr() for _, _, _, _, _, o, _ in part])
        acc = (ctypes.c_int * n)(*[int(a) for _, _, _, _, _, _, a in part])
        check(lib.vkas_finalize_many(n, src, P, nn, ld, dst, acc, _stream()), 'finalize_many')


def small_grad_views(params, Cp: int):
    """The parameters' flat gradient views if EVERY one of them can take a Cp-wide column sum directly (FlatBuffers attached,
    no channel padding), else None."""
    if _NO_FINALIZE_MANY:
        return None
    sinks = [grad_sink(p) for p in params]
    if any(s is None for s in sinks):
        return None
    if any(p.numel() != Cp or not p.grad.is_contiguous() or p.grad.dtype != _FLOAT for p in params):
        return None
    return sinks


def resize_fwd(x, size: Tuple[int, int], mode: int, out=None, accumulate=False):
    B, Hin, Win, Cp = x.shape
    Hout, Wout = size
    y = new_act(B, Hout, Wout, Cp, x) if out is None else out
    nbytes = B * Cp * x.element_size() * (Hin * Win + Hout * Wout * (2 if accumulate else 1))
    x2 = mode == 0 and Hout == 2 * Hin and Wout == 2 * Win and Hin > 1 and Win > 1  # the library's exact-x2 kernels
    _timed('resize2x_fwd_kernel' if x2 else 'resize_fwd_kernel', x, 0.0, B * Hout * Wout, Cp, 0,
           lambda: check(lib.vkas_resize_fwd(_p(x), act_ld(x), _p(y), act_ld(y), B, Hin, Win, Hout, Wout, Cp, mode,
                                             int(accumulate), _dt(x), _stream()), 'resize_fwd'), nbytes)
    return y


def resize_bwd(dy, in_size: Tuple[int, int], mode: int):
    B, Hout, Wout, Cp = dy.shape
    Hin, Win = in_size
    dx = new_act(B, Hin, Win, Cp, dy)
    nbytes = B * Cp * dy.element_size() * (Hin * Win + Hout * Wout)
    x2 = mode == 0 and Hout == 2 * Hin and Wout == 2 * Win and Hin > 1 and Win > 1
    wsb = lib.vkas_resize_bwd_ws_bytes(B, Hin, Win, Hout, Wout, Cp)  # > 0: the two-pass (separable) backward of large ratios
    ws = _ws(wsb, dy.device) if wsb else None
    name = 'resize2x_bwd_kernel' if x2 else ('resize_bwd_x+y_kernel' if wsb else 'resize_bwd_kernel')
    _timed(name, dy, 0.0, B * Hout * Wout, Cp, 0,
           lambda: check(lib.vkas_resize_bwd_ws(_p(dy), act_ld(dy), _p(dx), act_ld(dx), _p(ws), wsb, B, Hin, Win, Hout, Wout, Cp,
                                                mode, 0, _dt(dy), _stream()), 'resize_bwd'), nbytes)
    return dx


def copy_channels(x, out, accumulate=False):
    B, H, W, Cp = x.shape
    check(lib.vkas_copy_channels(_p(x), act_ld(x), _p(out), act_ld(out), B * H * W, Cp, int(accumulate), _dt(x),
                                 _stream()), 'copy_channels')
    return out


# ---------------------------------------------------------------------------------------- autograd ops
class ImageToAct(Function):
    """(B,3,H,W) fp32 NCHW raw pixels -> (B,H,W,8) activation (channels 3..7 zero).  The reference's stem is a plain
    nn.Conv2d (convnext.py:106-123), so an input that requires grad receives one: backward converts the stem
    convolution's NHWC input gradient back to (B,3,H,W) fp32.  Training feeds data (train.py:399-403): no gradient is
    computed then (the caller passes need_input_grad=False to the stem convolution as well)."""

    @staticmethod
    def forward(ctx, img: torch.Tensor, dtype: torch.dtype):
        _require_cuda(img)
        if img.dtype != _FLOAT:
            img = img.float()
        img = img.contiguous()
        B, C, H, W = img.shape
        out = torch.empty((B, H, W, 8), dtype=dtype, device=img.device)
        check(lib.vkas_image_nchw_to_nhwc8(_p(img), _p(out), B, C, H, W,
                                           _dtc(dtype), _stream()),
              'image_nchw_to_nhwc8')
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None
        g = as_act(g)
        B, H, W, Cp = g.shape
        out = torch.empty((B, ctx.C, H, W), dtype=_FLOAT, device=g.device)
        check(lib.vkas_nhwc_to_nchw_f32(_p(g), act_ld(g), _p(out), B, H, W, ctx.C, _dt(g), _stream()), 'nhwc_to_nchw_f32')
        return out, None


def images_to_act(imgs: Sequence[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    """Several (B_i,3,H,W) image batches -> ONE (sum B_i, H, W, 8) activation: the layout conversion of each batch writes its
    slice of the result, so the merged pass schedule (model.forward_both) needs no torch.cat of the fp32 images (200 MB copied
    per step at config #3).  Data only: images that require grad take ImageToAct + torch.cat."""
    if any(i.requires_grad for i in imgs) and torch.is_grad_enabled():
        return ImageToAct.apply(torch.cat(list(imgs), 0), dtype)
    _require_cuda(*imgs)
    H, W = imgs[0].shape[2], imgs[0].shape[3]
    out = torch.empty((sum(i.shape[0] for i in imgs), H, W, 8), dtype=dtype, device=imgs[0].device)
    b0 = 0
    for img in imgs:
        if img.dim() != 4 or img.shape[2:] != (H, W):
            raise ValueError('images_to_act: the batches must share their spatial size')
        img = img.detach()
        img = (img if img.dtype == _FLOAT else img.float()).contiguous()
        B, C = img.shape[0], img.shape[1]
        check(lib.vkas_image_nchw_to_nhwc8(_p(img), _p(out[b0:b0 + B]), B, C, H, W, _dtc(dtype), _stream()),
              'image_nchw_to_nhwc8')
        b0 += B
    return out


class Conv(Function):
    """helper.conv1x1 / conv3x3 / pconv2x2 / pconv4x4 (model/helper.py:18-58) as one implicit GEMM.

    x (B,Hin,Win,Cp) activation; weight in the reference layout (N,C,KH,KW) or (N,C); bias (N,) or None.
    Only stride-1 'same' convs (pad = (K-1)/2) and non-overlapping patch convs (stride = K, pad 0) exist on
    this path."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride: int, pad: int, need_input_grad: bool = True):
        _require_cuda(x, weight)
        x = as_act(x)
        w4 = weight if weight.dim() == 4 else weight.view(weight.shape[0], weight.shape[1], 1, 1)
        N, C, KH, KW = w4.shape
        B, Hin, Win, Cp = x.shape
        if Cp != rup8(C):
            raise ValueError(f'Conv: activation has {Cp} (padded) channels, weight expects {C}')
        Hout = (Hin + 2 * pad - KH) // stride + 1
        Wout = (Win + 2 * pad - KW) // stride + 1
        Np = rup8(N)
        Bw = pack_conv_weight(weight, Np, Cp, 0, x.dtype)
        out = new_act(B, Hout, Wout, Np, x)
        geom = _geom(B, Hin, Win, Hout, Wout, Cp, act_ld(x), KH, KW, stride, pad)
        conv_gemm(x, geom, Bw, Np, out, _lib.EPI_NONE, bias=pad_vector(bias, Np), nk=(N, C * KH * KW))
        ctx.save_for_backward(x, weight, bias if bias is not None else torch.empty(0, device=x.device))
        ctx.cfg = (stride, pad, bias is not None, need_input_grad)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        stride, pad, has_bias, need_input_grad = ctx.cfg
        dy = as_act(dy)
        w4 = weight if weight.dim() == 4 else weight.view(weight.shape[0], weight.shape[1], 1, 1)
        N, C, KH, KW = w4.shape
        B, Hin, Win, Cp = x.shape
        _, Hout, Wout, Np = dy.shape
        geom = _geom(B, Hin, Win, Hout, Wout, Cp, act_ld(x), KH, KW, stride, pad)
        gw, gb = conv_param_grads(x, geom, dy, Np, weight, bias if has_bias else None)
        dx = None
        if need_input_grad and ctx.needs_input_grad[0]:
            if stride == 1:
                # dgrad = same-size conv of dy with the 180-degree rotated, in/out swapped kernel
                Bt = pack_conv_weight(weight, Np, Cp, 1, x.dtype)
                dx = new_act(B, Hin, Win, Cp, x)
                g2 = _geom(B, Hout, Wout, Hin, Win, Np, act_ld(dy), KH, KW, 1, KH - 1 - pad)
                conv_gemm(dy, g2, Bt, Cp, dx, _lib.EPI_NONE, nk=(C, N * KH * KW))
            else:
                assert stride == KH == KW and pad == 0
                Bt = pack_conv_weight(weight, Np, Cp, 2, x.dtype)
                full = (Hout * stride == Hin and Wout * stride == Win)
                dx = new_act(B, Hin, Win, Cp, x) if full else torch.zeros_like(x, memory_format=torch.contiguous_format)
                g2 = _geom(B, Hout, Wout, Hout, Wout, Np, act_ld(dy), 1, 1, 1, 0)
                conv_gemm(dy, g2, Bt, KH * KW * Cp, dx, _lib.EPI_PATCH, patch=KH, patch_hw=(Hout, Wout), patch_Cp=Cp,
                          nk=(C * KH * KW, N))
        return dx, gw, gb, None, None, None


class LayerNorm(Function):
    """helper.ln (+ helper.gelu) on an NHWC activation: model/helper.py:96-101."""

    @staticmethod
    def forward(ctx, x, gamma, beta, act_gelu: bool):
        _require_cuda(x, gamma)
        x = as_act(x)
        C = gamma.numel()
        y, stats = layernorm_fwd(x, gamma.contiguous(), beta.contiguous(), C, act_gelu)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.act_gelu = act_gelu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        sinks = small_grad_views((gamma, beta), x.shape[3])
        if sinks is not None:  # the column sums go straight into the gradient views: one launch instead of finalize + add
            Cp = x.shape[3]
            dx, ws, parts = layernorm_bwd(x, gamma.contiguous(), beta.contiguous(), stats, as_act(dy), gamma.numel(),
                                          ctx.act_gelu, defer=True)
            finalize_many([(ws, 0, parts, Cp, 2 * Cp, gamma.grad, True), (ws, Cp, parts, Cp, 2 * Cp, beta.grad, True)])
            for sk in sinks:
                sk[0].grad_delivered(sk[1])
            return dx, None, None, None
        dx, dg, db = layernorm_bwd(x, gamma.contiguous(), beta.contiguous(), stats, as_act(dy), gamma.numel(),
                                   ctx.act_gelu)
        dg, db = deliver_small_grads([(gamma, dg), (beta, db)])
        return dx, dg, db, None


class MultiLayerNorm(Function):
    """Per-head LayerNorm(+GELU) over channel slices of one wide activation.

    The heads of a pass share their input, so their 3x3 convolutions run as ONE implicit GEMM whose output holds
    every head's channels side by side (slice h = rup8(C_h) channels).  This op normalises each slice separately
    (model/upernext.py:39-45 per head) and, in backward, writes each head's input gradient straight into its slice
    of one gradient buffer, so the shared conv sees a single dy: no per-head dgrad tensors, no gradient adds."""

    @staticmethod
    def forward(ctx, x, act_gelu: bool, *affine):
        _require_cuda(x)
        x = as_act(x)
        gammas, betas = affine[0::2], affine[1::2]
        outs, stats, off = [], [], 0
        for g, b in zip(gammas, betas):
            C = g.numel()
            Cp = rup8(C)
            y, st = layernorm_fwd(x[..., off:off + Cp], g, b, C, act_gelu)
            outs.append(y)
            stats.append(st)
            off += Cp
        if off != x.shape[3]:
            raise ValueError(f'MultiLayerNorm: slices cover {off} of {x.shape[3]} channels')
        ctx.save_for_backward(x, *gammas, *betas, *stats)
        ctx.n = len(gammas)
        ctx.act_gelu = act_gelu
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        x, gammas, betas, stats = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n], saved[1 + 2 * n:]
        B, H, W, Ct = x.shape
        dx = new_act(B, H, W, Ct, x)
        grads, off = [], 0
        M = B * H * W
        for g, b, st, dy in zip(gammas, betas, stats, dys):
            C = g.numel()
            Cp = rup8(C)
            dy = as_act(dy)
            xs, dxs = x[..., off:off + Cp], dx[..., off:off + Cp]
            dgb = torch.empty((2 * Cp,), dtype=_FLOAT, device=x.device)
            nbytes = lib.vkas_layernorm_bwd_ws_bytes(M, Cp)
            ws = _ws(nbytes, x.device)
            gp, bp = pad_vector(g, Cp), pad_vector(b, Cp)
            check(lib.vkas_layernorm_bwd(_p(xs), act_ld(xs), _p(gp), _p(bp), _p(st), _p(dy), act_ld(dy), _p(dxs),
                                         act_ld(dxs), _p(dgb[:Cp]), _p(dgb[Cp:]), _p(ws), nbytes, M, C, Cp,
                                         int(ctx.act_gelu), _dt(x), _stream()), 'layernorm_bwd')
            grads.append((dgb[:C], dgb[Cp:Cp + C]))
            off += Cp
        flat = []
        for dg, db in grads:
            flat.extend([dg, db])
        return (dx, None, *flat)


def _cached_pack_multi(params: Sequence[torch.Tensor], key, build):
    """_cached_pack for an operand derived from several parameters (stacked head weights, head affine tables)."""
    if not all(p.is_leaf and p.requires_grad for p in params):
        return build()
    k = (tuple(id(p) for p in params), key)
    ent = _PACK_CACHE.get(k)
    stamp = (tuple(p._version for p in params), _PACK_EPOCH[0], tuple(p.data_ptr() for p in params))
    if ent is not None and ent[0] == stamp and all(r() is p for r, p in zip(ent[1], params)):
        return ent[2]
    out = build()
    _PACK_CACHE[k] = (stamp, tuple(weakref.ref(p) for p in params), out)
    return out


def pack_head_weights(ws: Sequence[torch.Tensor], nps: Sequence[int], Cp: int, mode: int, dtype: torch.dtype) -> torch.Tensor:
    """The heads' 3x3 weights packed side by side (head h = output channels [n0_h, n0_h + np_h)) as one GEMM operand:
    mode 0 forward (Nt, 3, 3, Cp), mode 1 dgrad (Cp, 3, 3, Nt).  Replaces F.pad + torch.cat of the parameters."""
    Nt = sum(nps)

    def build():
        KH, KW = ws[0].shape[2], ws[0].shape[3]
        out = torch.empty((Nt * KH * KW * Cp,), dtype=dtype, device=ws[0].device)
        off = 0
        descs = []
        for w, np_ in zip(ws, nps):
            N, C = w.shape[0], w.shape[1]
            check(lib.vkas_pack_conv_weight_slice(_p(w.contiguous()), _p(out), N, C, KH, KW, np_, Cp, mode, off, Nt,
                                                  _dtc(dtype), _stream()),
                  'pack_conv_weight_slice')
            descs.append(_conv_desc(w, out, np_, Cp, mode, off, Nt, dtype))
            off += np_
        if all(w.is_leaf and w.requires_grad and w.is_contiguous() for w in ws):
            _plan_pack((tuple(id(w) for w in ws), key), descs, list(ws))
        return out
    key = ('heads', tuple(nps), Cp, mode, dtype)
    return _cached_pack_multi(list(ws), key, build)


def pack_head_bias(bs: Sequence[torch.Tensor], cs: Sequence[int], nps: Sequence[int]) -> torch.Tensor:
    """The heads' conv biases side by side, each padded to its np columns (fp32): the bias operand of the fused head GEMM.
    Rebuilt by vkas_pack_many with the other parameter images after an optimizer step (a bias is a 1 x 1 x 1 'weight')."""
    Nt = sum(nps)
    key = ('head_bias', tuple(nps))

    def build():
        t = torch.empty((Nt,), dtype=_FLOAT, device=bs[0].device)
        off, descs = 0, []
        for b, c, np_ in zip(bs, cs, nps):
            src = b.detach().contiguous()
            check(lib.vkas_pad_vector(_p(src), ctypes.c_void_p(t.data_ptr() + 4 * off), c, np_, _stream()), 'pad_vector')
            descs.append(_lib.PackDesc(src.data_ptr(), t.data_ptr(), np_, 0, c, 1, 1, 1, np_, 1, 0, off, Nt, _dtc(_FLOAT)))
            off += np_
        if all(b.is_leaf and b.requires_grad and b.is_contiguous() for b in bs):
            _plan_pack((tuple(id(b) for b in bs), key), descs, list(bs))
        return t
    return _cached_pack_multi(list(bs), key, build)


class HeadsFused(Function):
    """All heads of a pass (model/upernext.py:215-223 or model/fpn.py:165-183, shared upsampled input) as ONE implicit
    GEMM whose epilogue applies each head's LayerNorm -> GELU -> Linear(C -> out_channels) per pixel: forward writes
    only the pre-LN conv output z (needed by backward) and the 1..4 projected channels; the (M, C) activations and
    their gradients never reach HBM.  Backward recomputes them tile-locally (vkas_head_tail_bwd) and feeds one dz into
    the shared conv's dgrad / wgrad.  bf16 / fp16; out_channels <= 4; heads up to 224 channels in the GEMM epilogue, up to
    512 (ConvNeXt-Base / Large) through vkas_head_tail_fwd on z.

    inputs: x (B,H,W,Cp); then per head: conv weight (C_h, Cin, 3, 3), conv bias (C_h), gamma, beta, wproj (oc, C_h),
    bproj (oc).  The weights are packed side by side by the pack kernel (each head padded to a multiple of 8 rows).
    outputs: per head a (B,H,W,8) fp32 activation holding the oc projected channels (columns >= oc are zero)."""

    @staticmethod
    def eligible(x, channels, out_channels) -> bool:
        M = x.shape[0] * x.shape[1] * x.shape[2]
        return (x.dtype in _MFMA_DTYPES and M >= 16384 and len(channels) <= 4 and max(rup8(c) for c in channels) <= 512
                and max(out_channels) <= 4)

    @staticmethod
    def forward(ctx, x, keep: bool, *params):
        """keep: torch.is_grad_enabled() at the call site (Function.forward itself always runs with grad mode off)."""
        _require_cuda(x, params[0])
        x = as_act(x)
        n_heads = len(params) // 6
        ws, bs = params[0::6], params[1::6]
        gammas, betas, wps, bps = params[2::6], params[3::6], params[4::6], params[5::6]
        cs = [g.numel() for g in gammas]
        ocs = [w.shape[0] for w in wps]
        nps = [rup8(c) for c in cs]
        Nt = sum(nps)
        B, H, W, Cp = x.shape
        C = ws[0].shape[1]
        assert all(tuple(w.shape) == (c, C, 3, 3) for w, c in zip(ws, cs)) and Cp == rup8(C)
        M = B * H * W
        wmax = max(nps)
        # up to 224 columns the tail runs in the convolution's epilogue (one 256 x pw tile per head); wider heads
        # (ConvNeXt-Base / Large) take the plain epilogue + vkas_head_tail_fwd over z
        in_epilogue = wmax <= 224
        pw = (128 if wmax <= 128 else (192 if wmax <= 192 else 224)) if in_epilogue else wmax
        dev = x.device

        def build_hp():
            t = torch.empty((n_heads, 6 * pw + 8), dtype=_FLOAT, device=dev)
            for h in range(n_heads):
                check(lib.vkas_pack_head_params(_p(gammas[h].contiguous()), _p(betas[h].contiguous()),
                                                _p(wps[h].contiguous()), _p(bps[h].contiguous()), cs[h], ocs[h], pw,
                                                _p(t[h]), _stream()), 'pack_head_params')
            return t

        hp = _cached_pack_multi(list(gammas) + list(betas) + list(wps) + list(bps), ('head_params', pw), build_hp)
        b_cat = pack_head_bias(bs, cs, nps)
        # z and the row statistics only serve the backward pass: an inference (no-grad) call does not write them
        # (the wide-head path needs z as the tail kernel's input either way)
        z = new_act(B, H, W, Nt, x) if (keep or not in_epilogue) else None
        stats = torch.empty((n_heads, M, 2), dtype=_FLOAT, device=dev) if keep else None
        proj = torch.empty((n_heads, B, H, W, 8), dtype=_FLOAT, device=dev)
        head = _lib.HeadDesc()
        head.n_heads, head.pw = n_heads, pw
        off = 0
        for h in range(n_heads):
            head.n0[h], head.np[h], head.c[h], head.oc[h] = off, nps[h], cs[h], ocs[h]
            off += nps[h]
        head.params, head.stats, head.proj = hp.data_ptr(), stats.data_ptr() if keep else None, proj.data_ptr()
        geom = _geom(B, H, W, H, W, Cp, act_ld(x), 3, 3, 1, 1)
        Bw = pack_head_weights(ws, nps, Cp, 0, x.dtype)
        if in_epilogue:
            conv_gemm(x, geom, Bw, Nt, z, _lib.EPI_HEAD, bias=b_cat, nk=(sum(cs), C * 9), head=head)
        else:
            conv_gemm(x, geom, Bw, Nt, z, _lib.EPI_NONE, bias=b_cat, nk=(sum(cs), C * 9))
            check(lib.vkas_head_tail_fwd(_p(z), Nt, ctypes.byref(head), M, _dt(x), _stream()), 'head_tail_fwd')
        if keep:
            ctx.save_for_backward(x, z, stats, hp, *ws, *bs, *gammas, *betas, *wps, *bps)
        ctx.meta = (cs, ocs, nps, pw, C)
        return tuple(proj[h] for h in range(n_heads))

    @staticmethod
    def backward(ctx, *dprojs):
        cs, ocs, nps, pw, C = ctx.meta
        n_heads = len(cs)
        saved = ctx.saved_tensors
        x, z, stats, hp = saved[:4]
        ws, bs = saved[4:4 + n_heads], saved[4 + n_heads:4 + 2 * n_heads]
        B, H, W, Cp = x.shape
        Nt = z.shape[3]
        M = B * H * W
        K = 9 * Cp
        dev = x.device
        PS = 6 * pw + 8
        offs = [sum(nps[:h]) for h in range(n_heads + 1)]
        dparams = torch.empty((n_heads, PS), dtype=_FLOAT, device=dev)
        # packed weight gradient | bias gradient; with flat gradient sinks both are unpacked / added into the sinks below
        flat_sinks = all(grad_sink(p) is not None for p in list(ws) + list(bs))
        gbuf = zeros_f32(Nt * K + Nt, dev, flat_sinks)
        gwp, gbp = gbuf[:Nt * K], gbuf[Nt * K:]
        dps = []
        for h in range(n_heads):
            dp = dprojs[h]
            dps.append(torch.zeros((B, H, W, 8), dtype=_FLOAT, device=dev) if dp is None else dp.contiguous().float())

        def tail_bwd(h0, h1, z_ptr, ldz, stats_ptr, dp_ptrs, rows, dz):
            """vkas_head_tail_bwd for heads [h0, h1) whose z columns start at z_ptr: dz (rows, width) and dparams[h0:h1]."""
            head = _lib.HeadDesc()
            head.n_heads, head.pw = h1 - h0, pw
            ptrs = (ctypes.c_void_p * 4)()
            for j, h in enumerate(range(h0, h1)):
                head.n0[j], head.np[j], head.c[j], head.oc[j] = offs[h] - offs[h0], nps[h], cs[h], ocs[h]
                ptrs[j] = dp_ptrs[j]
            head.params, head.stats, head.proj = hp.data_ptr() + h0 * PS * 4, stats_ptr, None
            nbytes = lib.vkas_head_tail_bwd_ws_bytes(rows, pw)
            ws_buf = _ws(nbytes, dev)
            width = offs[h1] - offs[h0]
            _timed('head_tail_bwd_kernel', x, 0.0, rows, width, 0,
                   lambda: check(lib.vkas_head_tail_bwd(ctypes.c_void_p(z_ptr), ldz, ctypes.byref(head), ptrs, _p(dz), width,
                                                        ctypes.c_void_p(dparams.data_ptr() + h0 * PS * 4), _p(ws_buf), nbytes,
                                                        rows, _dt(x), _stream()), 'head_tail_bwd'),
                   float(rows) * (2 * width * x.element_size() + (h1 - h0) * 40))

        # Heads whose gradient arrives from a label-point loss (PreciseLoss marks it, see point_sparse) have B*P non-zero
        # rows: they take the compact path below and only the remaining heads pay for the dense backward.
        sp = _point_sparse_run(dprojs, B, H, W) if _POINT_SPARSE else None
        d0, d1 = (0, n_heads) if sp is None else ((0, sp[0]) if sp[0] > 0 else (sp[1], n_heads))
        es = x.element_size()
        dx = new_act(B, H, W, Cp, x) if ctx.needs_input_grad[0] else None
        if d1 > d0:
            Nd = offs[d1] - offs[d0]
            dz = new_act(B, H, W, Nd, x)
            tail_bwd(d0, d1, z.data_ptr() + offs[d0] * es, Nt, stats.data_ptr() + d0 * M * 8,
                     [dps[h].data_ptr() for h in range(d0, d1)], M, dz)
            geom = _geom(B, H, W, H, W, Cp, act_ld(x), 3, 3, 1, 1)
            conv_wgrad(x, geom, dz, Nd, nk=(sum(cs[d0:d1]), C * 9), with_bias=True,
                       gw_into=gwp[offs[d0] * K:offs[d1] * K], gb_into=gbp[offs[d0]:offs[d1]])
            if dx is not None:
                Bt = pack_head_weights(ws[d0:d1], nps[d0:d1], Cp, 1, x.dtype)
                g2 = _geom(B, H, W, H, W, Nd, Nd, 3, 3, 1, 1)
                conv_gemm(dz, g2, Bt, Cp, dx, _lib.EPI_NONE, nk=(C, sum(cs[d0:d1]) * 9))
            del dz
        elif dx is not None:
            dx.zero_()
        if sp is not None:
            s0, s1, py, px = sp
            Ns = offs[s1] - offs[s0]
            P = py.shape[1]
            Mp = -(-(B * P) // 64) * 64
            scratch = torch.empty((M + Mp,), dtype=torch.int32, device=dev)
            pmap, pix = scratch[:M], scratch[M:]
            check(lib.vkas_points_prepare(_p(py), _p(px), B, P, H, W, _p(pmap), _p(pix), Mp, _stream()), 'points_prepare')
            zs = new_act(1, 1, Mp, Ns, x)
            fbuf = torch.empty(((s1 - s0) * Mp * 10,), dtype=_FLOAT, device=dev)
            stats_s, dproj_s = fbuf[:(s1 - s0) * Mp * 2], fbuf[(s1 - s0) * Mp * 2:]
            ptrs = (ctypes.c_void_p * 4)(*[dps[h].data_ptr() for h in range(s0, s1)])
            check(lib.vkas_points_gather_rows(_p(z), Nt, offs[s0], Ns, ctypes.c_void_p(stats.data_ptr() + s0 * M * 8), ptrs,
                                              s1 - s0, M, _p(pix), Mp, _p(zs), _p(stats_s), _p(dproj_s), _dt(x), _stream()),
                  'points_gather_rows')
            dzs = new_act(1, 1, Mp, Ns, x)
            tail_bwd(s0, s1, zs.data_ptr(), Ns, stats_s.data_ptr(),
                     [dproj_s.data_ptr() + j * Mp * 32 for j in range(s1 - s0)], Mp, dzs)
            # weight / bias gradient: (Ns x Mp) . (Mp x 9 Cp) on the gathered 3x3 patches
            xs = new_act(1, 1, Mp, K, x)
            check(lib.vkas_points_gather_patches(_p(x), act_ld(x), Cp, B, H, W, _p(pix), Mp, _p(xs), _dt(x), _stream()),
                  'points_gather_patches')
            g1 = _geom(1, 1, Mp, 1, Mp, K, K, 1, 1, 1, 0)
            conv_wgrad(xs, g1, dzs, Ns, nk=(sum(cs[s0:s1]), C * 9), with_bias=True,
                       gw_into=gwp[offs[s0] * K:offs[s1] * K], gb_into=gbp[offs[s0]:offs[s1]])
            if dx is not None:
                # input gradient: D (Mp x 9 Cp, fp32) = dz_s . W with W (Ns x 9 Cp) the rows of the forward weight image -
                # the reduction runs over W's ROWS, i.e. this is the weight-gradient GEMM shape with dz_s^T as the "dy"
                # operand (fp32 result, no 16-bit rounding of the per-tap terms) - then summed per touched pixel onto dx
                Wf = pack_head_weights(ws, nps, Cp, 0, x.dtype)[offs[s0] * K:offs[s1] * K].view(1, 1, Ns, K)
                dzt = dzs.view(Mp, Ns).t().contiguous().view(1, 1, Ns, Mp)
                g3 = _geom(1, 1, Ns, 1, Ns, K, K, 1, 1, 1, 0)
                D = conv_wgrad(Wf, g3, dzt, Mp, nk=(B * P, C * 9), step_scratch=True, ordered=True)
                check(lib.vkas_points_scatter3x3(_p(D), _p(pix), _p(pmap), Mp, B, H, W, Cp, _p(dx), act_ld(dx), _dt(x),
                                                 _stream()), 'points_scatter3x3')
        gws, gbs = [], []
        for h in range(n_heads):
            gslice = gwp[offs[h] * K:offs[h + 1] * K]
            sw = grad_sink(ws[h])
            if sw is not None:  # straight into the flat gradient view
                unpack_wgrad(gslice, (cs[h], C, 3, 3), nps[h], Cp, into=ws[h].grad)
                sw[0].grad_delivered(sw[1])
                gws.append(None)
            else:
                gws.append(unpack_wgrad(gslice, (cs[h], C, 3, 3), nps[h], Cp))
            gbs.append(gbp[offs[h]:offs[h] + cs[h]])
        # the heads' small gradients (conv bias, LayerNorm affine, projection): with flat .grad views they are added in place by
        # vkas_accumulate_many, 16 contiguous pieces per launch (a projection weight row by row), instead of one autograd add
        # per parameter (30 launches per step)
        small = saved[4 + 2 * n_heads:]
        gammas, betas = small[:n_heads], small[n_heads:2 * n_heads]
        wps, bps = small[2 * n_heads:3 * n_heads], small[3 * n_heads:4 * n_heads]
        grads, pieces, delivered = [], [], []
        for h in range(n_heads):
            d = dparams[h]
            per = [(bs[h], gbs[h], [(gbs[h], 0, cs[h])]), (gammas[h], d[:cs[h]], [(d, 0, cs[h])]),
                   (betas[h], d[pw:pw + cs[h]], [(d, pw, cs[h])]),
                   (wps[h], d[2 * pw:6 * pw].view(4, pw)[:ocs[h], :cs[h]], [(d, (2 + q) * pw, cs[h]) for q in range(ocs[h])]),
                   (bps[h], d[6 * pw:6 * pw + ocs[h]], [(d, 6 * pw, ocs[h])])]
            out = [gws[h]]
            for param, g, rows in per:
                sk = grad_sink(param)
                if sk is None or not param.grad.is_contiguous():
                    # handed to autograd: never a view of the step's zero arena (see conv_param_grads)
                    out.append(g.clone() if (flat_sinks and g is gbs[h]) else g)
                    continue
                off = 0
                for src, so, cnt in rows:
                    pieces.append((src.data_ptr() + 4 * so, param.grad.data_ptr() + 4 * off, cnt))
                    off += cnt
                delivered.append(sk)
                out.append(None)
            grads.extend(out)
        for i in range(0, len(pieces), 16):
            part = pieces[i:i + 16]
            n = len(part)
            src = (ctypes.c_void_p * n)(*[a for a, _, _ in part])
            dst = (ctypes.c_void_p * n)(*[b for _, b, _ in part])
            cnt = (ctypes.c_int * n)(*[c for _, _, c in part])
            check(lib.vkas_accumulate_many(n, src, dst, cnt, _stream()), 'accumulate_many')
        for sk in delivered:
            sk[0].grad_delivered(sk[1])
        return (dx, None, *grads)


class HeadsAtPoints(Function):
    """Opt-in companion of HeadsFused for a TRAINING step: heads whose outputs the loss reads at the label points only
    (the precise pass's offset / angle / distance heads, loss_function/adaptive_scaling.py:167-179,235-262) evaluated AT
    those points only - the 3x3 patches at the (B, P) points are gathered and conv -> LayerNorm -> GELU -> Linear runs on
    B*P rows instead of B*H*W.  The returned (B,H,W,8) maps hold the heads' true outputs at the label points and zeros
    elsewhere, so this is NOT the module API of the reference (forward_precise returns dense maps): it exists for
    TwoPassStep(label_point_forward=True) and is never the default.  Losses and parameter gradients are those of the dense
    evaluation (the loss reads nothing else; tests/test_gpu_model.py::test_label_point_forward_matches_dense).

    inputs: x (B,H,W,Cp), py, px (B,P) int64, then per head the six parameters of HeadsFused."""

    @staticmethod
    def forward(ctx, x, py, px, *params):
        _require_cuda(x, py, px, params[0])
        x = as_act(x)
        if x.dtype not in _MFMA_DTYPES:
            raise ValueError('HeadsAtPoints: 16-bit activations only')
        n_heads = len(params) // 6
        ws, bs = params[0::6], params[1::6]
        gammas, betas, wps, bps = params[2::6], params[3::6], params[4::6], params[5::6]
        cs = [g.numel() for g in gammas]
        ocs = [w.shape[0] for w in wps]
        nps = [rup8(c) for c in cs]
        Ns = sum(nps)
        B, H, W, Cp = x.shape
        C = ws[0].shape[1]
        assert all(tuple(w.shape) == (c, C, 3, 3) for w, c in zip(ws, cs)) and Cp == rup8(C)
        if n_heads > 4 or max(ocs) > 4 or max(nps) > 512:
            raise ValueError('HeadsAtPoints: at most 4 heads of at most 512 channels and 4 outputs')
        py, px = py.contiguous().long(), px.contiguous().long()
        if py.dim() != 2 or py.shape[0] != B or px.shape != py.shape or py.shape[1] == 0:
            raise ValueError(f'HeadsAtPoints: label points must be (B={B}, P > 0)')
        P = py.shape[1]
        M, K = B * H * W, 9 * Cp
        Mp = -(-(B * P) // 64) * 64
        pw = max(nps)
        dev = x.device
        scratch = torch.empty((M + Mp,), dtype=torch.int32, device=dev)
        pmap, pix = scratch[:M], scratch[M:]
        check(lib.vkas_points_prepare(_p(py), _p(px), B, P, H, W, _p(pmap), _p(pix), Mp, _stream()), 'points_prepare')
        xs = new_act(1, 1, Mp, K, x)
        check(lib.vkas_points_gather_patches(_p(x), act_ld(x), Cp, B, H, W, _p(pix), Mp, _p(xs), _dt(x), _stream()),
              'points_gather_patches')

        def build_hp():
            t = torch.empty((n_heads, 6 * pw + 8), dtype=_FLOAT, device=dev)
            for h in range(n_heads):
                check(lib.vkas_pack_head_params(_p(gammas[h].contiguous()), _p(betas[h].contiguous()),
                                                _p(wps[h].contiguous()), _p(bps[h].contiguous()), cs[h], ocs[h], pw,
                                                _p(t[h]), _stream()), 'pack_head_params')
            return t

        hp = _cached_pack_multi(list(gammas) + list(betas) + list(wps) + list(bps), ('head_params', pw), build_hp)
        b_cat = pack_head_bias(bs, cs, nps)
        Wf = pack_head_weights(ws, nps, Cp, 0, x.dtype)
        zs = new_act(1, 1, Mp, Ns, x)
        g1 = _geom(1, 1, Mp, 1, Mp, K, K, 1, 1, 1, 0)
        conv_gemm(xs, g1, Wf, Ns, zs, _lib.EPI_NONE, bias=b_cat, nk=(sum(cs), C * 9))
        fbuf = torch.empty((n_heads * Mp * 10,), dtype=_FLOAT, device=dev)
        stats_s, proj_s = fbuf[:n_heads * Mp * 2], fbuf[n_heads * Mp * 2:]
        head = _lib.HeadDesc()
        head.n_heads, head.pw = n_heads, pw
        off = 0
        for h in range(n_heads):
            head.n0[h], head.np[h], head.c[h], head.oc[h] = off, nps[h], cs[h], ocs[h]
            off += nps[h]
        head.params, head.stats, head.proj = hp.data_ptr(), stats_s.data_ptr(), proj_s.data_ptr()
        check(lib.vkas_head_tail_fwd(_p(zs), Ns, ctypes.byref(head), Mp, _dt(x), _stream()), 'head_tail_fwd')
        proj = torch.zeros((n_heads, B, H, W, 8), dtype=_FLOAT, device=dev)
        for h in range(n_heads):
            check(lib.vkas_points_scatter_vec8(ctypes.c_void_p(proj_s.data_ptr() + h * Mp * 32), _p(pix), Mp, _p(proj[h]),
                                               _stream()), 'points_scatter_vec8')
        ctx.save_for_backward(xs, zs, stats_s, hp, scratch, *ws, *bs)
        ctx.meta = (cs, ocs, nps, pw, C, (B, H, W, Cp), P, Mp, x.dtype)
        return tuple(proj[h] for h in range(n_heads))

    @staticmethod
    def backward(ctx, *dprojs):
        cs, ocs, nps, pw, C, (B, H, W, Cp), P, Mp, dtype = ctx.meta
        n_heads = len(cs)
        saved = ctx.saved_tensors
        xs, zs, stats_s, hp, scratch = saved[:5]
        ws, bs = saved[5:5 + n_heads], saved[5 + n_heads:5 + 2 * n_heads]
        M, K = B * H * W, 9 * Cp
        Ns = sum(nps)
        dev = xs.device
        pmap, pix = scratch[:M], scratch[M:]
        PS = 6 * pw + 8
        offs = [sum(nps[:h]) for h in range(n_heads + 1)]
        dproj_s = torch.empty((n_heads, Mp, 8), dtype=_FLOAT, device=dev)
        for h in range(n_heads):
            if dprojs[h] is None:
                dproj_s[h].zero_()
            else:
                check(lib.vkas_points_gather_vec8(_p(dprojs[h].contiguous().float()), _p(pix), Mp, _p(dproj_s[h]), _stream()),
                      'points_gather_vec8')
        dparams = torch.empty((n_heads, PS), dtype=_FLOAT, device=dev)
        head = _lib.HeadDesc()
        head.n_heads, head.pw = n_heads, pw
        ptrs = (ctypes.c_void_p * 4)()
        for h in range(n_heads):
            head.n0[h], head.np[h], head.c[h], head.oc[h] = offs[h], nps[h], cs[h], ocs[h]
            ptrs[h] = dproj_s[h].data_ptr()
        head.params, head.stats, head.proj = hp.data_ptr(), stats_s.data_ptr(), None
        nbytes = lib.vkas_head_tail_bwd_ws_bytes(Mp, pw)
        ws_buf = _ws(nbytes, dev)
        dzs = new_act(1, 1, Mp, Ns, xs)
        check(lib.vkas_head_tail_bwd(_p(zs), Ns, ctypes.byref(head), ptrs, _p(dzs), Ns, _p(dparams), _p(ws_buf), nbytes, Mp,
                                     _dtc(dtype), _stream()), 'head_tail_bwd')
        g1 = _geom(1, 1, Mp, 1, Mp, K, K, 1, 1, 1, 0)
        gwp, gbp = conv_wgrad(xs, g1, dzs, Ns, nk=(sum(cs), C * 9), with_bias=True,
                              step_scratch=all(grad_sink(p) is not None for p in list(ws) + list(bs)))
        dx = None
        if ctx.needs_input_grad[0]:
            Wf = pack_head_weights(ws, nps, Cp, 0, dtype).view(1, 1, Ns, K)
            dzt = dzs.view(Mp, Ns).t().contiguous().view(1, 1, Ns, Mp)
            g3 = _geom(1, 1, Ns, 1, Ns, K, K, 1, 1, 1, 0)
            D = conv_wgrad(Wf, g3, dzt, Mp, nk=(B * P, C * 9), step_scratch=True, ordered=True)
            dx = torch.zeros((B, H, W, Cp), dtype=dtype, device=dev)
            check(lib.vkas_points_scatter3x3(_p(D), _p(pix), _p(pmap), Mp, B, H, W, Cp, _p(dx), Cp, _dtc(dtype), _stream()),
                  'points_scatter3x3')
        grads = []
        for h in range(n_heads):
            gslice = gwp[offs[h] * K:offs[h + 1] * K]
            sw = grad_sink(ws[h])
            if sw is not None:
                unpack_wgrad(gslice, (cs[h], C, 3, 3), nps[h], Cp, into=ws[h].grad)
                sw[0].grad_delivered(sw[1])
                gw = None
            else:
                gw = unpack_wgrad(gslice, (cs[h], C, 3, 3), nps[h], Cp)
            d = dparams[h]
            grads.extend([gw, gbp[offs[h]:offs[h] + cs[h]], d[:cs[h]], d[pw:pw + cs[h]],
                          d[2 * pw:6 * pw].view(4, pw)[:ocs[h], :cs[h]], d[6 * pw:6 * pw + ocs[h]]])
        return (dx, None, None, *grads)


_POINT_SPARSE = os.environ.get('VKAS_POINT_SPARSE_BWD', '1') != '0'


def point_sparse(grad: torch.Tensor, py: torch.Tensor, px: torch.Tensor) -> torch.Tensor:
    """Mark ``grad`` - a dense (B, C, H, W) or (B, H, W, C) gradient - as zero everywhere but at the label points (py, px)
    of its images.  The mark is a Python attribute of this very tensor object, stamped with the tensor's version counter:
    autograd hands the same object to the next backward function when the gradient has a single consumer; anything that
    builds a new tensor (an out-of-place sum with another consumer's gradient, a view, a cast) drops the attribute, and
    anything that writes into this tensor - autograd's input buffer accumulates IN PLACE (``old += new``) when it holds the
    last reference, a tensor hook may call ``g.add_()`` - bumps the version, which voids the mark (``point_mark``).  Only
    functions that keep the zero pattern (per-pixel maps) may pass it on (``pass_point_sparse``)."""
    grad._vkas_points = (py, px, grad._version)
    return grad


def point_mark(t: Optional[torch.Tensor]):
    """(py, px) if ``t`` carries a label-point mark that is still valid (nothing wrote into t since it was marked)."""
    m = getattr(t, '_vkas_points', None) if t is not None else None
    if m is None or t._version != m[2]:
        return None
    return m[0], m[1]


def pass_point_sparse(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    pts = point_mark(src)
    if pts is not None:
        point_sparse(dst, pts[0], pts[1])
    return dst


def _point_sparse_run(dprojs, B, H, W):
    """(h0, h1, py, px) when the heads [h0, h1) - a prefix or a suffix of the heads, so that the others stay one contiguous
    run of z columns - all carry the mark of the SAME label points and the compact path pays (points << pixels)."""
    marks = [point_mark(d) for d in dprojs]
    idx = [h for h, m in enumerate(marks) if m is not None]
    if not idx:
        return None
    h0, h1 = idx[0], idx[-1] + 1
    py, px = marks[h0]
    if idx != list(range(h0, h1)) or (h0 > 0 and h1 < len(dprojs)):
        return None
    if any(m[0] is not py or m[1] is not px for m in (marks[h] for h in idx)):
        return None
    if (py.dim() != 2 or py.shape != px.shape or py.shape[0] != B or py.dtype != torch.int64 or px.dtype != torch.int64
            or not py.is_cuda or not py.is_contiguous() or not px.is_contiguous() or py.shape[1] == 0):
        return None
    if py.numel() * 16 > B * H * W:
        return None
    return h0, h1, py, px


class ConvNextLayer(Function):
    """ConvNextBlockLayer.forward (model/convnext.py:29-59): dw7x7 -> LN -> MLP with the layer-scale / stochastic-depth /
    residual epilogue.  bf16 / fp16, C % 8 == 0, C <= 512: the MLP is ONE kernel per direction (csrc/mlp_chain.hip; the 4C-wide
    activation is written once for backward and never read back between the two matrix products); otherwise
    GEMM(C,4C)+GELU -> GEMM(4C,C)+epilogue.  ``rowscale`` is the per-sample keep mask already divided by the keep
    probability (:41-53) or None."""

    @staticmethod
    def forward(ctx, x, dw_w, dw_b, ln_g, ln_b, w1, b1, w2, b2, block_scale, rowscale, keep: bool = True):
        """keep: torch.is_grad_enabled() at the call site (Function.forward itself always runs with grad mode off);
        False = inference, the backward operands h and z are not written."""
        _require_cuda(x, dw_w)
        x = as_act(x)
        B, H, W, Cp = x.shape
        C = dw_w.shape[0]
        M = B * H * W
        dt, st = _dt(x), _stream()
        # depthwise 7x7
        wdw = pack_dw_weight(dw_w, C, Cp, 0)
        y = new_act(B, H, W, Cp, x)
        dwb = pad_vector(dw_b, Cp)
        dw_fwd_name = 'dwconv7x7_mfma_kernel' if x.dtype in _MFMA_DTYPES else 'dwconv7x7_fwd_kernel'  # rocprofv3 names
        _timed(dw_fwd_name, x, 2.0 * 49 * M * C, M, Cp, 49,
               lambda: check(lib.vkas_dwconv7x7_fwd(_p(x), act_ld(x), _p(wdw), _p(dwb), None, 0, _p(y), Cp, B, H, W, Cp, dt,
                                                    st), 'dwconv7x7_fwd'), 2.0 * M * Cp * x.element_size())
        C4 = w1.shape[0]
        C4p = rup8(C4)
        chain = C4 == 4 * C and mlp_chain_eligible(x, C)
        # LayerNorm: inside the fused MLP kernel where that runs (the rows are normalised on their way into the first matrix
        # product; yn / stats are written for backward only), its own launch otherwise
        fuse_ln = chain and not _NO_CHAIN_LN
        if not fuse_ln:
            yn, stats = layernorm_fwd(y, ln_g.contiguous(), ln_b.contiguous(), C, False)
        # MLP
        out = new_act(B, H, W, Cp, x)
        z = new_act(B, H, W, Cp, x)
        cs = pad_vector(block_scale, Cp)
        rs = None if rowscale is None else rowscale.to(_FLOAT).contiguous()
        if chain and fuse_ln:
            h = new_act(B, H, W, C4p, x) if keep else None
            yn = new_act(B, H, W, Cp, x) if keep else None
            stats = torch.empty((M, 2), dtype=_FLOAT, device=x.device) if keep else None
            if not keep:
                z = None
            g = None
            img = pack_mlp_chain(w1, w2, b1, C, 0, x.dtype)
            es = x.element_size()
            lg, lb = pad_vector(ln_g.contiguous(), Cp), pad_vector(ln_b.contiguous(), Cp)
            _timed('mlp_chain_pair_kernel<fwd>' if 256 < C <= 384 else 'mlp_chain_kernel<fwd>', x, 4.0 * M * C * C4, M, C, C4,
                   lambda: check(lib.vkas_mlp_chain_ln_fwd(_p(y), Cp, _p(lg), _p(lb), _p(yn), Cp, _p(stats), _p(img),
                                                           _p(pad_vector(b2, Cp)), _p(x), act_ld(x), _p(cs), _p(rs), H * W,
                                                           _p(h), C4p, _p(z), Cp, _p(out), Cp, M, C, dt, st),
                                 'mlp_chain_ln_fwd'),
                   float(M) * ((5 * Cp + C4p) if keep else 3 * Cp) * es)
            if not keep:
                return out
        elif chain:
            # one kernel: h = yn W1^T + b1 is written once (for backward), gelu(h) goes from the first matrix product
            # into the second in registers, the layer-scale / stochastic-depth / residual epilogue follows
            h = new_act(B, H, W, C4p, x) if keep else None
            if not keep:
                z = None
            g = None
            img = pack_mlp_chain(w1, w2, b1, C, 0, x.dtype)
            es = x.element_size()
            _timed('mlp_chain_pair_kernel<fwd>' if 256 < C <= 384 else 'mlp_chain_kernel<fwd>', x, 4.0 * M * C * C4, M, C, C4,
                   lambda: check(lib.vkas_mlp_chain_fwd(_p(yn), Cp, _p(img), _p(pad_vector(b2, Cp)),
                                                        _p(x), act_ld(x), _p(cs), _p(rs), H * W, _p(h), C4p, _p(z), Cp,
                                                        _p(out), Cp, M, C, dt, st), 'mlp_chain_fwd'),
                   float(M) * ((4 * Cp + C4p) if keep else 3 * Cp) * es)
            if not keep:
                return out
        else:
            h = new_act(B, H, W, C4p, x) if keep else None  # inference: the pre-activation and z are not written
            if not keep:
                z = None
            g = new_act(B, H, W, C4p, x)
            g1 = _geom(B, H, W, H, W, Cp, Cp, 1, 1, 1, 0)
            conv_gemm(yn, g1, pack_conv_weight(w1, C4p, Cp, 0, x.dtype), C4p, h, _lib.EPI_GELU, bias=pad_vector(b1, C4p),
                      out2=g)
            g2 = _geom(B, H, W, H, W, C4p, C4p, 1, 1, 1, 0)
            conv_gemm(g, g2, pack_conv_weight(w2, Cp, C4p, 0, x.dtype), Cp, out, _lib.EPI_SCALE_RES,
                      bias=pad_vector(b2, Cp), out2=z, aux=x, colscale=cs, rowscale=rs, rows_per_image=H * W)
            if not keep:
                return out
        empty = torch.empty(0, device=x.device)
        ctx.save_for_backward(x, y, stats, yn, h, g if g is not None else empty, z, dw_w, ln_g, ln_b, w1, w2, block_scale,
                              rs if rs is not None else empty, b1, dw_b, b2)
        ctx.chain = chain
        ctx.has_rs = rs is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, stats, yn, h, g, z, dw_w, ln_g, ln_b, w1, w2, block_scale, rs, b1, dw_b, b2 = ctx.saved_tensors
        rs = rs if ctx.has_rs else None
        dout = as_act(dout)
        B, H, W, Cp = x.shape
        C = dw_w.shape[0]
        C4 = w1.shape[0]
        C4p = rup8(C4)
        M = B * H * W
        dt, st, dev = _dt(x), _stream(), x.device
        cs = pad_vector(block_scale, Cp)
        # With flat gradient views on all five small parameters of the layer, the three second-stage column sums of this
        # backward (layer scale / b2, LayerNorm affine, depthwise weight / bias) and their delivery are ONE vkas_finalize_many
        # launch at the end instead of three finalize launches + one accumulate launch: the producers leave their partial rows.
        sinks = small_grad_views((block_scale, b2, ln_g, ln_b, dw_b), Cp)
        # out = x + rs*cs*z  ->  dz, d(block_scale), d(b2)
        dz = new_act(B, H, W, Cp, x)
        nbytes = lib.vkas_scale_res_bwd_ws_bytes(M, Cp)
        ws_sr = _ws(nbytes, dev)
        if sinks is None:
            dsb = torch.empty((2 * Cp,), dtype=_FLOAT, device=dev)
            dscale, db2 = dsb[:Cp], dsb[Cp:]
        else:
            dscale = db2 = None
        check(lib.vkas_scale_res_bwd(_p(dout), act_ld(dout), _p(z), Cp, _p(cs), _p(rs), H * W, _p(dz), Cp, _p(dscale),
                                     _p(db2), _p(ws_sr), nbytes, M, Cp, dt, st), 'scale_res_bwd')
        g2 = _geom(B, H, W, H, W, C4p, C4p, 1, 1, 1, 0)
        g1 = _geom(B, H, W, H, W, Cp, Cp, 1, 1, 1, 0)
        if ctx.chain:
            # dh = (dz W2) * gelu'(h) and dyn = dh W1 in one kernel; gelu(h) for the W2 weight gradient is applied
            # by that GEMM's operand loader
            dh = new_act(B, H, W, C4p, x)
            dyn = new_act(B, H, W, Cp, x)
            imgt = pack_mlp_chain(w1, w2, None, C, 1, x.dtype)
            _timed('mlp_chain_pair_kernel<bwd>' if 256 < C <= 384 else 'mlp_chain_kernel<bwd>', x, 4.0 * M * C * C4, M, C, C4,
                   lambda: check(lib.vkas_mlp_chain_bwd(_p(dz), Cp, _p(imgt), _p(h), C4p, _p(dh), C4p, _p(dyn), Cp, M, C,
                                                        dt, st), 'mlp_chain_bwd'),
                   float(M) * (2 * Cp + 2 * C4p) * x.element_size())
            gw2, _ = conv_param_grads(h, g2, dz, Cp, w2, None, x_gelu=True)
        else:
            # GEMM2: z = g W2^T + b2
            gw2, _ = conv_param_grads(g, g2, dz, Cp, w2, None)
            dh = new_act(B, H, W, C4p, x)
            conv_gemm(dz, g1, pack_conv_weight(w2, Cp, C4p, 1, x.dtype), C4p, dh, _lib.EPI_DGELU, aux=h)
        # GEMM1: h = yn W1^T + b1 (bias gradient fused into the wgrad kernel)
        gw1, db1 = conv_param_grads(yn, g1, dh, C4p, w1, b1)
        if not ctx.chain:
            dyn = new_act(B, H, W, Cp, x)
            conv_gemm(dh, g2, pack_conv_weight(w1, C4p, Cp, 1, x.dtype), Cp, dyn, _lib.EPI_NONE)
        # LayerNorm
        if sinks is None:
            dy, dlg, dlb = layernorm_bwd(y, ln_g.contiguous(), ln_b.contiguous(), stats, dyn, C, False)
        else:
            dy, ws_ln, parts_ln = layernorm_bwd(y, ln_g.contiguous(), ln_b.contiguous(), stats, dyn, C, False, defer=True)
            dlg = dlb = None
        # depthwise: wgrad, then dgrad (+ the residual path) in one kernel
        gdwb = torch.empty((50 * Cp,), dtype=_FLOAT, device=dev)
        gdw, gdb = gdwb[:49 * Cp], gdwb[49 * Cp:]
        nbytes = lib.vkas_dwconv7x7_wgrad_ws_bytes(B, H, W, Cp)
        ws = _ws(nbytes, dev)
        mfma = x.dtype in _MFMA_DTYPES
        _timed('dwconv7x7_wgrad_mfma_kernel' if mfma else 'dwconv7x7_wgrad_kernel', x, 2.0 * 49 * M * C, M, Cp, 49,
               lambda: check(lib.vkas_dwconv7x7_wgrad(_p(x), act_ld(x), _p(dy), Cp, _p(gdw) if sinks is None else None,
                                                      _p(gdb) if sinks is None else None, _p(ws), nbytes, B, H, W,
                                                      Cp, dt, st), 'dwconv7x7_wgrad'), 2.0 * M * Cp * x.element_size())
        if sinks is not None:
            p_sr = lib.vkas_scale_res_bwd_parts(M, Cp)
            p_dw = lib.vkas_dwconv7x7_wgrad_parts(B, H, W, Cp, dt)
            finalize_many([(ws_sr, 0, p_sr, Cp, 2 * Cp, block_scale.grad, True), (ws_sr, Cp, p_sr, Cp, 2 * Cp, b2.grad, True),
                           (ws_ln, 0, parts_ln, Cp, 2 * Cp, ln_g.grad, True), (ws_ln, Cp, parts_ln, Cp, 2 * Cp, ln_b.grad, True),
                           (ws, 0, p_dw, 49 * Cp, 50 * Cp, gdw, False), (ws, 49 * Cp, p_dw, Cp, 50 * Cp, dw_b.grad, True)])
            for sk in sinks:
                sk[0].grad_delivered(sk[1])
        sdw = grad_sink(dw_w)
        if sdw is not None:  # unpack straight onto the flat gradient view
            check(lib.vkas_unpack_dw_wgrad(_p(gdw), _p(dw_w.grad), C, Cp, 1, st), 'unpack_dw_wgrad')
            sdw[0].grad_delivered(sdw[1])
            gdw_ref = None
        else:
            gdw_ref = torch.empty((C, 1, 7, 7), dtype=_FLOAT, device=dev)
            check(lib.vkas_unpack_dw_wgrad(_p(gdw), _p(gdw_ref), C, Cp, 0, st), 'unpack_dw_wgrad')
        dx = None
        if ctx.needs_input_grad[0]:
            wflip = pack_dw_weight(dw_w, C, Cp, 1)
            dx = new_act(B, H, W, Cp, x)
            _timed('dwconv7x7_mfma_kernel' if mfma else 'dwconv7x7_fwd_kernel', x, 2.0 * 49 * M * C, M, Cp, 49,
                   lambda: check(lib.vkas_dwconv7x7_fwd(_p(dy), Cp, _p(wflip), None, _p(dout), act_ld(dout), _p(dx), Cp, B, H,
                                                        W, Cp, dt, st), 'dwconv7x7_dgrad'), 3.0 * M * Cp * x.element_size())
        if sinks is not None:
            return (dx, gdw_ref, None, None, None, gw1, db1, gw2, None, None, None, None)
        gdb_, dlg, dlb, db2_, dsc = deliver_small_grads([(dw_b, gdb[:C]), (ln_g, dlg), (ln_b, dlb), (b2, db2[:C]),
                                                         (block_scale, dscale[:C].view(block_scale.shape))])
        return (dx, gdw_ref, gdb_, dlg, dlb, gw1, db1, gw2, db2_, dsc, None, None)


class Resize(Function):
    """F.interpolate to an explicit size; mode 0 bilinear/align_corners=False (model/upernext.py:79,191-195,237-244),
    mode 1 nearest (model/fpn.py:138-142,197-204)."""

    @staticmethod
    def forward(ctx, x, size, mode: int):
        _require_cuda(x)
        x = as_act(x)
        ctx.in_size = (x.shape[1], x.shape[2])
        ctx.mode = mode
        return resize_fwd(x, tuple(size), mode)

    @staticmethod
    def backward(ctx, dy):
        return resize_bwd(as_act(dy), ctx.in_size, ctx.mode), None, None


class ResizeAdd(Function):
    """dst += F.interpolate(src, dst.shape) in place (model/upernext.py:174-182, model/fpn.py:121-129).  Legal for
    the same reason as in the reference: the producer of dst saved its input, not its output."""

    @staticmethod
    def forward(ctx, dst, src, mode: int):
        _require_cuda(dst, src)
        if not act_ok(dst):
            raise ValueError('ResizeAdd: dst must be a valid NHWC activation')
        src = as_act(src)
        ctx.in_size = (src.shape[1], src.shape[2])
        ctx.mode = mode
        resize_fwd(src, (dst.shape[1], dst.shape[2]), mode, out=dst, accumulate=True)
        ctx.mark_dirty(dst)
        return dst

    @staticmethod
    def backward(ctx, dy):
        dy = as_act(dy)
        return dy, resize_bwd(dy, ctx.in_size, ctx.mode), None


class AdaptiveAvgPool(Function):
    """nn.AdaptiveAvgPool2d(s) (model/upernext.py:62)."""

    @staticmethod
    def forward(ctx, x, s: int):
        _require_cuda(x)
        x = as_act(x)
        B, H, W, Cp = x.shape
        y = new_act(B, s, s, Cp, x)
        check(lib.vkas_adaptive_avgpool_fwd(_p(x), act_ld(x), _p(y), Cp, B, H, W, s, Cp, _dt(x), _stream()),
              'adaptive_avgpool_fwd')
        ctx.cfg = (H, W, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        H, W, s = ctx.cfg
        dy = as_act(dy)
        B, _, _, Cp = dy.shape
        dx = new_act(B, H, W, Cp, dy)
        check(lib.vkas_adaptive_avgpool_bwd(_p(dy), act_ld(dy), _p(dx), Cp, B, H, W, s, Cp, 0, _dt(dy), _stream()),
              'adaptive_avgpool_bwd')
        return dx, None


class AdaptiveAvgPools(Function):
    """[nn.AdaptiveAvgPool2d(s)(x) for s in scales] (the PPM's pooled branches, model/upernext.py:58-66) as one autograd node:
    backward adds the branches' gradients inside the pooling kernels (accumulate flag) instead of leaving len(scales) - 1 adds
    of full-size tensors to the autograd engine."""

    @staticmethod
    def forward(ctx, x, *scales: int):
        _require_cuda(x)
        x = as_act(x)
        B, H, W, Cp = x.shape
        ys = []
        for s in scales:
            y = new_act(B, s, s, Cp, x)
            check(lib.vkas_adaptive_avgpool_fwd(_p(x), act_ld(x), _p(y), Cp, B, H, W, s, Cp, _dt(x), _stream()),
                  'adaptive_avgpool_fwd')
            ys.append(y)
        ctx.cfg = (B, H, W, Cp, tuple(scales), x.dtype, x.device)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        B, H, W, Cp, scales, dtype, dev = ctx.cfg
        dx = torch.empty((B, H, W, Cp), dtype=dtype, device=dev)
        first = True
        for s, dy in zip(scales, dys):
            if dy is None:
                continue
            dy = as_act(dy)
            check(lib.vkas_adaptive_avgpool_bwd(_p(dy), act_ld(dy), _p(dx), Cp, B, H, W, s, Cp, 0 if first else 1, _dt(dy),
                                                _stream()), 'adaptive_avgpool_bwd')
            first = False
        if first:
            dx.zero_()
        return (dx,) + (None,) * len(scales)


class Cat(Function):
    """torch.cat along channels (model/upernext.py:82,197, model/fpn.py:144): each part is written into its channel
    slice of one NHWC buffer; backward hands out slices (views) of the incoming gradient."""

    @staticmethod
    def forward(ctx, *parts):
        _require_cuda(*parts)
        parts = [as_act(p) for p in parts]
        B, H, W, _ = parts[0].shape
        widths = [p.shape[3] for p in parts]
        out = new_act(B, H, W, sum(widths), parts[0])
        off = 0
        for p, w in zip(parts, widths):
            copy_channels(p, out[..., off:off + w])
            off += w
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = as_act(dy)
        outs, off = [], 0
        for w in ctx.widths:
            outs.append(dy[..., off:off + w])
            off += w
        return tuple(outs)


def copy_channel_range(x, c_src: int, out, c_dst: int, C: int, zero_tail: int = 0):
    """out[..., c_dst : c_dst + C] = x[..., c_src : c_src + C] (then ``zero_tail`` zero channels): channel offsets at
    element granularity, for concatenations whose parts are not multiples of 8 channels wide."""
    B, H, W, _ = x.shape
    check(lib.vkas_copy_channel_range(_p(x), act_ld(x), c_src, _p(out), act_ld(out), c_dst, B * H * W, C, zero_tail, _dt(x),
                                      _stream()), 'copy_channel_range')
    return out


class ResizeCat(Function):
    """torch.cat([first] + [F.interpolate(p, first's size) for p in others], channels) (model/upernext.py:184-197): the
    resize kernels write straight into their channel slice of the concatenated buffer, so only ``first`` is copied; backward
    reads each slice of the incoming gradient in place.

    ``widths``: the parts' LOGICAL channel counts, or None when every part is as wide as its activation.  The reference
    only asks for ``out_channels % len(levels) == 0`` (upernext.py:144, fpn.py:75) and its own test builds
    ``FpnNeck(..., out_channels=400)`` = 4 x 100 channels (tests/test_fpn.py:16-28): torch.cat puts the parts side by side
    WITHOUT pad channels, so part i starts at channel 100 i - not a 16-byte boundary.  Such a concatenation takes the
    compact path: every part is resized into a buffer of its own and moved to its element-granular channel offset by
    vkas_copy_channel_range; backward cuts the gradient's slices out into zero-padded buffers the same way."""

    @staticmethod
    def forward(ctx, mode: int, widths, first, *others):
        _require_cuda(first, *others)
        first = as_act(first)
        others = [as_act(p) for p in others]
        B, H, W, w0 = first.shape
        padded = [w0] + [p.shape[3] for p in others]
        widths = padded if widths is None else [int(w) for w in widths]
        if len(widths) != len(padded) or any(rup8(w) != pw for w, pw in zip(widths, padded)):
            raise ValueError(f'ResizeCat: logical widths {widths} do not match the activations\' channels {padded}')
        ctx.cfg = (mode, widths, [(p.shape[1], p.shape[2]) for p in others])
        if all(w % 8 == 0 for w in widths):
            out = new_act(B, H, W, sum(widths), first)
            copy_channels(first, out[..., :w0])
            off = w0
            for p in others:
                resize_fwd(p, (H, W), mode, out=out[..., off:off + p.shape[3]])
                off += p.shape[3]
            return out
        total = sum(widths)
        out = new_act(B, H, W, rup8(total), first)
        off = 0
        for i, (p, w) in enumerate(zip([first] + others, widths)):
            r = p if i == 0 else resize_fwd(p, (H, W), mode)
            last = i == len(widths) - 1
            copy_channel_range(r, 0, out, off, w, (rup8(total) - total) if last else 0)
            off += w
        return out

    @staticmethod
    def backward(ctx, dy):
        mode, widths, in_sizes = ctx.cfg
        dy = as_act(dy)
        if all(w % 8 == 0 for w in widths):
            grads = [dy[..., :widths[0]]]
            off = widths[0]
            for w, size in zip(widths[1:], in_sizes):
                grads.append(resize_bwd(dy[..., off:off + w], size, mode))
                off += w
            return (None, None, *grads)
        B, H, W, _ = dy.shape
        grads, off = [], 0
        for i, w in enumerate(widths):
            g = new_act(B, H, W, rup8(w), dy)
            copy_channel_range(dy, off, g, 0, w, rup8(w) - w)  # pad channels of an activation gradient are zero
            grads.append(g if i == 0 else resize_bwd(g, in_sizes[i - 1], mode))
            off += w
        return (None, None, *grads)


class SplitBatch(Function):
    """x -> (x[:b0], x[b0:]) as views of one NHWC activation; backward writes the two gradients into the halves of one
    buffer (no zero-fill + add, which is what slicing through autograd would do).  Used by the merged pass schedule
    (model.forward_both): the backbone runs once on the rough and the precise batch, each neck reads its half."""

    @staticmethod
    def forward(ctx, x, b0: int):
        _require_cuda(x)
        x = as_act(x)
        ctx.meta = (tuple(x.shape), b0, x.dtype)
        return x[:b0], x[b0:]

    @staticmethod
    def backward(ctx, d0, d1):
        shape, b0, dtype = ctx.meta
        ref = d0 if d0 is not None else d1
        out = torch.empty(shape, dtype=dtype, device=ref.device)
        for part, d in ((out[:b0], d0), (out[b0:], d1)):
            if d is None:
                part.zero_()
            else:
                copy_channels(as_act(d), part)
        return out, None


class ToNchw(Function):
    """(B,H,W,Cp) activation -> (B,C,H,W) fp32 NCHW: the permute back of model/upernext.py:223 / fpn.py:183 for the
    1..4-channel head outputs."""

    @staticmethod
    def forward(ctx, x, C: int):
        _require_cuda(x)
        x = as_act(x)
        B, H, W, Cp = x.shape
        out = torch.empty((B, C, H, W), dtype=_FLOAT, device=x.device)
        check(lib.vkas_nhwc_to_nchw_f32(_p(x), act_ld(x), _p(out), B, H, W, C, _dt(x), _stream()), 'nhwc_to_nchw_f32')
        ctx.cfg = (Cp, x.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        Cp, dtype = ctx.cfg
        mark = g
        g = g.contiguous().float()
        B, C, H, W = g.shape
        out = torch.empty((B, H, W, Cp), dtype=dtype, device=g.device)
        check(lib.vkas_nchw_f32_to_nhwc(_p(g), _p(out), Cp, B, H, W, C, Cp,
                                        _dtc(dtype), _stream()),
              'nchw_f32_to_nhwc')
        return pass_point_sparse(mark, out), None  # a permutation of the pixels' channels


class Softplus(Function):
    """nn.Softplus() (model/adaptive_scaling.py:101,140)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.vkas_softplus_fwd(_p(x), _p(y), x.numel(), _stream()), 'softplus_fwd')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mark = dy
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check(lib.vkas_softplus_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream()), 'softplus_bwd')
        return pass_point_sparse(mark, dx)  # elementwise: zero gradient stays zero


_DEFERRED = []  # (pinned host tensor, event, message): device-side argument checks awaiting their verdict


def _defer_check(min_margin: torch.Tensor, message: str):
    """min_margin: 1-element int64 device tensor that is negative when the check failed."""
    host = torch.empty((1,), dtype=min_margin.dtype, pin_memory=True)
    host.copy_(min_margin, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _DEFERRED.append((host, ev, message))


def check_deferred(wait: bool = False):
    """Raise for any device-side argument check that has completed (all of them with wait=True) and failed."""
    keep = []
    for host, ev, message in _DEFERRED:
        if wait:
            ev.synchronize()
        if ev.query():
            if int(host[0]) < 0:
                _DEFERRED.clear()
                raise IndexError(message)
        else:
            keep.append((host, ev, message))
    _DEFERRED[:] = keep


def _check_loss_maps(who, preds, channels, gt_mask, gt_score, up, left):
    """Shape / device contract of the dense losses (loss_function/adaptive_scaling.py:67-86,213-231): every tensor on
    the GPU, predictions (B, c, H, W), targets (B, CH, CW) with the crop inside the map.  The kernels index with these
    numbers, so a mismatch must raise here instead of reading out of bounds."""
    B, _, H, W = preds[0].shape
    for t, c in zip(preds, channels):
        if tuple(t.shape) != (B, c, H, W):
            raise ValueError(f'{who}: prediction must be ({B}, {c}, {H}, {W}), got {tuple(t.shape)}')
    if gt_mask.dim() != 3 or gt_mask.shape[0] != B or gt_score.shape != gt_mask.shape:
        raise ValueError(f'{who}: targets must be (B={B}, CH, CW) and equal, got {tuple(gt_mask.shape)} / '
                         f'{tuple(gt_score.shape)}')
    _, CH, CW = gt_mask.shape
    if up < 0 or left < 0 or up + CH > H or left + CW > W:
        raise ValueError(f'{who}: core box (up={up}, left={left}, {CH}x{CW}) does not fit the {H}x{W} map')


class RoughLoss(Function):
    """AdaptiveScalingRoughLossFunction.__call__ (loss_function/adaptive_scaling.py:53-131), default-active terms."""

    @staticmethod
    def forward(ctx, mask_feat, height_feat, gt_mask, gt_score, up, left, cfg):
        _require_cuda(mask_feat, height_feat, gt_mask, gt_score)
        mask_feat, height_feat = mask_feat.contiguous().float(), height_feat.contiguous().float()
        gt_mask, gt_score = gt_mask.contiguous().float(), gt_score.contiguous().float()
        _check_loss_maps('RoughLoss', (mask_feat, height_feat), (1, 1), gt_mask, gt_score, up, left)
        B, _, H, W = mask_feat.shape
        _, CH, CW = gt_mask.shape
        sums = torch.empty((8,), dtype=torch.float64, device=mask_feat.device)
        loss = torch.empty((), dtype=_FLOAT, device=mask_feat.device)
        check(lib.vkas_rough_loss_fwd(_p(mask_feat), _p(height_feat), _p(gt_mask), _p(gt_score), B, H, W, up, left, CH,
                                      CW, ctypes.byref(cfg), _p(sums), _p(loss), _stream()), 'rough_loss_fwd')
        ctx.save_for_backward(mask_feat, height_feat, gt_mask, gt_score, sums)
        ctx.cfg = (up, left, cfg)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        mask_feat, height_feat, gt_mask, gt_score, sums = ctx.saved_tensors
        up, left, cfg = ctx.cfg
        B, _, H, W = mask_feat.shape
        _, CH, CW = gt_mask.shape
        dm, dh = torch.empty_like(mask_feat), torch.empty_like(height_feat)
        dloss = dloss.contiguous().float()
        check(lib.vkas_rough_loss_bwd(_p(mask_feat), _p(height_feat), _p(gt_mask), _p(gt_score), B, H, W, up, left, CH,
                                      CW, ctypes.byref(cfg), _p(sums), _p(dloss), _p(dm), _p(dh), _stream()),
              'rough_loss_bwd')
        return dm, dh, None, None, None, None, None


class PreciseLoss(Function):
    """AdaptiveScalingPreciseLossFunction.__call__ (loss_function/adaptive_scaling.py:181-346), default-active terms."""

    @staticmethod
    def forward(ctx, prob, offset, angle, dist, gt_score, gt_mask, py, px, gt_off, gt_ang, gt_dist, up, left, cfg):
        _require_cuda(prob, offset, angle, dist, gt_score, gt_mask, py, px, gt_off, gt_ang, gt_dist)
        f = lambda t: t.contiguous().float()
        prob, offset, angle, dist = f(prob), f(offset), f(angle), f(dist)
        gt_score, gt_mask, gt_off, gt_ang, gt_dist = f(gt_score), f(gt_mask), f(gt_off), f(gt_ang), f(gt_dist)
        py, px = py.contiguous().long(), px.contiguous().long()
        _check_loss_maps('PreciseLoss', (prob, offset, angle, dist), (1, 2, 4, 4), gt_mask, gt_score, up, left)
        B, _, H, W = prob.shape
        _, CH, CW = gt_mask.shape
        if py.dim() != 2 or py.shape[0] != B or px.shape != py.shape:
            raise ValueError(f'PreciseLoss: label points must be (B={B}, P), got {tuple(py.shape)} / {tuple(px.shape)}')
        P = py.shape[1]
        for name, t, last in (('up_left_offsets', gt_off, 2), ('corner_angles', gt_ang, 4), ('corner_distances', gt_dist, 3)):
            if tuple(t.shape) != (B, P, last):
                raise ValueError(f'PreciseLoss: {name} must be ({B}, {P}, {last}), got {tuple(t.shape)}')
        if P > 0:
            # The reference's advanced indexing raises on out-of-range label points (and wraps negative ones); a corrupt
            # batch must not train silently on the clamped pixels the kernel falls back to.  The range test runs on the
            # device and its verdict travels to pinned host memory asynchronously: reading it here would stall the
            # launch pipeline once per step, so it is examined at the NEXT loss call / check_deferred() (one step late).
            check_deferred()
            lim = torch.empty((1,), dtype=torch.int64, device=py.device)
            check(lib.vkas_points_margin(_p(py), _p(px), B * P, H, W, _p(lim), _stream()), 'points_margin')
            _defer_check(lim, f'PreciseLoss: label points outside the {H}x{W} map')
        sums = torch.empty((8,), dtype=torch.float64, device=prob.device)
        loss = torch.empty((), dtype=_FLOAT, device=prob.device)
        check(lib.vkas_precise_loss_fwd(_p(prob), _p(offset), _p(angle), _p(dist), _p(gt_score), _p(gt_mask), _p(py),
                                        _p(px), _p(gt_off), _p(gt_ang), _p(gt_dist), B, H, W, up, left, CH, CW, P,
                                        ctypes.byref(cfg), _p(sums), _p(loss), _stream()), 'precise_loss_fwd')
        ctx.save_for_backward(prob, offset, angle, dist, gt_score, gt_mask, py, px, gt_off, gt_ang, gt_dist, sums)
        ctx.cfg = (up, left, cfg)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        prob, offset, angle, dist, gt_score, gt_mask, py, px, gt_off, gt_ang, gt_dist, sums = ctx.saved_tensors
        up, left, cfg = ctx.cfg
        B, _, H, W = prob.shape
        _, CH, CW = gt_mask.shape
        P = py.shape[1]
        dp, do, da, dd = (torch.empty_like(t) for t in (prob, offset, angle, dist))
        dloss = dloss.contiguous().float()
        check(lib.vkas_precise_loss_bwd(_p(prob), _p(offset), _p(angle), _p(dist), _p(gt_score), _p(gt_mask), _p(py),
                                        _p(px), _p(gt_off), _p(gt_ang), _p(gt_dist), B, H, W, up, left, CH, CW, P,
                                        ctypes.byref(cfg), _p(sums), _p(dloss), _p(dp), _p(do), _p(da), _p(dd),
                                        _stream()), 'precise_loss_bwd')
        # offset / angle / distance are read at the label points only (adaptive_scaling.py:235-262): their gradients are
        # zero elsewhere, which the heads' backward exploits
        for g in (do, da, dd):
            point_sparse(g, py, px)
        return (dp, do, da, dd) + (None,) * 10


class ElementwiseLoss(Function):
    """The mean-/masked-mean-type primitive losses and dice (loss_function/{focal_with_logits,dice,l1,l2}.py) as one
    reduction kernel + finalize; gradient for ``pred`` only (targets and masks are data)."""

    @staticmethod
    def forward(ctx, pred, gt, mask, kind: int, p0: float, p1: float, eps: float):
        _require_cuda(pred, gt, mask)
        if gt.shape != pred.shape or (mask is not None and mask.shape != pred.shape):
            raise ValueError(f'loss: pred {tuple(pred.shape)}, gt {tuple(gt.shape)}'
                             + (f', mask {tuple(mask.shape)}' if mask is not None else '') + ' must have equal shapes')
        shape = pred.shape
        pred, gt = pred.contiguous().float(), gt.contiguous().float()
        mask = None if mask is None else mask.contiguous().float()
        sums = torch.empty((4,), dtype=torch.float64, device=pred.device)
        loss = torch.empty((), dtype=_FLOAT, device=pred.device)
        check(lib.vkas_elementwise_loss_fwd(kind, _p(pred), _p(gt), _p(mask), pred.numel(), p0, p1, eps, _p(sums), _p(loss),
                                            _stream()), 'elementwise_loss_fwd')
        ctx.save_for_backward(pred, gt, mask if mask is not None else torch.empty(0, device=pred.device), sums)
        ctx.cfg = (kind, p0, p1, eps, mask is not None, shape)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        pred, gt, mask, sums = ctx.saved_tensors
        kind, p0, p1, eps, has_mask, shape = ctx.cfg
        dpred = torch.empty_like(pred)
        dloss = dloss.contiguous().float()
        check(lib.vkas_elementwise_loss_bwd(kind, _p(pred), _p(gt), _p(mask) if has_mask else None, pred.numel(), p0, p1, eps,
                                            _p(sums), _p(dloss), _p(dpred), _stream()), 'elementwise_loss_bwd')
        return dpred.view(shape), None, None, None, None, None, None


class CrossEntropy(Function):
    """F.cross_entropy on (rows, classes) logits with class-probability targets of the same shape or int64 class
    indices (loss_function/cross_entropy_with_logits.py:16-19); mean over rows."""

    @staticmethod
    def forward(ctx, logits, target):
        _require_cuda(logits, target)
        if logits.dim() != 2:
            raise ValueError(f'cross entropy: logits must be (rows, classes), got {tuple(logits.shape)}')
        rows, classes = logits.shape
        hard = not target.is_floating_point()
        if hard:
            if tuple(target.shape) != (rows,):
                raise ValueError(f'cross entropy: class-index target must be ({rows},), got {tuple(target.shape)}')
            target = target.contiguous().long()
            if rows > 0:
                lo, hi = (int(v) for v in torch.stack([target.min(), target.max()]).tolist())
                if lo < 0 or hi >= classes:
                    raise IndexError(f'cross entropy: target class outside [0, {classes})')
        else:
            if target.shape != logits.shape:
                raise ValueError(f'cross entropy: probability target must be {tuple(logits.shape)}, got {tuple(target.shape)}')
            target = target.contiguous().float()
        logits = logits.contiguous().float()
        sums = torch.empty((4,), dtype=torch.float64, device=logits.device)
        loss = torch.empty((), dtype=_FLOAT, device=logits.device)
        check(lib.vkas_cross_entropy_fwd(_p(logits), _p(target), int(hard), rows, classes, _p(sums), _p(loss), _stream()),
              'cross_entropy_fwd')
        ctx.save_for_backward(logits, target)
        ctx.hard = hard
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, target = ctx.saved_tensors
        rows, classes = logits.shape
        d = torch.empty_like(logits)
        dloss = dloss.contiguous().float()
        check(lib.vkas_cross_entropy_bwd(_p(logits), _p(target), int(ctx.hard), rows, classes, _p(dloss), _p(d), _stream()),
              'cross_entropy_bwd')
        return d, None
