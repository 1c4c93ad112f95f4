"""ORACLE — test infrastructure only.  ctypes wrapper of oracle/c/vkas_oracle.c (plain C, fp64, NCHW)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'libvkas_oracle.so')


def build():
    subprocess.run(['make', '-C', os.path.join(_HERE, 'c')], check=True, stdout=subprocess.DEVNULL)
    return _LIB


def _lib():
    if not os.path.exists(_LIB):
        build()
    return ctypes.CDLL(_LIB)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def conv2d(x, w, b, stride=1, pad=0, depthwise=False):
    x, w, b = _c(x), _c(w), _c(b)
    B, C, H, W = x.shape
    N, K = w.shape[0], w.shape[-1]
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    y = np.empty((B, N, Ho, Wo))
    _lib().vko_conv2d(_p(x), _p(w), _p(b), _p(y), B, C, H, W, N, K, stride, pad, int(depthwise))
    return y


def layernorm(x, g, b):
    x, g, b = _c(x), _c(g), _c(b)
    y = np.empty_like(x)
    _lib().vko_layernorm(_p(x), _p(g), _p(b), _p(y), *x.shape)
    return y


def _unary(name, x):
    x = _c(x)
    y = np.empty_like(x)
    getattr(_lib(), name)(_p(x), _p(y), ctypes.c_size_t(x.size))
    return y


def gelu(x):
    return _unary('vko_gelu', x)


def softplus(x):
    return _unary('vko_softplus', x)


def _resize(name, x, size):
    x = _c(x)
    B, C, H, W = x.shape
    y = np.empty((B, C, size[0], size[1]))
    getattr(_lib(), name)(_p(x), _p(y), B, C, H, W, size[0], size[1])
    return y


def bilinear(x, size):
    return _resize('vko_bilinear', x, size)


def nearest(x, size):
    return _resize('vko_nearest', x, size)


def adaptive_avgpool(x, s):
    x = _c(x)
    B, C, H, W = x.shape
    y = np.empty((B, C, s, s))
    _lib().vko_adaptive_avgpool(_p(x), _p(y), B, C, H, W, s)
    return y


def convnext_layer(x, sd, mask=None):
    """sd: dict with the reference's block.* keys (convnext.py:29-38)."""
    x = _c(x)
    B, C, H, W = x.shape
    out = np.empty_like(x)
    tmp = np.empty(x.size * 9)
    a = {k: _c(v) for k, v in sd.items()}
    _lib().vko_convnext_layer(_p(x), _p(a['block.0.weight']), _p(a['block.0.bias']), _p(a['block.2.weight']),
                              _p(a['block.2.bias']), _p(a['block.3.weight']), _p(a['block.3.bias']),
                              _p(a['block.5.weight']), _p(a['block.5.bias']), _p(a['block_scale'].reshape(-1)),
                              _p(_c(mask)), _p(out), _p(tmp), B, C, H, W)
    return out


def rough_loss(m, h, gm, gs):
    m, h, gm, gs = _c(m), _c(h), _c(gm), _c(gs)
    f = _lib().vko_rough_loss
    f.restype = ctypes.c_double
    return float(f(_p(m), _p(h), _p(gm), _p(gs), ctypes.c_size_t(m.size)))
