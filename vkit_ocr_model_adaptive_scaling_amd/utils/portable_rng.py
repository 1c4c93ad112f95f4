"""Counter-based, machine-independent pseudo random numbers.

Golden fixtures are generated in the build container from the imported reference and
checked on a different machine (the MI355X box), so weights and inputs must be
regenerated bit-for-bit from a seed instead of being shipped.  Everything here uses only
64-bit integer mixing (splitmix64 finaliser) and exact fp64 adds / multiplies, so numpy
on any IEEE machine produces identical bits.  No transcendental function is called.

The "normal-like" stream is a centred Irwin-Hall sum of four uniforms (variance 1/3,
rescaled to unit variance, support [-3.46, 3.46]), which is close enough to the
reference's trunc_normal_(std=0.02) initialisation (convnext.py:169-173) for synthetic
benchmarks and parity inputs.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _C1
    z = (z ^ (z >> np.uint64(27))) * _C2
    return z ^ (z >> np.uint64(31))


def key_from_name(name: str) -> int:
    """Stable 32-bit stream id for a parameter / tensor name."""
    return zlib.crc32(name.encode('utf-8')) & 0xFFFFFFFF


def raw_u64(seed: int, stream: int, n: int, lane: int = 0) -> np.ndarray:
    with np.errstate(over='ignore'):
        base = _mix(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * _GOLD + np.uint64(stream) * _C2 + np.uint64(lane) * _C1)
        idx = np.arange(n, dtype=np.uint64)
        return _mix((idx + np.uint64(1)) * _GOLD + base)


def uniform(seed: int, stream: int, n: int, lane: int = 0) -> np.ndarray:
    """fp64 uniform in [0, 1) with 53 random bits."""
    u = raw_u64(seed, stream, n, lane) >> np.uint64(11)
    return u.astype(np.float64) * (1.0 / 9007199254740992.0)


def normal_like(seed: int, stream: int, n: int) -> np.ndarray:
    """Unit-variance, zero-mean, bounded (|x| < 3.47) fp64 samples."""
    s = uniform(seed, stream, n, 0)
    for lane in (1, 2, 3):
        s = s + uniform(seed, stream, n, lane)
    return (s - 2.0) * 1.7320508075688772  # sqrt(3): var(sum of 4 U) = 1/3


def integers(seed: int, stream: int, n: int, low: int, high: int) -> np.ndarray:
    """int64 in [low, high)."""
    span = np.uint64(high - low)
    return (raw_u64(seed, stream, n) % span).astype(np.int64) + np.int64(low)


def fill_state_dict(shapes, seed: int, std: float = 0.02, block_scale: float = 1.0):
    """Deterministic parameter values for a name -> shape mapping.

    * conv / linear weights: std * normal_like
    * biases: 0.5 * std * normal_like (non-zero so that bias paths are exercised)
    * LayerNorm weights (1-D ``*.weight``): 1 + 0.1 * normal_like
    * ``block_scale``: block_scale * (1 + 0.25 * normal_like); the reference initialises it to
      1e-6 (convnext.py:38) which hides the whole residual branch from a parity check.
    Returns name -> fp64 numpy array.
    """
    out = {}
    for name, shape in shapes.items():
        n = int(np.prod(shape)) if len(shape) else 1
        z = normal_like(seed, key_from_name(name), n).reshape(shape)
        if name.endswith('block_scale'):
            v = block_scale * (1.0 + 0.25 * z)
        elif name.endswith('.bias'):
            v = 0.5 * std * z
        elif name.endswith('.weight') and len(shape) == 1:
            v = 1.0 + 0.1 * z
        else:
            v = std * z
        out[name] = v
    return out
