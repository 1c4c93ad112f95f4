"""Padding helpers of the inference path (mirror of vkit_open_model/inferencing/opt.py:16-41); host-side numpy."""
import math

import numpy as np


def pad_length_to_make_divisible(length: int, downsampling_factor: int):
    """opt.py:16-18 -> (padded length, padding)."""
    padded = math.ceil(length / downsampling_factor) * downsampling_factor
    return padded, padded - length


def pad_mat_to_make_divisible(mat: np.ndarray, downsampling_factor: int) -> np.ndarray:
    """opt.py:21-41: zero-pad an (H, W, *) array at the bottom / right so that H and W are multiples of the factor;
    the input itself when nothing has to be added."""
    height, width = mat.shape[:2]
    height_p, pad_h = pad_length_to_make_divisible(height, downsampling_factor)
    width_p, pad_w = pad_length_to_make_divisible(width, downsampling_factor)
    if pad_h == 0 and pad_w == 0:
        return mat
    out = np.zeros((height_p, width_p) + tuple(mat.shape[2:]), dtype=mat.dtype)
    out[:height, :width] = mat
    return out
