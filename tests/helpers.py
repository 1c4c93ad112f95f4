"""Shared test helpers (CPU-safe)."""
import os

import numpy as np
import torch

from tests.golden import recipe
from vkit_ocr_model_adaptive_scaling_amd.utils import portable_rng as prng

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)


def seeded_state_dict(shapes, seed, std, dtype=torch.float32, requires_grad=False, block_scale=1.0, device='cpu'):
    vals = prng.fill_state_dict(shapes, seed, std=std, block_scale=block_scale)
    sd = {}
    for k, v in vals.items():
        t = torch.from_numpy(v).to(dtype).to(device)
        if requires_grad:
            t.requires_grad_(True)
        sd[k] = t
    return sd


def rel_err(a, b) -> float:
    """Norm-wise relative error ||a - b|| / ||b|| in fp64."""
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    d = float((a - b).norm())
    n = float(b.norm())
    return d / n if n > 0 else d


def check_grad_summary(named_grads, g, tol, prefix='', min_checked=1, skip_missing=False, atol_frac=1e-6):
    """named_grads: name -> tensor with .grad (or name -> grad tensor).  Compares norm / sum / strided samples with
    the stored reference summaries.  Gradients whose reference norm is ~0 relative to the largest one are compared
    absolutely."""
    names = [k[len(prefix) + 6:] for k in g.files if k.startswith(prefix + 'gnorm/')]
    assert len(names) >= min_checked
    gmax = max(float(g[f'{prefix}gnorm/{n}']) for n in names)
    checked = 0
    for n in names:
        t = named_grads.get(n)
        if t is None:
            assert skip_missing, f'missing grad for {n}'
            continue
        gr = t.grad if (isinstance(t, torch.Tensor) and t.requires_grad and t.grad is not None) else t
        assert gr is not None, n
        gr = gr.detach().double().cpu().reshape(-1)
        ref_norm = float(g[f'{prefix}gnorm/{n}'])
        samp = gr[recipe.sample_indices(gr.numel())].numpy()
        ref_samp = g[f'{prefix}gsamp/{n}']
        floor = atol_frac * gmax
        assert abs(float(gr.norm()) - ref_norm) <= tol * ref_norm + floor, (n, float(gr.norm()), ref_norm)
        scale = ref_norm / max(1.0, np.sqrt(gr.numel()))
        assert np.max(np.abs(samp - ref_samp)) <= tol * max(np.max(np.abs(ref_samp)), scale) * 4 + floor, n
        checked += 1
    return checked
