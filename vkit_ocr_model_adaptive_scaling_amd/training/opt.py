"""Helpers of ``vkit_open_model.training`` (training/opt.py:18-57) the train loop imports."""
import random
from typing import Any, Dict

import numpy as np
import torch


def batch_to_device(batch: Dict[str, Any], device: torch.device):
    """Tensors move (asynchronously from pinned memory), everything else - shapes, boxes, rng states - stays."""
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


def device_is_cuda(device: torch.device) -> bool:
    return device.type == 'cuda'


def enable_cudnn_benchmark(device: torch.device):
    """The reference lets cuDNN autotune its convolutions; no library convolution runs on this path (every kernel and tile
    choice is fixed in csrc/), so there is nothing to switch on.  Kept for call-site compatibility."""


def enable_cudnn_deterministic(device: torch.device):
    """Forward is bit-reproducible by construction; weight gradients that split the pixel axis add with fp32 atomics
    (DESIGN.md section 3).  Kept for call-site compatibility."""


def setup_seeds(random_seed: int = 13370, numpy_seed: int = 1337, torch_seed: int = 133):
    random.seed(random_seed)
    np.random.seed(numpy_seed)
    torch.manual_seed(torch_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(torch_seed)


def calculate_iterable_dataset_num_samples(batch_size: int, num_batches: int) -> int:
    return batch_size * num_batches
