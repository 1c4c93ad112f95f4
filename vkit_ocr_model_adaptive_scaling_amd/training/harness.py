"""The epoch loop of ``experiment/adaptive_scaling/train.py`` (:52-88 configs, :340-605 loop) around ``TwoPassStep``.

What is reproduced: the config records (field for field: they are the JSON schema of a run), the order
train epoch -> dev evaluation -> checkpoint decision, the learning-rate rule - the reference calls
``scheduler.step(epoch_idx + (batch_idx - 1) / train_num_batches)`` AFTER ``optimizer.step()``, so batch b of epoch e trains
with the rate of the previous call -, the sliding-window loss logs every 4th batch, the dev means and the rule "save when
the dev loss improves, at a dataset switch, and after the last epoch" with the reference's file names, in the
``RestoreState`` schema (training/checkpoint.py).  What is not: the vkit dataset pipeline (any iterable of collated batches
is accepted: dataset/adaptive_scaling.py), TorchScript, JSON / folder handling (``iolite``)."""
import logging
import math
import os
import statistics
from enum import Enum, unique
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import attrs
import torch

from .checkpoint import save_restore_state, scheduler_state_dict
from .metrics import Metrics
from .opt import batch_to_device
from .optimizer import cosine_warm_restarts_lr

logger = logging.getLogger(__name__)


@attrs.define
class EpochConfig:
    """train.py:53-69"""
    torch_seed: int = 133
    num_epochs: int = 110
    num_page_char_regression_labels: int = 200
    train_num_batches: int = 1000
    train_batch_size: int = 6
    train_rng_seed: int = 13371
    train_num_processes: int = 10
    dev_num_batches: int = 70
    dev_batch_size: int = 22
    dev_rng_seed: int = 13
    dev_num_processes: int = 32
    avg_num_batches: int = 50
    enable_overfit_testing: bool = False
    enable_multitask_gradiant_inspection: bool = False


@attrs.define
class OptimizerConfig:
    """train.py:72-80"""
    adamw_lr: float = 8E-4
    adamw_betas: Tuple[float, float] = (0.9, 0.999)
    adamw_weight_decay: float = 0.01
    cosine_annealing_warm_restarts_t0: int = 10
    cosine_annealing_warm_restarts_tmulti: int = 10
    cosine_annealing_warm_restarts_eta_min: float = 8E-6
    clip_grad_norm_max_norm: Optional[float] = 2.5


@unique
class MetricsTag(Enum):
    """train.py:83-88"""
    TRAIN_ROUGH_LOSS = 'train_rough_loss'
    TRAIN_PRECISE_LOSS = 'train_precise_loss'
    DEV_ROUGH_LOSS = 'dev_rough_loss'
    DEV_PRECISE_LOSS = 'dev_precise_loss'


@attrs.define
class EpochResult:
    epoch_idx: int
    dev_rough_loss: float
    dev_precise_loss: float
    dev_loss: float
    state_dict_path: Optional[str]


def evaluate(step, dev_batches: Iterable[Dict], device, metrics: Metrics, epoch_idx: int, dev_num_batches: int):
    """train.py:491-571: no-grad forward of both passes, loss / 2 each, per-batch lists for the means."""
    model = step.model
    model.eval()
    metrics.reset([MetricsTag.DEV_ROUGH_LOSS, MetricsTag.DEV_PRECISE_LOSS])
    rough_losses: List[float] = []
    precise_losses: List[float] = []
    with torch.no_grad():
        for batch_idx, batch in enumerate(dev_batches, start=1):
            rb = batch_to_device(batch['rough'], device)
            rough = float(step._rough_loss(model.forward_rough(rb['image']), rb, 0.5))
            pb = batch_to_device(batch['precise'], device)
            precise = float(step._precise_loss(model.forward_precise(pb['image']), pb, 0.5))
            ra = metrics.update(MetricsTag.DEV_ROUGH_LOSS, rough)
            pa = metrics.update(MetricsTag.DEV_PRECISE_LOSS, precise)
            if batch_idx % 4 == 0 or batch_idx >= dev_num_batches:
                logger.info(f'E={epoch_idx}, B={batch_idx}/{dev_num_batches}, L_rough={ra:.5f}, L_precise={pa:.5f}, '
                            f'L_sum={ra + pa:.5f}, ')
            rough_losses.append(rough)
            precise_losses.append(precise)
    if not rough_losses:
        raise ValueError('the dev loader produced no batch')
    return (statistics.mean(rough_losses), statistics.mean(precise_losses),
            statistics.mean(r + p for r, p in zip(rough_losses, precise_losses)))


def run_training(step, train_batches_of_epoch: Callable[[int], Iterable[Dict]], dev_batches: Callable[[], Iterable[Dict]],
                 epoch_config: EpochConfig, optimizer_config: OptimizerConfig, output_folder: str, device,
                 start_epoch_idx: int = 0, dataset_switch_epochs: Sequence[int] = (),
                 on_epoch_end: Optional[Callable[[EpochResult], None]] = None) -> List[EpochResult]:
    """``step``: a ``TwoPassStep`` (its optimizer takes the learning rate per call).  ``train_batches_of_epoch(epoch_idx)``
    / ``dev_batches()``: iterables of collated batches (``{'rough': ..., 'precise': ...}``, host tensors)."""
    oc, ec = optimizer_config, epoch_config
    # the optimizer hyper-parameters of the run are those of its config record (train.py:287-298): applied here so that the
    # JSON written next to the checkpoints cannot disagree with what trained
    # ... unless the optimizer was restored from a RestoreState file: the reference constructs AdamW from the config and then
    # calls optimizer.load_state_dict (train.py:287-322), so a resumed run trains with the CHECKPOINT's betas / weight decay and
    # only the learning-rate pair is patched from the config; the clip norm is no optimizer state and always the config's
    opt = step.optimizer
    if not getattr(opt, 'restored_from_state', False):
        opt.betas, opt.weight_decay = tuple(oc.adamw_betas), oc.adamw_weight_decay
    opt.max_grad_norm = oc.clip_grad_norm_max_norm
    world = getattr(step, 'world', 1)
    rank = torch.distributed.get_rank() if (world > 1 and torch.distributed.is_initialized()) else 0

    def rule(epoch: float) -> float:
        return cosine_warm_restarts_lr(epoch, oc.adamw_lr, oc.cosine_annealing_warm_restarts_eta_min,
                                       oc.cosine_annealing_warm_restarts_t0, oc.cosine_annealing_warm_restarts_tmulti)

    os.makedirs(output_folder, exist_ok=True)
    metrics = Metrics(MetricsTag, avg_num_batches=ec.avg_num_batches)
    best_rough = best_precise = best = math.inf
    # the rate in force before the first step of this run: the scheduler's state after its last call (train.py:290-338)
    last_sched_epoch = 0.0 if start_epoch_idx == 0 else start_epoch_idx - 1 + (ec.train_num_batches - 1) / ec.train_num_batches
    lr = rule(last_sched_epoch)
    results: List[EpochResult] = []
    for epoch_idx in range(start_epoch_idx, ec.num_epochs):
        step.model.train()
        pending = []  # device scalars not yet read back: one host sync per log line instead of two per batch
        ra = pa = float('nan')
        for batch_idx, batch in enumerate(train_batches_of_epoch(epoch_idx), start=1):
            rough_loss, precise_loss = step(batch_to_device(batch['rough'], device), batch_to_device(batch['precise'], device),
                                            lr=lr)
            last_sched_epoch = epoch_idx + (batch_idx - 1) / ec.train_num_batches
            lr = rule(last_sched_epoch)
            pending.append((rough_loss, precise_loss))
            if batch_idx % 4 == 0 or batch_idx >= ec.train_num_batches:
                for r, p in pending:
                    ra = metrics.update(MetricsTag.TRAIN_ROUGH_LOSS, float(r))
                    pa = metrics.update(MetricsTag.TRAIN_PRECISE_LOSS, float(p))
                pending.clear()
                logger.info(f'E={epoch_idx}, B={batch_idx}/{ec.train_num_batches}, L_rough={ra:.5f}, L_precise={pa:.5f}, '
                            f'L_sum={ra + pa:.5f}, LR={lr:.6f}')
            if batch_idx >= ec.train_num_batches:
                break
        for r, p in pending:
            metrics.update(MetricsTag.TRAIN_ROUGH_LOSS, float(r))
            metrics.update(MetricsTag.TRAIN_PRECISE_LOSS, float(p))
        logger.info('Evaluating...')
        dev_rough, dev_precise, dev_loss = evaluate(step, dev_batches(), device, metrics, epoch_idx, ec.dev_num_batches)
        if world > 1:  # every rank sees its own dev shard: decide "best" on the mean over ranks, identically everywhere
            t = torch.tensor([dev_rough, dev_precise, dev_loss], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t)
            dev_rough, dev_precise, dev_loss = (t / world).tolist()
        logger.info(f'E={epoch_idx}, dev_rough_loss = {dev_rough}, dev_precise_loss = {dev_precise}, dev_loss = {dev_loss}')
        if dev_rough < best_rough:
            best_rough = dev_rough
            logger.info(f'E={epoch_idx}, FOR NOW THE BEST ROUGH LOSS.')
        if dev_precise < best_precise:
            best_precise = dev_precise
            logger.info(f'E={epoch_idx}, FOR NOW THE BEST PRECISE LOSS.')
        path = None
        if dev_loss < best or epoch_idx + 1 in dataset_switch_epochs or epoch_idx + 1 == ec.num_epochs:
            if dev_loss < best:
                best = dev_loss
                path = os.path.join(output_folder, f'state_dict_{epoch_idx}.pt')
                logger.info(f'E={epoch_idx}, FOR NOW THE BEST, SAVING TO {path}')
            else:
                path = os.path.join(output_folder, f'state_dict_{epoch_idx}_not_best.pt')
            sched = scheduler_state_dict(last_sched_epoch, oc.adamw_lr, oc.cosine_annealing_warm_restarts_eta_min,
                                         oc.cosine_annealing_warm_restarts_t0, oc.cosine_annealing_warm_restarts_tmulti)
            if rank == 0:  # replicas are identical: one writer
                save_restore_state(path, epoch_idx, step.model, step.optimizer, sched)
        result = EpochResult(epoch_idx, dev_rough, dev_precise, dev_loss, path)
        results.append(result)
        if on_epoch_end is not None:
            on_epoch_end(result)
    return results
