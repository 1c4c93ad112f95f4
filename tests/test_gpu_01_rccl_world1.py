"""RCCL on the one GPU a build box has: ``nccl`` (= RCCL on ROCm) initialised with ONE rank in a fresh
child process, the bucketed reducer told to issue its collectives anyway (``always_reduce=True``), and ``TwoPassStep`` on the
real model (ConvNeXt-T + UPerNext at 256 x 256) in both pass schedules.

What this covers that the ``gloo`` tests (tests/test_reducer_gloo.py, tests/test_gpu_00_ddp_world2.py) cannot: the RCCL
library loads and builds a communicator; ``dist.all_reduce(..., async_op=True)`` on slices of the flat gradient buffer is
issued from a callback INSIDE autograd's backward thread, on RCCL's side stream, ordered after the kernels that produced the
bucket on the compute stream; ``wait()`` joins it before the clip / AdamW kernels; nothing errors on that stream.  A sum over
one rank leaves the gradient as it is, so the assertion is exact: the flat gradient equals the no-reducer run bit for bit
(the weight-gradient kernels' fp32 atomics make two runs differ in the last bits, so the comparison is to the run-to-run
spread of the no-reducer path itself), the launch order is the designed one and no bucket is left to flush().  What it cannot
cover is the wire (xGMI between GPUs): the driver's N = 2/4/8 runs do.

The file sorts right behind test_gpu_00_*: the parent has not touched the GPU when it spawns the child (it only counts
devices, which does not initialise HIP on this image)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        import bench
        from vkit_ocr_model_adaptive_scaling_amd import ops
        from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                               AdaptiveScalingNeckHeadType)
        from vkit_ocr_model_adaptive_scaling_amd.loss_function import (
            AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg,
            AdaptiveScalingPreciseLossFunction, AdaptiveScalingPreciseLossFunctionConifg)
        from vkit_ocr_model_adaptive_scaling_amd.training import (FlatBuffers, FlatAdamW, BucketedGradReducer, TwoPassStep,
                                                                  adaptive_scaling_buckets)
        dev = torch.device('cuda', 0)
        torch.manual_seed(5)
        model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT),
                                compute_dtype=torch.bfloat16).to(dev).eval()  # eval: no random stochastic-depth masks
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith('block_scale'):
                    p.fill_(0.5)
        flat = FlatBuffers(model.named_parameters())
        red = BucketedGradReducer(flat, adaptive_scaling_buckets(model), always_reduce=True)
        assert red.world_size == 1 and red.collective and dist.get_backend() == 'nccl'
        rough, precise = bench.synthetic_batches(2, (256, 256), dev, 500)
        rl = AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg())
        pl = AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg())

        class KeepGrads:  # TwoPassStep's optimizer slot: leave the (reduced) gradient in the flat buffer
            def step(self, lr=None):
                pass

            def zero_grad(self):
                pass

        def run(reducer, merged):
            flat.zero_grad()
            TwoPassStep(model, rl, pl, KeepGrads(), reducer, merge_backbone=merged)(rough, precise)
            torch.cuda.synchronize()
            return flat.flat_grad.clone()

        backbone = ['backbone3', 'backbone2', 'backbone1', 'backbone0']
        left_armed = []
        real_flush = red.flush

        def flush(bucket_names=()):
            left_armed.extend(b.name for b in red.buckets.values() if b.armed)
            return real_flush(bucket_names)
        red.flush = flush
        report = []
        for merged in (False, True):
            a0, a1 = run(None, merged), run(None, merged)     # the no-reducer path twice: its own run-to-run spread
            spread = float((a0.double() - a1.double()).norm() / a0.double().norm())
            red.launch_log.clear()
            left_armed.clear()
            issued = red.collectives_issued
            g = run(red, merged)
            assert red.collectives_issued - issued == 6, red.collectives_issued - issued   # six buckets, six RCCL all-reduces
            assert not red._works and not left_armed, (merged, left_armed)
            if merged:
                assert sorted(red.launch_log[:2]) == ['precise', 'rough'] and red.launch_log[2:] == backbone, red.launch_log
            else:
                assert red.launch_log == ['rough', 'precise'] + backbone, red.launch_log
            err = float((g.double() - a0.double()).norm() / a0.double().norm())
            assert bool(torch.isfinite(g).all()) and float(g.norm()) > 0
            # a one-rank sum is the identity: equal to the no-reducer run up to that path's own atomics-order spread
            assert err <= max(2.0 * spread, 1e-6), (merged, err, spread)
            report.append('%s %.3e %.3e' % ('merged' if merged else 'two_pass', err, spread))
        # ... and a real optimizer step behind the collectives: wait() before the clip + AdamW kernels, three steps
        opt = FlatAdamW(None, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=2.5, flat=flat)
        step = TwoPassStep(model, rl, pl, opt, red, merge_backbone=True)
        p0 = flat.flat_param.clone()
        losses = [tuple(float(v) for v in step(rough, precise, lr=1e-4)) for _ in range(3)]
        torch.cuda.synchronize()
        assert all(v == v for l in losses for v in l) and float((flat.flat_param - p0).norm()) > 0
        ops.check_deferred(wait=True)
        open(os.path.join(out_dir, 'report'), 'w').write('\n'.join(report))
    finally:
        dist.destroy_process_group()


def test_real_model_reducer_world1_rccl(tmp_path):
    import torch.multiprocessing as mp
    assert torch.cuda.device_count() >= 1
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    from tests import parity_log
    for line in open(str(tmp_path / 'report')).read().splitlines():
        name, err, spread = line.split()
        parity_log.record('rccl_world1[%s]' % name, 'flat gradient behind 6 RCCL all-reduces vs the no-reducer run', float(err),
                          max(2.0 * float(spread), 1e-6), 'run-to-run spread of the no-reducer path %s' % spread)
