#!/usr/bin/env python3
"""Headline benchmark: images/s of the full adaptive-scaling train step (ConvNeXt-T + UPerNext + 6 heads, both
passes, both losses, backward, gradient all-reduce, clip, AdamW) on synthetic 1024x1024 inputs, bf16, B=8 per pass
and GPU (BASELINE.json configs[2]; weak scaling over GPUs = configs[3]).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  ``roofline`` describes the dominant kernel (the bf16 MFMA implicit-GEMM that runs
the head 3x3 convolutions, their dgrads and every 1x1): algorithmic FLOPs per launch / average launch duration,
both measured with HIP events on the launch stream during the timed steps.  ``cpu_baseline`` is the oracle
(oracle/torch_oracle.py, a CPU restatement of the reference pinned to the reference's outputs) timed on the host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0                  # same guide: HBM3E peak (about 6.3 TB/s achievable)


def synthetic_batches(batch, hw, device, seed):
    """raw 0..255 pixels; rough gt mask/score map; precise score map/mask/label points; margin 10."""
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed)
    H, W = hw
    dh, dw = H // 2, W // 2
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import Box
    box = Box(up=10, down=dh - 11, left=10, right=dw - 11)
    ch, cw = dh - 20, dw - 20
    P = 200  # train.py:58
    r = lambda *s: torch.rand(*s, generator=g)
    rough = dict(image=torch.randint(0, 256, (batch, 3, H, W), generator=g).float(),
                 downsampled_mask=(r(batch, ch, cw) > 0.5).float(), downsampled_score_map=r(batch, ch, cw) + 8.75,
                 downsampled_shape=(dh, dw), downsampled_core_box=box)
    precise = dict(image=torch.randint(0, 256, (batch, 3, H, W), generator=g).float(),
                   downsampled_mask=(r(batch, ch, cw) > 0.5).float(), downsampled_score_map=r(batch, ch, cw),
                   downsampled_shape=(dh, dw), downsampled_core_box=box,
                   downsampled_label_point_y=torch.randint(10, dh - 10, (batch, P), generator=g),
                   downsampled_label_point_x=torch.randint(10, dw - 10, (batch, P), generator=g),
                   up_left_offsets=torch.randint(-20, 21, (batch, P, 2), generator=g).float(),
                   corner_angles=torch.softmax(r(batch, P, 4), dim=-1), corner_distances=r(batch, P, 3))
    mv = lambda d: {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    return mv(rough), mv(precise)


def usable_cores() -> int:
    """CPU share of this process: affinity mask, capped by the cgroup quota (the GPU box exposes all host cores but
    grants a 16-core share per GPU; oversubscribing them makes the CPU leg meaningless)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get('VKAS_CPU_THREADS', '16'))))


def cpu_baseline(hw, seed):
    """One reference-semantics step (B=1 rough + B=1 precise) of the oracle on the host cores, fp32."""
    import torch
    from oracle import torch_oracle as O
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    torch.manual_seed(seed)
    cores = usable_cores()
    torch.set_num_threads(cores)
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    rough, precise = synthetic_batches(1, hw, 'cpu', seed)
    box = rough['downsampled_core_box']
    cb = (box.up, box.down, box.left, box.right)
    t0 = time.perf_counter()
    m, h = O.forward_rough(sd, rough['image'], 'upernext')
    (O.rough_loss(m, h, rough['downsampled_mask'], rough['downsampled_score_map'], cb) / 2).backward()
    outs = O.forward_precise(sd, precise['image'], 'upernext')
    (O.precise_loss(*outs, precise['downsampled_score_map'], precise['downsampled_mask'], cb,
                    precise['downsampled_label_point_y'], precise['downsampled_label_point_x'],
                    precise['up_left_offsets'], precise['corner_angles'], precise['corner_distances']) / 2).backward()
    dt = time.perf_counter() - t0
    return {'value': 2.0 / dt, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'1 step of the oracle (B=1 rough + B=1 precise, {hw[0]}x{hw[1]}, fp32, ConvNeXt-T+UPerNext, '
                      f'fwd+loss+bwd both passes, no optimizer): {dt:.1f} s'}


def roofline_from_timer(timer, elapsed, steps, use_pmc=False, mfma_peak=MFMA_BF16_DENSE_PEAK_TFLOPS):
    """``roofline`` object of the JSON line from the per-launch HIP-event records of the timed region (ops.LaunchTimer).
    Dominant kernel = the instantiation with the largest share of the timed region.  Its roof is whichever of the two bounds
    is tighter for the launches it ran: algorithmic flops / dense 16-bit MFMA peak or algorithmic bytes / HBM peak (the
    short-K layer GEMMs are HBM-bound, the 3x3 head convolutions MFMA-bound)."""
    summ = timer.summary()
    if not summ:
        return {'bound': 'mfma', 'kernel': 'none', 'achieved': 0.0, 'peak': mfma_peak, 'unit': 'TFLOP/s', 'frac': 0.0,
                'traffic': None}
    dom = max(summ, key=lambda k: summ[k]['ms'])

    def stats(k):
        fl = sum(r[3] for r in timer.records if r[0] == k)
        nb = sum(r[7] for r in timer.records if r[0] == k)
        ms_ = summ[k]['ms']
        t_mfma, t_hbm = fl / (mfma_peak * 1e12), nb / (HBM_PEAK_GBS * 1e9)
        bound = 'mfma' if t_mfma >= t_hbm else 'hbm'
        ach = fl / (ms_ * 1e-3) / 1e12 if bound == 'mfma' else nb / (ms_ * 1e-3) / 1e9
        peak = mfma_peak if bound == 'mfma' else HBM_PEAK_GBS
        return {'bound': bound, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s' if bound == 'mfma' else 'GB/s',
                'frac': round(ach / peak, 4), 'launches_per_step': summ[k]['launches'] / max(steps, 1),
                'avg_launch_ms': round(ms_ / max(summ[k]['launches'], 1), 4),
                'algorithmic_flops_per_launch': fl / max(summ[k]['launches'], 1),
                'algorithmic_bytes_per_launch': nb / max(summ[k]['launches'], 1),
                'share_of_step_time': round(ms_ / (1000.0 * elapsed), 3)}

    traffic, traffic_source = None, None
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if use_pmc and os.path.exists(pmc_path):  # per-launch HBM bytes from the rocprofv3 --pmc passes of this command
        pmc_all = json.load(open(pmc_path))
        pmc = pmc_all.get(dom)
        if pmc:  # measured by a separate rocprofv3 run of this command, not in this run: say which one
            traffic = pmc['bytes_per_launch']
            traffic_source = pmc_all.get('_meta', {}).get('source', 'profiles/pmc_traffic.json')
    is_gemm = lambda k: k.startswith(('gemm_', 'conv3x3_', 'mlp_chain_'))  # the depthwise / resize launches are timed too (HBM roofs)
    all_gemm_flops = sum(v['flops'] for k, v in summ.items() if is_gemm(k))
    all_gemm_ms = sum(v['ms'] for k, v in summ.items() if is_gemm(k))
    roof = stats(dom)
    return {'bound': roof['bound'], 'kernel': dom, 'achieved': roof['achieved'], 'peak': roof['peak'],
            'unit': roof['unit'], 'frac': roof['frac'], 'traffic': traffic, 'traffic_source': traffic_source,
            **{k: v for k, v in roof.items() if k not in ('bound', 'achieved', 'peak', 'unit', 'frac')},
            'all_gemm_kernels': {'tflops': round(all_gemm_flops / (all_gemm_ms * 1e-3) / 1e12, 2) if all_gemm_ms > 0 else 0.0,
                                 'share_of_step_time': round(all_gemm_ms / (1000.0 * elapsed), 3)},
            'other_kernels': {k: {kk: vv for kk, vv in stats(k).items()
                                  if kk in ('bound', 'achieved', 'unit', 'frac', 'avg_launch_ms', 'share_of_step_time')}
                              for k in summ if k != dom}}


CONFIG5_SHAPES = [(h, w) for h in range(1536, 2049, 128) for w in range(1024, 1537, 128)]  # long edge 1536..2048, x32


def config5_shape_sequence(n, seed):
    """Seeded sequence of (H, W) for BASELINE.json configs[4]: long edge 1536..2048, short edge 1024..1536, multiples of 32
    (inferencing/adaptive_scaling.py:95-107 pads every page to x32); the largest shape 2048 x 1536 always occurs."""
    import random
    rnd = random.Random(seed)
    seq = [rnd.choice(CONFIG5_SHAPES) for _ in range(n)]
    if n > 0 and (2048, 1536) not in seq:
        seq[rnd.randrange(n)] = (2048, 1536)
    return seq


def run_forward_config(args, world, rank, device, dist):
    """BASELINE.json configs[1] (--config 2: ConvNeXt-Tiny backbone forward, 640 x 640, bf16, batch 4) and configs[4]
    (--config 5: ConvNeXt-Base + UPerNext, fp16, no-grad inference of both passes + the device post-processing over a seeded
    SEQUENCE of page shapes, B = 1; N > 1: independent replicas, each rank its own pages).  One JSON line like config 3's."""
    import torch
    from vkit_ocr_model_adaptive_scaling_amd import ops
    from vkit_ocr_model_adaptive_scaling_amd._lib import lib, check
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType, ConvNext, set_compute_dtype)
    import ctypes
    torch.manual_seed(1234)
    g = torch.Generator(device='cpu').manual_seed(1337 + rank)
    if args.config == 2:
        dtype = torch.bfloat16
        model = set_compute_dtype(ConvNext.create_tiny().to(device).eval(), dtype)
        B = 4
        x = torch.randint(0, 256, (B, 3, 640, 640), generator=g).float().to(device)
        shapes = [(640, 640)] * (args.steps + args.warmup)

        def eager_step(i):
            return model.forward_act(x)

        def step(i):  # one captured HIP graph, replayed (the first call runs eagerly): inferencing/graphs.py
            return cache.run('config2', lambda x_: tuple(model.forward_act(x_)), [x], stamp)
        workload = 'ConvNeXt-Tiny backbone forward 640x640, batch 4 per GPU, no grad (BASELINE.json configs[1])'
        dt_name = 'bf16'
    else:
        dtype = torch.float16
        model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.BASE, AdaptiveScalingNeckHeadType.UPERNEXT),
                                compute_dtype=dtype).to(device).eval()
        B = 1
        shapes = config5_shape_sequence(args.steps + args.warmup, 4242 + rank)
        pages = {hw: torch.randint(0, 256, (1, 3, *hw), generator=g).float().to(device) for hw in sorted(set(shapes))}
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())

        def page_pass(x):
            H, W = x.shape[2], x.shape[3]
            mask, height = model.forward_rough(x)
            prob, offset, angle, dist_ = model.forward_precise(x)
            h2, w2 = H // 2, W // 2
            vh = torch.full((1,), h2, dtype=torch.int32, device=device)
            vw = torch.full((1,), w2, dtype=torch.int32, device=device)
            om = torch.empty((1, h2, w2), dtype=torch.uint8, device=device)
            oh = torch.empty((1, h2, w2), dtype=torch.float32, device=device)
            check(lib.vkas_rough_postprocess(ptr(mask), ptr(height), 1, h2, w2, ptr(vh), ptr(vw), 0.5, 3.0, ptr(om), ptr(oh),
                                             ops._stream()), 'rough_postprocess')
            op = torch.empty((1, h2, w2), dtype=torch.float32, device=device)
            oo = torch.empty((1, h2, w2, 2), dtype=torch.float32, device=device)
            oa = torch.empty((1, h2, w2, 4), dtype=torch.float32, device=device)
            od = torch.empty((1, h2, w2, 4), dtype=torch.float32, device=device)
            check(lib.vkas_precise_postprocess(ptr(prob), ptr(offset), ptr(angle), ptr(dist_), 1, h2, w2, ptr(vh), ptr(vw),
                                               ptr(op), ptr(oo), ptr(oa), ptr(od), ops._stream()), 'precise_postprocess')
            return om, oh, op, oo, oa, od

        def eager_step(i):
            return page_pass(pages[shapes[i]])

        def step(i):  # one captured HIP graph per page shape, replayed (the first page of a shape runs eagerly)
            return cache.run('config5', page_pass, [pages[shapes[i]]], stamp)
        workload = ('ConvNeXt-Base + UPerNext inference (forward_rough + forward_precise + device post-processing), fp16, '
                    'B = 1 per GPU, seeded sequence of page shapes with long edge 1536..2048 (BASELINE.json configs[4])')
        dt_name = 'f16'

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from vkit_ocr_model_adaptive_scaling_amd.inferencing import GraphCache, param_stamp
    use_graphs = not args.no_graphs
    cache = GraphCache(enabled=use_graphs)
    stamp = param_stamp(model)  # the parameters do not change during the run
    with torch.no_grad():
        if use_graphs:  # untimed: the eager first call + the capture of every signature of the run
            for i in sorted({shapes[j]: j for j in range(len(shapes))}.values()):
                step(i)
                step(i)
        for i in range(args.warmup):
            step(i)
        sync()
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            out = step(i)
        enqueue = time.perf_counter() - t0
        sync()
        elapsed = time.perf_counter() - t0
        out = [o.clone() for o in out]
        # per-kernel durations (the roofline object): HIP events cannot be recorded inside a graph replay, so the same steps run
        # once more eagerly with the launch timer, outside the timed region
        ops.TIMER = ops.LaunchTimer()
        n_k = min(args.steps, 5)
        torch.cuda.synchronize()
        tk = time.perf_counter()
        for i in range(args.warmup, args.warmup + n_k):
            eager_step(i)
        torch.cuda.synchronize()
        eager_elapsed = time.perf_counter() - tk
    timer, ops.TIMER = ops.TIMER, None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    for o in (out if isinstance(out, (list, tuple)) else [out]):
        if o.is_floating_point() and not bool(torch.isfinite(o.float()).all()):
            raise SystemExit('non-finite output in the timed region')
    if rank != 0:
        return
    print(f'[bench] config {args.config}: {args.steps} steps in {elapsed:.3f} s on {world} GPU(s); host enqueue '
          f'{enqueue:.3f} s', file=sys.stderr, flush=True)
    images = B * world
    timed_shapes = shapes[args.warmup:]
    out = {'metric': 'images/sec forward (%s)' % ('backbone, no grad' if args.config == 2 else 'both inference passes'),
           'value': round(images * args.steps / elapsed, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': round(1000.0 * elapsed / args.steps, 3), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': dt_name, 'data': 'synthetic',
           'config': {'workload': workload, 'global_batch': images, 'images_per_step': images,
                      'parallelism': f'dp{world} (independent replicas)',
                      'collective': {'backend': dist.get_backend() if world > 1 else None, 'world_size': world},
                      'shapes': sorted(set(timed_shapes)) if args.config == 5 else [(640, 640)],
                      'megapixels_per_step': round(sum(h * w for h, w in timed_shapes) / len(timed_shapes) / 1e6 * B, 3),
                      'host_enqueue_ms_per_step': round(1000.0 * enqueue / args.steps, 2),
                      'hip_graph_replay': use_graphs, 'graphs_captured': cache.captures,
                      'eager_ms_per_step': round(1000.0 * eager_elapsed / n_k, 3),
                      'kernel_timing': 'eager pass of %d steps behind the timed region' % n_k},
           'roofline': roofline_from_timer(timer, eager_elapsed, n_k)}
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_forward(args.config)
    print(json.dumps(out), flush=True)


def cpu_baseline_forward(config):
    """The oracle's forward on the host cores for the forward-only configurations, on a bounded sample."""
    import torch
    from oracle import torch_oracle as O
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType, ConvNext)
    torch.manual_seed(7)
    torch.set_num_threads(usable_cores())
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        if config == 2:
            sd = {k: v.detach() for k, v in ConvNext.create_tiny().state_dict().items()}
            x = torch.randint(0, 256, (1, 3, 640, 640), generator=g).float()
            t0 = time.perf_counter()
            O.convnext_forward(sd, x)
            dt = time.perf_counter() - t0
            return {'value': 1.0 / dt, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                    'sample': f'oracle ConvNeXt-Tiny backbone forward, 1 image 640x640, fp32: {dt:.1f} s'}
        model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.BASE, AdaptiveScalingNeckHeadType.UPERNEXT))
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        x = torch.randint(0, 256, (1, 3, 1024, 768), generator=g).float()
        t0 = time.perf_counter()
        O.forward_rough(sd, x, 'upernext')
        O.forward_precise(sd, x, 'upernext')
        dt = time.perf_counter() - t0
        mp_full = sum(h * w for h, w in CONFIG5_SHAPES) / len(CONFIG5_SHAPES) / 1e6
        scaled = dt * mp_full / (1024 * 768 / 1e6)
        return {'value': 1.0 / scaled, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                'sample': f'oracle ConvNeXt-Base+UPerNext forward_rough + forward_precise on ONE 1024x768 page, fp32: {dt:.1f} s; '
                          f'scaled by pixel count to the mean page of the shape set ({mp_full:.2f} MP): {scaled:.1f} s/image'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-per-pass', action='store_true',
                    help='skip the single-pass / other-schedule / dense-backward side measurements (profiling runs: the '
                         'kernel statistics then cover warmup + timed steps only)')
    ap.add_argument('--no-graphs', action='store_true', help='--config 2 / 5: enqueue every launch from Python instead of '
                    'replaying one captured HIP graph per input shape')
    ap.add_argument('--detail', action='store_true', help='per-shape GEMM timing table on stderr')
    ap.add_argument('--config', type=int, default=3, choices=(2, 3, 5),
                    help='BASELINE.json configuration (1-based): 3 = the headline train step (default; 4 = the same with '
                         '--gpus N), 2 = ConvNeXt-Tiny backbone forward 640x640 batch 4 bf16, 5 = ConvNeXt-Base fp16 '
                         'inference over a seeded sequence of page shapes')
    ap.add_argument('--schedule', choices=('merged', 'two-pass'), default='merged',
                    help='merged: one backbone pass over the rough + precise batches and one backward of the summed loss '
                         '(same gradients); two-pass: the reference order, rough fwd/bwd then precise fwd/bwd')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vkit_ocr_model_adaptive_scaling_amd import ops
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import (
        AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg, AdaptiveScalingPreciseLossFunction,
        AdaptiveScalingPreciseLossFunctionConifg)
    from vkit_ocr_model_adaptive_scaling_amd.training import (FlatBuffers, FlatAdamW, BucketedGradReducer, TwoPassStep,
                                                              adaptive_scaling_buckets, cosine_warm_restarts_lr)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1')
    # VKAS_DIST_BACKEND=gloo lets the N>1 control flow (bucket arming, hooks, flush) be rehearsed on a one-GPU box with
    # every rank on device 0; the real runs use nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get('VKAS_DIST_BACKEND', 'nccl')
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    # VKAS_FORCE_REDUCER=1 at N = 1: the bucketed reducer issues its (one-rank) RCCL all-reduces anyway - the N > 1 code
    # path measured on a one-GPU box (the line then carries config.collective.forced = true; never the default)
    force_reducer = world == 1 and os.environ.get('VKAS_FORCE_REDUCER', '0') == '1' and args.config == 3
    if force_reducer:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
    if world > 1 or force_reducer:
        dist.init_process_group(backend, rank=rank, world_size=world)

    if args.config != 3:
        run_forward_config(args, world, rank, device, dist)
        if world > 1:
            dist.destroy_process_group()
        return

    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    torch.manual_seed(1234)  # identical initial weights on every rank
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT),
                            compute_dtype=dtype).to(device).train()
    flat = FlatBuffers(model.named_parameters())
    opt = FlatAdamW(None, lr=8e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=2.5, flat=flat)
    reducer = BucketedGradReducer(flat, adaptive_scaling_buckets(model)) if (world > 1 or force_reducer) else None
    step = TwoPassStep(model, AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg()),
                       AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg()), opt, reducer,
                       merge_backbone=(args.schedule == 'merged'))
    hw = (args.size, args.size)
    rough, precise = synthetic_batches(args.batch, hw, device, 1337 + rank)
    torch.manual_seed(99 + rank)  # stochastic-depth masks differ per rank

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        step(rough, precise, lr=cosine_warm_restarts_lr(it / 1000.0, 8e-4, 8e-6, 10, 10))
        it += 1
    sync()
    ops.TIMER = ops.LaunchTimer()
    coll0 = reducer.collectives_issued if reducer is not None else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step(rough, precise, lr=cosine_warm_restarts_lr(it / 1000.0, 8e-4, 8e-6, 10, 10))
        it += 1
    enqueue = time.perf_counter() - t0  # host time to enqueue the timed steps (no sync inside a step)
    sync()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    coll_per_step = ((reducer.collectives_issued - coll0) / args.steps) if reducer is not None else 0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    rl, pl = float(losses[0]), float(losses[1])
    if not (rl == rl and pl == pl):
        raise SystemExit('non-finite loss in the timed region')

    # also the single-pass rates (forward + loss + backward of one batch of `batch` images, no optimizer),
    # so that either reading of "batch 8" is covered.  Outside the timed region; one rank only.
    per_pass = None
    if world == 1 and not args.no_per_pass:
        def pass_ms(run, n=3):
            run()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                run()
            torch.cuda.synchronize()
            return 1000.0 * (time.perf_counter() - t1) / n
        r_ms = pass_ms(lambda: step._rough_loss(model.forward_rough(rough['image']), rough, 0.5).backward())
        p_ms = pass_ms(lambda: step._precise_loss(model.forward_precise(precise['image']), precise, 0.5).backward())
        opt.zero_grad()
        per_pass = {'rough_only_images_per_s': round(args.batch / (r_ms * 1e-3), 2), 'rough_only_ms': round(r_ms, 2),
                    'precise_only_images_per_s': round(args.batch / (p_ms * 1e-3), 2), 'precise_only_ms': round(p_ms, 2)}
        # SURVEY 8(d) pins the unit of work as the reference's order (rough fwd/bwd, then precise fwd/bwd): report that
        # schedule's step time next to the timed one, whichever of the two was timed
        other = TwoPassStep(model, step.rough_loss_fn, step.precise_loss_fn, opt, reducer,
                            merge_backbone=(args.schedule != 'merged'))
        o_ms = pass_ms(lambda: other(rough, precise, lr=8e-4), n=max(2, min(args.steps, 5)))
        per_pass['two_pass_ms_per_step' if args.schedule == 'merged' else 'merged_ms_per_step'] = round(o_ms, 2)
        # The timed steps run the regression heads' backward on the label-point rows only (their gradient is zero
        # elsewhere: ops.point_sparse, csrc/points.hip - same gradients, fewer products with zero).  The same step with
        # that path switched off (every head's backward dense, as a framework without the loss's sparsity would run it):
        ops._POINT_SPARSE = False
        d_ms = pass_ms(lambda: step(rough, precise, lr=8e-4), n=max(2, min(args.steps, 5)))
        ops._POINT_SPARSE = True
        per_pass['dense_point_backward_ms_per_step'] = round(d_ms, 2)
        # ... and with the opt-in label-point FORWARD of the regression heads on top (ops.HeadsAtPoints: their maps are then
        # valid at the label points only - not the reference's module API, hence never the timed configuration)
        lp = TwoPassStep(model, step.rough_loss_fn, step.precise_loss_fn, opt, reducer,
                         merge_backbone=(args.schedule == 'merged'), label_point_forward=True)
        l_ms = pass_ms(lambda: lp(rough, precise, lr=8e-4), n=max(2, min(args.steps, 5)))
        per_pass['label_point_forward_ms_per_step'] = round(l_ms, 2)

    if rank == 0:
        print(f'[bench] {args.steps} steps in {elapsed:.3f} s on {world} GPU(s); host enqueue {enqueue:.3f} s; '
              f'losses {rl:.4f} {pl:.4f}', file=sys.stderr, flush=True)
        ms = 1000.0 * elapsed / args.steps
        images = 2 * args.batch * world  # one rough + one precise batch per rank per step
        summ = timer.summary()
        if args.detail:
            shapes = {}
            for kind, s_, e_, flops, M, N, K, nb_ in timer.records:
                d_ = shapes.setdefault((kind, M, N, K), [0, 0.0, 0.0, 0.0])
                d_[0] += 1
                d_[1] += s_.elapsed_time(e_)
                d_[2] += flops
                d_[3] += nb_
            # roof ms/step = max(flops / dense bf16 MFMA peak, algorithmic bytes / HBM peak)
            print('[bench] kind M N K launches/step ms/step TFLOP/s TB/s roof_ms/step', file=sys.stderr)
            tot_ms = tot_roof = 0.0
            for key, (n_, ms_, fl_, nb_) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                roof_ = max(fl_ / (MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12), nb_ / 8e12) * 1e3
                tot_ms += ms_
                tot_roof += roof_
                print(f'[bench] {key[0]:22s} {key[1]:8d} {key[2]:5d} {key[3]:5d} {n_ / args.steps:6.1f} '
                      f'{ms_ / args.steps:8.3f} {fl_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0:8.1f} '
                      f'{nb_ / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0:6.2f} {roof_ / args.steps:8.3f}', file=sys.stderr)
            print(f'[bench] all GEMMs: {tot_ms / args.steps:.2f} ms/step, roofline {tot_roof / args.steps:.2f} ms/step',
                  file=sys.stderr)
        default_cfg = (args.size == 1024 and args.batch == 8 and args.dtype == 'bf16')
        roof = roofline_from_timer(timer, elapsed, args.steps, use_pmc=default_cfg)
        out = {'metric': 'images/sec fwd+bwd @1024x1024 bf16 (train step: rough+precise passes, losses, backward, '
                         'clip, AdamW)', 'value': round(images / (elapsed / args.steps), 3), 'unit': 'images/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
               'data': 'synthetic', 'config': {'workload': f'ConvNeXt-T + UPerNext + adaptive-scaling heads fwd+bwd '
                                                           f'{args.size}x{args.size}, batch {args.batch} per pass per GPU '
                                                           f'(BASELINE.json configs[2]{"/[3]" if world > 1 else ""})',
                                               'global_batch': args.batch * world, 'images_per_step': images,
                                               'parallelism': f'dp{world}', 'pass_schedule': args.schedule,
                                               'collective': ({'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                                                               'forced': force_reducer,
                                                               'all_reduces_per_step': coll_per_step}
                                                              if reducer is not None else {'backend': None, 'world_size': 1}),
                                               'label_point_backward': 'compact (B*P rows)' if ops._POINT_SPARSE else 'dense',
                                               'per_pass': per_pass,
                                               'host_enqueue_ms_per_step': round(1000.0 * enqueue / args.steps, 2),
                                               'losses': [round(rl, 5), round(pl, 5)]},
               'roofline': roof}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(hw, 1337)
        print(json.dumps(out), flush=True)
    if world > 1 or force_reducer:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
