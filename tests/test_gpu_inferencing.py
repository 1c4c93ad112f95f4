"""Inference path (inferencing/adaptive_scaling.py:92-188,295-396 tensor side) on the MI355X: pad-to-32, the no-grad model
calls, device-side post-processing - against the oracle forward on the host followed by the reference's own numpy / torch
post-processing steps restated inline (sigmoid, threshold, padding forced negative, small heights cleared, softmax)."""
import math

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.golden import recipe
from tests.helpers import rel_err
from tests.test_gpu_model import seed_module

pytestmark = pytest.mark.gpu


def build(dtype):
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    from vkit_ocr_model_adaptive_scaling_amd.inferencing import AdaptiveScalingInferencing, AdaptiveScalingInferencingConfig
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT))
    seed_module(model, 71, 0.05)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    inf = AdaptiveScalingInferencing(AdaptiveScalingInferencingConfig(model_jit=model, compute_dtype=dtype,
                                                                      rough_valid_char_height_min=0.7))
    return inf, sd


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16], ids=['f32', 'f16'])
def test_rough_and_precise_infer_vs_oracle(dtype):
    from vkit_ocr_model_adaptive_scaling_amd.inferencing import pad_mat_to_make_divisible
    inf, sd = build(dtype)
    g = np.random.default_rng(5)
    img = g.integers(0, 256, (100, 150, 3), dtype=np.uint8)   # pads to 128 x 160
    padded = pad_mat_to_make_divisible(img, 32)
    assert padded.shape == (128, 160, 3) and (padded[100:] == 0).all() and (padded[:, 150:] == 0).all()
    x = torch.from_numpy(padded.transpose(2, 0, 1).astype(np.float32))[None]
    with torch.no_grad():
        m_ref, h_ref = O.forward_rough(sd, x, 'upernext')
        p_ref, o_ref, a_ref, d_ref = O.forward_precise(sd, x, 'upernext')
    tol = 2e-4 if dtype == torch.float32 else 1e-2
    # rough: reference steps :139-172
    r = inf.rough_infer(img)
    assert r.resized_shape == (50, 75) and r.padded_image.shape == (128, 160, 3)
    assert r.rough_char_mask.dtype == np.uint8 and r.rough_char_mask.shape == (64, 80)
    prob = torch.sigmoid(m_ref[0, 0]).numpy()
    mask_ref = (prob >= 0.5).astype(np.uint8)
    height_ref = h_ref[0, 0].numpy().copy()
    mask_ref[50:] = 0; height_ref[50:] = 0; mask_ref[:, 75:] = 0; height_ref[:, 75:] = 0
    height_ref[height_ref < 0.7] = 0
    sure = (np.abs(prob - 0.5) > 5 * tol)
    assert (r.rough_char_mask[sure] == mask_ref[sure]).all()
    assert (r.rough_char_mask[50:] == 0).all() and (r.rough_char_mask[:, 75:] == 0).all()
    hs = np.abs(h_ref[0, 0].numpy() - 0.7) > 5 * tol  # away from the height threshold
    assert np.allclose(r.rough_char_height_score_map[hs], height_ref[hs], rtol=tol, atol=tol)
    assert 0 < mask_ref.mean() < 1 and (height_ref > 0).any(), 'test image must exercise both branches'
    # precise: reference steps :343-396
    q = inf.precise_infer(img)
    pr = torch.sigmoid(p_ref[0, 0]).numpy().copy()
    pr[50:] = 0; pr[:, 75:] = 0
    assert np.allclose(q.precise_char_prob_score_map, pr, atol=tol)
    assert rel_err(q.precise_np_char_up_left_corner_offset, o_ref[0].permute(1, 2, 0)) < tol
    assert np.allclose(q.precise_np_char_corner_angle_distribution, torch.softmax(a_ref[0].permute(1, 2, 0), -1).numpy(), atol=tol)
    assert rel_err(q.precise_np_char_corner_distance, d_ref[0].permute(1, 2, 0)) < tol
    assert q.precise_char_mask is None and q.padded_image.shape == (128, 160, 3)


def test_precise_batch_groups_by_padded_shape_and_short_side_rule():
    from vkit_ocr_model_adaptive_scaling_amd.inferencing.adaptive_scaling import rough_resized_shape
    inf, _ = build(torch.float16)
    g = np.random.default_rng(6)
    imgs = [g.integers(0, 256, s, dtype=np.uint8) for s in ((60, 90, 3), (64, 96, 3), (40, 200, 3))]
    res = inf.precise_infer_batch(imgs)
    one = inf.precise_infer(imgs[0])
    assert np.array_equal(res[0].precise_char_prob_score_map, one.precise_char_prob_score_map), 'batching must not change results'
    assert [r.precise_char_prob_score_map.shape for r in res] == [(32, 48), (32, 48), (32, 112)]
    assert (res[0].precise_char_prob_score_map[30:] == 0).all() and (res[0].precise_char_prob_score_map[:, 45:] == 0).all()
    assert rough_resized_shape(1536, 2048, 720) == (720, 960) and rough_resized_shape(700, 3000, 720) == (700, 3000)
    big = g.integers(0, 256, (800, 1000, 3), dtype=np.uint8)
    with pytest.raises(ValueError):
        inf.rough_infer(big)  # needs the area-interpolation shrink: host-side image I/O, passed in as resize_fn
    shrink = lambda mat, h, w: mat[:h, :w]
    r = inf.rough_infer(big, resize_fn=shrink)
    assert r.padded_image.shape == (736, 928, 3) and r.rough_char_mask.shape == (368, 464)


def test_no_grad_forward_writes_no_backward_operands():
    """The no-grad fast path: same outputs as a grad-enabled forward, less memory (h / z / statistics are not written)."""
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT))
    seed_module(model, 72, 0.05)
    model.cuda().eval()
    x = torch.from_numpy(recipe.image(72, (2, 3, 256, 256))).float().cuda()
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
    outs_g = model.forward_precise(x)
    torch.cuda.synchronize(); peak_g = torch.cuda.max_memory_allocated() - base
    outs_g = [o.detach().clone() for o in outs_g]
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
    with torch.no_grad():
        outs_n = model.forward_precise(x)
    torch.cuda.synchronize(); peak_n = torch.cuda.max_memory_allocated() - base
    for a, b in zip(outs_g, outs_n):
        assert torch.equal(a, b)
    assert peak_n < 0.6 * peak_g, (peak_n, peak_g)


def test_inferencing_from_a_torchscript_file_matches_the_eager_module(tmp_path):
    """inferencing/adaptive_scaling.py:85-90 loads a TorchScript file: torch.jit.save(torch.jit.script(model)) of this mirror
    (fp16 storage, BASELINE.json configs[4]) loaded by path gives the eager module's results bit for bit."""
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    from vkit_ocr_model_adaptive_scaling_amd.inferencing import AdaptiveScalingInferencing, AdaptiveScalingInferencingConfig
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT),
                            compute_dtype=torch.float16)
    seed_module(model, 72, 0.05)
    path = str(tmp_path / 'model_jit.pt')
    torch.jit.save(torch.jit.script(model), path)
    img = np.random.default_rng(6).integers(0, 256, (90, 140, 3), dtype=np.uint8)
    eager = AdaptiveScalingInferencing(AdaptiveScalingInferencingConfig(model_jit=model, compute_dtype=torch.float16))
    a_r, a_p = eager.rough_infer(img), eager.precise_infer(img)
    del eager, model
    jit = AdaptiveScalingInferencing(AdaptiveScalingInferencingConfig(model_jit=path))
    assert isinstance(jit.model, torch.jit.ScriptModule) and not jit.model.training
    b_r, b_p = jit.rough_infer(img), jit.precise_infer(img)
    assert np.array_equal(a_r.rough_char_mask, b_r.rough_char_mask)
    assert np.array_equal(a_r.rough_char_height_score_map, b_r.rough_char_height_score_map)
    assert np.array_equal(a_p.precise_char_prob_score_map, b_p.precise_char_prob_score_map)
    assert np.array_equal(a_p.precise_np_char_corner_distance, b_p.precise_np_char_corner_distance)


def test_hip_graph_replay_matches_eager_inferencing():
    """AdaptiveScalingInferencingConfig.use_hip_graphs (default): the second call of a padded shape is captured into a HIP
    graph and replayed from then on; results equal the eager path's bit for bit, for interleaved shapes and for a batch."""
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    from vkit_ocr_model_adaptive_scaling_amd.inferencing import AdaptiveScalingInferencing, AdaptiveScalingInferencingConfig
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT))
    seed_module(model, 73, 0.05)
    cfg = dict(model_jit=model, compute_dtype=torch.float16, rough_valid_char_height_min=0.7)
    eager = AdaptiveScalingInferencing(AdaptiveScalingInferencingConfig(use_hip_graphs=False, **cfg))
    graphed = AdaptiveScalingInferencing(AdaptiveScalingInferencingConfig(**cfg))
    assert graphed.graphs.enabled and not eager.graphs.enabled
    g = np.random.default_rng(9)
    imgs = [g.integers(0, 256, s, dtype=np.uint8) for s in ((100, 150, 3), (60, 200, 3), (100, 150, 3), (120, 150, 3))]
    for rnd in range(3):            # round 0: eager first calls, round 1: captures, round 2: pure replays
        for im in imgs:
            er, gr = eager.rough_infer(im), graphed.rough_infer(im)
            assert np.array_equal(er.rough_char_mask, gr.rough_char_mask)
            assert np.array_equal(er.rough_char_height_score_map, gr.rough_char_height_score_map)
            ep, gp = eager.precise_infer(im), graphed.precise_infer(im)
            for f in ('precise_char_prob_score_map', 'precise_np_char_up_left_corner_offset',
                      'precise_np_char_corner_angle_distribution', 'precise_np_char_corner_distance'):
                assert np.array_equal(getattr(ep, f), getattr(gp, f)), (rnd, f)
    assert graphed.graphs.captures == 4 and graphed.graphs.replays >= 12   # 2 padded shapes x 2 passes
    eb, gb = eager.precise_infer_batch(imgs), graphed.precise_infer_batch(imgs)
    for a, b in zip(eb, gb):
        assert np.array_equal(a.precise_char_prob_score_map, b.precise_char_prob_score_map)


def test_scripted_file_runs_in_the_configs_storage_type(tmp_path):
    """A module scripted in bf16 (a training run's model_jit) loaded for inference with compute_dtype=float16 runs in fp16:
    the recipe string of the scripted module is rewritten (model/scripting.py::with_compute_dtype)."""
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    from vkit_ocr_model_adaptive_scaling_amd.inferencing import AdaptiveScalingInferencing, AdaptiveScalingInferencingConfig
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT),
                            compute_dtype=torch.bfloat16)
    seed_module(model, 74, 0.05)
    path = str(tmp_path / 'model_jit.pt')
    torch.jit.save(torch.jit.script(model), path)
    img = np.random.default_rng(7).integers(0, 256, (90, 140, 3), dtype=np.uint8)
    want = AdaptiveScalingInferencing(AdaptiveScalingInferencingConfig(model_jit=model, compute_dtype=torch.float16)).precise_infer(img)
    got = AdaptiveScalingInferencing(AdaptiveScalingInferencingConfig(model_jit=path, compute_dtype=torch.float16)).precise_infer(img)
    assert np.array_equal(want.precise_np_char_corner_distance, got.precise_np_char_corner_distance)
    bf = AdaptiveScalingInferencing(AdaptiveScalingInferencingConfig(model_jit=path, compute_dtype=torch.bfloat16)).precise_infer(img)
    assert not np.array_equal(bf.precise_np_char_corner_distance, got.precise_np_char_corner_distance)
