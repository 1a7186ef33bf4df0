"""Eval-time generation: host logic of the product module against the oracle restatement, and the
oracle's own properties (CPU only; the device parity test is tests/test_gpu_eval_generation.py)."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("n,bs,world", [(100, 32, 1), (100, 32, 3), (64, 32, 2), (5, 8, 2), (129, 16, 8)])
def test_eval_batch_split_matches_oracle_and_covers_all_images(n, bs, world):
    from oracle import eval_batch_sizes_ref
    from phendiff_amd.eval_generation import eval_batch_sizes
    got = [eval_batch_sizes(n, bs, world, r) for r in range(world)]
    assert got == [eval_batch_sizes_ref(n, bs, world, r) for r in range(world)]
    assert sum(sum(g) for g in got) == n
    lens = [len(g) for g in got]
    assert max(lens) - min(lens) <= 1 and lens == sorted(lens, reverse=True)


def test_best_model_bookkeeping():
    from phendiff_amd.eval_generation import get_initial_best_metric, is_it_best_model
    best = get_initial_best_metric()
    ok, best = is_it_best_model([10.0, 20.0], best)
    assert ok and best == 15.0
    ok, best = is_it_best_model([15.0, 15.0], best)           # strict '<' (utils_misc.py:357)
    assert not ok and best == 15.0
    ok, best = is_it_best_model([14.0], best)
    assert ok and best == 14.0


def test_latents_preview_and_uint8_match_oracle():
    from oracle import latents_preview_ref
    from phendiff_amd.eval_generation import images_to_uint8, latents_preview
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(5, 4, 16, 16, generator=g)
    assert np.array_equal(latents_preview(lat), latents_preview_ref(lat.clone()))
    p = latents_preview(lat)
    assert p.shape == (5, 1, 16, 16) and p.min() == 0 and p.max() == 255
    x = torch.rand(2, 8, 8, 3, generator=g).numpy()
    assert np.array_equal(images_to_uint8(x), (x * 255).round().astype("uint8"))


def test_oracle_generation_shares_one_generator_across_classes_and_restores_weights():
    from oracle import (CondUNet2DRef, DDIMSchedulerRef, EMASwapRef, TINY_CONFIG0_UNET, ConditionalDDIMPipelineRef,
                        eval_generation_ddim_ref)
    from test_oracle_training import SCHED
    torch.manual_seed(0)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in TINY_CONFIG0_UNET.items() if k in keys}).eval()
    pipe = ConditionalDDIMPipelineRef(r, DDIMSchedulerRef(**SCHED))
    params = list(r.parameters())
    shadow = [p.detach() * 0.5 for p in params]
    before = [p.detach().clone() for p in params]
    with EMASwapRef(shadow, params):
        assert all(torch.equal(p, s) for p, s in zip(params, shadow))
        out = eval_generation_ddim_ref(pipe, 2, [3, 2], None, 3, torch.Generator().manual_seed(11), eval_batch_size=3)
    assert all(torch.equal(p, b) for p, b in zip(params, before))
    assert out[0][0] == ["process_0_sample_0.png", "process_0_sample_1.png", "process_0_sample_2.png",
                         "process_0_sample_3.png", "process_0_sample_4.png"]
    assert out[0][1].shape == (5, 32, 32, 3) and out[1][1].shape == (5, 32, 32, 3)
    assert out[0][1].min() >= 0 and out[0][1].max() <= 1
    # class 1 continues the generator stream: regenerating class 1 alone from the same seed gives class 0's noise instead
    with EMASwapRef(shadow, params):
        again = eval_generation_ddim_ref(pipe, 2, [3, 2], None, 3, torch.Generator().manual_seed(11), eval_batch_size=3)
    assert np.array_equal(again[1][1], out[1][1])
    assert not np.allclose(out[0][1], out[1][1])
