"""Host-side logic of the product (no GPU): scheduler tables / grids / coefficients against the oracle and the
golden fixtures (bit-exact where integer), weight packing layout, module tree = diffusers names, config validation,
batch sharding, and the no-CPU-fallback rule."""
import os

import numpy as np
import pytest
import torch

import phendiff_amd as P
from oracle import CondUNet2DRef, DDIMInverseSchedulerRef, DDIMSchedulerRef

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", sorted(P.SCHEDULER_CONFIGS))
def test_scheduler_tables_bit_exact_vs_oracle(name):
    cfg = P.SCHEDULER_CONFIGS[name]
    got, ref = P.DDIMScheduler(**cfg), DDIMSchedulerRef(**cfg)
    assert torch.equal(got.alphas_cumprod, ref.alphas_cumprod)
    assert float(got.final_alpha_cumprod) == float(ref.final_alpha_cumprod)
    igot, iref = P.DDIMInverseScheduler.from_config(got.config), DDIMInverseSchedulerRef.from_config(ref.config)
    assert torch.equal(igot.alphas_cumprod, iref.alphas_cumprod)
    assert float(igot.final_alpha_cumprod) == float(iref.final_alpha_cumprod)
    for S in (1, 4, 50, 100, 999):
        got.set_timesteps(S); ref.set_timesteps(S); igot.set_timesteps(S); iref.set_timesteps(S)
        assert got.timesteps.dtype == torch.int64 and torch.equal(got.timesteps, ref.timesteps)
        assert torch.equal(igot.timesteps, iref.timesteps)
    for variant in ("0.18.2", "0.20+"):
        a = P.DDIMInverseScheduler.from_config(got.config, variant=variant)
        b = DDIMInverseSchedulerRef.from_config(ref.config, variant=variant)
        a.set_timesteps(50); b.set_timesteps(50)
        assert torch.equal(a.timesteps, b.timesteps) and torch.equal(a.alphas_cumprod, b.alphas_cumprod)


def test_scheduler_golden_fixture():
    d = np.load(os.path.join(GOLDEN, "scheduler_3k.npz"))
    s = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    s.set_timesteps(50)
    assert np.array_equal(s.alphas_cumprod.numpy(), d["alphas_cumprod"])
    assert np.array_equal(s.timesteps.numpy(), d["timesteps_50"])
    inv = P.DDIMInverseScheduler.from_config(s.config)
    inv.set_timesteps(50)
    assert np.array_equal(inv.alphas_cumprod.numpy(), d["inv_alphas_cumprod"])
    assert np.array_equal(inv.timesteps.numpy(), d["inv_timesteps_50"])


def test_step_coefficients_match_oracle_arithmetic():
    """The four fp32 coefficients handed to pd_ddim_step reproduce the oracle's update exactly on CPU."""
    cfg = P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]
    got, ref = P.DDIMScheduler(**cfg), DDIMSchedulerRef(**cfg)
    got.set_timesteps(50); ref.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn(2, 3, 8, 8, generator=g), torch.randn(2, 3, 8, 8, generator=g)
    for t in ref.timesteps[::7]:
        sa, sb, sap, dirc, sigma = got.step_coefficients(t)
        x0 = (torch.tensor(sa) * x - torch.tensor(sb) * v).clamp(-1, 1)
        eps = torch.tensor(sa) * v + torch.tensor(sb) * x
        mine = torch.tensor(sap) * x0 + torch.tensor(dirc) * eps
        assert torch.equal(mine, ref.step(v, t, x).prev_sample)
        assert sigma == 0.0


def test_no_cpu_fallback():
    s = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    s.set_timesteps(4)
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(P.PhenDiffHipError):
        s.step(x, s.timesteps[0], x)
    with pytest.raises(P.PhenDiffHipError):
        s.add_noise(x, x, torch.tensor([3]))
    m = P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], sample_size=32))
    with pytest.raises(P.PhenDiffHipError):
        m(x.expand(1, 3, 32, 32) if False else torch.zeros(1, 3, 32, 32), 1, class_labels=torch.tensor([0]))
    import phendiff_amd._lib as L
    import inspect
    for mod in (P.unet, P.schedulers, P.pipeline, P.img2img, L):
        assert "oracle" not in inspect.getsource(mod).replace("the oracle", "").replace("CPU oracle", ""), mod.__name__


def test_pack_conv_weight_layout():
    from phendiff_amd.packing import pack_conv_weight
    g = torch.Generator().manual_seed(0)
    w = torch.randn(40, 64, 3, 3, generator=g)
    p = pack_conv_weight(w, torch.float32, 64)
    assert p.shape == (2, 2, 9, 2, 64, 8)
    for (ct, ch, tap, s, lane, j) in [(0, 0, 0, 0, 0, 0), (1, 1, 8, 1, 63, 7), (0, 1, 4, 0, 37, 3), (1, 0, 2, 1, 5, 6)]:
        r, h = lane & 31, lane >> 5
        co, ci = 32 * ct + r, 32 * ch + 16 * s + 8 * h + j
        want = w[co, ci, tap // 3, tap % 3] if co < 40 else 0.0
        assert float(p[ct, ch, tap, s, lane, j]) == float(want)
    pb = pack_conv_weight(w, torch.bfloat16, 64)
    assert pb.dtype == torch.bfloat16 and torch.equal(pb.float(), p.to(torch.bfloat16).float())


@pytest.mark.parametrize("variant", [dict(center_input_sample=True), dict(resnet_time_scale_shift="scale_shift"),
                                     dict(class_embed_type="timestep"), dict(class_embed_type="identity", num_class_embeds=None)],
                         ids=lambda v: "+".join(sorted(v)))
def test_config_variants_build_the_reference_module_tree(variant):
    """Constructor switches no shipped config sets (cond_unet_2d.py:103,146-153,272-273): same parameters, names and default
    init as the restated reference class; the switches the HIP path does not implement still refuse at construction."""
    cfg = dict(P.UNET_CONFIGS["super_small"], **variant)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    torch.manual_seed(0)
    ref = CondUNet2DRef(**{k: v for k, v in cfg.items() if k in keys})
    torch.manual_seed(0)
    got = P.CustomCondUNet2DModel(**cfg)
    rs, gs = ref.state_dict(), got.state_dict()
    assert list(rs) == list(gs) and all(torch.equal(rs[k], gs[k]) for k in rs)
    if "resnet_time_scale_shift" in variant:
        assert got.down_blocks[0].resnets[0].time_emb_proj.weight.shape[0] == 2 * 64
    for bad in (dict(time_embedding_type="fourier"), dict(act_fn="mish"), dict(mid_block_scale_factor=2)):
        with pytest.raises(NotImplementedError):
            P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], **bad))
    with pytest.raises(ValueError):
        P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], class_embed_type="projection"))


@pytest.mark.parametrize("name", ["super_small", "small_denoiser_config"])
def test_module_tree_matches_diffusers_names(name):
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    torch.manual_seed(0)
    ref = CondUNet2DRef(**{k: v for k, v in P.UNET_CONFIGS[name].items() if k in keys})
    torch.manual_seed(0)
    got = P.CustomCondUNet2DModel(**P.UNET_CONFIGS[name])
    rs, gs = ref.state_dict(), got.state_dict()
    assert list(rs) == list(gs)
    assert all(rs[k].shape == gs[k].shape for k in rs)
    assert all(torch.equal(rs[k], gs[k]) for k in rs)       # same construction order => same default init
    assert got.time_embed_dim == 4 * P.UNET_CONFIGS[name]["block_out_channels"][0]
    assert sum(p.numel() for p in got.parameters()) == {"super_small": 15_725_443, "small_denoiser_config": 62_826_243}[name]
    # surface the reference touches: train.py:215-220 (.attentions on sub-modules), inspect.signature (class_emb)
    import inspect
    assert "class_emb" in inspect.signature(got.forward).parameters
    assert any(hasattr(mod, "attentions") for mod in got.modules())
    assert got.config.sample_size == 128 and got.config.in_channels == 3


def test_config_validation():
    with pytest.raises(ValueError):
        P.CustomCondUNet2DModel(down_block_types=("DownBlock2D",), up_block_types=("UpBlock2D", "UpBlock2D"), block_out_channels=(64,))
    with pytest.raises(ValueError):
        P.CustomCondUNet2DModel(down_block_types=("DownBlock2D",), up_block_types=("UpBlock2D",), block_out_channels=(64, 128))
    with pytest.raises(NotImplementedError):
        P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], time_embedding_type="fourier"))
    with pytest.raises(TypeError):
        P.CustomCondUNet2DModel(bogus=1)
    with pytest.raises(ValueError):
        P.DDIMScheduler(prediction_type="nope")
    with pytest.raises(ValueError):
        P.DDIMInverseScheduler(variant="1.0")


def test_pipeline_check_inputs_like_reference():
    pipe = P.ConditionalDDIMPipeline(P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], sample_size=32)),
                                     P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
    assert isinstance(pipe.scheduler, P.DDIMScheduler) and set(pipe.components) == {"unet", "scheduler"}
    lab = torch.tensor([0, 1])
    pipe.check_inputs(lab, None, 2.5, None, 0.5, torch.zeros(2, 3, 32, 32))
    with pytest.raises(AssertionError):
        pipe.check_inputs(lab, torch.zeros(2, 256))                      # both labels and emb
    with pytest.raises(AssertionError):
        pipe.check_inputs(lab, None, None, None, 0.5, None)             # frac without start image
    with pytest.raises(AssertionError):
        pipe.check_inputs(lab, None, torch.ones(3))                     # w of the wrong batch size
    with pytest.raises(ValueError):
        pipe.check_inputs(lab, None, None, [torch.Generator()])         # generator list of the wrong length
    with pytest.raises(AssertionError):
        pipe.check_inputs(lab, None, None, None, 1.5, torch.zeros(2, 3, 32, 32))


def test_shard_batches_batchsamplershard_semantics():
    assert P.shard_batches(8, 1, 4) == [1, 5]
    assert [P.shard_batches(5, r, 4) for r in range(4)] == [[0, 4], [1, 0], [2, 1], [3, 2]]   # tail wraps to the start
    assert [P.shard_batches(2, r, 4) for r in range(4)] == [[0], [1], [0], [1]]
    assert P.shard_batches(5, 1, 4, even_batches=False) == [1]
    assert P.shard_batches(0, 0, 2) == []
    for n in range(1, 20):
        for g in (1, 2, 3, 8):
            shards = [P.shard_batches(n, r, g) for r in range(g)]
            assert len({len(s) for s in shards}) == 1                     # every rank runs the same number of steps
            assert set(range(n)) <= {b for s in shards for b in s}      # every batch is processed
    assert torch.equal(P.swap_binary_labels(torch.tensor([0, 1, 1])), torch.tensor([1, 0, 0]))
    with pytest.raises(ValueError):
        P.shard_batches(4, 2, 2)


def test_pretrained_folder_roundtrip_and_deprecated_attention_keys(tmp_path):
    """diffusers save_pretrained layout incl. the 0.18 on-disk attention aliases."""
    import json
    from phendiff_amd.checkpoint import remap_deprecated_attention_keys
    torch.manual_seed(1)
    unet = P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], sample_size=32))
    pipe = P.ConditionalDDIMPipeline(unet, P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
    pipe.save_pretrained(str(tmp_path / "pipe"))
    idx = json.load(open(tmp_path / "pipe" / "model_index.json"))
    assert idx["unet"] == ["src.cond_unet_2d.cond_unet_2d", "CustomCondUNet2DModel"]
    back = P.ConditionalDDIMPipeline.from_pretrained(str(tmp_path / "pipe"), compute_dtype="f32")
    assert back.unet.compute_dtype == "f32" and back.unet.config.sample_size == 32
    assert all(torch.equal(a, b) for a, b in zip(unet.state_dict().values(), back.unet.state_dict().values()))
    assert vars(back.scheduler.config) == vars(pipe.scheduler.config)
    assert torch.equal(back.scheduler.alphas_cumprod, pipe.scheduler.alphas_cumprod)
    # a 0.18-style .bin with query/key/value/proj_attn names loads into to_q/to_k/to_v/to_out.0
    sd = unet.state_dict()
    old = {}
    for k, v in sd.items():
        for new, dep in ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn.")):
            k = k.replace(new, dep)
        old[k] = v
    assert any(".query." in k for k in old) and set(remap_deprecated_attention_keys(old)) == set(sd)
    folder = tmp_path / "old_unet"
    unet.save_pretrained(str(folder), safe_serialization=False)
    torch.save(old, folder / "diffusion_pytorch_model.bin")
    legacy = P.CustomCondUNet2DModel.from_pretrained(str(folder))
    assert all(torch.equal(a, b) for a, b in zip(sd.values(), legacy.state_dict().values()))


def test_training_param_order_makes_fused_gradients_contiguous():
    """Flat-buffer order for training: every parameter exactly once; stacked time_emb_proj and adjacent q/k/v so that the
    fused [proj_dim][tdim] / [3C][C] gradients pd_linear_wgrad / pd_conv_wgrad write are single contiguous blocks."""
    import phendiff_amd as P
    m = P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], sample_size=32))
    order = P.training_param_order(m)
    names = [n for n, _ in order]
    assert sorted(names) == sorted(n for n, _ in m.named_parameters()) and len(set(names)) == len(names)
    assert sum(p.numel() for _, p in order) == 15_725_443
    n_res = sum(1 for n in names if n.endswith("time_emb_proj.weight"))
    assert all(n.endswith("time_emb_proj.weight") for n in names[:n_res])
    assert all(n.endswith("time_emb_proj.bias") for n in names[n_res:2 * n_res])
    # module order == the order pd_temb's stacked projection uses (temb_off of each resnet)
    from phendiff_amd.unet import _Resnet
    assert [n[:-len(".time_emb_proj.weight")] for n in names[:n_res]] == [n for n, mod in m.named_modules() if isinstance(mod, _Resnet)]
    i = names.index("down_blocks.2.attentions.0.to_q.weight")
    assert names[i:i + 6] == [f"down_blocks.2.attentions.0.{w}.{s}" for s in ("weight", "bias") for w in ("to_q", "to_k", "to_v")]


def test_trainer_fails_loudly_without_gpu():
    import phendiff_amd as P
    m = P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], sample_size=32))
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    with pytest.raises(P.PhenDiffHipError):
        P.UNetTrainer(m, sched, lr=1e-4)


def test_plan_grad_buckets_cover_buffer_in_ready_order():
    from phendiff_amd.unet_train import plan_grad_buckets
    sizes = [10, 30, 5, 5, 50, 20, 40]
    ready = [99, 90, 80, 70, 40, 20, 5]        # backward finishes the END of the buffer first
    b = plan_grad_buckets(sizes, ready, 40)
    assert sorted((s, e) for s, e, _ in b) == [(0, 40), (40, 100), (100, 160)]       # contiguous, complete, no overlap
    assert [r for _, _, r in b] == sorted(r for _, _, r in b) and b[0] == (100, 160, 20)
    assert dict(((s, e), r) for s, e, r in b)[(0, 40)] == 99                        # a bucket is ready when its LAST gradient is
    assert plan_grad_buckets([7], [3], 1 << 20) == [(0, 7, 3)]


def test_sd_training_param_order_keeps_fused_projections_adjacent():
    """Host logic of the SD trainer (no GPU): the flat-buffer order covers every parameter once, stacks the time_emb_proj
    matrices and keeps attn1 q/k/v and attn2 k/v weights adjacent (the fused projections read them as one matrix)."""
    import torch
    from phendiff_amd.sd_unet import SDUNet2DConditionModel
    from phendiff_amd.sd_unet_train import sd_training_param_order
    from phendiff_amd.unet_train import plan_grad_buckets
    cfg = dict(in_channels=4, out_channels=4, block_out_channels=(64, 128), layers_per_block=1,
               down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
               attention_head_dim=(1, 2), cross_attention_dim=96, norm_num_groups=32)
    m = SDUNet2DConditionModel(compute_dtype="f32", **cfg)
    order = sd_training_param_order(m)
    names = [n for n, _ in order]
    assert sorted(names) == sorted(n for n, _ in m.named_parameters()) and len(set(names)) == len(names)
    pos = {n: i for i, n in enumerate(names)}
    tproj_w = [n for n in names if n.endswith("time_emb_proj.weight")]
    assert [pos[n] for n in tproj_w] == list(range(len(tproj_w)))                     # stacked first, in module order
    for n in names:
        if n.endswith("attn1.to_q.weight"):
            b = n[:-len("to_q.weight")]
            assert pos[b + "to_k.weight"] == pos[n] + 1 and pos[b + "to_v.weight"] == pos[n] + 2
        if n.endswith("attn2.to_k.weight"):
            assert pos[n[:-len("to_k.weight")] + "to_v.weight"] == pos[n] + 1
    # bucket planning over these sizes: contiguous cover, sorted by readiness
    sizes = [p.numel() for _, p in order]
    ready = list(range(len(sizes), 0, -1))
    buckets = plan_grad_buckets(sizes, ready, 1 << 16)
    cover = sorted((s, e) for s, e, _ in buckets)
    assert cover[0][0] == 0 and cover[-1][1] == sum(sizes) and all(a[1] == b[0] for a, b in zip(cover[:-1], cover[1:]))
    assert [r for _, _, r in buckets] == sorted(r for _, _, r in buckets)


def test_vae_image_processor_accepts_every_input_kind_of_the_reference():
    """phendiff_amd's VaeImageProcessor.preprocess is host-side formatting (no kernel): PIL / numpy / tensor / lists give exactly the
    oracle's restatement of diffusers 0.18.2 (custom_pipeline_stable_diffusion_img2img.py:638)."""
    from PIL import Image
    from oracle import vae_preprocess_ref
    from phendiff_amd.vae import VaeImageProcessor
    ip = VaeImageProcessor(vae_scale_factor=8)
    rng = np.random.default_rng(1)
    u8 = rng.integers(0, 256, size=(21, 34, 3), dtype=np.uint8)
    arr = rng.random((2, 16, 8, 3), dtype=np.float32)
    t = torch.rand(3, 16, 16)
    for inp in (Image.fromarray(u8), [Image.fromarray(u8), Image.fromarray(u8[:, ::-1].copy())], arr, [arr[0], arr[1]], arr * 2 - 1,
                t, [t, t], torch.rand(2, 3, 8, 8), torch.rand(2, 3, 8, 8) * 2 - 1):
        got, want = ip.preprocess(inp), vae_preprocess_ref(inp)
        assert got.dtype == torch.float32 and torch.equal(got, want)
    lat = torch.randn(2, 4, 5, 5)
    assert ip.preprocess(lat) is lat or torch.equal(ip.preprocess(lat), lat)
    for bad in (rng.random((1, 12, 8, 3), dtype=np.float32), torch.rand(1, 3, 12, 8), "x", []):
        with pytest.raises(ValueError):
            ip.preprocess(bad)
