"""Training-step building blocks: host schedules on CPU; fused loss / grad-norm / AdamW+EMA kernels on
the GPU against torch's own implementations (torch.nn.functional.mse_loss, clip_grad_norm_, torch.optim.AdamW)."""
import math
import os
import socket

import pytest
import torch
import torch.nn.functional as F

import phendiff_amd as P
from phendiff_amd import training as T


def test_ema_decay_schedule():
    # d_1 = 0, then 1 - (1 + step)^-0.75 capped at 0.9999
    assert T.ema_decay(1) == 0.0
    assert T.ema_decay(2) == pytest.approx(1 - 2 ** -0.75)
    assert T.ema_decay(101) == pytest.approx(1 - 101 ** -0.75)
    assert T.ema_decay(10 ** 9) == 0.9999
    assert T.ema_decay(5, use_ema_warmup=False) == pytest.approx(5 / 14)
    assert T.ema_decay(3, update_after_step=5) == 0.0


def test_cosine_lr_matches_lambda_lr():
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: T.cosine_lr_factor(s, 500, 30000))
    assert T.cosine_lr_factor(0, 500, 30000) == 0.0 and T.cosine_lr_factor(250, 500, 30000) == 0.5
    assert T.cosine_lr_factor(500, 500, 30000) == 1.0
    assert T.cosine_lr_factor(15250, 500, 30000) == pytest.approx(0.5)
    assert T.cosine_lr_factor(30000, 500, 30000) == pytest.approx(0.0, abs=1e-12)
    for _ in range(3):
        opt.step(); sched.step()
    assert sched.get_last_lr()[0] == pytest.approx(3 / 500)
    assert T.scaled_lr(1e-4, 8) == pytest.approx(1e-4 * math.sqrt(8))


def test_unconditional_flags_agree_across_ranks_without_traffic():
    a, b = T.UnconditionalStepFlags(1234, 0.1), T.UnconditionalStepFlags(1234, 0.1)
    fa, fb = [a.next() for _ in range(2000)], [b.next() for _ in range(2000)]
    assert fa == fb and 120 < sum(fa) < 280


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.SiLU(), torch.nn.Linear(5, 3))
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = rank * 4, rank * 4 + 4
    F.mse_loss(model(x[lo:hi]), y[lo:hi]).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    T.allreduce_mean_(flat, bucket_bytes=64)          # several small buckets on purpose
    # the non-overlapped path of a partly frozen model: only the trainable runs are exchanged, all collectives in flight together
    part = torch.arange(40, dtype=torch.float32) * (rank + 1)
    T.allreduce_mean_ranges_(part, [(2, 5), (10, 1), (20, 17)], bucket_bytes=16)
    want = torch.arange(40, dtype=torch.float32) * (rank + 1)
    for o, k in [(2, 5), (10, 1), (20, 17)]:
        want[o:o + k] = torch.arange(40, dtype=torch.float32)[o:o + k] * 1.5
    assert torch.equal(part, want)
    if rank == 0:
        out.put(flat.clone())
    dist.destroy_process_group()


def test_dp_gradient_average_equals_large_batch_gradient():
    """2-rank gloo: averaged per-rank gradients == single-process gradient of the concatenated batch (DDP semantics)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    flat = out.get(timeout=120)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.SiLU(), torch.nn.Linear(5, 3))
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    F.mse_loss(model(x), y).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(flat, ref, atol=1e-6)


def _bcast_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # ranks that were initialised differently ...
    flat = torch.randn(1000)
    mine = flat.clone()
    T.broadcast_from_rank0_(flat)                      # ... all continue from rank 0's parameters (DDP wrap, train.py:311-326)
    out.put((rank, mine, flat.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_parameter_broadcast_at_wrap_time():
    import torch.multiprocessing as mp
    import queue
    ctx = mp.get_context("spawn")
    for attempt in range(3):          # (the free port is probed, released, then bound by rank 0: another process can take it in between -- retry on a new one)
        out = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            got = dict((r, (a, b)) for r, a, b in (out.get(timeout=120) for _ in range(2)))
        except queue.Empty:
            got = None
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
        if got is not None and all(p.exitcode == 0 for p in procs):
            break
    else:
        raise AssertionError("two-rank gloo rendezvous failed three times")
    assert not torch.equal(got[0][0], got[1][0])                       # they did start apart
    assert torch.equal(got[0][1], got[0][0]) and torch.equal(got[1][1], got[0][0])
    x = torch.ones(3)
    assert T.broadcast_from_rank0_(x) is x                             # no process group: no-op


def test_randn_tensor_draws_per_sample_from_a_list_of_generators():
    """diffusers randn_tensor semantics used by scheduler.step(eta > 0) and the pipelines: one (1, ...) draw per generator."""
    from phendiff_amd.schedulers import randn_tensor
    gens = [torch.Generator().manual_seed(s) for s in (5, 6, 7)]
    got = randn_tensor((3, 2, 4, 4), gens, "cpu")
    want = torch.cat([torch.randn((1, 2, 4, 4), generator=torch.Generator().manual_seed(s)) for s in (5, 6, 7)], 0)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        randn_tensor((2, 2, 4, 4), gens, "cpu")
    one = randn_tensor((3, 2), torch.Generator().manual_seed(1), "cpu")
    assert torch.equal(one, torch.randn((3, 2), generator=torch.Generator().manual_seed(1)))


@pytest.mark.gpu
@pytest.mark.parametrize("pt", ["epsilon", "sample", "v_prediction"])
def test_diffusion_loss_and_gradient(pt):
    from oracle import DDIMSchedulerRef
    cfg = dict(P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"], prediction_type=pt)
    sched, ref_s = P.DDIMScheduler(**cfg), DDIMSchedulerRef(**cfg)
    g = torch.Generator().manual_seed(3)
    out, clean, noise = (torch.randn(4, 3, 16, 16, generator=g) for _ in range(3))
    ts = torch.tensor([5, 700, 1500, 2900])
    o = out.clone().requires_grad_(True)
    if pt == "epsilon":
        ref = F.mse_loss(o, noise)
    elif pt == "sample":
        a = ref_s.alphas_cumprod[ts].float()[:, None, None, None]          # extract_into_tensor
        ref = ((a / (1 - a)) * F.mse_loss(o, clean, reduction="none")).mean()
    else:
        ref = F.mse_loss(o, ref_s.get_velocity(clean, noise, ts))
    ref.backward()
    loss, grad = T.DiffusionLoss(sched, "cuda:0")(out.cuda(), clean.cuda(), noise.cuda(), ts.cuda())
    assert float(loss.cpu()) == pytest.approx(float(ref.detach()), rel=2e-6)
    assert torch.allclose(grad.cpu(), o.grad, rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_fused_clip_adamw_ema_matches_torch():
    torch.manual_seed(0)
    shapes = [(64, 3, 3, 3), (64,), (300, 17), (5,)]
    ref_p = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    dev_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    opt = torch.optim.AdamW(ref_p, lr=3e-3, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    fused = T.FlatAdamWEMA(dev_p, lr=3e-3)
    shadow = [p.detach().clone() for p in ref_p]
    g = torch.Generator().manual_seed(1)
    for step in range(1, 5):
        grads = [torch.randn(s, generator=g) * (3.0 if step % 2 else 0.01) for s in shapes]   # clipped / unclipped steps
        for p, q, gr in zip(ref_p, dev_p, grads):
            p.grad = gr.clone()
            q.grad.copy_(gr)
        lr = 3e-3 * T.cosine_lr_factor(step, 2, 10)
        for grp in opt.param_groups:
            grp["lr"] = lr
        norm = torch.nn.utils.clip_grad_norm_(ref_p, 1.0)
        opt.step()
        d = T.ema_decay(step)
        for s_, p in zip(shadow, ref_p):
            s_.sub_((1 - d) * (s_ - p.detach()))
        fused.step(lr=lr)
        torch.cuda.synchronize()
        assert float(fused.grad_norm.cpu()) == pytest.approx(float(norm), rel=1e-5)
        assert float(fused.grad.abs().max().cpu()) == 0.0                      # zero_grad
        for p, q in zip(ref_p, dev_p):
            assert torch.allclose(q.detach().cpu(), p.detach(), rtol=2e-5, atol=1e-7), step
        flat_shadow = torch.cat([s_.reshape(-1) for s_ in shadow])
        assert torch.allclose(fused.ema.cpu(), flat_shadow, rtol=2e-5, atol=1e-7)
    st = opt.state[ref_p[2]]
    off = sum(torch.Size(s).numel() for s in shapes[:2])
    k = torch.Size(shapes[2]).numel()
    assert torch.allclose(fused.exp_avg[off:off + k].cpu(), st["exp_avg"].reshape(-1), rtol=2e-5, atol=1e-8)
    assert torch.allclose(fused.exp_avg_sq[off:off + k].cpu(), st["exp_avg_sq"].reshape(-1), rtol=2e-5, atol=1e-10)


@pytest.mark.gpu
def test_sample_training_inputs_semantics():
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    clean = torch.rand(4, 3, 8, 8, device="cuda") * 2 - 1
    noise, ts, noisy = T.sample_training_inputs(clean, sched, cpu_generator=torch.Generator().manual_seed(0))
    assert torch.equal(noise.cpu(), torch.randn(clean.shape, generator=torch.Generator().manual_seed(0)))   # CPU RNG, then H2D
    assert ts.dtype == torch.int64 and ts.min() >= 0 and ts.max() < 3000
    a = sched.alphas_cumprod[ts.cpu()][:, None, None, None]
    assert torch.allclose(noisy.cpu(), a ** 0.5 * clean.cpu() + (1 - a) ** 0.5 * noise.cpu(), atol=1e-6)


def test_optimizer_state_in_accelerate_layout_roundtrips_through_torch_adamw(tmp_path):
    """optimizer.bin of accelerator.save_state = torch.optim.AdamW.state_dict(): built from the engine's flat moment buffers
    it must load into a real torch AdamW over the same module (parameters() order), land on the right parameters, and load
    back into flat buffers unchanged."""
    import phendiff_amd as P
    from phendiff_amd import train_state as TS
    m = P.CustomCondUNet2DModel(**dict(P.UNET_CONFIGS["super_small"], sample_size=32))
    order = P.training_param_order(m)
    fnames = [n for n, _ in order]
    pnames = [n for n, _ in m.named_parameters()]
    sizes = {n: p.shape for n, p in order}
    n = sum(p.numel() for _, p in order)
    g = torch.Generator().manual_seed(0)
    ea, eas = torch.randn(n, generator=g), torch.rand(n, generator=g)
    sd = TS.optimizer_state_dict(pnames, fnames, sizes, ea, eas, step=7, lr=3e-4, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6)
    for p in m.parameters():
        p.requires_grad_(True)
    opt = torch.optim.AdamW(m.parameters(), lr=1.0)
    opt.load_state_dict(sd)                                     # torch accepts the layout
    assert opt.param_groups[0]["lr"] == 3e-4 and opt.param_groups[0]["betas"] == (0.95, 0.999)
    # a parameter that sits elsewhere in the flat buffer than in parameters() order
    name = "down_blocks.2.attentions.0.to_k.weight"
    off = sum(sizes[k].numel() for k in fnames[:fnames.index(name)])
    st = opt.state[dict(m.named_parameters())[name]]
    assert float(st["step"]) == 7 and torch.equal(st["exp_avg"].reshape(-1), ea[off:off + sizes[name].numel()])
    torch.save(opt.state_dict(), tmp_path / "optimizer.bin")     # and torch's own file loads back into the flat layout
    ea2, eas2 = torch.empty(n), torch.empty(n)
    step = TS.load_optimizer_state_dict(torch.load(tmp_path / "optimizer.bin"), pnames, fnames, sizes, ea2, eas2)
    assert step == 7 and torch.equal(ea2, ea) and torch.equal(eas2, eas)
    # EMA state in diffusers EMAModel layout: shadow list in parameters() order
    esd = TS.ema_state_dict(ea, fnames, pnames, sizes, 7)
    assert len(esd["shadow_params"]) == len(pnames) and esd["optimization_step"] == 7
    assert torch.equal(esd["shadow_params"][pnames.index(name)].reshape(-1), ea[off:off + sizes[name].numel()])
    assert TS.lr_scheduler_state_dict(1e-4, 12, 5e-5)["last_epoch"] == 12


def test_checkpoint_folder_helpers(tmp_path):
    from phendiff_amd import train_state as TS
    assert TS.latest_checkpoint(str(tmp_path / "none")) is None
    for s in (10, 200, 30):
        (tmp_path / f"step_{s}").mkdir()
    assert TS.latest_checkpoint(str(tmp_path)).endswith("step_200")       # numeric, not lexicographic (utils_training.py:76)
    assert TS.resume_from_checkpoint(None, str(tmp_path / "none")) == (0, 0, 0, {})
