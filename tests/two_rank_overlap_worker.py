"""One rank of the two-rank data-parallel check (launched twice by tests/test_gpu_two_rank_overlap.py; not a test module).

Both ranks sit on cuda:0 of the one-GPU box and talk over the **gloo** backend on device tensors (RCCL refuses two ranks on one
device; gloo stages through the host -- fine for correctness): each rank runs the trainer's OVERLAPPED step -- the bucketed
all-reduce issued from inside the backward on a second stream (`UNetTrainer._forward_backward_overlapped`,
`unet_train.py`) -- on its half of a batch, at a different pace than its peer (rank 1 sleeps before each step, and its buckets
are handed over later: it also sleeps inside the first bucket hook), and checks what DDP guarantees in the reference
(train.py:62-74,311-326; utils_training.py:436):

  (i)   the flat gradient after the collectives == torch.autograd of the CPU oracle on the CONCATENATED batch (mean semantics),
  (ii)  parameters bit-identical across ranks after 3 optimisation steps (conditional, unconditional, conditional), although
        rank 1 started from different weights (the wrap-time broadcast from rank 0),
  (iii) the class table / CustomEmbedding untouched by the unconditional step on both ranks.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/two_rank_overlap_worker.py {pixel|sd}
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def _slow_hooks(tr, rank):
    """Rank 1 delays the hand-over of its first gradient bucket: the peers' collectives meet at different times."""
    if rank != 1:
        return
    orig = tr.forward_backward

    def fb(*a, after_op=None, **kw):
        if after_op:
            first = min(after_op)
            inner = after_op[first]
            after_op = dict(after_op)
            after_op[first] = lambda: (time.sleep(0.2), inner())
        return orig(*a, after_op=after_op, **kw)
    tr.forward_backward = fb


def _same_on_both_ranks(t, what):
    mine = t.detach().float().cpu().contiguous()
    both = [torch.empty_like(mine) for _ in range(2)]
    dist.all_gather(both, mine)
    assert torch.equal(both[0], both[1]), f"{what}: ranks differ (max |d| = {float((both[0] - both[1]).abs().max())})"


def pixel(rank):
    from test_gpu_unet_backward import batch, compare, oracle_grads
    from test_gpu_unet_ddib import make_pair
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", 32, "f32")
    if rank == 1:                                  # a rank that "loaded different weights": the trainer must overwrite them
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.01)
    sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
    _, ref = oracle_grads(r, noisy, ts, target, labels=labels)            # full batch of 4 = mean over both halves
    tr = UNetTrainer(m, sched, lr=2e-4, use_ema=True)
    _same_on_both_ranks(tr.opt.flat, "parameters after the wrap-time broadcast")
    _slow_hooks(tr, rank)
    sl = slice(2 * rank, 2 * rank + 2)
    half = [t[sl].cuda() for t in (noisy, ts, clean, noise)]
    if rank == 1:
        time.sleep(0.3)
    tr._forward_backward_overlapped(*half, labels[sl].cuda(), None, None, 2, 1 << 20)
    torch.cuda.synchronize()
    assert len(tr._buckets) >= 8, len(tr._buckets)
    compare(ref, tr.grads, 2e-4, 2e-5)                                    # (i)
    tr.opt.grad.zero_()
    # (ii) + (iii): three full steps through the public entry point; the oracle takes the same steps on the full batch
    opt = torch.optim.AdamW(r.parameters(), lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    zeros = torch.zeros(4, 256)
    for k, uncond in enumerate((False, True, False)):
        table_before = tr.params["class_embedding.weight"].clone()
        if uncond:
            loss_ref, _ = oracle_grads(r, noisy, ts, target, class_emb=zeros)
            r.class_embedding.weight.grad = None                          # torch leaves it None: AdamW skips the table
        else:
            loss_ref, _ = oracle_grads(r, noisy, ts, target, labels=labels)
        torch.nn.utils.clip_grad_norm_(r.parameters(), 1.0)
        opt.step()
        if rank == 1:
            time.sleep(0.15)
        kw = dict(class_emb=zeros[sl].cuda()) if uncond else dict(class_labels=labels[sl].cuda())
        loss = tr.step(*half, overlap=True, bucket_bytes=1 << 20, **kw)
        torch.cuda.synchronize()
        both = [torch.zeros(1) for _ in range(2)]
        dist.all_gather(both, loss.detach().float().cpu().reshape(1))
        assert abs(float(sum(both)) / 2 - float(loss_ref)) < 2e-4 * abs(float(loss_ref)), (k, both, float(loss_ref))
        if uncond:
            assert torch.equal(tr.params["class_embedding.weight"], table_before), "unconditional step touched the class table"
    _same_on_both_ranks(tr.opt.flat, "parameters after 3 overlapped steps")
    _same_on_both_ranks(tr.opt.ema, "EMA after 3 overlapped steps")
    sd = r.state_dict()
    num = sum(float((p.detach().cpu() - sd[n]).double().pow(2).sum()) for n, p in m.named_parameters())
    den = sum(float(sd[n].double().pow(2).sum()) for n, _ in m.named_parameters())
    assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5                  # == torch's AdamW on the full batch


def pixel_frozen(rank):
    """--attention_fine_tuning with two ranks (train.py:201-220 + 311-326): only the attention parameters are exchanged and stepped;
    frozen parameters still take rank 0's values at wrap time and never move."""
    from test_gpu_frozen_params import adamw, attention_fine_tuning, check_against_oracle, oracle_step
    from test_gpu_unet_backward import batch
    from test_gpu_unet_ddib import make_pair
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", 32, "f32")
    attention_fine_tuning(r, verbatim=False)
    attention_fine_tuning(m)
    before = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}       # rank 0's weights = the oracle's
    if rank == 1:
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.01)
    sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
    tr = UNetTrainer(m, sched, lr=2e-4, use_ema=True)
    trainable = set(tr.grads) - tr.frozen
    assert len(trainable) == 60
    _same_on_both_ranks(tr.opt.flat, "parameters after the wrap-time broadcast")
    _slow_hooks(tr, rank)
    sl = slice(2 * rank, 2 * rank + 2)
    half = [t[sl].cuda() for t in (noisy, ts, clean, noise)]
    opt = adamw(r, 2e-4)
    for k in range(3):
        loss_ref = oracle_step(r, opt, noisy, ts, target, class_labels=labels)
        if rank == 1:
            time.sleep(0.15)
        loss = tr.step(*half, class_labels=labels[sl].cuda(), overlap=True, bucket_bytes=64 << 10)
        torch.cuda.synchronize()
        both = [torch.zeros(1) for _ in range(2)]
        dist.all_gather(both, loss.detach().float().cpu().reshape(1))
        assert abs(float(sum(both)) / 2 - loss_ref) < 2e-4 * abs(loss_ref), (k, both, loss_ref)
    # every bucket lies inside a run of trainable parameters
    off, spans = 0, []
    for n in tr.grads:
        kk = tr.grads[n].numel()
        if n in tr.frozen:
            spans.append((off, off + kk))
        off += kk
    assert len(tr._buckets) >= 4
    for a, b, _ in tr._buckets:
        assert not any(a < hi and lo < b for lo, hi in spans), (a, b)
    _same_on_both_ranks(tr.opt.flat, "parameters after 3 overlapped steps")
    check_against_oracle(r, m, before, trainable)


def sd(rank):
    import phendiff_amd as P
    from test_gpu_sd_unet import TINY, make_pair
    from test_gpu_sd_unet_backward import batch, oracle_grads
    from test_gpu_unet_backward import compare
    r, emb, m, e2 = make_pair(TINY, "f32")
    if rank == 1:
        with torch.no_grad():
            for p in list(m.parameters()) + list(e2.parameters()):
                p.add_(0.01)
    sched, clean, noise, ts, labels, noisy, target = batch(4, 16)
    _, ref = oracle_grads(r, emb, noisy, ts, target, labels)
    tr = P.SDUNetTrainer(m, e2, sched, lr=2e-4, use_ema=True)
    _same_on_both_ranks(tr.opt.flat, "parameters after the wrap-time broadcast")
    _slow_hooks(tr, rank)
    sl = slice(2 * rank, 2 * rank + 2)
    half = [t[sl].cuda() for t in (noisy, ts, clean, noise)]
    tr._uncond = False
    if rank == 1:
        time.sleep(0.3)
    tr._forward_backward_overlapped(*half, labels[sl].cuda(), None, None, 2, 256 << 10)
    torch.cuda.synchronize()
    assert len(tr._buckets) >= 4, len(tr._buckets)
    compare(ref, tr.grads, 3e-4, 3e-5)                                    # (i)
    tr.opt.grad.zero_()
    allp = list(r.parameters()) + list(emb.parameters())
    opt = torch.optim.AdamW(allp, lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    name = "class_embedding.inner_module.weight"
    for uncond in (False, True, False):
        table_before = tr.params[name].clone()
        loss_ref, _ = oracle_grads(r, emb, noisy, ts, target, labels, unconditional=uncond)
        if uncond:
            emb.inner_module.weight.grad = None
        torch.nn.utils.clip_grad_norm_(allp, 1.0)
        opt.step()
        if rank == 1:
            time.sleep(0.15)
        loss = tr.step(*half, labels[sl].cuda(), unconditional=uncond, overlap=True, bucket_bytes=256 << 10)
        torch.cuda.synchronize()
        both = [torch.zeros(1) for _ in range(2)]
        dist.all_gather(both, loss.detach().float().cpu().reshape(1))
        assert abs(float(sum(both)) / 2 - float(loss_ref)) < 2e-4 * abs(float(loss_ref))
        if uncond:
            assert torch.equal(tr.params[name], table_before), "unconditional step touched the CustomEmbedding"
    _same_on_both_ranks(tr.opt.flat, "parameters after 3 overlapped steps")
    sd_ = dict(r.state_dict())
    num = sum(float((p.detach().cpu() - sd_[n]).double().pow(2).sum()) for n, p in m.named_parameters())
    den = sum(float(sd_[n].double().pow(2).sum()) for n, _ in m.named_parameters())
    assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5


def main():
    which = sys.argv[1]
    rank = int(os.environ["RANK"])
    assert int(os.environ["WORLD_SIZE"]) == 2
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        {"pixel": pixel, "sd": sd, "pixel_frozen": pixel_frozen}[which](rank)
        dist.barrier()
        print(f"two_rank_overlap_worker {which} rank {rank}: OK", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
