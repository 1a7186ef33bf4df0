"""Per-kernel-kind time of one training forward + backward (HIP events between launches).  Usage:
   python scripts/profile_train.py [B] [size] [bf16|f32] [model]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phendiff_amd as P  # noqa: E402
from phendiff_amd.unet_train import UNetTrainer  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    name = sys.argv[4] if len(sys.argv) > 4 else "super_small"
    torch.manual_seed(0)
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **dict(P.UNET_CONFIGS[name], sample_size=size)).to("cuda:0")
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    tr = UNetTrainer(m, sched, lr=1e-4)
    g = torch.Generator().manual_seed(1)
    clean = (torch.rand(B, 3, size, size, generator=g) * 2 - 1).cuda()
    noise = torch.randn(B, 3, size, size, generator=g).cuda()
    ts = torch.randint(0, 3000, (B,), generator=g).cuda()
    labels = (torch.arange(B) % 2).cuda()
    noisy = sched.add_noise(clean, noise, ts)
    for _ in range(2):
        loss = tr.step(noisy, ts, clean, noise, class_labels=labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        loss = tr.step(noisy, ts, clean, noise, class_labels=labels)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"step {dt * 1e3:.2f} ms  -> {B / dt:.1f} img/s  (B={B}, {size}x{size}, {mode}, {name}); loss {float(loss):.4f}")
    plan = tr.plan_for(B, size, size)
    st = torch.cuda.current_stream().cuda_stream
    for title, ops in (("forward", plan.ops), ("backward", plan.bwd_ops)):
        acc = plan._profile_ops(ops, st, reps=3)
        tot = sum(d["ms"] for d in acc.values())
        print(f"-- {title}: {tot:.3f} ms, {len(ops)} launches")
        for k, d in sorted(acc.items(), key=lambda kv: -kv[1]["ms"]):
            tf = d["flops"] / d["ms"] / 1e9 if d["ms"] > 0 else 0
            gb = d["bytes"] / d["ms"] / 1e6 if d["ms"] > 0 else 0
            print(f"   {k:14s} {d['ms']:8.3f} ms  x{d['launches']:5.0f}  {tf:8.1f} TF/s  {gb:8.1f} GB/s")
    # host-side pieces of a step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.refresh_weights()
    torch.cuda.synchronize()
    print(f"refresh_weights {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
    t0 = time.perf_counter()
    for _ in range(5):
        tr.opt.step()
    torch.cuda.synchronize()
    print(f"optimizer step {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
