"""configs[3] on one MI355X: SD-2.1 UNet (865.9 M parameters, random init) + CustomEmbedding fine-tuning step at 64x64 latents
(512 px): _SD_prediction_wrapper forward -> v-prediction MSE -> backward -> clip + AdamW + EMA -> weight re-pack.
   python scripts/bench_sd_train.py [B] [latent size] [bf16|f32] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phendiff_amd as P  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    torch.manual_seed(0)
    with torch.device("cuda:0"):
        unet = P.SDUNet2DConditionModel(compute_dtype=mode, **P.SD21_UNET_CONFIG)
        emb = P.CustomEmbedding(2, 1024)
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["SD_orig_config"])
    tr = P.SDUNetTrainer(unet, emb, sched, lr=1e-5)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    clean = torch.randn(B, 4, size, size, device="cuda:0", generator=g) * 0.8
    noise = torch.randn(B, 4, size, size, device="cuda:0", generator=g)
    ts = torch.randint(0, 1000, (B,), device="cuda:0", generator=g)
    labels = torch.arange(B, device="cuda:0") % 2
    noisy = sched.add_noise(clean, noise, ts)
    t0 = time.perf_counter()
    losses = [float(tr.step(noisy, ts, clean, noise, labels))]
    torch.cuda.synchronize()
    print(f"first step (plan build + pack): {time.perf_counter() - t0:.1f} s; memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        losses.append(float(tr.step(noisy, ts, clean, noise, labels, unconditional=(i == 1))))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    plan = tr.plan_for(B, size, size)
    st = torch.cuda.current_stream().cuda_stream
    fl = sum(op.flops for op in plan.ops) + sum(op.flops for op in plan.bwd_ops)
    print(f"SD UNet train step {size}x{size} latents, B={B}, {mode}: {dt * 1e3:.1f} ms/step = {B / dt:.2f} samples/s, "
          f"{fl / dt / 1e12:.0f} TF/s ({fl / B / 1e9:.0f} GF/sample fwd+bwd), {len(plan.ops)} + {len(plan.bwd_ops)} launches; losses {losses}")
    assert all(l == l for l in losses)
    for name, ops in (("forward", plan.ops), ("backward", plan.bwd_ops)):
        acc = plan._profile_ops(ops, st, reps=1)
        tot = sum(d["ms"] for d in acc.values())
        print(f" {name}: {tot:.1f} ms")
        for k, d in sorted(acc.items(), key=lambda kv: -kv[1]["ms"]):
            tf = d["flops"] / d["ms"] / 1e9 if d["ms"] > 0 else 0
            print(f"   {k:18s} {d['ms']:8.3f} ms  x{d['launches']:5.0f}  {tf:8.1f} TF/s  {d['bytes'] / max(d['ms'], 1e-9) / 1e6:8.1f} GB/s")
    t0 = time.perf_counter()
    tr.opt.step(1e-5)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tr.refresh_weights()
    torch.cuda.synchronize()
    print(f" optimizer (clip + AdamW + EMA over {tr.opt.flat.numel() / 1e6:.0f} M parameters): {(t1 - t0) * 1e3:.1f} ms; re-pack: {(time.perf_counter() - t1) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
