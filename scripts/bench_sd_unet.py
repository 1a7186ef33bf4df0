"""SD-2.1 UNet (865.9 M parameters, random init) forward on one MI355X: per-kernel-kind time at 64x64 latents (512 px).
   python scripts/bench_sd_unet.py [B] [latent size] [bf16|f32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phendiff_amd as P  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    torch.manual_seed(0)
    t0 = time.perf_counter()
    with torch.device("cuda:0"):
        m = P.SDUNet2DConditionModel(compute_dtype=mode, **P.SD21_UNET_CONFIG)
        emb = P.CustomEmbedding(2, 1024)
    print(f"model built on device in {time.perf_counter() - t0:.1f} s; params {sum(p.numel() for p in m.parameters()):,}")
    x = torch.randn(B, 4, size, size, device="cuda:0")
    ts = torch.randint(0, 1000, (B,), device="cuda:0")
    ehs = P.class_emb_to_encoder_hidden_states(emb(torch.arange(B, device="cuda:0") % 2))
    for _ in range(2):
        out = m(x, ts, ehs).sample
    torch.cuda.synchronize()
    if os.environ.get("PD_PMC_FORWARDS_ONLY"):      # scripts/collect_profiles.sh pmc_side: exactly two forwards under the counters, nothing else
        return
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        out = m(x, ts, ehs).sample
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    assert torch.isfinite(out).all()
    plan = next(iter(m._plans.values()))
    st = torch.cuda.current_stream().cuda_stream
    acc = plan._profile_ops(plan.ops, st, reps=2)
    flops = sum(d["flops"] for d in acc.values())
    print(f"forward {dt * 1e3:.2f} ms  (B={B}, {size}x{size} latents, {mode}): {B / dt:.2f} samples/s, {flops / dt / 1e12:.0f} TF/s "
          f"({flops / B / 1e9:.1f} GF/sample), {len(plan.ops)} launches")
    for k, d in sorted(acc.items(), key=lambda kv: -kv[1]["ms"]):
        tf = d["flops"] / d["ms"] / 1e9 if d["ms"] > 0 else 0
        gb = d["bytes"] / d["ms"] / 1e6 if d["ms"] > 0 else 0
        print(f"   {k:14s} {d['ms']:8.3f} ms  x{d['launches']:5.0f}  {tf:8.1f} TF/s  {gb:8.1f} GB/s")


if __name__ == "__main__":
    main()
