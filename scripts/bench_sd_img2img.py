"""configs[4] on one MI355X: custom_pipeline_stable_diffusion_img2img DDIB at 512x512 (64x64 latents) with the full-size
SD-2.1 UNet (865.9 M parameters) and SD VAE (83.7 M), random init: VAE encode -> S-step DDIM inversion under the original
class -> class swap -> S-step denoising -> VAE decode.
   python scripts/bench_sd_img2img.py [B] [image size] [S] [bf16|f32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phendiff_amd as P  # noqa: E402


def timed(fn, n=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    mode = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    torch.manual_seed(0)
    with torch.device("cuda:0"):
        unet = P.SDUNet2DConditionModel(compute_dtype=mode, **P.SD21_UNET_CONFIG)
        vae = P.AutoencoderKL(compute_dtype=mode)
        emb = P.CustomEmbedding(2, 1024)
    pipe = P.CustomStableDiffusionImg2ImgPipeline(vae, unet, P.DDIMScheduler(**P.SCHEDULER_CONFIGS["SD_orig_config"]), emb)
    g = torch.Generator().manual_seed(1234)
    labels = (torch.arange(B) % 2).cuda()
    x = (torch.rand(B, 3, size, size, generator=g) * 2 - 1).cuda()
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    # warm-up: builds every plan
    P.ddib(pipe, x, labels, 1 - labels, 2, generator=gen)
    t_enc, lat = timed(lambda: P.encode_to_latents(pipe, x, gen), 3)
    t_dec, img = timed(lambda: P.decode_to_images(pipe, lat), 3)
    assert bool(torch.isfinite(img).all())
    t_all, out = timed(lambda: P.ddib(pipe, x, labels, 1 - labels, S, generator=gen))
    assert out.shape == (B, size, size, 3)
    print(f"SD img2img DDIB {size}x{size}, B={B}, S={S}+{S}, {mode}: {t_all:.3f} s/batch = {B / t_all:.3f} images/s; "
          f"VAE encode {t_enc * 1e3:.1f} ms ({B * 1116.7 * (size / 512) ** 2 / t_enc / 1e3:.0f} TF/s), "
          f"decode {t_dec * 1e3:.1f} ms ({B * 2514.5 * (size / 512) ** 2 / t_dec / 1e3:.0f} TF/s), "
          f"UNet share {(t_all - t_enc - t_dec) / t_all:.3f}")
    st = torch.cuda.current_stream().cuda_stream
    for kind in ("enc", "dec"):
        plan = next(p for k, p in vae._plans.items() if k[0] == kind and k[1] == min(B, vae._max_batch(size, size, kind)))
        acc = plan._profile_ops(plan.ops, st, reps=2)
        print(f" VAE {kind} per-kernel ({len(plan.ops)} launches):")
        for k, d in sorted(acc.items(), key=lambda kv: -kv[1]["ms"]):
            tf = d["flops"] / d["ms"] / 1e9 if d["ms"] > 0 else 0
            print(f"   {k:14s} {d['ms']:8.3f} ms  x{d['launches']:5.0f}  {tf:8.1f} TF/s  {d['bytes'] / max(d['ms'], 1e-9) / 1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
