"""Generates the committed golden vectors from the CPU oracle (run here, in the build container):

    python tests/golden/make_golden.py

The reference itself cannot produce fixtures (its hot path imports diffusers==0.18.2, absent offline; SURVEY.md
section 8c), so these vectors pin the ORACLE's outputs: the GPU tests compare the HIP path with them, and the CPU
tests re-derive them from the oracle to catch drift.  Weights: torch default init under manual_seed(0) in the
oracle's construction order; inverse-scheduler variant "0.18.2".
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import (CondUNet2DRef, ConditionalDDIMPipelineRef, DDIMInverseSchedulerRef, DDIMSchedulerRef,  # noqa: E402
                    UNET_CONFIGS, ddib_ref, linear_interp_custom_guidance_inverted_start_ref)

SCHED_3K = dict(num_train_timesteps=3000, beta_start=1e-4, beta_end=0.02, beta_schedule="scaled_linear",
                clip_sample=True, clip_sample_range=1.0, prediction_type="v_prediction",
                rescale_betas_zero_snr=True, timestep_spacing="trailing")


def synth_batch(B, size, seed=1234):
    g = torch.Generator().manual_seed(seed)
    labels = torch.arange(B) % 2
    x = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    x = (x + 0.25 * (2 * labels.float() - 1).view(B, 1, 1, 1)).clamp(-1, 1)
    return x, labels


SD_TINY_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(64, 128), layers_per_block=1,
                    down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
                    attention_head_dim=(1, 2), cross_attention_dim=96, norm_num_groups=32)
SD_TINY_VAE = dict(block_out_channels=(32, 64), layers_per_block=1)
SD_SCHED = dict(beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085, clip_sample=False, clip_sample_range=1.0,
                num_train_timesteps=1000, prediction_type="v_prediction", rescale_betas_zero_snr=False,
                set_alpha_to_one=False, steps_offset=1, thresholding=False, timestep_spacing="leading")


def sd_tiny_pipe():
    """Tiny latent-diffusion stack (same block types as SD-2.1; construction order fixes the seeded weights)."""
    from oracle import AutoencoderKLRef, CustomEmbeddingRef, SDImg2ImgPipelineRef, UNet2DConditionRef
    torch.manual_seed(0)
    unet = UNet2DConditionRef(**SD_TINY_UNET).eval()
    vae = AutoencoderKLRef(**SD_TINY_VAE).eval()
    emb = CustomEmbeddingRef(2, SD_TINY_UNET["cross_attention_dim"])
    return SDImg2ImgPipelineRef(vae, unet, DDIMSchedulerRef(**SD_SCHED), emb)


def main_sd():
    """custom_pipeline_stable_diffusion_img2img: DDIB (VAE encode -> invert -> class swap -> denoise -> decode) and the
    CFG forward-start transfer, 32x32 images (16x16 latents), S = 4."""
    from oracle import sd_cfg_forward_start_ref, sd_ddib_ref
    pipe = sd_tiny_pipe()
    x, labels = synth_batch(4, 32)
    out, inverted, latents = sd_ddib_ref(pipe, x, labels, 1 - labels, 4, generator=torch.Generator().manual_seed(11))
    cfg_out, cfg_lat = sd_cfg_forward_start_ref(pipe, x, 1 - labels, 3.0, 0.5, 4, generator=torch.Generator().manual_seed(12),
                                                output_type="np+latent")
    np.savez_compressed(os.path.join(HERE, "sd_tiny_32_s4.npz"), images=x.numpy(), labels=labels.numpy(), latents=latents.numpy(),
                        inverted=inverted.numpy(), ddib_out=out, cfg_out=cfg_out, cfg_latents=cfg_lat.numpy())
    print("wrote sd_tiny_32_s4.npz")


def main_sd_guided():
    """Gradient-guided transfer, latent-diffusion branch (utils_Img2Img.py:651-760 with a CustomStableDiffusionImg2ImgPipeline):
    tiny stack, 32x32 images (16x16 latents), S = 3, p = 2; loss scale 0.5 so the gradient term is visible."""
    from oracle import sd_linear_interp_custom_guidance_inverted_start_ref
    pipe = sd_tiny_pipe()
    x, labels = synth_batch(2, 32)
    image, guided, inverted = sd_linear_interp_custom_guidance_inverted_start_ref(pipe, x, labels, 1 - labels, 2, 0.5, 3,
                                                                                  generator=torch.Generator().manual_seed(13))
    np.savez_compressed(os.path.join(HERE, "guided_sd_tiny_32_s3.npz"), images=x.numpy(), labels=labels.numpy(), p=np.float32(2),
                        guidance_loss_scale=np.float32(0.5), out=image.numpy(), guided_latents=guided.numpy(), inverted=inverted.numpy())
    print("wrote guided_sd_tiny_32_s3.npz")


def main():
    torch.manual_seed(0)
    unet = CondUNet2DRef(**dict(UNET_CONFIGS["super_small"], sample_size=32)).eval()
    sched = DDIMSchedulerRef(**SCHED_3K)
    pipe = ConditionalDDIMPipelineRef(unet, sched)
    x, labels = synth_batch(4, 32)
    out, inverted = ddib_ref(pipe, x, labels, 1 - labels, 4)
    with torch.no_grad():
        eps = unet(x, 1500, class_labels=labels).sample
    np.savez_compressed(os.path.join(HERE, "ddib_super_small_32_s4.npz"), images=x.numpy(), labels=labels.numpy(),
                        inverted=inverted.numpy(), out_images=out, unet_out_t1500=eps.numpy())
    # gradient-guided transfer (utils_Img2Img.py:651-760), p = 2, S = 3; loss scale 0.5 so the gradient term is visible
    guided = linear_interp_custom_guidance_inverted_start_ref(pipe, x[:2], labels[:2], 1 - labels[:2], 2, 0.5, 3)
    np.savez_compressed(os.path.join(HERE, "guided_super_small_32_s3.npz"), images=x[:2].numpy(), labels=labels[:2].numpy(),
                        p=np.float32(2), guidance_loss_scale=np.float32(0.5), out=guided.numpy())
    # scheduler tables (int64 grids are bit-exact requirements)
    sched.set_timesteps(50)
    inv = DDIMInverseSchedulerRef.from_config(sched.config)
    inv.set_timesteps(50)
    np.savez_compressed(os.path.join(HERE, "scheduler_3k.npz"), alphas_cumprod=sched.alphas_cumprod.numpy(),
                        timesteps_50=sched.timesteps.numpy(), inv_alphas_cumprod=inv.alphas_cumprod.numpy(),
                        inv_timesteps_50=inv.timesteps.numpy())
    print("wrote golden fixtures to", HERE)


def main_google():
    """models_configs/denoiser/orig_google_ddpm_model_denoiser.json (113.7 M parameters: one 512-wide attention head, six
    levels, eps 1e-6, freq_shift 1, pad-0 downsamplers, no class table): one UNet evaluation and a DDIB round trip (S = 2) at
    64x64.  The model is unconditional: the class labels pass through `_inversion` / the pipeline unused (cond_unet_2d.py:297)."""
    torch.manual_seed(0)
    unet = CondUNet2DRef(**dict(UNET_CONFIGS["orig_google_ddpm"], sample_size=64)).eval()
    pipe = ConditionalDDIMPipelineRef(unet, DDIMSchedulerRef(**SCHED_3K))
    x, labels = synth_batch(2, 64)
    with torch.no_grad():
        eps = unet(x, 1500).sample
    out, inverted = ddib_ref(pipe, x, labels, 1 - labels, 2)
    np.savez_compressed(os.path.join(HERE, "ddib_google_ddpm_64_s2.npz"), images=x.numpy(), labels=labels.numpy(),
                        inverted=inverted.numpy(), out_images=out, unet_out_t1500=eps.numpy())
    print("wrote ddib_google_ddpm_64_s2.npz")


def main_sd21_denoiser():
    """models_configs/denoiser/SD_2-1_config.json (641.9 M parameters: pixel-space class-conditional UNet with SD-2.1 widths,
    d = 8 attention with 40 / 80 / 160 heads on three levels): one UNet evaluation and a DDIB class transfer (S = 2) at 32x32."""
    torch.manual_seed(0)
    unet = CondUNet2DRef(**dict(UNET_CONFIGS["SD_2-1_config"], sample_size=32)).eval()
    pipe = ConditionalDDIMPipelineRef(unet, DDIMSchedulerRef(**SCHED_3K))
    x, labels = synth_batch(2, 32)
    with torch.no_grad():
        eps = unet(x, 1500, class_labels=labels).sample
    out, inverted = ddib_ref(pipe, x, labels, 1 - labels, 2)
    np.savez_compressed(os.path.join(HERE, "ddib_sd21_denoiser_32_s2.npz"), images=x.numpy(), labels=labels.numpy(),
                        inverted=inverted.numpy(), out_images=out, unet_out_t1500=eps.numpy())
    print("wrote ddib_sd21_denoiser_32_s2.npz")


if __name__ == "__main__":
    if "--sd21-denoiser" in sys.argv:
        main_sd21_denoiser()
    elif "--sd-guided" in sys.argv:
        main_sd_guided()
    elif "--sd" in sys.argv:
        main_sd()
    elif "--google" in sys.argv:
        main_google()
    else:
        main()
        main_sd()
        main_sd_guided()
        main_google()
        main_sd21_denoiser()
