"""Hyper-parameter VALUES of the reference's shipped configs (``models_configs/denoiser/*.json``,
``models_configs/noise_scheduler/*.json``) so the engine can be instantiated without the reference tree."""

UNET_CONFIGS = {
    # models_configs/denoiser/super_small.json -- 15 725 443 parameters
    "super_small": dict(
        act_fn="silu", attention_head_dim=8, block_out_channels=(64, 128, 256), center_input_sample=False,
        class_embed_type=None, down_block_types=("DownBlock2D", "DownBlock2D", "AttnDownBlock2D"),
        downsample_padding=1, flip_sin_to_cos=True, freq_shift=0, in_channels=3, layers_per_block=2,
        mid_block_scale_factor=1, norm_eps=1e-05, norm_num_groups=32, num_class_embeds=2, out_channels=3,
        resnet_time_scale_shift="default", sample_size=128, time_embedding_type="positional",
        up_block_types=("AttnUpBlock2D", "UpBlock2D", "UpBlock2D")),
    # models_configs/denoiser/small_denoiser_config.json -- 62 826 243 parameters
    "small_denoiser_config": dict(
        act_fn="silu", attention_head_dim=8, block_out_channels=(128, 256, 512), center_input_sample=False,
        class_embed_type=None, down_block_types=("DownBlock2D", "DownBlock2D", "AttnDownBlock2D"),
        downsample_padding=1, flip_sin_to_cos=True, freq_shift=0, in_channels=3, layers_per_block=2,
        mid_block_scale_factor=1, norm_eps=1e-05, norm_num_groups=32, num_class_embeds=2, out_channels=3,
        resnet_time_scale_shift="default", sample_size=128, time_embedding_type="positional",
        up_block_types=("AttnUpBlock2D", "UpBlock2D", "UpBlock2D")),
    # models_configs/denoiser/orig_google_ddpm_model_denoiser.json (UNet2DModel of google/ddpm-celebahq-256) -- 113 673 219
    # parameters: six levels, ONE attention head over all 512 channels (attention_head_dim null, cond_unet_2d.py:176-178,192-196),
    # eps 1e-6, sin-before-cos embedding with freq_shift 1, pad-0 (asymmetric) downsamplers, no class table
    "orig_google_ddpm_model_denoiser": dict(
        act_fn="silu", attention_head_dim=None, block_out_channels=(128, 128, 256, 256, 512, 512), center_input_sample=False,
        down_block_types=("DownBlock2D", "DownBlock2D", "DownBlock2D", "DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
        downsample_padding=0, flip_sin_to_cos=False, freq_shift=1, in_channels=3, layers_per_block=2, mid_block_scale_factor=1,
        norm_eps=1e-06, norm_num_groups=32, out_channels=3, sample_size=256, time_embedding_type="positional",
        up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D")),
    # models_configs/denoiser/SD_2-1_config.json -- 641 914 883 parameters: a pixel-space class-conditional CondUNet2DModel with
    # SD-2.1's widths; AttnDownBlock2D on the first three levels (head_dim 8 -> 40 / 80 / 160 heads; N = 16 384 tokens at 128^2).
    # The file's other keys (conv_in_kernel, conv_out_kernel, resnet_out_scale_factor, resnet_skip_time_act, upcast_attention,
    # use_linear_projection) are not arguments of CustomCondUNet2DModel: `from_config` drops them with a warning, as diffusers does.
    "SD_2-1_config": dict(
        act_fn="silu", attention_head_dim=8, block_out_channels=(320, 640, 1280, 1280), center_input_sample=False,
        class_embed_type=None, down_block_types=("AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
        downsample_padding=1, flip_sin_to_cos=True, freq_shift=0, in_channels=3, layers_per_block=2,
        mid_block_scale_factor=1, norm_eps=1e-05, norm_num_groups=32, num_class_embeds=2, out_channels=3,
        resnet_time_scale_shift="default", sample_size=128, time_embedding_type="positional",
        up_block_types=("UpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D")),
}
# keys of the shipped JSONs that CustomCondUNet2DModel.__init__ (cond_unet_2d.py:74-107) does not take
UNET_CONFIG_IGNORED_KEYS = ("conv_in_kernel", "conv_out_kernel", "resnet_out_scale_factor", "resnet_skip_time_act", "upcast_attention",
                            "use_linear_projection")

SCHEDULER_CONFIGS = {
    # models_configs/noise_scheduler/3k_steps_clipping_rescaling.json (paired with super_small in launch_script_DDIM.sh:46-47)
    "3k_steps_clipping_rescaling": dict(
        beta_schedule="scaled_linear", beta_end=0.02, beta_start=0.0001, clip_sample=True, clip_sample_range=1.0,
        num_train_timesteps=3000, prediction_type="v_prediction", rescale_betas_zero_snr=True,
        timestep_spacing="trailing"),
    # models_configs/noise_scheduler/1k_epsilon_pred.json
    "1k_epsilon_pred": dict(
        beta_schedule="scaled_linear", beta_end=0.02, beta_start=0.0001, clip_sample=True, clip_sample_range=1.0,
        num_train_timesteps=1000, prediction_type="epsilon", rescale_betas_zero_snr=True,
        timestep_spacing="trailing"),
    # models_configs/noise_scheduler/SD_orig_config.json
    "SD_orig_config": dict(
        beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085, clip_sample=False, clip_sample_range=1.0,
        num_train_timesteps=1000, prediction_type="v_prediction", rescale_betas_zero_snr=False,
        set_alpha_to_one=False, steps_offset=1, thresholding=False, timestep_spacing="leading"),
    # models_configs/noise_scheduler/better_SD_config.json
    "better_SD_config": dict(
        beta_schedule="scaled_linear", beta_end=0.015, beta_start=0.00001, clip_sample=False, thresholding=False,
        num_train_timesteps=3000, prediction_type="v_prediction", rescale_betas_zero_snr=True,
        timestep_spacing="trailing"),
}
