"""phendiff_amd -- MI355X-native engine for PhenDiff's diffusion hot path.

Drop-in (Python-level) replacements for the objects the reference's loops drive:
``CustomCondUNet2DModel`` (``src/cond_unet_2d``), ``DDIMScheduler`` / ``DDIMInverseScheduler`` (diffusers),
``ConditionalDDIMPipeline`` (``src/pipeline_conditional_ddim``), and the DDIB class-transfer loops of
``src/utils_Img2Img.py``.  All device work goes through ``libphendiff_hip.so`` (``include/phendiff_hip.h``).
"""
from ._lib import PhenDiffHipError, lib  # noqa: F401
from .unet import CustomCondUNet2DModel, UNet2DOutput  # noqa: F401
from .schedulers import DDIMScheduler, DDIMInverseScheduler  # noqa: F401
from .pipeline import ConditionalDDIMPipeline, ImagePipelineOutput  # noqa: F401
from .img2img import (inversion, ddib, inverted_regeneration, classifier_free_guidance_forward_start, DDIBGraph,  # noqa: F401
                      CFGForwardStartGraph, SDDDIBGraph, shard_batches, swap_binary_labels, custom_guided_generation,
                      linear_interp_custom_guidance_inverted_start, encode_to_latents, decode_to_images, LDM_preprocess, tensor_to_PIL)
from .configs import UNET_CONFIGS, SCHEDULER_CONFIGS  # noqa: F401
from . import configs  # noqa: F401
from . import diagnostics  # noqa: F401
from .comm import NativeComm  # noqa: F401
from . import training  # noqa: F401
from .unet_train import UNetTrainer, UNetTrainPlan, training_param_order  # noqa: F401
from . import train_state  # noqa: F401
from . import eval_generation  # noqa: F401
from .sd_unet import SDUNet2DConditionModel, CustomEmbedding, class_emb_to_encoder_hidden_states, SD21_UNET_CONFIG  # noqa: F401
from .vae import AutoencoderKL, VaeImageProcessor, DiagonalGaussianDistribution, SD_VAE_CONFIG  # noqa: F401
from .sd_pipeline import CustomStableDiffusionImg2ImgPipeline, hack_class_embedding  # noqa: F401
from .sd_unet_train import SDUNetTrainer, SDUNetTrainPlan, sd_training_param_order  # noqa: F401
from . import metrics  # noqa: F401,E402  (FID / IS / KID: InceptionV3Features, calculate_metrics, class_metrics_hook)
