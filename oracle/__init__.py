"""CPU oracle for the PhenDiff diffusion hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) restatement of the arithmetic that
the reference executes on its hot path (class-conditional UNet forward, DDIM /
inverse-DDIM scheduler updates, the conditional DDIM pipeline and the DDIB
invert -> class-swap -> denoise loop).  It exists so that the HIP kernels in
``phendiff_amd`` can be checked against an independent implementation.

**Parity unpinned.**  The reference (`/root/reference`) is pure-Python glue over
``diffusers==0.18.2`` (``environment.yaml:80``), which is neither vendored in the
reference tree nor installable here (no network), and the reference ships no
tests, golden vectors or fixtures for this path.  The restatement therefore
follows the reference's own call sites plus the published diffusers-0.18.2
algorithms (recorded in SURVEY.md Appendix A); it is pinned only by
known-answer checks (parameter counts of public checkpoints, closed-form
scheduler tables) -- see ``tests/test_oracle_*.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  Nothing under ``phendiff_amd/`` does.
"""

from .schedulers_ref import DDIMSchedulerRef, DDIMInverseSchedulerRef  # noqa: F401
from .unet_ref import CondUNet2DRef, UNET_CONFIGS  # noqa: F401
from .sd_unet_ref import (  # noqa: F401
    UNet2DConditionRef, CustomEmbeddingRef, SD21_UNET_CONFIG, class_emb_to_encoder_hidden_states)
from .vae_ref import AutoencoderKLRef, SD_VAE_CONFIG, vae_preprocess_ref, vae_postprocess_ref  # noqa: F401
from .sd_pipeline_ref import (  # noqa: F401
    SDImg2ImgPipelineRef, hack_class_embedding_ref, encode_to_latents_ref, sd_inversion_ref, sd_ddib_ref, sd_cfg_forward_start_ref,
    sd_linear_interp_custom_guidance_inverted_start_ref)
from .training_ref import TrainingLoopRef, TINY_CONFIG0_UNET, synthetic_two_class_batch, cosine_lr_lambda, ema_decay_ref  # noqa: F401
from .pipeline_ref import (  # noqa: F401
    ConditionalDDIMPipelineRef,
    inversion_ref,
    ddib_ref,
    numpy_to_uint8,
    lp_loss_ref,
    custom_guided_generation_ref,
    linear_interp_custom_guidance_inverted_start_ref,
    tensor_to_uint8_ref,
)
from .eval_generation_ref import (  # noqa: F401
    EMASwapRef, eval_batch_sizes_ref, eval_generation_ddim_ref, eval_generation_sd_ref, latents_preview_ref, split_ref)
from .inception_ref import (  # noqa: F401
    InceptionV3FeaturesRef, randomize_inception_, tf1_bilinear_resize_ref, fid_statistics_ref, fid_from_statistics_ref, isc_ref, kid_ref,
    calculate_metrics_ref,
)
