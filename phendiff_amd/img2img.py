"""DDIB class transfer: DDIM inversion under the original class -> denoising under the target class.

Mirrors ``src/utils_Img2Img.py``: ``_inversion`` (``:763-800``), ``_ddib`` (``:566-612``), the binary
class swap and batch sharding of ``perform_class_transfer_experiment`` (``:307-317,341-345``).

Two execution modes, same arithmetic:
  * eager  -- ``inversion`` / ``ddib`` walk the reference's Python loops over the drop-in objects;
  * graph  -- :class:`DDIBGraph` captures the whole 2*S-step trajectory (time/class embedding table, 2*S UNet
    evaluations, 2*S fused scheduler updates, post-processing) into ONE hipGraph and replays it per batch:
    no host round trip, no Python launch cost (~120 launches x 2*S per batch otherwise).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib as L
from .pipeline import ConditionalDDIMPipeline
from .schedulers import DDIMInverseScheduler


@torch.no_grad()
def inversion(pipe, input_images: torch.Tensor, class_labels: torch.Tensor, num_inference_steps: int,
              proc_idx: Optional[int] = None, variant: str = "0.18.2") -> torch.Tensor:
    """``_inversion`` (utils_Img2Img.py:763-800)."""
    gauss = input_images.clone().detach()
    inv = DDIMInverseScheduler.from_config(pipe.scheduler.config, variant=variant)
    inv.set_timesteps(num_inference_steps)
    for t in inv.timesteps:
        model_output = pipe.unet(gauss, t, class_labels).sample
        gauss = inv.step(model_output, t, gauss).prev_sample
    return gauss


@torch.no_grad()
def encode_to_latents(pipe, images: torch.Tensor, generator=None) -> torch.Tensor:
    """``_encode_to_latents`` (utils_Img2Img.py:827-836): ``vae.encode(x).latent_dist.sample() * scaling_factor`` (the scale
    is folded into the sampling kernel)."""
    return pipe.vae.encode(images).latent_dist.sample(generator, scale=float(pipe.vae.config.scaling_factor))


@torch.no_grad()
def decode_to_images(pipe, latents: torch.Tensor) -> torch.Tensor:
    """``_decode_to_images`` (utils_Img2Img.py:839-847)."""
    return pipe.vae.decode(latents / pipe.vae.config.scaling_factor, return_dict=False)[0]


@torch.no_grad()
def LDM_preprocess(pipe, images: torch.Tensor, class_labels_seq=None, generator=None):
    """``_LDM_preprocess`` (utils_Img2Img.py:803-824): images -> scaled latents, each label tensor -> (B, 77, D) embedding."""
    from .sd_pipeline import hack_class_embedding
    latents = encode_to_latents(pipe, images, generator)
    if class_labels_seq is None:
        return latents, None
    embeds = [hack_class_embedding(pipe._encode_class(class_labels=cl, device=latents.device, do_classifier_free_guidance=False))
              for cl in class_labels_seq]
    return latents, embeds


@torch.no_grad()
def ddib(pipe, clean_images, orig_class_labels, target_class_labels, num_inference_steps: int,
         process_idx: Optional[int] = None, variant: str = "0.18.2", output_type: str = "numpy", generator=None):
    """``_ddib`` (utils_Img2Img.py:566-612), both pipeline branches.  ``generator`` seeds the VAE posterior draw of the
    latent-diffusion branch (the reference draws it unseeded)."""
    from .sd_pipeline import CustomStableDiffusionImg2ImgPipeline
    if isinstance(pipe, CustomStableDiffusionImg2ImgPipeline):
        clean_images, [orig_class_cond] = LDM_preprocess(pipe, clean_images, [orig_class_labels], generator)
    elif isinstance(pipe, ConditionalDDIMPipeline):
        orig_class_cond = orig_class_labels
    else:
        raise NotImplementedError(type(pipe))
    inverted_gauss = inversion(pipe, clean_images, orig_class_cond, num_inference_steps, process_idx, variant)
    if isinstance(pipe, ConditionalDDIMPipeline):
        return pipe(class_labels=target_class_labels, w=0, num_inference_steps=num_inference_steps,
                    start_image=inverted_gauss, add_forward_noise_to_image=False, frac_diffusion_skipped=0,
                    output_type=output_type).images
    return pipe(image=inverted_gauss, class_labels=target_class_labels, strength=1, add_forward_noise_to_image=False,
                num_inference_steps=num_inference_steps, guidance_scale=0,     # guidance_scale <= 1.0 disables guidance
                output_type={"numpy": "np"}.get(output_type, output_type))


@torch.no_grad()
def inverted_regeneration(pipe, clean_images, orig_class_labels, num_inference_steps: int, **kw):
    """``"inverted_regeneration"`` (utils_Img2Img.py:374-384): DDIB with the original class as target."""
    return ddib(pipe, clean_images, orig_class_labels, orig_class_labels, num_inference_steps, **kw)


@torch.no_grad()
def classifier_free_guidance_forward_start(pipe, clean_images, target_class_labels, guidance_scale: float,
                                           frac_diffusion_skipped: float, num_inference_steps: int, generator=None,
                                           output_type: str = "numpy"):
    """``_classifier_free_guidance_forward_start`` (utils_Img2Img.py:615-648), both pipeline branches: noise
    the image up to ``(1 - frac_diffusion_skipped)`` of the trajectory, then denoise under the target class with
    classifier-free guidance."""
    from .sd_pipeline import CustomStableDiffusionImg2ImgPipeline
    if isinstance(pipe, CustomStableDiffusionImg2ImgPipeline):       # :638-645: strength = frac_diffusion_skipped
        return pipe(image=clean_images, class_labels=target_class_labels, strength=frac_diffusion_skipped,
                    num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, generator=generator,
                    output_type={"numpy": "np"}.get(output_type, output_type))
    return pipe(class_labels=target_class_labels, w=guidance_scale, num_inference_steps=num_inference_steps,
                start_image=clean_images, frac_diffusion_skipped=frac_diffusion_skipped, generator=generator,
                output_type=output_type).images


# fp16 engine: the factor the Lp loss gradient carries through the UNet backward (`custom_guided_generation`).  |d loss / d x0| of an
# Lp norm is <= 1 per element and ~ numel^-1/2 typically (2e-3 at 256 x 256 x 3): times 2^12 it sits mid-range of fp16 with 2^4 of
# head-room for what the backward adds; a step that overflows anyway halves it.
GUIDANCE_GRAD_SCALE = 4096.0


@torch.no_grad()
def custom_guided_generation(pipe, input_images: torch.Tensor, target_class_labels: torch.Tensor, p: float,
                             guidance_loss_scale: float, num_inference_steps: int, return_losses: bool = False):
    """``_custom_guided_generation`` (utils_Img2Img.py:699-760), both pipeline branches (latent diffusion: ``input_images`` are latents,
    ``target_class_labels`` the (B, 77, D) class embeddings, :663-671).  Per step: UNet forward
    (statistics kept), ``x0 = scheduler.step(...).pred_original_sample``, per-image ``Lp_loss(x0, input_images, p)``
    (``:245-270``), its gradient w.r.t. the image THROUGH the UNet (what ``torch.autograd.grad(losses_seq, images)`` returns:
    ``pd_lp_guidance`` -> input-gradient-only UNet backward), ``images -= guidance_loss_scale * grad``, then the scheduler
    step with the model output computed before the push.  No autograd graph exists: the backward is the HIP plan.

    fp16 engine (the reference runs this method under ``mixed_precision: fp16``, general_config.yaml:46 -- there autocast keeps the
    gradient in fp32 where it can; here every activation gradient is fp16): the Lp loss gradient enters the backward times a static
    power-of-two scale (:data:`GUIDANCE_GRAD_SCALE`, the role accelerate's GradScaler plays in training) and the image gradient is
    un-scaled before the ``guidance_loss_scale`` push; a step whose gradient is not finite halves the scale and is redone (one host
    read per step; the scale reached is kept for the following steps)."""
    from .sd_pipeline import CustomStableDiffusionImg2ImgPipeline
    ldm = isinstance(pipe, CustomStableDiffusionImg2ImgPipeline)
    if not ldm and not isinstance(pipe, ConditionalDDIMPipeline):
        raise NotImplementedError(type(pipe))
    if not input_images.is_cuda:
        raise L.PhenDiffHipError("phendiff_amd runs on MI355X only (no CPU fallback): move the inputs to 'cuda'")
    if isinstance(p, str) or not (1.0 <= float(p) < 1e6):
        raise NotImplementedError("Lp guidance: finite p >= 1 only (the reference config uses p = 2)")
    lib = L.lib()
    unet, sched = pipe.unet, pipe.scheduler
    dev = input_images.device
    B, _, H, W = input_images.shape
    st = torch.cuda.current_stream(dev).cuda_stream
    target = input_images.detach().contiguous().float()
    images = target.clone()
    if ldm:
        # latent-diffusion branch (:718-726): `target_class_labels` IS the (B, 77, D) class embedding `_LDM_preprocess` made, handed to
        # `pipe.unet(images, t, target_class_embeds)` as encoder_hidden_states; `images` are latents
        ehs = target_class_labels.to(device=dev, dtype=torch.float32).contiguous()
        if ehs.ndim != 3 or ehs.shape[0] != B:
            raise ValueError(f"latent-diffusion branch: target_class_labels must be the (B, tokens, D) class embeddings, got {tuple(ehs.shape)}")
        plan = unet.input_grad_plan(B, H, W, ehs.shape[1], dev)
        run_forward = lambda ts: plan.forward(images, ts, ehs, model_out, st)
    else:
        labels = target_class_labels.to(device=dev, dtype=torch.int64).contiguous()
        plan = unet.input_grad_plan(B, H, W, dev)
        run_forward = lambda ts: plan.forward(images, ts, labels, None, model_out, st)
    model_out, d_out, d_direct, pushed = (torch.empty_like(images) for _ in range(4))
    fp16 = getattr(unet, "compute_dtype", None) == "fp16"
    gscale = float(GUIDANCE_GRAD_SCALE) if fp16 else 1.0
    d_scaled = torch.empty_like(images) if fp16 else None
    splits = max(1, min(64, images[0].numel() // 4096))
    partial = torch.empty(B * splits, dtype=torch.float64, device=dev)
    losses = torch.empty(B, dtype=torch.float32, device=dev)
    all_losses = []
    c = sched.config
    sched.set_timesteps(num_inference_steps)
    for t in sched.timesteps:
        ts = torch.full((B,), float(t), dtype=torch.float32, device=dev)
        sa, sb, _, _, _ = sched.step_coefficients(t)
        a = L.LpGuidanceArgs(numel=images.numel(), per_sample=images[0].numel(), pred_type=L.PD_PRED[c.prediction_type],
                             clip=int(bool(c.clip_sample)), clip_range=float(c.clip_sample_range), sqrt_a=sa, sqrt_b=sb,
                             p=float(p), sample=images.data_ptr(), model_out=model_out.data_ptr(), target=target.data_ptr(),
                             partial=partial.data_ptr(), splits=splits, d_model_out=d_out.data_ptr(),
                             d_sample_direct=d_direct.data_ptr(), losses=losses.data_ptr())
        while True:
            run_forward(ts)
            L.check(lib.pd_lp_guidance(C.byref(a), st), "pd_lp_guidance")
            if not fp16:
                plan.backward(d_out, st)
                break
            torch.mul(d_out, gscale, out=d_scaled)
            plan.backward(d_scaled, st)
            if bool(torch.isfinite(plan.dsample).all()):       # (the one host read of an fp16 step)
                plan.dsample.mul_(1.0 / gscale)
                break
            gscale *= 0.5                                       # overflow somewhere in the fp16 chain: halve and redo the step
            if gscale < 2.0 ** -10:
                raise FloatingPointError("gradient-guided transfer (fp16): the UNet input gradient is not finite at any scale")
        g = L.GuidanceApplyArgs(numel=images.numel(), scale=float(guidance_loss_scale), x=images.data_ptr(),
                                g_direct=d_direct.data_ptr(), g_unet=plan.dsample.data_ptr(), out=pushed.data_ptr())
        L.check(lib.pd_guidance_apply(C.byref(g), st), "pd_guidance_apply")
        images = sched.step(model_out, t, pushed).prev_sample
        if return_losses:
            all_losses.append(losses.clone())
    return (images, all_losses) if return_losses else images


@torch.no_grad()
def linear_interp_custom_guidance_inverted_start(pipe, clean_images, orig_class_labels, target_class_labels, p: float,
                                                 guidance_loss_scale: float, num_inference_steps: int,
                                                 variant: str = "0.18.2", output_type: str = "numpy", generator=None):
    """``_linear_interp_custom_guidance_inverted_start`` (utils_Img2Img.py:651-696), both pipeline branches (``generator`` seeds the
    VAE posterior draw of the latent-diffusion branch; the reference draws it unseeded): inversion under the original class,
    then gradient-guided generation under the target class (``p`` / ``guidance_loss_scale``: the method's config keys,
    defaults 2 / 0.001).  ``output_type``: "pt" = the [-1, 1] tensor the reference hands to ``tensor_to_PIL``; "numpy" =
    NHWC float in [0, 1]; "pil"."""
    from .sd_pipeline import CustomStableDiffusionImg2ImgPipeline
    ldm = isinstance(pipe, CustomStableDiffusionImg2ImgPipeline)
    target_cond = target_class_labels
    if ldm:      # :663-671: images -> latents, both label tensors -> 77-token class embeddings
        clean_images, (orig_class_labels, target_cond) = LDM_preprocess(pipe, clean_images, [orig_class_labels, target_class_labels], generator)
    inverted = inversion(pipe, clean_images, orig_class_labels, num_inference_steps, None, variant)
    image = custom_guided_generation(pipe, inverted, target_cond, p, guidance_loss_scale, num_inference_steps)
    if ldm:      # :689-695: decode, then min-max renormalisation back to [-1, 1]
        image = decode_to_images(pipe, image)
        image = image - image.min()
        image = image / image.max()
        image = image * 2 - 1
    if output_type == "pt":
        return image
    arr = (image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
    return pipe.numpy_to_pil(arr) if output_type == "pil" else arr


@torch.no_grad()
def tensor_to_PIL(tensor: torch.Tensor, channel="mean"):
    """``tensor_to_PIL`` (utils_Img2Img.py:96-150): (N, 3, H, W) images in [-1, 1] -> RGB PIL images through ``pd_postproc``
    (``uint8 = round(255 * clamp(x/2 + 1/2, 0, 1))`` on the device); (N, 4, h, w) latents -> global min-max normalisation, then
    one channel or the channel mean as a greyscale image (visualisation only).  One image is returned bare, like the reference."""
    from PIL import Image
    assert tensor.ndim == 4, "Expecting a tensor of shape (N, C, H, W)"
    assert channel in ["mean"] + list(range(tensor.shape[1])), \
        f"Expecting a channel in {list(range(tensor.shape[1]))} or 'mean', got {channel}"
    if tensor.shape[1] == 4:
        img = tensor.detach().float().clone()
        img -= img.min()
        img /= img.max()
        img = img.clamp(0, 1)
        img = img[:, channel:channel + 1] if isinstance(channel, int) else img.mean(dim=1, keepdim=True)
        arr = (img.cpu().permute(0, 2, 3, 1).numpy() * 255).round().astype("uint8")
    else:
        assert float(tensor.min()) >= -1 and float(tensor.max()) <= 1, "Expecting values in [-1, 1]"
        x = tensor.detach().contiguous().float()
        if not x.is_cuda:
            raise L.PhenDiffHipError("phendiff_amd runs on MI355X only (no CPU fallback): move the tensor to 'cuda'")
        B, Cc, H, W = x.shape
        out = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=x.device)
        a = L.PostprocArgs(B=B, C=Cc, H=H, W=W, x=x.data_ptr(), out_f32=None, out_u8=out.data_ptr())
        L.check(L.lib().pd_postproc(C.byref(a), torch.cuda.current_stream(x.device).cuda_stream), "pd_postproc")
        arr = out.cpu().numpy()
    if arr.shape[-1] == 1:
        pil = [Image.fromarray(im.squeeze(-1), mode="L") for im in arr]
    else:
        pil = [Image.fromarray(im) for im in arr]
    return pil[0] if len(pil) == 1 else pil


def swap_binary_labels(orig_class_labels: torch.Tensor) -> torch.Tensor:
    """``target = 1 - orig`` (utils_Img2Img.py:343-344): strictly binary datasets."""
    return 1 - orig_class_labels


def shard_batches(num_batches: int, rank: int, world_size: int, even_batches: bool = True) -> List[int]:
    """Batch indices rank ``rank`` processes -- accelerate ``BatchSamplerShard`` semantics as reached through
    ``accelerator.prepare(dataloader)`` (utils_Img2Img.py:316): rank r takes batches r, r+G, r+2G, ...; with
    ``even_batches`` (accelerate's default) the tail is completed by wrapping around to the first batches so
    that every rank runs the same number of iterations.  No collective on the data path."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world_size")
    mine = list(range(rank, num_batches, world_size))
    if even_batches and num_batches > 0:
        # accelerate completes the last round with the samples of the first batches, in order: the k-th filler
        # batch is batch k (mod num_batches) and goes to the rank whose turn it is
        idx, k = num_batches, 0
        while idx % world_size != 0:
            if idx % world_size == rank:
                mine.append(k % num_batches)
            idx += 1
            k += 1
    return mine


class _ClassRows:
    """Per-(step, image) class conditioning of a captured trajectory, for every ``class_embed_type`` (cond_unet_2d.py:146-153,
    295-309).  The graph captures ONE ``pd_temb`` over all rows (+ one more for the class MLP of "timestep"); what changes between
    replays is the CONTENT of static buffers, filled by ``fill`` on the runner's stream before the replay:
      * nn.Embedding table (every shipped config) or no class conditioning: one int64 label per row;
      * "identity": the rows ARE the fp32 embedding vectors, [rows][time_embed_dim];
      * "timestep": the labels as fp32 go through the class MLP inside the graph (pre-allocated rows / scratch: a capture must not
        allocate), whose output rows the main ``pd_temb`` adds."""

    def __init__(self, plan, rows: int, device):
        self.plan, self.rows = plan, rows
        self.mode = getattr(plan.w, "class_mode", None)
        tdim = plan.m.time_embed_dim
        self.labels = torch.zeros((rows,), dtype=torch.int64, device=device) if self.mode is None else None
        self.emb = torch.zeros((rows, tdim), dtype=torch.float32, device=device) if self.mode in ("identity", "timestep") else None
        if self.mode == "timestep":
            self.vals = torch.zeros((rows,), dtype=torch.float32, device=device)
            self.scratch = torch.empty((rows, plan.w.proj_dim), dtype=torch.float32, device=device)

    def fill(self, sl: slice, steps: int, B: int, cond: torch.Tensor):
        """Rows ``sl`` (= ``steps`` x ``B``) <- the batch's conditioning repeated for every step."""
        dev = self.plan.device
        if self.mode is None:
            self.labels[sl].view(steps, B).copy_(cond.to(device=dev, dtype=torch.int64).view(1, B).expand(steps, B))
        elif self.mode == "identity":
            rows = cond.to(device=dev, dtype=torch.float32)
            if rows.numel() != B * self.emb.shape[1]:
                raise ValueError(f"class_embed_type='identity': pass the (B, time_embed_dim) embedding rows, got {tuple(cond.shape)}")
            self.emb[sl].view(steps, B, -1).copy_(rows.view(1, B, -1).expand(steps, B, -1))
        else:
            self.vals[sl].view(steps, B).copy_(cond.to(device=dev, dtype=torch.float32).view(1, B).expand(steps, B))

    def temb(self, ts_rows, st, out):
        """The launches that turn the rows into the [rows][proj_dim] projection table (captured)."""
        plan = self.plan
        if self.mode == "timestep":
            plan._class_rows_timestep(None, self.rows, st, emb=self.emb, scratch=self.scratch, vals=self.vals)
        plan.temb_rows(ts_rows, self.labels, self.emb, st, rows=self.rows, out=out)


class DDIBGraph:
    """One hipGraph for the whole invert -> class-swap -> denoise trajectory of a batch.

    ``run(images, orig_labels, target_labels)`` copies the batch into static buffers, replays the graph and
    returns the float NHWC ``[0,1]`` images on the device (``.images``), the inverted latents (``.inverted``)
    and optionally the uint8 quantisation."""

    def __init__(self, pipe: ConditionalDDIMPipeline, batch_size: int, num_inference_steps: int, height: int = None,
                 width: int = None, variant: str = "0.18.2", device=None, use_graph: bool = True, private_plan: bool = False):
        self.pipe = pipe
        unet = pipe.unet
        self.device = torch.device(device) if device is not None else unet.device
        ss = unet.config.sample_size
        H = height or (ss if isinstance(ss, int) else ss[0])
        W = width or (ss if isinstance(ss, int) else ss[1])
        self.B, self.S, self.H, self.W = batch_size, num_inference_steps, H, W
        B, S = self.B, self.S
        dev = self.device
        self.lib = L.lib()
        cin = unet.config.in_channels
        self._bounds = None
        mb = unet.max_batch(H, W)
        if B > mb:
            # One launch plan addresses each tensor with 32-bit byte offsets (< 2 GiB: 127 images of 256 x 256 x 64 bf16 channels), so a
            # larger batch (SURVEY 8(d) sweeps batch_size up to 128) is replayed as even slices -- the samples of a batch are independent
            # (utils_Img2Img.py:566-612: no cross-sample op anywhere on the path).  Slices of equal size share ONE captured graph.
            n_sl = -(-B // mb)
            step = -(-B // n_sl)
            self._bounds = [(b0, min(B, b0 + step)) for b0 in range(0, B, step)]
            self._runners = {}
            for b0, b1 in self._bounds:
                if b1 - b0 not in self._runners:
                    self._runners[b1 - b0] = DDIBGraph(pipe, b1 - b0, S, H, W, variant, dev, use_graph, private_plan)
            first = self._runners[self._bounds[0][1] - self._bounds[0][0]]
            self.plan, self.stream, self.inv_ts, self.gen_ts = first.plan, first.stream, first.inv_ts, first.gen_ts
            self.inverted = torch.empty((B, cin, H, W), dtype=torch.float32, device=dev)
            self.images = torch.empty((B, H, W, cin), dtype=torch.float32, device=dev)
            self.images_u8 = torch.empty((B, H, W, cin), dtype=torch.uint8, device=dev)
            self.graph = C.c_void_p(None)
            self.use_graph = use_graph
            return
        self.plan = unet.new_plan(B, H, W, dev) if private_plan else unet.plan_for(B, H, W, dev)
        # schedulers (host tables)
        self.inv = DDIMInverseScheduler.from_config(pipe.scheduler.config, variant=variant)
        self.inv.set_timesteps(S)
        fwd = pipe.scheduler
        fwd.set_timesteps(S)
        # pipeline __call__ with frac_diffusion_skipped=0: timesteps <= N*(1-0) => all of them (:250-258)
        gen_ts = fwd.timesteps[fwd.timesteps <= fwd.config.num_train_timesteps * (1 - 0)]
        self.inv_ts = [int(t) for t in self.inv.timesteps]
        self.gen_ts = [int(t) for t in gen_ts]
        nsteps = len(self.inv_ts) + len(self.gen_ts)
        # static device buffers
        self.x = torch.empty((B, cin, H, W), dtype=torch.float32, device=dev)
        self.model_out = torch.empty_like(self.x)
        self.inverted = torch.empty_like(self.x)
        self.images = torch.empty((B, H, W, cin), dtype=torch.float32, device=dev)
        self.images_u8 = torch.empty((B, H, W, cin), dtype=torch.uint8, device=dev)
        self.ts_rows = torch.empty((nsteps * B,), dtype=torch.float32, device=dev)
        self.class_rows = _ClassRows(self.plan, nsteps * B, dev)
        ts_host = torch.tensor(self.inv_ts + self.gen_ts, dtype=torch.float32).repeat_interleave(B)
        self.ts_rows.copy_(ts_host)
        self.temb = torch.empty((nsteps * B, self.plan.w.proj_dim), dtype=torch.float32, device=dev)
        # per-step scheduler args
        self.step_args = []
        for sched, ts in ((self.inv, self.inv_ts), (fwd, self.gen_ts)):
            c = sched.config
            for t in ts:
                sa, sb, sap, dirc, _ = sched.step_coefficients(t, 0.0)
                self.step_args.append(L.DdimStepArgs(
                    numel=self.x.numel(), per_sample=self.x[0].numel(), pred_type=L.PD_PRED[c.prediction_type],
                    clip=int(bool(c.clip_sample)), clip_range=float(c.clip_sample_range), use_clipped_model_output=0,
                    sqrt_a=sa, sqrt_b=sb, sqrt_ap=sap, dir_coef=dirc, sample=self.x.data_ptr(),
                    model_out=self.model_out.data_ptr(), uncond_out=None, w=None, w_per_sample=0, guidance_cfg=0,
                    prev_sample=self.x.data_ptr(), pred_x0=None))
        self.post_args = L.PostprocArgs(B=B, C=cin, H=H, W=W, x=self.x.data_ptr(), out_f32=self.images.data_ptr(),
                                        out_u8=self.images_u8.data_ptr())
        self.stream = torch.cuda.Stream(device=dev)
        self.graph = C.c_void_p(None)
        self.use_graph = use_graph
        if use_graph:
            self._capture()

    # the trajectory, as launches on `st`
    def _enqueue(self, st):
        lib, plan, B = self.lib, self.plan, self.B
        n_inv = len(self.inv_ts)
        self.class_rows.temb(self.ts_rows, st, self.temb)
        row_bytes = plan.w.proj_dim * 4
        for i, a in enumerate(self.step_args):
            plan.run(self.x.data_ptr(), self.temb.data_ptr() + i * B * row_bytes, self.model_out.data_ptr(), st)
            L.check(lib.pd_ddim_step(C.byref(a), st), "pd_ddim_step")
            if i == n_inv - 1:
                self._copy_inverted(st)
        L.check(lib.pd_postproc(C.byref(self.post_args), st), "pd_postproc")

    def _copy_inverted(self, st):
        # device-to-device snapshot of the inverted latents via the add_noise kernel: 1*x + 0*x
        a = L.AddNoiseArgs(numel=self.x.numel(), per_sample=self.x[0].numel(), velocity=0, x=self.x.data_ptr(),
                           noise=self.x.data_ptr(), sa=self._ones.data_ptr(), sb=self._zeros.data_ptr(),
                           out=self.inverted.data_ptr())
        L.check(self.lib.pd_add_noise(C.byref(a), st), "pd_add_noise")

    def _capture(self):
        st = self.stream.cuda_stream
        self._ones = torch.ones((self.B,), dtype=torch.float32, device=self.device)
        self._zeros = torch.zeros((self.B,), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        L.check(self.lib.pd_graph_begin(st), "pd_graph_begin")
        try:
            self._enqueue(st)
        finally:
            rc = self.lib.pd_graph_end(st, C.byref(self.graph))
        L.check(rc, "pd_graph_end")

    @torch.no_grad()
    def run(self, clean_images: torch.Tensor, orig_class_labels: torch.Tensor, target_class_labels: torch.Tensor,
            join: bool = True):
        """``join=False`` leaves the caller's stream un-joined (call ``self.join()`` before reading the outputs), so several
        runners can replay concurrently on their own streams."""
        B = self.B
        if self._bounds is not None:
            if tuple(clean_images.shape) != (B,) + tuple(self.inverted.shape[1:]):
                raise ValueError(f"expected images of shape {tuple(self.inverted.shape)}, got {tuple(clean_images.shape)}")
            for b0, b1 in self._bounds:
                r = self._runners[b1 - b0]
                r.run(clean_images[b0:b1], orig_class_labels[b0:b1], target_class_labels[b0:b1], join=False)
                with torch.cuda.stream(r.stream):      # stream-ordered before the runner's next replay overwrites its outputs
                    self.images[b0:b1].copy_(r.images, non_blocking=True)
                    self.images_u8[b0:b1].copy_(r.images_u8, non_blocking=True)
                    self.inverted[b0:b1].copy_(r.inverted, non_blocking=True)
            if join:
                self.join()
            return self
        if clean_images.shape != self.x.shape:
            raise ValueError(f"expected images of shape {tuple(self.x.shape)}, got {tuple(clean_images.shape)}")
        n_inv, n_gen = len(self.inv_ts), len(self.gen_ts)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.x.copy_(clean_images, non_blocking=True)
            self.class_rows.fill(slice(0, n_inv * B), n_inv, B, orig_class_labels)
            self.class_rows.fill(slice(n_inv * B, (n_inv + n_gen) * B), n_gen, B, target_class_labels)
            if self.use_graph:
                L.check(self.lib.pd_graph_launch(self.graph, self.stream.cuda_stream), "pd_graph_launch")
            else:
                if not hasattr(self, "_ones"):
                    self._ones = torch.ones((B,), dtype=torch.float32, device=self.device)
                    self._zeros = torch.zeros((B,), dtype=torch.float32, device=self.device)
                self._enqueue(self.stream.cuda_stream)
        if join:
            cur.wait_stream(self.stream)
        return self

    def join(self):
        if self._bounds is not None:
            for r in self._runners.values():
                r.join()
            return self
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return self

    def __del__(self):
        try:
            if self.graph:
                self.lib.pd_graph_destroy(self.graph)
        except Exception:
            pass


class SDDDIBGraph:
    """One hipGraph for the latent-diffusion DDIB transfer -- ``_ddib`` with a ``CustomStableDiffusionImg2ImgPipeline``
    (utils_Img2Img.py:566-612; custom_pipeline_stable_diffusion_img2img.py:667-711): VAE encode -> posterior sample (x
    ``scaling_factor``) -> S inversion steps under the original class -> class swap -> S denoising steps -> VAE decode ->
    ``(x / 2 + .5).clamp(0, 1)`` NHWC.  ~360 launches x 2S steps are captured once and replayed per batch; the result is the eager
    ``ddib(pipe, ...)`` bit for bit.  ``run(images, orig_labels, target_labels, generator=None, noise=None)``: the posterior noise is an
    input (drawn from ``generator`` outside the graph, like ``latent_dist.sample(generator)``)."""

    def __init__(self, pipe, batch_size: int, num_inference_steps: int, height: int, width: int, variant: str = "0.18.2",
                 device=None, use_graph: bool = True):
        from .schedulers import DDIMInverseScheduler as Inv
        self.pipe = pipe
        unet, vae = pipe.unet, pipe.vae
        self.device = dev = torch.device(device) if device is not None else unet.device
        self.B, self.S, self.H, self.W = B, S, H, W = batch_size, num_inference_steps, height, width
        self.lib = L.lib()
        sf = 1 << (len(vae.config.block_out_channels) - 1)
        h, w = H // sf, W // sf
        lc = vae.config.latent_channels
        # the VAE runs in sub-batches of what one of its launch plans addresses (32-bit byte offsets), like vae.encode / .decode
        def chunks(kind, ph, pw):
            step = vae._max_batch(H, W, kind)
            return [(b0, min(step, B - b0), vae._plan(kind, min(step, B - b0), ph, pw, dev)) for b0 in range(0, B, step)]
        self.enc_chunks, self.dec_chunks = chunks("enc", H, W), chunks("dec", h, w)
        self.plan = unet.plan_for(B, h, w, 77, dev)
        self.inv = Inv.from_config(pipe.scheduler.config, variant=variant)
        self.inv.set_timesteps(S)
        fwd = pipe.scheduler
        fwd.set_timesteps(S, device=dev)
        gen_ts, _ = pipe.get_timesteps(S, 1.0, dev)                     # strength = 1: all S steps (custom_pipeline...:375-382)
        self.inv_ts, self.gen_ts = [int(t) for t in self.inv.timesteps], [int(t) for t in gen_ts]
        nsteps = len(self.inv_ts) + len(self.gen_ts)
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        self.x, self.moments, self.noise = f32(B, 3, H, W), f32(B, 2 * lc, h, w), f32(B, lc, h, w)
        self.latents, self.model_out, self.inverted, self.dec_in = f32(B, lc, h, w), f32(B, lc, h, w), f32(B, lc, h, w), f32(B, lc, h, w)
        self.decoded, self.images = f32(B, 3, H, W), f32(B, H, W, 3)
        D = unet.config.cross_attention_dim
        self.ehs_orig, self.ehs_target = f32(B, 77, D), f32(B, 77, D)
        self.ts_rows = torch.tensor(self.inv_ts + self.gen_ts, dtype=torch.float32).repeat_interleave(B).to(dev)
        self.scaling = float(vae.config.scaling_factor)
        self.step_args = []
        for sched, ts in ((self.inv, self.inv_ts), (fwd, self.gen_ts)):
            c = sched.config
            for t in ts:
                sa, sb, sap, dirc, _ = sched.step_coefficients(t, 0.0)
                self.step_args.append(L.DdimStepArgs(
                    numel=self.latents.numel(), per_sample=self.latents[0].numel(), pred_type=L.PD_PRED[c.prediction_type],
                    clip=int(bool(c.clip_sample)), clip_range=float(c.clip_sample_range), use_clipped_model_output=0,
                    sqrt_a=sa, sqrt_b=sb, sqrt_ap=sap, dir_coef=dirc, sample=self.latents.data_ptr(),
                    model_out=self.model_out.data_ptr(), uncond_out=None, w=None, w_per_sample=0, guidance_cfg=0,
                    prev_sample=self.latents.data_ptr(), pred_x0=None))
        self.sample_args = L.LatentSampleArgs(B=B, C=lc, HW=h * w, scale=self.scaling, moments=self.moments.data_ptr(),
                                              noise=self.noise.data_ptr(), out=self.latents.data_ptr())
        self.post_args = L.PostprocArgs(B=B, C=3, H=H, W=W, x=self.decoded.data_ptr(), out_f32=self.images.data_ptr(), out_u8=None)
        self.stream = torch.cuda.Stream(device=dev)
        self.graph = C.c_void_p(None)
        self.use_graph = use_graph
        if use_graph:
            torch.cuda.synchronize(dev)
            L.check(self.lib.pd_graph_begin(self.stream.cuda_stream), "pd_graph_begin")
            try:
                with torch.cuda.stream(self.stream):
                    self._enqueue(self.stream.cuda_stream)
            finally:
                rc = self.lib.pd_graph_end(self.stream.cuda_stream, C.byref(self.graph))
            L.check(rc, "pd_graph_end")

    def _enqueue(self, st):
        """The trajectory as launches on the current stream (``st``); the three torch calls are plain device-to-device kernels
        between pre-allocated buffers (capturable)."""
        lib, plan, B = self.lib, self.plan, self.B
        for b0, nb, plan_ in self.enc_chunks:
            plan_.run(self.x[b0:b0 + nb].data_ptr(), self.moments[b0:b0 + nb].data_ptr(), st)
        L.check(lib.pd_latent_sample(C.byref(self.sample_args), st), "pd_latent_sample")
        n_inv = len(self.inv_ts)
        ta = plan.temb_args
        for i, a in enumerate(self.step_args):
            if i == 0 or i == n_inv:       # class conditioning of the phase: (B, 77, D) fp32 -> the plan's compute-dtype buffer
                plan.ehs.view(B, 77, -1).copy_(self.ehs_orig if i == 0 else self.ehs_target)
            ta.rows = B
            ta.timesteps, ta.labels, ta.class_emb = self.ts_rows.data_ptr() + 4 * i * B, None, None
            ta.emb, ta.proj = plan.temb_emb.data_ptr(), plan.temb_table.data_ptr()
            L.check(lib.pd_temb(C.byref(ta), st), "pd_temb")
            # the cross-attention k / v projections only on the first step of a phase (the context is the phase's class): round 6
            plan.run(self.latents.data_ptr(), plan.temb_table.data_ptr(), self.model_out.data_ptr(), st, context=(i == 0 or i == n_inv))
            L.check(lib.pd_ddim_step(C.byref(a), st), "pd_ddim_step")
            if i == n_inv - 1:
                self.inverted.copy_(self.latents)
        torch.div(self.latents, self.scaling, out=self.dec_in)        # vae.decode(latents / scaling_factor), custom_pipeline...:709
        for b0, nb, plan_ in self.dec_chunks:
            plan_.run(self.dec_in[b0:b0 + nb].data_ptr(), self.decoded[b0:b0 + nb].data_ptr(), st)
        L.check(lib.pd_postproc(C.byref(self.post_args), st), "pd_postproc")

    @torch.no_grad()
    def run(self, clean_images, orig_class_labels, target_class_labels, generator=None, noise=None):
        from .schedulers import randn_tensor
        from .sd_pipeline import hack_class_embedding
        if clean_images.shape != self.x.shape:
            raise ValueError(f"expected images of shape {tuple(self.x.shape)}, got {tuple(clean_images.shape)}")
        pipe, dev = self.pipe, self.device
        cur = torch.cuda.current_stream(dev)
        if noise is None:
            noise = randn_tensor(tuple(self.noise.shape), generator, dev)
        eo = hack_class_embedding(pipe._encode_class(class_labels=orig_class_labels, device=dev, do_classifier_free_guidance=False))
        et = hack_class_embedding(pipe._encode_class(class_labels=target_class_labels, device=dev, do_classifier_free_guidance=False))
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.x.copy_(clean_images, non_blocking=True)
            self.noise.copy_(noise, non_blocking=True)
            self.ehs_orig.copy_(eo)
            self.ehs_target.copy_(et)
            if self.use_graph:
                L.check(self.lib.pd_graph_launch(self.graph, self.stream.cuda_stream), "pd_graph_launch")
            else:
                self._enqueue(self.stream.cuda_stream)
        cur.wait_stream(self.stream)
        return self

    def __del__(self):
        try:
            if self.graph:
                self.lib.pd_graph_destroy(self.graph)
        except Exception:
            pass


class CFGForwardStartGraph:
    """hipGraph form of the CFG forward-start transfer (utils_Img2Img.py:615-648 +
    pipeline_conditionial_ddim.py:248-347): ``add_noise`` to the first kept timestep, then per step a conditional
    and an unconditional UNet evaluation (``class_emb = 0``, pipeline :310-317) feeding ONE fused
    guidance-combine + DDIM update; post-processing; all captured once and replayed per batch.

    ``run(clean_images, target_labels, noise)``: the forward noise is an input (the reference draws it from the
    caller's generator, ``randn_tensor``), so results are reproducible against the eager pipeline / the oracle."""

    def __init__(self, pipe: ConditionalDDIMPipeline, batch_size: int, num_inference_steps: int, guidance_scale: float = 2.5,
                 frac_diffusion_skipped: float = 0.5, guidance_eqn: str = "imagen", height: int = None, width: int = None,
                 device=None, use_graph: bool = True):
        if guidance_eqn not in ("imagen", "CFG"):
            raise ValueError(f"Unknown guidance equation '{guidance_eqn}'; should be 'imagen' or 'CFG'")
        unet = pipe.unet
        self.device = dev = torch.device(device) if device is not None else unet.device
        ss = unet.config.sample_size
        H = height or (ss if isinstance(ss, int) else ss[0])
        W = width or (ss if isinstance(ss, int) else ss[1])
        self.B, self.S = B, S = batch_size, num_inference_steps
        self.lib = L.lib()
        if B > unet.max_batch(H, W):
            raise ValueError(f"batch_size {B} exceeds what one launch plan holds at {H}x{W} ({unet.max_batch(H, W)} images)")
        self.plan = unet.plan_for(B, H, W, dev)
        cin = unet.config.in_channels
        sch = pipe.scheduler
        sch.set_timesteps(S)
        init_t = sch.config.num_train_timesteps * (1 - frac_diffusion_skipped)   # pipeline :252-258
        self.ts = [int(t) for t in sch.timesteps[sch.timesteps <= init_t]]
        n = len(self.ts)
        w = guidance_scale
        self.do_cfg = (guidance_eqn == "imagen" and w > 1) or (guidance_eqn == "CFG" and w > 0)   # :272-284
        self.x = torch.empty((B, cin, H, W), dtype=torch.float32, device=dev)
        self.clean = torch.empty_like(self.x)
        self.noise = torch.empty_like(self.x)
        self.cond_out = torch.empty_like(self.x)
        self.uncond_out = torch.empty_like(self.x)
        self.images = torch.empty((B, H, W, cin), dtype=torch.float32, device=dev)
        self.images_u8 = torch.empty((B, H, W, cin), dtype=torch.uint8, device=dev)
        self.ts_rows = torch.tensor(self.ts, dtype=torch.float32).repeat_interleave(B).to(dev)
        self.class_rows = _ClassRows(self.plan, n * B, dev)
        self.zero_emb = torch.zeros((n * B, unet.time_embed_dim), dtype=torch.float32, device=dev)
        pd = self.plan.w.proj_dim
        self.temb_c = torch.empty((n * B, pd), dtype=torch.float32, device=dev)
        self.temb_u = torch.empty((n * B, pd), dtype=torch.float32, device=dev)
        self.w_dev = torch.tensor([float(w)], dtype=torch.float32, device=dev)
        sa, sb = sch._per_sample_coefs(torch.tensor([self.ts[0]] * B), dev)
        self._sa, self._sb = sa, sb
        self.noise_args = L.AddNoiseArgs(numel=self.x.numel(), per_sample=self.x[0].numel(), velocity=0,
                                         x=self.clean.data_ptr(), noise=self.noise.data_ptr(), sa=sa.data_ptr(),
                                         sb=sb.data_ptr(), out=self.x.data_ptr())
        c = sch.config
        self.step_args = []
        for t in self.ts:
            sa_, sb_, sap, dirc, _ = sch.step_coefficients(t, 0.0)
            self.step_args.append(L.DdimStepArgs(
                numel=self.x.numel(), per_sample=self.x[0].numel(), pred_type=L.PD_PRED[c.prediction_type],
                clip=int(bool(c.clip_sample)), clip_range=float(c.clip_sample_range), use_clipped_model_output=0,
                sqrt_a=sa_, sqrt_b=sb_, sqrt_ap=sap, dir_coef=dirc, sample=self.x.data_ptr(), model_out=self.cond_out.data_ptr(),
                uncond_out=(self.uncond_out.data_ptr() if self.do_cfg else None), w=self.w_dev.data_ptr(), w_per_sample=0,
                guidance_cfg=int(guidance_eqn == "CFG"), prev_sample=self.x.data_ptr(), pred_x0=None))
        self.post_args = L.PostprocArgs(B=B, C=cin, H=H, W=W, x=self.x.data_ptr(), out_f32=self.images.data_ptr(),
                                        out_u8=self.images_u8.data_ptr())
        self.stream = torch.cuda.Stream(device=dev)
        self.graph = C.c_void_p(None)
        self.use_graph = use_graph
        if use_graph:
            torch.cuda.synchronize(dev)
            st = self.stream.cuda_stream
            L.check(self.lib.pd_graph_begin(st), "pd_graph_begin")
            try:
                self._enqueue(st)
            finally:
                rc = self.lib.pd_graph_end(st, C.byref(self.graph))
            L.check(rc, "pd_graph_end")

    def _enqueue(self, st):
        lib, plan, B = self.lib, self.plan, self.B
        rows = self.ts_rows.numel()
        self.class_rows.temb(self.ts_rows, st, self.temb_c)
        if self.do_cfg:
            plan.temb_rows(self.ts_rows, None, self.zero_emb, st, rows=rows, out=self.temb_u)
        L.check(lib.pd_add_noise(C.byref(self.noise_args), st), "pd_add_noise")
        row_bytes = plan.w.proj_dim * 4
        for i, a in enumerate(self.step_args):
            plan.run(self.x.data_ptr(), self.temb_c.data_ptr() + i * B * row_bytes, self.cond_out.data_ptr(), st)
            if self.do_cfg:
                plan.run(self.x.data_ptr(), self.temb_u.data_ptr() + i * B * row_bytes, self.uncond_out.data_ptr(), st)
            L.check(lib.pd_ddim_step(C.byref(a), st), "pd_ddim_step")
        L.check(lib.pd_postproc(C.byref(self.post_args), st), "pd_postproc")

    @torch.no_grad()
    def run(self, clean_images: torch.Tensor, target_class_labels: torch.Tensor, noise: torch.Tensor):
        n, B = len(self.ts), self.B
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.clean.copy_(clean_images, non_blocking=True)
            self.noise.copy_(noise, non_blocking=True)
            self.class_rows.fill(slice(0, n * B), n, B, target_class_labels)
            if self.use_graph:
                L.check(self.lib.pd_graph_launch(self.graph, self.stream.cuda_stream), "pd_graph_launch")
            else:
                self._enqueue(self.stream.cuda_stream)
        cur.wait_stream(self.stream)
        return self

    def __del__(self):
        try:
            if self.graph:
                self.lib.pd_graph_destroy(self.graph)
        except Exception:
            pass
