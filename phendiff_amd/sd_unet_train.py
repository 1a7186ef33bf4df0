"""Training step of the SD-2.1 ``UNet2DConditionModel`` + ``CustomEmbedding`` on MI355X (BASELINE
configs[3]): what ``_SD_prediction_wrapper`` + ``_diffusion_and_backward`` (``utils_training.py:459-496,371-456``) ask autograd
to do, as a static plan of HIP launches on the engine of :mod:`phendiff_amd.unet_train`.

``SDUNetTrainPlan`` = the forward of :class:`phendiff_amd.sd_unet.SDUNetPlan` (statistics / log-sum-exp kept) + the backward of
its block tape.  ResnetBlock2D / sampling convs / conv_out / the time-embedding path reuse the pixel-space UNet's emitters; a
``Transformer2DModel`` block adds, in reverse order of its forward: every ``nn.Linear`` as a 1x1 convolution (input gradient =
``pd_conv`` with the transposed weights, weight gradient = ``pd_conv_wgrad``), ``pd_geglu_bwd``, ``pd_layernorm_bwd`` (which also
folds the skip-connection gradient in), ``pd_attn_d64_bwd`` for the self attention and the 77-token cross attention, and the
gradient of the class token (``pd_token_embedding_grad``) -- the only trainable part of ``encoder_hidden_states``.
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .packing import dgrad_weight, pack_conv_weight, upsample_phase_weights_stacked
from .sd_unet import (CustomEmbedding, SDUNet2DConditionModel, SDUNetPlan, _SDPackedWeights, _Transformer2D,
                      class_emb_to_encoder_hidden_states)
from .unet import _Resnet, _Sampler, _copy_into
from .unet_train import UNetTrainer, UNetTrainPlan, _contiguous_after, run_pack_jobs

# diagnostic switch (same-box A/B): PD_BIAS_FUSE=0 -> every bias gradient of the transformer blocks is a pd_channel_sum pass over its dY again
_BIAS_FUSE = os.environ.get("PD_BIAS_FUSE", "1") != "0"

EMB_NAME = "class_embedding.inner_module.weight"


def sd_training_param_order(m: SDUNet2DConditionModel) -> List[Tuple[str, torch.nn.Parameter]]:
    """(name, parameter) pairs in the order of the flat training buffers: all ``time_emb_proj`` weights (then biases) stacked
    -- the one [proj_dim][tdim] matrix ``pd_temb`` sees --, each transformer's attn1 to_q/to_k/to_v and attn2 to_k/to_v weights
    adjacent (the fused projections), then everything else."""
    named = dict(m.named_parameters())
    out, seen = [], set()

    def take(n):
        out.append((n, named[n]))
        seen.add(n)

    res = [n for n, mod in m.named_modules() if isinstance(mod, _Resnet)]
    for suffix in ("weight", "bias"):
        for n in res:
            take(f"{n}.time_emb_proj.{suffix}")
    for n, mod in m.named_modules():
        if isinstance(mod, _Transformer2D):
            b = f"{n}.transformer_blocks.0"
            for which in ("to_q", "to_k", "to_v"):
                take(f"{b}.attn1.{which}.weight")
            for which in ("to_k", "to_v"):
                take(f"{b}.attn2.{which}.weight")
    for n in named:
        if n not in seen:
            take(n)
    return out


class SDTrainWeights:
    """Input-gradient weights in ``pd_conv``'s packed layout (W'[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx])."""

    def __init__(self, m: SDUNet2DConditionModel, device, tdt):
        # fp16 (round 5): `--mixed_precision fp16` fine-tuning (args_parser.py:381-390) under training.LossScaler, set up by SDUNetTrainer
        self.tdt, self.device = tdt, device
        pk = lambda w, cp=None: pack_conv_weight(dgrad_weight(w.detach().to(device=device, dtype=torch.float32)), tdt, cp)
        lin = lambda w: w.detach()[:, :, None, None]
        self.resnets, self.transformers, self.samplers = {}, {}, {}
        for name, mod in m.named_modules():
            if isinstance(mod, _Resnet):
                e = SimpleNamespace(w1d=pk(mod.conv1.weight), w2d=pk(mod.conv2.weight))
                if mod.conv_shortcut is not None:
                    e.wsd = pk(mod.conv_shortcut.weight)
                self.resnets[name] = e
            elif isinstance(mod, _Transformer2D):
                blk = mod.transformer_blocks[0]
                a1, a2 = blk.attn1, blk.attn2
                self.transformers[name] = SimpleNamespace(
                    w_in_d=pk(lin(mod.proj_in.weight)), w_out_d=pk(lin(mod.proj_out.weight)),
                    wqkv1_d=pk(lin(torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], 0))),
                    wo1_d=pk(lin(a1.to_out[0].weight)), wq2_d=pk(lin(a2.to_q.weight)),
                    wkv2_d=pk(lin(torch.cat([a2.to_k.weight, a2.to_v.weight], 0))), wo2_d=pk(lin(a2.to_out[0].weight)),
                    wff1_d=pk(lin(blk.ff.net[0].proj.weight)), wff2_d=pk(lin(blk.ff.net[2].weight)))
            elif isinstance(mod, _Sampler):
                self.samplers[name] = SimpleNamespace(wd=pk(mod.conv.weight))
                if ".upsamplers." in name:      # input-gradient weights of the four sub-pixel phases (UNetTrainPlan's "up" backward, pd_conv phase_in)
                    k4 = upsample_phase_weights_stacked(mod.conv.weight.detach().to(device=device, dtype=torch.float32))
                    self.samplers[name].wd4 = tuple(pk(k4[p]) for p in range(4))
        co = m.conv_out.weight.shape[0]
        wo = torch.zeros((((co + 31) // 32) * 32,) + tuple(m.conv_out.weight.shape[1:]), dtype=torch.float32, device=device)
        wo[:co] = m.conv_out.weight.detach().to(device=device, dtype=torch.float32)
        self.conv_out_d = pk(wo)
        # conv_in's input gradient (the latent gradient of the gradient-guided transfer): block_out_channels[0] -> 4 channels (pad 32)
        self.conv_in_d = pk(m.conv_in.weight, 32)


class SDUNetTrainPlan(UNetTrainPlan, SDUNetPlan):
    """Forward (with saved statistics) + backward launch plan of the SD UNet for a fixed (B, H, W, context tokens).
    ``params`` / ``grads``: state_dict-name -> fp32 device tensor, laid out as :func:`sd_training_param_order` prescribes, plus
    ``EMB_NAME`` -> the ``CustomEmbedding`` table (optional).  ``backward`` ACCUMULATES into ``grads``."""

    def __init__(self, m: SDUNet2DConditionModel, w: _SDPackedWeights, tw: SDTrainWeights, B, H, W, tokens, device,
                 params: Optional[Dict[str, torch.Tensor]] = None, grads: Optional[Dict[str, torch.Tensor]] = None, frozen=(),
                 input_grad: bool = False):
        """``input_grad``: also produce d loss / d latents (fp32 NCHW, ``self.dsample``) -- the gradient-guided transfer's
        ``torch.autograd.grad(losses_seq, images)`` through the UNet (utils_Img2Img.py:718-745, latent-diffusion branch)."""
        self.train = True
        SDUNetPlan.__init__(self, m, w, B, H, W, tokens, device)
        self._init_train(tw, params, grads, input_grad, frozen)

    # ---- layout checks -----------------------------------------------------------------------------------------------
    def _check_layout(self):
        for d in (self.params, self.grads):
            res = [n for n, mod in self.m.named_modules() if isinstance(mod, _Resnet)]
            for suffix in ("weight", "bias"):
                for a, b in zip(res[:-1], res[1:]):
                    if not _contiguous_after(d[f"{a}.time_emb_proj.{suffix}"], d[f"{b}.time_emb_proj.{suffix}"]):
                        raise ValueError("time_emb_proj parameters must be stacked contiguously (use sd_training_param_order)")
            for n, mod in self.m.named_modules():
                if isinstance(mod, _Transformer2D):
                    b = f"{n}.transformer_blocks.0"
                    q, k, v = (d[f"{b}.attn1.{x}.weight"] for x in ("to_q", "to_k", "to_v"))
                    k2, v2 = (d[f"{b}.attn2.{x}.weight"] for x in ("to_k", "to_v"))
                    if not (_contiguous_after(q, k) and _contiguous_after(k, v) and _contiguous_after(k2, v2)):
                        raise ValueError("attn1 to_q/to_k/to_v and attn2 to_k/to_v weights must be adjacent (use sd_training_param_order)")
            for t in d.values():
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != torch.device(self.device):
                    raise ValueError("training parameters / gradients must be contiguous fp32 tensors on the plan's device")

    def _zero_bias_len(self):
        return 8 * max(self.m.config.block_out_channels) + 64

    # ---- forward -----------------------------------------------------------------------------------------------------
    def forward(self, sample: torch.Tensor, timesteps: torch.Tensor, ehs: torch.Tensor, out: torch.Tensor, stream,
                labels: Optional[torch.Tensor] = None):
        """One training forward: fp32 NCHW latents + (B,) timesteps + (B, tokens, D) context -> fp32 NCHW prediction.
        ``labels``: the class ids whose embedding is token 0 of ``ehs`` (None on an unconditional step)."""
        self.ehs.view(self.B, self.tokens, -1).copy_(ehs)
        a = self.temb_args
        a.rows = self.B
        a.timesteps, a.labels, a.class_emb = timesteps.data_ptr(), None, None
        a.emb, a.proj = self.t_emb.data_ptr(), self.temb_table.data_ptr()
        a.feat, a.z1 = self.t_feat.data_ptr(), self.t_z1.data_ptr()
        L.check(self.lib.pd_temb(C.byref(a), stream), "pd_temb")
        a.feat, a.z1 = None, None
        self.run(sample.data_ptr(), self.temb_table.data_ptr(), out.data_ptr(), stream)
        self._labels = None
        if self._token_grad_args is not None:
            self._token_grad_args.labels = L.ptr(labels)
        self.keepalive = (sample, timesteps, ehs, out, labels)

    # ---- backward emitters -------------------------------------------------------------------------------------------
    def _build_backward(self):
        self._dehs = [torch.empty_like(self.ehs), False]
        self.bufs.append(self._dehs[0])
        self._token_grad_args = None
        super()._build_backward()
        if self._want_ehs_grad():
            dehs, dt = self._dehs[0], self.grads[EMB_NAME]
            a = L.TokenEmbeddingGradArgs(dtype=self.code, rows=self.B, dim=dt.shape[1], num_classes=dt.shape[0],
                                         row_stride=self.tokens * dt.shape[1], labels=None, d=dehs.data_ptr(),
                                         dtable=self._G(EMB_NAME).data_ptr())
            self._token_grad_args = a
            self._b(self.lib.pd_token_embedding_grad, a, "token_embedding_grad")

    def _bwd_record(self, rec):
        if rec.kind == "transformer":
            self._transformer_bwd(rec)
        elif rec.kind == "sd_conv_in":
            if self.input_grad:
                dout = self._g(rec.out)[0]
                ops, self.ops = self.ops, self.bwd_ops
                try:
                    self._conv(dout, None, self.tw.conv_in_d, self._zero_bias, self.m.config.in_channels, out_mode=L.PD_OUT_NCHW_F32,
                               cout_pad=32, y=self.dsample, stats=False)
                finally:
                    self.ops = ops
                self.bwd_ops[-1].what = "dgrad3x3"
            if not self.param_grads:
                return
            dout = self._g(rec.out)[0]
            self._bias_grad(dout, self._G("conv_in.bias"))
            self._wgrad(rec.x, None, None, 0, dout, self._G("conv_in.weight"), cin_valid=self.m.config.in_channels)
        else:
            super()._bwd_record(rec)

    def _ln_bwd(self, x, ln, pname, dy, res, tag, bias_name=None):
        """LayerNorm backward (+ the gradient arriving over the skip connection).  ``bias_name``: the bias of the Linear layer whose output gradient the
        returned dx is -- its gradient (the column sums of dx) comes out of the same launch (round 6, ``pd_layernorm_bwd_args.dxsum``) instead of a
        ``pd_channel_sum`` pass over dx; returns (dx, whether that happened)."""
        gamma, _, eps = ln
        B, h, w, ch = x.shape
        rows = B * h * w
        dx = self._tmp((B, h, w, ch), tag)
        partial = None
        dgam, dbet = self._G2(pname + ".weight", pname + ".bias")
        fuse = bias_name is not None and dgam is not None and ch <= 1536 and _BIAS_FUSE and bias_name not in self.frozen
        dxsum = self._G(bias_name) if fuse else None
        if dgam is not None:
            partial = self._tmp((self.lib.pd_layernorm_bwd_blocks(rows) * (3 if fuse else 2) * ch,), "lnpart3" if fuse else "lnpart", torch.float32)
        a = L.LayerNormBwdArgs(dtype=self.code, rows=rows, C=ch, eps=eps, x=x.data_ptr(), dy=dy.data_ptr(), gamma=gamma.data_ptr(),
                               res=L.ptr(res), dx=dx.data_ptr(), dgamma=L.ptr(dgam), dbeta=L.ptr(dbet), partial=L.ptr(partial), dxsum=L.ptr(dxsum))
        self._b(self.lib.pd_layernorm_bwd, a, "layernorm_bwd", 0.0, (3 + (res is not None)) * x.numel() * self._esz())
        return dx, fuse

    def _attn_bwd64(self, q, qs, k, v, kvs, o, do, lse, heads, nq, nkv, dq, dqs, dk, dv, dkvs):
        delta = self._tmp((self.B, heads, nq), "delta64", torch.float32)
        a = L.AttnD64BwdArgs(dtype=self.code, B=self.B, heads=heads, Nq=nq, Nkv=nkv, q=q, q_stride=qs, k=k, v=v, kv_stride=kvs,
                             o=o.data_ptr(), dout=do.data_ptr(), o_stride=heads * 64, lse=lse.data_ptr(), delta=delta.data_ptr(),
                             dq=dq, dq_stride=dqs, dk=dk, dv=dv, dkv_stride=dkvs)
        self._b(self.lib.pd_attn_d64_bwd, a, "attn_d64_bwd", 10.0 * self.B * heads * nq * nkv * 64,
                (4.0 * nq + 4.0 * nkv) * self.B * heads * 64 * self._esz())

    def _transformer_bwd(self, rec):
        e, te, n = rec.e, self.tw.transformers[rec.name], rec.name
        blk = n + ".transformer_blocks.0"
        G = self._G
        B, h, w, ch = rec.x.shape
        N, esz, T = h * w, self._esz(), self.tokens
        lin_w = lambda x, dy, wname, gn=None: self._wgrad(x, None, gn, 0, dy, G(*wname) if isinstance(wname, tuple) else G(wname),
                                                          ksize=1, pad=0)
        dout = self._g(rec.out)[0]
        # proj_out (+ residual x: folded into the GroupNorm backward at the end)
        self._bias_grad(dout, G(n + ".proj_out.bias"))
        lin_w(rec.h3, dout, n + ".proj_out.weight")
        dh3 = self._dgrad(dout, te.w_out_d, ch, ksize=1, tag="t_dh3")
        # feed-forward: h3 = ff2(geglu(ff1(LN3(h2)))) + h2
        self._bias_grad(dh3, G(blk + ".ff.net.2.bias"))
        lin_w(rec.gg, dh3, blk + ".ff.net.2.weight")
        dgg = self._dgrad(dh3, te.wff2_d, 4 * ch, ksize=1, tag="t_dgg")
        dff = self._tmp((B, h, w, 8 * ch), "t_dff")
        ga = L.GegluBwdArgs(dtype=self.code, rows=B * N, inner=4 * ch, x=rec.ff.data_ptr(), dy=dgg.data_ptr(), dx=dff.data_ptr())
        if _BIAS_FUSE and self.param_grads and (4 * ch) % 256 == 0 and (blk + ".ff.net.0.proj.bias") not in self.frozen:
            # the gate's backward also leaves the per-split column sums of dff (the widest gradient of the block): the bias gradient folds
            # those (B x splits x 8 ch floats) instead of reading dff back
            gs = max(1, min(64, N // 64))
            gws = torch.empty((B * gs * 8 * ch,), dtype=torch.float32, device=self.device)      # this block's own (its fold may run on the second stream)
            self.bufs.append(gws)
            ga.sums, ga.sum_splits, ga.B = gws.data_ptr(), gs, B
            self._fused_sums[id(dff)] = (gws, gs, True)
        self._b(self.lib.pd_geglu_bwd, ga, "geglu_bwd", 0.0, 5.0 * dgg.numel() * esz)
        self._bias_grad(dff, G(blk + ".ff.net.0.proj.bias"))
        self._fused_sums.pop(id(dff), None)               # (dff is a scratch buffer other blocks reuse)
        lin_w(rec.y3, dff, blk + ".ff.net.0.proj.weight")
        dy3 = self._dgrad(dff, te.wff1_d, ch, ksize=1, tag="t_dy")
        dh2, fb = self._ln_bwd(rec.h2, e.ln3, blk + ".norm3", dy3, dh3, "t_dh2", bias_name=blk + ".attn2.to_out.0.bias")
        # cross attention: h2 = to_out(attn(to_q(LN2(h1)), to_k(ehs), to_v(ehs))) + h1
        if not fb:
            self._bias_grad(dh2, G(blk + ".attn2.to_out.0.bias"))
        lin_w(rec.a2, dh2, blk + ".attn2.to_out.0.weight")
        da2 = self._dgrad(dh2, te.wo2_d, ch, ksize=1, tag="t_da")
        dq2 = self._tmp((B, h, w, ch), "t_dq2")
        dkv = self._tmp((B, 1, T, 2 * ch), "t_dkv")
        kvp = rec.kv.data_ptr()
        self._attn_bwd64(rec.q2.data_ptr(), ch, kvp, kvp + ch * esz, 2 * ch, rec.a2, da2, rec.lse2, e.heads, N, T,
                         dq2.data_ptr(), ch, dkv.data_ptr(), dkv.data_ptr() + ch * esz, 2 * ch)
        lin_w(rec.y2, dq2, blk + ".attn2.to_q.weight")
        lin_w(self.ehs, dkv, (blk + ".attn2.to_k.weight", (blk + ".attn2.to_v.weight",)))       # fused [2C][D] gradient
        if self._want_ehs_grad():
            self._dgrad(dkv, te.wkv2_d, self.ehs.shape[3], ksize=1, into=self._dehs)
        dy2 = self._dgrad(dq2, te.wq2_d, ch, ksize=1, tag="t_dy")
        dh1, fb = self._ln_bwd(rec.h1, e.ln2, blk + ".norm2", dy2, dh2, "t_dh1", bias_name=blk + ".attn1.to_out.0.bias")
        # self attention: h1 = to_out(attn(qkv(LN1(h0)))) + h0
        if not fb:
            self._bias_grad(dh1, G(blk + ".attn1.to_out.0.bias"))
        lin_w(rec.a1, dh1, blk + ".attn1.to_out.0.weight")
        da1 = self._dgrad(dh1, te.wo1_d, ch, ksize=1, tag="t_da")
        dqkv = self._tmp((B, h, w, 3 * ch), "t_dqkv")
        p, dp = rec.qkv.data_ptr(), dqkv.data_ptr()
        self._attn_bwd64(p, 3 * ch, p + ch * esz, p + 2 * ch * esz, 3 * ch, rec.a1, da1, rec.lse1, e.heads, N, N,
                         dp, 3 * ch, dp + ch * esz, dp + 2 * ch * esz, 3 * ch)
        lin_w(rec.y1, dqkv, (blk + ".attn1.to_q.weight", (blk + ".attn1.to_k.weight", blk + ".attn1.to_v.weight")))
        dy1 = self._dgrad(dqkv, te.wqkv1_d, ch, ksize=1, tag="t_dy")
        dh0, fb = self._ln_bwd(rec.h0, e.ln1, blk + ".norm1", dy1, dh1, "t_dh0", bias_name=n + ".proj_in.bias")
        # proj_in over GroupNorm(x) (no SiLU)
        if not fb:
            self._bias_grad(dh0, G(n + ".proj_in.bias"))
        lin_w(rec.x, dh0, n + ".proj_in.weight", gn=rec.gn)
        dz = self._dgrad(dh0, te.w_in_d, ch, ksize=1, tag="t_dz")
        self._gn_bwd(rec.gn, dz, 0, res=dout, wname=n + ".norm")

    def _want_ehs_grad(self):
        return self.param_grads and EMB_NAME in self.grads and EMB_NAME not in self.frozen


class _SDRepacker:
    """After an optimizer step: fp32 master parameters -> every kernel-layout copy the plans read, IN PLACE
    (``pd_pack_weight`` launches + a few small fp32 copies).  fp32 vectors the kernels read directly (norm affines, biases)
    alias the flat master buffer."""

    def __init__(self, m: SDUNet2DConditionModel, w: _SDPackedWeights, tw: SDTrainWeights):
        self.lib = L.lib()
        self.jobs, self.small, self.pre = [], [], []
        self.jobs_device = m.conv_in.weight.device
        self.w = w
        code = w.code

        def job(dst, src, cout, cin, k, *, dgrad=0, cout_pad=None, cin_pad=None, ct_stride=None, dst_off=0):
            cp = cout_pad or ((cout + 31) // 32) * 32
            ip = cin_pad or ((cin + 31) // 32) * 32
            per_ct = (ip // 32) * k * k * 2 * 64 * 8
            self.jobs.append(L.PackWeightArgs(dtype=code, cout=cout, cin=cin, cout_pad=cp, cin_pad=ip, ksize=k,
                                              src_in=(cout if dgrad else cin), dgrad=dgrad, src=src.data_ptr(),
                                              dst=dst.data_ptr() + dst_off * dst.element_size(), dst_ct_stride=ct_stride or per_ct))

        def both(dst, dstd, weight, cout, cin, k=1):
            job(dst, weight, cout, cin, k)
            job(dstd, weight, cin, cout, k, dgrad=1)

        ci, c0 = m.conv_in.weight.shape[1], m.conv_in.weight.shape[0]
        job(w.conv_in_w, m.conv_in.weight, c0, ci, 3, cin_pad=32)
        job(tw.conv_in_d, m.conv_in.weight, ci, c0, 3, dgrad=1, cout_pad=32)
        for name, mod in m.named_modules():
            if isinstance(mod, _Resnet):
                e, t = w.resnets[name], tw.resnets[name]
                cin, cout = mod.in_channels, mod.out_channels
                job(e.w1, mod.conv1.weight, cout, cin, 3)
                stride = e.w2[0].numel()
                job(e.w2, mod.conv2.weight, cout, cout, 3, ct_stride=stride)
                job(t.w1d, mod.conv1.weight, cin, cout, 3, dgrad=1)
                job(t.w2d, mod.conv2.weight, cout, cout, 3, dgrad=1)
                if mod.conv_shortcut is not None:
                    job(e.w2, mod.conv_shortcut.weight, cout, cin, 1, ct_stride=stride, dst_off=(cout // 32) * 9 * 2 * 512)
                    job(t.wsd, mod.conv_shortcut.weight, cin, cout, 1, dgrad=1)
                    b2, bs, dst = mod.conv2.bias, mod.conv_shortcut.bias, e.b2
                    self.small.append(lambda b2=b2, bs=bs, dst=dst: torch.add(b2.data, bs.data, out=dst))
            elif isinstance(mod, _Transformer2D):
                e, t = w.transformers[name], tw.transformers[name]
                blk = mod.transformer_blocks[0]
                a1, a2 = blk.attn1, blk.attn2
                ch, D = e.ch, a2.to_k.weight.shape[1]
                if not (_contiguous_after(a1.to_q.weight.data, a1.to_k.weight.data) and _contiguous_after(a1.to_k.weight.data, a1.to_v.weight.data)
                        and _contiguous_after(a2.to_k.weight.data, a2.to_v.weight.data)):
                    raise ValueError("fused projection weights must be adjacent (use sd_training_param_order)")
                both(e.w_in, t.w_in_d, mod.proj_in.weight, ch, ch)
                both(e.w_out, t.w_out_d, mod.proj_out.weight, ch, ch)
                both(e.wqkv1, t.wqkv1_d, a1.to_q.weight, 3 * ch, ch)
                both(e.wo1, t.wo1_d, a1.to_out[0].weight, ch, ch)
                both(e.wq2, t.wq2_d, a2.to_q.weight, ch, ch)
                both(e.wkv2, t.wkv2_d, a2.to_k.weight, 2 * ch, D)
                both(e.wo2, t.wo2_d, a2.to_out[0].weight, ch, ch)
                both(e.wff1, t.wff1_d, blk.ff.net[0].proj.weight, 8 * ch, ch)
                wf = blk.ff.net[0].proj.weight                       # inference copy: value / gate tiles interleaved
                tile = (ch // 32) * 2 * 512
                job(e.wff1_glu, wf.data[:4 * ch], 4 * ch, ch, 1, ct_stride=2 * tile)
                job(e.wff1_glu, wf.data[4 * ch:], 4 * ch, ch, 1, ct_stride=2 * tile, dst_off=tile)
                both(e.wff2, t.wff2_d, blk.ff.net[2].weight, ch, 4 * ch)
            elif isinstance(mod, _Sampler):
                ch = mod.conv.weight.shape[0]
                both(w.samplers[name].w, tw.samplers[name].wd, mod.conv.weight, ch, ch, 3)
                if getattr(w.samplers[name], "w4_src", None) is not None:      # the inference plans' sub-pixel phase kernels follow the weights too
                    src4, wt = w.samplers[name].w4_src, mod.conv.weight
                    self.pre.append(lambda src4=src4, wt=wt: upsample_phase_weights_stacked(wt.data, out=src4))
                    for p in range(4):
                        job(w.samplers[name].w4[p], src4[p], ch, ch, 2)
                        job(tw.samplers[name].wd4[p], src4[p], ch, ch, 2, dgrad=1)
        co, cc = m.conv_out.weight.shape[0], m.conv_out.weight.shape[1]
        job(w.conv_out_w, m.conv_out.weight, co, cc, 3, cout_pad=w.conv_out_pad)
        job(tw.conv_out_d, m.conv_out.weight, cc, co, 3, dgrad=1, cin_pad=w.conv_out_pad)
        te = m.time_embedding
        res = [mod for _, mod in m.named_modules() if isinstance(mod, _Resnet)]
        pd_, tdim = w.proj_dim, m.time_embed_dim
        first = res[0].time_emb_proj
        self.small += [
            lambda: w.w1T.copy_(te.linear_1.weight.data.t()),
            lambda: w.w2T.copy_(te.linear_2.weight.data.t()),
            lambda: w.wpT.copy_(torch.as_strided(first.weight.data, (pd_, tdim), (tdim, 1)).t()),
            lambda: w.bp.copy_(torch.as_strided(first.bias.data, (pd_,), (1,))),
            lambda: w.conv_out_b[:co].copy_(m.conv_out.bias.data),
        ]
        tf0 = next(iter(w.transformers.values()))
        for a, b in ((w.b1, te.linear_1.bias), (w.conv_in_b, m.conv_in.bias), (w.gn_out[0], m.conv_norm_out.weight),
                     (tf0.ln1[0], next(mod for mod in m.modules() if isinstance(mod, _Transformer2D)).transformer_blocks[0].norm1.weight)):
            if a.data_ptr() != b.data_ptr():
                raise RuntimeError("kernel-side fp32 vectors must alias the master parameters (build the packed weights "
                                   "after the parameters were moved into the flat training buffer)")

    def run(self, stream):
        with torch.no_grad():
            for f in self.pre:
                f()
        run_pack_jobs(self.lib, self.jobs, stream, self.__dict__.setdefault("_batch", {}), self.jobs_device)
        with torch.no_grad():
            for f in self.small:
                f()
        self.w.version = getattr(self.w, "version", 0) + 1      # inference plans drop what they cached of the old weights (cross-attention k / v)


class SDUNetTrainer(UNetTrainer):
    """One optimisation step of ``perform_training_epoch`` for ``model_type == "StableDiffusion"``
    (``utils_training.py:244-454``; components_to_train = denoiser [+ class_embedding], ``train.py:189-199``): latents + noise +
    timesteps -> ``_SD_prediction_wrapper`` forward -> loss -> backward (+ overlapped RCCL gradient all-reduce) -> clip + AdamW
    + EMA -> re-pack.  Call ``step(noisy_latents, timesteps, clean_latents, noise, class_labels, unconditional=False)``."""

    def __init__(self, model: SDUNet2DConditionModel, class_embedding: CustomEmbedding, scheduler, lr: float, *, device=None,
                 train_class_embedding: bool = True, use_ema: bool = True, max_grad_norm: Optional[float] = 1.0, group=None,
                 trainable=None, **adamw):
        """``trainable``: as for :class:`UNetTrainer` (names carry ``EMB_NAME`` for the class table); default: the ``requires_grad``
        flags of the UNet's parameters decide for the UNet (``components_to_train`` / ``--attention_fine_tuning``, train.py:189-220),
        ``train_class_embedding`` and the table's own flag for the ``CustomEmbedding``."""
        from .training import DiffusionLoss, FlatAdamWEMA, broadcast_from_rank0_, resolve_trainable
        self.model, self.class_embedding, self.scheduler = model, class_embedding, scheduler
        dev = device or model.device
        if torch.device(dev).type != "cuda":
            raise L.PhenDiffHipError("phendiff_amd trains on MI355X only (no CPU fallback): move the model to 'cuda'")
        order = sd_training_param_order(model)
        if trainable is None:
            flags = resolve_trainable(order, model)
            if train_class_embedding:
                flags = flags + resolve_trainable([(EMB_NAME, class_embedding.inner_module.weight)], class_embedding)
        if train_class_embedding:
            order = order + [(EMB_NAME, class_embedding.inner_module.weight)]
        if trainable is not None:
            flags = resolve_trainable(order, None, trainable)
        if not any(flags):
            raise ValueError("SDUNetTrainer: no trainable parameter (every parameter is frozen)")
        self.frozen = frozenset(n for (n, _), f in zip(order, flags) if not f)
        self.opt = FlatAdamWEMA([p for _, p in order], lr, use_ema=use_ema, max_grad_norm=max_grad_norm, **adamw)
        self.opt.set_trainable(flags)
        if train_class_embedding and flags[-1]:
            self.opt.set_tail(class_embedding.inner_module.weight.numel(), (EMB_NAME,))
        # DDP's wrap-time broadcast (train.py:311-326): rank 0's parameters everywhere; the EMA shadow starts from them
        broadcast_from_rank0_(self.opt.flat, group)
        if self.opt.ema is not None:
            self.opt.ema.copy_(self.opt.flat)
        self.params = {n: p.data for n, p in order}
        self.grads = {n: p.grad for n, p in order}
        model.invalidate()
        self.loss_fn = DiffusionLoss(scheduler, dev)
        if getattr(model, "compute_dtype", None) == "fp16":      # --mixed_precision fp16: accelerate's GradScaler (training.LossScaler)
            from .training import LossScaler
            self.opt.scaler = LossScaler()
        self.device = dev
        self._plans = {}
        self._tw = None
        self._repack = None
        self._uncond = False

    def checkpoint_modules(self):
        """accelerate's numbering of the prepared models (train.py:318-326): unet 0, vae 1 (frozen, not the trainer's), class
        embedding 2 -- flat names of the embedding carry the ``class_embedding.`` prefix."""
        mods = [(0, self.model, "")]
        if EMB_NAME in self.params:
            mods.append((2, self.class_embedding, "class_embedding."))
        return mods

    def _optimizer_step(self, lr):
        # an unconditional step leaves the CustomEmbedding without a gradient: torch's AdamW skips it (EMA still steps)
        self.opt.step(lr, tail_active=not self._uncond)

    def _make_packed(self):
        return _SDPackedWeights(self.model, self.device)

    def _make_train_weights(self):
        return SDTrainWeights(self.model, self.device, self.model._weights.tdt)

    def _make_repacker(self):
        return _SDRepacker(self.model, self.model._weights, self._tw)

    def _make_plan(self, key):
        m = self.model
        return SDUNetTrainPlan(m, m._weights, self._tw, *key, self.device, self.params, self.grads, frozen=self.frozen)

    def plan_for(self, B, H, W, tokens=77):
        self._bind_weights()
        key = (B, H, W, tokens)
        p = self._plans.get(key)
        if p is None:
            p = self._make_plan(key)
            self._plans[key] = p
        return p

    def encoder_hidden_states(self, class_labels, unconditional: bool):
        """``_SD_prediction_wrapper`` (utils_training.py:470-484): zeros(B, 77, D) on an unconditional step, else the class
        embedding as token 0 followed by 76 zero tokens."""
        table = self.class_embedding.inner_module.weight
        if unconditional:
            return torch.zeros((class_labels.shape[0], 77, table.shape[1]), dtype=torch.float32, device=self.device)
        return class_emb_to_encoder_hidden_states(table.data[class_labels.to(device=self.device, dtype=torch.int64)])

    def forward_backward(self, noisy, timesteps, clean, noise, class_labels=None, class_emb=None, after_op=None, unconditional=None):
        if unconditional is None:
            unconditional = getattr(self, "_uncond", False)
        B, _, H, W = noisy.shape
        plan = self.plan_for(B, H, W)
        st = torch.cuda.current_stream(self.device).cuda_stream
        x = noisy.contiguous().float()
        ts = timesteps.to(device=self.device, dtype=torch.float32).contiguous()
        labels = class_labels.to(device=self.device, dtype=torch.int64).contiguous()
        ehs = self.encoder_hidden_states(labels, unconditional)
        out = torch.empty_like(x)
        plan.forward(x, ts, ehs, out, st, labels=None if unconditional else labels)
        loss, dout = self.loss_fn(out, clean, noise, timesteps, grad_scale=self.opt.scaler.scale if self.opt.scaler is not None else 1.0)
        plan.backward(dout, st, after_op=after_op)
        return loss, out

    def step(self, noisy, timesteps, clean, noise, class_labels, unconditional: bool = False, lr: Optional[float] = None,
             group=None, overlap: bool = True, bucket_bytes: int = 64 << 20):
        """3.46 GB of fp32 gradients per step: 64 MB buckets (xGMI rings are per-link bound: few, large messages)."""
        self._uncond = bool(unconditional)
        return super().step(noisy, timesteps, clean, noise, class_labels=class_labels, class_emb=None, lr=lr, group=group,
                            overlap=overlap, bucket_bytes=bucket_bytes)
